/* Test-only stand-in for librccl (CPU tier): the five entry points tetraear_amd/rccl.py binds, over a Unix-domain
 * socket between the ranks of ONE node and HOST memory.  It exists so that the id hand-off, the collective bring-up
 * decision and the all-reduce plumbing of RcclGroup execute in the CPU test tier with world size 2
 * (tests/test_rccl_stub.py); it is never loaded by the product.                                                     */
#define _GNU_SOURCE
#include <errno.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <sys/socket.h>
#include <sys/un.h>
#include <time.h>
#include <unistd.h>

typedef struct { char internal[128]; } ncclUniqueId;
typedef struct stub_comm {
    int rank, nranks;
    int fds[64]; /* rank 0: one per peer (index = peer rank); others: fds[0] = link to rank 0 */
} stub_comm;
typedef stub_comm *ncclComm_t;

static int rd(int fd, void *p, size_t n)
{
    char *c = (char *)p;
    while (n) {
        ssize_t k = read(fd, c, n);
        if (k <= 0) return -1;
        c += k;
        n -= (size_t)k;
    }
    return 0;
}
static int wr(int fd, const void *p, size_t n)
{
    const char *c = (const char *)p;
    while (n) {
        ssize_t k = write(fd, c, n);
        if (k <= 0) return -1;
        c += k;
        n -= (size_t)k;
    }
    return 0;
}

const char *ncclGetErrorString(int rc) { return rc == 0 ? "no error" : "rccl stub: failure"; }

int ncclGetUniqueId(ncclUniqueId *id)
{
    memset(id, 0, sizeof(*id));
    /* abstract socket name unique to this call */
    snprintf(id->internal, sizeof(id->internal), "tdm_rccl_stub_%d_%ld", (int)getpid(), (long)time(NULL));
    return 0;
}

static void addr_of(const ncclUniqueId *id, struct sockaddr_un *a, socklen_t *len)
{
    memset(a, 0, sizeof(*a));
    a->sun_family = AF_UNIX;
    a->sun_path[0] = 0; /* abstract namespace */
    strncpy(a->sun_path + 1, id->internal, sizeof(a->sun_path) - 2);
    *len = (socklen_t)(sizeof(a->sun_family) + 1 + strlen(id->internal));
}

int ncclCommInitRank(ncclComm_t *comm, int nranks, ncclUniqueId id, int rank)
{
    if (nranks < 1 || nranks > 64 || rank < 0 || rank >= nranks) return 4;
    stub_comm *c = (stub_comm *)calloc(1, sizeof(stub_comm));
    c->rank = rank;
    c->nranks = nranks;
    struct sockaddr_un a;
    socklen_t alen;
    addr_of(&id, &a, &alen);
    if (nranks > 1 && rank == 0) {
        int s = socket(AF_UNIX, SOCK_STREAM, 0);
        if (s < 0 || bind(s, (struct sockaddr *)&a, alen) != 0 || listen(s, nranks) != 0) return 2;
        for (int i = 1; i < nranks; ++i) {
            int fd = accept(s, NULL, NULL);
            int32_t r = -1;
            if (fd < 0 || rd(fd, &r, 4) != 0 || r <= 0 || r >= nranks) return 2;
            c->fds[r] = fd;
        }
        close(s);
    } else if (nranks > 1) {
        int fd = -1;
        for (int tries = 0; tries < 3000; ++tries) { /* rank 0 may not be listening yet */
            fd = socket(AF_UNIX, SOCK_STREAM, 0);
            if (connect(fd, (struct sockaddr *)&a, alen) == 0) break;
            close(fd);
            fd = -1;
            usleep(10000);
        }
        int32_t r = rank;
        if (fd < 0 || wr(fd, &r, 4) != 0) return 2;
        c->fds[0] = fd;
    }
    *comm = c;
    return 0;
}

/* dtype 8 = float64, 4 = int64; op 0 = sum, 2 = max (the two combinations rccl.py uses) */
int ncclAllReduce(const void *send, void *recv, size_t count, int dtype, int op, ncclComm_t c, void *stream)
{
    (void)stream;
    if (count != 1 || !((dtype == 8 && op == 2) || (dtype == 4 && op == 0))) return 5;
    uint64_t acc;
    memcpy(&acc, send, 8);
    if (c->nranks > 1 && c->rank == 0) {
        for (int r = 1; r < c->nranks; ++r) {
            uint64_t v;
            if (rd(c->fds[r], &v, 8) != 0) return 2;
            if (dtype == 8) {
                double a, b;
                memcpy(&a, &acc, 8);
                memcpy(&b, &v, 8);
                if (b > a) a = b;
                memcpy(&acc, &a, 8);
            } else {
                int64_t a, b;
                memcpy(&a, &acc, 8);
                memcpy(&b, &v, 8);
                a += b;
                memcpy(&acc, &a, 8);
            }
        }
        for (int r = 1; r < c->nranks; ++r)
            if (wr(c->fds[r], &acc, 8) != 0) return 2;
    } else if (c->nranks > 1) {
        if (wr(c->fds[0], &acc, 8) != 0 || rd(c->fds[0], &acc, 8) != 0) return 2;
    }
    memcpy(recv, &acc, 8);
    return 0;
}

int ncclCommDestroy(ncclComm_t c)
{
    if (!c) return 0;
    for (int i = 0; i < 64; ++i)
        if (c->fds[i] > 0) close(c->fds[i]);
    free(c);
    return 0;
}
