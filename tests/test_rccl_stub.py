"""CPU tier, world size 2: tetraear_amd.rccl.RcclGroup against a stub librccl (tests/rccl_stub) over host memory.
What executes here is what a multi-GPU bench run depends on before its first kernel: the TCP rendezvous on
MASTER_ADDR, the collective "is librccl usable on every rank" decision, the ncclUniqueId hand-off,
ncclCommInitRank, and the barrier / max / sum all-reduces."""
import ctypes as C
import multiprocessing as mp
import os
import socket
import subprocess

import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
STUB = os.path.join(HERE, "rccl_stub", "librccl_stub.so")


@pytest.fixture(scope="module", autouse=True)
def _build_stub():
    subprocess.run(["make", "-C", os.path.join(HERE, "rccl_stub")], check=True, capture_output=True)


class HostMemory:
    """the exchange buffer in host memory (the stub reduces host pointers)"""

    def __init__(self):
        self.buf = (C.c_byte * 16)()
        self.ptr = C.cast(self.buf, C.c_void_p)

    def upload(self, v):
        C.memmove(self.buf, C.byref(v), 8)

    def download(self, v):
        C.memmove(C.byref(v), self.buf, 8)

    def free(self):
        pass


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _rank(rank, world, port, lib_path, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), TORCHELASTIC_RUN_ID="stubtest")
    from tetraear_amd.rccl import RcclGroup, RcclUnavailable
    from tetraear_amd.shard import reduce_job
    try:
        g = RcclGroup(rank, world, 0, timeout_s=30.0, lib_path=lib_path, memory=HostMemory())
    except RcclUnavailable as e:
        q.put((rank, "unavailable", str(e)))
        return
    g.barrier()
    t, s, bad = reduce_job(g, 0.25 * (rank + 1), 1000 + rank, n_failed=1 if rank == 1 else 0)
    mx = g.max_f64(-1.5 - rank)
    g.close()
    q.put((rank, "ok", (t, s, bad, mx)))


def _run(world, port, lib_paths):
    ctx = mp.get_context("fork")
    q = ctx.Queue()
    procs = [ctx.Process(target=_rank, args=(r, world, port, lib_paths[r], q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=60) for _ in procs)
    for p in procs:
        p.join(30)
        assert p.exitcode == 0
    return res


@pytest.mark.timeout(120)
def test_two_ranks_rendezvous_and_allreduce():
    res = _run(2, _free_port(), [STUB, STUB])
    for rank, status, val in res:
        assert status == "ok", val
        t, s, bad, mx = val
        assert t == 0.5 and s == 2001 and bad == 1 and mx == -1.5   # max, sum, sum (the failure flag of rank 1 reaches rank 0), max


@pytest.mark.timeout(120)
def test_three_ranks():
    res = _run(3, _free_port(), [STUB] * 3)
    assert [r[1] for r in res] == ["ok"] * 3 and all(r[2][1] == 3003 for r in res)


@pytest.mark.timeout(120)
def test_fallback_decision_is_collective():
    """one rank cannot load the library: EVERY rank raises RcclUnavailable (nobody is left inside ncclCommInitRank)"""
    for broken in (0, 1):
        libs = [STUB, STUB]
        libs[broken] = "/nonexistent/librccl.so"
        res = _run(2, _free_port(), libs)
        assert [r[1] for r in res] == ["unavailable", "unavailable"], res


@pytest.mark.timeout(120)
def test_foreign_listener_on_the_first_candidate_port_is_skipped():
    """something else listens on MASTER_PORT + 1 and never speaks the protocol: rank 0 binds the next port, the others
    recognise the stranger by the missing acknowledgement and move on"""
    port = _free_port()
    stranger = socket.socket()
    stranger.setsockopt(socket.SOL_SOCKET, socket.SO_REUSEADDR, 1)
    try:
        stranger.bind(("127.0.0.1", port + 1))
    except OSError:
        pytest.skip("port taken")
    stranger.listen(4)
    try:
        res = _run(2, port, [STUB, STUB])
        assert [r[1] for r in res] == ["ok", "ok"]
    finally:
        stranger.close()
