"""RCCL through ctypes: the only collective traffic of this project is a barrier and two 8-byte all-reduces around the
timed region of a multi-GPU run (carriers are sharded, SURVEY.md section 8(e): no data-path collective), so the
launcher's ranks talk to librccl directly instead of importing a tensor framework.

One process per GPU on ONE node.  Bring-up is collective by construction:

1. every rank loads librccl and binds its device, and notes whether that worked;
2. the ranks meet on a TCP socket that rank 0 opens on MASTER_ADDR (ports MASTER_PORT + 1 + k, k = 0..63: the first one
   it can bind; MASTER_PORT itself belongs to the launcher's store).  A connection starts with a token made of the
   launcher's run id and pid (or TDM_RCCL_TOKEN, for launchers whose ranks do not share a parent), the master port and
   the world size, so a foreign listener or a stale run on one of those ports is recognised and skipped;
3. rank 0 collects every rank's "librccl usable" flag and answers each with the SAME decision: all usable -> the
   ncclUniqueId follows and all ranks enter ncclCommInitRank; otherwise every rank raises RcclUnavailable and the caller
   may choose another backend -- on ALL ranks, never on some (a per-rank fallback leaves the rest blocked inside
   ncclCommInitRank for ever).

After the decision there is no fallback: an error inside ncclCommInitRank / ncclAllReduce is fatal for the job.
"""
import contextlib
import ctypes as C
import os
import socket
import struct
import sys
import time

from . import _lib

NCCL_UNIQUE_ID_BYTES = 128
ncclFloat64, ncclInt64 = 8, 4
ncclSum, ncclMax = 0, 2
_PORT_SPAN = 64
_MAGIC = b"TDMRCCL3"


def _port_base(master_port):
    """first rendezvous port for a launch whose store listens on `master_port`: 1000 + above it, folded back under 65536 - span"""
    hi = 65536 - _PORT_SPAN
    p = master_port + 1000
    return p if p < hi else 1024 + (p - hi) % (hi - 1024)


class RcclUnavailable(RuntimeError):
    """raised on EVERY rank when at least one rank cannot use librccl (the decision is collective)"""


@contextlib.contextmanager
def _stdout_to_stderr():
    """librccl prints a version banner on the C stdout when a communicator is created; the bench's stdout carries
    exactly one JSON line, so the library's chatter is sent to stderr for the duration of the call."""
    libc = C.CDLL(None)
    sys.stdout.flush()
    saved = os.dup(1)
    os.dup2(2, 1)
    try:
        yield
    finally:
        libc.fflush(None)
        os.dup2(saved, 1)
        os.close(saved)


class _UniqueId(C.Structure):
    _fields_ = [("internal", C.c_byte * NCCL_UNIQUE_ID_BYTES)]


class DeviceMemory:
    """the 16-byte exchange buffer of the all-reduces in device memory, through libtetrahip's helpers"""

    def __init__(self, device):
        self.device = int(device)
        self.tdm = _lib.load()
        _lib.check(self.tdm.tdm_dev_sync(self.device))          # binds this process to its device (hipSetDevice)
        p = C.c_void_p()
        _lib.check(self.tdm.tdm_dev_alloc(self.device, 16, C.byref(p)))
        self.ptr = p

    def upload(self, cvalue):
        _lib.check(self.tdm.tdm_dev_upload(self.device, self.ptr, C.byref(cvalue), 8))

    def download(self, cvalue):
        _lib.check(self.tdm.tdm_dev_sync(self.device))
        _lib.check(self.tdm.tdm_dev_download(self.device, C.byref(cvalue), self.ptr, 8))

    def free(self):
        self.tdm.tdm_dev_free(self.device, self.ptr)


def _recv_exact(sock, n):
    buf = b""
    while len(buf) < n:
        part = sock.recv(n - len(buf))
        if not part:
            raise ConnectionError("rendezvous peer closed the connection")
        buf += part
    return buf


_CONN_TIMEOUT_S = 10.0   # rank 0's patience with one peer's handshake step


def _token(world):
    """what identifies this launch to its own ranks: TDM_RCCL_TOKEN if the launcher sets one; else the elastic launcher's
    run id plus the launcher's pid (torch.distributed.run: every rank is its child); the master port and world size always"""
    tag = os.environ.get("TDM_RCCL_TOKEN") or f"{os.environ.get('TORCHELASTIC_RUN_ID', 'run')}|{os.getppid()}"
    run = f"{tag}|{os.environ.get('MASTER_PORT', '')}|{world}".encode()
    return _MAGIC + struct.pack("<H", len(run)) + run


def rendezvous(rank, world, usable, make_payload, timeout_s=120.0, addr=None, base_port=None):
    """Collective decision + payload hand-off.  Every rank passes `usable`; rank 0 also passes make_payload() -> bytes
    (called only when every rank is usable).  Returns (all_usable, payload or b"")."""
    addr = addr or os.environ.get("MASTER_ADDR", "127.0.0.1")
    # the rendezvous ports lie well away from the launcher's own range: a second launcher on the node normally takes
    # MASTER_PORT + 1, + 2, ... as ITS master port, and a rank 0 of ours squatting there would stop that job from starting
    # (TDM_RCCL_PORT overrides the base)
    if base_port is None:
        env = os.environ.get("TDM_RCCL_PORT")
        base_port = int(env) if env else _port_base(int(os.environ.get("MASTER_PORT", "29500")))
    base = int(base_port)
    token = _token(world)
    deadline = time.time() + timeout_s
    if world == 1:
        return bool(usable), (make_payload() if usable else b"")
    if rank == 0:
        srv = None
        for k in range(_PORT_SPAN):
            s = socket.socket(socket.AF_INET, socket.SOCK_STREAM)
            s.setsockopt(socket.SOL_SOCKET, socket.SO_REUSEADDR, 1)
            try:
                s.bind((addr, base + k))
                s.listen(world)
                srv = s
                break
            except OSError:
                s.close()
        if srv is None:
            raise RuntimeError(f"rccl rendezvous: no free port in {base}..{base + _PORT_SPAN - 1} on {addr}")
        # peers[r]: the NEWEST connection of rank r; confirmed: the ranks that acknowledged the decision.  A rank whose
        # handshake broke (before or after our acknowledgement) simply connects again: the newer socket replaces the stale
        # one, and a decision that could not be delivered (send error, no acknowledgement) puts the rank back among the
        # awaited ones instead of leaving it to time out.
        # unacked: ranks whose decision was SENT without error but whose acknowledgement did not come within the socket
        # timeout (a peer that closed without acknowledging is different: it never read the decision).  They count as served -- the rank has most likely returned already and will never connect
        # again, so waiting for it would time the job out while the others hang in ncclCommInitRank -- but if such a rank does
        # come back before the last rank is served, it is served again.
        peers, flags, confirmed, unacked = {}, {0: bool(usable)}, set(), set()
        decision, payload = None, b""
        try:
            while len(confirmed) < world - 1:
                while len(peers) + len(confirmed) < world - 1:
                    srv.settimeout(max(0.1, deadline - time.time()))
                    try:
                        conn, _ = srv.accept()
                    except socket.timeout:
                        raise TimeoutError(f"rccl rendezvous: {world - 1 - len(peers) - len(confirmed)} rank(s) never arrived") from None
                    try:
                        conn.settimeout(_CONN_TIMEOUT_S)
                        if _recv_exact(conn, len(token)) != token:
                            conn.close()          # not one of this launch's ranks
                            continue
                        r, ok = struct.unpack("<iB", _recv_exact(conn, 5))
                        conn.sendall(_MAGIC)      # immediate acknowledgement: the peer knows it found this launch's rank 0
                    except (OSError, ConnectionError, struct.error):
                        conn.close()
                        continue
                    if 0 < r < world and (r not in confirmed or r in unacked):
                        if r in unacked:          # it did not get the decision after all
                            unacked.discard(r)
                            confirmed.discard(r)
                        if r in peers:
                            peers[r].close()
                        peers[r] = conn
                        flags.setdefault(r, bool(ok))
                    else:
                        conn.close()
                if decision is None:
                    decision = all(flags.values())
                    payload = make_payload() if decision else b""
                for r, conn in list(peers.items()):
                    try:
                        conn.sendall(struct.pack("<BI", 1 if decision else 0, len(payload)) + payload)
                    except (OSError, ConnectionError):
                        pass                      # not delivered: the rank connects again (or the deadline passes)
                    else:
                        try:
                            if _recv_exact(conn, 1) == b"K":
                                confirmed.add(r)
                        except socket.timeout:
                            confirmed.add(r)      # sent, the acknowledgement late: served unless it comes back (see above)
                            unacked.add(r)
                        except (OSError, ConnectionError):
                            pass                  # the peer closed without acknowledging: it never read the decision and connects again
                    conn.close()
                    del peers[r]
        finally:
            for conn in peers.values():
                conn.close()
            srv.close()
        return decision, payload
    # ranks > 0: find rank 0's socket among the candidate ports (a listener that is not ours never answers the token)
    last = None
    while time.time() < deadline:
        for k in range(_PORT_SPAN):
            try:
                s = socket.create_connection((addr, base + k), timeout=1.0)
            except OSError as e:
                last = e
                continue
            try:
                s.settimeout(3.0)
                s.sendall(token + struct.pack("<iB", rank, 1 if usable else 0))
                if _recv_exact(s, len(_MAGIC)) != _MAGIC:
                    continue                  # something else listens on this port
                s.settimeout(max(1.0, deadline - time.time()))   # the decision comes when every rank has arrived
                head = _recv_exact(s, 5)
                decision, n = struct.unpack("<BI", head)
                payload = _recv_exact(s, n) if n else b""
                s.sendall(b"K")               # delivered: rank 0 stops waiting for this rank
                return bool(decision), payload
            except (OSError, ConnectionError, struct.error) as e:
                last = e
            finally:
                s.close()
        time.sleep(0.05)
    raise TimeoutError(f"rccl rendezvous: rank {rank} found no rank 0 on {addr}:{base}..{base + _PORT_SPAN - 1} ({last})")


class RcclGroup:
    """barrier / max_f64 / sum_i64 over librccl.  `lib_path` and `memory` exist for the CPU-tier test, which runs this class
    against a stub library over host memory (tests/rccl_stub)."""

    def __init__(self, rank, world, device, timeout_s=120.0, lib_path=None, memory=None):
        self.rank, self.world, self.device = int(rank), int(world), int(device)
        # (TDM_RCCL_LIB: another library with the same five entry points -- the CPU tier's stub)
        lib_path = lib_path or os.environ.get("TDM_RCCL_LIB", "librccl.so")
        self.lib, self.mem, err = None, None, None
        try:
            self.lib = C.CDLL(lib_path)
            self.lib.ncclGetErrorString.restype = C.c_char_p
            self.mem = memory if memory is not None else DeviceMemory(self.device)
        except Exception as e:  # noqa: BLE001 -- reported to the other ranks, then raised on all of them
            err = e
        uid = _UniqueId()

        def make_id():
            with _stdout_to_stderr():
                self._ok(self.lib.ncclGetUniqueId(C.byref(uid)))
            return bytes(uid.internal)

        ok, payload = rendezvous(self.rank, self.world, err is None, make_id, timeout_s)
        if not ok:
            if self.mem is not None and memory is None:
                self.mem.free()
            raise RcclUnavailable(f"librccl is not usable on every rank (this rank: {err or 'ok'})")
        C.memmove(uid.internal, payload, NCCL_UNIQUE_ID_BYTES)
        self.comm = C.c_void_p()
        with _stdout_to_stderr():
            self._ok(self.lib.ncclCommInitRank(C.byref(self.comm), self.world, uid, self.rank))

    def _ok(self, rc):
        if rc != 0:
            raise RuntimeError(f"rccl: {self.lib.ncclGetErrorString(rc).decode()}")

    def _allreduce(self, ctype, nccl_type, op, value):
        v = ctype(value)
        self.mem.upload(v)
        self._ok(self.lib.ncclAllReduce(self.mem.ptr, self.mem.ptr, C.c_size_t(1), nccl_type, op, self.comm, C.c_void_p(0)))
        self.mem.download(v)
        return v.value

    def max_f64(self, x):
        return float(self._allreduce(C.c_double, ncclFloat64, ncclMax, float(x)))

    def sum_i64(self, x):
        return int(self._allreduce(C.c_int64, ncclInt64, ncclSum, int(x)))

    def barrier(self):
        self.sum_i64(1)

    def close(self):
        try:
            self.barrier()
            with _stdout_to_stderr():
                self.lib.ncclCommDestroy(self.comm)
            self.mem.free()
        except Exception:
            pass
