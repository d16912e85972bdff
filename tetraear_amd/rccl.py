"""RCCL through ctypes: the only collective traffic of this project is a barrier and two 8-byte all-reduces around the
timed region of a multi-GPU run (carriers are sharded, SURVEY.md section 8(e): no data-path collective), so the
launcher's ranks talk to librccl directly instead of importing a tensor framework.

One process per GPU on ONE node.  The ncclUniqueId travels from rank 0 to the others through a file in /tmp named
after MASTER_PORT, the launcher's run id and the launcher's process id (all ranks share the node's /tmp and their parent)."""
import contextlib
import ctypes as C
import os
import sys
import time

from . import _lib

NCCL_UNIQUE_ID_BYTES = 128
ncclFloat64, ncclInt64 = 8, 4
ncclSum, ncclMax = 0, 2


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


class RcclGroup:
    def __init__(self, rank, world, device, tag=None, timeout_s=120.0):
        self.rank, self.world, self.device = int(rank), int(world), int(device)
        self.lib = C.CDLL("librccl.so")
        self.lib.ncclGetErrorString.restype = C.c_char_p
        self.tdm = _lib.load()
        _lib.check(self.tdm.tdm_dev_sync(self.device))          # binds this process to its device (hipSetDevice)
        # (the launcher's pid is the same for all ranks of one launch and differs between launches: an id file left behind
        #  by a crashed earlier run on the same port can never be taken for this run's)
        tag = tag or f"{os.environ.get('MASTER_PORT', '0')}_{os.environ.get('TORCHELASTIC_RUN_ID', 'run')}_{os.getppid()}"
        path = f"/tmp/tdm_rccl_{tag}_{self.world}.id"
        uid = _UniqueId()
        if self.rank == 0:
            with _stdout_to_stderr():
                self._ok(self.lib.ncclGetUniqueId(C.byref(uid)))
            tmp = path + f".{os.getpid()}"
            with open(tmp, "wb") as f:
                f.write(bytes(uid.internal))
            os.replace(tmp, path)
        else:
            t0 = time.time()
            while not (os.path.exists(path) and os.path.getsize(path) == NCCL_UNIQUE_ID_BYTES):
                if time.time() - t0 > timeout_s:
                    raise TimeoutError(f"no ncclUniqueId at {path}")
                time.sleep(0.01)
            with open(path, "rb") as f:
                C.memmove(uid.internal, f.read(), NCCL_UNIQUE_ID_BYTES)
        self.comm = C.c_void_p()
        with _stdout_to_stderr():
            self._ok(self.lib.ncclCommInitRank(C.byref(self.comm), self.world, uid, self.rank))
        self._path = path
        p = C.c_void_p()
        _lib.check(self.tdm.tdm_dev_alloc(self.device, 16, C.byref(p)))
        self.buf = p

    def _ok(self, rc):
        if rc != 0:
            raise RuntimeError(f"rccl: {self.lib.ncclGetErrorString(rc).decode()}")

    def _allreduce(self, ctype, nccl_type, op, value):
        v = ctype(value)
        _lib.check(self.tdm.tdm_dev_upload(self.device, self.buf, C.byref(v), 8))
        self._ok(self.lib.ncclAllReduce(self.buf, self.buf, C.c_size_t(1), nccl_type, op, self.comm, C.c_void_p(0)))
        _lib.check(self.tdm.tdm_dev_sync(self.device))
        _lib.check(self.tdm.tdm_dev_download(self.device, C.byref(v), self.buf, 8))
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
            self.tdm.tdm_dev_free(self.device, self.buf)
            if self.rank == 0 and os.path.exists(self._path):
                os.remove(self._path)
        except Exception:
            pass
