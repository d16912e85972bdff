#!/usr/bin/env python3
"""Benchmark of the hot path: IQ -> pi/4-DQPSK hard symbols (reference-parity mode).

    python bench.py [--gpus N] [--steps K] [--warmup W] [--carriers C] [--chunk S] [--fmt cu8]

One "step" = one pass of SignalProcessor.process over a batch of C independent 2.4 MS/s carriers
of S samples each per GPU (1024 carriers x 262144-sample chunks across
8 GPUs = 128 per GPU; weak scaling, carriers are sharded across ranks with no data-path
collective).  Inputs are resident in HBM before the timed region.  For N > 1 the driver launches
one process per GPU with torch.distributed.run; RCCL is used only for the barrier and the
max-over-ranks of the elapsed time.

Prints ONE JSON line (rank 0).
"""
import argparse
import json
import os
import sys
import time

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)

SAMPLE_RATE = 2.4e6
PEAK_FP64_TFLOPS = 78.6   # MI355X fp64 vector FMA peak (= fp64 matrix peak); MI355X_MICROARCH.md has 157.3 fp32
PEAK_HBM_GBS = 8000.0     # MI355X_MICROARCH.md: HBM3E 8 TB/s spec (6.3 TB/s achievable)
SETTLE_STEPS = 150   # hot-path passes (~0.14 s) after which the GPU's clocks have settled
# algorithmic work of the decimator stage, SURVEY.md 8(d): 4 biquads x 2 passes on complex data
FLOP_PER_INPUT_SAMPLE = 150.0
# what the parallel-form kernel executes on cu8 input: 5080 fp64 FMAs per lane of 120 input samples (recurrences 3840,
# two-tap outputs 384, start-state responses 384, lane scans 448, direct term 24), 2 flop each
EXECUTED_FLOP_PER_INPUT_SAMPLE = 5080 * 2 / 120.0
# the part of those no implementation of the filter can avoid: the recurrences themselves, 3840 FMAs per 120 samples
IRREDUCIBLE_FLOP_PER_INPUT_SAMPLE = 3840 * 2 / 120.0

_CEILING = None
PMC_CHILD = "--pmc-child" in sys.argv   # this process is the short run a parent bench counts HBM traffic on


def hbm_ceiling(device=0):
    """Measured HBM ceilings of THIS box (the roofline denominator is an on-box copy-kernel figure): the
    library's own copy / read / write kernels over 2 x 1 GiB, once per bench process (~0.3 s)."""
    global _CEILING
    if PMC_CHILD:
        return {"skipped": "pmc child"}
    if _CEILING is None:
        import ctypes as C
        from tetraear_amd import _lib
        g = (C.c_double * 3)()
        try:
            _lib.check(_lib.load().tdm_hbm_ceiling(device, 1 << 30, 10, g))
            _CEILING = {"copy": g[0], "read": g[1], "write": g[2], "unit": "GB/s",
                        "how": "tdm_hbm_ceiling: 16 B per lane over 1 GiB buffers, best of the flat form (one access per lane, one workgroup per 4 KB) and grid-stride forms (plain / non-temporal, 2048 / 8192 workgroups), 10 launches after 3, HIP events"}
        except Exception as e:  # noqa: BLE001
            _CEILING = {"error": str(e)}
    return _CEILING


def hbm_roofline(kernel, bytes_alg, ms, read_bytes=None, traffic=None, traffic_src=None, device=0, **extra):
    """roofline object of an HBM-bound kernel: algorithmic bytes over the measured launch time against the data-sheet
    peak (`frac`) and against the box's measured ceilings (`frac_of_measured`: the copy figure; `frac_of_mix`: the time
    the same read / write split would take at the measured read-only and write-only rates)."""
    ach = bytes_alg / (ms * 1e-3) / 1e9
    out = {"kernel": kernel, "bound": "hbm", "achieved": ach, "peak": PEAK_HBM_GBS, "unit": "GB/s", "frac": ach / PEAK_HBM_GBS,
           "traffic": traffic, "algorithmic_bytes_per_launch": bytes_alg, "avg_launch_ms": ms}
    if traffic_src:
        out["traffic_source"] = traffic_src
    if traffic:
        out["traffic_over_algorithmic"] = traffic / bytes_alg
    c = hbm_ceiling(device)
    if "copy" in c:
        out["peak_measured"] = c["copy"]
        out["frac_of_measured"] = ach / c["copy"]
        if read_bytes is not None:
            t_mix = read_bytes / c["read"] + (bytes_alg - read_bytes) / c["write"]   # (GB/s -> bytes / 1e9 per second)
            out["peak_measured_mix"] = bytes_alg / t_mix
            out["frac_of_mix"] = ach / (bytes_alg / t_mix)
    out.update(extra)
    return out


class TorchGroup:
    """barrier / max / sum on torch.distributed: the fallback of the multi-GPU bench when librccl cannot be used through
    ctypes on every rank, and the group of the gloo test (tests/test_dist_cpu.py).  Lives here, not in the package: the
    product path imports no tensor framework."""

    def __init__(self, dist, device=None):
        self.dist, self.device = dist, device

    def _reduce(self, value, dtype, op):
        import torch
        t = torch.tensor([value], dtype=dtype, device=self.device)
        self.dist.all_reduce(t, op=op)
        return t.item()

    def max_f64(self, x):
        import torch
        return float(self._reduce(float(x), torch.float64, self.dist.ReduceOp.MAX))

    def sum_i64(self, x):
        import torch
        return int(self._reduce(int(x), torch.int64, self.dist.ReduceOp.SUM))

    def barrier(self):
        self.dist.barrier()
        if self.device is not None:
            import torch
            torch.cuda.synchronize()

    def close(self):
        self.dist.destroy_process_group()


def load_checks():
    """Definition-pinned expectations of the north-star legs (tests/golden/bench_checks.npz, written on the CPU by
    tests/golden/make_bench_checks.py from the fp64 definitions of oracle/): data only, the bench never imports oracle/
    outside its cpu_baseline leg."""
    path = os.path.join(HERE, "tests", "golden", "bench_checks.npz")
    try:
        return np.load(path, allow_pickle=False)
    except OSError:
        return None


def rows_digest(hard, n_soft, rows):
    import hashlib
    h = hashlib.sha256()
    for r in rows:
        ns = int(n_soft[r])
        h.update(np.int32(ns).tobytes())
        h.update(np.ascontiguousarray(hard[r, :max(ns - 1, 0)]).tobytes())
    return h.hexdigest()


# ---- workloads of the north-star legs (shared with tests/golden/make_bench_checks.py) ---------------------------------
TETRA_FS, TETRA_N, TETRA_DISTINCT = 72000.0, 32768, 8
PFB_M, PFB_D, PFB_NIN, PFB_FS = 400, 125, 1048576, 10e6
WIDEBAND_CHANNELS = (0, 1, 57, 133, 199, 201, 310, 398, 399)   # band edges, both sides of DC, neighbours


def tetra_rows():
    """the 8 distinct channelised carriers of the tetra leg: cf32 at 72 kS/s (4 samples/symbol), own symbol seed and
    timing offset each, noise at 20 dB"""
    from tetraear_amd import synth
    base = [synth.dqpsk_baseband(TETRA_N, TETRA_FS, 700 + i, timing_offset=0.07 * i)[0].astype(np.complex64) for i in range(TETRA_DISTINCT)]
    rng = np.random.default_rng(5)
    return [b + (0.07 * (rng.standard_normal(TETRA_N) + 1j * rng.standard_normal(TETRA_N))).astype(np.complex64) for b in base]


def pfb_stream():
    """the channeliser leg's 10 MS/s cu8 stream (uniform random bytes: full-scale wideband noise)"""
    from tetraear_amd import synth
    return synth.noise_cu8(PFB_NIN, 1)


def wideband_stream():
    """BASELINE config 5's stream for the end-to-end leg: 10 MS/s, 1 Mi samples, pi/4-DQPSK carriers on nine channels of
    the 25 kHz grid at 25 dB, cu8.  Returns (u8, {channel: dibits})."""
    from tetraear_amd import synth
    x, dibs = synth.grid_carriers(PFB_NIN, PFB_FS, WIDEBAND_CHANNELS, PFB_M, seed0=300, snr_db=25.0)
    return synth.quantise_cu8(x, scale=1.0 / (4.0 * np.sqrt(len(WIDEBAND_CHANNELS) + 1.0))), dibs


STREAM_SEED0 = 1000   # stream g of the job has seed 1000 + g


def carrier_offsets(first, carriers):
    """post-decimation frequency offset of the job's carriers first .. first + carriers - 1 (21 values on the gate's AFC grid)"""
    g = np.arange(first, first + carriers)
    return (((g * 5) % 21) - 10) * 117.1875


def make_batch(carriers, chunk, fmt, first=0, workers=None):
    """Synthetic batch: `carriers` DISTINCT pi/4-DQPSK streams -- the job's carriers first .. first + carriers - 1, stream
    g with symbol / noise seed 1000 + g ("1024 separate cu8 streams (seeds 1000 + i)") and its own
    frequency offset.  Weak scaling: rank r holds the carriers r * C .. r * C + C - 1; strong scaling: its slice of the
    one job.  (Up to round 3 a batch tiled 8 distinct streams.)"""
    from tetraear_amd import synth
    u8 = synth.dqpsk_cu8_streams(chunk, SAMPLE_RATE, [STREAM_SEED0 + first + i for i in range(carriers)], workers).reshape(-1)
    foffs = carrier_offsets(first, carriers).astype(np.float64)
    if fmt == "cu8":
        return u8, foffs
    x = synth.cu8_to_c128(u8)
    if fmt == "cf32":
        return x.astype(np.complex64), foffs
    if fmt == "cf64":
        return x, foffs
    raise ValueError(fmt)


def make_shared_stream(chunk, tchunks, fmt, rank=0):
    """--shared: ONE stream of `tchunks` consecutive chunks (a single chunk: the job's stream `rank`, as before; more: one
    longer stream of its own seed, so that the chunks are consecutive pieces of one signal)"""
    if tchunks == 1:
        return make_batch(1, chunk, fmt, rank)
    from tetraear_amd import synth
    u8 = synth.dqpsk_cu8_streams(chunk * tchunks, SAMPLE_RATE, [STREAM_SEED0 + 5000 + rank], 1).reshape(-1)
    if fmt == "cu8":
        return u8, np.zeros(1)
    x = synth.cu8_to_c128(u8)
    return (x.astype(np.complex64) if fmt == "cf32" else x), np.zeros(1)


def shared_offsets(carriers):
    """--shared: the carriers' input-rate shifts, a 25 kHz grid centred on the stream (squeezed when more than 64 share it)"""
    return (np.arange(carriers) - (carriers - 1) / 2.0) * 25000.0 * (64.0 / max(carriers, 64))


def output_digest(hard, n_soft, best_phase):
    """SHA-256 over everything the hot path decided for the batch: per carrier the symbol count, the timing
    phase picked and the hard symbols.  Bit-exact work, so the digest is the same on every MI355X."""
    import hashlib
    h = hashlib.sha256()
    h.update(np.ascontiguousarray(n_soft, dtype=np.int32).tobytes())
    h.update(np.ascontiguousarray(best_phase, dtype=np.int32).tobytes())
    for r in range(len(n_soft)):
        h.update(np.ascontiguousarray(hard[r, :max(int(n_soft[r]) - 1, 0)]).tobytes())
    return h.hexdigest()


def digest_key(carriers, chunk, fmt, rate, rank, shared):
    return f"{fmt}:{carriers}x{chunk}@{rate:g}:rank{rank}" + (":shared" if shared else "")


def expected_digest(key):
    """Digests of the default workloads, pinned to the CPU oracle by tools/make_bench_digest.py (which compares every
    carrier of the batch with the oracle before it writes the file) and re-checked by tests/test_gpu_parity.py."""
    path = os.path.join(HERE, "tests", "golden", "bench_digest.json")
    if os.environ.get("TDM_BENCH_TEST_HOOK") and os.environ.get("TDM_BENCH_NO_PINNED"):
        return None     # (tests/: a stand-in device's outputs have no pinned digest; only honoured under the test hook)
    try:
        with open(path) as f:
            return json.load(f).get(key)
    except OSError:
        return None


def measured_traffic(samples_per_launch, fmt, key="k1"):
    """HBM bytes per K1 launch from the last committed rocprofv3 PMC profile (FETCH_SIZE x2 gfx950
    correction + WRITE_SIZE, separate passes; profiles/*_pmc_traffic.json), scaled per input sample.
    PMC counters cannot be read from inside this script, hence the committed figure."""
    import glob
    files = sorted(glob.glob(os.path.join(HERE, "profiles", "*_pmc_traffic.json")))
    if not files or fmt not in ("cu8", "tetra-cf32"):
        return None, None
    with open(files[-1]) as f:
        prof = json.load(f)
    if key not in prof:
        return None, None
    return prof[key]["hbm_bytes_per_input_sample"] * samples_per_launch, os.path.basename(files[-1])


def live_traffic(child_args, kernel_match, timeout_s=240):
    """HBM bytes per launch of the kernel whose name contains `kernel_match`, measured NOW on this box: two rocprofv3
    counter passes (FETCH_SIZE, WRITE_SIZE -- separate passes, with --kernel-trace only, as MI355X_MICROARCH.md's HBM
    section prescribes; FETCH_SIZE doubled on gfx950, counter unit KB) over a short child run of this script
    (`child_args` + 2 steps; the child skips its own measurements of this kind).  None when rocprofv3 is not there, the
    process is itself being profiled, TDM_BENCH_LIVE_PMC=0, or a pass fails / times out -- the caller then falls back to
    the last committed profile and says so in `traffic_source`."""
    import collections
    import csv
    import glob
    import shutil
    import subprocess
    import tempfile
    if os.environ.get("TDM_BENCH_LIVE_PMC", "1") == "0" or shutil.which("rocprofv3") is None:
        return None
    if any(k.startswith(("ROCPROF", "ROCP_")) for k in os.environ):   # nested profilers do not mix
        return None
    env = dict(os.environ, TDM_BENCH_LIVE_PMC="0", TMPDIR="/tmp")
    out = {}
    try:
        with tempfile.TemporaryDirectory(prefix="tdm_pmc_", dir="/tmp") as tmp:
            for counter in ("FETCH_SIZE", "WRITE_SIZE"):
                d = os.path.join(tmp, counter)
                cmd = ["rocprofv3", "--pmc", counter, "--kernel-trace", "--output-format", "csv", "-d", d, "-o", "c", "--",
                       sys.executable, os.path.abspath(__file__)] + list(child_args) + ["--steps", "2", "--warmup", "1", "--pmc-child"]
                r = subprocess.run(cmd, env=env, cwd="/tmp", stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, timeout=timeout_s)
                if r.returncode != 0:
                    return None
                vals = collections.defaultdict(list)
                for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
                    with open(f) as fh:
                        for row in csv.DictReader(fh):
                            if row["Counter_Name"] == counter and kernel_match in row["Kernel_Name"]:
                                vals[row["Kernel_Name"]].append(float(row["Counter_Value"]))
                if len(vals) != 1:
                    return None
                name, v = next(iter(vals.items()))
                out["kernel"] = name
                out[counter] = sum(v) / len(v) * 1024.0
                out["launches_" + counter] = len(v)
    except Exception:  # noqa: BLE001 -- a side measurement never breaks the bench line
        return None
    return {"kernel": out["kernel"], "fetch_bytes_raw": out["FETCH_SIZE"], "fetch_bytes_corrected_x2": 2 * out["FETCH_SIZE"],
            "write_bytes": out["WRITE_SIZE"], "hbm_bytes_per_launch": 2 * out["FETCH_SIZE"] + out["WRITE_SIZE"],
            "launches_averaged": [out["launches_FETCH_SIZE"], out["launches_WRITE_SIZE"]],
            "how": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes, --kernel-trace only) over a 2-step child run of "
                   "this command on this box, inside this bench run; FETCH_SIZE x 2 (gfx950), unit KB"}


def traffic_now(child_args, kernel_match, samples_per_launch, fmt, key):
    """(traffic bytes per launch, source, detail): measured live if possible, else the last committed profile's figure"""
    live = live_traffic(child_args, kernel_match)
    if live is not None:
        return live["hbm_bytes_per_launch"], "live: rocprofv3 PMC passes inside this bench run", live
    t, src = measured_traffic(samples_per_launch, fmt, key)
    return t, (src + " (committed profile: live PMC passes unavailable)") if src else None, None


def cpu_baseline(chunk, budget_s=10.0):
    """The CPU oracle (C restatement of the reference chain, single thread) on the same workload,
    bounded to ~budget_s of CPU work.  Reported next to the GPU number; never the thing shipped."""
    from oracle.oracle import OracleSignalProcessor
    from tetraear_amd import synth
    u8, _ = synth.dqpsk_cu8(chunk, SAMPLE_RATE, seed=4242)
    x = synth.cu8_to_c128(u8)
    o = OracleSignalProcessor(SAMPLE_RATE)
    o.process(x, 117.1875)  # warm
    n_sym, reps = 0, 0
    t0 = time.perf_counter()
    while time.perf_counter() - t0 < budget_s:
        n_sym += len(o.process(x, 117.1875))
        reps += 1
    dt = time.perf_counter() - t0
    return {"value": n_sym / dt / 1e6, "unit": "Msym/s", "cores": 1, "kind": "port",
            "sample": f"{reps} chunks of {chunk} samples @2.4 MS/s (cf64 in), C oracle liboracle.so, "
                      f"{dt:.1f} s on 1 thread of {os.cpu_count()} host CPUs"}


def _cpu_worker(args):
    chunk, budget_s = args
    from oracle.oracle import OracleSignalProcessor
    from tetraear_amd import synth
    x = synth.cu8_to_c128(synth.noise_cu8(chunk, 99))
    o = OracleSignalProcessor(SAMPLE_RATE)
    o.process(x, 117.1875)
    n, t0 = 0, time.perf_counter()
    while time.perf_counter() - t0 < budget_s:
        n += len(o.process(x, 117.1875))
    return n, time.perf_counter() - t0


def cpu_baseline_allcore(chunk, budget_s=5.0):
    """Same oracle, one process per host CPU (the path is single-threaded; carriers are independent)."""
    import multiprocessing as mp
    procs = os.cpu_count() or 1
    with mp.get_context("fork").Pool(procs) as pool:
        res = pool.map(_cpu_worker, [(chunk, budget_s)] * procs)
    rate = sum(n / dt for n, dt in res) / 1e6
    return {"value": rate, "unit": "Msym/s", "cores": procs, "kind": "port",
            "sample": f"{procs} processes x {budget_s:.0f} s of {chunk}-sample chunks, C oracle"}


def emit(out):
    """Print the ONE JSON line.  Under TDM_BENCH_TEST_HOOK (tests/: a stand-in device) the line says so, in a field of its own
    and in `data`, so that no hooked run can pass for a measurement."""
    hook = os.environ.get("TDM_BENCH_TEST_HOOK")
    if hook:
        out = dict(out, test_hook=os.path.abspath(hook), data="test hook: stand-in device, NOT a measurement")
    print(json.dumps(out), flush=True)


def spawn_local_ranks(n):
    """`python bench.py --gpus N` without a launcher: start N ranks of this very command on this node -- rank r on device r,
    rendezvous on 127.0.0.1 (RcclGroup: librccl through ctypes) -- pass rank 0's one JSON line through, fail if any rank
    fails.  The environment is what torch.distributed.run would have set."""
    import socket
    import subprocess
    if not os.environ.get("TDM_BENCH_TEST_HOOK"):      # (the CPU-tier test's stand-in device has no count to ask for)
        from tetraear_amd import _lib
        have = int(_lib.load().tdm_device_count())
        if have < n:
            raise SystemExit(f"bench: --gpus {n} needs {n} devices on this node, the library sees " + (str(have) if have >= 0 else f"none (tdm_device_count: status {have})"))
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    run_id = f"bench-self-{os.getpid()}"
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), TORCHELASTIC_RUN_ID=run_id, TDM_RCCL_TOKEN=f"{run_id}|{port}")
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env,
                                      stdout=None if r == 0 else subprocess.DEVNULL))
    rcs = [p.wait() for p in procs]
    if any(rcs):
        raise SystemExit(f"bench: ranks exited with {rcs}")
    return 0


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20,
                    help="untimed steps right before the timed region (untimed settling passes are added in front of them, see SETTLE_STEPS)")
    ap.add_argument("--carriers", type=int, default=None,
                    help="carriers per GPU: weak scaling, every rank its own (default at N = 1: 1024, SURVEY 8(d) C4's batch)")
    ap.add_argument("--chunk", type=int, default=262144, help="samples per carrier per step")
    ap.add_argument("--fmt", default="cu8", choices=["cu8", "cf32", "cf64"])
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--mode", default="reference", choices=["reference", "tetra", "pfb", "stream", "wideband"],
                    help="reference = parity mode (the metric); tetra = RRC/timing/Farrow receiver on channelised cf32")
    ap.add_argument("--zero-foff", action="store_true", help="experiment: all freq offsets 0 (NCO skipped)")
    ap.add_argument("--shared", action="store_true",
                    help="BASELINE config 3: all carriers read ONE shared wideband stream, each shifted to baseband by its "
                         "own offset on load (process(frequency_shift(x, f_k)) per carrier)")
    ap.add_argument("--chunks", type=int, default=1,
                    help="--shared: T consecutive chunks of the stream per step, every carrier out of every chunk (plan rows = T x "
                         "carriers, plan option rows_per_chunk): the reference's chunk-after-chunk caller loop, T turns per launch")
    ap.add_argument("--exact-shift", action="store_true",
                    help="--shared: the input-rate shift reproduces the reference's rounding of its phase sample by sample (the "
                         "library's default) instead of the plan option fast_pre_shift (ideal phase ramp, exactly anchored per lane)")
    ap.add_argument("--rate", type=float, default=SAMPLE_RATE, help="sample rate (experiments; metric config is 2.4e6)")
    ap.add_argument("--total-carriers", type=int, default=None,
                    help="strong scaling: this many carriers in total, block-partitioned over the ranks.  The default at N > 1 when "
                         "neither this nor --carriers is given: 1024 = BASELINE config 4 as written (\"1024 independent carriers "
                         "sharded across 8 GPUs\"), with the weak-scaling figure (every rank its own 1024) as the side field `weak`")
    ap.add_argument("--no-extra", action="store_true", help="skip the tetra / pfb / wideband / single-carrier legs appended at N = 1")
    ap.add_argument("--depth", default="auto",
                    help="plans per rank that take the steps in turn, each on its own stream (tetraear_amd.batch.PipelinedBatchDemodulator): "
                         "auto = 3 (up to three steps in flight); 1 = one plan, steps strictly one after the other")
    ap.add_argument("--pmc-child", action="store_true",
                    help="(internal) the short run rocprofv3 counts HBM traffic on: noise input, no output check, no side measurements")
    args = ap.parse_args()
    # which workload: see --total-carriers.  (N = 1: the two coincide -- 1024 carriers on the one GPU.)
    world_env = int(os.environ.get("WORLD_SIZE", "1"))
    args.weak_side = False
    if args.carriers is None and args.total_carriers is None:
        if world_env > 1 and args.mode == "reference" and not args.shared:
            args.total_carriers, args.weak_side = 1024, True
        args.carriers = 1024
    elif args.carriers is None:
        args.carriers = 1024
    args.total_carriers = args.total_carriers or 0

    hook = os.environ.get("TDM_BENCH_TEST_HOOK")
    if hook:
        # tests/ only: replaces the device with a stand-in so that the launch logic runs on a CPU-only box.  A measurement
        # script says so itself when it is not measuring: the hook is REFUSED on a box that has a device, and every JSON line
        # printed under it carries "test_hook": <path> and "data": "test hook ..." (emit()).
        from tetraear_amd import _lib
        if int(_lib.load().tdm_device_count()) > 0:
            raise SystemExit("bench: TDM_BENCH_TEST_HOOK is set but this box has a GPU; refusing to run a measurement with a stand-in device")
        import runpy
        runpy.run_path(hook, init_globals={"bench": sys.modules[__name__]})
    if args.mode == "tetra":
        return main_tetra(args)
    if args.mode == "pfb":
        return main_pfb(args)
    if args.mode == "wideband":
        return main_wideband(args)
    if args.mode == "stream":
        return main_stream(args)
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        # `python bench.py --gpus N` by itself: N local ranks, one per device, spawned here (the driver's torch.distributed.run
        # line sets WORLD_SIZE and never comes this way)
        return spawn_local_ranks(args.gpus)
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus != world:
        # a line that says n_gpus: N must come from N ranks
        raise SystemExit(f"bench: --gpus {args.gpus} but the launcher started {world} rank(s) (WORLD_SIZE); refusing to report")
    group, collective = None, "none (single process)"
    if world > 1 or os.environ.get("TDM_FORCE_DIST") == "1":   # the env switch lets a 1-GPU box exercise this path
        # barrier + three 8-byte all-reduces: librccl through ctypes (no tensor framework in the product path).  The choice
        # of backend is COLLECTIVE: RcclGroup's bring-up ends with the same decision on every rank (tetraear_amd/rccl.py), so
        # either all ranks run on librccl or all of them take torch.distributed's "nccl" backend (the same RCCL); any other
        # failure is fatal for the job instead of being papered over on one rank.
        from tetraear_amd.rccl import RcclGroup, RcclUnavailable
        try:
            if os.environ.get("TDM_DIST_BACKEND", "rccl") != "rccl":   # (an environment switch is the same on every rank)
                raise RcclUnavailable("torch.distributed requested")
            group = RcclGroup(rank, world, local_rank)
            collective = "librccl via ctypes (ncclAllReduce x3 + barrier)"
        except RcclUnavailable as e:
            import torch
            import torch.distributed as dist
            if os.environ.get("TDM_DIST_BACKEND") == "gloo":
                # dry run of the N > 1 logic where the ranks cannot have a device each (several ranks on ONE GPU: RCCL refuses
                # duplicate devices): host-side reductions; the line says so, its timing means nothing
                dist.init_process_group("gloo")
                group = TorchGroup(dist, None)
                collective = "torch.distributed gloo (TDM_DIST_BACKEND=gloo: a dry run of the multi-rank logic, NOT a measurement)"
            else:
                torch.cuda.set_device(local_rank)
                dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
                group = TorchGroup(dist, "cuda")
                collective = f"torch.distributed nccl (librccl through ctypes not usable on every rank: {e})"

    from tetraear_amd import batch as batch_mod
    from tetraear_amd.shard import carrier_range, reduce_job

    # weak scaling (default): every rank demodulates --carriers carriers of its own.  Strong scaling (--total-carriers T,
    # BASELINE config 4: 1024 carriers over 8 GPUs): the T carriers of ONE batch are block-partitioned over the ranks.
    strong = args.total_carriers > 0
    lo, hi = carrier_range(args.total_carriers, rank, world) if strong else (0, args.carriers)
    carriers = hi - lo
    t_plan = time.perf_counter()
    tchunks = max(1, args.chunks) if args.shared else 1
    per_chunk = carriers            # carriers shifted out of one chunk of the stream
    carriers = carriers * tchunks   # plan rows
    bd = batch_mod.batch_demodulator(args.rate, args.chunk, carriers, args.fmt, device=local_rank,
                                     depth=args.depth if args.depth == "auto" else int(args.depth))
    depth = getattr(bd, "depth", 1)
    if tchunks > 1:
        bd.set_rows_per_chunk(per_chunk)
    bd.sync()
    plan_create_ms = (time.perf_counter() - t_plan) * 1e3
    bd.alloc_device_io(shared_input=args.shared and tchunks == 1)
    fast_shift = args.shared and not args.exact_shift
    if fast_shift:
        bd.set_fast_pre_shift()
    gen_workers = max(1, min(64, (os.cpu_count() or 1) // max(1, int(os.environ.get("LOCAL_WORLD_SIZE", world)))))
    if args.pmc_child:
        # (the counters do not care what the bytes say: uniform noise, made in a second instead of the job's 1024 streams)
        from tetraear_amd import synth
        iq = synth.noise_cu8((tchunks if args.shared else carriers) * args.chunk, 7)
        if args.fmt != "cu8":
            iq = synth.cu8_to_c128(iq).astype(np.complex64 if args.fmt == "cf32" else np.complex128)
        if args.shared:
            bd.upload(iq, freq_offsets=None, pre_shifts=np.tile(shared_offsets(per_chunk), tchunks))
        else:
            bd.upload(iq, freq_offsets=carrier_offsets(0, carriers).astype(np.float64))
    elif args.shared:
        iq, foffs = make_shared_stream(args.chunk, tchunks, args.fmt, rank)
        foffs = np.zeros(carriers)
        pre = np.tile(shared_offsets(per_chunk), tchunks)
        bd.upload(iq, freq_offsets=None, pre_shifts=pre)
    else:
        # every carrier of the job is its own stream: this rank's are first .. first + carriers - 1
        first = lo if strong else rank * carriers
        iq, foffs = make_batch(carriers, args.chunk, args.fmt, first, gen_workers)
        if args.zero_foff:
            foffs = foffs * 0.0
        bd.upload(iq, freq_offsets=foffs)

    def barrier():
        bd.sync()
        if group is not None:
            group.barrier()

    # Clock settling: after idle the GPU's clocks ramp for ~40 launches (~40 ms; per-launch kernel times under rocprofv3 go
    # 0.56 -> 0.68 -> 0.50 ms for the decimator, profiles/*_kernel_timed.json), so a short --warmup would put the timed
    # region inside the ramp.  Untimed passes of the same hot path are added in front of the warm-up until at least
    # SETTLE_STEPS passes have run; the W warm-up steps and the K timed steps follow unchanged.
    settle_steps = 0 if args.pmc_child else max(0, SETTLE_STEPS - args.warmup)
    for _ in range(settle_steps):
        bd.enqueue()
    for _ in range(args.warmup):
        bd.enqueue()
    barrier()
    # the timed region: exactly `steps` passes, launches back to back as a caller issues them, one HIP-event pair around
    # the whole region on the stream the kernels run on
    bd.time_begin(per_stage=False)
    t0 = time.perf_counter()
    for _ in range(args.steps):
        bd.enqueue()
    ev_ms = bd.time_end()
    barrier()
    dt = time.perf_counter() - t0
    # the per-kernel pass: the same `steps` passes again with a HIP-event pair around every launch (the roofline's kernel
    # time).  The events themselves cost ~5 us per launch, which is why this is not the region `value` is taken from.
    bd.time_begin(per_stage=True)
    t1 = time.perf_counter()
    for _ in range(args.steps):
        bd.enqueue()
    ev_ms_staged = bd.time_end()
    barrier()
    dt_staged = time.perf_counter() - t1
    stage_ms = bd.stage_times()   # average per launch over the per-kernel pass

    hard, soft, n_soft, bp, mm = bd.download()
    sym_per_step = int(np.sum(np.maximum(n_soft.astype(np.int64) - 1, 0)))
    # the output of the last timed step is checked, not just counted: its digest must equal the one pinned to the oracle --
    # and so must the last output of every other plan that took steps in turn with it
    dkey = digest_key(per_chunk, args.chunk, args.fmt, args.rate, rank, args.shared) + (f":chunks{tchunks}" if tchunks > 1 else "") + (f":strong{lo}-{hi}of{args.total_carriers}" if strong and world > 1 else "")
    digest, want = output_digest(hard, n_soft, bp), expected_digest(dkey)
    if hasattr(bd, "download_all"):
        for o in bd.download_all():
            d2 = output_digest(o[0], o[2], o[3])
            if d2 != digest:
                digest = "plans disagree: " + digest + " / " + d2
                break
            mm = np.minimum(mm, o[4])
    if args.zero_foff or args.pmc_child:
        want = None
    mismatch = want is not None and digest != want
    fast_shift_report = None
    if fast_shift:
        # the guard of the fast pre-shift: no carrier's smallest decision margin within reach of the phase rounding it skips
        # (1e-8 rad is 100x that rounding), and -- below -- the digest pinned to the ORACLE's exact-phase decisions
        fast_shift_report = {"min_margin_rad": float(np.min(mm)), "guard_rad": 1e-8, "carriers_below_guard": int(np.sum(mm < 1e-8))}
    output_check = {"key": dkey, "sha256": digest,
                    "status": "matches oracle-pinned digest" if want == digest else
                              ("not compared" if (args.zero_foff or args.pmc_child) else
                               ("no pinned digest for this workload" if want is None else "DIFFERS from the oracle-pinned digest"))}

    # (a rank whose check failed still takes part in the reductions: the failure count is reduced with the job and ALL
    # ranks stop together below; leaving early would strand the others inside ncclAllReduce)
    dt, total_sym_per_step, n_bad = reduce_job(group, dt, sym_per_step, 1 if mismatch else 0)
    if n_bad:
        if mismatch:
            print(f"bench: rank {rank}: output digest {digest} differs from the oracle-pinned {want} for {dkey}", file=sys.stderr)
        bd.close()
        if group is not None:
            group.close()
        raise SystemExit(f"bench: the output check failed on {n_bad} rank(s)")

    info_n_dec, info_engine = int(bd.info.n_dec), int(bd.info.dec_engine)
    weak = None
    if args.weak_side:
        # BASELINE config 4's other reading, beside the headline: every rank its own 1024 carriers (8192 on 8 GPUs)
        bd.close()
        bd = None
        weak = weak_side(args, group, rank, world, local_rank, gen_workers)
    if rank == 0:
        value = total_sym_per_step * args.steps / dt / 1e6
        sym_rate_per_carrier = SAMPLE_RATE / 10 / 13   # 18461.5 sym/s in reference mode @2.4 MS/s
        k1_ms = stage_ms.get("dec_block", float("nan"))
        samples_per_launch = carriers * args.chunk
        in_bytes = {"cu8": 2, "cf32": 8, "cf64": 16}[args.fmt]
        n_dec = info_n_dec
        k1_bytes = samples_per_launch * in_bytes + carriers * n_dec * 16
        contract_tf = samples_per_launch * FLOP_PER_INPUT_SAMPLE / (k1_ms * 1e-3) / 1e12
        # executed flops: exact for the raw-byte kernel (dec_engine 3, the bench's case); the double-based kernel executes
        # 52.5 FMAs per sample and the cascade engine 133 issue slots: other engines report the same field from their counts
        # (a call with an input-rate pre-shift -- --shared -- runs the double-based kernel whatever the batch size: the raw-byte
        # kernel works on integers, and a rotated sample is none; tdm_plan_info cannot know the call's arguments)
        engine = 2 if (args.shared and info_engine == 3) else info_engine
        # --shared adds, per input sample, the conversion (4 flop) and the input-rate NCO that reproduces the reference's own
        # rounding of its phase (NcoRunT: Markstein quotient 10, phase 1, phasor advance 6, correction 7, rotation 6: ~35 flop)
        exec_flop = {3: EXECUTED_FLOP_PER_INPUT_SAMPLE, 2: 105.0, 1: 266.0}.get(engine, EXECUTED_FLOP_PER_INPUT_SAMPLE) + ((16.0 if fast_shift else 39.0) if args.shared else 0.0)
        executed_tf = samples_per_launch * exec_flop / (k1_ms * 1e-3) / 1e12
        irreducible_tf = samples_per_launch * IRREDUCIBLE_FLOP_PER_INPUT_SAMPLE / (k1_ms * 1e-3) / 1e12
        ceil = hbm_ceiling(local_rank) if (world == 1 and not args.pmc_child) else {}
        traffic, traffic_src, traffic_detail = None, None, None
        if world == 1 and not args.pmc_child:
            child = ["--carriers", str(carriers), "--chunk", str(args.chunk), "--fmt", args.fmt, "--rate", repr(args.rate),
                     "--no-cpu-baseline", "--no-extra", "--depth", str(depth)] + (["--shared", "--chunks", str(tchunks)] if args.shared else [])
            traffic, traffic_src, traffic_detail = traffic_now(child, "k_pz_raw<" if engine == 3 else ("k_pz_block<" if engine == 2 else "k_zp_block<"),
                                                               samples_per_launch, args.fmt, "k1")
        total = args.total_carriers if strong else carriers * world
        out = {
            "metric": "Msymbols/s demodulated (reference-parity mode, hard symbols written)",
            "value": value,
            "unit": "Msym/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": dt / args.steps * 1e3,
            "higher_is_better": True,
            "scaling": "strong" if strong else "weak",
            "total_carriers": total,
            "vs_baseline": None,
            "dtype": "f64",
            "data": "synthetic",
            "config": {"workload": (f"{per_chunk} carriers shifted out of ONE shared {args.fmt} stream @{args.rate / 1e6:g} MS/s, "
                                    f"{tchunks} consecutive chunk(s) of {args.chunk} samples per step (SURVEY 8(d) C3)") if args.shared else
                                   (f"{total} independent 25 kHz carriers, every one its own stream (seeds 1000 + g), over {world} GPU(s) ({carriers} on rank 0), "
                                    f"{args.chunk}-sample {args.fmt} chunks @{args.rate / 1e6:g} MS/s (SURVEY 8(d) C4"
                                    f"{'' if strong else ' per-GPU share x ' + str(world)})"),
                       "carriers_per_gpu": carriers, "chunk_samples": args.chunk, "in_fmt": args.fmt,
                       "steps_in_flight": {"plans": depth,
                                           "how": ("one plan, one stream: steps strictly one after the other" if depth == 1 else
                                                   f"{depth} plans of the whole batch take the steps in turn, each on its own stream and work buffers "
                                                   "(tetraear_amd.batch.PipelinedBatchDemodulator): a step's carries / low-rate stage / finish run beside "
                                                   "the next step's decimator; every launch is the full batch, every plan's output is checked (digest below)")},
                       "mode": "reference", "parallelism": f"carriers sharded over {world} GPU(s), no data-path collective",
                       "collective": collective},
            "realtime_carriers": value * 1e6 / sym_rate_per_carrier,
            "output_check": output_check,
            "pre_shift": (None if not args.shared else
                          ({"phase": "ideal ramp from an exactly anchored sample per lane (plan option fast_pre_shift)", **fast_shift_report}
                           if fast_shift else {"phase": "the reference's rounding of theta reproduced sample by sample"})),
            "rccl_ranks": world if group is not None else 0,
            "plan_create_ms": plan_create_ms,
            "settle_steps": settle_steps,   # untimed passes before the warm-up (clock ramp after idle)
            "event_ms_per_step_rank0": ev_ms / args.steps,
            "ms_per_step_per_kernel_pass_rank0": dt_staged / args.steps * 1e3,   # same steps with events around every launch
            "stage_ms_per_launch": stage_ms,
            "stage_timing": ("HIP events around every launch on its plan's stream" if depth == 1 else
                             f"HIP events around every launch on its plan's stream, the steps ordered ONE AFTER THE OTHER on the device in this pass "
                             "(tdm_plan_wait_for: every launch alone on the device); in the timed region consecutive steps overlap, so a launch "
                             "there takes longer than here while a step takes less than the sum of its launches"),
            "roofline": {
                "kernel": ("k_pz_raw<10,12,27> (zero-phase Chebyshev-8 decimator in parallel form: causal + anticausal all-pole banks on the raw samples)"
                           if engine == 3 else
                           ("k_pz_block<shift> (the same decimator on doubles, every sample first rotated by its carrier's input-rate NCO: frequency_shift(x, f_k) of the shared stream)"
                            if args.shared else
                            "k_pz_block (the same decimator with the samples held as doubles: few carriers, or a wire format other than cu8)")
                           if engine == 2 else "k_zp_block (cascade engine)"),
                "bound": "valu_fp64",
                # frac = what the fp64 vector ALUs execute over their peak (never > 1); round 2 reported SURVEY 8(d)'s
                # operation count over the measured time here, which the parallel form undercuts (1.02 "of peak")
                "achieved": executed_tf,
                "peak": PEAK_FP64_TFLOPS,
                "unit": "TFLOP/s",
                "frac": executed_tf / PEAK_FP64_TFLOPS,
                "flop_per_input_sample": exec_flop,
                "avg_launch_ms": k1_ms,
                "irreducible": {"flop_per_input_sample": IRREDUCIBLE_FLOP_PER_INPUT_SAMPLE, "achieved": irreducible_tf,
                                "frac": irreducible_tf / PEAK_FP64_TFLOPS,
                                "note": "the recurrences alone (2 FMAs per real sample, pole pair and direction): what no form of this filter avoids"},
                "contract_survey_8d": {"flop_per_input_sample": FLOP_PER_INPUT_SAMPLE, "algorithmic_flop_per_launch": samples_per_launch * FLOP_PER_INPUT_SAMPLE,
                                       "rate": contract_tf, "rate_over_peak": contract_tf / PEAK_FP64_TFLOPS,
                                       "note": "SURVEY 8(d) prices the cascade (150 flop/sample); the parallel form executes 84.7, so this "
                                               "rate can exceed the ALU peak and is NOT a fraction of anything the kernel does"},
                "traffic": traffic,
                "traffic_source": traffic_src,
                "traffic_detail": traffic_detail,
                "hbm": {"achieved": k1_bytes / (k1_ms * 1e-3) / 1e9, "peak": PEAK_HBM_GBS, "unit": "GB/s",
                        "frac": k1_bytes / (k1_ms * 1e-3) / 1e9 / PEAK_HBM_GBS,
                        "peak_measured": ceil.get("copy"),
                        "frac_of_measured": (k1_bytes / (k1_ms * 1e-3) / 1e9 / ceil["copy"]) if "copy" in ceil else None,
                        "algorithmic_bytes_per_launch": k1_bytes},
                "note": "no MFMA on this path; the kernel is fp64-vector-ALU bound (78.6 TFLOP/s = 16 fp64 FMA lanes per SIMD and "
                        "clock); the HBM view of the same launch is under 'hbm'",
            },
            "hbm_ceiling_measured": ceil or None,
        }
        if not args.no_cpu_baseline and world == 1 and not args.pmc_child:   # reported on rank 0 at N = 1 only
            out["cpu_baseline"] = cpu_baseline(args.chunk)
            try:
                out["cpu_baseline_allcore"] = cpu_baseline_allcore(args.chunk)
            except Exception as e:  # never let the side measurement break the bench line
                out["cpu_baseline_allcore"] = {"error": str(e)}
    if bd is not None:
        bd.close()
    if rank == 0:
        if weak is not None:
            out["weak"] = weak
        if (world == 1 and not args.no_extra and not strong and not args.shared and not args.pmc_child and args.fmt == "cu8"
                and depth > 1):
            # beside the headline: the same batch on ONE plan, steps strictly one after the other (the round-1..5 headline)
            try:
                out["one_plan"] = leg_depth(iq, foffs, carriers, args.chunk, args.rate, args.steps, want, 1)
            except Exception as e:  # noqa: BLE001
                out["one_plan"] = {"error": str(e)}
        if world == 1 and not args.no_extra and not strong and not args.shared:
            # the north-star stages beside the headline (own workloads, HIP-event timing; none of them is `value`)
            for key, leg, c in (("single_carrier", leg_single, 1), ("tetra", leg_tetra, 4096), ("pfb", leg_pfb, 12800),
                                ("wideband", leg_wideband, 12800)):
                try:
                    out[key] = leg(c, 100, 60)   # (steady state: see --warmup)
                except Exception as e:  # noqa: BLE001
                    out[key] = {"error": str(e)}
    if rank == 0:
        emit(out)
    if group is not None:
        group.close()
    if rank == 0:
        bad = [k for k in ("tetra", "pfb", "wideband", "two_plans", "one_plan") if isinstance(out.get(k), dict)
               and str(out[k].get("output_check", {}).get("status", "")).startswith("DIFFERS")]
        if bad:   # (the line is printed for the record; the run does not count as a success)
            raise SystemExit(f"bench: output check failed in leg(s) {bad}")


def weak_side(args, group, rank, world, local_rank, gen_workers):
    """The weak-scaling run beside a strong-scaling headline (N > 1, default arguments): every rank demodulates 1024 carriers
    of its own (stream g of the job = seed 1000 + g, rank r holds r * 1024 ...), same settle / warm-up / timed steps, barrier
    and device synchronisation on both sides, max over ranks; every rank's output against its oracle-pinned digest.
    Collective: every rank calls it.  Returns the side field on rank 0 (None elsewhere); raises on a failed check."""
    from tetraear_amd import batch as batch_mod
    from tetraear_amd.shard import reduce_job
    carriers = 1024
    bd = batch_mod.batch_demodulator(args.rate, args.chunk, carriers, args.fmt, device=local_rank,
                                     depth=args.depth if args.depth == "auto" else int(args.depth))
    bd.alloc_device_io()
    iq, foffs = make_batch(carriers, args.chunk, args.fmt, rank * carriers, gen_workers)
    bd.upload(iq, freq_offsets=foffs)

    def barrier():
        bd.sync()
        if group is not None:
            group.barrier()
    for _ in range(max(0, SETTLE_STEPS - args.warmup) + args.warmup):
        bd.enqueue()
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        bd.enqueue()
    barrier()
    dt = time.perf_counter() - t0
    hard, soft, n_soft, bp, mm = bd.download()
    bd.close()
    sym = int(np.sum(np.maximum(n_soft.astype(np.int64) - 1, 0)))
    dkey = digest_key(carriers, args.chunk, args.fmt, args.rate, rank, False)
    digest, want = output_digest(hard, n_soft, bp), expected_digest(dkey)
    bad = want is not None and digest != want
    dt, total_sym, n_bad = reduce_job(group, dt, sym, 1 if bad else 0)
    if n_bad:
        if bad:
            print(f"bench: rank {rank}: weak-scaling output digest {digest} differs from the oracle-pinned {want} for {dkey}", file=sys.stderr)
        if group is not None:
            group.close()
        raise SystemExit(f"bench: the weak-scaling side run's output check failed on {n_bad} rank(s)")
    if rank != 0:
        return None
    return {"scaling": "weak", "value": total_sym * args.steps / dt / 1e6, "unit": "Msym/s", "ms_per_step": dt / args.steps * 1e3,
            "total_carriers": carriers * world, "carriers_per_gpu": carriers,
            "output_check_rank0": {"key": dkey, "status": "matches oracle-pinned digest" if want == digest else "no pinned digest for this workload"},
            "note": "every rank its own 1024 distinct carriers; the headline `value` is the strong-scaling run of ONE 1024-carrier job"}


def leg_depth(iq, foffs, carriers, chunk, rate, steps, want, depth):
    """The same batch with `depth` plans taking the steps in turn (1: one plan, one stream), wall clock between device
    synchronisations, every plan's digest checked.  Side figure beside the headline."""
    from tetraear_amd.batch import batch_demodulator
    bd = batch_demodulator(rate, chunk, carriers, "cu8", depth=depth)
    bd.alloc_device_io()
    bd.upload(iq, freq_offsets=foffs)
    for _ in range(SETTLE_STEPS):
        bd.enqueue()
    bd.sync()
    t0 = time.perf_counter()
    for _ in range(steps):
        bd.enqueue()
    bd.sync()
    ms = (time.perf_counter() - t0) / steps * 1e3
    outs = bd.download_all() if hasattr(bd, "download_all") else [bd.download()]
    bd.close()
    nsym = int(np.sum(np.maximum(outs[0][2].astype(np.int64) - 1, 0)))
    digests = sorted({output_digest(o[0], o[2], o[3]) for o in outs})
    return {"workload": f"{depth} plan(s), one stream each, wall clock between device synchronisations", "ms_per_step": ms,
            "value": nsym / (ms * 1e-3) / 1e6, "unit": "Msym/s",
            "output_check": {"sha256": digests[0] if len(digests) == 1 else digests,
                             "status": ("matches oracle-pinned digest" if digests == [want] else
                                        ("no pinned digest for this workload" if want is None else "DIFFERS from the oracle-pinned digest"))}}


def leg_single(carriers, steps, warmup):
    """The reference's own use: ONE carrier, one 262 144-sample chunk per call (launch- and latency-bound)."""
    from tetraear_amd.batch import BatchDemodulator
    t0 = time.perf_counter()
    bd = BatchDemodulator(SAMPLE_RATE, 262144, carriers, "cu8")
    bd.sync()
    create_ms = (time.perf_counter() - t0) * 1e3
    bd.alloc_device_io()
    iq, foffs = make_batch(carriers, 262144, "cu8", 0)
    bd.upload(iq, freq_offsets=foffs)
    for _ in range(warmup):
        bd.enqueue()
    bd.sync()
    # per-stage pass (events around every launch), then the calls as a capture loop makes them: back to back, HIP events
    # around the whole loop only
    bd.time_begin()
    for _ in range(steps):
        bd.enqueue()
    ms_staged = bd.time_end() / steps
    st = bd.stage_times()
    for _ in range(3):
        bd.enqueue()
    bd.sync()
    bd.time_begin(per_stage=False)
    for _ in range(steps):
        bd.enqueue()
    ms = bd.time_end() / steps
    # a length the plan has not seen, then one it has (tdm_plan_resize: ragged read sizes on one plan)
    t0 = time.perf_counter()
    bd.resize(131072)
    resize_new_ms = (time.perf_counter() - t0) * 1e3
    t0 = time.perf_counter()
    bd.resize(262144)
    resize_seen_ms = (time.perf_counter() - t0) * 1e3
    hard, soft, n_soft, bp, mm = bd.download()
    bd.close()
    nsym = int(np.maximum(n_soft - 1, 0).sum())
    # the drop-in call itself, host arrays in and out (PCIe and the Python shim included; never `value`): what a caller of
    # tetraear.signal.SignalProcessor.process sees per 262144-sample read
    from tetraear_amd import synth
    from tetraear_amd.signal import SignalProcessor
    sp = SignalProcessor(SAMPLE_RATE)
    u8 = np.ascontiguousarray(iq[:2 * 262144])
    x128 = synth.cu8_to_c128(u8)
    host = {}
    for name, call in (("process_complex128", lambda: sp.process(x128, float(foffs[0]))), ("process_cu8", lambda: sp.process_cu8(u8, float(foffs[0])))):
        for _ in range(5):
            out = call()
        t0 = time.perf_counter()
        for _ in range(30):
            out = call()
        host[name + "_ms_per_call"] = (time.perf_counter() - t0) / 30 * 1e3
        host[name + "_symbols"] = int(len(out))
    return {"workload": f"{carriers} carrier x 262144 cu8 samples @2.4 MS/s", "ms_per_step": ms, "value": nsym / (ms * 1e-3) / 1e6,
            "host_call": host,
            "unit": "Msym/s", "x_realtime": (262144 / SAMPLE_RATE) / (ms * 1e-3), "plan_create_ms": create_ms,
            "plan_resize_ms": {"new_length": resize_new_ms, "seen_length": resize_seen_ms},
            "ms_per_step_staged": ms_staged, "stage_ms_per_launch": st,
            "timing": "HIP events on the plan's stream; ms_per_step = calls back to back, events around the loop; "
                      "ms_per_step_staged = the per-stage pass (events around each of the launches)"}


def main_stream(args):
    """Host-fed leg (PCIe inclusive; never the headline `value`): cu8 batches in pageable host memory,
    pinned in place, H2D / kernels / D2H overlapped (tdm_process_pipelined)."""
    from tetraear_amd.batch import BatchDemodulator
    rows = min(args.carriers, 256)
    n_batches = 8
    bd = BatchDemodulator(SAMPLE_RATE, args.chunk, rows, "cu8")
    iq, foffs = make_batch(rows, args.chunk, "cu8", 0)
    allq = np.concatenate([iq] * n_batches)
    bd.process_stream(allq, n_batches, foffs)   # warm
    t0 = time.perf_counter()
    reps = max(1, args.steps // 4)
    for _ in range(reps):
        hard, soft, n_soft, bp, mm = bd.process_stream(allq, n_batches, foffs)
    dt = time.perf_counter() - t0
    nsym = int(np.sum(np.maximum(n_soft.astype(np.int64) - 1, 0))) * reps
    gb = allq.nbytes * reps / 1e9
    emit({"metric": "Msymbols/s demodulated, host-fed cu8 (PCIe inclusive)", "value": nsym / dt / 1e6,
                      "unit": "Msym/s", "n_gpus": 1, "config": {"workload": f"{n_batches} batches x {rows} carriers x {args.chunk} cu8 samples from host memory"},
          "host_to_device_GBps": gb / dt, "seconds": dt, "realtime_carriers": nsym / dt / (SAMPLE_RATE / 130)})
    bd.close()


def leg_pfb(carriers, steps, warmup):
    """Channeliser leg (BASELINE config 5): 10 MS/s cu8 -> 400 x 25 kHz channels at 80 kS/s (D = 125),
    1 048 576-sample chunks.  Algorithmic bytes: n_in*2 in + 400*n_out*8 out (SURVEY 8(d) 'PFB stage').
    Timed with HIP events on the stream the kernel runs on (a plan's stream made current); the output of the last step is
    compared with the fp64 definition's values at the pinned probe points."""
    import ctypes as C
    from tetraear_amd import _lib
    from tetraear_amd._lib import MODE_TETRA
    from tetraear_amd.batch import BatchDemodulator, DeviceBuffer
    L = _lib.load()
    M, D, n_in = PFB_M, PFB_D, int(os.environ.get("TDM_BENCH_PFB_NIN", PFB_NIN))
    n_out = (n_in + D - 1) // D
    streams = max(1, carriers // 400)
    if n_in == PFB_NIN:
        u8 = pfb_stream()
    else:   # (experiment sizes: a stream of exactly the length the kernel reads, never a slice shorter than that)
        from tetraear_amd import synth
        u8 = synth.noise_cu8(n_in, 1)
    assert len(u8) == 2 * n_in
    din = DeviceBuffer(0, streams * n_in * 2)
    pitch = (n_out + 15) // 16 * 16   # 128-byte aligned channel rows
    dout = DeviceBuffer(0, streams * M * pitch * 8)
    din.upload(np.concatenate([np.roll(u8, 2 * 977 * i) for i in range(streams)]))   # (stream 0 is the pinned one)
    no = C.c_int64()
    clock = BatchDemodulator(80000.0, 4096, 1, "cf32", mode=MODE_TETRA)   # (only its stream and event pair are used)
    clock.make_stream_current()
    try:
        def step():
            # one launch for all streams (grid.y = stream)
            _lib.check(L.tdm_channelise_batch(din.ptr, 0, n_in, streams, M, D, dout.ptr, pitch, C.byref(no), 1, 0))
        for _ in range(warmup if PMC_CHILD else max(warmup, SETTLE_STEPS)):   # (clock settling: see main())
            step()
        clock.sync()
        clock.time_begin()
        for _ in range(steps):
            step()
        ms = clock.time_end() / steps
    finally:
        clock.release_stream()
        clock.close()
    check = {"status": "no pinned probes for this workload"}
    chk = load_checks()
    if chk is not None and n_in == PFB_NIN and "pfb_channels" in chk.files:
        y0 = dout.download(np.complex64, M * pitch).reshape(M, pitch)   # stream 0
        ref, chans, stride = chk["pfb_ref"], chk["pfb_channels"], int(chk["pfb_time_stride"])
        scale = float(chk["pfb_scale"])
        err = max(float(np.max(np.abs(y0[int(k), 0:n_out:stride] - ref[i]))) for i, k in enumerate(chans)) / scale
        check = {"against": "oracle/pfb_np.py (fp64 definition), pinned by tests/golden/make_bench_checks.py",
                 "probes": int(ref.size), "channels": [int(k) for k in chans], "max_err_rel": err, "tolerance": 2e-5,
                 "status": "matches the definition at the pinned probes" if err < 2e-5 else "DIFFERS from the definition"}
    din.free()
    dout.free()
    bytes_alg = streams * (n_in * 2 + M * n_out * 8)
    traffic, traffic_src, traffic_detail = (None, None, None) if PMC_CHILD else traffic_now(
        ["--mode", "pfb", "--carriers", str(carriers)], "k_pfb_fft<", streams * n_in, "tetra-cf32", "pfb")
    return {"metric": "channeliser throughput (tetra mode, polyphase DFT filter bank)", "value": streams * n_in / (ms * 1e-3) / 1e6,
            "unit": "Msamples/s in", "n_gpus": 1, "steps": steps, "warmup": warmup, "ms_per_step": ms,
            "higher_is_better": True, "dtype": "f32", "data": "synthetic",
            "config": {"workload": f"{streams} x (10 MS/s cu8, {n_in} samples -> 400 channels x {n_out} cf32 @80 kS/s)"},
            "realtime_10MSps_streams": streams * n_in / (ms * 1e-3) / 10e6,
            "output_check": check,
            "roofline": hbm_roofline("k_pfb_fft<20,20,3,32>", bytes_alg, ms, read_bytes=streams * n_in * 2, traffic=traffic,
                                     traffic_src=traffic_src, traffic_detail=traffic_detail,
                                     timing="HIP events on the kernel's stream, one kernel per step")}


def _print_leg(out):
    """a north-star leg run on its own: the JSON line, then -- as main() does for the headline -- a non-zero exit when the
    leg's output differs from the pinned definition (a number from wrong output must not look like a result)"""
    emit(out)
    status = str(out.get("output_check", {}).get("status", ""))
    side = str(((out.get("gardner_mode") or {}).get("output_check") or {}).get("status", ""))
    if status.startswith("DIFFERS") or side.startswith("DIFFERS"):
        raise SystemExit(f"bench: output check failed: {status} {side}")


def main_pfb(args):
    _print_leg(leg_pfb(args.carriers, args.steps, args.warmup))


def leg_wideband(carriers, steps, warmup):
    """BASELINE config 5 end to end on the device: `streams` x (10 MS/s cu8, 1 048 576 samples) -> polyphase
    channeliser (400 x 80 kS/s, row pitch 8400) -> TETRA-mode demodulation of every channel (RRC, timing,
    Farrow, slicer), without host synchronisation.  The stream carries nine pi/4-DQPSK carriers on the 25 kHz grid (every
    stream of the batch is a copy of it); the hard decisions of the occupied channels are compared with the digest of the
    fp64 definition chain (oracle/pfb_np.py -> oracle/tetra_np.py), every stream against stream 0.
    Two figures: `ms_per_step_one_stream` = one batch's channeliser and demodulator back to back on one stream (HIP events;
    round 2's figure), `ms_per_step` = consecutive batches on two alternating slots, each with its own stream: the
    channeliser of batch k+1 (bound by its output stores) runs beside the demodulator of batch k (bound by instruction
    issue).  Wall clock between device synchronisations over `steps` batches, launches issued back to back."""
    from tetraear_amd.wideband import WidebandReceiver
    M, D, n_in, fs = PFB_M, PFB_D, PFB_NIN, PFB_FS
    streams = max(1, carriers // 400)
    u8, _ = wideband_stream()
    rx = WidebandReceiver(fs, n_in, M, D, streams=streams, fmt="cu8", slots=2)
    rx.d_in.upload(np.tile(u8, streams))
    bd = rx.demod
    for _ in range(warmup if PMC_CHILD else max(warmup, SETTLE_STEPS)):   # (clock settling: see main())
        rx.enqueue()
    bd.sync()
    bd.time_begin()
    for _ in range(steps):
        rx.enqueue()
    ms_one = bd.time_end() / steps
    st = bd.stage_times()
    # two slots, two streams
    for k in range(20):
        rx.enqueue(slot=k & 1)
    rx.sync()
    t0 = time.perf_counter()
    for k in range(steps):
        rx.enqueue(slot=k & 1)
    rx.sync()
    ms = (time.perf_counter() - t0) / steps * 1e3
    occupied = [int(k) for k in WIDEBAND_CHANNELS]
    chk = load_checks()
    want = str(chk["wideband_digest"]) if chk is not None and "wideband_digest" in chk.files else None
    digests, same, nsym = [], True, 0
    for _, demod in rx.slots:
        hard, soft, n_soft, bp, mm = demod.download()
        nsym = int(np.maximum(n_soft - 1, 0).sum())
        digests.append(rows_digest(hard, n_soft, occupied))
        same = same and all(rows_digest(hard, n_soft, [sidx * M + k for k in occupied]) == digests[-1] for sidx in range(1, streams))
    ok = want is not None and all(d == want for d in digests) and same
    check = {"against": "oracle/pfb_np.py -> oracle/tetra_np.py (fp64 definitions) on the nine occupied channels, pinned by tests/golden/make_bench_checks.py",
             "sha256": digests[0], "both_slots_equal": digests[0] == digests[-1], "streams_identical": bool(same),
             "status": ("matches the definition-pinned digest" if ok else
                        ("no pinned digest for this workload" if want is None else "DIFFERS from the definition-pinned digest"))}
    n_out = rx.n_out
    rx.close()
    # ---- the same chain with the occupancy gate (the reference demodulates only where its gate sees a signal,
    # ui/modern.py:1921-2022): channeliser -> tdm_occupancy_gate -> receiver over the listed rows, all on the device.
    # Both figures ride in the line; `value` stays the ungated one (every channel demodulated).
    gated = None
    try:
        hard_all, _, n_soft_all, _, _ = hard, soft, n_soft, bp, mm   # (the last slot's ungated outputs, kept for the comparison)
        rxg = WidebandReceiver(fs, n_in, M, D, streams=streams, fmt="cu8", slots=2, gated=True)
        rxg.d_in.upload(np.tile(u8, streams))
        for k in range(20):
            rxg.enqueue(slot=k & 1)
        rxg.sync()
        rxg.demod.time_begin()
        for _ in range(steps):
            rxg.enqueue()
        g_one = rxg.demod.time_end() / steps
        g_st = rxg.demod.stage_times()
        for k in range(4):
            rxg.enqueue(slot=k & 1)
        rxg.sync()
        t0 = time.perf_counter()
        for k in range(steps):
            rxg.enqueue(slot=k & 1)
        rxg.sync()
        g_ms = (time.perf_counter() - t0) / steps * 1e3
        gh, gs, gn, gb, gm = rxg.demod.download()
        _, _, occ = rxg.occupancy()
        want_rows = sorted(sidx * M + k for sidx in range(streams) for k in occupied)
        got_rows = sorted(int(r) for r in np.where(occ.reshape(-1))[0])
        same_rows = all(int(gn[r]) == int(n_soft_all[r]) and np.array_equal(gh[r, :max(int(gn[r]) - 1, 0)], hard_all[r, :max(int(gn[r]) - 1, 0)])
                        for r in want_rows)
        others_empty = int(np.count_nonzero(np.delete(gn, want_rows))) == 0
        g_digest = rows_digest(gh, gn, occupied)
        g_ok = got_rows == want_rows and same_rows and others_empty and (want is None or g_digest == want)
        g_sym = int(np.maximum(gn - 1, 0).sum())
        gated = {"ms_per_step": g_ms, "ms_per_step_one_stream": g_one, "stage_ms_per_launch": g_st,
                 "occupied_rows": len(got_rows), "rows": streams * M, "value": g_sym / (g_ms * 1e-3) / 1e6, "unit": "Msym/s of the occupied channels",
                 "realtime_10MSps_streams": streams * n_in / (g_ms * 1e-3) / fs,
                 "output_check": {"occupied_rows_are_the_transmitted_channels": got_rows == want_rows,
                                  "gated_rows_bit_identical_to_ungated": bool(same_rows), "other_rows_report_no_symbols": bool(others_empty),
                                  "sha256": g_digest,
                                  "status": "gated rows equal the ungated run's and the definition-pinned digest" if g_ok else "DIFFERS"}}
        rxg.close()
    except Exception as e:  # noqa: BLE001  (a side figure must not take the leg down; a wrong result does, below)
        gated = {"error": str(e)}
    out = {"metric": "Msymbols/s demodulated from wideband IQ (tetra mode: channeliser + per-channel demod)",
            "value": nsym / (ms * 1e-3) / 1e6, "unit": "Msym/s", "n_gpus": 1, "steps": steps, "warmup": warmup,
            "ms_per_step": ms, "ms_per_step_one_stream": ms_one, "higher_is_better": True, "dtype": "f32", "data": "synthetic",
            "config": {"workload": f"{streams} x (10 MS/s cu8, {n_in} samples, 9 pi/4-DQPSK carriers on the 25 kHz grid) -> {streams * M} channels x {n_out} cf32 -> symbols",
                       "pipelining": "consecutive batches on two slots / two streams (channeliser of batch k+1 beside the demodulator of batch k)"},
            "realtime_10MSps_streams": streams * n_in / (ms * 1e-3) / fs,
            "realtime_carriers_18ksym": nsym / (ms * 1e-3) / 18000.0,
            "output_check": check,
            "all_channels": {"ms_per_step": ms, "rows": streams * M, "note": "every channel row demodulated (the figure `value` is made of)"},
            "gated": gated,
            "stage_ms_per_launch": st, "timing": "ms_per_step: wall clock between device synchronisations, two streams; ms_per_step_one_stream and stage_ms_per_launch: HIP events on one stream", "vs_baseline": None}
    if isinstance(gated, dict) and str(gated.get("output_check", {}).get("status", "")).startswith("DIFFERS"):
        out["output_check"] = dict(check, status="DIFFERS: the gated chain (see gated.output_check)")
    return out


def main_wideband(args):
    _print_leg(leg_wideband(args.carriers, args.steps, args.warmup))


def main_tetra(args):
    _print_leg(leg_tetra(args.carriers, args.steps, args.warmup))


def leg_rrc(bd, rows, n, steps):
    """The RRC stage alone (tdm_plan_rrc_filter: LDS-tiled sliding-window matched filter, cf32 in, cf32 out at the sample rate),
    the stage `north_star`'s ">= 70 % of the HBM roofline for the RRC stage" names: launches back to back on the resident
    batch of the tetra leg, HIP events on the plan's stream; algorithmic bytes 16 per sample (SURVEY 8(d) "unfused": 8 in +
    8 out).  Its output is checked by tests/test_tetra_mode.py (every tap count, against the fp64 definition)."""
    from tetraear_amd.batch import DeviceBuffer
    pitch = (n + 1) & ~1
    ybuf = DeviceBuffer(0, rows * pitch * 8)
    try:
        for _ in range(20 if PMC_CHILD else SETTLE_STEPS):
            bd.enqueue_rrc_filter(ybuf, pitch)
        bd.sync()
        bd.time_begin()
        for _ in range(steps):
            bd.enqueue_rrc_filter(ybuf, pitch)
        ms = bd.time_end() / steps
    finally:
        ybuf.free()
    traffic, traffic_src, traffic_detail = (None, None, None) if PMC_CHILD else traffic_now(
        ["--mode", "tetra", "--carriers", str(rows)], "k_tetra_mf<", rows * n, "tetra-cf32", "tetra_mf")
    return hbm_roofline("k_tetra_mf<33> (RRC matched filter alone: LDS-tiled sliding window, 8 consecutive outputs per thread from one LDS read per sample, "
                        "taps in scalar registers, outputs transposed through LDS; cf32 in and out)",
                        rows * n * 16, ms, read_bytes=rows * n * 8, traffic=traffic, traffic_src=traffic_src, traffic_detail=traffic_detail,
                        bytes_per_sample=16, steps=steps, timing="HIP events on the plan's stream, launches back to back")


def leg_gardner(rows, base, chk, steps):
    """The same carriers through TDM_MODE_TETRA_GARDNER -- the timing recovery `north_star` names (Gardner TED + PI loop +
    period-controlled Farrow), a recurrence over a carrier's symbols run with four lanes per carrier -- timed beside the
    feed-forward receiver that is the leg's headline.  Output check: every prototype row against the decisions of the fp64
    definition's loop (oracle/tetra_np.demod_gardner via the fixture): the device's loop runs in fp32, so the count may
    differ by one symbol at the chunk's end and at most 1e-3 of the decisions (symbols the definition itself puts within
    rounding of a boundary)."""
    from tetraear_amd._lib import MODE_TETRA_GARDNER
    from tetraear_amd.batch import BatchDemodulator
    bd = BatchDemodulator(TETRA_FS, TETRA_N, rows, "cf32", mode=MODE_TETRA_GARDNER)
    bd.alloc_device_io()
    bd.upload(np.concatenate([base[i % TETRA_DISTINCT] for i in range(rows)]))
    for _ in range(max(25, steps)):     # (the loop kernel is clock-bound: at least 25 untimed passes, about 20 ms, so the clocks have settled)
        bd.enqueue()
    bd.sync()
    bd.time_begin()
    for _ in range(steps):
        bd.enqueue()
    ms = bd.time_end() / steps
    st = bd.stage_times()
    hard, soft, n_soft, tm, mm = bd.download()
    halves = int(bd.info.gardner_segments)

    def timed(b):
        for _ in range(max(25, steps)):
            b.enqueue()
        b.sync()
        b.time_begin()
        for _ in range(steps):
            b.enqueue()
        return b.time_end() / steps
    # beside it: the number of pieces FITTED TO THIS BATCH (plan option -1: the fastest for this row count on this device,
    # and the one setting under which a carrier's soft symbols depend on the plan's size -- the default's pieces follow from
    # the chunk alone), the same batch as whole chunks, and a quarter of it
    side = {"loops_per_carrier_rule": "default: the largest of 2, 4, 8 pieces the chunk's length allows (chunk-only: the same "
                                      "carrier gives the same symbols in a plan of any size)"}
    if halves > 1:
        bd.set_gardner_segments(-1)
        side["fitted_to_batch"] = {"loops_per_carrier": int(bd.info.gardner_segments), "ms_per_step": timed(bd),
                                   "note": "tdm_plan_option gardner_segments = -1: fewer, longer pieces when the batch fills the device by itself"}
        bd.set_gardner_segments(0)
        side["whole_chunks_ms_per_step"] = timed(bd)
    bd.close()
    if rows >= 4 * TETRA_DISTINCT:
        bq = BatchDemodulator(TETRA_FS, TETRA_N, rows // 4, "cf32", mode=MODE_TETRA_GARDNER)
        bq.alloc_device_io()
        bq.upload(np.concatenate([base[i % TETRA_DISTINCT] for i in range(rows // 4)]))
        side["quarter_batch"] = {"carriers": rows // 4, "loops_per_carrier": int(bq.info.gardner_segments), "ms_per_step": timed(bq)}
        bq.close()
    nsym = int(np.sum(np.maximum(n_soft.astype(np.int64) - 1, 0)))
    check = {"status": "no pinned decisions for this workload"}
    if chk is not None and "gardner_hard" in chk.files and rows >= TETRA_DISTINCT:
        ref_n, ref_h = chk["gardner_n_sym"], chk["gardner_hard"]
        worst, ok = 0.0, True
        for r in range(TETRA_DISTINCT):
            ns = int(n_soft[r])
            m = min(ns, int(ref_n[r])) - 1
            frac = float(np.mean(hard[r, :m] != ref_h[r, :m]))
            worst = max(worst, frac)
            ok = ok and abs(ns - int(ref_n[r])) <= 1 and frac <= 1e-3
        same = all(rows_digest(hard, n_soft, [r]) == rows_digest(hard, n_soft, [r % TETRA_DISTINCT]) for r in range(TETRA_DISTINCT, rows))
        check = {"against": "oracle/tetra_np.py demod_gardner (fp64 loop), pinned by tests/golden/make_bench_checks.py",
                 "worst_fraction_of_differing_decisions": worst, "rows_equal_their_prototype": bool(same),
                 "status": "decisions match the definition's loop (<= 1e-3 differing, count within one)" if (ok and same) else "DIFFERS from the definition"}
    return {"what": "TDM_MODE_TETRA_GARDNER: matched filter (producer wavefronts) -> LDS ring -> Gardner TED + PI loop + Farrow, four lanes per carrier, one kernel -> decisions"
                    + (" (2 launches)" if halves == 1 else "; every carrier's chunk as %d independently started loops (each later one starts at a feed-forward "
                       "timing estimate and warms up over 384 symbols before the seam it takes over at), joined by a copy of the later pieces' symbols (3 launches)" % halves),
            "loops_per_carrier": halves, **side,

            "ms_per_step": ms, "value": nsym / (ms * 1e-3) / 1e6, "unit": "Msym/s", "steps": steps, "stage_ms_per_launch": st,
            "output_check": check}


def leg_tetra_int8(rows, base, steps):
    """The same carriers as cu8 bytes (north_star: "coalesced complex-int8/float loads"; round 6): quantised at a quarter of full
    scale, demodulated by a cu8 TDM_MODE_TETRA plan -- 2 instead of 8 bytes per sample in, ONE bf16 plane per component, two
    matrix-core products per step instead of four.  Algorithmic bytes R*2 + 8 + 1 per symbol (17 at R = 4).  Output check: every
    row's decisions equal to those of the cf32 kernel on the samples the bytes mean (u / 127.5 - 1), soft symbols within 1e-5."""
    from tetraear_amd import synth
    from tetraear_amd._lib import MODE_TETRA
    from tetraear_amd.batch import BatchDemodulator
    fs, n = TETRA_FS, TETRA_N
    raws, xqs = [], []
    for x in base:
        x = np.asarray(x).astype(np.complex128)
        raw = synth.quantise_cu8(x / (4.0 * np.max(np.abs(x))), scale=1.0)
        raws.append(raw)
        xqs.append(synth.cu8_to_c128(raw).astype(np.complex64))
    outs = {}
    for fmt, rows_in in (("cf32", xqs), ("cu8", raws)):
        bd = BatchDemodulator(fs, n, rows, fmt, mode=MODE_TETRA)
        bd.alloc_device_io()
        bd.upload(np.concatenate([rows_in[i % TETRA_DISTINCT] for i in range(rows)]))
        for _ in range(max(25, steps)):
            bd.enqueue()
        bd.sync()
        bd.time_begin()
        for _ in range(steps):
            bd.enqueue()
        ms = bd.time_end() / steps
        outs[fmt] = (ms, bd.stage_times(), bd.download())
        if fmt == "cu8":
            # the RRC stage alone on the bytes: 2 B in + 8 B out per sample
            from tetraear_amd.batch import DeviceBuffer
            pitch = (n + 1) & ~1
            ybuf = DeviceBuffer(0, rows * pitch * 8)
            try:
                for _ in range(max(25, steps)):
                    bd.enqueue_rrc_filter(ybuf, pitch)
                bd.sync()
                bd.time_begin()
                for _ in range(steps):
                    bd.enqueue_rrc_filter(ybuf, pitch)
                rrc8_ms = bd.time_end() / steps
            finally:
                ybuf.free()
        bd.close()
    ms8, st8, (h8, s8, n8, t8, m8) = outs["cu8"]
    msf, stf, (hf, sf, nf, tf, mf) = outs["cf32"]
    same = bool(np.array_equal(n8, nf)) and all(np.array_equal(h8[r, :max(int(n8[r]) - 1, 0)], hf[r, :max(int(nf[r]) - 1, 0)]) for r in range(rows))
    err = max(float(np.max(np.abs(s8[r, :n8[r]] - sf[r, :nf[r]])) / np.max(np.abs(sf[r, :nf[r]]))) for r in range(min(rows, TETRA_DISTINCT))) if same else float("nan")
    nsym = int(np.sum(np.maximum(n8.astype(np.int64) - 1, 0)))
    k_ms = st8.get("tetra_fused", ms8)
    bytes_alg = rows * n * 2 + int(np.sum(n8.astype(np.int64))) * 9
    return {"what": "the fused receiver on cu8 input: bytes converted where the window is staged, one exact bf16 plane per component, "
                    "two matrix-core products per step instead of four",
            "ms_per_step": ms8, "value": nsym / (ms8 * 1e-3) / 1e6, "unit": "Msym/s", "cf32_on_the_same_samples_ms_per_step": msf,
            "output_check": {"against": "the cf32 kernel on the dequantised samples (u / 127.5 - 1), every row", "soft_max_err_rel": err,
                             "status": "decisions equal to the cf32 kernel's on the same samples, soft within 1e-5" if (same and err < 1e-5) else "DIFFERS from the cf32 kernel"},
            "rrc_stage": hbm_roofline("k_tetra_mf<33, cu8> (the RRC stage alone on cu8 input: bytes converted where the window is staged, cf32 out)",
                                      rows * n * 10, rrc8_ms, read_bytes=rows * n * 2, bytes_per_sample=10),
            "roofline": hbm_roofline("k_tetra_fused<33, cu8>", bytes_alg, k_ms, read_bytes=rows * n * 2, bytes_per_symbol=bytes_alg / max(nsym, 1),
                                     note="2 + 2.25 algorithmic bytes per input sample: with a quarter of the input bytes the kernel is bound by "
                                          "instruction issue, not by HBM -- the fraction is reported all the same")}


def leg_tetra(carriers, steps, warmup):
    """TETRA-mode leg (no reference oracle; SURVEY 8(d) 'tetra mode'): `carriers` channelised carriers,
    cf32 at 72 kS/s (4 samples/symbol), chunks of 32768 samples.  ONE kernel (matched filter -> timing ->
    Farrow -> carrier offset -> decisions), HBM-bound: algorithmic bytes = the input once + soft + hard symbols,
    R*8 + 8 + 1 B/symbol (SURVEY 8(d) 'fused': 41 B/symbol at R = 4).  The output of the last step is checked: hard
    decisions of every row against the digest of the fp64 definition's, soft symbols at the pinned probes within 1e-5."""
    from tetraear_amd._lib import MODE_TETRA
    from tetraear_amd.batch import BatchDemodulator
    fs, n = TETRA_FS, TETRA_N
    rows = carriers
    bd = BatchDemodulator(fs, n, rows, "cf32", mode=MODE_TETRA)
    bd.alloc_device_io()
    base = tetra_rows()
    bd.upload(np.concatenate([base[i % TETRA_DISTINCT] for i in range(rows)]))
    for _ in range(warmup if PMC_CHILD else max(warmup, SETTLE_STEPS)):   # (clock settling: see main())
        bd.enqueue()
    bd.sync()
    bd.time_begin()
    t0 = time.perf_counter()
    for _ in range(steps):
        bd.enqueue()
    ev_ms = bd.time_end()
    bd.sync()
    dt = time.perf_counter() - t0
    st = bd.stage_times()
    hard, soft, n_soft, tm, mm = bd.download()
    nsym = int(np.sum(np.maximum(n_soft.astype(np.int64) - 1, 0)))
    rrc_ms = st.get("tetra_fused", float("nan"))
    bytes_alg = rows * n * 8 + int(np.sum(n_soft.astype(np.int64))) * 9
    traffic, traffic_src, traffic_detail = (None, None, None) if PMC_CHILD else traffic_now(
        ["--mode", "tetra", "--carriers", str(rows)], "k_tetra_fused<", rows * n, "tetra-cf32", "tetra_fused")
    # ---- output check against the definition's pinned results
    d8 = [rows_digest(hard, n_soft, [r]) for r in range(min(rows, TETRA_DISTINCT))]
    same = all(rows_digest(hard, n_soft, [r]) == d8[r % TETRA_DISTINCT] for r in range(TETRA_DISTINCT, rows))
    chk = load_checks()
    check = {"status": "no pinned digest for this workload"}
    if chk is not None and "tetra_digests" in chk.files and rows >= TETRA_DISTINCT:
        want = [str(x) for x in chk["tetra_digests"]]
        idx, ref = chk["tetra_soft_idx"], chk["tetra_soft_ref"]
        err = max(float(np.max(np.abs(soft[r, idx[r]] - ref[r])) / np.max(np.abs(ref[r]))) for r in range(TETRA_DISTINCT))
        ok = want == d8 and same and err < 1e-5
        check = {"against": "oracle/tetra_np.py (fp64 definition, unquantised RRC for the soft probes), pinned by tests/golden/make_bench_checks.py",
                 "hard_sha256_row0": d8[0], "rows_equal_their_prototype": bool(same), "soft_probes": int(idx.size),
                 "soft_max_err_rel": err, "soft_tolerance": 1e-5,
                 "status": "hard decisions match the definition-pinned digests, soft within 1e-5" if ok else "DIFFERS from the definition"}
    try:
        rrc_stage = leg_rrc(bd, rows, n, 3 if PMC_CHILD else steps)
    except Exception as e:  # noqa: BLE001
        rrc_stage = {"error": str(e)}
    gardner = None
    if not PMC_CHILD:
        try:
            gardner = leg_gardner(rows, base, chk, max(5, steps // 5))
        except Exception as e:  # noqa: BLE001 -- a side leg never breaks the line
            gardner = {"error": str(e)}
    int8 = None
    if not PMC_CHILD:
        try:
            int8 = leg_tetra_int8(rows, base, max(5, steps // 5))
        except Exception as e:  # noqa: BLE001 -- a side leg never breaks the line
            int8 = {"error": str(e)}
    out = {"metric": "Msymbols/s demodulated (TETRA mode: RRC + feed-forward timing + Farrow + quadrant slicer)",
           "value": nsym * steps / dt / 1e6, "unit": "Msym/s", "n_gpus": 1, "steps": steps,
           "warmup": warmup, "ms_per_step": dt / steps * 1e3, "higher_is_better": True, "scaling": "weak",
           "vs_baseline": None, "dtype": "f32 (matched filter: split-bf16 products accumulated in fp32; soft symbols max 6.2e-6 of full scale from the unquantised fp64 definition over 2048 carriers, tests/test_tetra_precision.py)", "data": "synthetic",
           "config": {"workload": f"{rows} channelised 25 kHz carriers, cf32 @72 kS/s, {n}-sample chunks", "mode": "tetra"},
           "realtime_carriers": nsym * steps / dt / 18000.0, "event_ms_per_step": ev_ms / steps,
           "stage_ms_per_launch": st,
           "output_check": check,
           "rrc_stage": rrc_stage,
           "gardner_mode": gardner,
           "int8_input": int8,
           "roofline": hbm_roofline("k_tetra_fused<33> (RRC matched filter on the matrix cores (split-bf16 products, fp32 accumulate) -> timing -> Farrow -> slicer, one pass over the input)",
                                    bytes_alg, rrc_ms, read_bytes=rows * n * 8, traffic=traffic, traffic_src=traffic_src,
                                    traffic_detail=traffic_detail, bytes_per_symbol=bytes_alg / max(nsym, 1))}
    bd.close()
    return out


if __name__ == "__main__":
    main()
