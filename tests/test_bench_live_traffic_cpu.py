"""CPU tier: bench.py's live HBM-traffic measurement (two rocprofv3 counter passes over a child run of the bench) against a
stand-in `rocprofv3` that writes counter CSVs of the shape the real one does: the parsing, the gfx950 corrections
(FETCH_SIZE x 2, unit KB), and every way the measurement declines (no tool, nested profiler, failed pass, ambiguous
kernel) -- in which case the bench falls back to the last committed profile and says so."""
import os
import stat
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

FAKE = r'''#!/usr/bin/env python3
import os, sys
a = sys.argv[1:]
counter = a[a.index("--pmc") + 1]
d = a[a.index("-d") + 1]
child = a[a.index("--") + 1:]
assert "--pmc-child" in child and "--steps" in child, child
assert os.environ.get("TDM_BENCH_LIVE_PMC") == "0"
if os.environ.get("FAKE_ROCPROF_FAIL") == counter:
    sys.exit(3)
os.makedirs(os.path.join(d, "host", "123"), exist_ok=True)
rows = [("void tdm::k_pz_raw<10, 12, 27, 0>(tdm::ZpParams, void const*, long, int)", {"FETCH_SIZE": 1000.0, "WRITE_SIZE": 500.0}),
        ("void tdm::k_pz_raw<10, 12, 27, 0>(tdm::ZpParams, void const*, long, int)", {"FETCH_SIZE": 1200.0, "WRITE_SIZE": 700.0}),
        ("void tdm::k_lp2<tdm::Lp2SrcDec>(tdm::Lp2Params, tdm::Lp2SrcDec)", {"FETCH_SIZE": 9.0, "WRITE_SIZE": 9.0})]
if os.environ.get("FAKE_ROCPROF_TWO_KERNELS"):
    rows.append(("void tdm::k_pz_raw<8, 15, 27, 0>(tdm::ZpParams, void const*, long, int)", {"FETCH_SIZE": 1.0, "WRITE_SIZE": 1.0}))
with open(os.path.join(d, "host", "123", "c_counter_collection.csv"), "w") as f:
    f.write("Correlation_Id,Dispatch_Id,Agent_Id,Queue_Id,Process_Id,Thread_Id,Grid_Size,Kernel_Id,Kernel_Name,Workgroup_Size,LDS_Block_Size,Scratch_Size,VGPR_Count,Accum_VGPR_Count,SGPR_Count,Counter_Name,Counter_Value,Start_Timestamp,End_Timestamp\n")
    for i, (k, v) in enumerate(rows):
        f.write(f'{i},{i},1,1,1,1,64,1,"{k}",64,0,0,256,0,96,{counter},{v[counter]},0,1\n')
'''


@pytest.fixture
def fake_rocprof(tmp_path, monkeypatch):
    p = tmp_path / "rocprofv3"
    p.write_text(FAKE)
    p.chmod(p.stat().st_mode | stat.S_IEXEC)
    monkeypatch.setenv("PATH", str(tmp_path) + os.pathsep + os.environ["PATH"])
    for k in list(os.environ):
        if k.startswith(("ROCPROF", "ROCP_")):
            monkeypatch.delenv(k)
    monkeypatch.delenv("TDM_BENCH_LIVE_PMC", raising=False)
    return p


def test_live_traffic_parses_counters_and_applies_the_gfx950_corrections(fake_rocprof):
    import bench
    r = bench.live_traffic(["--carriers", "4"], "k_pz_raw<")
    assert r is not None and r["kernel"].startswith("void tdm::k_pz_raw<10, 12, 27")
    assert r["launches_averaged"] == [2, 2]
    assert r["fetch_bytes_raw"] == 1100.0 * 1024 and r["write_bytes"] == 600.0 * 1024
    assert r["hbm_bytes_per_launch"] == 2 * 1100.0 * 1024 + 600.0 * 1024
    t, src, detail = bench.traffic_now(["--carriers", "4"], "k_pz_raw<", 268435456, "cu8", "k1")
    assert t == r["hbm_bytes_per_launch"] and src.startswith("live") and detail["kernel"] == r["kernel"]


def test_live_traffic_declines_and_the_bench_falls_back_to_the_committed_profile(fake_rocprof, monkeypatch):
    import bench
    monkeypatch.setenv("FAKE_ROCPROF_FAIL", "WRITE_SIZE")            # a pass fails
    assert bench.live_traffic([], "k_pz_raw<") is None
    monkeypatch.delenv("FAKE_ROCPROF_FAIL")
    monkeypatch.setenv("FAKE_ROCPROF_TWO_KERNELS", "1")              # the name matches two kernels: not a measurement
    assert bench.live_traffic([], "k_pz_raw<") is None
    monkeypatch.delenv("FAKE_ROCPROF_TWO_KERNELS")
    assert bench.live_traffic([], "k_no_such_kernel<") is None
    monkeypatch.setenv("ROCPROFILER_SOMETHING", "1")                 # this process is itself being profiled
    assert bench.live_traffic([], "k_pz_raw<") is None
    monkeypatch.delenv("ROCPROFILER_SOMETHING")
    monkeypatch.setenv("TDM_BENCH_LIVE_PMC", "0")                    # switched off (the child of a measuring bench)
    assert bench.live_traffic([], "k_pz_raw<") is None
    t, src, detail = bench.traffic_now([], "k_pz_raw<", 268435456, "cu8", "k1")
    assert detail is None and t is not None and "committed profile" in src and src.endswith("live PMC passes unavailable)")
    monkeypatch.delenv("TDM_BENCH_LIVE_PMC")
    monkeypatch.setenv("PATH", "/nonexistent")                       # no rocprofv3 at all
    assert bench.live_traffic([], "k_pz_raw<") is None
