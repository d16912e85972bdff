#!/usr/bin/env python3
"""Turn the raw output of tools/profile_round.sh (gpurun_out/<tag>/) into the committed summaries under
profiles/: bench lines, rocprofv3 kernel-stats tables and the HBM traffic figures (FETCH_SIZE x2 on gfx950,
WRITE_SIZE as is, counter unit KB; separate passes).   usage: tools/summarize_profile.py <tag>"""
import collections
import csv
import glob
import json
import os
import shutil
import sys

HERE = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def counters(path):
    """mean counter value per launch and kernel from a *_counter_collection.csv"""
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for f in glob.glob(os.path.join(path, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            acc[r["Kernel_Name"]][r["Counter_Name"]].append(float(r["Counter_Value"]))
    return {k: {c: sum(v) / len(v) for c, v in d.items()} for k, d in acc.items()}


def traffic(fetch_dir, write_dir, match, samples):
    f, w = counters(fetch_dir), counters(write_dir)
    name = next((k for k in f if match in k), None)
    if name is None or name not in w:
        return None
    fb = f[name]["FETCH_SIZE"] * 1024.0
    wb = w[name]["WRITE_SIZE"] * 1024.0
    return {"name": name, "fetch_bytes_raw": fb, "fetch_bytes_corrected_x2": 2 * fb, "write_bytes": wb,
            "hbm_bytes_per_launch": 2 * fb + wb, "input_samples_per_launch": samples,
            "hbm_bytes_per_input_sample": (2 * fb + wb) / samples}


def main():
    tag = sys.argv[1]
    src = os.path.join(HERE, "gpurun_out", tag)
    dst = os.path.join(HERE, "profiles")
    os.makedirs(dst, exist_ok=True)
    for name in ("hbm_ceiling", "bench", "bench_forcedist_rccl", "bench_strong1024", "bench_tetra", "bench_pfb", "bench_single", "bench_cf64_256", "bench_shared64", "bench_wideband",
                 "bench_shared64_chunks4", "bench_carriers128", "bench_carriers256", "bench_carriers512", "bench_depth1"):
        p = os.path.join(src, name + ".json")
        if os.path.exists(p) and os.path.getsize(p) > 0:
            shutil.copy(p, os.path.join(dst, f"{tag}_{name}.json"))
    for sub, out in (("trace", "reference"), ("trace_tetra", "tetra"), ("trace_pfb", "pfb")):
        for f in glob.glob(os.path.join(src, sub, "**", "*kernel_stats.csv"), recursive=True):
            shutil.copy(f, os.path.join(dst, f"{tag}_{out}_kernel_stats.csv"))
    # per-kernel launch durations in launch order (rocprofv3 --stats averages every launch of the process, the untimed
    # warm-up ones with their clock ramp included): the same trace, split into the warm-up and the timed launches
    for sub, out, warm in (("trace", "reference", 150), ("trace_tetra", "tetra", 150), ("trace_pfb", "pfb", 150)):
        for f in glob.glob(os.path.join(src, sub, "**", "*kernel_trace.csv"), recursive=True):
            per = collections.defaultdict(list)
            for r in csv.DictReader(open(f)):
                per[r["Kernel_Name"]].append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]) - int(r["Start_Timestamp"])))
            rows = {}
            for k, v in per.items():
                d = [x[1] / 1e6 for x in sorted(v)]
                if len(d) <= warm:
                    continue
                steps = (len(d) - warm) // 2 if out == "reference" else 0   # reference bench: timed region, then the per-kernel pass
                timed = d[warm:warm + steps] if steps else d[warm:]
                rows[k] = {"launches": len(d), "warmup_launches": warm, "warmup_avg_ms": sum(d[:warm]) / warm,
                           "warmup_max_ms": max(d[:warm]), "timed_avg_ms": sum(timed) / len(timed),
                           "timed_min_ms": min(timed), "timed_max_ms": max(timed), "first_10_ms": [round(x, 4) for x in d[:10]]}
                if steps:
                    alone = d[warm + steps:warm + 2 * steps]
                    rows[k].update({"per_kernel_pass_avg_ms": sum(alone) / len(alone), "per_kernel_pass_min_ms": min(alone),
                                    "per_kernel_pass_max_ms": max(alone)})
            with open(os.path.join(dst, f"{tag}_{out}_kernel_timed.json"), "w") as fo:
                json.dump({"note": "launch durations from the rocprofv3 kernel trace of the bench command, in launch order: "
                                   "the first `warmup_launches` are bench.py's untimed settling + warm-up passes (GPU clocks ramp for ~40 "
                                   "launches after idle); `timed_*` are the launches of the timed region -- since round 6 up to three steps "
                                   "are in flight there (PipelinedBatchDemodulator), so a launch shares the device with the neighbouring "
                                   "steps' kernels and takes longer than alone; `per_kernel_pass_*` are the same number of steps again with "
                                   "the steps ordered one after the other on the device (tdm_plan_wait_for): every launch alone -- the "
                                   "figure bench.py's `roofline.avg_launch_ms` / `stage_ms_per_launch` report from HIP events",
                           "kernels": rows}, fo, indent=1)
    prof = {"command": "rocprofv3 --pmc FETCH_SIZE | --pmc WRITE_SIZE (separate passes) --kernel-trace -- python bench.py "
                       "[--mode tetra --carriers 4096 | --mode pfb --carriers 12800] --steps 2 --warmup 1",
            "note": "gfx950: FETCH_SIZE counts 64 B per 128-B request for wide coalesced reads (MI355X_MICROARCH.md, HBM "
                    "section) -> doubled; WRITE_SIZE taken as is; counter unit KB"}
    t = traffic(os.path.join(src, "pmc_fetch"), os.path.join(src, "pmc_write"), "k_pz_raw<10, 12, 27", 1024 * 262144)
    if t:
        prof["k1"] = t
    t = traffic(os.path.join(src, "pmc_fetch"), os.path.join(src, "pmc_write"), "k_lp2<", 1024 * 26215)
    if t:
        prof["lp2"] = t
    t = traffic(os.path.join(src, "pmc_tetra_fetch"), os.path.join(src, "pmc_tetra_write"), "k_tetra_fused", 4096 * 32768)
    if t:
        prof["tetra_fused"] = t
    t = traffic(os.path.join(src, "pmc_pfb_fetch"), os.path.join(src, "pmc_pfb_write"), "k_pfb_fft", 32 * 1048576)
    if t:
        prof["pfb"] = t
    prof["all_reference"] = {k: {c + "_KB": v for c, v in d.items()} for k, d in
                             {**counters(os.path.join(src, "pmc_fetch"))}.items()}
    for k, d in counters(os.path.join(src, "pmc_write")).items():
        prof["all_reference"].setdefault(k, {}).update({c + "_KB": v for c, v in d.items()})
    with open(os.path.join(dst, f"{tag}_pmc_traffic.json"), "w") as f:
        json.dump(prof, f, indent=1)
    print(json.dumps({k: (v.get("hbm_bytes_per_input_sample") if isinstance(v, dict) else None)
                      for k, v in prof.items() if k in ("k1", "tetra_fused", "pfb")}))


if __name__ == "__main__":
    main()
