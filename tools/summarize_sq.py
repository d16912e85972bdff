#!/usr/bin/env python3
"""Turn the raw output of tools/pmc_sq.sh (gpurun_out/<tag>/sq, sq2: two rocprofv3 --pmc passes) into profiles/<name>.json:
per-launch averages per kernel plus the derived figures DESIGN.md quotes.  usage: tools/summarize_sq.py <tag> <name> [note]"""
import collections
import csv
import glob
import json
import os
import sys

HERE = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag, name = sys.argv[1], sys.argv[2]
note = sys.argv[3] if len(sys.argv) > 3 else ""
short = {"k_pz_raw": "k_pz_raw", "k_pz_carry": "k_pz_carry", "k_lp2": "k_lp2", "k_finish": "k_finish", "k_pz_block": "k_pz_block"}
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for sub in ("sq", "sq2"):
    for f in glob.glob(os.path.join(HERE, "gpurun_out", tag, sub, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            k = next((v for s, v in short.items() if s in r["Kernel_Name"]), None)
            if k:
                acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
out = {}
for k, d in acc.items():
    row = {c: sum(v) / len(v) for c, v in d.items()}
    row["launches"] = max(len(v) for v in d.values())
    if row.get("SQ_WAVE_CYCLES"):
        row["frac_wave_cycles_waiting_any"] = row.get("SQ_WAIT_ANY", 0.0) / row["SQ_WAVE_CYCLES"]
        row["frac_wave_cycles_waiting_to_issue"] = row.get("SQ_WAIT_INST_ANY", 0.0) / row["SQ_WAVE_CYCLES"]
    if row.get("SQ_WAVES"):
        row["valu_insts_per_wave"] = row.get("SQ_INSTS_VALU", 0.0) / row["SQ_WAVES"]
    if row.get("SQ_BUSY_CYCLES"):
        row["valu_active_per_busy_cycle"] = row.get("SQ_ACTIVE_INST_VALU", 0.0) / row["SQ_BUSY_CYCLES"]
    if row.get("SQ_LDS_IDX_ACTIVE"):
        row["lds_bank_conflict_per_lds_active_cycle"] = row.get("SQ_LDS_BANK_CONFLICT", 0.0) / row["SQ_LDS_IDX_ACTIVE"]
    out[k] = row
with open(os.path.join(HERE, "profiles", name + ".json"), "w") as f:
    json.dump({"note": "rocprofv3 --pmc SQ_* (two passes, --kernel-trace only) over a 2-step run of the bench command on one MI355X; "
                       "per-launch averages; tools/pmc_sq.sh " + tag + ("; " + note if note else ""), "kernels": out}, f, indent=1)
print({k: {c: round(v, 3) for c, v in r.items() if c.startswith(("valu_", "frac_", "lds_"))} for k, r in out.items()})
