#!/bin/bash
# SQ issue/stall breakdown of any command (runs on the GPU box).  usage: tools/pmc_sq_cmd.sh <tag> <command...>
set -u
TAG=$1; shift
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --kernel-trace --output-format csv -d $OUT/sq -o sq -- "$@" > /dev/null 2> $OUT/sq.err
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVES SQ_BUSY_CYCLES SQ_WAIT_INST_LDS --kernel-trace --output-format csv -d $OUT/sq2 -o sq2 -- "$@" > /dev/null 2> $OUT/sq2.err
python - <<PY
import csv, glob, collections
for pat in ("$OUT/sq/**/*counter_collection.csv", "$OUT/sq2/**/*counter_collection.csv"):
    for f in glob.glob(pat, recursive=True):
        acc = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"][:60]
            acc[k][r["Counter_Name"]] += float(r["Counter_Value"]); n[(k, r["Counter_Name"])] += 1
        for k, d in acc.items():
            print(k, {c: f"{v / n[(k, c)]:.4g}" for c, v in d.items()})
PY
