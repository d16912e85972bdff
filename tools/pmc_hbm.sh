#!/bin/bash
# HBM traffic of one bench command (separate passes; FETCH_SIZE needs the x2 gfx950 correction).  usage: tools/pmc_hbm.sh <tag> <bench args...>
set -u
TAG=$1; shift
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/f -o f -- python bench.py "$@" --steps 2 --warmup 1 > /dev/null 2> $OUT/f.err
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/w -o w -- python bench.py "$@" --steps 2 --warmup 1 > /dev/null 2> $OUT/w.err
rocprofv3 --pmc TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum --kernel-trace --output-format csv -d $OUT/e -o e -- python bench.py "$@" --steps 2 --warmup 1 > /dev/null 2> $OUT/e.err
python - <<PY
import csv, glob, collections
for pat in ("$OUT/f/**/*counter_collection.csv", "$OUT/w/**/*counter_collection.csv", "$OUT/e/**/*counter_collection.csv"):
    for f in glob.glob(pat, recursive=True):
        acc = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"][:50]
            acc[k][r["Counter_Name"]] += float(r["Counter_Value"]); n[(k, r["Counter_Name"])] += 1
        for k, d in acc.items():
            if "rocclr" in k: continue
            print(k, {c: f"{v / n[(k, c)]:.5g} per launch" for c, v in d.items()})
PY
