#!/bin/bash
# Runs on the GPU box (via gpurun): bench JSON + rocprofv3 kernel stats + HBM traffic counters.
# usage: tools/profile_round.sh <tag> [bench args...]
set -u
TAG=$1; shift
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
python bench.py "$@" > $OUT/bench.json 2> $OUT/bench.err
tail -c 600 $OUT/bench.json
# per-kernel time (same command, fewer steps)
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o trace -- python bench.py "$@" --steps 5 --warmup 2 --no-cpu-baseline > $OUT/trace_bench.json 2> $OUT/trace.err
# HBM traffic counters, separate passes (TCC slots: FETCH_SIZE 3, WRITE_SIZE 2)
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/pmc_fetch -o fetch -- python bench.py "$@" --steps 2 --warmup 1 --no-cpu-baseline > /dev/null 2> $OUT/pmc_fetch.err
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/pmc_write -o write -- python bench.py "$@" --steps 2 --warmup 1 --no-cpu-baseline > /dev/null 2> $OUT/pmc_write.err
find $OUT -name "*.csv" | head -20
