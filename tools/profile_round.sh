#!/bin/bash
# Runs on the GPU box (via gpurun): bench JSON + rocprofv3 kernel stats + HBM traffic counters.
# usage: tools/profile_round.sh <tag> [bench args...]
set -u
TAG=$1; shift
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
python tools/hbm_ceiling.py $OUT/hbm_ceiling.json > /dev/null 2>&1
python bench.py "$@" > $OUT/bench.json 2> $OUT/bench.err
tail -c 600 $OUT/bench.json
# per-kernel time (same command, fewer steps)
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o trace -- python bench.py "$@" --no-cpu-baseline --no-extra > $OUT/trace_bench.json 2> $OUT/trace.err
# HBM traffic counters, separate passes (TCC slots: FETCH_SIZE 3, WRITE_SIZE 2)
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/pmc_fetch -o fetch -- python bench.py "$@" --steps 2 --warmup 1 --no-cpu-baseline --no-extra > /dev/null 2> $OUT/pmc_fetch.err
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/pmc_write -o write -- python bench.py "$@" --steps 2 --warmup 1 --no-cpu-baseline --no-extra > /dev/null 2> $OUT/pmc_write.err
find $OUT -name "*.csv" | head -20
# tetra-mode legs (no reference oracle): RRC stage HBM roofline, channeliser
python bench.py --mode tetra --carriers 4096 --steps 100 --warmup 60 > $OUT/bench_tetra.json 2> $OUT/bench_tetra.err
python bench.py --mode pfb --carriers 12800 --steps 100 --warmup 60 > $OUT/bench_pfb.json 2> $OUT/bench_pfb.err
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_tetra -o tetra -- python bench.py --mode tetra --carriers 4096 --steps 100 --warmup 60 > /dev/null 2> $OUT/trace_tetra.err
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/pmc_tetra_fetch -o fetch -- python bench.py --mode tetra --carriers 4096 --steps 2 --warmup 1 > /dev/null 2> $OUT/pmc_tetra_fetch.err
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/pmc_tetra_write -o write -- python bench.py --mode tetra --carriers 4096 --steps 2 --warmup 1 > /dev/null 2> $OUT/pmc_tetra_write.err
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_pfb -o pfb -- python bench.py --mode pfb --carriers 12800 --steps 100 --warmup 60 > /dev/null 2> $OUT/trace_pfb.err
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/pmc_pfb_fetch -o fetch -- python bench.py --mode pfb --carriers 12800 --steps 2 --warmup 1 > /dev/null 2> $OUT/pmc_pfb_fetch.err
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/pmc_pfb_write -o write -- python bench.py --mode pfb --carriers 12800 --steps 2 --warmup 1 > /dev/null 2> $OUT/pmc_pfb_write.err
python bench.py --carriers 1 --no-cpu-baseline --no-extra > $OUT/bench_single.json 2> /dev/null
python bench.py --shared --carriers 64 --no-cpu-baseline --no-extra > $OUT/bench_shared64.json 2> /dev/null
python bench.py --shared --carriers 64 --chunks 4 --no-cpu-baseline --no-extra > $OUT/bench_shared64_chunks4.json 2> /dev/null
for c in 128 256 512; do python bench.py --carriers $c --no-cpu-baseline --no-extra > $OUT/bench_carriers$c.json 2> /dev/null; done
python bench.py --depth 1 --no-cpu-baseline --no-extra > $OUT/bench_depth1.json 2> /dev/null
TDM_FORCE_DIST=1 MASTER_PORT=29517 python bench.py --no-cpu-baseline --no-extra > $OUT/bench_forcedist_rccl.json 2> /dev/null
python bench.py --no-cpu-baseline --no-extra --total-carriers 1024 > $OUT/bench_strong1024.json 2> /dev/null
python bench.py --mode wideband --carriers 12800 --steps 100 --warmup 60 > $OUT/bench_wideband.json 2> /dev/null
python bench.py --carriers 256 --fmt cf64 --no-cpu-baseline --no-extra > $OUT/bench_cf64_256.json 2> /dev/null
tail -c 300 $OUT/bench_tetra.json; echo; tail -c 200 $OUT/bench_single.json
