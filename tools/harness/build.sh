#!/bin/bash
# builds harness variants into tools/harness/bin/ (git-ignored; travels to the GPU box)
# usage: build.sh <name> <kernel header path> [extra flags...]
set -e
cd "$(dirname "$0")"
mkdir -p bin
NAME=$1; HDR=$2; shift 2
/opt/rocm/bin/hipcc -w -O3 -std=c++17 --offload-arch=gfx950 -mllvm -pragma-unroll-threshold=131072 -DTDM_KERNEL_HEADER="\"$HDR\"" "$@" -o bin/$NAME tetra_bench.hip
