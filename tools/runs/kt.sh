#!/bin/bash
# usage: bash tools/runs/kt.sh <tag> [bench args]: rocprofv3 kernel-trace stats of a short bench run -> gpurun_out/<tag>_kernel_stats.csv
tag=$1; shift
export TMPDIR=/tmp
out=gpurun_out/kt_$tag
mkdir -p $out
B="--no-cpu-baseline --no-end-to-end --no-shaped-leg --gen-workers 1 --no-tree-compare"
rocprofv3 --kernel-trace --stats -d $out/kt -o kt --output-format csv -- python3 bench.py --steps 3 --warmup 1 $B "$@" > $out/kt_bench_line.json 2> $out/kt.log
cp $(find $out/kt -name '*kernel_stats.csv') gpurun_out/${tag}_kernel_stats.csv
head -25 gpurun_out/${tag}_kernel_stats.csv | cut -c1-200
