#!/bin/bash
# usage (GPU box, repo root): bash tools/runs/trace_step.sh <tag>   -- a kernel trace of three bench steps and where the GPU idles inside the last one (tools/step_timeline.py)
tag=$1
export TMPDIR=/tmp
out=gpurun_out/trace_$tag
mkdir -p $out
env | grep -i -E "^(HSA|ROC|HIP|GPU|AMD)" > $out/env.txt
rocprofv3 --kernel-trace -d $out/kt -o kt --output-format csv -- python3 bench.py --steps 3 --warmup 2 --no-cpu-baseline --no-end-to-end --no-shaped-leg --gen-workers 1 --no-tree-compare --no-two-core --no-kernel-breakdown > $out/line.json 2> $out/kt.log
python3 tools/step_timeline.py $(find $out/kt -name '*kernel_trace.csv') 15 $out/launches.txt > $out/timeline.txt
rm -rf $out/kt
head -60 $out/timeline.txt
