#!/bin/bash
# k_tb_walk variants (tools/runs/build/libfzphase_rpw<N>.so, built with -DFZP_TBW_RPW=<N>: walkers per wave): parity against the twin, then the bench step
export TMPDIR=/tmp
out=gpurun_out/tbwv; mkdir -p $out; rm -f $out/*
B="--no-cpu-baseline --no-end-to-end --no-shaped-leg --no-two-core --steps 20 --warmup 3"
for t in default "$@"; do
  if [ $t = default ]; then unset FZP_LIB; else export FZP_LIB=$PWD/tools/runs/build/libfzphase_$t.so; fi
  timeout 600 python3 -m pytest tests/test_gpu_align.py -x -q -m gpu -k "matches_cpu_twin or randomized or long_reads" > $out/$t.test 2>&1; tail -1 $out/$t.test
  python3 bench.py $B > $out/$t.json 2> $out/$t.log
  python3 - <<PY
import json
d=json.loads([l for l in open("$out/$t.json") if l.startswith("{")][0])
k=d["kernel_ms_per_step"]
print("$t", "ms/step", d["ms_per_step"], "k1_sw", k.get("k1_sw"), "k1_traceback", k.get("k1_traceback"), "roofline avg", d["roofline"]["avg_launch_ms"], "value", d["value"])
PY
done
