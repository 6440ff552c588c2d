#!/bin/bash
# k_swb's work units and register budgets (r6): parity of the unit form against the twin, then the bench step per (FZP_SWB_WAVES, FZP_SWB_UNIT) pair.
# usage: bash tools/runs/swb_units_ab.sh "<waves>:<unit>[:<hyst>[:<dbg>]] ..."   (dbg 4 = the two waves of a SIMD take turns at the higher priority)      (unit 1048576 = whole groups, the r5 schedule)
export TMPDIR=/tmp
out=gpurun_out/swbu; mkdir -p $out; rm -f $out/*
B="--no-cpu-baseline --no-end-to-end --no-shaped-leg --no-two-core --no-from-files --steps 20 --warmup 3"
FZP_SWB_UNIT=3 timeout 900 python3 -m pytest tests/test_gpu_align.py -x -q -m gpu -k "matches_cpu_twin or randomized or long_reads" > $out/parity_u3.test 2>&1; tail -1 $out/parity_u3.test
FZP_SWB_UNIT=3 FZP_SWB_WAVES=1 timeout 900 python3 -m pytest tests/test_gpu_align.py -x -q -m gpu -k "matches_cpu_twin or randomized or long_reads" > $out/parity_u3w1.test 2>&1; tail -1 $out/parity_u3w1.test
for v in $1; do
  IFS=: read w u h d <<< "$v"; h=${h:-0}; d=${d:-0}
  FZP_SWB_WAVES=$w FZP_SWB_UNIT=$u FZP_SWB_HYST=$h FZP_SWB_DBG=$d python3 bench.py $B > $out/w${w}_u${u}_h${h}_d$d.json 2> $out/w${w}_u${u}_h${h}_d$d.log
  python3 - <<PY
import json
d=json.loads([l for l in open("$out/w${w}_u${u}_h${h}_d$d.json") if l.startswith("{")][0])
k=d["kernel_ms_per_step"]
print("waves $w unit $u hyst $h dbg $d", "ms/step", d["ms_per_step"], "k1_sw", k.get("k1_sw"), "k1_traceback", k.get("k1_traceback"), "roofline avg", d["roofline"]["avg_launch_ms"], "value", d["value"], "cigars differing", d.get("k1_cigars_differing_from_hip"))
PY
done
