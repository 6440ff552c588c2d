#!/bin/bash
# the walk of the bit-sliced kernel's slots: 32 walkers per wave / 32-step pieces (k_tb_walk_h) against the 16-walker form (FZP_TBW_OLD=1): parity, then the bench step's kernel times
export TMPDIR=/tmp
out=gpurun_out/tbh; mkdir -p $out; rm -f $out/*
B="--no-cpu-baseline --no-end-to-end --no-shaped-leg --no-two-core --steps 20 --warmup 3"
timeout 900 python3 -m pytest tests/test_gpu_align.py -x -q -m gpu > $out/new.test 2>&1; tail -1 $out/new.test
for t in new old new2 old2; do
  if [ ${t:0:3} = old ]; then export FZP_TBW_OLD=1; else unset FZP_TBW_OLD; fi
  python3 bench.py $B > $out/$t.json 2> $out/$t.log
  python3 - <<PY
import json
d=json.loads([l for l in open("$out/$t.json") if l.startswith("{")][0])
k=d["kernel_ms_per_step"]
print("$t", "ms/step", d["ms_per_step"], "k1_sw", k.get("k1_sw"), "k1_traceback", k.get("k1_traceback"), "k1_join", k.get("k1_join"), "value", d["value"], "real shape tb", d.get("k1_on_real_read_shape"))
PY
done
