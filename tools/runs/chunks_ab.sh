#!/bin/bash
# K1 in several chunks of reads (FZP_SW_CHUNKS): the walk of chunk k on stream2 under the DP of chunk k + 1 -- does the step get shorter?
export TMPDIR=/tmp
out=gpurun_out/chunks; mkdir -p $out; rm -f $out/*
B="--no-cpu-baseline --no-end-to-end --no-shaped-leg --no-two-core --steps 20 --warmup 3"
for c in 1 2 3 4 6 1; do
  FZP_SW_CHUNKS=$c python3 bench.py $B > $out/c$c.json 2> $out/c$c.log
  python3 - <<PY
import json
d=json.loads([l for l in open("$out/c$c.json") if l.startswith("{")][0])
k=d["kernel_ms_per_step"]
print("chunks $c", "ms/step", d["ms_per_step"], "instrumented", d.get("ms_per_step_instrumented"), "k1 wall", d["host_wall_ms_per_step"]["k1"], "k1_sw", k.get("k1_sw"), "k1_traceback", k.get("k1_traceback"), "value", d["value"])
PY
done
