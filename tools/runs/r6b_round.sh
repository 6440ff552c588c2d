#!/bin/bash
# r6, second half: the in-place file rewrite (two-core step), k_swb's two-wave build at band 32
export TMPDIR=/tmp
out=gpurun_out/r6b; mkdir -p $out; rm -f $out/*
timeout 900 python3 -m pytest tests/test_gpu_pipe.py -x -q -m gpu > $out/pipe.test 2>&1; tail -2 $out/pipe.test
B="--no-cpu-baseline --no-end-to-end --no-shaped-leg --no-from-files"
python3 bench.py $B > $out/inplace.json 2> $out/inplace.log
python3 bench.py $B --fresh-trees > $out/fresh.json 2> $out/fresh.log
python3 - <<PY
import json
for t in ("inplace","fresh"):
    d=json.loads([l for l in open("$out/%s.json"%t) if l.startswith("{")][0])
    print(t, "ms/step", d["ms_per_step"], "cpu", d["host_cpu_ms_per_step"], "two_core", {k:d["two_core"].get(k) for k in ("ms_per_step","host_cpu_ms_per_step","vs_unconstrained","cpu_ms_per_step_by_thread","fresh_trees")}, "fresh_cmp", d.get("fresh_trees"))
PY
B2="$B --no-two-core --steps 20"
for v in 1:1048576:0:0 2:1048576:0:0 2:16:0:4 2:16:0:0 2:32:0:4; do
  IFS=: read w u h d <<< "$v"
  FZP_SWB_WAVES=$w FZP_SWB_UNIT=$u FZP_SWB_HYST=$h FZP_SWB_DBG=$d python3 bench.py $B2 > $out/w${w}_u${u}_d$d.json 2> $out/w${w}_u${u}_d$d.log
  python3 - <<PY
import json
d=json.loads([l for l in open("$out/w${w}_u${u}_d$d.json") if l.startswith("{")][0])
k=d["kernel_ms_per_step"]
print("waves $w unit $u dbg $d", "ms/step", d["ms_per_step"], "k1_sw", k.get("k1_sw"), "k1_traceback", k.get("k1_traceback"), "value", d["value"])
PY
done
