#!/bin/bash
# bench step with alternative builds of the library (tools/runs/build/libfzphase_<tag>.so) on one box: bash tools/runs/lib_ab.sh <tag> ...
export TMPDIR=/tmp
out=gpurun_out/libab; mkdir -p $out; rm -f $out/*
B="--no-cpu-baseline --no-end-to-end --no-shaped-leg --no-two-core --no-from-files --steps 20 --warmup 3"
for t in default "$@" default; do
  if [ $t = default ]; then unset FZP_LIB; else export FZP_LIB=$PWD/tools/runs/build/libfzphase_$t.so; fi
  python3 bench.py $B > $out/$t.json 2> $out/$t.log
  python3 - <<PY
import json
d=json.loads([l for l in open("$out/$t.json") if l.startswith("{")][0])
k=d["kernel_ms_per_step"]
print("$t", "ms/step", d["ms_per_step"], {x: k[x] for x in ("k1_index","k1_seed","k1_sw","k1_traceback","k1_cigar")}, "value", d["value"])
PY
done
