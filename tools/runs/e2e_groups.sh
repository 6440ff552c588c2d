#!/bin/bash
# end to end (host buffers) and from files with groups of 10 / 7 / 5 / 4 contigs on 2 and 3 lanes: which shape of the lanes pipeline is fastest for the bench step's 20 contigs
export TMPDIR=/tmp
out=gpurun_out/e2eg; mkdir -p $out; rm -f $out/*
for shape in "10 2" "7 2" "5 2" "4 2" "5 3" "7 3" "4 3"; do
  set -- $shape
  python3 bench.py --no-cpu-baseline --no-shaped-leg --no-two-core --no-kernel-breakdown --steps 3 --warmup 1 --e2e-group-contigs $1 --e2e-lanes $2 > $out/g$1_l$2.json 2> $out/g$1_l$2.log
  python3 - <<PY
import json
d=json.loads([l for l in open("$out/g$1_l$2.json") if l.startswith("{")][0])
print("group $1 lanes $2: end_to_end", d["end_to_end"]["ms"], d["end_to_end"]["lanes"], "from_files", d["from_files"]["ms"], d["from_files"].get("groups"))
PY
done
