#!/bin/bash
# usage (GPU box, repo root): bash tools/runs/sq_cycles_only.sh <tag> [ENV=VALUE ...]  -- the two SQ cycle-counter passes of tools/profile_round.sh alone (one bench step each)
tag=$1; shift
for kv in "$@"; do export "$kv"; done
export TMPDIR=/tmp
out=gpurun_out/sqc_$tag
mkdir -p $out
C="--no-cpu-baseline --no-end-to-end --no-shaped-leg --gen-workers 1 --no-tree-compare --no-kernel-breakdown --no-two-core"
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY -d $out/sq2 -o p --output-format csv -- python3 bench.py --steps 1 --warmup 0 $C > /dev/null 2> $out/sq2.log
rocprofv3 --pmc SQ_WAIT_ANY SQ_INSTS_LDS SQ_INSTS_VALU SQ_WAVES -d $out/sq3 -o p --output-format csv -- python3 bench.py --steps 1 --warmup 0 $C > /dev/null 2> $out/sq3.log
cat $(find $out/sq2 -name '*counter_collection.csv') > $out/a.csv; cat $(find $out/sq3 -name '*counter_collection.csv') > $out/b.csv
(cd tools && python3 sq_cycles_summary.py ../$out/a.csv ../$out/b.csv ../$out/sq_cycles.json)
rm -rf $out/sq2 $out/sq3 $out/a.csv $out/b.csv
python3 - <<PY
import json
d=json.load(open("$out/sq_cycles.json"))
for k,v in d.items():
    if "tb_walk" in k or "swb" in k: print(k, json.dumps(v))
PY
