#!/bin/bash
# usage (GPU box, repo root): bash tools/runs/pmc_one.sh <tag> "<COUNTER ...>" [kernel substring]  -- one counter pass over one bench step, per-kernel sums
tag=$1; ctrs=$2; pat=${3:-k_}
export TMPDIR=/tmp
out=gpurun_out/pmc_$tag; mkdir -p $out
C="--no-cpu-baseline --no-end-to-end --no-shaped-leg --gen-workers 1 --no-tree-compare --no-kernel-breakdown --no-two-core"
rocprofv3 --pmc $ctrs -d $out/p -o p --output-format csv -- python3 bench.py --steps 1 --warmup 0 $C > /dev/null 2> $out/log.txt
python3 - <<PY
import csv, glob, collections, re
tot = collections.defaultdict(collections.Counter)
for f in glob.glob("$out/p/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        n = re.sub(r"\(.*", "", r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", ""))
        tot[n][r["Counter_Name"]] += float(r["Counter_Value"])
for n, c in sorted(tot.items(), key=lambda kv: -sum(kv[1].values()))[:40]:
    if "$pat" in n: print(n, dict(c))
PY
rm -rf $out/p
