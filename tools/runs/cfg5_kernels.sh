#!/bin/bash
# kernel stats of configs[4] at half scale with consensus (rocprofv3 --kernel-trace --stats): which kernels a genome-scale run spends its time in
export TMPDIR=/tmp
out=gpurun_out/cfg5k; mkdir -p $out; rm -rf $out/*
cat > $out/run.py <<PY
import os, sys
sys.path.insert(0, "tools"); sys.path.insert(0, ".")
import run_cfg5
r = run_cfg5.run(scale=0.5, lanes=2, workers=1, consensus=True)
print(r["wall_s"], r["first_call_wall_s"], {k: r["stats"][k] for k in ("ms_upload", "ms_k1", "ms_phase", "ms_results", "ms_text", "n_groups")})
PY
rocprofv3 --kernel-trace --stats -d $out/kt -o kt --output-format csv -- python3 $out/run.py > $out/line.txt 2> $out/log.txt
cat $out/line.txt
python3 - <<PY
import csv, glob
for f in glob.glob("$out/kt/**/*kernel_stats.csv", recursive=True):
    rows = list(csv.DictReader(open(f)))
    tot = sum(float(r["TotalDurationNs"]) for r in rows)
    print("all kernels %.1f ms (two passes of the half-scale genome)" % (tot / 1e6))
    for r in rows[:28]:
        print("  %8.2f ms %6s calls  %s" % (float(r["TotalDurationNs"]) / 1e6, r["Calls"], r["Name"].replace("(anonymous namespace)::", "").replace("void ", "")[:70]))
PY
rm -rf $out/kt
