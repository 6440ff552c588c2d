cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
out=gpurun_out/prof_k6
rm -rf $out; mkdir -p $out
B="--no-cpu-baseline --no-end-to-end --gen-workers 1 --with-consensus --steps 2 --warmup 1"
rocprofv3 --kernel-trace --stats -d $out/kt -o kt --output-format csv -- python3 bench.py $B > /dev/null 2> $out/kt.log
python3 - <<'PY'
import csv, glob
f = glob.glob("gpurun_out/prof_k6/kt/**/*kernel_stats.csv", recursive=True)[0]
for r in list(csv.DictReader(open(f)))[:14]:
    n = r['Name'].replace('(anonymous namespace)::','').split('(')[0]
    print("%-28s calls %4s avg %9.3f ms min %9.3f max %9.3f" % (n[:28], r['Calls'], float(r['AverageNs'])/1e6, float(r['MinNs'])/1e6, float(r['MaxNs'])/1e6))
PY
