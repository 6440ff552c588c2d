cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
out=gpurun_out/prof_chain
rm -rf $out; mkdir -p $out
B="--no-cpu-baseline --no-end-to-end --gen-workers 1 --steps 1 --warmup 0"
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES -d $out/a -o p --output-format csv -- python3 bench.py $B > /dev/null 2> $out/a.log
python3 - <<'PY'
import csv, glob, collections
agg = collections.defaultdict(lambda: collections.defaultdict(float))
for f in glob.glob("gpurun_out/prof_chain/a/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        n = r['Kernel_Name'].replace('(anonymous namespace)::','').split('(')[0]
        if n in ("k_chain","k_seed","k_tb_walk","k_cig_ckpt","k_gather16","k_vmap_sites"): agg[n][r['Counter_Name']] += float(r['Counter_Value'])
for n, v in agg.items(): print(n, {k: round(x/1e6,2) for k,x in v.items()})
PY
