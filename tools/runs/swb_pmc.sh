#!/bin/bash
# issue / wait counters of the two bit-sliced DP kernels (k_swb2 default, k_swb with FZP_SWB_64=1); usage on the GPU box: bash tools/runs/swb_pmc.sh
export TMPDIR=/tmp
C="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_ACTIVE_INST_ANY SQ_INST_CYCLES_VMEM_WR"
for v in pair lane64; do
  if [ $v = lane64 ]; then export FZP_SWB_64=1; else unset FZP_SWB_64; fi
  out=gpurun_out/swb_pmc_$v
  rm -rf $out; mkdir -p $out
  timeout 300 rocprofv3 --pmc $C -d $out -o p --output-format csv -- python3 tools/runs/swb_probe.py 0 > $out/log.txt 2>&1
  python3 - $out <<'PY'
import csv,sys,glob,collections
f=glob.glob(sys.argv[1]+'/**/*counter_collection.csv',recursive=True)
tot=collections.defaultdict(lambda: collections.Counter())
for r in csv.DictReader(open(f[0])):
    k=r["Kernel_Name"]
    k = "k_swb2" if "k_swb2" in k else ("k_swb" if "k_swb" in k else None)
    if k: tot[k][r["Counter_Name"]]+=float(r["Counter_Value"]); tot[k]["n_"+r["Counter_Name"]]+=1
for k,v in tot.items():
    n=v["n_SQ_WAVE_CYCLES"]
    print(sys.argv[1],k,"dispatches",n,{c:round(x/n/1e6,2) for c,x in v.items() if not c.startswith("n_")})
PY
done
