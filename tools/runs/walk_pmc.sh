#!/bin/bash
# issue / wait counters of the trace-back walk (k_tb_walk) on the bench workload; usage on the GPU box: bash tools/runs/walk_pmc.sh
export TMPDIR=/tmp
for C in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_ACTIVE_INST_ANY SQ_INSTS_SALU" "SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_VMEM SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_SCA SQ_WAVES"; do
  out=gpurun_out/walk_pmc
  rm -rf $out; mkdir -p $out
  timeout 300 rocprofv3 --pmc $C -d $out -o p --output-format csv -- python3 tools/runs/swb_probe.py 0 > $out/log.txt 2>&1
  python3 - $out <<'PY'
import csv,sys,glob,collections
f=glob.glob(sys.argv[1]+'/**/*counter_collection.csv',recursive=True)
tot=collections.defaultdict(lambda: collections.Counter())
for r in csv.DictReader(open(f[0])):
    k=r["Kernel_Name"]
    k = "k_tb_walk<true>" if "k_tb_walk<true>" in k else None
    if k: tot[k][r["Counter_Name"]]+=float(r["Counter_Value"]); tot[k]["n_"+r["Counter_Name"]]+=1
for k,v in tot.items():
    n=max(v[c] for c in v if c.startswith("n_"))
    print(k,"dispatches",n,{c:round(x/n/1e6,2) for c,x in v.items() if not c.startswith("n_")})
PY
done
