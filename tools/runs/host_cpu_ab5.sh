#!/bin/bash
export TMPDIR=/tmp
out=gpurun_out/hostcpu5; mkdir -p $out; rm -f $out/*.json $out/*.log
B="--no-cpu-baseline --no-end-to-end --no-shaped-leg --steps 20 --warmup 3"
run() { tag=$1; shift; env "$@" FZP_BENCH_2C_STEPS=40 python3 bench.py $B > $out/$tag.json 2> $out/$tag.log; }
for i in 1 2 3; do
run nice_$i
run nonice_$i FZP_NO_NICE=1
run old_$i FZP_SCHED=auto FZP_FETCH_SPIN_US=100000000 FZP_WRITER_THREADS=16 ROC_SIGNAL_POOL_SIZE=64 FZP_NO_NICE=1
done
python3 - <<'PY'
import json,glob,os
for f in sorted(glob.glob("gpurun_out/hostcpu5/*.json")):
    try:
        d=json.loads([l for l in open(f) if l.startswith("{")][0])
        print(os.path.basename(f), "ms/step", d["ms_per_step"], "cpu ms/step", d["host_cpu_ms_per_step"], "two_core", d["two_core"]["ms_per_step"], d["two_core"]["host_cpu_ms_per_step"], d["two_core"]["vs_unconstrained"])
    except Exception as e:
        print(f, "failed", e)
PY
