#!/bin/bash
# runtime knobs against the CPU a rank burns per step (the async-events thread of direct dispatch); tmpfs outputs
export TMPDIR=/tmp
out=gpurun_out/hostcpu2; mkdir -p $out; rm -f $out/*.json $out/*.log
B="--no-cpu-baseline --no-end-to-end --no-shaped-leg --steps 50 --warmup 3"
run() { tag=$1; shift; env "$@" FZP_BENCH_THREAD_CPU=1 python3 bench.py $B > $out/$tag.json 2> $out/$tag.log; }
run a_blk60 FZP_SCHED=blocking
run b_blk60_dd0 FZP_SCHED=blocking AMD_DIRECT_DISPATCH=0
run c_auto_dd0 FZP_SCHED=auto AMD_DIRECT_DISPATCH=0 FZP_FETCH_SPIN_US=100000000
run d_blk60_pool4k FZP_SCHED=blocking ROC_SIGNAL_POOL_SIZE=4096
run e_blk60_pool64k FZP_SCHED=blocking ROC_SIGNAL_POOL_SIZE=65536
run f_blk60_dd0_w4 FZP_SCHED=blocking AMD_DIRECT_DISPATCH=0 FZP_WRITER_THREADS=4
run g_blk60_dd0_noprof FZP_SCHED=blocking AMD_DIRECT_DISPATCH=0 FZP_BENCH_NO_PROF=1
run h_yield_dd0 FZP_SCHED=yield AMD_DIRECT_DISPATCH=0
python3 - <<'PY'
import json,glob,os
for f in sorted(glob.glob("gpurun_out/hostcpu2/*.json")):
    try:
        d=json.loads([l for l in open(f) if l.startswith("{")][0])
        print(os.path.basename(f), "ms/step", d["ms_per_step"], "cpu ms/step", d["host_cpu_ms_per_step"], d["out_fs"], "two_core", d.get("two_core"), d["host_wall_ms_per_step"])
    except Exception as e:
        print(f, "failed", e)
PY
grep -H "thread cpu" $out/*.log | cut -c1-600
