#!/bin/bash
# usage (GPU box, repo root): bash tools/runs/host_cpu_ab.sh -- how the launch thread waits: device scheduling flag x fzp_fetch spin budget x writer threads; the bench line
# carries host_cpu_ms_per_step and two_core_step_ms (a child on two CPUs with LOCAL_WORLD_SIZE=8), stderr the per-thread CPU split (user + system, of which system)
export TMPDIR=/tmp
out=gpurun_out/hostcpu; mkdir -p $out; rm -f $out/*.json $out/*.log
B="--no-cpu-baseline --no-end-to-end --no-shaped-leg --steps 50 --warmup 3"
run() { tag=$1; shift; env "$@" FZP_BENCH_THREAD_CPU=1 python3 bench.py $B > $out/$tag.json 2> $out/$tag.log; }
run a_old FZP_SCHED=auto FZP_FETCH_SPIN_US=100000000
run b_blk60 FZP_SCHED=blocking
run c_blk300 FZP_SCHED=blocking FZP_FETCH_SPIN_US=300
run d_blk60_noprof FZP_SCHED=blocking FZP_BENCH_NO_PROF=1
run e_blk60_w4 FZP_SCHED=blocking FZP_WRITER_THREADS=4
run f_blk60_w2 FZP_SCHED=blocking FZP_WRITER_THREADS=2
run g_old_w4 FZP_SCHED=auto FZP_FETCH_SPIN_US=100000000 FZP_WRITER_THREADS=4
run h_yield60 FZP_SCHED=yield
python3 - <<'PY'
import json,glob,os
for f in sorted(glob.glob("gpurun_out/hostcpu/*.json")):
    try:
        d=json.loads([l for l in open(f) if l.startswith("{")][0])
        print(os.path.basename(f), "ms/step", d["ms_per_step"], "cpu ms/step", d["host_cpu_ms_per_step"], d["out_fs"], "two_core", d.get("two_core"), d["host_wall_ms_per_step"])
    except Exception as e:
        print(f, "failed", e)
PY
grep -H "thread cpu" $out/*.log | cut -c1-900
