#!/bin/bash
export TMPDIR=/tmp
out=gpurun_out/hostcpu4; mkdir -p $out; rm -f $out/*.json $out/*.log
B="--no-cpu-baseline --no-end-to-end --no-shaped-leg --steps 50 --warmup 3"
run() { tag=$1; shift; env "$@" FZP_BENCH_THREAD_CPU=1 python3 bench.py $B > $out/$tag.json 2> $out/$tag.log; }
run a_default
run b_nonice FZP_NO_NICE=1
run c_spin20 FZP_FETCH_SPIN_US=20
run d_spin200 FZP_FETCH_SPIN_US=200
run e_default_again
python3 - <<'PY'
import json,glob,os
for f in sorted(glob.glob("gpurun_out/hostcpu4/*.json")):
    try:
        d=json.loads([l for l in open(f) if l.startswith("{")][0])
        print(os.path.basename(f), "ms/step", d["ms_per_step"], "instr", d["ms_per_step_instrumented"], "cpu ms/step", d["host_cpu_ms_per_step"], "two_core", d.get("two_core"), d["host_wall_ms_per_step"], "sum kernels", round(sum(d["kernel_ms_per_step"].values()),2))
    except Exception as e:
        print(f, "failed", e)
PY
