#!/bin/bash
# the round's closing set (GPU box, repo root): bash tools/runs/r6_final.sh <tag>
# profile_round (kernel stats, HBM passes, SQ counters, default bench line), a step's timeline, the bench with polish / consensus, configs[4] on one GPU from memory and from files
tag=$1
export TMPDIR=/tmp
bash tools/profile_round.sh $tag > gpurun_out/prof_$tag.out 2>&1
bash tools/runs/trace_step.sh $tag > gpurun_out/trace_$tag.out 2>&1
out=gpurun_out/prof_$tag
python3 bench.py --with-polish --no-cpu-baseline --no-end-to-end --no-from-files --no-shaped-leg --no-two-core > $out/bench_line_with_polish.json 2> $out/polish.log
python3 bench.py --with-consensus --no-cpu-baseline --no-end-to-end --no-from-files --no-shaped-leg --no-two-core > $out/bench_line_with_consensus.json 2> $out/consensus.log
python3 bench.py --gpus 1 --steps 20 --warmup 5 > $out/bench_line_steps20.json 2> $out/steps20.log
timeout 900 python3 tools/run_cfg5.py > $out/cfg5_one_gpu.json 2> $out/cfg5.log
timeout 900 python3 tools/run_cfg5.py --from-files > $out/cfg5_one_gpu_from_files.json 2> $out/cfg5f.log
ls -la $out | head -40
python3 - <<PY
import json
for f in ("bench_line.json","bench_line_steps20.json","bench_line_with_polish.json","bench_line_with_consensus.json"):
    try:
        d=json.loads([l for l in open("$out/"+f) if l.startswith("{")][0])
        print(f, d["ms_per_step"], d["value"], d.get("two_core",{}) and {k:d["two_core"].get(k) for k in ("ms_per_step","vs_unconstrained","host_cpu_ms_per_step")}, d.get("roofline",{}).get("frac"), d.get("polish_tigs"))
    except Exception as e: print(f, "ERR", e)
for f in ("cfg5_one_gpu.json","cfg5_one_gpu_from_files.json"):
    try:
        d=json.loads([l for l in open("$out/"+f) if l.startswith("{")][-1]); print(f, {k:d.get(k) for k in ("wall_s","reads_per_s","peak_hbm_gb","reads")})
    except Exception as e: print(f, "ERR", e)
PY
