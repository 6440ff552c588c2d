#!/bin/bash
# the round's last look (GPU box, repo root): every -m gpu test, the smoke run, the default bench line and the driver's command on the tree as it stands
tag=${1:-r6f}
export TMPDIR=/tmp
mkdir -p gpurun_out/$tag
timeout 2300 python3 -m pytest tests -m gpu -x -q > gpurun_out/$tag/full_gpu.log 2>&1; grep -E "passed|failed|error" gpurun_out/$tag/full_gpu.log | tail -2
python3 -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
python3 bench.py > gpurun_out/$tag/bench_line.json 2> gpurun_out/$tag/bench.log
python3 bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/$tag/bench_line_steps20.json 2> /dev/null
python3 - <<PY
import json
for f in ("bench_line.json","bench_line_steps20.json"):
    d=json.loads([l for l in open("gpurun_out/$tag/"+f) if l.startswith("{")][0])
    print(f, d["ms_per_step"], d["value"], d["host_cpu_ms_per_step"], {k:d["two_core"].get(k) for k in ("ms_per_step","vs_unconstrained","host_cpu_ms_per_step")}, d["roofline"]["frac"], d["value_from_files"], d["value_end_to_end"], d["two_steps_in_flight"]["ms_per_step"], d["cpu_baseline"]["value"], d["cpu_baseline"]["k1_cigars_differing_from_hip"])
PY
