export TMPDIR=/tmp
mkdir -p gpurun_out/r2t
timeout 1500 python3 -m pytest tests/test_gpu_scale.py -x -q --durations=5 2>&1 | tail -12
timeout 600 python3 bench.py > gpurun_out/r2t/bench.json 2> gpurun_out/r2t/bench.err
python3 -c "
import json; d=json.load(open('gpurun_out/r2t/bench.json')); print(d['value'], d['ms_per_step'], d['cpu_baseline'])"
