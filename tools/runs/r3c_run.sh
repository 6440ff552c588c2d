export TMPDIR=/tmp
mkdir -p gpurun_out/r3c
timeout 900 python3 -m pytest tests/test_gpu_pipe.py tests/test_gpu_pipeline.py -x -q 2>&1 | tail -3
python3 bench.py --no-cpu-baseline > gpurun_out/r3c/bench.json 2> gpurun_out/r3c/bench.err
python3 -c "
import json; d=json.load(open('gpurun_out/r3c/bench.json')); print(d['value'], d['ms_per_step'], d['host_wall_ms_per_step'], d['end_to_end']['ms'], d['end_to_end']['lanes'], d['two_steps_in_flight'])"
