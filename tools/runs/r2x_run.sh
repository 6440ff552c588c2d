export TMPDIR=/tmp
mkdir -p gpurun_out/r2x
timeout 1500 python3 -m pytest tests/test_gpu_parity.py tests/test_gpu_cns.py tests/test_gpu_scale.py -x -q 2>&1 | tail -5
timeout 600 python3 bench.py --no-cpu-baseline --no-end-to-end > gpurun_out/r2x/bench.json 2> gpurun_out/r2x/bench.err
python3 -c "
import json; d=json.load(open('gpurun_out/r2x/bench.json')); print(d['value'], d['ms_per_step'], {k:v for k,v in d['kernel_ms_per_step'].items() if k.startswith('k2')}, d['host_wall_ms_per_step'])"
