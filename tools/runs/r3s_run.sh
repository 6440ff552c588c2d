export TMPDIR=/tmp
timeout 1500 python3 -m pytest tests/test_gpu_parity.py tests/test_gpu_scale.py tests/test_gpu_pipe.py -x -q 2>&1 | tail -3
python3 bench.py --no-cpu-baseline --no-end-to-end > gpurun_out/as.json 2> gpurun_out/as.err
python3 -c "
import json; d=json.load(open('gpurun_out/as.json')); print(d['value'], d['ms_per_step'], {k:v for k,v in d['kernel_ms_per_step'].items() if k.startswith('k3')}, d['host_wall_ms_per_step'])"
