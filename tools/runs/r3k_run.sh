export TMPDIR=/tmp
timeout 1200 python3 -m pytest tests/test_gpu_cns.py tests/test_gpu_parity.py tests/test_gpu_scale.py -x -q 2>&1 | tail -3
python3 bench.py --no-cpu-baseline --no-end-to-end --with-consensus --steps 5 --warmup 2 > gpurun_out/cns.json 2> gpurun_out/cns.err
python3 -c "
import json; d=json.load(open('gpurun_out/cns.json')); print('cns', d['value'], d['ms_per_step'], {k:v for k,v in d['kernel_ms_per_step'].items() if k.startswith('k6') or k=='k2_pileup_count'}, d['host_wall_ms_per_step'])"
python3 bench.py --no-cpu-baseline --no-end-to-end > gpurun_out/nocns.json 2> gpurun_out/nocns.err
python3 -c "
import json; d=json.load(open('gpurun_out/nocns.json')); print('std', d['value'], d['ms_per_step'], {k:v for k,v in d['kernel_ms_per_step'].items() if k.startswith('k2')}, d['host_wall_ms_per_step'])"
