export TMPDIR=/tmp
timeout 1200 python3 -m pytest tests/test_gpu_align.py -x -q 2>&1 | tail -3
python3 bench.py --no-cpu-baseline --no-end-to-end > gpurun_out/tb.json 2> gpurun_out/tb.err
python3 -c "
import json; d=json.load(open('gpurun_out/tb.json')); print(d['value'], d['ms_per_step'], {k:v for k,v in d['kernel_ms_per_step'].items() if k.startswith('k1')})"
