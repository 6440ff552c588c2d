export TMPDIR=/tmp
python3 bench.py --no-cpu-baseline --no-end-to-end --steps 3 --warmup 1 > gpurun_out/tb.json 2> gpurun_out/tb.err
python3 -c "
import json; d=json.load(open('gpurun_out/tb.json')); print(d['value'], d['ms_per_step'], {k:v for k,v in d['kernel_ms_per_step'].items() if k in ('k1_sw','k1_traceback')})"
