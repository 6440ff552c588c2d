export TMPDIR=/tmp
for x in 0 1 2; do
FZP_TBX=$x python3 bench.py --no-cpu-baseline --no-end-to-end --steps 3 --warmup 1 > gpurun_out/tbx.json 2> gpurun_out/tbx.err
python3 -c "
import json; d=json.load(open('gpurun_out/tbx.json')); print('tbx=$x', {k:v for k,v in d['kernel_ms_per_step'].items() if k in ('k1_traceback','k1_sw')})" || tail -3 gpurun_out/tbx.err
done
