export TMPDIR=/tmp
B="--no-cpu-baseline --no-end-to-end --no-shaped-leg --gen-workers 1 --steps 8 --warmup 2"
for v in 0 1 0 1; do
if [ $v = 1 ]; then export FZP_SW_SERIAL=1; else unset FZP_SW_SERIAL; fi
timeout 200 python3 bench.py $B 2>/dev/null | python3 -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); k=d['kernel_ms_per_step']; print('serial $v ms', d['ms_per_step'], 'k1_sw', k['k1_sw'], 'tb', k['k1_traceback'])"
done
