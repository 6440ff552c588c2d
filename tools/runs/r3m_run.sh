export TMPDIR=/tmp
for e in 12345 12346; do
FZP_CNS_EXP=$e python3 bench.py --no-cpu-baseline --no-end-to-end --with-consensus --steps 3 --warmup 1 > gpurun_out/cnsx.json 2> gpurun_out/cnsx.err
python3 -c "
import json; d=json.load(open('gpurun_out/cnsx.json')); print('$e', {k:v for k,v in d['kernel_ms_per_step'].items() if k.startswith('k6_tally')})"
done
