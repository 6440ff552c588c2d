export TMPDIR=/tmp
df -h /tmp | tail -1
python3 bench.py --no-cpu-baseline --no-end-to-end --steps 100 --warmup 3 > gpurun_out/soak.json 2> gpurun_out/soak.err; tail -2 gpurun_out/soak.err
python3 -c "
import json; d=json.load(open('gpurun_out/soak.json')); print(d['value'], d['ms_per_step'], d['host_wall_ms_per_step'])"
df -h /tmp | tail -1
