export TMPDIR=/tmp
mkdir -p gpurun_out/r2f
FZP_PIPE_TIMING=1 timeout 900 python3 bench.py --no-cpu-baseline --steps 3 > gpurun_out/r2f/bench.json 2> gpurun_out/r2f/bench.err
grep fzp_pipe gpurun_out/r2f/bench.err | tail -3
python3 -c "
import json,sys; d=json.load(open('gpurun_out/r2f/bench.json')); print(d['value'], d['ms_per_step'], d['host_wall_ms_per_step'], d['end_to_end'])"
df -h /tmp | tail -2; mount | grep -E " /tmp| / " | head
