export TMPDIR=/tmp
echo "mem.max: $(cat /sys/fs/cgroup/memory.max 2>/dev/null)"; free -g | head -2; df -h /tmp /dev/shm . | cat; nproc
B="--no-cpu-baseline --no-end-to-end --no-shaped-leg --gen-workers 1 --steps 5 --warmup 2"
for dbg in 0 1 2 3; do
  FZP_SWB_DBG=$dbg timeout 100 python3 bench.py $B 2>/dev/null | python3 -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); k=d['kernel_ms_per_step']; print('dbg $dbg', 'ms', d['ms_per_step'], 'k1_sw', k['k1_sw'], 'tb', k['k1_traceback'], 'seed', k['k1_seed'])"
done
