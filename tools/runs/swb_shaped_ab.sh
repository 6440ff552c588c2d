export TMPDIR=/tmp
B="--no-cpu-baseline --no-end-to-end --no-two-core --no-from-files --steps 10 --warmup 3"
for v in "1:1048576:0:0" "2:16:0:4" "1:16:0:0" "2:32:0:4" "1:1048576:0:0" "2:16:0:4"; do
  IFS=: read w u h d <<< "$v"
  FZP_SWB_WAVES=$w FZP_SWB_UNIT=$u FZP_SWB_HYST=$h FZP_SWB_DBG=$d timeout 300 python3 bench.py $B 2>/dev/null | python3 -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][0])
print('$v', 'ms/step', d['ms_per_step'], 'k1_sw', d['kernel_ms_per_step']['k1_sw'], 'shaped', d['k1_on_real_read_shape']['k1_sw_ms'], d['k1_on_real_read_shape']['k1_traceback_ms'])"
done
