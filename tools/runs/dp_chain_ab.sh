# A/B: forward DPs of different jobs chained one behind the other (default) or free to run side by side (FZP_DP_NO_CHAIN=1): two steps in flight, end to end on two lanes
for m in chain free chain free; do
  if [ $m = free ]; then export FZP_DP_NO_CHAIN=1; else unset FZP_DP_NO_CHAIN; fi
  timeout 280 python bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-shaped-leg 2>/dev/null | python -c "
import json,sys;d=json.loads(sys.stdin.read());print('$m', d['ms_per_step'], 'two in flight', d['two_steps_in_flight']['ms_per_step'], d['two_steps_in_flight']['reads_per_s'], 'e2e', d['value_end_to_end'], d['end_to_end'].get('lanes'), d['end_to_end'].get('other_shape'))"
done
