# A/B in one session: k_swb's base streams through LDS rings (default) or straight from HBM (FZP_SWB_NO_RING=1)
for m in ring hbm ring hbm ring hbm; do
  if [ $m = hbm ]; then export FZP_SWB_NO_RING=1; else unset FZP_SWB_NO_RING; fi
  timeout 250 python bench.py --no-end-to-end --steps 5 --warmup 2 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys;d=json.loads(sys.stdin.read());k=d['kernel_ms_per_step'];print('$m', d['ms_per_step'], 'sw', k['k1_sw'], 'tb', k['k1_traceback'], 'back', k['k1_back'], 'shaped', d['k1_on_real_read_shape']['longest_first']['k1_sw_ms'])"
done
