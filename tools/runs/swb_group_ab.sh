# A/B of k_swb's mask-store group (SWB_GROUP in fzp_align.hip): build the variants as tools/runs/libs/g<N>.so first (FZP_LIB picks the library); measured with the LDS rings: 4 -> 9.1-9.2 ms, 8 -> 8.7-8.8, 16 -> 8.8-8.9
for m in 8 4 16 8 4 16; do
  if [ $m = 8 ]; then unset FZP_LIB; else export FZP_LIB=$PWD/tools/runs/libs/g$m.so; fi
  timeout 250 python bench.py --no-end-to-end --steps 5 --warmup 2 --no-cpu-baseline --no-shaped-leg 2>/dev/null | python -c "
import json,sys;d=json.loads(sys.stdin.read());k=d['kernel_ms_per_step'];print('group $m', d['ms_per_step'], 'sw', k['k1_sw'])"
done
