# the shaped leg several times over (its two DP kernels run side by side: how stable is the pair?)
for i in 1 2 3 4; do
  timeout 250 python bench.py --no-end-to-end --steps 2 --warmup 1 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys;d=json.loads(sys.stdin.read());k=d['kernel_ms_per_step'];print(d['ms_per_step'], k['k1_sw'], d['k1_on_real_read_shape']['longest_first']['k1_sw_ms'], d['k1_on_real_read_shape']['longest_first_no_priority']['k1_sw_ms'])"
done
