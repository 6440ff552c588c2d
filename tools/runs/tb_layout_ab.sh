# A/B of the mask stream layout (interleaved within launch groups of 64 vs every stream on its own): bench step and the shaped leg, twice each
for m in interleaved contig interleaved contig; do
  if [ $m = contig ]; then export FZP_TB_CONTIG=1; else unset FZP_TB_CONTIG; fi
  timeout 250 python bench.py --no-end-to-end --steps 4 --warmup 2 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys;d=json.loads(sys.stdin.read());k=d['kernel_ms_per_step'];print('$m', d['ms_per_step'], k['k1_sw'], k['k1_traceback'], d['k1_on_real_read_shape']['longest_first'])"
done
