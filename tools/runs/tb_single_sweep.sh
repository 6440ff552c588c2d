for m in 40960 16384 8192 4096; do
  FZP_TB_SINGLE_STEPS=$m timeout 200 python bench.py --no-end-to-end --steps 3 --warmup 1 --no-cpu-baseline --no-shaped-leg 2>/dev/null | python -c "
import json,sys;d=json.loads(sys.stdin.read());print($m, d['ms_per_step'], d['kernel_ms_per_step']['k1_traceback'], d['traceback'])"
done
