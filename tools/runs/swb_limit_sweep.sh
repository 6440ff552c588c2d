for m in 24576 32768 40960 49152 57344; do
  FZP_SWB_MAX_STEPS=$m timeout 200 python bench.py --no-end-to-end --steps 2 --warmup 1 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys;d=json.loads(sys.stdin.read());print($m, d['k1_on_real_read_shape']['longest_first'])"
done
