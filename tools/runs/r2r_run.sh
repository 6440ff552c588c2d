export TMPDIR=/tmp
mkdir -p gpurun_out/r2r
for v in 1 0; do
FZP_SW_SPLIT_ROUNDS=$v timeout 600 python3 bench.py --no-cpu-baseline --no-end-to-end > gpurun_out/r2r/bench$v.json 2> gpurun_out/r2r/bench.err
python3 -c "
import json; d=json.load(open('gpurun_out/r2r/bench$v.json')); k=d['kernel_ms_per_step']; print($v, d['value'], d['ms_per_step'], d['host_wall_ms_per_step']['k1'], k['k1_sw'], k['k1_traceback'])"
done
timeout 900 python3 -m pytest tests/test_gpu_align.py tests/test_gpu_scale.py -x -q 2>&1 | tail -2
