export TMPDIR=/tmp
mkdir -p gpurun_out/r2z
timeout 1500 python3 -m pytest tests/test_gpu_align.py tests/test_gpu_pipe.py tests/test_gpu_scale.py -x -q 2>&1 | tail -4
python3 bench.py --no-cpu-baseline > gpurun_out/r2z/bench.json 2> gpurun_out/r2z/bench.err
python3 -c "
import json; d=json.load(open('gpurun_out/r2z/bench.json')); print(d['value'], d['ms_per_step'], d['host_wall_ms_per_step'], d['end_to_end']['ms'], d['upload_ms'], d['kernel_ms_per_step'].get('k1_index'))"
FZP_INDEX_PER_RUN=1 python3 bench.py --no-cpu-baseline --no-end-to-end > gpurun_out/r2z/bench_perrun.json 2> gpurun_out/r2z/bench_perrun.err
python3 -c "
import json; d=json.load(open('gpurun_out/r2z/bench_perrun.json')); print(d['value'], d['ms_per_step'], d['kernel_ms_per_step'].get('k1_index'))"
