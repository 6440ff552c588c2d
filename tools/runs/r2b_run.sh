export TMPDIR=/tmp
mkdir -p gpurun_out/r2b
timeout 900 python3 -m pytest tests/test_gpu_align.py -x -q > gpurun_out/r2b/pytest_align.txt 2>&1
tail -15 gpurun_out/r2b/pytest_align.txt
timeout 600 python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline > gpurun_out/r2b/bench.json 2> gpurun_out/r2b/bench.err
python3 -c "
import json; d=json.load(open('gpurun_out/r2b/bench.json')); print(d['value'], d['ms_per_step'], d['kernel_ms_per_step'], d['aligned_frac'])"
./tools/ubench/valu_issue > gpurun_out/r2b/valu_issue.txt 2>&1
