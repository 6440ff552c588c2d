export TMPDIR=/tmp
mkdir -p gpurun_out/r2j
timeout 1200 python3 -m pytest tests/test_gpu_align.py tests/test_gpu_pipe.py -x -q > gpurun_out/r2j/pytest.txt 2>&1
tail -8 gpurun_out/r2j/pytest.txt
timeout 600 python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-end-to-end > gpurun_out/r2j/bench.json 2> gpurun_out/r2j/bench.err
python3 -c "
import json; d=json.load(open('gpurun_out/r2j/bench.json')); print(d['value'], d['ms_per_step'], {k:v for k,v in d['kernel_ms_per_step'].items() if k.startswith('k1')}, d['aligned_frac'])"
