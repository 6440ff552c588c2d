export TMPDIR=/tmp
mkdir -p gpurun_out/r2o
timeout 1500 python3 -m pytest tests/test_gpu_align.py tests/test_gpu_pipe.py tests/test_gpu_parity.py tests/test_gpu_scale.py -x -q > gpurun_out/r2o/pytest.txt 2>&1
tail -6 gpurun_out/r2o/pytest.txt
timeout 600 python3 bench.py --no-cpu-baseline --no-end-to-end > gpurun_out/r2o/bench.json 2> gpurun_out/r2o/bench.err
python3 -c "
import json; d=json.load(open('gpurun_out/r2o/bench.json')); print(d['value'], d['ms_per_step'], d['kernel_ms_per_step']['k1_plan'])"
