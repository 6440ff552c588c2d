export TMPDIR=/tmp
mkdir -p gpurun_out/r2w
timeout 1500 python3 -m pytest tests/test_gpu_scale.py tests/test_gpu_align.py tests/test_gpu_pipe.py -x -q 2>&1 | tail -5
timeout 600 python3 bench.py > gpurun_out/r2w/bench.json 2> gpurun_out/r2w/bench.err
python3 -c "
import json; d=json.load(open('gpurun_out/r2w/bench.json')); print(d['value'], d['ms_per_step'], d['upload_ms'], d['end_to_end'])"
