export TMPDIR=/tmp
mkdir -p gpurun_out/r2d
timeout 900 python3 bench.py > gpurun_out/r2d/bench.json 2> gpurun_out/r2d/bench.err
tail -3 gpurun_out/r2d/bench.err
python3 -c "
import json; d=json.load(open('gpurun_out/r2d/bench.json')); print(d['value'], d['ms_per_step'], d['host_wall_ms_per_step'], d['end_to_end'], d['upload_ms'], d['cpu_baseline'], d['roofline']['valu'])"
timeout 900 python3 -m pytest tests/test_gpu_bench_ranks.py -x -q 2>&1 | tail -5
