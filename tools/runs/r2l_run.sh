export TMPDIR=/tmp
mkdir -p gpurun_out/r2l
timeout 1200 python3 -m pytest tests/test_gpu_pipe.py tests/test_gpu_bench_ranks.py -x -q > gpurun_out/r2l/pytest.txt 2>&1
tail -8 gpurun_out/r2l/pytest.txt
for i in 1 2; do
timeout 600 python3 bench.py --no-cpu-baseline > gpurun_out/r2l/bench$i.json 2> gpurun_out/r2l/bench.err
python3 -c "
import json; d=json.load(open('gpurun_out/r2l/bench$i.json')); print(d['value'], d['ms_per_step'], d['host_wall_ms_per_step'], d['end_to_end']['reads_per_s'])"
done
