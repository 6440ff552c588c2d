export TMPDIR=/tmp
mkdir -p gpurun_out/r2y
( time timeout 1500 python3 -m pytest tests/ -x -q -m gpu --durations=8 ) 2>&1 | tail -18
( time python3 bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r2y/bench20.json 2> gpurun_out/r2y/bench20.err ) 2>&1 | tail -4
python3 -c "
import json; d=json.load(open('gpurun_out/r2y/bench20.json')); print(d['value'], d['ms_per_step'], d['host_wall_ms_per_step'], d['end_to_end']['ms'])"
python3 -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
