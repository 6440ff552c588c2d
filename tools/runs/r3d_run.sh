export TMPDIR=/tmp
mkdir -p gpurun_out/r3d
( time python3 bench.py > gpurun_out/r3d/bench.json 2> gpurun_out/r3d/bench.err ) 2>&1 | tail -3
python3 -c "
import json; d=json.load(open('gpurun_out/r3d/bench.json')); print(d['value'], d['ms_per_step'], d['steps'], d['host_wall_ms_per_step'], d['end_to_end']['ms'], d['end_to_end']['lanes'], d['two_steps_in_flight'], d['cpu_baseline']['value'])"
