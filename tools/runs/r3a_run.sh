export TMPDIR=/tmp
mkdir -p gpurun_out/r3a
python3 bench.py --no-cpu-baseline > gpurun_out/r3a/bench.json 2> gpurun_out/r3a/bench.err; tail -3 gpurun_out/r3a/bench.err
python3 -c "
import json; d=json.load(open('gpurun_out/r3a/bench.json')); print(d['value'], d['ms_per_step'], d['end_to_end']['ms'], d['two_steps_in_flight'])"
