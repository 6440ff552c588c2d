export TMPDIR=/tmp
for st in 10 20 40; do
python3 bench.py --no-cpu-baseline --no-end-to-end --steps $st --warmup 5 > gpurun_out/st.json 2> gpurun_out/st.err
python3 -c "
import json; d=json.load(open('gpurun_out/st.json')); print($st, d['value'], d['ms_per_step'], d['host_wall_ms_per_step'])"
done
