export TMPDIR=/tmp
mkdir -p gpurun_out/r2e
for g in 5 10; do for l in 2 3; do
timeout 900 python3 bench.py --no-cpu-baseline --steps 2 --e2e-lanes $l --e2e-group-contigs $g > gpurun_out/r2e/bench_${g}_${l}.json 2> gpurun_out/r2e/bench.err
python3 -c "
import json,sys; d=json.load(open('gpurun_out/r2e/bench_${g}_${l}.json')); print($g,$l,d['value'], d['ms_per_step'], d['end_to_end']['reads_per_s'], d['end_to_end']['ms'], d['end_to_end']['first_call_ms'], d['end_to_end']['host_section_ms_summed_over_lanes'], d['upload_ms'])"
done; done
