export TMPDIR=/tmp
timeout 900 python3 tools/cns_accuracy.py 2>&1 | tail -6
timeout 600 python3 bench.py --strong --contigs 40 --no-cpu-baseline --no-end-to-end --steps 3 --warmup 1 2>/dev/null | python3 -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('strong', d['value'], d['ms_per_step'], d['scaling'], d['config']['reads_total'])"
