export TMPDIR=/tmp
FZP_BENCH_BACKEND=gloo timeout 300 python3 bench.py --gpus 2 --contigs 2 --contig-len 300000 --reads-per-contig 150 --read-len 8000 --window 120000 --steps 2 --warmup 1 --no-cpu-baseline --gen-workers 1 --strong-leg-contigs 5 --strong-leg-contig-len 200000 2>&1 | grep -v "^{" | grep -B2 -A12 "Traceback" | head -60 | cut -c1-300
