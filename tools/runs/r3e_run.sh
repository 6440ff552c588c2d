export TMPDIR=/tmp
timeout 900 python3 -m pytest tests/test_gpu_bench_ranks.py -x -q 2>&1 | tail -12
