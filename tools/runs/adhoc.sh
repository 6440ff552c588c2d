export TMPDIR=/tmp
timeout 600 python3 -m pytest -m gpu -x -q tests/test_gpu_align.py 2>&1 | tail -6
