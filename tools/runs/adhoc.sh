export TMPDIR=/tmp
timeout 600 python3 -m pytest -m gpu -x -q tests/test_gpu_align.py -k "outside_the_recorded" 2>&1 | tail -3
