export TMPDIR=/tmp
timeout 900 python3 -m pytest tests/test_gpu_scale.py -x -q -k cfg5 2>&1 | tail -3
