export TMPDIR=/tmp
timeout 1500 python3 -m pytest tests/test_gpu_parity.py -x -q -k "long_clips or random" 2>&1 | tail -8
