export TMPDIR=/tmp
timeout 1500 python3 -m pytest tests/test_gpu_scale.py tests/test_gpu_align.py tests/test_gpu_pipe.py -x -q 2>&1 | tail -3
