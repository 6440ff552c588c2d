export TMPDIR=/tmp
timeout 900 python3 -m pytest tests/test_gpu_pipe.py tests/test_gpu_pipeline.py tests/test_gpu_scale.py -x -q 2>&1 | tail -3
