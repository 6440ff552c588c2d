export TMPDIR=/tmp
mkdir -p gpurun_out/r2c
timeout 1200 python3 -m pytest tests/test_gpu_pipe.py tests/test_gpu_pipeline.py tests/test_gpu_align.py -x -q > gpurun_out/r2c/pytest.txt 2>&1
tail -25 gpurun_out/r2c/pytest.txt
