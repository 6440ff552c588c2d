export TMPDIR=/tmp
mkdir -p gpurun_out/r2q
timeout 1200 python3 -m pytest tests/test_gpu_pipe.py -x -q > gpurun_out/r2q/pytest.txt 2>&1
tail -12 gpurun_out/r2q/pytest.txt
