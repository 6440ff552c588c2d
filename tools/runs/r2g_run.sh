export TMPDIR=/tmp
mkdir -p gpurun_out/r2g
timeout 900 python3 -m pytest tests/test_gpu_comm.py -x -q -rs > gpurun_out/r2g/pytest.txt 2>&1
tail -15 gpurun_out/r2g/pytest.txt
