export TMPDIR=/tmp
mkdir -p gpurun_out/r2i
timeout 1200 python3 -m pytest tests/test_gpu_cns.py -x -q > gpurun_out/r2i/pytest.txt 2>&1
tail -25 gpurun_out/r2i/pytest.txt
