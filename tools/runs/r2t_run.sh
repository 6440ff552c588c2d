export TMPDIR=/tmp
mkdir -p gpurun_out/r2t
timeout 1500 python3 -m pytest tests/test_gpu_scale.py -x -q --durations=5 2>&1 | tail -12
true
