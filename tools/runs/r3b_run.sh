export TMPDIR=/tmp
timeout 900 python3 -m pytest tests/test_gpu_pipe.py tests/test_gpu_align.py -x -q 2>&1 | tail -3
SWEEP_TORCH_FIRST=1 timeout 900 python3 tools/e2e_sweep.py 2>&1 | grep "group_contigs" | cut -c1-250
