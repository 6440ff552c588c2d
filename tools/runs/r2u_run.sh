export TMPDIR=/tmp
mkdir -p gpurun_out/r2u
echo "--- library first"
timeout 900 python3 tools/e2e_sweep.py 2>&1 | tail -5
echo "--- torch first"
SWEEP_TORCH_FIRST=1 timeout 900 python3 tools/e2e_sweep.py 2>&1 | tail -5
