export TMPDIR=/tmp
SWEEP_TORCH_FIRST=1 timeout 900 python3 tools/e2e_sweep.py 2>&1 | grep "group_contigs" | cut -c1-250
echo "--- system runtime"
timeout 900 python3 tools/e2e_sweep.py 2>&1 | grep "group_contigs" | cut -c1-250
