export TMPDIR=/tmp
for mode in staged direct register; do
echo "--- $mode, library first"; FZP_UPLOAD_MODE=$mode timeout 900 python3 tools/e2e_sweep.py 2>&1 | grep group_contigs | cut -c1-330
echo "--- $mode, torch first"; FZP_UPLOAD_MODE=$mode SWEEP_TORCH_FIRST=1 timeout 900 python3 tools/e2e_sweep.py 2>&1 | grep group_contigs | cut -c1-330
done
