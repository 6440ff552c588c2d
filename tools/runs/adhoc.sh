export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 600 python3 -m pytest -m gpu -x -q tests/test_gpu_align.py -k "border_reached or three_dp" > gpurun_out/r4b_fixed.log 2>&1; echo "fixed rc=$?"
sed -i 's/safe = min(rows_left, cols_left) - 1;/safe = min(rows_left, cols_left);/' falcon_unzip_amd/csrc/fzp_align.hip
make -C falcon_unzip_amd/csrc -j4 > gpurun_out/r4b_make.log 2>&1; echo "make rc=$?"
timeout 600 python3 -m pytest -m gpu -x -q tests/test_gpu_align.py -k "border_reached" > gpurun_out/r4b_old.log 2>&1; echo "old-safe rc=$? (expected non-zero)"
tail -5 gpurun_out/r4b_fixed.log; grep -n "^E  " gpurun_out/r4b_old.log | head -5
