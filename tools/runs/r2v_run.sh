export TMPDIR=/tmp
mkdir -p gpurun_out/r2v
echo "--- system runtime"; timeout 300 python3 tools/ubench/h2d_driver.py 2>&1 | tee gpurun_out/r2v/h2d_system.txt
echo "--- torch runtime"; H2D_TORCH_FIRST=1 timeout 300 python3 tools/ubench/h2d_driver.py 2>&1 | tee gpurun_out/r2v/h2d_torch.txt
