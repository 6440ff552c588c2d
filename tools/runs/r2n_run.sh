export TMPDIR=/tmp
mkdir -p gpurun_out/r2n
python3 -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/r2n/smoke.txt 2>&1; tail -2 gpurun_out/r2n/smoke.txt
timeout 2000 python3 -m pytest tests -m gpu -x -q > gpurun_out/r2n/pytest_gpu.txt 2>&1
tail -5 gpurun_out/r2n/pytest_gpu.txt | head -3
