bash tools/profile_round.sh r2_a > gpurun_out/prof_r2_a.log 2>&1
tail -3 gpurun_out/prof_r2_a.log
timeout 1500 python3 -m pytest tests -m gpu -x -q > gpurun_out/pytest_gpu_r2_a.txt 2>&1
tail -5 gpurun_out/pytest_gpu_r2_a.txt
