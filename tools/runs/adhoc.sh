export TMPDIR=/tmp
FZP_PIPE_TIMING=1 timeout 600 python3 tools/run_cfg5.py --from-files > gpurun_out/r4s_cfg5_files.json 2> gpurun_out/r4s_cfg5_files.err; grep "load_group\|phase_contigs_files" gpurun_out/r4s_cfg5_files.err | tail -6
FZP_PIPE_TIMING=1 timeout 600 python3 bench.py --no-cpu-baseline --no-shaped-leg > gpurun_out/r4k_bench.json 2> gpurun_out/r4k_bench.err; grep "phase_contigs_files\|load_group" gpurun_out/r4k_bench.err | tail -3
timeout 300 python3 -m pytest -m gpu -x -q tests/test_gpu_ranks.py -k "files_entry or two_ranks" 2>&1 | tail -2
