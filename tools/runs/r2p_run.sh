export TMPDIR=/tmp
mkdir -p gpurun_out/r2p
rocprofv3 --kernel-trace --stats -d gpurun_out/r2p/kt -o kt --output-format csv -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-end-to-end --gen-workers 1 > gpurun_out/r2p/line.json 2> gpurun_out/r2p/kt.log
grep -E "k_plan|k_rank|k_gather|k_sec_count|k_pick" gpurun_out/r2p/kt/kt_kernel_stats.csv | cut -c1-60,200-
