export TMPDIR=/tmp
timeout 600 python3 bench.py > gpurun_out/r4z_bench.json 2> gpurun_out/r4z_bench.err; tail -c 200 gpurun_out/r4z_bench.err
