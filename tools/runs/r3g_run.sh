export TMPDIR=/tmp
mkdir -p gpurun_out/r3g
timeout 1500 python3 tools/run_cfg5.py --workers 32 > gpurun_out/r3g/cfg5.json 2> gpurun_out/r3g/cfg5.err; tail -3 gpurun_out/r3g/cfg5.err; cat gpurun_out/r3g/cfg5.json | cut -c1-1500
