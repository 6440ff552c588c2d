export TMPDIR=/tmp
mkdir -p gpurun_out/r2h
free -g | head -2
timeout 1500 python3 tools/run_cfg5.py --scale 0.1 --workers 4 > gpurun_out/r2h/cfg5_s01.json 2> gpurun_out/r2h/cfg5_s01.err; tail -3 gpurun_out/r2h/cfg5_s01.err; cat gpurun_out/r2h/cfg5_s01.json
timeout 2400 python3 tools/run_cfg5.py --scale 1.0 --workers 12 > gpurun_out/r2h/cfg5.json 2> gpurun_out/r2h/cfg5.err; tail -3 gpurun_out/r2h/cfg5.err; cat gpurun_out/r2h/cfg5.json
