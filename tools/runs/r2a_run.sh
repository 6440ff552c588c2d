export TMPDIR=/tmp
mkdir -p gpurun_out/r2a
./tools/ubench/valu_issue > gpurun_out/r2a/valu_issue.txt 2>&1
rocm-smi --showclocks > gpurun_out/r2a/clocks.txt 2>&1
rocprofv3 --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_WAVES -d gpurun_out/r2a/sq1 -o p --output-format csv -- python3 bench.py --steps 1 --warmup 0 --no-cpu-baseline --gen-workers 1 > gpurun_out/r2a/sq1_line.json 2> gpurun_out/r2a/sq1.log
rocprofv3 --pmc SQ_INSTS_SALU SQ_INSTS_SMEM SQ_WAVE_CYCLES SQ_INST_CYCLES_VMEM -d gpurun_out/r2a/sq2 -o p --output-format csv -- python3 bench.py --steps 1 --warmup 0 --no-cpu-baseline --gen-workers 1 > gpurun_out/r2a/sq2_line.json 2> gpurun_out/r2a/sq2.log
rocprofv3 --pmc SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_SCA SQ_WAIT_ANY -d gpurun_out/r2a/sq3 -o p --output-format csv -- python3 bench.py --steps 1 --warmup 0 --no-cpu-baseline --gen-workers 1 > gpurun_out/r2a/sq3_line.json 2> gpurun_out/r2a/sq3.log
python3 -m pytest tests -m gpu -x -q > gpurun_out/r2a/pytest.txt 2>&1
tail -3 gpurun_out/r2a/pytest.txt
head -30 gpurun_out/r2a/valu_issue.txt
