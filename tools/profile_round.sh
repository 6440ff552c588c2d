#!/bin/bash
# usage (on the GPU box, from the repo root): bash tools/profile_round.sh <tag>
# kernel-trace stats of the bench, the two HBM counter passes and the SQ issue counters (each in its own rocprofv3 run, counters only;
# the program sits directly after `--` and never forks: --gen-workers 1 --no-tree-compare), then the default bench line
tag=$1
export TMPDIR=/tmp
out=gpurun_out/prof_$tag
mkdir -p $out
B="--no-cpu-baseline --no-end-to-end --no-shaped-leg --gen-workers 1 --no-tree-compare"
C="$B --no-kernel-breakdown"      # counter passes: exactly the launches of ONE step
rocprofv3 --kernel-trace --stats -d $out/kt -o kt --output-format csv -- python3 bench.py --steps 2 --warmup 1 $B > $out/kt_bench_line.json 2> $out/kt.log
rocprofv3 --pmc FETCH_SIZE -d $out/fetch -o p --output-format csv -- python3 bench.py --steps 1 --warmup 0 $C > /dev/null 2> $out/fetch.log
rocprofv3 --pmc WRITE_SIZE -d $out/write -o p --output-format csv -- python3 bench.py --steps 1 --warmup 0 $C > /dev/null 2> $out/write.log
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_WAVES -d $out/sq -o p --output-format csv -- python3 bench.py --steps 1 --warmup 0 $C > /dev/null 2> $out/sq.log
# where a wave's cycles go (r5): busy / waiting / issuing, LDS and store instructions, per kernel (two passes: the SQ block has few counters)
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY -d $out/sq2 -o p --output-format csv -- python3 bench.py --steps 1 --warmup 0 $C > /dev/null 2> $out/sq2.log
rocprofv3 --pmc SQ_WAIT_ANY SQ_INSTS_LDS SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD -d $out/sq3 -o p --output-format csv -- python3 bench.py --steps 1 --warmup 0 $C > /dev/null 2> $out/sq3.log
cat $(find $out/sq2 -name '*counter_collection.csv') > $out/sq_cycles_a.csv; cat $(find $out/sq3 -name '*counter_collection.csv') > $out/sq_cycles_b.csv
(cd tools && python3 sq_cycles_summary.py ../$out/sq_cycles_a.csv ../$out/sq_cycles_b.csv ../$out/sq_cycles.json)
python3 tools/pmc_hbm_summary.py $(find $out/fetch -name '*counter_collection.csv') $(find $out/write -name '*counter_collection.csv') $out/pmc_hbm_bytes.csv $out/k1_sw_hbm_traffic.json
cp $(find $out/kt -name '*kernel_stats.csv') $out/kernel_stats.csv
cp $(find $out/sq -name '*counter_collection.csv') $out/sq_counters.csv
(cd tools && python3 sq_counters_summary.py ../$out/sq_counters.csv ../$out/sq_counters.json ../$out/kt_bench_line.json ../$out/k1_sw_counters.json "profiles/${tag}_sq_counters.json (rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_WAVES, bench.py --steps 1)")
python3 bench.py > $out/bench_line.json 2> $out/bench.log
ls $out
