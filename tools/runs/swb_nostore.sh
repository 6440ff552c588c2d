#!/bin/bash
# measurement aid: k_swb's own duration (rocprofv3 kernel stats) with and without its mask stores (FZP_SWB_DBG=1: results invalid -- the run that follows is thrown away), default build and register-budget variants
export TMPDIR=/tmp
out=gpurun_out/swbn; mkdir -p $out; rm -rf $out/*
B="--no-cpu-baseline --no-end-to-end --no-shaped-leg --no-two-core --no-from-files --no-kernel-breakdown --gen-workers 1 --steps 3 --warmup 1"
for t in default "$@"; do
  if [ $t = default ]; then unset FZP_LIB; else export FZP_LIB=$PWD/tools/runs/build/libfzphase_$t.so; fi
  for dbg in 0 1 3; do
    FZP_SWB_DBG=$dbg rocprofv3 --kernel-trace --stats -d $out/kt_${t}_$dbg -o kt --output-format csv -- python3 bench.py $B > /dev/null 2> $out/${t}_$dbg.log
    echo "$t dbg=$dbg $(grep -h 'k_swb<' $(find $out/kt_${t}_$dbg -name '*kernel_stats.csv') | cut -d, -f1-6 | cut -c1-40,200-)"
    grep -h 'k_swb<' $(find $out/kt_${t}_$dbg -name '*kernel_stats.csv') | awk -F'","' '{print "   calls", $2, "total ns", $3, "avg ns", $4}'
  done
done
