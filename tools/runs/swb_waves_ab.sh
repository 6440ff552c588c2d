#!/bin/bash
# the schedule of k_swb's waves (FZP_SWB_WAVE_LOG) per "<waves>:<unit>:<hyst>" triple
export TMPDIR=/tmp
mkdir -p gpurun_out/swbw
for v in $1; do
  IFS=: read w u h d <<< "$v"
  echo "== waves $w unit $u hyst ${h:-0}"
  FZP_SWB_WAVES=$w FZP_SWB_UNIT=$u FZP_SWB_HYST=${h:-0} FZP_SWB_DBG=${d:-0} FZP_WAVES_DUMP=gpurun_out/swbw/w${w}_u${u}_h${h:-0}_d${d:-0}.npz python3 tools/runs/swb_waves.py 2>&1 | tail -22
done
