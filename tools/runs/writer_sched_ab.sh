#!/bin/bash
# the background file writers' scheduling class (FZP_WRITER_SCHED; default: idle for a rank with <= 4 cores): the two-core step with and without
export TMPDIR=/tmp
out=gpurun_out/wsched; mkdir -p $out; rm -f $out/*
for t in default other default2 other2; do
  case $t in other*) export FZP_WRITER_SCHED=other;; *) unset FZP_WRITER_SCHED;; esac
  python3 bench.py --no-cpu-baseline --no-end-to-end --no-shaped-leg --no-kernel-breakdown --steps 10 --warmup 3 > $out/$t.json 2> $out/$t.log
  python3 - <<PY
import json
d=json.loads([l for l in open("$out/$t.json") if l.startswith("{")][0])
print("$t", "ms/step", d["ms_per_step"], "two-core", d["two_core"]["ms_per_step"], d["two_core"]["vs_unconstrained"], d["two_core"]["cpu_ms_per_step_by_thread"])
PY
done
