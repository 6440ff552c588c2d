#!/bin/bash
# fzalign v1.6 sampling (FZP_SEED_ANCHORED=0) against v1.7 (anchored k-mers) on the same box: alignment tests, then the bench step
export TMPDIR=/tmp
out=gpurun_out/seedab; mkdir -p $out; rm -f $out/*
B="--no-cpu-baseline --no-end-to-end --no-two-core --no-from-files --steps 20 --warmup 3"
for a in 1 0 1; do
  FZP_SEED_ANCHORED=$a python3 bench.py $B > $out/a$a.json 2> $out/a$a.log
  python3 - <<PY
import json
d=json.loads([l for l in open("$out/a$a.json") if l.startswith("{")][0])
k=d["kernel_ms_per_step"]
print("anchored=$a", "ms/step", d["ms_per_step"], {x: k[x] for x in ("k1_index","k1_seed","k1_sw","k1_traceback","k1_cigar")}, "aligned", d["aligned_frac"], "shaped", {x: d["k1_on_real_read_shape"][x] for x in ("k1_seed_ms","k1_sw_ms","aligned_frac","bases_inside_alignments_frac")})
PY
done
timeout 1200 python3 -m pytest tests/test_gpu_align.py tests/test_gpu_scale.py -x -q -m gpu 2>&1 | tail -3
