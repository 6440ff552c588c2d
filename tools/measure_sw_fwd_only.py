#!/usr/bin/env python3
"""DESIGN section 14, the trace-back design question: what would a checkpoint + recompute trace-back cost?  Its forward pass is k_sw
without the per-step mask stores, and its recompute pass is a second DP over every step of the winning path's prefix (the masks of a
64-step segment depend on ALL 64 lanes of the checkpointed rows: a band-limited recompute is not exact) -- so the price is two launches
of the store-free kernel against today's one launch with stores + the walk.  This tool times k_sw with and without its
s_store_dwordx4 (FZP_SW_NO_MASKS=1: a measurement switch, the alignments of such a run are not used) on the bench workload.
usage (GPU box): python3 tools/measure_sw_fwd_only.py > gpurun_out/r3_sw_fwd_only.json"""
import json
import os
import sys

import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
import bench  # noqa: E402


def main():
    n_ctg = int(os.environ.get("N_CTG", "20"))
    contigs, blob, off, read_ctg = bench.make_inputs(2, list(range(n_ctg)), 5_000_000, lambda ci: 2000, 15000, 750_000, 8)
    from falcon_unzip_amd import _lib
    eng = _lib.Engine(0)
    job = _lib.align_job_raw(eng, contigs, blob, off, read_ctg)
    out = {}
    for mode in ("with_mask_stores", "without_mask_stores", "with_mask_stores_again"):
        if mode == "without_mask_stores":
            os.environ["FZP_SW_NO_MASKS"] = "1"
        else:
            os.environ.pop("FZP_SW_NO_MASKS", None)
        job.run()
        eng.synchronize()
        eng.prof_reset()
        eng.prof_enable(True)
        for _ in range(3):
            job.run()
        eng.synchronize()
        eng.prof_enable(False)
        pr = eng.prof()
        out[mode] = {k: round(pr[k][0] / max(1, pr[k][1]), 3) for k in ("k1_sw", "k1_traceback", "k1_cigar") if k in pr}
    cells = float(job.summaries()["cells"].sum())
    steps = cells / 64
    out["dp_steps_per_launch"] = steps
    out["mask_bytes_per_launch"] = steps * 16
    out["reading"] = ("checkpointed trace-back = forward pass without stores + a full-band recompute of the same steps + the walk in LDS: "
                      ">= 2 x k1_sw(without) ms, against k1_sw(with) + k1_traceback today; footprint per 15 kb read: 16 B/step x ~33 750 steps = 540 KB of masks "
                      "today, 8 B/step of checkpoints (H and X rows every 64 steps) = 270 KB with checkpoints every 64 steps, 34 KB with one every 512")
    job.close()
    eng.close()
    print(json.dumps(out))


if __name__ == "__main__":
    main()
