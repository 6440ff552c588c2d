#!/usr/bin/env python3
"""Measurement aid: the schedule of k_tb_walk_h's waves on the bench workload (FZP_TBH_WAVE_LOG with the -DFZP_TBH_LOG build, FZP_LIB): when each wave ran, how many
iterations, how its shader cycles split between the step loops and the staging (park, move words, next prefetch)."""
import ctypes as C
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
os.environ["FZP_TBH_WAVE_LOG"] = "1"
import bench
from falcon_unzip_amd import _lib

contigs, blob, off, rc = bench.make_inputs(2, list(range(20)), 5_000_000, lambda ci: 2000, 15000, 750_000, 1)
eng = _lib.Engine(0)
job = _lib.align_job_raw(eng, contigs, blob, off, rc)
job.run()
job.run()
lib = _lib.load()
cap = 1 << 16
buf = np.zeros((cap, 4), np.uint64)
lib.fzp_debug_swb_waves.restype = C.c_int64
lib.fzp_debug_swb_waves.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64]
n = lib.fzp_debug_swb_waves(eng._p, job._p, buf.ctypes.data_as(C.c_void_p), cap)
w = buf[:n].reshape(-1, 8)
w = w[w[:, 1] > 0]
t0 = w[:, 0].min()
st, en = (w[:, 0] - t0) / 100.0, (w[:, 1] - t0) / 100.0      # microseconds
inner, stage, iters = w[:, 2].astype(np.float64), w[:, 3].astype(np.float64), w[:, 4].astype(np.float64)
life = en - st
print("waves", len(w), "span %.1f us" % en.max(), "sum of lifetimes %.1f ms -> mean resident waves %.0f" % (life.sum() / 1e3, life.sum() / en.max()))
print("lifetime us: median %.0f p10 %.0f p90 %.0f max %.0f" % (np.median(life), *np.percentile(life, [10, 90]), life.max()))
print("iterations per wave: median %.0f max %.0f" % (np.median(iters), iters.max()))
cyc = inner + stage
print("shader-clock counts per wave: step loops %.2e staging %.2e (%.0f %% / %.0f %%)" % (np.median(inner), np.median(stage), 100 * inner.sum() / cyc.sum(), 100 * stage.sum() / cyc.sum()))
print("per iteration: step loop %.0f counts, staging %.0f counts; counts per us of lifetime %.1f" % (np.median(inner / np.maximum(iters, 1)), np.median(stage / np.maximum(iters, 1)), np.median(cyc / np.maximum(life, 1e-3))))
for lo in np.arange(0, en.max(), en.max() / 16):
    hi = lo + en.max() / 16
    busy = np.clip(np.minimum(en, hi) - np.maximum(st, lo), 0, None).sum() / (hi - lo)
    print("  %7.0f-%7.0f us: %6.0f waves running, %5d started, %5d ended" % (lo, hi, busy, ((st >= lo) & (st < hi)).sum(), ((en >= lo) & (en < hi)).sum()))
