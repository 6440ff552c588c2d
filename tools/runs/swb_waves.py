#!/usr/bin/env python3
"""Measurement aid: the schedule of k_swb's workgroups on the bench workload (FZP_SWB_WAVE_LOG): when each ran, how many steps, how busy the SIMD slots were."""
import ctypes as C
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
os.environ["FZP_SWB_WAVE_LOG"] = "1"
import bench
from falcon_unzip_amd import _lib

contigs, blob, off, rc = bench.make_inputs(2, list(range(20)), 5_000_000, lambda ci: 2000, 15000, 750_000, 1)
eng = _lib.Engine(0)
job = _lib.align_job_raw(eng, contigs, blob, off, rc)
for _ in range(int(os.environ.get("FZP_WAVES_RUNS", "2"))):      # (the schedule of the LAST run; many runs back to back show what the chip sustains)
    job.run()
lib = _lib.load()
cap = 1 << 19
buf = np.zeros((cap, 4), np.uint64)
lib.fzp_debug_swb_waves.restype = C.c_int64
lib.fzp_debug_swb_waves.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64]
n = lib.fzp_debug_swb_waves(eng._p, job._p, buf.ctypes.data_as(C.c_void_p), cap)
w = buf[:n]
w = w[w[:, 1] > 0]
t0 = w[:, 0].min()
st, en = (w[:, 0] - t0) / 100.0, (w[:, 1] - t0) / 100.0      # microseconds
steps = w[:, 3].astype(np.int64)
grp = (w[:, 2] >> np.uint64(32)).astype(np.int64)      # r6: an entry is a work unit (FZP_SWB_UNIT blocks of a launch group)
if os.environ.get("FZP_WAVES_DUMP"):
    np.savez_compressed(os.environ["FZP_WAVES_DUMP"], st=st, en=en, steps=steps, grp=grp, hw=(w[:, 2] & np.uint64(0xffffffff)))
print("units", len(w), "groups", len(np.unique(grp)), "span %.1f us" % en.max(), "sum of run times %.1f ms" % ((en - st).sum() / 1e3), "-> mean busy slots %.0f" % ((en - st).sum() / en.max()))
print("steps: total %d, ns/step median %.1f p10 %.1f p90 %.1f" % (steps.sum(), np.median((en - st)[steps > 500] * 1e3 / steps[steps > 500]), *np.percentile((en - st)[steps > 500] * 1e3 / steps[steps > 500], [10, 90])))
full = steps == steps.max()
if full.sum() > 10:
    d = (en - st)[full]
    print("full units (%d steps): %d, us median %.1f p10 %.1f p90 %.1f p99 %.1f max %.1f" % (steps.max(), full.sum(), np.median(d), *np.percentile(d, [10, 90, 99]), d.max()))
for lo in np.arange(0, en.max(), en.max() / 16):
    hi = lo + en.max() / 16
    busy = np.clip(np.minimum(en, hi) - np.maximum(st, lo), 0, None).sum() / (hi - lo)
    print("  %7.0f-%7.0f us: %6.0f waves running, %5d started" % (lo, hi, busy, ((st >= lo) & (st < hi)).sum()))
hw = w[:, 2]
print("distinct hardware ids (cu/simd/se/xcc bits)", len(np.unique(hw & np.uint64(0xfffffff0))))
