#!/usr/bin/env python3
"""Measurement aid: fzp_polish_tigs on the bench step's inputs (every contig a tig), three calls; run under rocprofv3 --kernel-trace --stats for the kernels' share."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bench
from falcon_unzip_amd import _lib

contigs, blob, off, rc = bench.make_inputs(2, list(range(20)), 5_000_000, lambda ci: 2000, 15000, 750_000, 1)
eng = _lib.Engine(0)
for k in range(3):
    t0 = time.perf_counter()
    t = _lib.polish_tigs(eng, contigs, blob, off, rc)
    print("call %d: %.1f ms" % (k, (time.perf_counter() - t0) * 1e3), flush=True)
    t.close()
eng.close()
