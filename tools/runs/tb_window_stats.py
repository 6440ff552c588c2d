#!/usr/bin/env python3
"""Measurement aid: how often the traced path leaves a 32-lane window of the 64-lane band -- the fixed middle lanes 16..47, and a window centred
where the band's edge scores (E2, in the move words) put the path.  Bench workload and reads of real shape."""
import ctypes as C
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
os.environ["FZP_TB_STATS"] = "1"
os.environ["FZP_SWB_64"] = "1"
import bench
from falcon_unzip_amd import _lib

lib = _lib.load()
lib.fzp_debug_tb_stats.restype = C.c_int
lib.fzp_debug_tb_stats.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
eng = _lib.Engine(0)
for tag, gen in (("cfg2 (15 kb reads, 13 % error)", None), ("reads of real shape (3-60 kb, bursts at 30 %)", bench.gen_contig_shaped)):
    contigs, blob, off, rc = bench.make_inputs(2, list(range(4)), 5_000_000, lambda ci: 2000, 15000, 750_000, 1, gen=gen)
    job = _lib.align_job_raw(eng, contigs, blob, off, rc)
    job.run()
    st = np.zeros(16, np.uint64)
    assert lib.fzp_debug_tb_stats(eng._p, job._p, st.ctypes.data_as(C.c_void_p)) == 0
    st = st.astype(np.float64)
    print(tag)
    print("  pieces walked %d, steps %d" % (st[0], st[3]))
    print("  fixed window 16..47   : %.4f %% of the pieces leave it (%.5f %% of the steps)" % (100 * st[1] / st[0], 100 * st[4] / st[3]))
    print("  window around 32+E2/3 : %.4f %% of the pieces leave it (%.5f %% of the steps)" % (100 * st[2] / st[0], 100 * st[5] / st[3]))
    print("  pieces by largest distance from the band's centre, lanes 31|32 (0-1, 2-3, ..., 18+):", [int(x) for x in st[6:16]])
    job.close()
