#!/usr/bin/env python3
"""Measurement aid: where a bench step's host time goes between the body of fzp_job_phase_write and the next call (the Python binding's argument set-up and result copy, the
library's epilogue)."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bench
from falcon_unzip_amd import _lib

contigs, blob, off, rc = bench.make_inputs(2, list(range(20)), 5_000_000, lambda ci: 2000, 15000, 750_000, 1)
ids = ["%06dF" % i for i in range(20)]
name_tab, maps = bench.make_names_and_maps(rc, off, ids, 0)
eng = _lib.Engine(0)
job = _lib.align_job_raw(eng, contigs, blob, off, rc)
root = "/dev/shm/fzp_host_tail"
acc = {"args": 0.0, "call": 0.0, "result": 0.0}
orig_args, orig_res = _lib._pipe_args, _lib._pipe_result
lib = _lib.load()
orig_call = lib.fzp_job_phase_write


def t_args(*a, **k):
    t = time.perf_counter(); r = orig_args(*a, **k); acc["args"] += time.perf_counter() - t; return r


def t_res(*a, **k):
    t = time.perf_counter(); r = orig_res(*a, **k); acc["result"] += time.perf_counter() - t; return r


_lib._pipe_args, _lib._pipe_result = t_args, t_res
for k in range(5):
    job.phase_write(ids, names=name_tab, out_dir="%s/w%d" % (root, k), read_maps=maps, ctg_index=list(range(20)), async_writes=True, rebuild_index=True)
eng.synchronize(); eng.pipe_flush()
for k in acc:
    acc[k] = 0.0
N = 20
t0 = time.perf_counter()
for k in range(N):
    t = time.perf_counter()
    job.phase_write(ids, names=name_tab, out_dir="%s/s%d" % (root, k), read_maps=maps, ctg_index=list(range(20)), async_writes=True, rebuild_index=True)
    acc["call"] += time.perf_counter() - t
eng.synchronize(); eng.pipe_flush()
dt = time.perf_counter() - t0
print("per step: loop %.3f ms, phase_write (python) %.3f, of it _pipe_args %.3f, _pipe_result %.3f" % (dt / N * 1e3, acc["call"] / N * 1e3, acc["args"] / N * 1e3, acc["result"] / N * 1e3))
job.close(); eng.close()
import shutil
shutil.rmtree(root, ignore_errors=True)
