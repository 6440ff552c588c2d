export TMPDIR=/tmp
python3 - <<'PY'
import sys, time, os, tempfile
sys.path.insert(0, '.')
import numpy as np
import bench
from falcon_unzip_amd import _lib
import ctypes as C
mine = list(range(20))
contigs, blob, off, rc = bench.make_inputs(2, mine, 5_000_000, lambda ci: 2000, 15000, 750_000, 8)
ids = ["%06dF" % ci for ci in mine]
name_tab, maps = bench.make_names_and_maps(rc, off, ids, 0)
eng = _lib.Engine(0)
job = _lib.align_job_raw(eng, contigs, blob, off, rc)
root = tempfile.mkdtemp(dir="/tmp")
lib = _lib.load()
for k in range(8):
    t0 = time.perf_counter()
    nm, opts, keep = _lib._pipe_args(ids, name_tab, os.path.join(root, "s%d" % k), maps, mine, 0, 0, 0, None, _lib.PIPE_ASYNC_WRITES)
    out = _lib.PipeOut()
    t1 = time.perf_counter()
    rc_ = lib.fzp_job_phase_write(eng._p, job._p, C.byref(nm), C.byref(opts), C.byref(out))
    t2 = time.perf_counter()
    st, recs = _lib._pipe_result(out)
    t3 = time.perf_counter()
    print("step %d: args %.3f ms, C call %.3f ms (sections %.3f), result %.3f ms" % (k, (t1-t0)*1e3, (t2-t1)*1e3, st['ms_k1']+st['ms_phase']+st['ms_results']+st['ms_text'], (t3-t2)*1e3))
eng.pipe_flush()
PY
