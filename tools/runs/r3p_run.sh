export TMPDIR=/tmp
python3 - <<'PY'
import sys, ctypes as C, numpy as np
sys.path.insert(0, '.')
import bench
from falcon_unzip_amd import _lib
contigs, blob, off, rc = bench.make_inputs(2, [0,1], 5_000_000, lambda ci: 2000, 15000, 750_000, 2)
eng = _lib.Engine(0)
job = _lib.align_job_raw(eng, contigs, blob, off, rc)
job.run()
h = (C.c_ulonglong * 16)()
_lib.load().fzp_debug_khist(h)
h = np.array(list(h), dtype=np.float64)
print("steps", h.sum()); print("lane/4 histogram (%):", np.round(100*h/h.sum(), 4))
print("outside [16,48):", 100*(h[:4].sum()+h[12:].sum())/h.sum(), "%   outside [8,56):", 100*(h[:2].sum()+h[14:].sum())/h.sum(), "%")
PY
