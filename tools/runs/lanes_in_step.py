"""measurement aid: the bench step's 20 contigs as P sub-jobs over L contexts (lanes), one batch at a time -- what does a step cost when the lanes' gaps and tails overlap?
usage: python3 tools/runs/lanes_in_step.py [steps]"""
import os, sys, time, threading, tempfile, shutil
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import bench
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 8
NC = 20
contigs, blob, off, rc = bench.make_inputs(2, list(range(NC)), 5_000_000, lambda ci: 2000, 15000, 750_000, 8)
ids = ["%06dF" % c for c in range(NC)]
(noff, names), maps = bench.make_names_and_maps(rc, off, ids, 0)
from falcon_unzip_amd import _lib
root = tempfile.mkdtemp(prefix="lanes_", dir="/dev/shm")
blob_np = np.frombuffer(blob, np.uint8)

def sub(c0, c1):
    r = np.nonzero((rc >= c0) & (rc < c1))[0]
    r0, r1 = int(r[0]), int(r[-1]) + 1
    o = off[r0:r1 + 1] - off[r0]
    b = blob_np[off[r0]:off[r1]].tobytes()
    rs = (rc[r0:r1] - c0).astype(np.int32)
    (no, nm), m = bench.make_names_and_maps(rs, o, ids[c0:c1], 0)      # (the read_map files of a sub-job number its reads from 0)
    return contigs[c0:c1], b, o, rs, ids[c0:c1], (no, nm), m, list(range(c0, c1))

def run(P, L, tag):
    engs = [_lib.Engine(0) for _ in range(L)]
    bounds = [NC * k // P for k in range(P + 1)]
    parts = []
    for k in range(P):
        cs, b, o, r, idk, nt, m, mine = sub(bounds[k], bounds[k + 1])
        e = engs[k % L]
        parts.append((k % L, _lib.align_job_raw(e, cs, b, o, r), idk, nt, m, mine))
    err = []
    n = [0]
    def lane(li, d):
        try:
            for (l, job, idk, nt, m, mine) in parts:
                if l == li:
                    job.phase_write(idk, names=nt, out_dir=d, read_maps=m, ctg_index=mine, async_writes=True, rebuild_index=True)
        except Exception as e:
            err.append(repr(e))
    def step():
        n[0] += 1
        d = os.path.join(root, "%s_%d" % (tag, n[0]))
        th = [threading.Thread(target=lane, args=(li, d)) for li in range(1, L)]
        for t in th: t.start()
        lane(0, d)
        for t in th: t.join()
    for _ in range(3): step()
    for e in engs: e.synchronize(); e.pipe_flush()
    t0, c0 = time.perf_counter(), time.process_time()
    for _ in range(steps): step()
    for e in engs: e.synchronize(); e.pipe_flush()
    dt, dc = time.perf_counter() - t0, time.process_time() - c0
    print("P=%d L=%d  %.3f ms per batch   host cpu %.1f ms per batch %s" % (P, L, dt / steps * 1e3, dc / steps * 1e3, err[:1]), flush=True)
    for p in parts: p[1].close()
    for e in engs: e.close()
    shutil.rmtree(root, ignore_errors=True); os.makedirs(root, exist_ok=True)

for P, L in ((1, 1), (2, 2), (4, 2), (3, 3), (6, 3), (2, 1), (1, 1)):
    run(P, L, "p%dl%d" % (P, L))
shutil.rmtree(root, ignore_errors=True)
