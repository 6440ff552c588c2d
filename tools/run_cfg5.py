#!/usr/bin/env python3
"""BASELINE configs[4] on ONE MI355X: an Arabidopsis-scale primary assembly (120 Mb in contigs of 1-10 Mb, log-uniform),
40x of 15 kb CLR reads (~320 k reads, 4.8 Gb), full phase + phased-pile consensus, through fzp_phase_contigs (contig groups
streamed over two lanes, files written).  Prints one JSON object: wall, reads/s, peak HBM, stage counts.
`--scale 0.1` is the slice tests/test_gpu_scale.py runs."""
import argparse
import json
import multiprocessing as mp
import os
import shutil
import sys
import tempfile
import threading
import time

import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)


def contig_lengths(total, rng):
    out = []
    while sum(out) < total:
        out.append(int(np.exp(rng.uniform(np.log(1e6), np.log(1e7)))))
    return out


def gen_contig(args):
    ci, L, cov, R, slab = args
    from falcon_unzip_amd import sim
    rng = sim.rng_for(5, ci)
    hap0, hap1, _ = sim.make_diploid(L, rng)
    n = int(L * cov / R)
    blobs, lens = [], []
    for s0 in range(0, n, slab):                       # slabs bound the generator's working set (~0.2 GB per 1000 reads)
        codes, off, *_ = sim.simulate_raw_reads_bulk(hap0, hap1, min(slab, n - s0), R, rng)
        blobs.append(sim.ACGT[codes].tobytes())
        lens.append(np.diff(off))
    return sim.ACGT[hap0].tobytes(), b"".join(blobs), np.concatenate(lens) if lens else np.zeros(0, np.int64)


def make_job(total_bases, cov=40, R=15000, workers=8, slab=2000, seed=20265000):
    rng = np.random.Generator(np.random.PCG64(seed))
    Ls = contig_lengths(total_bases, rng)
    jobs = [(ci, L, cov, R, slab) for ci, L in enumerate(Ls)]
    if workers > 1:
        with mp.get_context("spawn").Pool(min(workers, len(jobs))) as pool:      # (spawn: a caller that has initialised the GPU -- pytest -- must not fork)
            res = pool.map(gen_contig, jobs)
    else:
        res = [gen_contig(j) for j in jobs]
    contigs = [r[0] for r in res]
    blob = b"".join(r[1] for r in res)
    lens = np.concatenate([r[2] for r in res])
    off = np.zeros(len(lens) + 1, np.int64)
    off[1:] = np.cumsum(lens)
    read_ctg = np.concatenate([np.full(len(r[2]), c, np.int32) for c, r in enumerate(res)])
    ids = ["%06dF" % c for c in range(len(contigs))]
    return contigs, blob, off, read_ctg, ids


def run(scale=1.0, lanes=2, workers=8, out_root=None, keep=False, consensus=True, from_files=False):
    t0 = time.perf_counter()
    contigs, blob, off, read_ctg, ids = make_job(int(120e6 * scale), workers=workers)
    t_gen = time.perf_counter() - t0
    from bench import make_names_and_maps
    name_tab, maps = make_names_and_maps(read_ctg, off, ids, 0)
    from falcon_unzip_amd import _lib
    reads_dir = None
    if from_files:      # the reference's own input files on a memory file system; the library parses them (fzp_phase_contigs_files)
        from bench import write_reads_tree
        reads_dir = write_reads_tree(contigs, blob, off, read_ctg, ids, name_tab, "/dev/shm" if os.path.isdir("/dev/shm") else out_root)
    eng = _lib.Engine(0)
    mon = _lib.Engine(0)                       # a second context only to read the device's memory counters from the sampling thread
    total_mem = mon.mem_info()[1]
    peak = {"used": 0, "stop": False}

    def sample():
        while not peak["stop"]:
            fr, tot = mon.mem_info()
            peak["used"] = max(peak["used"], tot - fr)
            time.sleep(0.02)
    th = threading.Thread(target=sample, daemon=True)
    th.start()
    if out_root is None:                       # (r6: like bench.py, the output trees go to a memory file system when it has room -- on the box's disk-backed /tmp what a call
        from bench import shm_with_room        # takes depends on what earlier processes left in the page cache: 0.21 s or 0.44 s for the same tree)
        out_root = shm_with_room(int(4e9 * max(scale, 0.05)))
    out_dir = tempfile.mkdtemp(prefix="fzp_cfg5_", dir=out_root)
    # the first call also fills the contexts' block caches (150 GB of hipMalloc: 0.6 - 1.8 s depending on the box); the second is the job itself
    walls = []
    for k in range(4):
        t1 = time.perf_counter()
        if from_files:
            stats, recs = _lib.phase_contigs_files(eng, reads_dir, ids, out_dir=os.path.join(out_dir, "run%d" % k), read_maps=maps, n_lanes=lanes, consensus=consensus)
        else:
            stats, recs = _lib.phase_contigs(eng, contigs, blob, off, read_ctg, ids, names=name_tab, out_dir=os.path.join(out_dir, "run%d" % k), read_maps=maps, n_lanes=lanes,
                                             consensus=consensus)
        walls.append(time.perf_counter() - t1)
    wall = min(walls[1:])                      # (r6: the best of three calls behind the first, all four in `walls_s`: single calls vary by a factor of two on some boxes)
    out_dir = os.path.join(out_dir, "run3")
    peak["stop"] = True
    th.join()
    n_files = sum(len(f) for _, _, f in os.walk(out_dir))
    res = {"config": "configs[4] at scale %.2f: %d contigs, %.1f Mb, %d reads x 15 kb (%.2f Gb), 40x; K1..K5 + K6 consensus + all files, fzp_phase_contigs on %d lanes"
                     % (scale, len(contigs), sum(len(c) for c in contigs) / 1e6, len(read_ctg), len(blob) / 1e9, lanes),
           "wall_s": round(wall, 3), "first_call_wall_s": round(walls[0], 3), "walls_s": [round(w, 3) for w in walls], "reads_per_s": round(len(read_ctg) / wall, 1), "input_generation_s": round(t_gen, 1),
           "peak_hbm_gb": round(peak["used"] / 2**30, 2), "hbm_total_gb": round(total_mem / 2**30, 1),
           "dp_gcell_per_s_wall": round(stats["dp_cells"] / wall / 1e9, 1), "files_written": n_files, "r2p_records": int(len(recs)),
           "reads_phased": int((recs["block"] != -1).sum()), "stats": {k: (round(v, 2) if isinstance(v, float) else int(v)) for k, v in stats.items()},
           "longest_contig_reads": int(np.bincount(read_ctg).max()), "out_root": out_root or tempfile.gettempdir()}
    mon.close()
    eng.close()
    if reads_dir:
        res["inputs"] = "FASTA files on a memory file system (%s), parsed by the library" % os.path.dirname(reads_dir)
        shutil.rmtree(reads_dir, ignore_errors=True)
    if keep:
        res["out_dir"] = out_dir
    else:
        shutil.rmtree(os.path.dirname(out_dir), ignore_errors=True)
    return res


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--scale", type=float, default=1.0)
    ap.add_argument("--lanes", type=int, default=2)
    ap.add_argument("--workers", type=int, default=8)
    ap.add_argument("--out-root", default=None)
    ap.add_argument("--from-files", action="store_true")
    a = ap.parse_args()
    print(json.dumps(run(a.scale, a.lanes, a.workers, a.out_root, from_files=a.from_files)))
