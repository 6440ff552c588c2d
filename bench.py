#!/usr/bin/env python3
"""bench.py -- the hot path on synthetic input, one JSON line on stdout (rank 0).

Workload (BASELINE.json configs[1], SURVEY.md section 8d "cfg2"): per GPU 20 primary contigs of 5 Mb,
2 000 simulated PacBio CLR reads of 15 kb template each (sub 1 % / ins 8 % / del 4 %, both strands);
the reads of a contig are drawn from a 750 kb window of it, so the covered region is at 40x and the
het-call gate `total >= 10` (phasing.py:112) has something to call, while seeding and the banded DP
still run against the full 5 Mb contig.

One step = one pass of the whole hot path over that batch, inputs already resident (packed) in HBM, through the
library's fzp_job_phase_write:
  K1 index + seed + chain + banded DP + trace-back -> records ("samtools sort" order, record filters)
  K2 pileup + het call -> K3 association table -> K4 phase blocks -> K5 read phasing
  text of all seven files of every contig (variant_map / atable serialised on the device, the rest by host threads),
  the files WRITTEN under a scratch directory (the page-cache copies run on background threads of the library and may still be
  going on while the next step's kernels start; the closing barrier waits for every file of every step: fzp_pipe_flush),
  get_phasing_readmap -> rid_to_phase records
  -> one all-gather of the records across ranks (skipped at world size 1).
`value` = reads processed by all ranks / max-over-ranks step time.  `dp_gcell_per_s_per_gpu` is the banded-DP rate of
the dominant kernel (k1_sw) from HIP events on the library's stream.  `end_to_end` (not `value`: the bench contract keeps
inputs resident) times fzp_phase_contigs on the same workload from host buffers: pinned staging + H2D + 2-bit packing
included, contig groups on two lanes so that uploads and file writes hide behind kernels.
`--strong` switches to BASELINE configs[2]'s shape: a fixed set of contigs of uneven size dealt LPT over the ranks.

Ranks.  `python bench.py --gpus N` with N > 1 and no torch.distributed environment starts its own N ranks: before torch is imported or
anything touches the GPU, the process spawns `python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1
bench.py <same arguments>` as a CHILD, lets rank 0's JSON line through and exits with the children's code (never an exec).  Under a launcher
(RANK / WORLD_SIZE set) WORLD_SIZE must equal --gpus.  An N > 1 run also carries `strong_cfg3`: BASELINE configs[2] (500 contigs x 750 kb,
0.5x..2x 2 000 reads each, LPT over the ranks, streamed from host buffers through fzp_phase_contigs, one all-gather) inside the same line.
"""
from __future__ import annotations

import argparse
import json
import multiprocessing as mp
import os
import shutil
import sys
import tempfile
import time

import numpy as np

REPO = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, REPO)

HBM_PEAK_GBS = 8000.0            # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
SW_BYTES_PER_CELL = 0.25         # algorithmic: 2 trace-back bits per cell (16 B per 64-cell step); sequence
                                 # reads add 2 bits per band step, i.e. < 0.01 B/cell
# VALU view of k1_sw (the DP stage: k_swb, the bit-sliced kernel with one read per lane, + k_sw for long reads), from measurements kept under profiles/:
#   profiles/k1_sw_counters.json     SQ_INSTS_VALU of both kernels / 64-cell band steps of the bench step (k_swb: ~123 wave instructions per step of 64 pieces)
SW_VALU_PER_STEP = 2.0
N_SIMD, CLK_GHZ = 1024, 2.4


def gen_contig(args):
    cfg, ci, L, n_reads, R, win = args
    from falcon_unzip_amd import sim
    rng = sim.rng_for(cfg, ci)
    hap0, hap1, _ = sim.make_diploid(L, rng)
    lo = int(rng.integers(0, L - win + 1))
    codes, off, st, hp, sd = sim.simulate_raw_reads_bulk(hap0, hap1, n_reads, R, rng, lo=lo, hi=lo + win)
    return sim.ACGT[hap0].tobytes(), sim.ACGT[codes].tobytes(), off


def gen_contig_shaped(args):
    """the same contig as gen_contig, with reads of real CLR shape (sim.simulate_raw_reads_shaped: log-normal lengths, bursty errors) adding up to about
    the same number of bases"""
    cfg, ci, L, n_reads, R, win = args
    from falcon_unzip_amd import sim
    rng = sim.rng_for(cfg, ci)
    hap0, hap1, _ = sim.make_diploid(L, rng)
    lo = int(rng.integers(0, L - win + 1))
    lens = sim.lognormal_lengths(int(n_reads * R / 14200), rng)
    codes, off, *_ = sim.simulate_raw_reads_shaped(hap0, hap1, len(lens), rng, lens=lens, lo=lo, hi=lo + win)
    return sim.ACGT[hap0].tobytes(), sim.ACGT[codes].tobytes(), off


def make_inputs(cfg, contig_ids, L, reads_of, R, win, workers, gen=None):
    """contig_ids: global contig indices this rank processes; reads_of(ci) -> number of reads of that contig"""
    jobs = [(cfg, ci, L, reads_of(ci), R, win) for ci in contig_ids]
    if workers > 1 and len(jobs) > 1:
        with mp.get_context("fork").Pool(min(workers, len(jobs))) as pool:     # before any GPU initialisation
            res = pool.map(gen or gen_contig, jobs)
    else:
        res = [(gen or gen_contig)(j) for j in jobs]
    contigs = [r[0] for r in res]
    blob = b"".join(r[1] for r in res)
    offs, read_ctg, base = [np.zeros(1, np.int64)], [], 0
    for c, r in enumerate(res):
        offs.append(r[2][1:] + base)
        base += int(r[2][-1])
        read_ctg.append(np.full(len(r[2]) - 1, c, np.int32))
    return contigs, blob, np.concatenate(offs), np.concatenate(read_ctg) if read_ctg else np.zeros(0, np.int32)


def make_names_and_maps(read_ctg, off, ids, arid_base):
    """read names as <ctg>_reads.fa would carry them and the three read_map files of fc_phasing_readmap.py
    (phasing_readmap.py:15-16,36): raw read i == pread i, pread_ids' second field = raw id * 10."""
    n = len(read_ctg)
    lens = np.diff(off)
    names = [b"sim/%d/0_%d" % (arid_base + i, lens[i]) for i in range(n)]
    noff = np.zeros(n + 1, np.int64)
    noff[1:] = np.cumsum([len(x) for x in names])
    rawread_ids = b"\n".join(names) + b"\n"
    pread_ids = b"".join(b"pread/%d/0_%d\n" % (10 * i, lens[i]) for i in range(n))
    p2c = b"".join(b"%09d %s 15000 0 15000 1\n" % (i, ids[read_ctg[i]].encode()) for i in range(n))
    return (noff, b"".join(names)), (rawread_ids, pread_ids, p2c)


def write_reads_tree(contigs, blob, off, read_ctg, ids, name_tab, root):
    """<root>/fzp_bench_reads_*/: <ctg>_ref.fa and <ctg>_reads.fa as fc_unzip.py's fetch_reads leaves them (unzip.py:204,233-234): one line per sequence"""
    import tempfile
    d = tempfile.mkdtemp(prefix="fzp_bench_reads_", dir=root)
    noff, nblob = name_tab
    mv = memoryview(blob)
    for c, cid in enumerate(ids):
        with open(os.path.join(d, "%s_ref.fa" % cid), "wb") as f:
            f.write(b">" + cid.encode() + b"\n")
            f.write(contigs[c])
            f.write(b"\n")
        with open(os.path.join(d, "%s_reads.fa" % cid), "wb") as f:
            parts = []
            for r in np.flatnonzero(read_ctg == c):
                parts += [b">", nblob[noff[r]:noff[r + 1]], b"\n", mv[off[r]:off[r + 1]], b"\n"]
            f.write(b"".join(parts))
    return d


def tree_digest(root):
    import hashlib
    h = hashlib.sha256()
    for d, _, files in sorted(os.walk(root)):
        for fn in sorted(files):
            with open(os.path.join(d, fn), "rb") as fh:
                h.update(os.path.relpath(os.path.join(d, fn), root).encode())
                h.update(fh.read())
    return h.hexdigest()


def cgroup_throttle():
    """{nr_throttled, throttled_ms} of this process's CPU cgroup (v2 cpu.stat, v1 cpu,cpuacct/cpu.stat), None where neither exists: a rank behind a CPU quota that wants
    more than its share shows up here as periods in which it was stopped, which a wall clock alone cannot tell from a slow kernel"""
    for path in ("/sys/fs/cgroup/cpu.stat", "/sys/fs/cgroup/cpu,cpuacct/cpu.stat", "/sys/fs/cgroup/cpu/cpu.stat"):
        try:
            kv = dict(l.split()[:2] for l in open(path).read().splitlines() if len(l.split()) >= 2)
        except OSError:
            continue
        if "nr_throttled" in kv:
            us = float(kv.get("throttled_usec", 0)) if "throttled_usec" in kv else float(kv.get("throttled_time", 0)) / 1e3
            return {"nr_throttled": int(kv["nr_throttled"]), "throttled_ms": round(us / 1e3, 2)}
    return None


def throttle_delta(a, b):
    return None if (a is None or b is None) else {"nr_throttled": b["nr_throttled"] - a["nr_throttled"], "throttled_ms": round(b["throttled_ms"] - a["throttled_ms"], 2)}


def thread_cpu():
    """{tid: (name, user + system seconds, system seconds)} of this process's threads (/proc/self/task)"""
    out = {}
    tck = os.sysconf("SC_CLK_TCK")
    for t in os.listdir("/proc/self/task"):
        try:
            with open("/proc/self/task/%s/stat" % t) as f:
                st = f.read()
            nm = st[st.index("(") + 1:st.rindex(")")]
            fld = st[st.rindex(")") + 2:].split()
            out[int(t)] = (nm, (int(fld[11]) + int(fld[12])) / tck, int(fld[12]) / tck)
        except (OSError, ValueError):
            pass
    return out


def host_cores():
    """threads this process may really use: the affinity mask and the cgroup CPU quota, not the machine's thread count (the GPU boxes show 256
    hardware threads behind a 16-CPU quota)"""
    n = os.cpu_count() or 1
    try:
        n = min(n, len(os.sched_getaffinity(0)))
    except (AttributeError, OSError):
        pass
    try:
        with open("/sys/fs/cgroup/cpu.max") as f:
            q, p = f.read().split()[:2]
            if q != "max":
                n = min(n, max(1, int(int(q) / int(p) + 0.5)))
    except (OSError, ValueError):
        pass
    return max(1, n)


def shm_with_room(need_bytes):
    """/dev/shm when it is there and has `need_bytes` free (a container's default one has 64 MB), else None"""
    try:
        st = os.statvfs("/dev/shm")
        return "/dev/shm" if os.path.isdir("/dev/shm") and st.f_bavail * st.f_frsize >= need_bytes else None
    except OSError:
        return None


def fs_of(path):
    """file system type of the mount that holds `path` (/proc/mounts)"""
    best, typ = "", ""
    try:
        rp = os.path.realpath(path)
        with open("/proc/mounts") as f:
            for l in f:
                t = l.split()
                if len(t) >= 3 and (rp == t[1] or rp.startswith(t[1].rstrip("/") + "/")) and len(t[1]) >= len(best):
                    best, typ = t[1], t[2]
    except OSError:
        pass
    return typ


def cpu_model():
    try:
        with open("/proc/cpuinfo") as f:
            for l in f:
                if l.startswith("model name"):
                    return l.split(":", 1)[1].strip()
    except OSError:
        pass
    return ""


def cpu_baseline(contigs, blob, off, read_ctg, ids, job, hip_summ, budget_s=30.0):
    """The oracle ("port": scalar C restatement) on the host's cores, the reference's own shape: contigs side by side (unzip.py:255), each
    contig's reads on its share of the threads (blasr --nproc, unzip.py:88).  Sample = as many whole contigs of the workload as ~budget_s
    of CPU time allow at 80 Mcell/s per thread (all of them on a many-core GPU host).  Per contig the twin aligner
    (oracle/align_oracle.c) reports its index build and its seeding + DP separately; then the oracle/phasing_oracle.c chain runs on the
    SAM text of those contigs, one contig per thread (phasing.py:496-498 max_jobs=1).  `job` (the HIP job of the timed run) supplies the
    SAM text the chain reads and, the checker's other use, the parity count."""
    from concurrent.futures import ThreadPoolExecutor
    from falcon_unzip_amd import _lib
    from tests import oracle_lib
    orc = oracle_lib.load()
    cores = host_cores()
    n_ctg = len(contigs)
    cells_per_ctg = float(hip_summ["cells"].sum()) / max(1, n_ctg)
    n_sample = max(1, min(n_ctg, int(budget_s * 80e6 * cores / max(1.0, cells_per_ctg))))
    side = min(n_sample, cores)                                 # contigs in flight
    thr = max(1, -(-cores // side))                             # threads per contig (rounded up: 9 contigs on 16 cores take 2 each)
    idxs = [np.flatnonzero(read_ctg == c) for c in range(n_sample)]

    def one(c):
        reads = [blob[off[i]:off[i + 1]] for i in idxs[c]]
        sec = []
        summ, cigs = oracle_lib.align_reads(orc, contigs[c], reads, n_threads=thr, seconds=sec)
        return summ, sec, cigs

    t0 = time.perf_counter()
    with ThreadPoolExecutor(max_workers=side) as ex:
        res = list(ex.map(one, range(n_sample)))
    t_aln = time.perf_counter() - t0
    cells = float(sum(r[0]["cells"].sum() for r in res))
    n_used = int(sum(len(x) for x in idxs))
    t_index = max(r[1][0] for r in res)                         # the contigs' indexes are built side by side: the slowest one is on the clock
    t_dp = max(1e-9, t_aln - t_index)
    mismatched = 0
    fields = ("aligned", "strand", "pos", "ref_end", "q_start", "q_end", "score", "n_cigar", "cells", "n_columns", "n_match")
    hip_hash = job.cigar_hashes()                               # a 64-bit fingerprint of every read's CIGAR words: gap placement, not only the summaries
    cig_diff = 0
    for c in range(n_sample):
        mismatched += int(sum(int((hip_summ[f][idxs[c]] != res[c][0][f]).sum()) for f in fields))
        cig_diff += int((hip_hash[idxs[c]] != _lib.cigar_hash_of_words(res[c][2])).sum())      # (the checker's work: outside the baseline's clock)
    sams = []
    for c in range(n_sample):
        aln, _ = job.alnset(c)
        sams.append((_lib.format_sam(aln, ids[c]), contigs[c], ids[c]))
    t0 = time.perf_counter()
    with ThreadPoolExecutor(max_workers=side) as ex:
        list(ex.map(lambda a: orc.phase_all(*a), sams))
    t_ph = time.perf_counter() - t0
    return {"value": round(n_used / (t_aln + t_ph), 3), "unit": "reads/s", "cores": cores, "hardware_threads": os.cpu_count(), "kind": "port", "cpu_model": cpu_model(),
            "sample": "all %d reads of %d of the %d contigs: %d contigs side by side x %d threads each -- oracle/align_oracle.c (index build, then seeding + DP + "
                      "trace-back), then the oracle/phasing_oracle.c chain, one contig per thread; the reference's blasr and Python 2 cannot run here"
                      % (n_used, n_sample, n_ctg, side, thr),
            "contigs": n_sample, "threads_per_contig": thr,
            "align_s": round(t_aln, 3), "index_s": round(t_index, 3), "dp_s": round(t_dp, 3), "phasing_s": round(t_ph, 3),
            "dp_gcell_per_s": round(cells / t_dp / 1e9, 4), "dp_mcell_per_s_per_thread": round(cells / t_dp / 1e6 / min(cores, side * thr), 2),
            "k1_fields_differing_from_hip": mismatched, "k1_cigars_differing_from_hip": cig_diff}


def roofline(cells_per_launch, sw_avg_ms, sw_launches, dp_gcells, traffic, band=32):
    """k1_sw, the dominant stage: the banded DP of every extension piece of the step in ONE launch pair (k_swb, bit-sliced, one piece per lane; k_sw, one wave per
    piece, for the pieces narrower than the band -- DESIGN section 5).  What binds it is VALU issue, so that is the headline: wave64 VALU instructions per second
    (SQ_INSTS_VALU from profiles/, named in `counter_source`) against the chip's issue peak, 256 CU x 4 SIMD x 2.4 GHz / 2 cycles per wave64 instruction on a SIMD-32
    (MI355X_MICROARCH.md).  Beside it the two HBM views: SURVEY 8d's algorithmic 0.25 B per cell (the north star's figure), and the bytes the kernel really moves since
    round 4 -- band lanes 16..47 only, 0.125 B/cell; `traffic` = the measured HBM bytes of the largest DP dispatch (FETCH + WRITE, from profiles/)."""
    counters = {"valu_per_step": SW_VALU_PER_STEP, "source": "(default: no profiles/k1_sw_counters.json)"}
    cf = os.path.join(REPO, "profiles", "k1_sw_counters.json")
    if os.path.exists(cf):
        with open(cf) as f:
            counters = json.load(f)
    per_step = float(counters["valu_per_step"])
    # the committed counter files belong to a particular kernel and workload: say so when this run no longer looks like the one they were taken on
    stale = []
    moved_per_cell = 8.0 / band      # 8 B of mask record per band step: the band's middle 32 lanes of v1.7's 64-cell band, the WHOLE mask of v1.8's 32-cell band
    if counters.get("band", 64) != band:
        stale.append("counters: taken with a band of %d cells, this run's has %d" % (counters.get("band", 64), band))
    if counters.get("band_steps_per_bench_step") and abs(counters["band_steps_per_bench_step"] * float(band) / max(cells_per_launch, 1.0) - 1.0) > 0.02:
        stale.append("counters: %.4g band steps per launch when they were taken, %.4g now" % (counters["band_steps_per_bench_step"], cells_per_launch / float(band)))
    if traffic and abs(traffic / max(cells_per_launch * moved_per_cell, 1.0) - 1.0) > 0.25:
        stale.append("traffic: %.3g B per launch on file, %.3g B of mask records planned by this run" % (traffic, cells_per_launch * moved_per_cell))
    steps_per_s = dp_gcells * 1e9 / float(band)
    valu = steps_per_s * per_step / 1e9
    valu_peak = N_SIMD * CLK_GHZ / 2.0
    secs = sw_avg_ms * 1e-3 if sw_avg_ms else 0.0
    gbs_alg = cells_per_launch * SW_BYTES_PER_CELL / secs / 1e9 if secs else 0.0
    gbs_moved = (traffic / secs / 1e9) if (traffic and secs) else None
    return {"bound": "valu", "kernel": "k1_sw (k_swb + k_sw)", "achieved": round(valu, 2), "peak": round(valu_peak, 1), "unit": "G wave64-inst/s", "frac": round(valu / valu_peak, 4),
            "traffic": traffic, "profile_files_stale": stale or None, "avg_launch_ms": round(sw_avg_ms, 3), "launches": int(sw_launches),
            "valu_insts_per_band_step": per_step, "counter_source": counters.get("source"), "band_cells": band,
            "note": "the DP stage keeps every SIMD busy with one wave (~3 500 waves of 64 pieces on 1 024 SIMDs) and is bound by VALU issue: wave64 VALU instructions per second "
                    "(SQ_INSTS_VALU of the committed counter pass x this run's band steps per second) against 256 CU x 4 SIMD x 2.4 GHz / 2 cycles per instruction.  The two HBM "
                    "views sit beside it: `hbm_algorithmic` prices SURVEY 8d's 0.25 B per cell (the 2 trace-back bits of every cell), `hbm_moved` what the kernel really moves "
                    "(`traffic`: 8 B of mask record per band step -- with fzalign v1.8's 32-cell band the whole mask -- + the base streams).  (SURVEY 8d's third view, 12 integer "
                    "ops per cell, is not priced any more: a bit-parallel kernel spends ~2 lane-operations per cell, the 'fraction' came out at 2.0.)",
            # the kernel holds ONE wave per SIMD (265 registers a lane); a lone wave issues a VALU instruction every 4 cycles at best (the guide's figure; measured 4.8-6.6 for
            # this mix, tools/ubench/bitops_issue) -- the ceiling of that shape, beside the chip's
            "one_wave_per_simd": {"peak": round(valu_peak / 2.0, 1), "frac": round(valu / (valu_peak / 2.0), 4), "unit": "G wave64-inst/s",
                                  "note": "1 024 SIMDs x 2.4 GHz / 4 cycles per instruction of a lone wave; a second wave per SIMD was built and measured slower (profiles/r6_swb_units.txt, r6_not_kept.txt)"},
            "hbm_algorithmic": {"achieved": round(gbs_alg, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(gbs_alg / HBM_PEAK_GBS, 4), "bytes_per_cell": SW_BYTES_PER_CELL},
            "hbm_moved": {"achieved": round(gbs_moved, 2) if gbs_moved else None, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(gbs_moved / HBM_PEAK_GBS, 4) if gbs_moved else None,
                          "bytes_per_cell": moved_per_cell, "bytes_per_launch": traffic}}


def polish_leg(eng, contigs, blob, off, read_ctg):
    """fzp_polish_tigs on the step's own inputs: every contig a tig, its reads the tig's pile -- K1 aligns them, K6's packed tally calls the whole tig (the consensus role of
    run_quiver.py:82-97 per tig).  Reported beside `value`: tigs per second, template bases per second."""
    from falcon_unzip_amd import _lib
    walls, n_rec, n_out = [], 0, 0
    for _ in range(3):
        t0 = time.perf_counter()
        t = _lib.polish_tigs(eng, contigs, blob, off, read_ctg)
        walls.append(time.perf_counter() - t0)
        n_rec, n_out = int(t.tigs["n_records"].sum()), int(t.tigs["seq_len"].sum())
        t.close()
    w = min(walls[1:])
    return {"tigs": len(contigs), "template_mb": round(sum(len(c) for c in contigs) / 1e6, 1), "reads": int(len(read_ctg)), "records_in_piles": n_rec, "bases_out": n_out,
            "ms_per_call": round(w * 1e3, 2), "first_call_ms": round(walls[0] * 1e3, 2), "tigs_per_s": round(len(contigs) / w, 1), "template_mb_per_s": round(sum(len(c) for c in contigs) / 1e6 / w, 1),
            "reads_per_s": round(len(read_ctg) / w, 1),
            "what": "fzp_polish_tigs from host buffers (upload, pack, index, K1, packed hand-off, K6 over every whole tig, download): every contig of the step as a tig with its own reads"}


def shaped_leg(eng, inp):
    """K1 alone on the same contigs with reads of real CLR shape (fzalign v1.6 cuts every read into pieces of ~3 kb, so uneven read lengths no longer
    shape the DP launch: one figure, where r3 compared launch orders)."""
    from falcon_unzip_amd import _lib
    contigs, blob, off, read_ctg = inp
    job = _lib.align_job_raw(eng, contigs, blob, off, read_ctg)
    job.run()
    eng.synchronize()
    eng.prof_reset()
    eng.prof_enable(True)
    for _ in range(2):
        job.run()
    eng.synchronize()
    eng.prof_enable(False)
    pr = eng.prof()
    summ = job.summaries()
    sw_ms, sw_n = pr.get("k1_sw", (0.0, 0))
    res = {"k1_sw_ms": round(sw_ms / max(1, sw_n), 3), "k1_traceback_ms": round(pr.get("k1_traceback", (0.0, 0))[0] / max(1, sw_n), 3),
           "k1_seed_ms": round(pr.get("k1_seed", (0.0, 0))[0] / max(1, sw_n), 3),
           "dp_gcell_per_s": round(float(summ["cells"].sum()) / (sw_ms / max(1, sw_n) * 1e-3) / 1e9, 2) if sw_ms else 0.0}
    lens = np.diff(off)
    inside = (summ["q_end"] - summ["q_start"])[summ["aligned"] == 1]
    job.close()
    res.update({"reads": int(len(lens)), "read_len_min_median_max": [int(lens.min()), int(np.median(lens)), int(lens.max())], "aligned_frac": round(float(summ["aligned"].mean()), 4),
                "bases_inside_alignments_frac": round(float(inside.sum()) / float(lens.sum()), 4),
                "workload": "the step's contigs with log-normal read lengths (median 12 kb, 3-60 kb) and bursty errors (0.3-1 kb at 30 %, a quarter of the reads at their head), ~ the same bases; K1 only"})
    return res


def strong_inputs(args, rank, world, workers):
    """the strong_cfg3 leg's shard of this rank, generated BEFORE anything initialises the GPU (the generator forks workers)"""
    from falcon_unzip_amd import dist as fdist
    nc, L, R = args.strong_leg_contigs, args.strong_leg_contig_len, args.read_len
    u = np.random.Generator(np.random.PCG64(20263000)).random(nc)
    n_reads_c = (args.reads_per_contig * (0.5 + 1.5 * u)).astype(np.int64)
    mine = fdist.shard_contigs((n_reads_c * R).tolist(), world)[rank]
    try:
        contigs, blob, off, read_ctg = make_inputs(3, mine, L, lambda ci: int(n_reads_c[ci]), R, L, workers)
    except Exception as e:      # noqa: BLE001 -- e.g. MemoryError on a small host: the leg reports it, the main line is unaffected
        return {"error": "rank %d: input generation: %r" % (rank, e), "mine": mine, "n_reads_c": n_reads_c}
    return {"mine": mine, "n_reads_c": n_reads_c, "contigs": contigs, "blob": blob, "off": off, "read_ctg": read_ctg}


def strong_leg(args, rank, world, eng, comm, coll_dev, inp, out_root):
    """BASELINE configs[2] inside an N-rank run: a FIXED job of --strong-leg-contigs contigs (750 kb) with 0.5x..2x
    --reads-per-contig reads each, dealt LPT by read bases (dist.shard_contigs), every rank streaming its shard from HOST buffers through
    fzp_phase_contigs (contig groups on lanes; PCIe, packing and index inside), then the one all-gather.  Two passes, the second is timed;
    the figure is all reads / slowest rank.  A rank that fails says so and the leg is reported as failed by every rank -- no collective is
    entered unless all ranks got there."""
    import torch
    import torch.distributed as dist
    from falcon_unzip_amd import _lib
    from falcon_unzip_amd import dist as fdist
    nc, L, R = args.strong_leg_contigs, args.strong_leg_contig_len, args.read_len
    mine, n_reads_c = inp["mine"], inp["n_reads_c"]
    err, dt, n_mine, recs, st = inp.get("error", ""), 0.0, 0, np.zeros(0, _lib.R2P), None
    try:
        if err:
            raise RuntimeError(err)
        contigs, blob, off, read_ctg = inp["contigs"], inp["blob"], inp["off"], inp["read_ctg"]
        n_mine = len(read_ctg)
        ids = ["%06dF" % ci for ci in mine]
        name_tab, maps = make_names_and_maps(read_ctg, off, ids, 0)
        for k in range(2):
            eng.synchronize()
            t0 = time.perf_counter()
            if mine:
                st, recs = _lib.phase_contigs(eng, contigs, blob, off, read_ctg, ids, names=name_tab, out_dir=os.path.join(out_root, "strong_%d" % k), read_maps=maps,
                                              ctg_index=mine, n_lanes=args.e2e_lanes, consensus=args.with_consensus, async_writes=True)
            recs["arid"] += 10_000_000 * rank
            eng.synchronize()
            eng.pipe_flush()
            dt = time.perf_counter() - t0
    except Exception as e:      # noqa: BLE001 -- reported in the line
        err = err or "rank %d: %r" % (rank, e)
        print("bench.py strong_cfg3: " + err, file=sys.stderr, flush=True)
    ok = torch.tensor([0 if err else 1], dtype=torch.int32, device=coll_dev)
    dist.all_reduce(ok, op=dist.ReduceOp.MIN)
    if int(ok.item()) == 0:
        return {"error": err or "another rank failed"}
    t0 = time.perf_counter()
    allr = comm.allgather_r2p(recs) if comm is not None else fdist.allgather_r2p(recs, device=coll_dev)
    t_gather = time.perf_counter() - t0
    per_rank = [torch.zeros(3, dtype=torch.float64, device=coll_dev) for _ in range(world)]
    dist.all_gather(per_rank, torch.tensor([dt + t_gather, float(n_mine), float(len(mine))], dtype=torch.float64, device=coll_dev))
    slowest = max(float(p[0].item()) for p in per_rank)
    total = int(sum(float(p[1].item()) for p in per_rank))
    return {"workload": "cfg3: %d contigs x %d bp IN TOTAL, %d..%d reads x %d bp each, LPT over %d ranks; host ASCII -> fzp_phase_contigs (upload, pack, k-mer tables, K1..K5, "
                        "files written, readmap) -> one all-gather" % (nc, L, int(n_reads_c.min()), int(n_reads_c.max()), R, world),
            "scaling": "strong", "n_gpus": world, "reads_total": total, "reads_per_s": round(total / slowest, 1), "s": round(slowest, 4),
            "r2p_records": int(len(allr)), "gather_ms": round(t_gather * 1e3, 3),
            "groups_rank0": int(st["n_groups"]) if st else 0,
            "rank_load": [{"rank": r, "contigs": int(p[2].item()), "reads": int(p[1].item()), "s": round(float(p[0].item()), 4)} for r, p in enumerate(per_rank)]}


def constrained_child(args, n_cores, contigs, blob, off, read_ctg, ids, name_tab, maps, mine):
    """The resident step once more with the CPU a rank of a full node has: a CHILD forked here -- before this process imports torch or touches the GPU; a fork, never an
    exec -- confines itself to `n_cores` CPUs (sched_setaffinity), says LOCAL_WORLD_SIZE=8 (the library sizes its host thread pools by the rank's share of the cores) and
    runs warm-up + timed steps of the same fzp_job_phase_write on the same inputs (inherited copy-on-write), alone on the GPU; the parent waits for it and only then brings
    up its own context.  8 ranks behind the GPU boxes' 16-CPU quota have two cores each: `two_core_step_ms` against `ms_per_step` is what the 8-GPU curve will feel first."""
    rd, wr = os.pipe()
    pid = os.fork()
    if pid == 0:
        code = 1
        try:
            os.close(rd)
            os.sched_setaffinity(0, sorted(os.sched_getaffinity(0))[:n_cores])
            os.environ["LOCAL_WORLD_SIZE"] = "8"
            from falcon_unzip_amd import _lib
            eng = _lib.Engine(0)
            job = _lib.align_job_raw(eng, contigs, blob, off, read_ctg)
            root = tempfile.mkdtemp(prefix="fzp_bench_2c_", dir=(shm_with_room(4 << 30) if not args.out_root else args.out_root))
            n_steps = int(os.environ.get("FZP_BENCH_2C_STEPS", "20"))      # (twenty: ten steps of a fresh process varied by +-2 ms run to run)

            def one(k, fresh=args.fresh_trees):
                job.phase_write(ids, names=name_tab, out_dir=os.path.join(root, ("s%03d" % k) if fresh else ("t%d" % (k & 1))), read_maps=maps, ctg_index=mine,
                                consensus=args.with_consensus, async_writes=True, rebuild_index=not args.index_at_create)
            for k in range(3):
                one(k)
            eng.synchronize(); eng.pipe_flush()
            th0 = thread_cpu()
            cg0 = cgroup_throttle()
            t0, c0 = time.perf_counter(), time.process_time()
            for k in range(n_steps):
                one(3 + k)
            eng.synchronize(); eng.pipe_flush()
            dt, cpu = time.perf_counter() - t0, time.process_time() - c0
            cg_step = throttle_delta(cg0, cgroup_throttle())
            th1 = thread_cpu()
            by_name = {}      # who burns the rank's two cores: CPU ms per step by thread name (the runtime's own threads carry the process's name)
            for tid, (nm, tot, sy) in th1.items():
                d = tot - th0.get(tid, (nm, 0.0, 0.0))[1]
                key = nm.rstrip("0123456789") if nm.startswith("fzp-") else ("main" if tid == os.getpid() else "runtime/" + nm)
                by_name[key] = by_name.get(key, 0.0) + d
            by_name = {k: round(v / n_steps * 1e3, 2) for k, v in sorted(by_name.items(), key=lambda kv: -kv[1]) if v / n_steps * 1e3 >= 0.05}
            # the same child the other way (a fresh tree per step: every page of every file newly allocated; two trees rewritten in turn: the files overwritten in place)
            for k in range(3):
                one(100 + k, fresh=not args.fresh_trees)
            eng.synchronize(); eng.pipe_flush()
            t1, c1 = time.perf_counter(), time.process_time()
            for k in range(10):
                one(103 + k, fresh=not args.fresh_trees)
            eng.synchronize(); eng.pipe_flush()
            other = {"trees": "fresh" if not args.fresh_trees else "rewritten", "ms_per_step": round((time.perf_counter() - t1) / 10 * 1e3, 3),
                     "host_cpu_ms_per_step": round((time.process_time() - c1) / 10 * 1e3, 2), "steps": 10}
            job.close()
            ff = None
            if not args.no_from_files:      # the same two cores reading the step's FASTA files: what the reader, the copies and the lanes' launch threads need side by side
                try:
                    gcf, lanesf = args.e2e_group_contigs, args.e2e_lanes
                    t_b, st_b, _, _, mb = files_leg(args, eng, contigs, blob, off, read_ctg, ids, name_tab, maps, mine, root, gcf, lanesf)
                    ff = {"ms": round(t_b * 1e3, 2), "host_cpu_ms": st_b.get("host_cpu_ms"), "cgroup": st_b.get("cgroup"), "lanes": lanesf, "groups": int(st_b["n_groups"])}
                except Exception as e:      # noqa: BLE001 -- reported in the line
                    ff = {"error": repr(e)}
            eng.close()
            shutil.rmtree(root, ignore_errors=True)
            os.write(wr, json.dumps({"ms_per_step": round(dt / n_steps * 1e3, 3), "host_cpu_ms_per_step": round(cpu / n_steps * 1e3, 2), "steps": n_steps, "cpus": n_cores,
                                     "local_world_size": 8, "cpu_ms_per_step_by_thread": by_name, "cgroup": cg_step, "from_files": ff, "other_tree_mode": other}).encode())
            code = 0
        except BaseException as e:      # noqa: BLE001 -- reported by the parent
            try:
                os.write(wr, json.dumps({"error": repr(e)}).encode())
            except OSError:
                pass
        finally:
            os._exit(code)
    os.close(wr)
    data = b""
    while True:
        chunk = os.read(rd, 65536)
        if not chunk:
            break
        data += chunk
    os.close(rd)
    os.waitpid(pid, 0)
    try:
        return json.loads(data.decode())
    except ValueError:
        return {"error": "the child left no result"}


def files_leg(args, eng, contigs, blob, off, read_ctg, ids, name_tab, maps, mine, out_root, gc, lanes, sync=None):
    """The workload from the reference's own input files (unzip.py:204,233-234: reads/<ctg>_ref.fa, <ctg>_reads.fa) on a memory file system: FASTA parsing (the library's,
    a contig group ahead of the lanes) inside the clock -- what scripts/fc_unzip_phase_gpu.py does per rank.  Three calls, the best of the last two; `sync` (N > 1): called
    before every call so that the ranks run theirs side by side on the node's shared host cores.  -> (seconds, stats, records, output dir of the last call)"""
    from falcon_unzip_amd import _lib
    reads_dir = write_reads_tree(contigs, blob, off, read_ctg, ids, name_tab, shm_with_room((2 << 30) * max(1, int(os.environ.get("LOCAL_WORLD_SIZE", "1")))) or out_root)
    try:
        gb = int(gc * args.reads_per_contig * args.read_len * 1.09)      # (file sizes: bases + names)
        runs = []
        for k in range(3):
            if sync:
                sync()
            t1, c1, th1 = time.perf_counter(), time.process_time(), cgroup_throttle()
            st, recs_f = _lib.phase_contigs_files(eng, reads_dir, ids, out_dir=os.path.join(out_root, "files_%d" % k), read_maps=maps, ctg_index=mine, n_lanes=lanes, group_bases=gb,
                                                  consensus=args.with_consensus, async_writes=True)
            st = dict(st)
            st["host_cpu_ms"] = round((time.process_time() - c1) * 1e3, 2)      # user + system time of every thread of the rank during the call (readers, launch threads, writers, the runtime's)
            st["cgroup"] = throttle_delta(th1, cgroup_throttle())
            runs.append((time.perf_counter() - t1, st))
        best = min(runs[1:], key=lambda x: x[0])
        mb = sum(os.path.getsize(os.path.join(reads_dir, f)) for f in os.listdir(reads_dir)) / 1e6
        return best[0], best[1], recs_f, os.path.join(out_root, "files_2"), mb
    finally:
        shutil.rmtree(reads_dir, ignore_errors=True)


def launch_ranks(n):
    """--gpus N outside a launcher: N fresh ranks as children of this process, which has not imported torch nor touched the GPU."""
    import socket
    import subprocess
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.setdefault("OMP_NUM_THREADS", "1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n), "--master-addr", "127.0.0.1", "--master-port", str(port),
           os.path.abspath(__file__)] + sys.argv[1:]
    return subprocess.call(cmd, env=env)        # stdout is inherited: rank 0's JSON line goes straight through


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--contigs", type=int, default=20, help="contigs per GPU (weak scaling) or in total (--strong)")
    ap.add_argument("--contig-len", type=int, default=5_000_000)
    ap.add_argument("--reads-per-contig", type=int, default=2000)
    ap.add_argument("--read-len", type=int, default=15000)
    ap.add_argument("--window", type=int, default=750_000)
    ap.add_argument("--strong", action="store_true", help="configs[2] shape: --contigs contigs IN TOTAL with 0.5x..2x the reads each, dealt LPT over the ranks")
    ap.add_argument("--cpu-budget-s", type=float, default=30.0, help="CPU seconds the cpu_baseline sample is sized for (whole contigs; all of them on a many-core host)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--index-at-create", action="store_true", help="leave the k-mer tables fzp_align_create built in place (r2's step); default: the step rebuilds them, as a job that sees its contigs once would")
    ap.add_argument("--no-shaped-leg", action="store_true", help="skip the K1-only leg on reads of real CLR shape (log-normal lengths, bursty errors)")
    ap.add_argument("--strong-leg-contigs", type=int, default=500, help="N > 1 runs: contigs of the strong_cfg3 leg (0 = no such leg)")
    ap.add_argument("--strong-leg-contig-len", type=int, default=750_000)
    ap.add_argument("--no-end-to-end", action="store_true")
    ap.add_argument("--no-from-files", action="store_true", help="skip the from_files leg (the end-to-end workload from FASTA files on a memory file system)")
    ap.add_argument("--no-kernel-breakdown", action="store_true", help="skip the instrumented pass behind the timed steps (counter-collection runs: the pass would add its launches to the sums)")
    ap.add_argument("--no-two-core", action="store_true", help="skip two_core_step_ms (the resident step in a child confined to two CPUs with LOCAL_WORLD_SIZE=8)")
    ap.add_argument("--e2e-lanes", type=int, default=2)
    ap.add_argument("--e2e-group-contigs", type=int, default=10)
    ap.add_argument("--rewrite-trees", dest="fresh_trees", action="store_false", help="two output trees written in turn -- from the third step on every file is overwritten in place, "
                    "a re-run into an existing tree (default: every step writes into an empty directory of its own, a first run; `other_tree_mode` in the line times a few steps the other way)")
    ap.add_argument("--no-tree-compare", action="store_true", help="skip `other_tree_mode` (eleven more steps behind the timed ones; counter-collection runs want exactly the timed steps' launches)")
    ap.add_argument("--with-polish", action="store_true", help="also time fzp_polish_tigs on the step's inputs (every contig a tig, its reads the pile): tigs/s beside `value`")
    ap.add_argument("--with-consensus", action="store_true", help="also run K6 (phased-pile consensus, BASELINE config 4) inside every step")
    ap.add_argument("--out-root", default=None, help="where the per-step output trees go (default: a scratch directory under $TMPDIR)")
    ap.add_argument("--gen-workers", type=int, default=0, help="processes for input generation (0 = auto; forced to 1 under rocprofv3)")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(launch_ranks(args.gpus))          # nothing has touched the GPU yet; the ranks are children, this process only waits
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit("bench.py: --gpus %d but the launcher started %d rank(s) (WORLD_SIZE); they must agree" % (args.gpus, world))
    workers = args.gen_workers or max(1, min(16 if world > 1 else 8, host_cores() // max(1, world)))
    if "rocprof" in os.environ.get("LD_PRELOAD", "") or os.environ.get("ROCPROFILER_REGISTER_FORCE_LOAD") or os.environ.get("ROCP_TOOL_LIBRARIES"):
        workers = 1   # the profiler's preloaded library may have initialised the GPU already: do not fork
    from falcon_unzip_amd import dist as fdist
    win = min(args.window, args.contig_len)
    if args.strong:
        # a fixed job: contig c carries (0.5 + 1.5 u_c) x reads-per-contig reads; LPT by read bases (dist.shard_contigs)
        u = np.random.Generator(np.random.PCG64(20263000)).random(args.contigs)
        n_reads_c = (args.reads_per_contig * (0.5 + 1.5 * u)).astype(np.int64)
        shards = fdist.shard_contigs((n_reads_c * args.read_len).tolist(), world)
        mine = shards[rank]
        cfg = 3
        reads_of = lambda ci: int(n_reads_c[ci])
    else:
        mine = list(range(rank * args.contigs, (rank + 1) * args.contigs))
        cfg = 2
        reads_of = lambda ci: args.reads_per_contig
    contigs, blob, off, read_ctg = make_inputs(cfg, mine, args.contig_len, reads_of, args.read_len, win, workers)
    n_reads = len(read_ctg)
    ids = ["%06dF" % ci for ci in mine]
    arid_base = int(sum(reads_of(ci) for ci in range(mine[0]))) if (mine and not args.strong) else (1_000_000 * rank)
    name_tab, maps = make_names_and_maps(read_ctg, off, ids, arid_base)
    shaped_inp = None
    if rank == 0 and world == 1 and not args.strong and not args.no_shaped_leg:
        shaped_inp = make_inputs(cfg, mine, args.contig_len, reads_of, args.read_len, win, workers, gen=gen_contig_shaped)      # before the GPU is touched (forks)
    s_inp = strong_inputs(args, rank, world, workers) if (world > 1 and not args.strong and args.strong_leg_contigs > 0) else None
    two_core = None
    if rank == 0 and world == 1 and not args.strong and not args.no_two_core and workers > 1 and mine:      # (workers == 1: under a profiler -- no forks)
        two_core = constrained_child(args, 2, contigs, blob, off, read_ctg, ids, name_tab, maps, mine)      # before torch is imported or the GPU touched; the child has the GPU to itself

    os.environ.setdefault("ROC_SIGNAL_POOL_SIZE", "4096")      # (before ANY HIP runtime of this process comes up -- under backend nccl torch's does first: include/fzphase.h, fzp_sched_status)
    import torch
    import torch.distributed as dist
    from falcon_unzip_amd import _lib
    backend = os.environ.get("FZP_BENCH_BACKEND", "nccl")     # "gloo": ranks may share a GPU (single-GPU dry run of the N>1 flow)
    n_dev = torch.cuda.device_count()
    dev_index = local_rank if backend == "nccl" else local_rank % max(1, n_dev)
    coll_dev = ("cuda:%d" % dev_index) if backend == "nccl" else "cpu"
    if world > 1:
        if backend == "nccl":
            torch.cuda.set_device(dev_index)
            dist.init_process_group(backend="nccl", device_id=torch.device("cuda", dev_index))
        else:
            dist.init_process_group(backend=backend)
    eng = _lib.Engine(dev_index)
    # the exchange step: the library's own RCCL all-gather (fzp_allgather_rid_to_phase) when the ranks sit on their own GPUs;
    # torch.distributed's all_gather otherwise (gloo dry runs) or if RCCL cannot be brought up identically on every rank
    comm, gather_note = None, None
    if world > 1 and backend == "nccl" and os.environ.get("FZP_BENCH_GATHER", "cabi") == "cabi":
        from falcon_unzip_amd import dist as fdist
        comm, gather_note = fdist.make_comm(eng, rank, world, coll_dev)      # the agree-before-ncclCommInitRank handshake (dist.make_comm)
        if comm is None:
            print("bench.py: " + gather_note, file=sys.stderr, flush=True)
    try:
        rccl_lib = _lib.comm_library()
    except Exception as e:      # noqa: BLE001 -- reported in the line
        rccl_lib = (None, repr(e))
    out_root = None
    # --out-root, else a memory file system (what the step's files cost on a disk-backed /tmp depends on what was written there before -- measured: the same step's
    # writer threads took 2 ms or 30 ms of system time each, in the order the runs came), else $TMPDIR, else wherever a directory can be made
    for cand in ((args.out_root,) if args.out_root else ()) + ("/dev/shm", None, REPO):
        try:
            # (room for every step's tree, the from-files leg's inputs and trees of this rank AND of the ranks beside it: a memory file system of a few GB -- a container's
            #  default /dev/shm -- is passed over rather than filled up in the middle of a run)
            st_fs = os.statvfs(cand or tempfile.gettempdir())
            if cand != args.out_root and st_fs.f_bavail * st_fs.f_frsize < (6 << 30) * max(1, int(os.environ.get("LOCAL_WORLD_SIZE", world))):
                continue
            out_root = tempfile.mkdtemp(prefix="fzp_bench_r%d_" % rank, dir=cand)
            break
        except OSError:
            continue
    if out_root is None:
        raise SystemExit("bench.py: no writable scratch directory for the output trees")
    prof_on = not os.environ.get("FZP_BENCH_NO_PROF")      # (measurement aid: the step without the library's HIP-event brackets; the line then has no kernel times)
    eng.prof_enable(prof_on)
    t_up = time.perf_counter()
    # (a rank whose shard is empty -- --strong with fewer contigs than ranks -- has no job; it still takes part in every collective)
    job = _lib.align_job_raw(eng, contigs, blob, off, read_ctg) if mine else None     # upload + 2-bit pack: inputs now resident in HBM
    eng.synchronize()
    upload_ms = (time.perf_counter() - t_up) * 1e3                  # PCIe + packing + the first k-mer tables, outside the timed region
    index_ms_at_create = eng.prof().get("k1_index", (0.0, 0))[0] if args.index_at_create else 0.0

    def barrier():
        eng.synchronize()
        eng.pipe_flush()                               # every file of every step so far is on the file system
        if world > 1:
            dist.barrier()
            if backend == "nccl":
                torch.cuda.synchronize(dev_index)

    stats = {}
    host_t = {"phase_write": 0.0, "allgather": 0.0}
    sect = {"ms_k1": 0.0, "ms_phase": 0.0, "ms_results": 0.0, "ms_text": 0.0}

    step_no = [0]

    def step():
        step_no[0] += 1
        # the output tree: an empty directory per step, as a job's first run would write it.  --rewrite-trees: two trees written in turn, so that from the third step on every
        # file is OVERWRITTEN IN PLACE -- a re-run into an existing tree, which is what a restarted unzip job does (the library cuts a file to its new length instead of
        # truncating it first: no page is freed and allocated again).  `other_tree_mode` in the line times a few steps the other way.
        out_dir = os.path.join(out_root, ("step%03d" % step_no[0]) if args.fresh_trees else ("step_%d" % (step_no[0] & 1)))
        t_a = time.perf_counter()
        if job is not None:
            st, recs = job.phase_write(ids, names=name_tab, out_dir=out_dir, read_maps=maps, ctg_index=mine, consensus=args.with_consensus, async_writes=True,
                                         rebuild_index=not args.index_at_create)
        else:
            st, recs = {k: 0 for k in ("n_aligned", "n_rec", "n_sites", "n_rows", "n_arows", "n_pvars", "n_preads", "bytes_written", "ms_k1", "ms_phase", "ms_results", "ms_text")}, np.zeros(0, _lib.R2P)
        recs["arid"] += arid_base                      # the read_map files of a rank number its preads from 0: make the ids job-wide
        t_b = time.perf_counter()
        allr = comm.allgather_r2p(recs) if comm is not None else fdist.allgather_r2p(recs, device=coll_dev if world > 1 else None)
        t_c = time.perf_counter()
        stats.update({k: st[k] for k in ("n_aligned", "n_rec", "n_sites", "n_rows", "n_arows", "n_pvars", "n_preads", "bytes_written")})
        stats["reads_phased"] = int((recs["block"] != -1).sum())
        stats["r2p_records"] = len(allr)
        host_t["phase_write"] += t_b - t_a
        host_t["allgather"] += t_c - t_b
        for k in sect:
            sect[k] += st[k]

    # N > 1: what ONE of these GPUs does on its own, on this node, in this process group -- rank 0 alone runs a few steps (no collective: nobody else is there) while the
    # others wait at the barrier; `scaling_estimate` in the line holds the N-rank rate against it, so that a sub-linear SCALE curve can be read off ONE line
    solo_ms = None
    if world > 1 and not args.strong:
        barrier()
        if rank == 0 and job is not None:
            def solo(k):
                st_, _ = job.phase_write(ids, names=name_tab, out_dir=os.path.join(out_root, ("solo%02d" % k) if args.fresh_trees else ("solo_%d" % (k & 1))), read_maps=maps, ctg_index=mine, consensus=args.with_consensus,
                                         async_writes=True, rebuild_index=not args.index_at_create)
            for k in range(2):
                solo(k)
            eng.synchronize(); eng.pipe_flush()
            t_s = time.perf_counter()
            for k in range(5):
                solo(2 + k)
            eng.synchronize(); eng.pipe_flush()
            solo_ms = (time.perf_counter() - t_s) / 5 * 1e3
        barrier()
    for _ in range(args.warmup):
        step()
    for k in host_t:
        host_t[k] = 0.0
    for k in sect:
        sect[k] = 0.0
    eng.prof_reset()
    eng.prof_enable(2 if prof_on else 0)      # the timed steps bracket the DP stage only (what the roofline needs); every other kernel's time comes from an instrumented pass below
    barrier()
    thr0 = thread_cpu() if os.environ.get("FZP_BENCH_THREAD_CPU") else None
    cg0 = cgroup_throttle()
    t0 = time.perf_counter()
    cpu0 = time.process_time()
    for _ in range(args.steps):
        step()
    barrier()
    dt = time.perf_counter() - t0
    cpu_ms_per_step = (time.process_time() - cpu0) / args.steps * 1e3      # user + system time of every thread of this rank (launch thread, host workers, file writers)
    cg_steps = throttle_delta(cg0, cgroup_throttle())
    if thr0 is not None and rank == 0:                                     # measurement aid: which threads that time belongs to
        thr1 = thread_cpu()
        rows = sorted(((thr1[t][1] - thr0.get(t, (thr1[t][0], 0.0, 0.0))[1], thr1[t][2] - thr0.get(t, (thr1[t][0], 0.0, 0.0))[2], thr1[t][0], t) for t in thr1), reverse=True)
        print("thread cpu ms/step (of which system): " + ", ".join("%s[%d] %.2f (%.2f)" % (nm, t, d / args.steps * 1e3, sy / args.steps * 1e3) for d, sy, nm, t in rows[:40] if d > 0),
              file=sys.stderr, flush=True)
    eng.prof_enable(False)
    prof = eng.prof()
    # the same steps once more with every kernel bracketed by HIP events (outside the timed region: sixty brackets cost a step ~1.3 ms and the rank ~6 ms of CPU)
    prof_all, ms_instr, n_instr = {}, None, 0
    host_t_timed, sect_timed = dict(host_t), dict(sect)      # (the timed steps' host sections: `step` goes on adding to the dictionaries it closes over)
    if prof_on and not args.no_kernel_breakdown:      # (every rank, also one without a job: the steps hold collectives)
        n_instr = max(2, min(args.steps, 5))
        eng.prof_reset()
        eng.prof_enable(1)
        barrier()
        t_i = time.perf_counter()
        for _ in range(n_instr):
            step()
        barrier()
        ms_instr = (time.perf_counter() - t_i) / n_instr * 1e3
        eng.prof_enable(False)
        prof_all = eng.prof()
    fresh_cmp = None
    if world == 1 and job is not None and not args.no_tree_compare:      # the same step with the output trees handled the other way, beside `value`
        def other_step(k):
            job.phase_write(ids, names=name_tab, out_dir=os.path.join(out_root, ("other%03d" % k) if not args.fresh_trees else ("other_%d" % (k & 1))), read_maps=maps, ctg_index=mine,
                            consensus=args.with_consensus, async_writes=True, rebuild_index=not args.index_at_create)
        for k in range(3):
            other_step(k)
        eng.synchronize(); eng.pipe_flush()
        t_f, c_f = time.perf_counter(), time.process_time()
        for k in range(8):
            other_step(3 + k)
        eng.synchronize(); eng.pipe_flush()
        fresh_cmp = {"trees": "fresh" if not args.fresh_trees else "rewritten", "ms_per_step": round((time.perf_counter() - t_f) / 8 * 1e3, 3),
                     "host_cpu_ms_per_step": round((time.process_time() - c_f) / 8 * 1e3, 2), "steps": 8,
                     "note": "fresh: an empty output directory per step (a first run); rewritten: two trees written in turn, files overwritten in place (a re-run into an existing tree)"}
    n_total = n_reads
    if world > 1:
        tt = torch.tensor([dt, float(n_reads)], dtype=torch.float64, device=coll_dev)
        mx = tt.clone()
        dist.all_reduce(mx, op=dist.ReduceOp.MAX)
        dist.all_reduce(tt, op=dist.ReduceOp.SUM)
        my_dt, dt, n_total = dt, float(mx[0].item()), int(tt[1].item())
        # every rank says what ITS steps were made of: a slow rank then names its cause in the one line rank 0 prints (host CPU per step, the calling thread's time in
        # phase_write and in the gather, its longest kernel bracket, how often its cgroup stopped it)
        slowest_k = max(((v[0] / max(1, v[1]), k) for k, v in prof.items()), default=(0.0, ""))
        knames = sorted(prof)
        mine_v = [my_dt, float(n_reads), cpu_ms_per_step, host_t["phase_write"] / args.steps * 1e3, host_t["allgather"] / args.steps * 1e3, slowest_k[0],
                  float(knames.index(slowest_k[1])) if slowest_k[1] in knames else -1.0,
                  float(cg_steps["nr_throttled"]) if cg_steps else -1.0, float(cg_steps["throttled_ms"]) if cg_steps else -1.0]
        per_rank = [torch.zeros(len(mine_v), dtype=torch.float64, device=coll_dev) for _ in range(world)]
        dist.all_gather(per_rank, torch.tensor(mine_v, dtype=torch.float64, device=coll_dev))
        rank_load = [{"rank": r, "reads": int(p[1].item()), "ms_per_step": round(float(p[0].item()) / args.steps * 1e3, 3), "host_cpu_ms_per_step": round(float(p[2].item()), 2),
                      "phase_write_ms": round(float(p[3].item()), 3), "gather_ms": round(float(p[4].item()), 3),
                      "slowest_bracket": {"name": knames[int(p[6].item())] if 0 <= int(p[6].item()) < len(knames) else None, "ms": round(float(p[5].item()), 3)},
                      "cgroup": None if p[7].item() < 0 else {"nr_throttled": int(p[7].item()), "throttled_ms": round(float(p[8].item()), 2)}} for r, p in enumerate(per_rank)]
    else:
        slowest_k = max(((v[0] / max(1, v[1]), k) for k, v in prof.items()), default=(0.0, ""))
        rank_load = [{"rank": 0, "reads": n_reads, "ms_per_step": round(dt / args.steps * 1e3, 3), "host_cpu_ms_per_step": round(cpu_ms_per_step, 2),
                      "phase_write_ms": round(host_t["phase_write"] / args.steps * 1e3, 3), "gather_ms": round(host_t["allgather"] / args.steps * 1e3, 3),
                      "slowest_bracket": {"name": slowest_k[1] or None, "ms": round(slowest_k[0], 3)}, "cgroup": cg_steps}]

    summ = job.summaries() if job is not None else np.zeros(0, dtype=[("cells", "<i8"), ("aligned", "<i4")])
    cells_per_step = float(summ["cells"].sum())
    sw_ms, sw_launches = prof.get("k1_sw", (0.0, 0))
    sw_avg_ms = sw_ms / max(1, sw_launches)
    cells_per_launch = cells_per_step * args.steps / max(1, sw_launches)
    dp_gcells = cells_per_launch / (sw_avg_ms * 1e-3) / 1e9 if sw_avg_ms > 0 else 0.0
    aligned_frac = float(summ["aligned"].mean()) if n_reads else 0.0
    cpu = None
    if rank == 0 and not args.no_cpu_baseline and job is not None and n_reads:
        cpu = cpu_baseline(contigs, blob, off, read_ctg, ids, job, summ, args.cpu_budget_s)      # rank 0's host cores; the other ranks wait at the next collective
    if job is not None:
        job.close()

    shaped = None
    if shaped_inp is not None:
        shaped = shaped_leg(eng, shaped_inp)
        shaped_inp = None
    polish = polish_leg(eng, contigs, blob, off, read_ctg) if (args.with_polish and rank == 0 and n_reads) else None
    strong = None
    if world > 1 and not args.strong and args.strong_leg_contigs > 0:
        strong = strong_leg(args, rank, world, eng, comm, coll_dev, s_inp, out_root)

    e2e = None
    if rank == 0 and world == 1 and not args.no_end_to_end:
        # the same workload from HOST buffers (never `value`): staging + H2D + packing inside, groups of contigs on lanes
        # two shapes -- the whole shard as one group on one lane, and groups of --e2e-group-contigs on --e2e-lanes lanes -- the faster is reported
        shapes = [(len(mine), 1), (args.e2e_group_contigs, args.e2e_lanes)]
        tried = []
        for gc, lanes in shapes:
            gb = int(gc * args.reads_per_contig * args.read_len * 1.06)
            runs = []
            for k in range(3):
                t1 = time.perf_counter()
                st, recs = _lib.phase_contigs(eng, contigs, blob, off, read_ctg, ids, names=name_tab, out_dir=os.path.join(out_root, "e2e_%d_%d_%d" % (gc, lanes, k)), read_maps=maps,
                                              ctg_index=mine, n_lanes=lanes, group_bases=gb, consensus=args.with_consensus, async_writes=True)
                runs.append((time.perf_counter() - t1, st))
            best = min(runs[1:], key=lambda x: x[0])
            tried.append({"reads_per_s": round(n_reads / best[0], 1), "ms": round(best[0] * 1e3, 2), "lanes": lanes, "groups": int(best[1]["n_groups"]),
                          "first_call_ms": round(runs[0][0] * 1e3, 2),
                          "host_section_ms_summed_over_lanes": {k: round(best[1][k], 2) for k in ("ms_upload", "ms_k1", "ms_phase", "ms_results", "ms_text")}})
        e2e = dict(min(tried, key=lambda x: x["ms"]))
        e2e["other_shape"] = {k: v for k, v in max(tried, key=lambda x: x["ms"]).items() if k in ("reads_per_s", "ms", "lanes", "groups")}
        e2e["note"] = "fzp_phase_contigs: host ASCII -> H2D -> pack -> K1..K5 -> texts -> files; PCIe-inclusive, reported beside `value`, never as it"

    from_files = None
    if e2e is not None and not args.no_from_files:
        # the same workload from the reference's own input files (unzip.py:204,233-234: reads/<ctg>_ref.fa, <ctg>_reads.fa) on a memory file system: FASTA parsing
        # (the library's, a contig group ahead of the lanes) inside the clock as well -- what scripts/fc_unzip_phase_gpu.py does per rank.  Never `value`.
        try:
            gc, lanes = (len(mine), 1) if e2e["lanes"] == 1 else (args.e2e_group_contigs, args.e2e_lanes)
            t_best, st_best, recs_f, last_dir, mb = files_leg(args, eng, contigs, blob, off, read_ctg, ids, name_tab, maps, mine, out_root, gc, lanes)
            _, recs_m = _lib.phase_contigs(eng, contigs, blob, off, read_ctg, ids, names=name_tab, out_dir=os.path.join(out_root, "files_ref"), read_maps=maps, ctg_index=mine,
                                           n_lanes=lanes, group_bases=int(gc * args.reads_per_contig * args.read_len * 1.06), consensus=args.with_consensus, async_writes=True)
            same = bool(np.array_equal(recs_f, recs_m)) and tree_digest(last_dir) == tree_digest(os.path.join(out_root, "files_ref"))
            from_files = {"reads_per_s": round(n_reads / t_best, 1), "ms": round(t_best * 1e3, 2), "lanes": lanes, "groups": int(st_best["n_groups"]),
                          "vs_end_to_end": round(t_best * 1e3 / e2e["ms"], 3), "same_bytes_as_from_memory": same, "input_mb": round(mb, 1),
                          "host_cpu_ms": st_best.get("host_cpu_ms"), "cgroup": st_best.get("cgroup"), "reader": "host" if os.environ.get("FZP_FASTA_HOST") else "device",
                          "h2d_floor_ms": round(mb / 37.0, 1),
                          "note": "fzp_phase_contigs_files: <ctg>_ref.fa / <ctg>_reads.fa on a memory file system -> pread into a pinned block, every 4 MB piece handed to the copy engine "
                                  "as it lands -> records found on the device (csrc/fzp_fasta.hip) -> pack -> K1..K5 -> texts -> files; reported beside `value`, never as it.  "
                                  "h2d_floor_ms: the files' bytes at the ~37 GB/s this host-to-device link sustains -- the last group's kernels cannot start before it"}
        except Exception as e:      # noqa: BLE001 -- reported in the line
            from_files = {"error": repr(e)}
    if world > 1 and not args.strong and not args.no_from_files:
        # N > 1: every rank its own shard's files, all ranks side by side (a barrier before every call), all reads / the slowest rank -- SURVEY 8d's reads phased/sec
        # "end-to-end incl. host I/O" for the whole node.  A rank that fails says so; the collectives below are entered by every rank either way.
        err, t_mine, passed = "", 0.0, [0]

        def side_by_side():
            dist.barrier()
            passed[0] += 1
        try:
            t_mine, st_best, _, _, mb = files_leg(args, eng, contigs, blob, off, read_ctg, ids, name_tab, maps, mine, out_root, args.e2e_group_contigs, args.e2e_lanes, sync=side_by_side)
        except Exception as e:      # noqa: BLE001
            err = "rank %d: %r" % (rank, e)
            print("bench.py from_files: " + err, file=sys.stderr, flush=True)
            while passed[0] < 3:
                side_by_side()      # the barriers the other ranks' remaining calls wait at
        per_rank = [torch.zeros(3, dtype=torch.float64, device=coll_dev) for _ in range(world)]
        dist.all_gather(per_rank, torch.tensor([t_mine, float(n_reads), 0.0 if err else 1.0], dtype=torch.float64, device=coll_dev))
        if all(float(p[2].item()) == 1.0 for p in per_rank):
            slowest = max(float(p[0].item()) for p in per_rank)
            from_files = {"reads_per_s": round(sum(float(p[1].item()) for p in per_rank) / slowest, 1), "ms": round(slowest * 1e3, 2), "lanes": args.e2e_lanes, "n_gpus": world,
                          "rank_ms": [round(float(p[0].item()) * 1e3, 2) for p in per_rank],
                          "note": "every rank its own 20 contigs from <ctg>_ref.fa / <ctg>_reads.fa on a memory file system through fzp_phase_contigs_files, all ranks at once: all reads / slowest rank"}
        else:
            from_files = {"error": err or "another rank failed"}

    pipelined = None
    if rank == 0 and world == 1 and not args.no_end_to_end and not os.environ.get("FZP_BENCH_NO_PIPELINED"):
        # the resident step again, two steps in flight: two contexts, each with its own resident copy of the job, alternate steps on two host
        # threads, so the record / text downloads and the host formatting of one step run under the kernels of the next (what
        # fzp_phase_contigs' lanes do for a stream of contig groups).  Reported beside `value`, which stays the one-step-at-a-time figure.
        import threading
        eng2 = _lib.Engine(dev_index)
        jobs = [_lib.align_job_raw(eng, contigs, blob, off, read_ctg), _lib.align_job_raw(eng2, contigs, blob, off, read_ctg)]
        n_each = max(2, min(4, (args.steps + 1) // 2))
        err = []

        def lane(li, n, tag):
            try:
                for k in range(n):
                    jobs[li].phase_write(ids, names=name_tab, out_dir=os.path.join(out_root, "pipe_%s_%d_%d" % (tag, li, k)), read_maps=maps, ctg_index=mine,
                                         consensus=args.with_consensus, async_writes=True)
            except Exception as e:      # noqa: BLE001 -- reported below
                err.append(repr(e))

        def both(n, tag):
            th = [threading.Thread(target=lane, args=(li, n, tag)) for li in range(2)]
            for t in th:
                t.start()
            for t in th:
                t.join()
            eng.synchronize(); eng2.synchronize()
            eng.pipe_flush(); eng2.pipe_flush()

        both(1, "w")
        t1 = time.perf_counter()
        both(n_each, "t")
        dt2 = time.perf_counter() - t1
        for jb in jobs:
            jb.close()
        eng2.close()
        pipelined = ({"error": err[0]} if err else
                     {"steps_in_flight": 2, "steps": 2 * n_each, "ms_per_step": round(dt2 / (2 * n_each) * 1e3, 3), "reads_per_s": round(n_reads * 2 * n_each / dt2, 1),
                      "note": "same step, same resident inputs (one copy per context), two steps in flight; not `value`"})

    if rank == 0:
        ms_per_step = dt / args.steps * 1e3
        traffic = None
        tf = os.path.join(REPO, "profiles", "k1_sw_hbm_traffic.json")
        if os.path.exists(tf):
            with open(tf) as f:
                traffic = json.load(f).get("bytes_per_launch")
        out = {
            # BASELINE.json's metric, verbatim; `value` is its reads-phased/sec half (whole job), the DP half is
            # `dp_gcell_per_s_per_gpu` below
            "metric": "DP Gcell/s/GPU + reads phased/sec, 15 kb reads x 5 Mb contigs, 1/2/4/8 GPUs",
            "value": round(n_total * args.steps / dt, 2),
            "unit": "reads/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(ms_per_step, 3),
            "higher_is_better": True, "scaling": "strong" if args.strong else "weak", "vs_baseline": None,
            "dtype": "int32", "data": "synthetic",
            "config": {"workload": ("cfg3 (strong): %d contigs x %d bp in total, 0.5x..2x %d reads x %d bp each, LPT over ranks; " % (args.contigs, args.contig_len, args.reads_per_contig, args.read_len)
                                    if args.strong else
                                    "cfg2: per GPU %d contigs x %d bp, %d reads x %d bp template each (CLR 1/8/4 %% errors, both strands), reads drawn from a %d bp window per contig; "
                                    % (args.contigs, args.contig_len, args.reads_per_contig, args.read_len, win))
                                   + "inputs resident in HBM (2-bit packed reads and contigs); inside the step: " + ("" if not args.index_at_create else "[k-mer tables from fzp_align_create, NOT in the step] ")
                                   + "K1 " + ("k-mer tables + " if not args.index_at_create else "") + "align (fzalign) + K2 het call + K3 atable + K4 blocks + K5 reads + all seven files of every contig "
                                     "serialised AND written + readmap + r2p all-gather" + (" + K6 consensus" if args.with_consensus else ""),
                       "reads_total": n_total, "reads_per_gpu": n_reads, "parallelism": "contigs sharded, %d rank(s)" % world},
            "dp_gcell_per_s_per_gpu": round(dp_gcells, 2),
            "upload_ms": round(upload_ms, 1),
            "dp_cells_per_step": cells_per_step,
            "aligned_frac": round(aligned_frac, 4),
            "stage_counts": {k: int(v) for k, v in stats.items()},
            "kernel_ms_per_step": {k: round(v[0] / max(1, n_instr), 3) for k, v in sorted(prof_all.items())},      # from the instrumented pass (every kernel bracketed), not the timed steps
            "ms_per_step_instrumented": round(ms_instr, 3) if ms_instr else None, "instrumented_steps": n_instr,
            "host_cpu_ms_per_step": round(cpu_ms_per_step, 2),
            "out_fs": fs_of(out_root), "out_tree": "a fresh tree per step" if args.fresh_trees else "two trees written in turn (files overwritten in place from the third step on: a re-run)",
            "other_tree_mode": fresh_cmp, "sched_flag_rc": _lib.sched_status(),      # 0: the runtime took the blocking-sync scheduling flag
            "host_wall_ms_per_step": dict({k: round(v / args.steps * 1e3, 3) for k, v in host_t_timed.items()}, **{k[3:]: round(v / args.steps, 3) for k, v in sect_timed.items()}),
            "rank_load": rank_load,
            # N > 1: this run's rate against rank 0 running the same step ALONE on its GPU a moment earlier (same node, same process group): what the driver's SCALE curve
            # will show for this N, from one line -- with rank_load saying which rank held the others up and with what
            "scaling_estimate": ({"solo_ms_per_step": round(solo_ms, 3), "solo_reads_per_s": round(n_reads / (solo_ms * 1e-3), 1), "n_gpus": world,
                                  "speedup_vs_one_gpu": round((n_total / (dt / args.steps)) / (n_reads / (solo_ms * 1e-3)), 3),
                                  "slowest_rank": max(rank_load, key=lambda r: r["ms_per_step"])["rank"], "out_fs": out_root}
                                 if (solo_ms and world > 1) else None),
            "gather": "fzp_allgather_rid_to_phase (RCCL, C-ABI)" if comm is not None else ("torch.distributed all_gather (%s)" % backend if world > 1 else "none (1 rank)"),
            "rccl_ranks": comm.ranks()[1] if comm is not None else 0,      # size of the RCCL communicator as ncclCommCount reports it (0: no RCCL communicator in this run)
            "gather_fallback": gather_note,                                  # why the C-ABI RCCL gather was not used (the library's own error text inside), or None
            "rccl_path": rccl_lib[0], "rccl_version": rccl_lib[1],        # which RCCL the library bound in this process (dladdr of ncclAllGather, ncclGetVersion) -- or why none
            "index_in_step": not args.index_at_create,
            "index_ms": round(prof_all.get("k1_index", (0.0, 0))[0] / max(1, n_instr), 3) if not args.index_at_create else round(index_ms_at_create, 3),
            "value_from_files": from_files.get("reads_per_s") if from_files else None,      # SURVEY 8d's "reads phased/sec (end-to-end incl. host I/O)": FASTA files in, files out
            "two_core_step_ms": two_core.get("ms_per_step") if two_core else None,           # the resident step on TWO CPUs with LOCAL_WORLD_SIZE=8 (a rank's share of a 16-CPU, 8-GPU node)
            "two_core": dict(two_core, vs_unconstrained=round(two_core["ms_per_step"] / ms_per_step, 3),
                             from_files_vs_unconstrained=(round(two_core["from_files"]["ms"] / from_files["ms"], 3)
                                                          if (two_core.get("from_files") and "ms" in two_core["from_files"] and from_files and "ms" in from_files) else None))
            if (two_core and "ms_per_step" in two_core) else two_core,
            "value_end_to_end": e2e["reads_per_s"] if e2e else None,       # host ASCII in, PCIe + packing + index inside (SURVEY 8d's reads-phased/sec); `value` keeps inputs resident (bench contract)
            "end_to_end": e2e,
            "from_files": from_files,
            "two_steps_in_flight": pipelined,
            "roofline": roofline(cells_per_launch, sw_avg_ms, sw_launches, dp_gcells, traffic, band=_lib.align_band()),
        }
        if cpu is not None:
            out["cpu_baseline"] = cpu
        if shaped is not None:
            out["k1_on_real_read_shape"] = shaped
        if polish is not None:
            out["polish_tigs"] = polish
        if strong is not None:
            out["strong_cfg3"] = strong
        print(json.dumps(out), flush=True)
    if comm is not None:
        comm.close()
    eng.close()
    shutil.rmtree(out_root, ignore_errors=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
