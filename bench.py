#!/usr/bin/env python3
"""bench.py -- the hot path on synthetic input, one JSON line on stdout (rank 0).

Workload (BASELINE.json configs[1], SURVEY.md section 8d "cfg2"): per GPU 20 primary contigs of 5 Mb,
2 000 simulated PacBio CLR reads of 15 kb template each (sub 1 % / ins 8 % / del 4 %, both strands);
the reads of a contig are drawn from a 750 kb window of it, so the covered region is at 40x and the
het-call gate `total >= 10` (phasing.py:112) has something to call, while seeding and the banded DP
still run against the full 5 Mb contig.

One step = one pass of the whole hot path over that batch, inputs already resident (packed) in HBM:
  K1 index + seed + banded DP + trace-back -> records ("samtools sort" order, record filters)
  K2 pileup + het call -> K3 association table -> K4 phase blocks -> K5 read phasing
  rid_to_phase records -> one all-gather across ranks (skipped at world size 1).
`value` = reads processed by all ranks / max-over-ranks step time.  `dp_gcell_per_s_per_gpu` is the
banded-DP rate of the dominant kernel (k1_sw) from HIP events on the library's stream.
"""
from __future__ import annotations

import argparse
import json
import multiprocessing as mp
import os
import sys
import time

import numpy as np

REPO = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, REPO)

HBM_PEAK_GBS = 8000.0            # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
VALU_PEAK_GINST = 614.4          # wave64 int32 VALU instructions/s: 256 CUs x 4 SIMDs x 2.4 GHz / 4 cycles per wave64 op
                                 # (the 157.3 TFLOP/s vector figure counts packed fp32, which integer ops do not have)
SW_VALU_PER_STEP = 12.5            # VALU instructions per 64-cell band step in k_sw's interior block (fzp_align.hip: sw_block)
SW_BYTES_PER_CELL = 0.25         # algorithmic: 2 trace-back bits per cell (16 B per 64-cell step); sequence
                                 # reads add 2 bits per band step, i.e. < 0.01 B/cell


def gen_contig(args):
    cfg, ci, L, n_reads, R, win = args
    from falcon_unzip_amd import sim
    rng = sim.rng_for(cfg, ci)
    hap0, hap1, _ = sim.make_diploid(L, rng)
    lo = int(rng.integers(0, L - win + 1))
    codes, off, st, hp, sd = sim.simulate_raw_reads_bulk(hap0, hap1, n_reads, R, rng, lo=lo, hi=lo + win)
    return sim.ACGT[hap0].tobytes(), sim.ACGT[codes].tobytes(), off


def make_inputs(rank, n_ctg, L, n_reads, R, win, workers):
    jobs = [(2, rank * n_ctg + c, L, n_reads, R, win) for c in range(n_ctg)]
    if workers > 1:
        with mp.get_context("fork").Pool(workers) as pool:     # before any GPU initialisation
            res = pool.map(gen_contig, jobs)
    else:
        res = [gen_contig(j) for j in jobs]
    contigs = [r[0] for r in res]
    blob = b"".join(r[1] for r in res)
    offs, read_ctg, base = [np.zeros(1, np.int64)], [], 0
    for c, r in enumerate(res):
        offs.append(r[2][1:] + base)
        base += int(r[2][-1])
        read_ctg.append(np.full(len(r[2]) - 1, c, np.int32))
    return contigs, blob, np.concatenate(offs), np.concatenate(read_ctg)


def cpu_baseline(contigs, blob, off, read_ctg, eng, sample_reads):
    """The oracle ("port": scalar C restatement) on a bounded sample: the first `sample_reads` reads of
    contig 0 through the CPU twin aligner, then the oracle phasing chain on the SAM text of those reads."""
    from falcon_unzip_amd import _lib
    from tests import oracle_lib
    orc = oracle_lib.load()
    idx = np.flatnonzero(read_ctg == 0)[:sample_reads]
    reads = [blob[off[i]:off[i + 1]] for i in idx]
    t0 = time.perf_counter()
    summ, _ = oracle_lib.align_reads(orc, contigs[0], reads)
    t_aln = time.perf_counter() - t0
    job = _lib.align_job(eng, [contigs[0]], reads)      # only to obtain the SAM text the oracle chain reads
    job.run()
    aln, _ = job.alnset(0)
    sam = _lib.format_sam(aln, "c0")
    job.close()
    t0 = time.perf_counter()
    orc.phase_all(sam, contigs[0], "c0")
    t_ph = time.perf_counter() - t0
    cells = float(summ["cells"].sum())
    return {"value": round(len(reads) / (t_aln + t_ph), 3), "unit": "reads/s", "cores": 1, "kind": "port",
            "sample": "%d reads (15 kb) of contig 0 vs its 5 Mb contig: oracle/align_oracle.c then oracle/phasing_oracle.c chain, 1 thread; "
                      "the reference's blasr and Python 2 cannot run here" % len(reads),
            "align_s": round(t_aln, 3), "phasing_s": round(t_ph, 3), "dp_gcell_per_s": round(cells / t_aln / 1e9, 4)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--contigs", type=int, default=20)
    ap.add_argument("--contig-len", type=int, default=5_000_000)
    ap.add_argument("--reads-per-contig", type=int, default=2000)
    ap.add_argument("--read-len", type=int, default=15000)
    ap.add_argument("--window", type=int, default=750_000)
    ap.add_argument("--cpu-sample-reads", type=int, default=400)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--with-consensus", action="store_true", help="also run K6 (phased-pile consensus, BASELINE config 4) inside every step")
    ap.add_argument("--gen-workers", type=int, default=0, help="processes for input generation (0 = auto; forced to 1 under rocprofv3)")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    workers = args.gen_workers or max(1, min(8, (os.cpu_count() or 1) // max(1, world)))
    if "rocprof" in os.environ.get("LD_PRELOAD", "") or os.environ.get("ROCPROFILER_REGISTER_FORCE_LOAD") or os.environ.get("ROCP_TOOL_LIBRARIES"):
        workers = 1   # the profiler's preloaded library may have initialised the GPU already: do not fork
    contigs, blob, off, read_ctg = make_inputs(rank, args.contigs, args.contig_len, args.reads_per_contig, args.read_len,
                                               min(args.window, args.contig_len), workers)
    n_reads = len(read_ctg)

    import torch
    import torch.distributed as dist
    from falcon_unzip_amd import _lib
    from falcon_unzip_amd import dist as fdist
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
    t_up = time.perf_counter()
    job = _lib.align_job_raw(eng, contigs, blob, off, read_ctg)     # upload + 2-bit pack: inputs now resident in HBM
    eng.synchronize()
    upload_ms = (time.perf_counter() - t_up) * 1e3                  # PCIe + packing, outside the timed region

    def barrier():
        eng.synchronize()
        if world > 1:
            dist.barrier()
            torch.cuda.synchronize(dev_index)

    stats = {}
    reads_per_ctg = np.bincount(read_ctg, minlength=args.contigs).tolist()

    host_t = {"align_run": 0.0, "to_batch": 0.0, "phase_run": 0.0, "results": 0.0, "allgather": 0.0}

    def step():
        t_a = time.perf_counter()
        job.run()
        t_b = time.perf_counter()
        b = job.to_batch()
        t_c = time.perf_counter()
        b.run(_lib.STAGE_ALL)
        t_d = time.perf_counter()
        recs = []
        if args.with_consensus:
            tg = b.consensus()
            stats["n_tigs"] = len(tg.tigs)
            stats["tig_bases"] = len(tg.seq)
            tg.close()
        b.results(copy=False)
        full, beg = b.last_full                      # whole-batch records: one vectorised pass instead of one per contig
        local = fdist.r2p_from_batch(full.preads, beg["pread"], reads_per_ctg, rank * n_reads, rank * args.contigs)   # reads_per_ctg: upper bound, aligned reads get q_ids
        n_phased = int((local["block"] != -1).sum())
        recs.append(local)
        stats.update(b.counts())
        stats["reads_phased"] = n_phased
        b.close()
        t_e = time.perf_counter()
        allr = fdist.allgather_r2p(np.concatenate(recs), device=coll_dev if world > 1 else None)
        stats["r2p_records"] = len(allr)
        t_f = time.perf_counter()
        for k, v in zip(host_t, (t_b - t_a, t_c - t_b, t_d - t_c, t_e - t_d, t_f - t_e)):
            host_t[k] += v

    for _ in range(args.warmup):
        step()
    for k in host_t:
        host_t[k] = 0.0
    eng.prof_reset()
    eng.prof_enable(True)
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    barrier()
    dt = time.perf_counter() - t0
    eng.prof_enable(False)
    prof = eng.prof()
    if world > 1:
        tt = torch.tensor([dt], dtype=torch.float64, device=coll_dev)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt.item())

    summ = job.summaries()
    cells_per_step = float(summ["cells"].sum())
    sw_ms, sw_launches = prof.get("k1_sw", (0.0, 0))
    sw_avg_ms = sw_ms / max(1, sw_launches)
    cells_per_launch = cells_per_step * args.steps / max(1, sw_launches)
    dp_gcells = cells_per_launch / (sw_avg_ms * 1e-3) / 1e9 if sw_avg_ms > 0 else 0.0

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
            "value": round(world * n_reads * args.steps / dt, 2),
            "unit": "reads/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(ms_per_step, 3),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "int32", "data": "synthetic",
            "config": {"workload": "cfg2: per GPU %d contigs x %d bp, %d reads x %d bp template each (CLR 1/8/4 %% errors, both strands), "
                                   "reads drawn from a %d bp window per contig; K1 align + K2 het call + K3 atable + K4 blocks + K5 reads + r2p all-gather"
                                   % (args.contigs, args.contig_len, args.reads_per_contig, args.read_len, min(args.window, args.contig_len))
                                   + (" + K6 consensus" if args.with_consensus else ""),
                       "reads_per_gpu": n_reads, "parallelism": "contigs sharded, %d rank(s)" % world},
            "dp_gcell_per_s_per_gpu": round(dp_gcells, 2),
            "upload_ms": round(upload_ms, 1),
            "dp_cells_per_step": cells_per_step,
            "aligned_frac": round(float(summ["aligned"].mean()), 4),
            "stage_counts": {k: int(v) for k, v in stats.items()},
            "kernel_ms_per_step": {k: round(v[0] / args.steps, 3) for k, v in sorted(prof.items())},
            "host_wall_ms_per_step": {k: round(v / args.steps * 1e3, 3) for k, v in host_t.items()},
            # the segment that is bit-exact against the reference (alignments given -> phased reads): K2..K5 + record download
            "phasing_only": {"ms_per_step": round((host_t["phase_run"] + host_t["results"]) / args.steps * 1e3, 3),
                             "reads_per_s": round(n_reads * args.steps / max(1e-9, host_t["phase_run"] + host_t["results"]), 1)},
            "roofline": {"bound": "hbm", "kernel": "k1_sw", "achieved": round(cells_per_launch * SW_BYTES_PER_CELL / (sw_avg_ms * 1e-3) / 1e9, 2) if sw_avg_ms else 0.0,
                         "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": round(cells_per_launch * SW_BYTES_PER_CELL / (sw_avg_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 5) if sw_avg_ms else 0.0,
                         "traffic": traffic, "avg_launch_ms": round(sw_avg_ms, 3), "launches": int(sw_launches),
                         "note": "k1_sw is VALU-issue-bound by construction (0.25 algorithmic B/cell); the issue view is in `valu`",
                         "valu": {"insts_per_step": SW_VALU_PER_STEP, "achieved_ginst": round(dp_gcells / 64.0 * SW_VALU_PER_STEP, 2),
                                  "peak_ginst": VALU_PEAK_GINST, "frac": round(dp_gcells / 64.0 * SW_VALU_PER_STEP / VALU_PEAK_GINST, 4)}},
        }
        if not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(contigs, blob, off, read_ctg, eng, args.cpu_sample_reads)
        print(json.dumps(out))
    job.close()
    eng.close()
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
