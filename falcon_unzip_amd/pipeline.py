"""Batched replacement for the per-contig job fan-out of `unzip_all` (falcon_unzip/unzip.py:221-288).

The reference starts, per contig, one blasr job (`task_run_blasr`, unzip.py:61-99) and one phasing job
(`task_phasing`, unzip.py:102-133: fc_phasing.py + fc_phasing_readmap.py), then concatenates every
`rid_to_phase.<ctg>` (`get_rid_to_phase_all`, unzip.py:303-314).  On a GPU node the same work is one
process per GPU: this rank's contigs (LPT by read bases) go through K1 -> K5 in batched launches, the
files the reference would have written are produced from the records, and one all-gather assembles
`rid_to_phase.all`.

Inputs and outputs use the reference's directory contract:
    <unzip_dir>/reads/ctg_list, <ctg>_ref.fa, <ctg>_reads.fa                       (unzip.py:204,233-234)
    <unzip_dir>/0-phasing/<ctg>/{het_call/*, g_atable/atable, get_phased_blocks/phased_variants,
                                  phased_reads, rid_to_phase.<ctg>, blasr/<ctg>_sorted.bam(.bai), cns/phased_blocks.fa}
    <unzip_dir>/1-hasm/rid-to-phase-all/rid_to_phase.all                          (unzip.py:285)
"""
from __future__ import annotations

import os
import sys

import numpy as np

from . import _lib
from . import dist as fdist


def read_fasta(path):
    """-> list of (name (first word), sequence bytes)."""
    out, name, chunks = [], None, []
    with open(path, "rb") as f:
        for line in f:
            line = line.rstrip(b"\r\n")
            if line.startswith(b">"):
                if name is not None:
                    out.append((name, b"".join(chunks)))
                name, chunks = (line[1:].split() or [b""])[0], []
            else:
                chunks.append(line.strip())
    if name is not None:
        out.append((name, b"".join(chunks)))
    return out


def load_contig_jobs(unzip_dir, ctg_ids=None):
    """Contig ids (from reads/ctg_list unless given) with their reference sequence and reads."""
    reads_dir = os.path.join(unzip_dir, "reads")
    if ctg_ids is None:
        with open(os.path.join(reads_dir, "ctg_list")) as f:
            ctg_ids = [l.strip() for l in f if l.strip()]
    jobs = []
    for ctg in ctg_ids:
        ref = None
        for name, seq in read_fasta(os.path.join(reads_dir, "%s_ref.fa" % ctg)):
            if name.decode() == ctg:          # phasing.py:489-494
                ref = seq.upper()
        if ref is None:
            ref = b""
        reads = read_fasta(os.path.join(reads_dir, "%s_reads.fa" % ctg))
        jobs.append((ctg, ref, reads))
    return jobs


def _mkdirs(*paths):
    for p in paths:
        os.makedirs(p, exist_ok=True)


def _slurp(p):
    with open(p, "rb") as f:
        return f.read()


def load_read_maps(read_map_dir):
    """The three files fc_phasing_readmap.py reads (phasing_readmap.py:15-16,36), once per rank."""
    return (_slurp(os.path.join(read_map_dir, "dump_rawread_ids", "rawread_ids")), _slurp(os.path.join(read_map_dir, "dump_pread_ids", "pread_ids")),
            _slurp(os.path.join(read_map_dir, "pread_to_contigs")))


def _touch(path):
    os.makedirs(os.path.dirname(path), exist_ok=True)
    with open(path, "a"):
        pass


def phase_contigs(eng, jobs, unzip_dir, read_map_dir=None, write_sam=False, ctg_indices=None, consensus=True, n_lanes=0, group_bases=0, sentinels=True):
    """K1 -> K5 (+ K6) for `jobs` (list of (ctg_id, ref bytes, [(read name, seq)])) on one GPU; writes the per-contig files;
    returns the rid_to_phase records of these contigs (empty unless read_map_dir is given).  ctg_indices: the contigs'
    indices in the job-wide sorted contig list.

    Everything runs inside the library in ONE alignment pass (fzp_phase_contigs: contig groups streamed through the device, files written
    by its host threads).  write_sam=True additionally leaves the blasr task's artefacts (<ctg>/blasr/<ctg>_sorted.bam + .bai,
    unzip.py:86-91) -- made from the same pass's records, group by group, compressed on the library's writer threads.  sentinels=True
    leaves what an unchanged fc_unzip.py looks for to call the two per-contig tasks done (unzip.py:239-241,267-269):
    <ctg>/blasr/aln_<ctg>_done (with write_sam) and <ctg>/phasing/p_<ctg>_done, each with its `.exit` twin (the scripts' `trap ... EXIT`,
    unzip.py:81,120); if the call fails as a whole, every contig of it gets the `.exit` files only."""
    if ctg_indices is None:
        ctg_indices = list(range(len(jobs)))
    contigs = [j[1] for j in jobs]
    names, blobs, lens, read_ctg = [], [], [], []
    for c, (_, _, reads) in enumerate(jobs):
        for nm, seq in reads:
            names.append(nm)
            blobs.append(seq)
            lens.append(len(seq))
            read_ctg.append(c)
    offs = np.zeros(len(lens) + 1, np.int64)
    offs[1:] = np.cumsum(lens)
    blob = b"".join(blobs)
    read_ctg = np.array(read_ctg, np.int32)
    enc = [nm.encode() if isinstance(nm, str) else nm for nm in names]
    noff = np.zeros(len(enc) + 1, np.int64)
    noff[1:] = np.cumsum([len(e) for e in enc])
    name_tab = (noff, b"".join(enc))                          # built once for the whole rank
    maps = load_read_maps(read_map_dir) if read_map_dir is not None else None
    out_dir = os.path.join(unzip_dir, "0-phasing")
    os.makedirs(out_dir, exist_ok=True)
    try:
        stats, recs = _lib.phase_contigs(eng, contigs, blob, offs, read_ctg, [j[0] for j in jobs], names=name_tab, out_dir=out_dir, read_maps=maps,
                                         ctg_index=ctg_indices, n_lanes=n_lanes, group_bases=group_bases, consensus=consensus, bam=write_sam, sentinels=sentinels)
    except Exception:
        if sentinels:                                          # the tasks exited without finishing: `.exit` only (contigs the library did finish keep their _done files)
            for ctg, _, _ in jobs:
                if write_sam:
                    _touch(os.path.join(out_dir, ctg, "blasr", "aln_%s_done.exit" % ctg))
                _touch(os.path.join(out_dir, ctg, "phasing", "p_%s_done.exit" % ctg))
        raise
    phase_contigs.last_stats = stats
    return recs


def phase_contigs_files(eng, unzip_dir, ctg_ids, read_map_dir=None, write_sam=False, ctg_indices=None, consensus=True, n_lanes=0, group_bases=0, sentinels=True):
    """phase_contigs with the inputs read by the library itself from `<unzip_dir>/reads/<ctg>_ref.fa` / `<ctg>_reads.fa` (unzip.py:204,233-234; the
    record of <ctg>_ref.fa named <ctg> is the contig, phasing.py:489-494): fzp_phase_contigs_files.  Same outputs, same sentinels."""
    if ctg_indices is None:
        ctg_indices = list(range(len(ctg_ids)))
    maps = load_read_maps(read_map_dir) if read_map_dir is not None else None
    out_dir = os.path.join(unzip_dir, "0-phasing")
    os.makedirs(out_dir, exist_ok=True)
    try:
        stats, recs = _lib.phase_contigs_files(eng, os.path.join(unzip_dir, "reads"), list(ctg_ids), out_dir=out_dir, read_maps=maps, ctg_index=ctg_indices,
                                               n_lanes=n_lanes, group_bases=group_bases, consensus=consensus, bam=write_sam, sentinels=sentinels)
    except Exception:
        if sentinels:
            for ctg in ctg_ids:
                if write_sam:
                    _touch(os.path.join(out_dir, ctg, "blasr", "aln_%s_done.exit" % ctg))
                _touch(os.path.join(out_dir, ctg, "phasing", "p_%s_done.exit" % ctg))
        raise
    phase_contigs_files.last_stats = stats
    return recs


def run(unzip_dir, read_map_dir=None, ctg_ids=None, device=None, write_sam=True, consensus=True):
    """Whole phasing section of unzip_all for this rank (one process per GPU under torch.distributed.run): the rank's contigs streamed from
    `reads/<ctg>_{ref,reads}.fa` by the library (fzp_phase_contigs_files: the FASTA files are parsed by its host threads a contig group ahead of the
    device, no whole-rank lists in Python), one exchange step -- the library's own RCCL all-gather (fzp_allgather_rid_to_phase) when the ranks
    sit on their own GPUs under backend nccl, torch.distributed's otherwise (gloo dry runs; said on stderr when it is a fallback)."""
    import torch.distributed as tdist
    rank = tdist.get_rank() if tdist.is_available() and tdist.is_initialized() else 0
    world = tdist.get_world_size() if tdist.is_available() and tdist.is_initialized() else 1
    if ctg_ids is None:
        with open(os.path.join(unzip_dir, "reads", "ctg_list")) as f:
            ctg_ids = [l.strip() for l in f if l.strip()]
    ctg_ids = sorted(ctg_ids)                                  # rid_to_phase.all is in sorted-path order (unzip.py:306-307)
    weights = []
    for ctg in ctg_ids:                                        # LPT weight ~ read bases ~ DP cells
        try:
            weights.append(os.path.getsize(os.path.join(unzip_dir, "reads", "%s_reads.fa" % ctg)))
        except OSError:
            weights.append(0)
    mine = fdist.shard_contigs(weights, world)[rank]
    dev = int(os.environ.get("LOCAL_RANK", "0")) if device is None else device
    if world > 1 and tdist.get_backend() != "nccl":            # dry runs: several ranks share what devices there are
        import torch
        dev = dev % max(1, torch.cuda.device_count())
    eng = _lib.Engine(dev)
    comm, note = None, None
    if world > 1 and tdist.get_backend() == "nccl" and os.environ.get("FZP_GATHER", "cabi") == "cabi":
        comm, note = fdist.make_comm(eng, rank, world, "cuda:%d" % dev)
        if comm is None:
            print("pipeline.run: " + note, file=sys.stderr, flush=True)
    # A rank whose phasing fails must not leave its peers waiting in the exchange step (fzp_allgather_rid_to_phase has no watchdog of its own): every rank says how it
    # went over the process group that started the ranks FIRST; the gather is entered only when all went well; the failing rank's exception is raised on it, the others
    # raise a RuntimeError that names the failure; communicator and engine are closed either way.
    local, failure = np.zeros(0, _lib.R2P), None
    try:
        try:
            if mine:
                # contig indices are global so that the gathered records sort like the reference's file list
                local = phase_contigs_files(eng, unzip_dir, [ctg_ids[i] for i in mine], read_map_dir, write_sam=write_sam, ctg_indices=mine, consensus=consensus)
        except Exception as e:      # noqa: BLE001 -- re-raised below, after the ranks have agreed not to gather
            failure = e
        if world > 1:
            import torch
            flag = torch.tensor([0 if failure is None else 1], dtype=torch.int32, device=("cuda:%d" % dev) if tdist.get_backend() == "nccl" else "cpu")
            tdist.all_reduce(flag, op=tdist.ReduceOp.MAX)
            if int(flag.item()) and failure is None:
                failure = RuntimeError("pipeline.run: another rank failed in its phasing step; rid_to_phase.all was not assembled")
        if failure is not None:
            raise failure
        allr = fdist.gather_r2p(local, comm)
        run.last_gather = "fzp_allgather_rid_to_phase (RCCL)" if comm is not None else ("torch.distributed (%s)" % tdist.get_backend() if world > 1 else "none (1 rank)")
    finally:
        if comm is not None:
            comm.close()
        eng.close()
    if rank == 0 and read_map_dir is not None:
        out_dir = os.path.join(unzip_dir, "1-hasm", "rid-to-phase-all")
        os.makedirs(out_dir, exist_ok=True)
        with open(os.path.join(out_dir, "rid_to_phase.all"), "wb") as f:
            f.write(fdist.format_rid_to_phase_all(allr, ctg_ids))
    return allr
