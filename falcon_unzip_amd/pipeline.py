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


def phase_contigs(eng, jobs, unzip_dir, read_map_dir=None, write_sam=True, ctg_indices=None, consensus=True):
    """K1 -> K5 for `jobs` (list of (ctg_id, ref bytes, [(read name, seq)])) on one GPU, all contigs in the same
    launches; writes the per-contig files; returns the rid_to_phase records of these contigs (empty unless
    read_map_dir is given).  ctg_indices: the contigs' indices in the job-wide sorted contig list."""
    if ctg_indices is None:
        ctg_indices = list(range(len(jobs)))
    contigs = [j[1] for j in jobs]
    names, blobs, offs, read_ctg = [], [], [0], []
    for c, (_, _, reads) in enumerate(jobs):
        for nm, seq in reads:
            names.append(nm)
            blobs.append(seq)
            offs.append(offs[-1] + len(seq))
            read_ctg.append(c)
    job = _lib.align_job_raw(eng, contigs, b"".join(blobs), np.array(offs, np.int64), np.array(read_ctg, np.int32))
    job.run()
    summ = job.summaries()
    batch = job.to_batch()
    batch.run(_lib.STAGE_ALL)
    tigs = batch.consensus() if consensus else None          # K6 before results(): results() hands out pinned views
    results = batch.results()
    all_recs = []
    for c, (ctg, ref, reads) in enumerate(jobs):
        base = os.path.join(unzip_dir, "0-phasing", ctg)
        _mkdirs(os.path.join(base, "het_call"), os.path.join(base, "g_atable"), os.path.join(base, "get_phased_blocks"),
                os.path.join(base, "blasr"))
        aln, idx = job.alnset(c, names)                      # records in (POS, read) order + the q_id table
        r = results[c]
        qoff, qnames = aln.qname_table()

        def put(rel, data):
            with open(os.path.join(base, rel), "wb") as f:
                f.write(data)
        put("het_call/variant_pos", _lib.format_variant_pos(r.sites))
        put("het_call/variant_map", _lib.format_variant_map(r.sites, r.vmap_qid))
        put("het_call/q_id_map", _lib.format_q_id_map(aln))
        put("g_atable/atable", _lib.format_atable(r.sites, r.arows))
        put("get_phased_blocks/phased_variants", _lib.format_phased_variants(r.sites, r.pvars))
        phased_reads = _lib.format_phased_reads(r.preads, ctg, qoff, qnames)
        put("phased_reads", phased_reads)
        if tigs is not None:                                 # K6: consensus of every (block, phase) pile
            _mkdirs(os.path.join(base, "cns"))
            put("cns/phased_blocks.fa", tigs.fasta(c, ctg))
        if write_sam:                                        # the blasr task's artefacts (unzip.py:86-91): sorted BAM + index
            flags = (summ["strand"][idx] * 16).astype(np.int32)
            bam, bai = _lib.format_bam(aln, ctg, len(ref), flags)
            put("blasr/%s_sorted.bam" % ctg, bam)
            put("blasr/%s_sorted.bam.bai" % ctg, bai)
        if read_map_dir is not None:                         # fc_phasing_readmap.py (unzip.py:126)
            def slurp(p):
                with open(p, "rb") as f:
                    return f.read()
            recs, text = _lib.readmap(phased_reads, slurp(os.path.join(read_map_dir, "dump_rawread_ids", "rawread_ids")),
                                      slurp(os.path.join(read_map_dir, "dump_pread_ids", "pread_ids")),
                                      slurp(os.path.join(read_map_dir, "pread_to_contigs")), ctg, ctg_indices[c])
            put("rid_to_phase.%s" % ctg, text)
            all_recs.append(recs)
    if tigs is not None:
        tigs.close()
    batch.close()
    job.close()
    return np.concatenate(all_recs) if all_recs else np.zeros(0, _lib.R2P)


def run(unzip_dir, read_map_dir=None, ctg_ids=None, device=None):
    """Whole phasing section of unzip_all for this rank (one process per GPU under torch.distributed.run)."""
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
    eng = _lib.Engine(int(os.environ.get("LOCAL_RANK", "0")) if device is None else device)
    local = np.zeros(0, _lib.R2P)
    if mine:
        jobs = load_contig_jobs(unzip_dir, [ctg_ids[i] for i in mine])
        # contig indices are global so that the gathered records sort like the reference's file list
        local = phase_contigs(eng, jobs, unzip_dir, read_map_dir, ctg_indices=mine)
    eng.close()
    allr = fdist.allgather_r2p(local)
    if rank == 0 and read_map_dir is not None:
        out_dir = os.path.join(unzip_dir, "1-hasm", "rid-to-phase-all")
        os.makedirs(out_dir, exist_ok=True)
        with open(os.path.join(out_dir, "rid_to_phase.all"), "wb") as f:
            f.write(fdist.format_rid_to_phase_all(allr, ctg_ids))
    return allr
