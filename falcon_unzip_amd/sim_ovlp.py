"""Synthetic `LA4Falcon -mo` overlap dumps + `rid_to_phase.all` maps for the overlap filter
(ovlp_filter_with_phase.py reads 13 whitespace-separated columns per overlap:
 q_id t_id -len idt q_strand q_s q_e q_l t_strand t_s t_e t_l tag; falcon_unzip/ovlp_filter_with_phase.py:60-106).

Reads are intervals on contigs with a (block, phase) label; overlaps are derived from interval geometry, so the
counts / containment / best-n logic sees realistic structure.  numpy PCG64, seeded like falcon_unzip_amd.sim.
"""
from __future__ import annotations

import numpy as np


def rng_for(case, idx=0):
    return np.random.Generator(np.random.PCG64(20270000 + 1000 * case + idx))


def make_reads(rng, n_ctg=2, reads_per_ctg=120, ctg_len=200_000, mean_len=9000, unphased_frac=0.3, n_blocks=3):
    """-> list of dicts (rid, ctg, start, end, block, phase) sorted by rid."""
    reads = []
    rid = 0
    for c in range(n_ctg):
        starts = np.sort(rng.integers(0, ctg_len - 2000, reads_per_ctg))
        for s in starts:
            ln = int(max(1500, rng.normal(mean_len, 2500)))
            e = int(min(ctg_len, s + ln))
            blk = int((s * n_blocks) // ctg_len)
            if rng.random() < unphased_frac:
                b, p = -1, 0
            else:
                b, p = blk, int(rng.integers(0, 2))
            reads.append(dict(rid=rid, ctg="%06dF" % c, start=int(s), end=e, block=b, phase=p))
            rid += 1
    return reads


def rid_phase_map_text(reads, drop_frac=0.0, rng=None):
    """rid_to_phase.all rows ('%09d ctg block phase', phasing_readmap.py:47-51); some reads can be left out."""
    out = []
    for r in reads:
        if rng is not None and drop_frac > 0 and rng.random() < drop_frac:
            continue
        out.append("%09d %s %d %d\n" % (r["rid"], r["ctg"], r["block"], r["phase"]))
    return "".join(out)


def overlap_lines(reads, rng, min_ovl=800, noise_pairs=0.02, low_idt_frac=0.05, dup_frac=0.02, odd_tag_frac=0.02):
    """LA4Falcon -mo style rows grouped by q (ascending rid), every geometric overlap >= min_ovl bases reported."""
    by_ctg = {}
    for r in reads:
        by_ctg.setdefault(r["ctg"], []).append(r)
    lines = []
    n = len(reads)
    for q in reads:
        rows = []
        for t in by_ctg[q["ctg"]]:
            if t["rid"] == q["rid"]:
                continue
            lo, hi = max(q["start"], t["start"]), min(q["end"], t["end"])
            if hi - lo < min_ovl:
                continue
            rows.append(_row(q, t, lo, hi, rng, low_idt_frac, odd_tag_frac))
            if rng.random() < dup_frac:       # a second local alignment of the same pair (tie material)
                rows.append(_row(q, t, lo, hi, rng, low_idt_frac, odd_tag_frac))
        for _ in range(int(rng.poisson(noise_pairs * 10))):   # cross-contig / repeat-induced hits
            t = reads[int(rng.integers(0, n))]
            if t["rid"] == q["rid"]:
                continue
            ql, tl = q["end"] - q["start"], t["end"] - t["start"]
            ov = int(rng.integers(500, max(501, min(ql, tl))))
            rows.append("%09d %09d %d %.2f 0 %d %d %d %d %d %d %d %s" % (
                q["rid"], t["rid"], -ov, float(rng.uniform(85, 99)), 0, ov, ql, int(rng.integers(0, 2)), tl - ov, tl, tl, "overlap"))
        order = rng.permutation(len(rows))
        lines.extend(rows[i] for i in order)
    return lines


def _row(q, t, lo, hi, rng, low_idt_frac, odd_tag_frac):
    ql, tl = q["end"] - q["start"], t["end"] - t["start"]
    q_s, q_e = lo - q["start"], hi - q["start"]
    t_s, t_e = lo - t["start"], hi - t["start"]
    # snap near-end coordinates to the ends, as the aligner's local alignments reaching an end do
    if q_s < 30:
        q_s = 0
    if ql - q_e < 30:
        q_e = ql
    if t_s < 30:
        t_s = 0
    if tl - t_e < 30:
        t_e = tl
    if q_s == 0 and q_e == ql:
        tag = "contained"
    elif t_s == 0 and t_e == tl:
        tag = "contains"
    else:
        tag = "overlap"
    if rng.random() < odd_tag_frac:
        tag = "none"
    idt = float(rng.uniform(70, 89.99)) if rng.random() < low_idt_frac else float(rng.uniform(90, 99.9))
    strand = int(rng.integers(0, 2))
    return "%09d %09d %d %.2f 0 %d %d %d %d %d %d %d %s" % (q["rid"], t["rid"], -(q_e - q_s), idt, q_s, q_e, ql, strand, t_s, t_e, tl, tag)


def split_files(lines, n_files):
    """Consecutive chunks, cut only between different q ids (one .las block per file) unless n_files < 0: then cut anywhere."""
    if n_files <= 1:
        return ["".join(l + "\n" for l in lines)]
    cuts = [0]
    target = len(lines) / n_files
    for k in range(1, n_files):
        i = int(k * target)
        while 0 < i < len(lines) and lines[i].split()[0] == lines[i - 1].split()[0]:
            i += 1
        cuts.append(min(i, len(lines)))
    cuts.append(len(lines))
    return ["".join(l + "\n" for l in lines[a:b]) for a, b in zip(cuts[:-1], cuts[1:])]
