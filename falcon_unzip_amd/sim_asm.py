"""Synthetic inputs for the haplotig layout (graphs_to_h_tigs.py): a diploid locus with heterozygous "bubbles", error-free
p-reads tiling both haplotypes, the primary assembly's string graph (one linear contig through haplotype A), the phased
assembly's string graph (haplotype-A edges again plus the haplotype-B paths through every bubble, hooked to the unphased
flank reads), the p-read FASTA and the rid_to_phase map.  Test / golden input generator only.

String-graph conventions (FALCON): nodes are read ends '<rid>:B' / '<rid>:E'; for reads X = [x0, x1) then Y = [y0, y1) on the
same strand (x0 < y0, x1 < y1, overlapping) the edge X:E -> Y:E spells Y[x1 - y0 : len(Y)] and its dual Y:B -> X:B spells the
reverse complement read off X from index y0 - x0 down to (not including) 0; a label (seq_id, s, t) with s > t means that."""
from __future__ import annotations

import os

import numpy as np

from . import sim


def make_locus(rng, segments, read_len=8000, step=3000, het_rate=1.0 / 300, rid_base=0, pread_sub=0.0, err_rng=None):
    """segments: list of ('hom' | 'het' | 'tie', length).  -> dict with haplotypes, reads, phases, graph edge lists.
    pread_sub > 0: every p-read carries substitution errors at that rate (drawn from err_rng, a generator of its own: the locus is the same with and without them;
    substitutions only -- the string graph's edge coordinates are read offsets, an indel would move them), so the tigs the layout spells carry their reads' errors.
    'tie': a bubble like 'het' whose reads were NOT phased (no block for them in rid_to_phase): both branches score alike, the layout has to
    break a tie between them"""
    L = sum(n for _, n in segments)
    hapA = rng.integers(0, 4, size=L, dtype=np.uint8)
    hapB = hapA.copy()
    bubbles, unphased, pos = [], set(), 0
    for kind, n in segments:
        if kind in ("het", "tie"):
            k = max(3, int(n * het_rate))
            p = np.sort(rng.choice(np.arange(pos + 200, pos + n - 200), size=k, replace=False))
            hapB[p] = (hapA[p] + rng.integers(1, 4, size=k, dtype=np.uint8)) & 3
            if kind == "tie":
                unphased.add(len(bubbles))
            bubbles.append((pos, pos + n))
        pos += n
    reads = []          # (rid, hap, start, end)

    def block_of(mid):
        for b, (u0, u1) in enumerate(bubbles):
            if u0 <= mid < u1:
                return -1 if b in unphased else b + 1
        return -1
    rid = rid_base
    a_reads = []
    for s in range(0, L - read_len + 1, step):
        a_reads.append(("%09d" % rid, 0, s, s + read_len))
        rid += 1
    if a_reads[-1][3] < L:
        a_reads.append(("%09d" % rid, 0, L - read_len, L))
        rid += 1
    b_paths = []
    for bi, (u0, u1) in enumerate(bubbles):
        starts, s = [], u0 - 2 * step + step // 2
        while s + read_len <= min(L, u1 + 2 * step):
            starts.append(s)
            s += step
        if bi in unphased:
            # one read fewer over the same span: the branch then has as many edges between its hooks as the primary path beside it -- with
            # unphased reads on both (every edge scores 1) the two routes cost the same and the layout must break the tie
            m = len(starts) - 1
            assert m >= 3 and (starts[-1] - starts[0]) % (m - 1) == 0, "tie bubble: pick a length that lets the branch lose one read evenly"
            starts = [starts[0] + i * (starts[-1] - starts[0]) // (m - 1) for i in range(m)]
        path = []
        for s in starts:
            path.append(("%09d" % rid, 1, s, s + read_len))
            rid += 1
        b_paths.append(path)
    phase = {}
    for r in a_reads:
        b = block_of((r[2] + r[3]) // 2)
        phase[r[0]] = (b, 0) if b > 0 else (-1, 0)
    for k, path in enumerate(b_paths):
        for r in path:
            mid = (r[2] + r[3]) // 2
            phase[r[0]] = (k + 1, 1) if (bubbles[k][0] <= mid < bubbles[k][1] and k not in unphased) else (-1, 0)
    seqs = {}
    for r in a_reads:
        seqs[r[0]] = sim.codes_to_str(hapA[r[2]:r[3]])
    for path in b_paths:
        for r in path:
            seqs[r[0]] = sim.codes_to_str(hapB[r[2]:r[3]])

    if pread_sub > 0:
        lut = np.zeros(256, np.uint8)
        lut[[65, 67, 71, 84]] = [0, 1, 2, 3]
        for rid_ in sorted(seqs):
            codes = lut[np.frombuffer(seqs[rid_].encode(), np.uint8)]
            hit = np.flatnonzero(err_rng.random(codes.size) < pread_sub)
            codes[hit] = (codes[hit] + err_rng.integers(1, 4, size=hit.size, dtype=np.uint8)) & 3
            seqs[rid_] = sim.codes_to_str(codes)

    def edge_pair(x, y):
        """forward edge X:E -> Y:E and its dual Y:B -> X:B"""
        ovl = x[3] - y[2]
        assert x[2] < y[2] and x[3] < y[3] and ovl > 0
        ly = y[3] - y[2]
        return [("%s:E" % x[0], "%s:E" % y[0], y[0], ovl, ly, ovl, 99.9), ("%s:B" % y[0], "%s:B" % x[0], x[0], y[2] - x[2], 0, ovl, 99.9)]
    p_edges, h_edges = [], []
    for x, y in zip(a_reads[:-1], a_reads[1:]):
        p_edges += [e + ("G",) for e in edge_pair(x, y)]
    for x, y in zip(a_reads[:-2], a_reads[2:]):                      # transitive edges, reduced away: type 'TR'
        if x[3] > y[2]:
            p_edges += [e + ("TR",) for e in edge_pair(x, y)]
    h_edges += [e for e in p_edges if e[-1] == "G"]
    for path in b_paths:
        # left hook: the last haplotype-A read that starts before the path's first read and ends inside it; right hook likewise
        first, last = path[0], path[-1]
        left = [r for r in a_reads if r[2] < first[2] and first[2] < r[3] < first[3] and phase[r[0]][0] == -1]
        right = [r for r in a_reads if last[2] < r[2] < last[3] and r[3] > last[3] and phase[r[0]][0] == -1]
        chain = ([left[-1]] if left else []) + path + ([right[0]] if right else [])
        for x, y in zip(chain[:-1], chain[1:]):
            h_edges += [e + ("G",) for e in edge_pair(x, y)]
    return {"L": L, "hapA": hapA, "hapB": hapB, "bubbles": bubbles, "a_reads": a_reads, "b_paths": b_paths, "phase": phase, "seqs": seqs,
            "p_edges": p_edges, "h_edges": h_edges, "next_rid": rid}


def write_assembly(asm_dir, loci_edges, contigs):
    """sg_edges_list / utg_data / ctg_paths of one assembly directory.  contigs: list of (ctg_id, [node, ...] path)"""
    os.makedirs(asm_dir, exist_ok=True)
    with open(os.path.join(asm_dir, "sg_edges_list"), "w") as f:
        for edges in loci_edges:
            for v, w, sid, s, t, score, idt, typ in edges:
                f.write("%s %s %s %d %d %d %.2f %s\n" % (v, w, sid, s, t, score, idt, typ))
    with open(os.path.join(asm_dir, "utg_data"), "w") as fu, open(os.path.join(asm_dir, "ctg_paths"), "w") as fc:
        for ctg_id, nodes in contigs:
            if len(nodes) < 3:
                continue
            # two simple unitigs per contig (the second starts where the first ends) so that multi-unitig paths are exercised
            cut = len(nodes) // 2
            utgs = [nodes[:cut + 1], nodes[cut:]] if cut >= 2 and len(nodes) - cut >= 3 else [nodes]
            for u in utgs:
                fu.write("%s %s %s simple %d %d %s\n" % (u[0], u[1], u[-1], 1000 * len(u), 100 * len(u), "~".join(u)))
            fc.write("%s ctg_linear %s~%s %s %d %d %s\n" % (ctg_id, nodes[0], nodes[1], nodes[-1], 1000 * len(nodes), 100 * len(nodes),
                                                          "|".join("%s~%s~%s" % (u[0], u[1], u[-1]) for u in utgs)))


def make_case(root, seed, layouts, pread_sub=0.0):
    """layouts: list of (ctg_id, segments).  Writes <root>/{2-asm-falcon,1-hasm}/..., preads4falcon.fasta, rid_to_phase.all;
    returns the loci (for truth checks).  pread_sub: substitution errors in the p-reads (make_locus)."""
    rng = np.random.Generator(np.random.PCG64(seed))
    err_rng = np.random.Generator(np.random.PCG64(seed + 7919))
    loci, rid = [], 0
    for ctg_id, segments in layouts:
        loc = make_locus(rng, segments, rid_base=rid, pread_sub=pread_sub, err_rng=err_rng)
        rid = loc["next_rid"]
        loc["ctg_id"] = ctg_id
        loci.append(loc)
    write_assembly(os.path.join(root, "2-asm-falcon"), [l["p_edges"] for l in loci],
                   [(l["ctg_id"], ["%s:E" % r[0] for r in l["a_reads"]]) for l in loci] +
                   [(l["ctg_id"][:-1] + "R", ["%s:B" % r[0] for r in reversed(l["a_reads"])]) for l in loci])
    h_contigs = []
    for l in loci:
        h_contigs.append((l["ctg_id"], ["%s:E" % r[0] for r in l["a_reads"]]))
    write_assembly(os.path.join(root, "1-hasm"), [l["h_edges"] for l in loci], h_contigs)
    with open(os.path.join(root, "preads4falcon.fasta"), "w") as f:
        for l in loci:
            for rid_, s in l["seqs"].items():
                f.write(">%s\n%s\n" % (rid_, s.lower() if int(rid_) % 5 == 0 else s))       # the layout upper-cases (graphs_to_h_tigs.py:36)
    with open(os.path.join(root, "rid_to_phase.all"), "w") as f:
        for l in loci:
            if not l["bubbles"]:
                continue                                    # a contig without phased reads has no rows: the layout skips it (:668-669)
            for rid_ in sorted(l["phase"]):
                f.write("%s %s %d %d\n" % (rid_, l["ctg_id"], l["phase"][rid_][0], l["phase"][rid_][1]))
    return loci
