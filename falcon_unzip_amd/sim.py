"""Synthetic diploid + PacBio-CLR-like read simulator (numpy PCG64, deterministic).

This is the input generator SURVEY.md section 8(d) prescribes for every config:
hap0 = iid uniform ACGT (it is also the primary contig the reads are aligned to),
hap1 = hap0 with SNPs at rate 1/500; reads of fixed template length R start
uniformly, pick a haplotype with p=1/2, and go through a CLR error model
(sub / ins / del).  The simulator knows the true alignment of every read, so it can
emit the coordinate-sorted SAM text that `samtools view <bam> <ctg>` would hand to
`make_het_call` (reference: falcon_unzip/phasing.py:27,42-59) without any aligner.

Nothing here is on the product path: tests, the golden generator and bench.py use it
to make inputs.
"""
from __future__ import annotations

import numpy as np

ACGT = np.frombuffer(b"ACGT", dtype=np.uint8)
_COMP = np.array([3, 2, 1, 0], dtype=np.uint8)  # A<->T, C<->G on 2-bit codes

# op codes used internally (also the BAM op numbering the C-ABI uses)
OP_M, OP_I, OP_D, OP_N, OP_S, OP_H, OP_P, OP_EQ, OP_X = range(9)
OP_CHARS = "MIDNSHP=X"


def rng_for(config: int, contig_idx: int = 0) -> np.random.Generator:
    """seed = 20260000 + 1000*config + contig_idx."""
    return np.random.Generator(np.random.PCG64(20260000 + 1000 * config + contig_idx))


def make_diploid(L: int, rng: np.random.Generator, het_rate: float = 1.0 / 500):
    """Return (hap0 codes, hap1 codes, sorted het positions)."""
    hap0 = rng.integers(0, 4, size=L, dtype=np.uint8)
    n_het = int(round(L * het_rate))
    pos = np.sort(rng.choice(L, size=n_het, replace=False))
    hap1 = hap0.copy()
    hap1[pos] = (hap0[pos] + rng.integers(1, 4, size=n_het, dtype=np.uint8)) & 3
    return hap0, hap1, pos


def codes_to_str(codes: np.ndarray) -> str:
    return ACGT[codes].tobytes().decode("ascii")


def revcomp_codes(codes: np.ndarray) -> np.ndarray:
    return _COMP[codes[::-1]]


class SimRead:
    __slots__ = ("name", "hap", "start", "strand", "seq", "ops", "lens", "clip5", "clip3")

    def __init__(self, name, hap, start, strand, seq, ops, lens, clip5, clip3):
        self.name = name      # QNAME
        self.hap = hap        # 0/1 haplotype of origin
        self.start = start    # 0-based contig position of the first aligned column
        self.strand = strand  # 0 forward, 1 reverse (the *sequenced* read is revcomp)
        self.seq = seq        # uint8 codes, on the CONTIG strand, incl. soft clips
        self.ops = ops        # uint8 op codes (run-length encoded CIGAR)
        self.lens = lens      # int32 run lengths
        self.clip5 = clip5
        self.clip3 = clip3

    def cigar(self) -> str:
        return "".join("%d%s" % (l, OP_CHARS[o]) for o, l in zip(self.ops, self.lens))

    def raw_seq_codes(self) -> np.ndarray:
        """Sequence as it came off the instrument (what <ctg>_reads.fa holds)."""
        return revcomp_codes(self.seq) if self.strand else self.seq

    def ref_span(self) -> int:
        m = (self.ops == OP_EQ) | (self.ops == OP_X) | (self.ops == OP_D) | (self.ops == OP_M)
        return int(self.lens[m].sum())


def _rle(op_seq: np.ndarray):
    if op_seq.size == 0:
        return np.zeros(0, np.uint8), np.zeros(0, np.int32)
    brk = np.flatnonzero(op_seq[1:] != op_seq[:-1]) + 1
    starts = np.concatenate(([0], brk))
    ends = np.concatenate((brk, [op_seq.size]))
    return op_seq[starts].astype(np.uint8), (ends - starts).astype(np.int32)


def simulate_read(contig, hap, start, R, rng, sub=0.01, ins=0.08, dele=0.04,
                  clip5=0, clip3=0, hp_bias=1.0):
    """One read of template length R from `hap` at `start`; CIGAR is vs `contig`.

    Returns (seq codes on contig strand incl. clips, ops, lens).
    """
    t = hap[start:start + R]
    c = contig[start:start + R]
    R = t.size
    u = rng.random(R)
    w = rng.random(R)
    if hp_bias != 1.0:
        # homopolymer-biased indels: inside a run (base equal to its predecessor) deletions and insertions are hp_bias times as likely,
        # and an inserted base repeats the run's base -- the non-iid error shape of single-molecule reads
        inrun = np.zeros(R, bool)
        inrun[1:] = t[1:] == t[:-1]
        dele_k = np.where(inrun, min(0.45, dele * hp_bias), dele)
        ins_k = np.where(inrun, min(0.45, ins * hp_bias), ins)
    else:
        inrun, dele_k, ins_k = None, dele, ins
    is_del = u < dele_k
    is_sub = (u >= dele_k) & (u < dele_k + sub)
    n_ins = (w < ins_k).astype(np.int64)
    # keep the alignment anchored: first/last template bases are emitted, no leading insert
    is_del[0] = is_del[-1] = False
    n_ins[0] = 0
    base = t.copy()
    nsub = int(is_sub.sum())
    if nsub:
        base[is_sub] = (t[is_sub] + rng.integers(1, 4, size=nsub, dtype=np.uint8)) & 3
    emit = (~is_del).astype(np.int64)
    per = n_ins + 1                       # path steps contributed by template base k
    tot_steps = int(per.sum())
    step_off = np.cumsum(per) - per       # first step of template base k
    op_seq = np.full(tot_steps, OP_I, dtype=np.uint8)
    last = step_off + n_ins               # the M/D step of base k
    op_seq[last] = np.where(is_del, OP_D, np.where(base == c, OP_EQ, OP_X))
    # read bases: inserted bases random, emitted bases = base
    q_per = n_ins + emit
    qlen = int(q_per.sum())
    q_off = np.cumsum(q_per) - q_per
    seq = rng.integers(0, 4, size=qlen, dtype=np.uint8)   # fills the insert slots
    if inrun is not None:
        sel = inrun & (n_ins > 0)
        seq[q_off[sel]] = t[sel]
    seq[(q_off + n_ins)[~is_del]] = base[~is_del]
    ops, lens = _rle(op_seq)
    if clip5:
        seq = np.concatenate((rng.integers(0, 4, size=clip5, dtype=np.uint8), seq))
        ops = np.concatenate(([OP_S], ops)).astype(np.uint8)
        lens = np.concatenate(([clip5], lens)).astype(np.int32)
    if clip3:
        seq = np.concatenate((seq, rng.integers(0, 4, size=clip3, dtype=np.uint8)))
        ops = np.concatenate((ops, [OP_S])).astype(np.uint8)
        lens = np.concatenate((lens, [clip3])).astype(np.int32)
    return seq, ops, lens


def simulate_reads(hap0, hap1, n_reads, R, rng, sub=0.01, ins=0.08, dele=0.04,
                   strand_mix=0.0, clip_frac=0.0, clip_max=300, name_prefix="sim"):
    """`n_reads` reads vs contig = hap0.  Returned list is in simulation order."""
    L = hap0.size
    reads = []
    starts = rng.integers(0, max(1, L - R + 1), size=n_reads)
    haps = rng.integers(0, 2, size=n_reads)
    strands = (rng.random(n_reads) < strand_mix).astype(np.int64)
    for i in range(n_reads):
        c5 = c3 = 0
        if clip_frac > 0 and rng.random() < clip_frac:
            c5 = int(rng.integers(0, clip_max))
            c3 = int(rng.integers(0, clip_max))
        hap = hap1 if haps[i] else hap0
        seq, ops, lens = simulate_read(hap0, hap, int(starts[i]), R, rng, sub, ins, dele, c5, c3)
        name = "%s/%d/0_%d" % (name_prefix, i, seq.size)
        reads.append(SimRead(name, int(haps[i]), int(starts[i]), int(strands[i]), seq, ops, lens, c5, c3))
    return reads


def sam_lines(reads, ctg_id, header=True, L=None):
    """Coordinate-sorted SAM text lines (sorted by (POS, simulation index))."""
    out = []
    if header:
        out.append("@HD\tVN:1.5\tSO:coordinate")
        if L is not None:
            out.append("@SQ\tSN:%s\tLN:%d" % (ctg_id, L))
    order = sorted(range(len(reads)), key=lambda i: (reads[i].start, i))
    for i in order:
        r = reads[i]
        flag = 16 if r.strand else 0
        out.append("\t".join((r.name, str(flag), ctg_id, str(r.start + 1), "254", r.cigar(),
                              "*", "0", "0", codes_to_str(r.seq), "*")))
    return out


def write_fasta(path, records, width=0):
    with open(path, "w") as f:
        for name, seq in records:
            f.write(">%s\n" % name)
            if width:
                for i in range(0, len(seq), width):
                    f.write(seq[i:i + width] + "\n")
            else:
                f.write(seq + "\n")


def simulate_raw_reads_bulk(hap0, hap1, n_reads, R, rng, lo=0, hi=None, sub=0.01, ins=0.08, dele=0.04, strand_mix=0.5):
    """Vectorised generator for benchmarks: raw (as-sequenced) reads only, no truth CIGARs.

    Same error model as simulate_read (per template base: optional 1-base insert, then match / sub / del),
    with 16-bit thresholds; read starts are uniform in [lo, hi - R].
    Returns (codes uint8 [total], off int64 [n+1], start, hap, strand)."""
    L = hap0.size
    hi = L if hi is None else hi
    starts = rng.integers(lo, max(lo + 1, hi - R + 1), size=n_reads)
    haps = rng.integers(0, 2, size=n_reads)
    strands = (rng.random(n_reads) < strand_mix)
    T = np.empty((n_reads, R), dtype=np.uint8)
    for i in range(n_reads):
        T[i] = (hap1 if haps[i] else hap0)[starts[i]:starts[i] + R]
    u = rng.integers(0, 65536, size=(n_reads, R), dtype=np.uint16)
    t_del, t_sub, t_ins = int(dele * 65536), int((dele + sub) * 65536), int(ins * 65536)
    is_del = u < t_del
    is_sub = (u >= t_del) & (u < t_sub)
    v = rng.integers(0, 65536, size=(n_reads, R), dtype=np.uint16)
    has_ins = v < t_ins
    is_del[:, 0] = False
    is_del[:, -1] = False
    has_ins[:, 0] = False
    rnd = rng.integers(0, 256, size=(n_reads, R), dtype=np.uint8)
    T[is_sub] = (T[is_sub] + 1 + (rnd[is_sub] % 3)) & 3
    E = np.empty((n_reads, R, 2), dtype=np.uint8)
    E[:, :, 0] = (rnd >> 4) & 3            # inserted base
    E[:, :, 1] = T
    keep = np.empty((n_reads, R, 2), dtype=bool)
    keep[:, :, 0] = has_ins
    keep[:, :, 1] = ~is_del
    lens = keep.reshape(n_reads, -1).sum(axis=1)
    codes = E[keep]
    off = np.zeros(n_reads + 1, np.int64)
    off[1:] = np.cumsum(lens)
    for i in np.flatnonzero(strands):
        a, b = off[i], off[i + 1]
        codes[a:b] = _COMP[codes[a:b][::-1]]
    return codes, off, starts, haps, strands.astype(np.int64)


def lognormal_lengths(n, rng, median=12000, sigma=0.55, lo=3000, hi=60000):
    """CLR-like subread lengths: log-normal around `median`, redrawn (clipped after 8 tries) into [lo, hi]"""
    x = np.exp(rng.normal(np.log(median), sigma, size=n))
    for _ in range(8):
        bad = (x < lo) | (x > hi)
        if not bad.any():
            break
        x[bad] = np.exp(rng.normal(np.log(median), sigma, size=int(bad.sum())))
    return np.clip(x, lo, hi).astype(np.int64)


def simulate_raw_reads_shaped(hap0, hap1, n_reads, rng, lens=None, lo=0, hi=None, length_model=None, batch=128, **kw):
    """_shaped_flat in batches of `batch` reads (the flat arrays stay cache-sized); same return tuple.  with_truth=True appends the
    template index of every emitted base, in TEMPLATE order (= the oriented read K1 reports q_start / q_end on): base q of oriented read i
    sits at contig position start[i] + truth[off[i] + q]."""
    L = hap0.size
    hi = L if hi is None else hi
    if lens is None:
        lens = lognormal_lengths(n_reads, rng, **(length_model or {}))
    lens = np.asarray(lens, dtype=np.int64)
    parts = [_shaped_flat(hap0, hap1, min(batch, n_reads - b), rng, lens[b:b + batch], lo, hi, **kw) for b in range(0, n_reads, batch)]
    if not parts:
        z = np.zeros(0, np.int64)
        return np.zeros(0, np.uint8), np.zeros(1, np.int64), z, z, z, z, np.zeros(0)
    off = np.zeros(n_reads + 1, np.int64)
    off[1:] = np.cumsum(np.concatenate([np.diff(p[1]) for p in parts]))
    return (np.concatenate([p[0] for p in parts]), off) + tuple(np.concatenate([p[k] for p in parts]) for k in range(2, len(parts[0])))


def _shaped_flat(hap0, hap1, n_reads, rng, lens, lo, hi, sub=0.01, ins=0.08, dele=0.04, strand_mix=0.5,
                 burst_rate=1.0 / 10000, burst_len=(300, 1000), burst_err=(0.06, 0.16, 0.08), head_burst=0.25, with_truth=False):
    """Reads of REAL shape for K1: template lengths from `lens` (default: lognormal_lengths -- median 12 kb, 3-60 kb) and
    bursty errors: stretches of burst_len template bases at burst_err = (sub, ins, del) -- 30 % error -- starting at rate burst_rate per
    base, and with probability head_burst one of them right at the read's head (the as-sequenced 5' end, whichever strand).  Outside the
    bursts the iid CLR model of simulate_raw_reads_bulk.  Flat vectorisation over all template bases.
    Returns (codes uint8 [total], off int64 [n+1], start, hap, strand, template length, burst fraction per read)."""
    lens = np.minimum(np.asarray(lens, dtype=np.int64), hi - lo)
    starts = (lo + rng.random(n_reads) * (hi - lo - lens + 1)).astype(np.int64)
    haps = rng.integers(0, 2, size=n_reads)
    strands = rng.random(n_reads) < strand_mix
    toff = np.zeros(n_reads + 1, np.int64)
    toff[1:] = np.cumsum(lens)
    total = int(toff[-1])
    rid = np.repeat(np.arange(n_reads), lens)
    within = np.arange(total, dtype=np.int64) - toff[rid]
    gpos = starts[rid] + within
    T = np.where(haps[rid] == 1, hap1[gpos], hap0[gpos]).astype(np.uint8)
    # burst intervals -> per-base flag (difference array)
    diff = np.zeros(total + 1, np.int32)
    n_b = rng.poisson(lens * burst_rate)
    for r in np.flatnonzero(n_b):
        for _ in range(int(n_b[r])):
            bl = int(rng.integers(burst_len[0], burst_len[1] + 1))
            b0 = int(rng.integers(0, max(1, lens[r] - bl)))
            diff[toff[r] + b0] += 1
            diff[min(toff[r] + b0 + bl, toff[r + 1])] -= 1
    for r in np.flatnonzero(rng.random(n_reads) < head_burst):
        bl = int(min(rng.integers(burst_len[0], burst_len[1] + 1), lens[r] // 2))
        if strands[r]:                                   # sequenced 5' end = the template's far end
            diff[toff[r + 1] - bl] += 1
            diff[toff[r + 1]] -= 1
        else:
            diff[toff[r]] += 1
            diff[toff[r] + bl] -= 1
    inb = np.cumsum(diff[:-1]) > 0
    t_del = np.where(inb, int(burst_err[2] * 65536), int(dele * 65536)).astype(np.uint32)
    t_sub = t_del + np.where(inb, int(burst_err[0] * 65536), int(sub * 65536)).astype(np.uint32)
    t_ins = np.where(inb, int(burst_err[1] * 65536), int(ins * 65536)).astype(np.uint32)
    u = rng.integers(0, 65536, size=total, dtype=np.uint16).astype(np.uint32)
    v = rng.integers(0, 65536, size=total, dtype=np.uint16).astype(np.uint32)
    rnd = rng.integers(0, 256, size=total, dtype=np.uint8)
    is_del = u < t_del
    is_sub = (u >= t_del) & (u < t_sub)
    has_ins = v < t_ins
    first = toff[:-1]
    last = toff[1:] - 1
    is_del[first] = False
    is_del[last] = False
    has_ins[first] = False
    T[is_sub] = (T[is_sub] + 1 + (rnd[is_sub] % 3)) & 3
    E = np.empty((total, 2), dtype=np.uint8)
    E[:, 0] = (rnd >> 4) & 3
    E[:, 1] = T
    keep = np.empty((total, 2), dtype=bool)
    keep[:, 0] = has_ins
    keep[:, 1] = ~is_del
    per_base = keep.sum(axis=1)
    rl = np.add.reduceat(per_base, toff[:-1]) if n_reads else np.zeros(0, np.int64)
    codes = E[keep]
    off = np.zeros(n_reads + 1, np.int64)
    off[1:] = np.cumsum(rl)
    for i in np.flatnonzero(strands):
        a, b = off[i], off[i + 1]
        codes[a:b] = _COMP[codes[a:b][::-1]]
    bfrac = np.add.reduceat(inb.astype(np.int64), toff[:-1]) / np.maximum(lens, 1) if n_reads else np.zeros(0)
    if with_truth:
        return codes, off, starts, haps, strands.astype(np.int64), lens, bfrac, np.repeat(within, per_base).astype(np.int32)
    return codes, off, starts, haps, strands.astype(np.int64), lens, bfrac


def _diverge(codes, rng, div, indel_frac=0.2):
    """A diverged copy of `codes`: substitutions at rate div*(1-indel_frac), 1-base indels at rate div*indel_frac."""
    n = codes.size
    out = codes.copy()
    u = rng.random(n)
    sub = u < div * (1.0 - indel_frac)
    k = int(sub.sum())
    if k:
        out[sub] = (out[sub] + rng.integers(1, 4, size=k, dtype=np.uint8)) & 3
    v = rng.random(n)
    dele = v < div * indel_frac * 0.5
    ins = (v >= div * indel_frac * 0.5) & (v < div * indel_frac)
    keep = ~dele
    pieces = np.repeat(keep.astype(np.int64) + ins.astype(np.int64), 1)
    res = np.empty(int(pieces.sum()), np.uint8)
    off = np.cumsum(pieces) - pieces
    res[off[keep]] = out[keep]
    ni = int(ins.sum())
    if ni:
        res[(off + keep.astype(np.int64))[ins]] = rng.integers(0, 4, size=ni, dtype=np.uint8)
    return res


def make_repeat_diploid(L, rng, het_rate=1.0 / 500, n_families=6, copies=(2, 4), fam_len=(2000, 6000),
                        n_tandem=6, unit_len=(80, 500), tandem_copies=(4, 10), divergence=(0.01, 0.05)):
    """Diploid with repeats: interspersed families of 2-6 kb copies at 95-99 % identity and
    tandem arrays (unit 80-500 bp, 4-10 copies, same divergence), written over an iid background; hap1 = hap0 + SNPs.
    Returns (hap0, hap1, het positions, list of (start, end, kind) repeat intervals)."""
    hap0 = rng.integers(0, 4, size=L, dtype=np.uint8)
    spans = []

    def place(seg):
        for _ in range(200):
            s = int(rng.integers(0, L - seg.size))
            if all(s + seg.size <= a or s >= b for a, b, _k in spans):
                return s
        return None

    for _ in range(n_families):
        flen = int(rng.integers(fam_len[0], fam_len[1] + 1))
        master = rng.integers(0, 4, size=flen, dtype=np.uint8)
        for _c in range(int(rng.integers(copies[0], copies[1] + 1))):
            seg = _diverge(master, rng, float(rng.uniform(*divergence)))
            if rng.random() < 0.3:
                seg = revcomp_codes(seg)
            s = place(seg)
            if s is None:
                continue
            hap0[s:s + seg.size] = seg
            spans.append((s, s + seg.size, "interspersed"))
    for _ in range(n_tandem):
        ulen = int(rng.integers(unit_len[0], unit_len[1] + 1))
        unit = rng.integers(0, 4, size=ulen, dtype=np.uint8)
        arr = np.concatenate([_diverge(unit, rng, float(rng.uniform(*divergence))) for _c in range(int(rng.integers(tandem_copies[0], tandem_copies[1] + 1)))])
        s = place(arr)
        if s is None:
            continue
        hap0[s:s + arr.size] = arr
        spans.append((s, s + arr.size, "tandem"))
    n_het = int(round(L * het_rate))
    pos = np.sort(rng.choice(L, size=n_het, replace=False))
    hap1 = hap0.copy()
    hap1[pos] = (hap0[pos] + rng.integers(1, 4, size=n_het, dtype=np.uint8)) & 3
    return hap0, hap1, pos, sorted(spans)


def make_diploid_indels(L, rng, het_rate=1.0 / 400, indel_frac=0.25, indel_len=(2, 5)):
    """hap0 iid; hap1 = hap0 with heterozygous SNPs and small insertions / deletions (indel_len bases).  Returns
    (hap0, hap1, map01, events): map01[p] = index in hap1 of hap0 position p (length L + 1), events = [(pos, kind, length)]."""
    hap0 = rng.integers(0, 4, size=L, dtype=np.uint8)
    n_het = int(round(L * het_rate))
    pos = np.sort(rng.choice(np.arange(50, L - 50), size=n_het, replace=False))
    pieces, map01, events, prev, shift = [], np.zeros(L + 1, np.int64), [], 0, 0
    for p in pos:
        p = int(p)
        if p < prev:
            continue
        pieces.append(hap0[prev:p])
        map01[prev:p] = np.arange(prev, p) + shift
        r = rng.random()
        if r < indel_frac / 2:                              # insertion after p - 1 (kept away from homopolymer ambiguity: first inserted base differs from both neighbours)
            n = int(rng.integers(indel_len[0], indel_len[1] + 1))
            ins = rng.integers(0, 4, size=n, dtype=np.uint8)
            while ins[0] == hap0[p - 1] or ins[-1] == hap0[p]:
                ins = rng.integers(0, 4, size=n, dtype=np.uint8)
            pieces.append(ins)
            shift += n
            events.append((p, "ins", n))
            prev = p
        elif r < indel_frac:                                # deletion of hap0[p : p + n]
            n = int(rng.integers(indel_len[0], indel_len[1] + 1))
            map01[p:p + n] = p + shift
            shift -= n
            events.append((p, "del", n))
            prev = p + n
        else:
            pieces.append(np.array([(hap0[p] + rng.integers(1, 4)) & 3], dtype=np.uint8))
            map01[p] = p + shift
            events.append((p, "snp", 1))
            prev = p + 1
    pieces.append(hap0[prev:])
    map01[prev:L] = np.arange(prev, L) + shift
    map01[L] = L + shift
    return hap0, np.concatenate(pieces).astype(np.uint8), map01, events


def simulate_raw_reads_from(hap, n_reads, R, rng, strand_mix=0.5, hp_bias=1.0, name_prefix="sim"):
    """raw reads (as sequenced) of template length R drawn from one haplotype sequence; -> [(name, bytes, start, strand)]"""
    out = []
    starts = rng.integers(0, max(1, hap.size - R + 1), size=n_reads)
    strands = rng.random(n_reads) < strand_mix
    for i in range(n_reads):
        seq, _, _ = simulate_read(hap, hap, int(starts[i]), R, rng, hp_bias=hp_bias)
        if strands[i]:
            seq = revcomp_codes(seq)
        out.append(("%s/%d/0_%d" % (name_prefix, i, seq.size), ACGT[seq].tobytes(), int(starts[i]), int(strands[i])))
    return out
