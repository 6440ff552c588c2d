"""K1 (read->contig aligner): the HIP kernels against their scalar CPU twin (oracle/align_oracle.c),
bit-exact, plus quality against the simulator's true alignments.  Parity vs blasr is unpinned."""
import numpy as np
import pytest

from tests import oracle_lib

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def eng():
    from falcon_unzip_amd import _lib
    e = _lib.Engine(0)
    yield e
    e.close()


def _sim(seed, L, n, R, strand_mix=0.5, **kw):
    from falcon_unzip_amd import sim
    rng = np.random.Generator(np.random.PCG64(seed))
    hap0, hap1, _ = sim.make_diploid(L, rng)
    reads = sim.simulate_reads(hap0, hap1, n, R, rng, strand_mix=strand_mix, **kw)
    ctg = sim.codes_to_str(hap0).encode()
    raw = [sim.codes_to_str(r.raw_seq_codes()).encode() for r in reads]
    return ctg, reads, raw


FIELDS = ("aligned", "strand", "pos", "ref_end", "q_start", "q_end", "score", "n_cigar", "cells", "n_columns", "n_match")


@pytest.mark.parametrize("seed,L,n,R", [(21, 60000, 48, 5000), (22, 400000, 64, 15000), (23, 30000, 40, 2500)])
def test_matches_cpu_twin(eng, oracle, seed, L, n, R):
    from falcon_unzip_amd import _lib
    ctg, reads, raw = _sim(seed, L, n, R)
    # a junk read and a too-short read exercise the unaligned paths
    rng = np.random.Generator(np.random.PCG64(seed + 100))
    raw.append(bytes(rng.choice(np.frombuffer(b"ACGT", np.uint8), size=3000)))
    raw.append(b"ACGTACGTAC")
    exp, exp_cig = oracle_lib.align_reads(oracle, ctg, raw)
    job = _lib.align_job(eng, [ctg], raw)
    job.run()
    got = job.summaries()
    for f in FIELDS:
        assert np.array_equal(got[f], exp[f]), (f, np.flatnonzero(got[f] != exp[f])[:5], got[f][:4], exp[f][:4])
    assert got["aligned"][-2:].sum() == 0
    aln, idx = job.alnset(0)
    assert aln.n_rec > 0.8 * n
    for k, r in enumerate(idx):
        words = np.array([(l << 4) | o for l, o in aln.cigar_of(k)], dtype=np.uint32)
        assert np.array_equal(words, exp_cig[r]), (k, r)
    job.close()


def test_quality_vs_truth(eng):
    """Placement and extent against the simulator's truth (15 kb CLR reads, 13 % error, both strands)."""
    from falcon_unzip_amd import _lib
    ctg, reads, raw = _sim(31, 1000000, 200, 15000)
    job = _lib.align_job(eng, [ctg], raw)
    job.run()
    s = job.summaries()
    start = np.array([r.start for r in reads])
    end = np.array([r.start + r.ref_span() for r in reads])
    strand = np.array([r.strand for r in reads])
    assert s["aligned"].mean() >= 0.99
    ok = s["aligned"] == 1
    assert np.all(s["strand"][ok] == strand[ok])
    assert np.mean(np.abs(s["ref_end"][ok] - end[ok]) <= 5) >= 0.98
    assert np.mean(np.abs(s["pos"][ok] - start[ok]) <= 10) >= 0.9       # origin = seed diagonal at the read's first base
    assert np.abs(s["pos"][ok] - start[ok]).max() <= 64
    assert np.mean((s["q_end"][ok] - s["q_start"][ok]) / np.array([len(x) for x in raw])[ok]) >= 0.995
    job.close()


def test_align_then_phase_matches_oracle_chain(eng, oracle):
    """K1 -> K2..K5 without leaving the device == (K1 records as SAM text) -> oracle phasing chain."""
    from falcon_unzip_amd import _lib, sim
    rng = np.random.Generator(np.random.PCG64(41))
    L = 60000
    hap0, hap1, _ = sim.make_diploid(L, rng, het_rate=1.0 / 300)
    reads = sim.simulate_reads(hap0, hap1, 260, 8000, rng, strand_mix=0.5)
    ctg = sim.codes_to_str(hap0).encode()
    raw = [sim.codes_to_str(r.raw_seq_codes()).encode() for r in reads]
    names = [r.name for r in reads]
    job = _lib.align_job(eng, [ctg], raw)
    job.run()
    aln, idx = job.alnset(0, names)
    sam = _lib.format_sam(aln, "ctgA")
    exp = oracle.phase_all(sam, ctg, "ctgA")
    b = job.to_batch()
    b.run(_lib.STAGE_ALL)
    r = b.result(0)
    off, nm = aln.qname_table()
    assert _lib.format_variant_pos(r.sites) == exp["variant_pos"]
    assert _lib.format_variant_map(r.sites, r.vmap_qid) == exp["variant_map"]
    assert _lib.format_atable(r.sites, r.arows) == exp["atable"]
    assert _lib.format_phased_variants(r.sites, r.pvars) == exp["phased_variants"]
    assert _lib.format_phased_reads(r.preads, "ctgA", off, nm) == exp["phased_reads"]
    assert len(r.sites) > 50 and len(r.preads) > 150
    # phasing accuracy vs the simulator's haplotypes: reads of one (block, phase) come from one haplotype
    hap = {rd.name: rd.hap for rd in reads}
    qn = aln.qnames()
    agree = 0
    for blk in np.unique(r.preads["block"]):
        m = r.preads["block"] == blk
        h = np.array([hap[qn[q]] for q in r.preads["q_id"][m]])
        ph = r.preads["phase"][m]
        agree += max(np.sum(h == ph), np.sum(h != ph))
    assert agree >= 0.97 * len(r.preads)
    b.close()
    job.close()


def test_edge_cases_match_twin(eng, oracle):
    """Overhangs at both contig ends, a read longer than the contig, lower-case / non-ACGT symbols, junk."""
    from falcon_unzip_amd import _lib, sim
    rng = np.random.Generator(np.random.PCG64(77))
    L = 40000
    hap0, hap1, _ = sim.make_diploid(L, rng)
    big = np.concatenate((rng.integers(0, 4, 3000, dtype=np.uint8), hap0, rng.integers(0, 4, 3000, dtype=np.uint8)))
    ctg = sim.codes_to_str(hap0).encode()

    def noisy(codes):
        seq, _, _ = sim.simulate_read(codes, codes, 0, len(codes), rng)
        return sim.codes_to_str(seq).encode()

    raw = [
        noisy(big[1000:9000]),                 # hangs over the contig start by 2000 bases
        noisy(big[L - 2000:L + 5500]),         # hangs over the contig end
        noisy(big),                            # longer than the whole contig
        noisy(sim.revcomp_codes(big[500:8000])),
        noisy(hap0[5000:12000]).lower(),       # lower case
        noisy(hap0[15000:22000]).replace(b"A", b"N", 40),   # other symbols are read as 'A'
        bytes(rng.choice(np.frombuffer(b"ACGT", np.uint8), size=5000)),   # junk
        b"ACGT" * 3, b"",                      # shorter than k, empty
    ]
    exp, exp_cig = oracle_lib.align_reads(oracle, ctg, raw)
    job = _lib.align_job(eng, [ctg], raw)
    job.run()
    got = job.summaries()
    for f in FIELDS:
        assert np.array_equal(got[f], exp[f]), (f, got[f], exp[f])
    assert list(got["aligned"]) == [1, 1, 1, 1, 1, 1, 0, 0, 0]
    assert got["q_start"][0] > 1500 and got["pos"][0] < 20          # clipped prefix, starts at the contig's first bases
    assert got["ref_end"][1] > L - 20 and got["ref_end"][2] > L - 20
    assert got["strand"][3] == 1
    aln, idx = job.alnset(0)
    for k, r in enumerate(idx):
        words = np.array([(l << 4) | o for l, o in aln.cigar_of(k)], dtype=np.uint32)
        assert np.array_equal(words, exp_cig[r]), (k, r)
    job.close()


def test_reads_given_as_spans_of_a_buffer(eng):
    """fzp_align_create_spans (what fzp_phase_contigs_files hands the bytes of <ctg>_reads.fa to): the reads as spans of one buffer -- FASTA headers and line ends between
    them, the spans in any order, an empty one, two that share bytes -- give the job fzp_align_create makes from the same reads back to back: every summary field and every record."""
    from falcon_unzip_amd import _lib
    ctg, reads, raw = _sim(41, 80000, 40, 4000)
    raw.append(b"")
    raw.append(raw[3][100:2500])                       # lies inside another read's bytes
    order = np.random.Generator(np.random.PCG64(5)).permutation(len(raw) - 1)
    buf, where = bytearray(b"junk before the first record\n"), {}
    for k in order:                                    # a FASTA file's bytes: one line per sequence, records in shuffled order
        buf += b">read/%d some description\n" % k
        where[int(k)] = (len(buf), len(buf) + len(raw[k]))
        buf += raw[k] + b"\n"
    where[len(raw) - 1] = (where[3][0] + 100, where[3][0] + 2500)
    be = np.array([where[k] for k in range(len(raw))], np.int64)
    a = _lib.align_job(eng, [ctg], raw)
    b = _lib.align_job_spans(eng, [ctg], bytes(buf), be, np.zeros(len(raw), np.int32))
    a.run()
    b.run()
    sa, sb = a.summaries(), b.summaries()
    assert sa.tobytes() == sb.tobytes() and sa["aligned"].sum() >= 38
    ra, ia = a.alnset(0)
    rb, ib = b.alnset(0)
    assert np.array_equal(ia, ib) and ra.n_rec == rb.n_rec
    assert all(ra.cigar_of(k) == rb.cigar_of(k) for k in range(ra.n_rec))
    a.close()
    b.close()
    with pytest.raises(_lib.FzpError):
        _lib.align_job_spans(eng, [ctg], bytes(buf), np.array([[10, 5]], np.int64), np.zeros(1, np.int32))      # end before begin


def test_empty_and_unaligned_only_jobs(eng):
    from falcon_unzip_amd import _lib
    ctg = b"ACGTTGCA" * 500
    job = _lib.align_job(eng, [ctg], [])
    job.run()
    assert len(job.summaries()) == 0
    b = job.to_batch()
    b.run(_lib.STAGE_ALL)
    r = b.results()[0]
    assert len(r.sites) == len(r.preads) == 0
    b.close()
    job.close()
    rng = np.random.Generator(np.random.PCG64(3))
    junk = [bytes(rng.choice(np.frombuffer(b"ACGT", np.uint8), size=3000)) for _ in range(5)]
    ctg2 = bytes(rng.choice(np.frombuffer(b"ACGT", np.uint8), size=50000))
    job = _lib.align_job(eng, [ctg2], junk)
    job.run()
    assert job.summaries()["aligned"].sum() == 0
    a, idx = job.alnset(0)
    assert a.n_rec == 0 and a.n_qid == 0
    b = job.to_batch()
    b.run(_lib.STAGE_ALL)
    assert b.counts()["n_sites"] == 0
    b.close()
    job.close()


def test_long_reads_match_twin(eng, oracle):
    """Reads of 40-110 kb (masks of one read: up to ~4 MB, op streams past one LDS load) against the twin, then K2..K6 run."""
    from falcon_unzip_amd import _lib, sim
    rng = np.random.Generator(np.random.PCG64(123))
    L = 260000
    hap0, hap1, _ = sim.make_diploid(L, rng, het_rate=1.0 / 400)
    ctg = sim.codes_to_str(hap0).encode()
    raw = []
    for n, strand in ((40000, 0), (65000, 1), (110000, 0), (90000, 1)):
        s0 = int(rng.integers(0, L - n))
        codes = hap0[s0:s0 + n]
        seq, _, _ = sim.simulate_read(codes, codes, 0, n, rng)
        raw.append(sim.codes_to_str(sim.revcomp_codes(seq) if strand else seq).encode())
    exp, exp_cig = oracle_lib.align_reads(oracle, ctg, raw)
    job = _lib.align_job(eng, [ctg], raw)
    job.run()
    got = job.summaries()
    for f in FIELDS:
        assert np.array_equal(got[f], exp[f]), (f, got[f], exp[f])
    assert got["aligned"].all() and (got["q_end"] - got["q_start"] > 0.98 * np.array([len(r) for r in raw])).all()
    aln, idx = job.alnset(0)
    for k, r in enumerate(idx):
        words = np.array([(l << 4) | o for l, o in aln.cigar_of(k)], dtype=np.uint32)
        assert np.array_equal(words, exp_cig[r]), (k, r)
    b = job.to_batch()
    b.run(_lib.STAGE_ALL)
    t = b.consensus()
    assert len(b.results()[0].sites) == 0 and len(t.tigs) == 0        # 4-fold coverage: nothing reaches total >= 10
    t.close(); b.close(); job.close()


def _cmp_twin(job, oracle, ctg, raw, params=None, cigars=True):
    exp, exp_cig = oracle_lib.align_reads(oracle, ctg, raw, params)
    got = job.summaries()
    for f in FIELDS:
        assert np.array_equal(got[f], exp[f]), (f, np.flatnonzero(got[f] != exp[f])[:5], got[f][:4], exp[f][:4])
    if cigars:
        aln, idx = job.alnset(0)
        for k, r in enumerate(idx):
            words = np.array([(l << 4) | o for l, o in aln.cigar_of(k)], dtype=np.uint32)
            assert np.array_equal(words, exp_cig[r]), (k, r)
    return got, exp


def test_headline_shape_5mb_contig_matches_twin(eng, oracle):
    """The bench's shape: 15 kb CLR reads, both strands, against a 5 Mb contig -- a 2^23-slot
    k-mer table and ~4 900 vote bins -- summaries and CIGARs equal the twin's."""
    from falcon_unzip_amd import _lib
    ctg, reads, raw = _sim(51, 5_000_000, 120, 15000)
    job = _lib.align_job(eng, [ctg], raw)
    job.run()
    got, _ = _cmp_twin(job, oracle, ctg, raw)
    assert got["aligned"].mean() >= 0.99 and set(np.unique(got["strand"])) == {0, 1}
    job.close()


def test_vote_bin_shift_above_10_matches_twin(eng, oracle):
    """A 9 Mb contig: (Lc + n) >> 10 exceeds the 8192 vote bins, so the bins are 2 048 diagonals wide (shift 11)."""
    from falcon_unzip_amd import _lib
    ctg, reads, raw = _sim(52, 9_000_000, 40, 15000)
    job = _lib.align_job(eng, [ctg], raw)
    job.run()
    got, _ = _cmp_twin(job, oracle, ctg, raw)
    assert got["aligned"].all()
    job.close()


def _sim_repeats(seed, L, n, R, **kw):
    from falcon_unzip_amd import sim
    rng = np.random.Generator(np.random.PCG64(seed))
    hap0, hap1, _, spans = sim.make_repeat_diploid(L, rng, **kw)
    reads = sim.simulate_reads(hap0, hap1, n, R, rng, strand_mix=0.5)
    ctg = sim.codes_to_str(hap0).encode()
    raw = [sim.codes_to_str(r.raw_seq_codes()).encode() for r in reads]
    return ctg, reads, raw, spans


def test_repeat_genome_matches_twin_and_truth(eng, oracle):
    """Tandem arrays and 2-6 kb interspersed copies at 95-99 % identity: every position of a k-mer is
    indexed, second placements are extended and the better one kept, anchors come from chains.  HIP == twin, and the
    placements hold against the simulator's truth."""
    from falcon_unzip_amd import _lib
    ctg, reads, raw, spans = _sim_repeats(33, 300000, 300, 15000, n_families=10, n_tandem=12)
    job = _lib.align_job(eng, [ctg], raw)
    job.run()
    s, _ = _cmp_twin(job, oracle, ctg, raw)
    assert job.n_second() > 0                              # some reads did carry a second candidate
    start = np.array([r.start for r in reads]); end = np.array([r.start + r.ref_span() for r in reads]); strand = np.array([r.strand for r in reads])
    ok = s["aligned"] == 1
    assert ok.mean() >= 0.99 and np.all(s["strand"][ok] == strand[ok])
    # inside a tandem array a start / end may sit one unit off: judge by overlap with the true interval
    ov = np.minimum(s["ref_end"], end) - np.maximum(s["pos"], start)
    assert np.mean(ov[ok] >= 0.97 * (end - start)[ok]) >= 0.99
    assert np.mean((np.abs(s["pos"] - start) <= 64)[ok]) >= 0.95
    assert np.mean(((s["q_end"] - s["q_start"]) / np.array([len(x) for x in raw]))[ok] >= 0.99) >= 0.99
    job.close()


def test_second_candidate_can_win(eng, oracle):
    """A read drawn from the diverged copy of a duplicated 12 kb segment: both copies collect votes; the extension score decides."""
    from falcon_unzip_amd import _lib, sim
    rng = np.random.Generator(np.random.PCG64(91))
    L = 200000
    hap0 = rng.integers(0, 4, size=L, dtype=np.uint8)
    seg = hap0[20000:32000].copy()
    hap0[120000:120000 + 12000] = sim._diverge(seg, rng, 0.02)[:12000]
    ctg = sim.codes_to_str(hap0).encode()
    raw, truth = [], []
    for s0 in (20500, 120500, 21000, 121000, 19000, 119500):
        seq, _, _ = sim.simulate_read(hap0, hap0, s0, 10000, rng)
        raw.append(sim.codes_to_str(seq).encode()); truth.append(s0)
    job = _lib.align_job(eng, [ctg], raw)
    job.run()
    got, _ = _cmp_twin(job, oracle, ctg, raw)
    assert job.n_second() >= 4
    assert np.all(np.abs(got["pos"] - np.array(truth)) <= 64), (got["pos"], truth)
    job.close()


def test_second_window_next_to_the_first(eng, oracle):
    """Reads that skip 3-4 vote bins of the contig half way (a deletion of 3.3-4.6 kb): the second window lies on the same strand right next to the first, the two windows
    share hits, and both chains hand out waypoints -- each window's chain must keep its own waypoint links (their waves run side by side).  HIP == twin on every field."""
    from falcon_unzip_amd import _lib, sim
    rng = np.random.Generator(np.random.PCG64(93))
    L = 400000
    hap0 = rng.integers(0, 4, size=L, dtype=np.uint8)
    ctg = sim.codes_to_str(hap0).encode()
    raw = []
    for k in range(160):
        s0 = int(rng.integers(0, L - 40000))
        gap = 3300 + 10 * k
        a = int(rng.integers(5000, 12000))
        tpl = np.concatenate((hap0[s0:s0 + a], hap0[s0 + a + gap:s0 + a + gap + int(rng.integers(5000, 12000))]))
        seq, _, _ = sim.simulate_read(tpl, tpl, 0, len(tpl), rng)
        raw.append(sim.codes_to_str(sim.revcomp_codes(seq) if k % 2 else seq).encode())
    job = _lib.align_job(eng, [ctg], raw)
    job.run()
    got, _ = _cmp_twin(job, oracle, ctg, raw)
    assert job.n_second() >= 100 and got["aligned"].all()
    job.close()


def test_identity_gate(eng, oracle):
    """--minPctIdentity 70 (unzip.py:87): a read at ~60 % identity is dropped by the gate and kept without it; n_match is exact."""
    from falcon_unzip_amd import _lib, sim
    rng = np.random.Generator(np.random.PCG64(92))
    L = 100000
    hap0 = rng.integers(0, 4, size=L, dtype=np.uint8)
    ctg = sim.codes_to_str(hap0).encode()
    raw = []
    for sub, ins, dele in ((0.01, 0.08, 0.04), (0.16, 0.12, 0.08), (0.22, 0.14, 0.10)):
        seq, _, _ = sim.simulate_read(hap0, hap0, 30000, 8000, rng, sub=sub, ins=ins, dele=dele)
        raw.append(sim.codes_to_str(seq).encode())
    for pid in (70, 0):
        job = _lib.align_job(eng, [ctg], raw, params={"min_pct_identity": pid})
        job.run()
        got, _ = _cmp_twin(job, oracle, ctg, raw, {"min_pct_identity": pid})
        ident = 100.0 * got["n_match"] / np.maximum(1, (got["q_end"] - got["q_start"]) + (got["ref_end"] - got["pos"]) - got["n_columns"])
        if pid:
            assert got["aligned"][0] == 1 and np.all(ident[got["aligned"] == 1] >= 70)
        else:
            assert got["aligned"][0] == 1
        job.close()
    # n_match against a direct count over the =/X CIGAR
    job = _lib.align_job(eng, [ctg], raw[:1])
    job.run()
    aln, _ = job.alnset(0)
    assert sum(l for l, o in aln.cigar_of(0) if o == 7) == job.summaries()["n_match"][0]
    job.close()


def _full_matrix_best(q, t, match=2, mismatch=4, gap=3, border=False):
    """Unbanded extension DP from the origin, H(-1,-1) = 0, linear gaps, no zero floor: best score over all cells, or (border=True) over the
    cells of the last row and the last column only.  Independent of the band, the steering and the trace-back of the spec (numpy, row by row)."""
    nq, nt = len(q), len(t)
    ar = np.arange(nt + 1, dtype=np.int64)
    prev = -gap * ar                      # row -1: H(-1, j-1) at index j, index 0 = the corner
    best = -(1 << 60)
    for i in range(nq):
        s = np.where(t == q[i], match, -mismatch)
        base = np.empty(nt + 1, np.int64)
        base[0] = -gap * (i + 1)          # H(i, -1)
        base[1:] = np.maximum(prev[:-1] + s, prev[1:] - gap)
        cur = np.maximum.accumulate(base + gap * ar) - gap * ar      # the left-gap chain
        if not border:
            best = max(best, int(cur[1:].max()))
        else:
            best = max(best, int(cur[nt]), int(cur[1:].max()) if i == nq - 1 else best)
        prev = cur
    return best


def _two_way(ori, ctg_codes, i_a, c_a, border=False):
    """(forward, backward) full-matrix optimum of the extensions from the anchor (i_a, c_a): best over all cells (the local optimum) or over the border
    cells (fzalign v1.5's terminal).  The backward one is the same DP on the reversed read prefix and contig window; None when the anchor is at an edge."""
    L = len(ctg_codes)
    q = ori[i_a:]
    nt = min(L - c_a, len(q) + len(q) // 4 + 64)
    f = _full_matrix_best(q, ctg_codes[c_a:c_a + nt], border=border)
    b = None
    if i_a > 0 and c_a > 0:
        ntb = min(c_a, i_a + i_a // 4 + 64)
        b = _full_matrix_best(ori[:i_a][::-1], ctg_codes[c_a - ntb:c_a][::-1], border=border)
    return f, b


def _check_against_full_matrix(oracle, ctg, raw, ori, hap0, strand_true, score, cig):
    """fzalign v1.5 against a plain full-matrix DP, for one aligned read: (1) an origin of the twin's seeding exists whose two extensions reach exactly the
    full-matrix optimum over the border cells -- the band never cut the best path; (2) the reported score is the score of the reported CIGAR; (3) it is at
    most the sum of the two local optima from that origin and, the reported piece being the best stretch of the path to the border, within a few columns of it."""
    cs = sum(2 * l if o == 7 else -4 * l if o == 8 else -3 * l if o in (1, 2) else 0 for l, o in cig)
    assert cs == score
    first = [o for _, o in cig if o != 4]
    assert first[0] == 7 and first[-1] == 7
    for strand, i_a, c_a, tf, tb in oracle_lib.align_origins(oracle, ctg, raw):
        if strand != strand_true:
            continue
        f, b = _two_way(ori, hap0, i_a, c_a, border=True)
        if f != tf or (b is not None and b != tb):
            continue
        lf, lb = _two_way(ori, hap0, i_a, c_a)
        top = lf + (lb if lb is not None and lb > 0 else 0)
        if top - 40 <= score <= top:
            return True
    return False


def test_scores_equal_unbanded_dp_on_short_reads(eng, oracle):
    """Spec-independent check: for reads of ~2.5 kb the extensions must reach the optimum of a plain full-matrix DP from the
    same origin -- the band never cut the best path, and twin and kernel do not share an arithmetic bug (_check_against_full_matrix; the kernel's
    fields equal the twin's, whose terminal scores the hook reports)."""
    from falcon_unzip_amd import _lib, sim
    rng = np.random.Generator(np.random.PCG64(93))
    L = 60000
    hap0, hap1, _ = sim.make_diploid(L, rng)
    reads = sim.simulate_reads(hap0, hap1, 24, 2300, rng, strand_mix=0.5)
    ctg = sim.codes_to_str(hap0).encode()
    raw = [sim.codes_to_str(r.raw_seq_codes()).encode() for r in reads]
    job = _lib.align_job(eng, [ctg], raw)
    job.run()
    s = job.summaries()
    exp, _ = oracle_lib.align_reads(oracle, ctg, raw)
    for f in FIELDS:
        assert np.array_equal(s[f], exp[f]), f
    aln, idx = job.alnset(0)
    assert s["aligned"].sum() >= 22
    cig_of = {int(r): aln.cigar_of(k) for k, r in enumerate(idx)}
    checked = 0
    for r, rd in enumerate(reads):
        if not s["aligned"][r] or r not in cig_of:
            continue
        assert _check_against_full_matrix(oracle, ctg, raw[r], rd.seq, hap0, rd.strand, int(s["score"][r]), cig_of[r]), (r, s[r])
        checked += 1
    assert checked >= 20
    job.close()


def _shaped(seed, L, n, **kw):
    from falcon_unzip_amd import sim
    rng = np.random.Generator(np.random.PCG64(seed))
    hap0, hap1, _ = sim.make_diploid(L, rng)
    codes, off, st, hp, sd, lens, bf, truth = sim.simulate_raw_reads_shaped(hap0, hap1, n, rng, with_truth=True, **kw)
    return sim.ACGT[hap0].tobytes(), sim.ACGT[codes].tobytes(), off, st, sd, lens, bf, truth


def test_real_read_shape_matches_twin_and_truth(eng, oracle):
    """Reads of real CLR shape: log-normal lengths (median 12 kb, 3-60 kb) and bursty errors (0.3-1 kb stretches at
    30 %, a quarter of the reads with one at their head).  HIP == twin on every field; >= 99 % of the reads start within 64 bp of where
    their first aligned base truly lies and end within 64 bp of their true end; >= 95 % of all read bases are inside alignments and
    >= 98 % of the reads have >= 95 % of their own bases inside (the rest lost the band behind a noisy head: forward extension only)."""
    from falcon_unzip_amd import _lib
    n = 1200
    ctg, blob, off, st, sd, lens, bf, truth = _shaped(47, 2_000_000, n)
    rl = np.diff(off)
    assert rl.min() < 4500 and rl.max() > 45000 and 11000 < np.median(lens) < 13500 and (bf > 0).mean() > 0.5
    job = _lib.align_job_raw(eng, [ctg], blob, off, np.zeros(n, np.int32))
    job.run()
    s = job.summaries()
    exp, _ = oracle_lib.align_reads(oracle, ctg, [blob[off[i]:off[i + 1]] for i in range(n)], n_threads=8)
    for f in FIELDS:
        assert np.array_equal(s[f], exp[f]), (f, np.flatnonzero(s[f] != exp[f])[:5])
    ok = s["aligned"] == 1
    assert ok.mean() >= 0.995 and np.all(s["strand"][ok] == sd[ok])
    true_pos = st + truth[off[:-1] + np.clip(s["q_start"], 0, rl - 1)]
    placed = (np.abs(s["pos"] - true_pos) <= 64) & (np.abs(s["ref_end"] - (st + lens)) <= 64)
    assert placed[ok].mean() >= 0.99, placed[ok].mean()
    inside = (s["q_end"] - s["q_start"])[ok]
    assert inside.sum() >= 0.95 * rl[ok].sum() and np.mean(inside / rl[ok] >= 0.95) >= 0.98, (inside.sum() / rl[ok].sum(), np.mean(inside / rl[ok] >= 0.95))
    job.close()


def test_the_three_dp_kernels_agree(eng, oracle, monkeypatch):
    """The banded DP exists three times: k_sw (a wave per piece, integer scores), k_swb (bit-sliced, a piece per lane) and k_swb2 (bit-sliced, a piece per pair of
    lanes).  Same reads through each (FZP_SW_NO_BITS / FZP_SWB_64 / FZP_SWB_PAIR; the default mixes them by piece size): every summary field and every CIGAR equal, and
    equal to the twin's.  Reads from 70 bases up, so that extensions narrower than the band (which stay with k_sw) and pieces over a (lowered) step
    limit of the bit-sliced kernels are both in the set; once more with the reads cut into three chunks (every chunk its own slots, launch lists and mask buffers)."""
    from falcon_unzip_amd import _lib
    n = 400
    ctg, blob, off, *_ = _shaped(51, 800_000, n, length_model={"median": 4000, "sigma": 1.3, "lo": 70, "hi": 30000})
    reads = [blob[off[i]:off[i + 1]] for i in range(n)]
    got = {}
    B64 = {"FZP_ALIGN_BAND": "64"}      # (fzalign v1.8: the default band has 32 cells; the 64-cell band and its kernels -- the pair-of-lanes form exists for it alone -- stay selectable)
    for mode, env in (("default", {}), ("wave_per_read", {"FZP_SW_NO_BITS": "1"}), ("lane64", {"FZP_SWB_64": "1"}), ("three_chunks", {"FZP_SW_CHUNKS": "3"}),
                      ("bases_from_hbm", {"FZP_SWB_NO_RING": "1", "FZP_SWB_64": "1"}),
                      ("band64", dict(B64)), ("band64_wave_per_read", dict(B64, FZP_SW_NO_BITS="1")), ("band64_lane64", dict(B64, FZP_SWB_64="1")), ("band64_pair", dict(B64, FZP_SWB_PAIR="1")),
                      ("band64_pair_short_limit", dict(B64, FZP_SWB_PAIR="1", FZP_SWB_MAX_STEPS="9000")), ("band64_units_two_waves", dict(B64, FZP_SWB_64="1", FZP_SWB_WAVES="2", FZP_SWB_UNIT="3", FZP_SWB_GRID="8")),
                      ("walk16", {"FZP_TBW_OLD": "1"}),       # (the 16-walkers-per-wave walk on the records laid out for the 32-walker one)
                      # r6: k_swb's groups cut into work units of 2 / 3 blocks that waves park and pick up (longest remaining first), with both register budgets, with and
                      # without the turns at the issue priority, on a handful of waves (FZP_SWB_GRID) so that the job's ~30 groups wait for each other -- hundreds of hand-overs through HBM, the same bits
                      ("units_one_wave", {"FZP_SWB_64": "1", "FZP_SWB_WAVES": "1", "FZP_SWB_UNIT": "2", "FZP_SWB_GRID": "5"}),
                      ("units_two_waves", {"FZP_SWB_64": "1", "FZP_SWB_WAVES": "2", "FZP_SWB_UNIT": "3", "FZP_SWB_GRID": "8", "FZP_SWB_DBG": "4"}),
                      ("units_hysteresis", {"FZP_SWB_64": "1", "FZP_SWB_WAVES": "2", "FZP_SWB_UNIT": "1", "FZP_SWB_HYST": "2", "FZP_SWB_GRID": "3"})):
        for k, v in env.items():
            monkeypatch.setenv(k, v)
        job = _lib.align_job_raw(eng, [ctg], blob, off, np.zeros(n, np.int32))
        job.run()
        aln, idx = job.alnset(0)
        got[mode] = (job.summaries().copy(), {int(r): aln.cigar_of(k) for k, r in enumerate(idx)})
        job.close()
        for k in env:
            monkeypatch.delenv(k)
    exp = {32: oracle_lib.align_reads(oracle, ctg, reads, {"band": 32}, n_threads=8)[0], 64: oracle_lib.align_reads(oracle, ctg, reads, {"band": 64}, n_threads=8)[0]}
    both = (exp[32]["aligned"] == 1) & (exp[64]["aligned"] == 1)
    assert np.mean(exp[32]["score"][both] >= exp[64]["score"][both]) >= 0.97 and not np.array_equal(exp[32]["cells"], exp[64]["cells"])      # (on nearly every read the narrower band finds the same score)
    for mode, (s, cg) in got.items():
        band = 64 if mode.startswith("band64") else 32
        for f in FIELDS:
            assert np.array_equal(s[f], exp[band][f]), (mode, f, np.flatnonzero(s[f] != exp[band][f])[:5])
        assert cg == got["band64" if band == 64 else "default"][1], mode
    assert got["default"][0]["aligned"].mean() > 0.8            # (reads of a few hundred bases rarely gather 8 seed votes)


def test_terminal_on_the_border_reached_along_the_band_edge(eng, oracle, monkeypatch):
    """An extension whose last steps all go one way puts an edge lane of the band ON the matrix border at the very step a 32-step interior block would
    have ended: that cell is a terminal candidate and only the checked steps look at it (k_sw's `safe` count let it into the unchecked
    block).  Error-free reads with 16..31 extra bases 0..70 bases before their end (the path jumps towards lane 63 and the band runs DOWN after it, into
    the read's last row) and, at the contig's end, with as many bases missing (it runs RIGHT into the last column); every DP kernel against the twin."""
    from falcon_unzip_amd import _lib
    rng = np.random.Generator(np.random.PCG64(97))
    L = 24000
    ctg_codes = rng.integers(0, 4, L, dtype=np.uint8)
    acgt = np.frombuffer(b"ACGT", np.uint8)
    ctg = acgt[ctg_codes].tobytes()
    reads = []
    for x in range(16, 32):
        for tail in range(0, 71):
            s0 = 2 * int(rng.integers(0, (L - 9000) // 2))                          # (even: the index holds every 2nd contig position, an error-free read never changes parity)
            n_body = 2400 + int(rng.integers(0, 64))
            body = ctg_codes[s0:s0 + n_body]
            reads.append(acgt[np.concatenate((body, rng.integers(0, 4, x, dtype=np.uint8), ctg_codes[s0 + n_body:s0 + n_body + tail]))].tobytes())
            e0 = L - int(rng.integers(0, 8))                                        # the contig ends 0..7 bases after the read's last base ...
            b2 = ctg_codes[(e0 - 2400 - int(rng.integers(0, 64))) & ~1:e0]
            cut = len(b2) - tail
            reads.append(acgt[np.concatenate((b2[:cut - x], b2[cut:]))].tobytes())  # ... and x contig bases are missing from the read `tail` bases before its end
    exp, exp_cig = oracle_lib.align_reads(oracle, ctg, reads, n_threads=8)
    assert exp["aligned"].mean() > 0.95
    for mode, env in (("default", {}), ("wave_per_read", {"FZP_SW_NO_BITS": "1"}), ("lane64", {"FZP_SWB_64": "1"}), ("pair", {"FZP_SWB_PAIR": "1"})):
        for k, v in env.items():
            monkeypatch.setenv(k, v)
        job = _lib.align_job(eng, [ctg], reads)
        job.run()
        s = job.summaries()
        aln, idx = job.alnset(0)
        for f in FIELDS:
            assert np.array_equal(s[f], exp[f]), (mode, f, np.flatnonzero(s[f] != exp[f])[:5])
        for k, r in enumerate(idx):
            assert np.array_equal(np.array([(l << 4) | o for l, o in aln.cigar_of(k)], dtype=np.uint32), exp_cig[r]), (mode, k, r)
        job.close()
        for k in env:
            monkeypatch.delenv(k)


def test_paths_outside_the_recorded_lanes_are_computed_again(eng, oracle, monkeypatch):
    """The bit-sliced DP records band lanes 16..47 of its masks (8 B per step); a walk that needs another lane puts its piece on the fail list, the piece is computed again
    by the wave-per-piece kernel with whole masks and walked from those.  Real paths stay within ~12 lanes of the centre, so the test narrows what the walker accepts
    (FZP_TB_WINDOW): with +-3 lanes most pieces come back -- same summaries, same CIGARs as the twin's; more of them than there is room for and the run is done again with whole masks throughout."""
    from falcon_unzip_amd import _lib
    ctg, reads, raw = _sim(61, 400000, 50, 15000)
    exp, exp_cig = oracle_lib.align_reads(oracle, ctg, raw, n_threads=8)
    for mode, env in (("lane64", {"FZP_SWB_64": "1", "FZP_TB_WINDOW": "3"}), ("pair", {"FZP_SWB_PAIR": "1", "FZP_TB_WINDOW": "3"}), ("edge", {"FZP_SWB_64": "1", "FZP_TB_WINDOW": "9"})):
        for k, v in env.items():
            monkeypatch.setenv(k, v)
        job = _lib.align_job(eng, [ctg], raw)
        job.run()
        s = job.summaries()
        aln, idx = job.alnset(0)
        for f in FIELDS:
            assert np.array_equal(s[f], exp[f]), (mode, f, np.flatnonzero(s[f] != exp[f])[:5])
        for k, r in enumerate(idx):
            assert np.array_equal(np.array([(l << 4) | o for l, o in aln.cigar_of(k)], dtype=np.uint32), exp_cig[r]), (mode, k, r)
        job.close()
        for k in env:
            monkeypatch.delenv(k)
    # more of them than there is room for: the run is done again with whole masks for every piece (or, with FZP_TB_NO_RETRY, refused with a message)
    ctg, reads, raw = _sim(62, 1000000, 400, 15000)
    exp, _ = oracle_lib.align_reads(oracle, ctg, raw, n_threads=8)
    monkeypatch.setenv("FZP_SWB_64", "1")
    monkeypatch.setenv("FZP_TB_WINDOW", "2")
    job = _lib.align_job(eng, [ctg], raw)
    job.run()
    s = job.summaries()
    for f in FIELDS:
        assert np.array_equal(s[f], exp[f]), ("retry", f)
    monkeypatch.setenv("FZP_TB_NO_RETRY", "1")
    with pytest.raises(_lib.FzpError) as e:
        job.run()
    assert "whole trace-back masks" in str(e.value)
    job.close()


@pytest.mark.parametrize("seed", [201, 202])
def test_randomized_reads_of_every_kind_match_the_twin(eng, oracle, seed, monkeypatch):
    """A wide draw against a contig with repeats: read lengths log-uniform over 100 .. 120 000 bases (from fewer than one band width to 30 pieces), clean / CLR / bursty /
    far diverged error profiles, both strands, reads hanging over either contig end, chimeras of two loci, junk, N's and lower case in reads and contig, duplicates -- every
    summary field and every CIGAR equal to the twin's; the second seed also with the reads cut into chunks (several plans, fail lists, joins per run)."""
    from falcon_unzip_amd import _lib, sim
    rng = np.random.Generator(np.random.PCG64(seed))
    L = 600000
    hap0, hap1, _, spans = sim.make_repeat_diploid(L, rng, n_families=12, n_tandem=12)
    acgt = np.frombuffer(b"ACGT", np.uint8)
    ctg_b = bytearray(acgt[hap0].tobytes())
    for x in rng.integers(0, L, 40):                       # a few N's and a lower-case stretch in the contig
        ctg_b[x] = ord("N")
    ctg_b[5000:9000] = bytes(ctg_b[5000:9000]).lower()
    ctg = bytes(ctg_b)
    raw = []
    n_reads = 700
    for k in range(n_reads):
        n = int(np.exp(rng.uniform(np.log(100), np.log(120000 if k % 9 == 0 else 30000))))
        kind = k % 7
        s0 = int(rng.integers(0, max(1, L - n)))
        if kind == 5:                                       # hangs over an end of the contig
            s0 = 0 if k % 2 else max(0, L - n)
        tpl = hap0[s0:s0 + n] if k % 2 else hap1[s0:s0 + n]
        if kind == 0:
            seq = tpl.copy()                                # error-free
        elif kind == 4:
            seq, _, _ = sim.simulate_read(tpl, tpl, 0, len(tpl), rng, sub=0.10, ins=0.12, dele=0.08)        # far diverged (around the identity gate)
        else:
            seq, _, _ = sim.simulate_read(tpl, tpl, 0, len(tpl), rng)
        if kind == 3 and len(seq) > 2000:                   # a burst of noise in the middle
            b0 = int(rng.integers(0, len(seq) - 1000)); bl = int(rng.integers(200, 1000))
            seq = np.concatenate((seq[:b0], rng.integers(0, 4, bl, dtype=np.uint8), seq[b0 + bl // 2:]))
        if kind == 6 and len(seq) > 4000:                   # a chimera: the second half from elsewhere
            s1 = int(rng.integers(0, L - len(seq)))
            other, _, _ = sim.simulate_read(hap0[s1:s1 + len(seq) // 2], hap0[s1:s1 + len(seq) // 2], 0, len(seq) // 2, rng)
            seq = np.concatenate((seq[:len(seq) // 2], other))
        if kind == 5 and n > 500:                           # bases beyond the contig's end
            ext = rng.integers(0, 4, int(rng.integers(50, 400)), dtype=np.uint8)
            seq = np.concatenate((ext, seq)) if k % 2 else np.concatenate((seq, ext))
        if k % 3 == 0:
            seq = sim.revcomp_codes(seq)
        b = bytearray(acgt[seq].tobytes())
        if k % 11 == 0 and len(b) > 50:
            for x in rng.integers(0, len(b), 5):
                b[x] = ord("N")
            b[10:40] = bytes(b[10:40]).lower()
        raw.append(bytes(b))
    raw += [bytes(rng.choice(acgt, size=5000)), raw[3], raw[3], b"ACGT" * 3, b""]      # junk, duplicates, a read shorter than a seed, an empty one
    if seed == 202:
        monkeypatch.setenv("FZP_SW_CHUNKS", "5")
    exp, exp_cig = oracle_lib.align_reads(oracle, ctg, raw, n_threads=8)
    job = _lib.align_job(eng, [ctg], raw)
    job.run()
    got = job.summaries()
    for f in FIELDS:
        assert np.array_equal(got[f], exp[f]), (f, np.flatnonzero(got[f] != exp[f])[:8])
    aln, idx = job.alnset(0)
    for k, r in enumerate(idx):
        assert np.array_equal(np.array([(l << 4) | o for l, o in aln.cigar_of(k)], dtype=np.uint32), exp_cig[r]), (k, r)
    assert got["aligned"].mean() > 0.5 and got["aligned"][-5] == 0 and got["aligned"][-1] == 0 and got["aligned"][-2] == 0      # (a third of the draw is too short, too diverged or junk)
    job.close()


def test_record_planning_at_deep_coverage(eng):
    """A contig with 24 000 reads (many starting in the same 256-bp bin, many at the same POS): the device's record planning (binned rank,
    fzp_align_to_batch) must order records exactly like the host's sort behind fzp_align_alnset ('samtools sort' order: POS, then read index)."""
    from falcon_unzip_amd import _lib, sim
    rng = np.random.Generator(np.random.PCG64(95))
    L = 40000
    hap0, hap1, _ = sim.make_diploid(L, rng, het_rate=1.0 / 300)
    codes, off, *_ = sim.simulate_raw_reads_bulk(hap0, hap1, 24000, 2600, rng)
    ctg = sim.ACGT[hap0].tobytes()
    job = _lib.align_job_raw(eng, [ctg], sim.ACGT[codes].tobytes(), off, np.zeros(24000, np.int32))
    job.run()
    aln, idx = job.alnset(0)
    assert aln.n_rec > 23000 and np.all(np.diff(aln.rec_pos()) >= 0)
    b1 = job.to_batch()
    b1.run(_lib.STAGE_ALL)
    b2 = eng.batch([aln], [ctg])
    b2.run(_lib.STAGE_ALL)
    r1, r2 = b1.result(0), b2.result(0)
    for f in ("sites", "vmap_qid", "arows", "pvars", "preads"):
        assert np.array_equal(getattr(r1, f), getattr(r2, f)), f
    assert len(r1.sites) > 50 and len(r1.preads) > 10000
    b1.close(); b2.close(); job.close()


def test_packed_hand_off_equals_the_byte_hand_off(eng, monkeypatch):
    """r5: K2 reads K1's alignments where K1 leaves them -- 2-bit op streams (END first), 2-bit reads, 256-op checkpoints -- instead of run-length CIGAR words and byte SEQ.
    The same job through both forms (FZP_K2_BYTES makes the byte form and runs the byte kernels on it): every record of every stage equal.  Reads of real shape (3-60 kb,
    bursts) so that streams end on and off word, checkpoint and tile boundaries, on both strands, with clips at either end; then K6, which asks a packed batch for its bytes."""
    from falcon_unzip_amd import _lib
    n = 900
    ctg, blob, off, st, sd, lens, bf, truth = _shaped(53, 600_000, n)
    job = _lib.align_job_raw(eng, [ctg], blob, off, np.zeros(n, np.int32))
    job.run()
    s = job.summaries()
    assert (s["aligned"] == 1).mean() > 0.99 and (s["strand"] == 1).mean() > 0.3 and (s["q_start"] > 0).mean() > 0.05
    res = {}
    for form in ("packed", "bytes"):
        if form == "bytes":
            monkeypatch.setenv("FZP_K2_BYTES", "1")
        b = job.to_batch()
        b.run(_lib.STAGE_ALL)
        res[form] = b.result(0)
        if form == "packed":
            tig_p = b.consensus()                      # (a packed batch makes its run-length records when K6 asks)
            fa_p = tig_p.fasta(0, "c")
            tig_p.close()
        else:
            tig_b = b.consensus()
            assert tig_b.fasta(0, "c") == fa_p and len(fa_p) > 1000
            tig_b.close()
        b.close()
    monkeypatch.delenv("FZP_K2_BYTES")
    a, c = res["packed"], res["bytes"]
    for f in ("sites", "vmap_qid", "arows", "pvars", "preads"):
        assert np.array_equal(getattr(a, f), getattr(c, f)), f
    assert len(a.sites) > 100 and len(a.vmap_qid) > 2000 and len(a.preads) > 300
    job.close()


def test_a_batch_borrows_its_job(eng):
    """fzp_align_to_batch's lifetime rule (include/fzphase.h): the batch reads the job's packed records in place, so the job may neither run again nor go away under it.
    The binding refuses both; the library itself refuses the run and, once the job IS destroyed, fails the batch's next stage instead of reading freed memory."""
    from falcon_unzip_amd import _lib
    import ctypes as C
    ctg, blob, off, *_ = _shaped(77, 200_000, 60, length_model={"median": 5000, "sigma": 0.5, "lo": 3000, "hi": 9000})
    job = _lib.align_job_raw(eng, [ctg], blob, off, np.zeros(60, np.int32))
    job.run()
    b = job.to_batch()
    with pytest.raises(_lib.FzpError):
        job.run()
    with pytest.raises(_lib.FzpError):
        job.close()
    lib = _lib.load()
    assert lib.fzp_align_run(eng._p, job._p) != 0 and b"still open" in lib.fzp_last_error()
    b.run(_lib.STAGE_HET)                                   # the batch itself is fine
    lib.fzp_align_destroy(eng._p, job._p)                   # (past the binding: what a C caller could do)
    job._p = None
    with pytest.raises(_lib.FzpError, match="destroyed"):
        b.run(_lib.STAGE_ALL)
    b.close()
    # the right order: batch first, then the job runs again
    job = _lib.align_job_raw(eng, [ctg], blob, off, np.zeros(60, np.int32))
    job.run()
    b = job.to_batch()
    b.run(_lib.STAGE_ALL)
    b.close()
    job.run()
    job.close()


def _dev_index(eng, job, ctg):
    import ctypes as C
    from falcon_unzip_amd import _lib
    lib = _lib.load()
    p, n = C.c_void_p(), C.c_int64()
    f = lib.fzp_debug_index_entries
    f.restype = C.c_int
    f.argtypes = [C.c_void_p, C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p]
    assert f(eng._p, job._p, ctg, C.byref(p), C.byref(n)) == 0, lib.fzp_last_error()
    out = np.frombuffer(C.string_at(p, 8 * n.value), np.uint64).copy()
    lib.fzp_free(p)
    return out


def _dev_hits(eng, job, r):
    import ctypes as C
    from falcon_unzip_amd import _lib
    lib = _lib.load()
    out = np.zeros((4096, 2), np.uint32)
    n = C.c_int32()
    f = lib.fzp_debug_read_hits
    f.restype = C.c_int
    f.argtypes = [C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p]
    assert f(eng._p, job._p, r, out.ctypes.data_as(C.c_void_p), C.byref(n)) == 0, lib.fzp_last_error()
    return out[:n.value].copy()


def _dev_fingerprint(eng, job, rebuild=False):
    import ctypes as C
    from falcon_unzip_amd import _lib
    lib = _lib.load()
    if rebuild:
        f = lib.fzp_debug_rebuild_index
        f.restype = C.c_int
        f.argtypes = [C.c_void_p, C.c_void_p]
        assert f(eng._p, job._p) == 0, lib.fzp_last_error()
    out = np.zeros(3, np.uint64)
    f = lib.fzp_debug_index_fingerprint
    f.restype = C.c_int
    f.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
    assert f(eng._p, job._p, out.ctypes.data_as(C.c_void_p)) == 0, lib.fzp_last_error()
    return tuple(int(x) for x in out)


def test_kmer_table_and_hit_lists_equal_the_twins(eng, oracle):
    """K1's INTERMEDIATES (seeding is redundant by design -- ~85 hits per read --, so a few wrong k-mers need not move an alignment, and no test looked at
    them).  The k-mer table of every contig, as a sorted list of entries, and the hit list of every one of 2 000 reads, in spec order: equal to the twin's, entry for entry.
    Both samplings (v1.7's anchored k-mers, v1.6's strides), contigs whose lengths put the last k-mers at every offset of a packed word."""
    from falcon_unzip_amd import _lib, sim
    rng = np.random.Generator(np.random.PCG64(4711))
    ctgs, blobs, offs, rctg = [], [], [np.zeros(1, np.int64)], []
    base = 0
    for c, L in enumerate((300_000, 300_007, 123_461, 77_777)):
        hap0, hap1, _ = sim.make_diploid(L, rng)
        codes, off, *_ = sim.simulate_raw_reads_bulk(hap0, hap1, 500, 9000, rng)
        ctgs.append(sim.ACGT[hap0].tobytes()); blobs.append(sim.ACGT[codes].tobytes())
        offs.append(off[1:] + base); base += int(off[-1]); rctg.append(np.full(500, c, np.int32))
    blob, off, rctg = b"".join(blobs), np.concatenate(offs), np.concatenate(rctg)
    for params in ({}, {"seed_anchored": 0}):
        job = _lib.align_job_raw(eng, ctgs, blob, off, rctg, params=params)
        job.run()
        for c, ctg in enumerate(ctgs):
            dev, twin = _dev_index(eng, job, c), oracle_lib.debug_index(oracle, ctg, params)
            assert len(dev) == len(twin) > len(ctg) // 20 and np.array_equal(dev, twin), (params, c, len(dev), len(twin))
        n_hits = 0
        for r in range(len(rctg)):
            dev = _dev_hits(eng, job, r)
            twin = oracle_lib.debug_hits(oracle, ctgs[rctg[r]], blob[off[r]:off[r + 1]], params)
            assert dev.shape == twin.shape and np.array_equal(dev, twin), (params, r, dev.shape, twin.shape)
            n_hits += len(dev)
        assert n_hits > 40 * len(rctg)
        job.close()


def test_ten_builds_of_the_bench_index_are_the_same_table(eng):
    """The k-mer tables of the bench step's contigs (20 x 5 Mb) built ten times: which slot an entry lands in is a race between inserts, WHAT is in the table is not -- the
    number of entries and an order-free 128-bit fingerprint of them (sum and xor of a 64-bit mix of every entry) stay the same; and one contig's sorted entries, downloaded
    after the first and after the last build, are equal word for word."""
    from falcon_unzip_amd import _lib, sim
    rng = np.random.Generator(np.random.PCG64(20260002))
    ctgs = [sim.ACGT[rng.integers(0, 4, 5_000_000, dtype=np.uint8)].tobytes() for _ in range(20)]
    rd = ctgs[0][1000:9000]
    job = _lib.align_job_raw(eng, ctgs, rd, np.array([0, len(rd)], np.int64), np.zeros(1, np.int32))
    first = _dev_fingerprint(eng, job)
    assert first[0] > 20 * 5_000_000 // 10                      # about an eighth of the positions are anchored k-mers
    e0 = _dev_index(eng, job, 7)
    for _ in range(9):
        assert _dev_fingerprint(eng, job, rebuild=True) == first
    assert np.array_equal(_dev_index(eng, job, 7), e0)
    job.close()
