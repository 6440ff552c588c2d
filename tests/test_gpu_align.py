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


FIELDS = ("aligned", "strand", "pos", "ref_end", "q_start", "q_end", "score", "n_cigar", "cells", "n_columns")


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
