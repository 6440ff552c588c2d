"""The K1 twin alone (CPU): spec "fzalign v1.2" against the simulator's truth, on an iid genome and on one with tandem arrays
and interspersed repeats, plus the identity gate.  (HIP == twin is tests/test_gpu_align.py.)  The twin aborts if the
score-derived match count ever differs from a direct count of '=' columns."""
import numpy as np

from tests import oracle_lib


def _run(oracle, hap0, reads, params=None):
    from falcon_unzip_amd import sim
    ctg = sim.codes_to_str(hap0).encode()
    raw = [sim.codes_to_str(r.raw_seq_codes()).encode() for r in reads]
    s, cig = oracle_lib.align_reads(oracle, ctg, raw, params)
    start = np.array([r.start for r in reads]); end = np.array([r.start + r.ref_span() for r in reads]); strand = np.array([r.strand for r in reads])
    return s, cig, start, end, strand, raw


def test_iid_genome(oracle):
    from falcon_unzip_amd import sim
    rng = np.random.Generator(np.random.PCG64(31))
    hap0, hap1, _ = sim.make_diploid(400000, rng)
    reads = sim.simulate_reads(hap0, hap1, 60, 15000, rng, strand_mix=0.5)
    s, cig, start, end, strand, raw = _run(oracle, hap0, reads)
    assert s["aligned"].all() and np.all(s["strand"] == strand)
    assert np.all(np.abs(s["pos"] - start) <= 64) and np.mean(np.abs(s["ref_end"] - end) <= 5) >= 0.98
    for k in range(len(reads)):        # '=' columns of the CIGAR == n_match; CIGAR consumes the read
        ops = [(int(w) >> 4, int(w) & 15) for w in cig[k]]
        assert sum(l for l, o in ops if o == 7) == s["n_match"][k]
        assert sum(l for l, o in ops if o in (1, 4, 7, 8)) == len(raw[k])


def test_repeat_genome(oracle):
    from falcon_unzip_amd import sim
    rng = np.random.Generator(np.random.PCG64(32))
    hap0, hap1, _, spans = sim.make_repeat_diploid(1000000, rng, n_families=20, n_tandem=20)
    reads = sim.simulate_reads(hap0, hap1, 150, 15000, rng, strand_mix=0.5)
    s, cig, start, end, strand, raw = _run(oracle, hap0, reads)
    ok = s["aligned"] == 1
    assert ok.mean() >= 0.99 and np.all(s["strand"][ok] == strand[ok])
    ov = np.minimum(s["ref_end"], end) - np.maximum(s["pos"], start)
    assert np.mean(ov[ok] >= 0.97 * (end - start)[ok]) >= 0.99
    assert np.mean(((s["q_end"] - s["q_start"]) / np.array([len(x) for x in raw]))[ok] >= 0.99) >= 0.99
    assert len(spans) > 30


def test_identity_gate(oracle):
    from falcon_unzip_amd import sim
    rng = np.random.Generator(np.random.PCG64(92))
    hap0 = rng.integers(0, 4, size=100000, dtype=np.uint8)
    ctg = sim.codes_to_str(hap0).encode()
    seq, _, _ = sim.simulate_read(hap0, hap0, 30000, 8000, rng, sub=0.22, ins=0.14, dele=0.10)
    raw = [sim.codes_to_str(seq).encode()]
    off, _ = oracle_lib.align_reads(oracle, ctg, raw, {"min_pct_identity": 0})
    on, _ = oracle_lib.align_reads(oracle, ctg, raw)
    if off["aligned"][0]:
        ident = 100.0 * off["n_match"][0] / ((off["q_end"][0] - off["q_start"][0]) + (off["ref_end"][0] - off["pos"][0]) - off["n_columns"][0])
        assert (ident >= 70) == bool(on["aligned"][0])
    assert on["cells"][0] == off["cells"][0]


def test_twin_scores_equal_unbanded_dp(oracle):
    """The twin's extensions reach the optimum of a plain full-matrix DP from the same origin (reads of ~2.5 kb); the reported score is the
    reported CIGAR's and lies within a few columns of the local optimum."""
    from falcon_unzip_amd import sim
    from tests.test_gpu_align import _check_against_full_matrix
    rng = np.random.Generator(np.random.PCG64(93))
    L = 60000
    hap0, hap1, _ = sim.make_diploid(L, rng)
    reads = sim.simulate_reads(hap0, hap1, 12, 2300, rng, strand_mix=0.5)
    s, cig, *_ = _run(oracle, hap0, reads)
    ctg = sim.codes_to_str(hap0).encode()
    for r, rd in enumerate(reads):
        assert s["aligned"][r]
        raw = sim.codes_to_str(rd.raw_seq_codes()).encode()
        ops = [(int(w) >> 4, int(w) & 15) for w in cig[r]]
        assert _check_against_full_matrix(oracle, ctg, raw, rd.seq, hap0, rd.strand, int(s["score"][r]), ops), (r, s[r])


def test_threaded_twin_equals_single_thread(oracle):
    from falcon_unzip_amd import sim
    rng = np.random.Generator(np.random.PCG64(35))
    hap0, hap1, _ = sim.make_diploid(200000, rng)
    reads = sim.simulate_reads(hap0, hap1, 40, 8000, rng, strand_mix=0.5)
    ctg = sim.codes_to_str(hap0).encode()
    raw = [sim.codes_to_str(r.raw_seq_codes()).encode() for r in reads]
    a, ca = oracle_lib.align_reads(oracle, ctg, raw)
    b, cb = oracle_lib.align_reads(oracle, ctg, raw, n_threads=4)
    assert a.tobytes() == b.tobytes() and all(np.array_equal(x, y) for x, y in zip(ca, cb))
