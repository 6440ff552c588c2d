"""K6, the phased-pile consensus (fzp_batch_consensus): HIP == CPU twin byte for byte, and accuracy of the whole chain
K1 -> K5 -> K6 against the simulator's true haplotypes."""
import numpy as np
import pytest

from tests import cns_util, oracle_lib
from tests.golden_util import Case

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def eng():
    from falcon_unzip_amd import _lib
    e = _lib.Engine(0)
    yield e
    e.close()


def hip_fasta(eng, alns, refs, ids):
    from falcon_unzip_amd import _lib
    b = eng.batch(alns, refs)
    b.run(_lib.STAGE_ALL)
    t = b.consensus()
    out = [t.fasta(c, ids[c]) for c in range(len(ids))]
    n = len(t.tigs)
    t.close()
    b.close()
    return out, n


@pytest.mark.parametrize("name", ["g1_cfg1_clean", "g2_cfg1_clr", "g4_multiblock", "g7_lowcov", "g8_noisy", "g9_fragmented", "g3_quirks"])
def test_matches_twin_on_goldens(eng, oracle, name):
    from falcon_unzip_amd import _lib
    c = Case(name)
    aln = _lib.parse_sam(c.sam)
    got, n = hip_fasta(eng, [aln], [c.ref_seq], [c.ctg_id])
    res = oracle.phase_all(c.sam, c.ref_seq, c.ctg_id)
    exp = oracle_lib.consensus(oracle, c.sam, c.ref_seq, res["phased_reads"], res["phased_variants"], c.ctg_id)
    assert got[0] == exp
    assert n == exp.count(b">")


def test_batch_of_contigs_matches_twin(eng, oracle):
    """Several contigs in one batch (block ids restart per contig)."""
    from falcon_unzip_amd import _lib, sim
    alns, refs, ids, exps = [], [], [], []
    for k in range(3):
        hap0, hap1, het, reads = cns_util.diploid_case(40 + k, L=30000 + 7000 * k, n_reads=150 + 40 * k, R=7000)
        ctg = sim.codes_to_str(hap0).encode()
        sam = ("\n".join(sim.sam_lines(reads, "c%d" % k, header=False)) + "\n").encode()
        res = oracle.phase_all(sam, ctg, "c%d" % k)
        exps.append(oracle_lib.consensus(oracle, sam, ctg, res["phased_reads"], res["phased_variants"], "c%d" % k))
        alns.append(_lib.parse_sam(sam)); refs.append(ctg); ids.append("c%d" % k)
    got, n = hip_fasta(eng, alns, refs, ids)
    assert got == exps
    assert n == sum(e.count(b">") for e in exps) and n >= 6


def test_chain_from_raw_reads_recovers_haplotypes(eng):
    """K1 aligns the raw reads, K2..K5 phase them, K6 calls the consensus: each tig must be one of the two true
    haplotypes over its span (>= 99.5 % identical) and clearly not the other."""
    from falcon_unzip_amd import _lib, sim
    rng = np.random.Generator(np.random.PCG64(77))
    L = 80000
    hap0, hap1, het = sim.make_diploid(L, rng, het_rate=1.0 / 400)
    reads = sim.simulate_reads(hap0, hap1, 520, 9000, rng, strand_mix=0.5)
    ctg = sim.codes_to_str(hap0).encode()
    raw = [sim.codes_to_str(r.raw_seq_codes()).encode() for r in reads]
    job = _lib.align_job(eng, [ctg], raw)
    job.run()
    b = job.to_batch()
    b.run(_lib.STAGE_ALL)
    t = b.consensus()
    truth = [sim.codes_to_str(hap0).encode(), sim.codes_to_str(hap1).encode()]
    assert len(t.tigs) >= 2
    covered = 0
    for i, tig in enumerate(t.tigs):
        lo, hi = int(tig["lo"]), int(tig["hi"])
        s = t.sequence(i)
        d = [cns_util.banded_edit_distance(s, tr[lo:hi + 1]) for tr in truth]
        n_het = int(((het >= lo) & (het <= hi)).sum())
        assert min(d) <= 0.005 * (hi - lo + 1), (tig, d)
        assert max(d) >= min(d) + 0.6 * n_het, (tig, d, n_het)
        covered += hi - lo + 1
    assert covered >= 2 * 0.8 * L            # both phases of blocks spanning most of the contig
    t.close(); b.close(); job.close()


def _indel_case(eng, hp_bias, seed=81, L=80000, per_hap=270):
    from falcon_unzip_amd import _lib, sim
    rng = np.random.Generator(np.random.PCG64(seed))
    hap0, hap1, map01, events = sim.make_diploid_indels(L, rng)
    raw = [r[1] for r in sim.simulate_raw_reads_from(hap0, per_hap, 9000, rng, hp_bias=hp_bias, name_prefix="a")]
    raw += [r[1] for r in sim.simulate_raw_reads_from(hap1, per_hap, 9000, rng, hp_bias=hp_bias, name_prefix="b")]
    ctg = sim.codes_to_str(hap0).encode()
    job = _lib.align_job(eng, [ctg], raw)
    job.run()
    b = job.to_batch()
    b.run(_lib.STAGE_ALL)
    t0 = sim.codes_to_str(hap0).encode()
    t1 = sim.codes_to_str(hap1).encode()
    res = {}
    for ver in (1, 2, 3):
        t = b.consensus(version=ver)
        tot = err = 0
        for i, tig in enumerate(t.tigs):
            lo, hi = int(tig["lo"]), int(tig["hi"])
            s = t.sequence(i)
            d = min(cns_util.banded_edit_distance(s, t0[lo:hi + 1]), cns_util.banded_edit_distance(s, t1[int(map01[lo]):int(map01[hi]) + 1]))
            tot += hi - lo + 1
            err += d
        res[ver] = (tot, err, len(t.tigs))
        t.close()
    b.close(); job.close()
    return res, events


def test_multi_base_indel_hets_are_recovered(eng):
    """hets that are 2-5 base insertions / deletions: v1 (one inserted base) cannot spell them; v2 (prefix-linked majorities) spells those the noisy
    reads agree on letter for letter; v3 (length by the pile's median first, then the bases) also gets the ones every read spells a little differently"""
    res, events = _indel_case(eng, hp_bias=1.0)
    (tot1, err1, n1), (tot2, err2, n2), (tot3, err3, n3) = res[1], res[2], res[3]
    n_ins_bases = sum(n - 1 for _, kind, n in events if kind == "ins")
    assert n3 >= 2 and tot3 >= 100000
    assert err3 <= 8 and err3 < err2 < err1, res           # 7 errors in 131 636 consensus bases (v2: 18, v1: 28)
    assert err1 - err2 >= 0.15 * n_ins_bases, (res, n_ins_bases)


def test_homopolymer_biased_errors(eng):
    """non-iid errors: indels 3x as likely inside homopolymer runs, inserted bases repeat the run's base; v3's length vote must not make more of them than v2 did"""
    res, _ = _indel_case(eng, hp_bias=3.0, seed=82)
    tot3, err3, n3 = res[3]
    assert n3 >= 2 and err3 <= 0.002 * tot3 and err3 <= res[2][1], res           # >= 99.8 %


def test_v2_still_matches_its_twin(eng, oracle):
    from falcon_unzip_amd import _lib
    c = Case("g2_cfg1_clr")
    b = eng.batch([_lib.parse_sam(c.sam)], [c.ref_seq])
    b.run(_lib.STAGE_ALL)
    t = b.consensus(version=2)
    res = oracle.phase_all(c.sam, c.ref_seq, c.ctg_id)
    assert t.fasta(0, c.ctg_id) == oracle_lib.consensus(oracle, c.sam, c.ref_seq, res["phased_reads"], res["phased_variants"], c.ctg_id, version=2)
    t.close(); b.close()


def test_v1_still_matches_its_twin(eng, oracle):
    from falcon_unzip_amd import _lib
    c = Case("g2_cfg1_clr")
    b = eng.batch([_lib.parse_sam(c.sam)], [c.ref_seq])
    b.run(_lib.STAGE_ALL)
    t = b.consensus(version=1)
    res = oracle.phase_all(c.sam, c.ref_seq, c.ctg_id)
    assert t.fasta(0, c.ctg_id) == oracle_lib.consensus(oracle, c.sam, c.ref_seq, res["phased_reads"], res["phased_variants"], c.ctg_id, version=1)
    t.close(); b.close()


def test_packed_tally_equals_the_run_length_tally_on_indel_hets(eng, monkeypatch):
    """r5: fzcns v3 tallies a batch that came from K1 in K1's own form -- 2-bit op streams, 2-bit reads (k_cns_tiles_pk) -- where it used to have the run-length CIGAR words and
    byte SEQ made first.  Reads over a genome with multi-base indel hets and homopolymer-biased errors (insertion runs of every length, on and across word boundaries, on both
    strands): the same tigs, byte for byte, as the run-length tally (FZP_K6_BYTES) of the same batch."""
    from falcon_unzip_amd import _lib, sim
    rng = np.random.Generator(np.random.PCG64(91))
    L = 90000
    hap0, hap1, map01, events = sim.make_diploid_indels(L, rng)
    raw = [r[1] for r in sim.simulate_raw_reads_from(hap0, 300, 9000, rng, hp_bias=True, name_prefix="a")]
    raw += [r[1] for r in sim.simulate_raw_reads_from(hap1, 300, 9000, rng, hp_bias=True, name_prefix="b")]
    ctg = sim.codes_to_str(hap0).encode()
    job = _lib.align_job(eng, [ctg], raw)
    job.run()
    fa = {}
    for form in ("packed", "bytes"):
        if form == "bytes":
            monkeypatch.setenv("FZP_K6_BYTES", "1")
        b = job.to_batch()
        b.run(_lib.STAGE_ALL)
        t = b.consensus()
        fa[form] = t.fasta(0, "c")
        assert len(t.tigs) >= 2
        t.close(); b.close()
    monkeypatch.delenv("FZP_K6_BYTES")
    assert fa["packed"] == fa["bytes"] and len(fa["packed"]) > 100000
    job.close()
