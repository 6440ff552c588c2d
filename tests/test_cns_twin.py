"""The consensus twin (oracle/cns_oracle.c, "fzcns v1") on its own: CPU only.  Parity is unpinned (the reference has
no consensus code), so what is asserted is accuracy against the simulator's true haplotypes and the output format."""
import numpy as np

from tests import cns_util, oracle_lib


def test_twin_recovers_the_haplotypes(oracle):
    from falcon_unzip_amd import sim
    hap0, hap1, het, reads = cns_util.diploid_case(31)
    ctg = sim.codes_to_str(hap0).encode()
    sam = ("\n".join(sim.sam_lines(reads, "ctg", header=False)) + "\n").encode()
    res = oracle.phase_all(sam, ctg, "ctg")
    fa = oracle_lib.consensus(oracle, sam, ctg, res["phased_reads"], res["phased_variants"], "ctg").decode().splitlines()
    assert len(fa) >= 4 and len(fa) % 2 == 0
    truth = [sim.codes_to_str(hap0).encode(), sim.codes_to_str(hap1).encode()]
    seen = set()
    for h, s in zip(fa[0::2], fa[1::2]):
        name, lo, hi, n = h[1:].split()
        ctg_id, blk, ph = name.rsplit("_", 2)
        lo, hi, n = int(lo), int(hi), int(n)
        assert ctg_id == "ctg" and n >= 5 and hi > lo
        seen.add((int(blk), int(ph)))
        d = [cns_util.banded_edit_distance(s.encode(), t[lo - 1:hi]) for t in truth]
        own = min(d)
        span = hi - lo + 1
        n_het = int(((het >= lo - 1) & (het <= hi - 1)).sum())
        assert own <= 0.004 * span, (h, d)                  # >= 99.6 % identical to one haplotype ...
        assert max(d) >= own + 0.6 * n_het, (h, d, n_het)   # ... and clearly not the other one
    assert all((b, 0) in seen and (b, 1) in seen for b, _ in seen)


def test_twin_with_a_template_polishes_a_tig(oracle):
    """orc_polish (the twin of fzp_polish_tigs): the whole tig one pile of every accepted record.  Reads simulated against the TRUE sequence and handed over with their true
    alignments -- to a template that IS the truth -- give the truth back; a tig without records comes back upper-cased and unchanged."""
    from falcon_unzip_amd import sim
    rng = np.random.Generator(np.random.PCG64(12))
    hap = rng.integers(0, 4, 30000, dtype=np.uint8)
    tig = sim.codes_to_str(hap).encode()
    reads = sim.simulate_reads(hap, hap, 120, 8000, rng)
    sam = ("\n".join(sim.sam_lines(reads, "tig", header=False)) + "\n").encode()
    got = oracle_lib.polish(oracle, sam, tig)
    # (reads start uniformly inside the tig: its first and last few hundred bases see little coverage -- the interior is what a pile of 30 x decides; the simulator's own
    #  alignments put an inserted base anywhere inside a run of its kind, so a vote over them is noisier than over an aligner's consistently placed gaps: 1 error per kb here,
    #  none in 2 kb with K1's records, tests/test_gpu_polish.py)
    d0 = cns_util.banded_edit_distance(got[1500:-1500], tig[1500:-1500], band=100)
    assert d0 <= 0.0015 * len(tig), d0
    assert oracle_lib.polish(oracle, b"", tig.lower()) == tig
    # a template that differs from what the reads say is called towards the reads: 40 substitutions planted in the template's interior are gone
    bad = bytearray(tig)
    for p in rng.choice(np.arange(3000, 27000), 40, replace=False):
        bad[p] = b"ACGT"[(b"ACGT".index(bad[p]) + 1) % 4]
    fixed = oracle_lib.polish(oracle, sam, bytes(bad))
    assert cns_util.banded_edit_distance(fixed[1500:-1500], tig[1500:-1500], band=100) <= d0 + 2
