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
