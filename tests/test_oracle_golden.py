"""The C oracle (oracle/phasing_oracle.c) against the vectors produced by running the reference.

This is what pins the oracle: every stage is fed the GOLDEN output of the previous stage (so a
mistake cannot hide behind an earlier one) and the chained run is checked as well.
"""
import pytest

from tests.golden_util import Case, cases

CASES = cases()


@pytest.mark.parametrize("name", CASES)
def test_make_het_call(oracle, name):
    c = Case(name)
    vpos, vmap, qmap = oracle.make_het_call(c.sam, c.ref_seq)
    c.check("variant_pos", vpos)
    c.check("variant_map", vmap)
    c.check("q_id_map", qmap)


@pytest.mark.parametrize("name", CASES)
def test_chain(oracle, name):
    c = Case(name)
    out = oracle.phase_all(c.sam, c.ref_seq, c.ctg_id)
    for k in ("variant_pos", "variant_map", "q_id_map", "atable", "phased_variants", "phased_reads"):
        c.check(k, out[k])


@pytest.mark.parametrize("name", CASES)
def test_stages_from_golden_inputs(oracle, name):
    c = Case(name)
    vmap = c.expected("variant_map")
    atable = c.expected("atable")
    if vmap is None:
        pytest.skip("variant_map pinned by hash only")
    got_atable = oracle.generate_association_table(vmap)
    c.check("atable", got_atable)
    atable = atable if atable is not None else got_atable
    pv = oracle.get_phased_blocks(vmap, atable)
    c.check("phased_variants", pv)
    pr = oracle.get_phased_reads(vmap, c.expected("q_id_map"), c.expected("phased_variants"), c.ctg_id)
    c.check("phased_reads", pr)


@pytest.mark.parametrize("name", [n for n in CASES if Case(n).has("rid_to_phase")])
def test_readmap(oracle, name):
    c = Case(name)
    rm = c.readmap_inputs()
    got = oracle.phasing_readmap(c.expected("phased_reads"), rm["rawread_ids"], rm["pread_ids"],
                                 rm["pread_to_contigs"], c.ctg_id)
    c.check("rid_to_phase", got)
