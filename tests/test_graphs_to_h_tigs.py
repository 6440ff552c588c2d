"""falcon_unzip_amd/graphs_to_h_tigs.py (mirror of the reference's haplotig layout) against fixtures produced by
RUNNING the reference (tests/golden_htigs/make_golden_htigs.py), byte for byte; plus what the outputs must mean on the
simulated locus: the primary contig spells haplotype A from the second read on, every haplotig spells haplotype B through one bubble."""
import gzip
import hashlib
import json
import os
import subprocess
import sys

import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(HERE)
GOLD = os.path.join(HERE, "golden_htigs")
CASES = sorted(d for d in os.listdir(GOLD) if os.path.isfile(os.path.join(GOLD, d, "manifest.json")))


def _run(case, tmp_path, via_cli=False):
    from falcon_unzip_amd import graphs_to_h_tigs, sim_asm
    with open(os.path.join(GOLD, case, "manifest.json")) as f:
        man = json.load(f)
    work = str(tmp_path / case)
    loci = sim_asm.make_case(work, man["seed"], [(c, [tuple(s) for s in segs]) for c, segs in man["layouts"]])
    argv = ["fc_graphs_to_h_tigs.py", "--fc_asm_path", "2-asm-falcon", "--fc_hasm_path", "1-hasm", "--ctg_id", "all", "--rid_phase_map", "rid_to_phase.all",
            "--fasta", "preads4falcon.fasta"]
    if via_cli:
        subprocess.check_call([sys.executable, os.path.join(REPO, "scripts", "fc_graphs_to_h_tigs.py")] + argv[1:], cwd=work)
    else:
        cwd = os.getcwd()
        os.chdir(work)
        try:
            graphs_to_h_tigs.main(argv)
        finally:
            os.chdir(cwd)
    return man, work, loci


@pytest.mark.parametrize("case", CASES)
def test_matches_reference_outputs(case, tmp_path):
    man, work, _ = _run(case, tmp_path, via_cli=(case == CASES[0]))
    for ctg, files in man["outputs"].items():
        if files is None:
            assert not os.path.exists(os.path.join(work, ctg)), "contig without phase rows must be skipped (graphs_to_h_tigs.py:668-669)"
            continue
        for fn, meta in files.items():
            with open(os.path.join(work, ctg, fn), "rb") as f:
                got = f.read()
            with gzip.open(os.path.join(GOLD, case, "%s.%s.gz" % (ctg, fn)), "rb") as g:
                exp = g.read()
            assert hashlib.sha256(exp).hexdigest() == meta["sha256"]
            assert got == exp, (case, ctg, fn, got[:200], exp[:200])


def test_tigs_spell_the_haplotypes(tmp_path):
    from falcon_unzip_amd import sim
    man, work, loci = _run("h1_two_bubbles", tmp_path)
    loc = loci[0]
    ctg = loc["ctg_id"]
    with open(os.path.join(work, ctg, "p_ctg.%s.fa" % ctg)) as f:
        name, seq = f.read().split("\n")[:2]
    a0 = loc["a_reads"][0]
    hapA = sim.codes_to_str(loc["hapA"])
    assert name == ">" + ctg and seq == hapA[a0[3]:], "the primary tig = haplotype A after the first read"
    hapB = sim.codes_to_str(loc["hapB"])
    with open(os.path.join(work, ctg, "h_ctg_all.%s.fa" % ctg)) as f:
        rec = f.read().split("\n")
    tigs = {rec[i][1:]: rec[i + 1] for i in range(0, len(rec) - 1, 2)}
    assert len(tigs) == len(loc["bubbles"])
    for (u0, u1), s in zip(loc["bubbles"], tigs.values()):
        at = hapB.find(s)
        assert at >= 0 and at < u0 + 8000 and at + len(s) > u1 and hapA.find(s) < 0     # haplotype B from just after the left hook read to beyond the bubble; differs from A
    with open(os.path.join(work, ctg, "h_ctg_edges.%s" % ctg)) as f:
        for l in f:
            t = l.split()
            assert (t[5], t[6]) == ("-1", "0") or t[6] == "1"          # phase-1 reads or unphased hooks
            assert t[3] == "N" and t[4] == "H"


def test_consumes_rid_to_phase_written_by_the_pipeline_format(tmp_path):
    """rows as fzp_readmap / fzp_phase_contigs write them ('%09d ctg block phase') load into the same table"""
    from falcon_unzip_amd import graphs_to_h_tigs
    p = tmp_path / "rid_to_phase.all"
    p.write_text("000000012 000000F 3 1\n000000013 000000F -1 0\n000000099 000001F 1 0\n")
    table, ids = graphs_to_h_tigs.load_rid_to_phase(str(p))
    assert table == {"000000F": {"000000012": (3, 1), "000000013": (-1, 0)}, "000001F": {"000000099": (1, 0)}} and len(ids) == 3


def test_tied_routes_are_broken_by_the_pinned_rule_not_by_networkx(tmp_path, monkeypatch):
    """h4: a bubble of unphased reads whose branches have the same number of edges -- two source-to-sink routes of equal score.  The mirror's choice
    (least score, fewest edges, smallest predecessor name from the target back) equals the reference's output on networkx 3 (the golden), and it does
    not move when the graph's edges are inserted in another order, which is what networkx's own tie-break follows."""
    import networkx as nx
    from falcon_unzip_amd import graphs_to_h_tigs as g
    man, work, loci = _run("h4_tied_bubble", tmp_path)
    loc = loci[0]
    ctg = loc["ctg_id"]
    assert any(seg[0] == "tie" for seg in man["layouts"][0][1])
    with open(os.path.join(work, ctg, "p_ctg_path.%s" % ctg)) as f:
        primary = [l.split()[1] for l in f]
    a_names = {"%s:E" % r[0] for r in loc["a_reads"]}
    assert set(primary) <= a_names                                   # through the tied bubble the primary tig keeps to the primary assembly's reads
    # the rule itself, on a diamond with equal costs, whatever the insertion order
    for order in (("s", "b", "t", "s", "a", "t"), ("s", "a", "t", "s", "b", "t")):
        G = nx.DiGraph()
        for v, w in zip(order[0:2] + order[3:5], order[1:3] + order[4:6]):
            G.add_edge(v, w, score=1)
        assert g._best_path(G, "s", "t", weight="score") == ["s", "a", "t"]
    G = nx.DiGraph()
    G.add_edge("s", "x", score=2); G.add_edge("x", "t", score=0)    # equal score, more edges than the direct one below: the shorter route wins
    G.add_edge("s", "t", score=2)
    assert g._best_path(G, "s", "t", weight="score") == ["s", "t"]
    with pytest.raises(nx.exception.NetworkXNoPath):
        g._best_path(G, "t", "s", weight="score")
