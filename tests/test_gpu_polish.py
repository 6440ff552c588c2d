"""fzp_polish_tigs: a tig as the template (the consensus role of run_quiver.py:82-97) -- K1 aligns a tig's routed reads to it, K6's packed tally (fzcns v3) calls the
whole tig as one pile.  HIP == twin (oracle/cns_oracle.c: orc_polish over the device's own alignment records) byte for byte, and what it is for: haplotigs and the
primary tig the layout spells from error-carrying p-reads come out >= 99.95 % identical to their true haplotype and stay away from the other one."""
import os

import numpy as np
import pytest

from tests import cns_util, oracle_lib

pytestmark = pytest.mark.gpu

LAYOUT = [("000000F", [("hom", 20000), ("het", 30000), ("hom", 15000), ("het", 25000), ("hom", 20000)])]


@pytest.fixture(scope="module")
def eng():
    from falcon_unzip_amd import _lib
    e = _lib.Engine(0)
    yield e
    e.close()


def _layout(work, pread_sub):
    """the haplotig layout (graphs_to_h_tigs.py) on a simulated locus -> loci, [(tig name, sequence)]: the primary tig first, then the haplotigs"""
    from falcon_unzip_amd import graphs_to_h_tigs, polish_tigs, sim_asm
    loci = sim_asm.make_case(work, 4242, LAYOUT, pread_sub=pread_sub)
    cwd = os.getcwd()
    os.chdir(work)
    try:
        graphs_to_h_tigs.main(["fc_graphs_to_h_tigs.py", "--fc_asm_path", "2-asm-falcon", "--fc_hasm_path", "1-hasm", "--ctg_id", "all", "--rid_phase_map", "rid_to_phase.all",
                               "--fasta", "preads4falcon.fasta"])
    finally:
        os.chdir(cwd)
    ctg = loci[0]["ctg_id"]
    tigs = polish_tigs.read_fasta(os.path.join(work, ctg, "p_ctg.%s.fa" % ctg)) + polish_tigs.read_fasta(os.path.join(work, ctg, "h_ctg_all.%s.fa" % ctg))
    return loci, tigs


def _indels(seq: bytes, rate, rng):
    """p-read indels the layout would have carried into the tig (sim_asm's p-reads can only carry substitutions: the graph's edges are read offsets)"""
    out = bytearray()
    for ch in seq:
        u = rng.random()
        if u < rate / 2:
            continue                                      # a deleted base
        out.append(ch)
        if u > 1.0 - rate / 2:
            out.append(b"ACGT"[int(rng.integers(0, 4))])  # an inserted one
    return bytes(out)


def _case(tmp_path):
    from falcon_unzip_amd import sim
    loci_t, truth = _layout(str(tmp_path / "clean"), 0.0)          # the same locus, error-free p-reads: what every tig truly is
    loci, draft = _layout(str(tmp_path / "noisy"), 0.0025)          # p-reads with 0.25 % substitutions ...
    rng = np.random.Generator(np.random.PCG64(99))
    draft = [(nm, _indels(s, 0.0025, rng)) for nm, s in draft]      # ... and 0.25 % indels: 0.5 % in all
    assert [n for n, _ in truth] == [n for n, _ in draft] and len(draft) == 3
    loc = loci_t[0]
    hapA, hapB = sim.codes_to_str(loc["hapA"]).encode(), sim.codes_to_str(loc["hapB"]).encode()
    # where every true tig lies on its haplotype
    spans = []
    for k, (nm, s) in enumerate(truth):
        hap = hapA if k == 0 else hapB
        at = hap.find(s)
        assert at >= 0
        spans.append((at, at + len(s)))
    # 30 x of CLR reads per haplotype, both strands; routed the way the tracker would: haplotype-B reads that lie mostly inside a haplotig's span go to it, the rest home
    R, L = 9000, loc["L"]
    n = int(30 * L / R)
    reads, read_tig = [], []
    for h, hap in enumerate((loc["hapA"], loc["hapB"])):
        for r in sim.simulate_reads(hap, hap, n, R, rng, strand_mix=0.5, name_prefix="hap%d" % h):
            tig = 0
            if h == 1:
                for k in (1, 2):
                    a, b = spans[k]
                    if min(r.start + R, b) - max(r.start, a) >= R // 2:
                        tig = k
                if tig == 0 and any(min(r.start + R, b) > max(r.start, a) for a, b in spans[1:]):
                    continue                                  # (straddles a haplotig's end: phased away from the primary, too little of it inside the haplotig)
            reads.append((r.name, sim.codes_to_str(r.raw_seq_codes()).encode()))
            read_tig.append(tig)
    return loc, truth, draft, spans, hapA, hapB, reads, np.asarray(read_tig, np.int32)


def test_polished_tigs_equal_the_twin_and_recover_their_haplotypes(eng, oracle, tmp_path):
    from falcon_unzip_amd import _lib, polish_tigs
    loc, truth, draft, spans, hapA, hapB, reads, read_tig = _case(tmp_path)
    res, t = polish_tigs.polish(eng, draft, reads, read_tig)
    assert [r[0] for r in res] == [n for n, _ in draft]
    # ---- HIP == twin: the same K1 records (an alignment job of its own, same inputs: K1 is deterministic) as SAM text through orc_polish
    job = _lib.align_job(eng, [s for _, s in draft], [s for _, s in reads], read_tig)
    job.run()
    names = [nm for nm, _ in reads]
    for c, (nm, seq, n_rec) in enumerate(res):
        aln, idx = job.alnset(c, names)
        assert n_rec == aln.n_rec and n_rec > 50
        sam = _lib.format_sam(aln, nm)
        assert seq == oracle_lib.polish(oracle, sam, draft[c][1]), "tig %s: HIP != twin" % nm
    job.close()
    # ---- what it is for
    for c, (nm, seq, n_rec) in enumerate(res):
        true = truth[c][1]
        a, b = spans[c]
        other = (hapB if c == 0 else hapA)[a:b]
        n_het = sum(x != y for x, y in zip(true, other))
        d_draft = cns_util.banded_edit_distance(draft[c][1], true, band=400)
        d_true = cns_util.banded_edit_distance(seq, true, band=400)
        d_other = cns_util.banded_edit_distance(seq, other, band=400)
        assert d_draft >= 0.003 * len(true), (nm, d_draft)                       # the draft did carry its 0.5 %
        assert d_true <= 0.0005 * len(true), (nm, d_true, len(true))             # >= 99.95 % identical to its own haplotype
        assert n_het >= 40 and d_other >= 0.6 * n_het, (nm, d_other, n_het)      # and not pulled over to the other one
    t.close()


def test_tigs_without_reads_and_empty_calls(eng):
    """A tig no read aligns to comes back upper-cased and unchanged with n_records 0; no reads at all is not an error."""
    from falcon_unzip_amd import polish_tigs, sim
    rng = np.random.Generator(np.random.PCG64(5))
    t0 = sim.codes_to_str(rng.integers(0, 4, 30000, dtype=np.uint8)).encode()
    t1 = sim.codes_to_str(rng.integers(0, 4, 12000, dtype=np.uint8)).encode()
    rd = [("r%d" % i, t0[s:s + 6000]) for i, s in enumerate(range(0, 24000, 1500))]
    res, t = polish_tigs.polish(eng, [("a", t0), ("b", t1.lower())], rd, [0] * len(rd))
    assert res[0][1] == t0 and res[0][2] == len(rd)          # error-free reads: the call is the template
    assert res[1] == ("b", t1, 0)
    t.close()
    res, t = polish_tigs.polish(eng, [("a", t0[:5000])], [], [])
    assert res == [("a", t0[:5000], 0)]
    t.close()


def test_cli(eng, tmp_path):
    """scripts/fc_polish_tigs.py: --ref_fasta / --read_fasta / --cns_fasta (gzip by name), the reference task's own input and output names (run_quiver.py:64-68)"""
    import gzip
    import subprocess
    import sys
    from falcon_unzip_amd import polish_tigs, sim
    rng = np.random.Generator(np.random.PCG64(6))
    hap = rng.integers(0, 4, 56000, dtype=np.uint8)
    true = sim.codes_to_str(hap[8000:48000]).encode()          # (the tig lies inside what the reads cover: its ends see full coverage, as a tig inside a genome does)
    draft = _indels(true, 0.004, rng)
    reads = sim.simulate_reads(hap, hap, 210, 8000, rng, strand_mix=0.5)
    with open(tmp_path / "tig_ref.fa", "wb") as f:
        f.write(b">000000F_001 some words\n" + b"\n".join(draft[i:i + 70] for i in range(0, len(draft), 70)) + b"\n")
    with open(tmp_path / "reads.fa", "wb") as f:
        for r in reads:
            f.write((">%s\n%s\n" % (r.name, sim.codes_to_str(r.raw_seq_codes()))).encode())
    repo = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    subprocess.check_call([sys.executable, os.path.join(repo, "scripts", "fc_polish_tigs.py"), "--ref_fasta", str(tmp_path / "tig_ref.fa"), "--read_fasta", str(tmp_path / "reads.fa"),
                           "--cns_fasta", str(tmp_path / "cns-000000F_001.fasta.gz")], cwd=str(tmp_path), env=dict(os.environ, PYTHONPATH=repo))
    with gzip.open(tmp_path / "cns-000000F_001.fasta.gz", "rb") as f:
        hdr, seq = f.read().split(b"\n")[:2]
    assert hdr.startswith(b">000000F_001|fzcns ")
    assert cns_util.banded_edit_distance(seq, true, band=300) <= 0.0005 * len(true)
    assert polish_tigs.read_fasta(str(tmp_path / "tig_ref.fa"))[0] == ("000000F_001", draft)
