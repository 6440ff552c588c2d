"""The batched driver (falcon_unzip_amd/pipeline.py) on a miniature 3-unzip tree: FASTA in, the reference's file
layout out; every file compared with the oracle run on the SAM the aligner produced."""
import os
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_unzip_tree_end_to_end(tmp_path, oracle):
    from falcon_unzip_amd import _lib, pipeline, sim
    from tests.golden.make_golden import make_readmap
    unzip = tmp_path / "3-unzip"
    (unzip / "reads").mkdir(parents=True)
    ctgs = ["000001F", "000000F"]                       # deliberately unsorted in ctg_list
    all_reads = {}
    rng0 = np.random.Generator(np.random.PCG64(99))
    for k, ctg in enumerate(ctgs):
        rng = np.random.Generator(np.random.PCG64(100 + k))
        L = 40000 + 5000 * k
        hap0, hap1, _ = sim.make_diploid(L, rng, het_rate=1.0 / 300)
        reads = sim.simulate_reads(hap0, hap1, 180, 7000, rng, strand_mix=0.5, name_prefix="m%d" % k)
        all_reads[ctg] = reads
        sim.write_fasta(str(unzip / "reads" / ("%s_ref.fa" % ctg)), [(ctg + " some description", sim.codes_to_str(hap0).lower())], width=80)
        sim.write_fasta(str(unzip / "reads" / ("%s_reads.fa" % ctg)), [(r.name, sim.codes_to_str(r.raw_seq_codes())) for r in reads])
    (unzip / "reads" / "ctg_list").write_text("".join(c + "\n" for c in ctgs))
    # one shared read_maps tree: raw read ids over both contigs
    flat = [r for c in sorted(ctgs) for r in all_reads[c]]
    rm = make_readmap(flat, "000000F", rng0)
    # make_readmap assigns every pread to one contig; route preads of 000001F's reads to 000001F
    n0 = len(all_reads["000000F"])
    lines = []
    pread_ids = rm["pread_ids"].split("\n")
    for l in rm["pread_to_contigs"].split("\n"):
        if not l:
            continue
        f = l.split()
        raw = int(pread_ids[int(f[0])].split("/")[1]) // 10
        if raw >= n0:
            f[1] = f[1].replace("000000F", "000001F")
        lines.append(" ".join(f))
    rmd = tmp_path / "read_maps"
    (rmd / "dump_rawread_ids").mkdir(parents=True)
    (rmd / "dump_pread_ids").mkdir(parents=True)
    (rmd / "dump_rawread_ids" / "rawread_ids").write_text(rm["rawread_ids"])
    (rmd / "dump_pread_ids" / "pread_ids").write_text(rm["pread_ids"])
    (rmd / "pread_to_contigs").write_text("".join(l + "\n" for l in lines))

    allr = pipeline.run(str(unzip), str(rmd), device=0)

    cat = b""
    for ctg in sorted(ctgs):
        base = unzip / "0-phasing" / ctg
        bam = (base / "blasr" / ("%s_sorted.bam" % ctg)).read_bytes()      # what the blasr task leaves behind (unzip.py:86-91)
        assert (base / "blasr" / ("%s_sorted.bam.bai" % ctg)).read_bytes()[:4] == b"BAI\x01"
        sam = _lib.bam_to_sam(bam, ctg)                                     # `samtools view <bam> <ctg>` (phasing.py:27)
        assert sam.count(b"\n") >= 175
        # what an unchanged fc_unzip.py checks to call the contig's two tasks done (unzip.py:239-241,267-269; `trap ... EXIT` at :81,:120)
        for rel in ("blasr/aln_%s_done", "blasr/aln_%s_done.exit", "phasing/p_%s_done", "phasing/p_%s_done.exit"):
            assert (base / (rel % ctg)).exists() and (base / (rel % ctg)).stat().st_size == 0, rel
        ref = sim.codes_to_str(sim.make_diploid(40000 + 5000 * ctgs.index(ctg), np.random.Generator(np.random.PCG64(100 + ctgs.index(ctg))), het_rate=1.0 / 300)[0]).encode()
        exp = oracle.phase_all(sam, ref, ctg)
        for rel, key in (("het_call/variant_pos", "variant_pos"), ("het_call/variant_map", "variant_map"), ("het_call/q_id_map", "q_id_map"),
                         ("g_atable/atable", "atable"), ("get_phased_blocks/phased_variants", "phased_variants"), ("phased_reads", "phased_reads")):
            assert (base / rel).read_bytes() == exp[key], (ctg, key)
        assert exp["phased_reads"].count(b"\n") > 100
        from tests import oracle_lib
        cns = oracle_lib.consensus(oracle, sam, ref, exp["phased_reads"], exp["phased_variants"], ctg)     # K6 == its twin
        assert (base / "cns" / "phased_blocks.fa").read_bytes() == cns and cns.count(b">") >= 2
        r2p = oracle.phasing_readmap(exp["phased_reads"], rm["rawread_ids"].encode(), rm["pread_ids"].encode(),
                                     (rmd / "pread_to_contigs").read_bytes(), ctg)
        assert (base / ("rid_to_phase.%s" % ctg)).read_bytes() == r2p
        cat += r2p
    assert (unzip / "1-hasm" / "rid-to-phase-all" / "rid_to_phase.all").read_bytes() == cat     # unzip.py:303-314
    # the emitted BAM closes the loop: the per-contig CLI (task_phasing's script, unzip.py:125) run on it without
    # samtools writes the same files the batched pipeline wrote
    ctg = sorted(ctgs)[0]
    again = tmp_path / "again"
    (again / ctg).mkdir(parents=True)
    env = dict(os.environ, PYTHONPATH=REPO + os.pathsep + os.environ.get("PYTHONPATH", ""))
    subprocess.check_call([sys.executable, os.path.join(REPO, "scripts", "fc_phasing.py"), "--bam", str(unzip / "0-phasing" / ctg / "blasr" / ("%s_sorted.bam" % ctg)),
                           "--fasta", str(unzip / "reads" / ("%s_ref.fa" % ctg)), "--ctg_id", ctg, "--base_dir", "..", "--samtools", "builtin"],
                          cwd=str(again / ctg), env=env)
    for rel in ("het_call/variant_pos", "het_call/variant_map", "het_call/q_id_map", "g_atable/atable", "get_phased_blocks/phased_variants", "phased_reads"):
        assert (again / ctg / rel).read_bytes() == (unzip / "0-phasing" / ctg / rel).read_bytes(), rel
    assert len(allr) == cat.count(b"\n")
