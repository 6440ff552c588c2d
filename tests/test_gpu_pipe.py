"""The library's own multi-contig pipeline (fzp_job_phase_write / fzp_phase_contigs, csrc/fzp_pipe.hip) and the device text
serialiser (csrc/fzp_text.hip): every file byte-identical to the per-contig path (host serialisers, fzp_readmap), whatever
the grouping of contigs and the number of lanes; two engines in one process stay independent."""
import os

import numpy as np
import pytest

from tests import golden_util

pytestmark = pytest.mark.gpu
FILES = ("het_call/variant_pos", "het_call/variant_map", "het_call/q_id_map", "g_atable/atable", "get_phased_blocks/phased_variants", "phased_reads")


@pytest.fixture(scope="module")
def eng():
    from falcon_unzip_amd import _lib
    e = _lib.Engine(0)
    yield e
    e.close()


def _make_job(n_ctg=4, seed=300, n_reads=150, R=6000):
    from falcon_unzip_amd import sim
    contigs, blobs, names, read_ctg, ids = [], [], [], [], []
    for c in range(n_ctg):
        rng = np.random.Generator(np.random.PCG64(seed + c))
        L = 30000 + 4000 * c
        hap0, hap1, _ = sim.make_diploid(L, rng, het_rate=1.0 / 300)
        reads = sim.simulate_reads(hap0, hap1, n_reads - 10 * c, R, rng, strand_mix=0.5, name_prefix="m%d" % c)
        contigs.append(sim.codes_to_str(hap0).encode() if c != 1 else sim.codes_to_str(hap0).lower().encode())     # one contig in lower case
        for r in reads:
            blobs.append(sim.codes_to_str(r.raw_seq_codes()).encode())
            names.append(r.name)
            read_ctg.append(c)
        ids.append("%06dF" % c)
    off = np.zeros(len(blobs) + 1, np.int64)
    off[1:] = np.cumsum([len(b) for b in blobs])
    return contigs, b"".join(blobs), off, np.array(read_ctg, np.int32), names, ids


def _read_maps(names, read_ctg, ids):
    """raw read i == pread i; pread_ids carries raw id * 10 in its second field (phasing_readmap.py:20-23)"""
    rawread_ids = "\n".join(names) + "\n"
    pread_ids = "\n".join("pread/%d/0_100" % (10 * i + 3) for i in range(len(names))) + "\n"
    rows = []
    for i, c in enumerate(read_ctg):
        rows.append("%09d %s 12 0 100 1" % (i, ids[c]))
        if i % 7 == 0:
            rows.append("%09d %s 3 1 100 1" % (i, ids[(c + 1) % len(ids)]))      # a second-best hit elsewhere: rank 1, ignored
    return rawread_ids.encode(), pread_ids.encode(), ("\n".join(rows) + "\n").encode()


def _legacy(eng, contigs, blob, off, read_ctg, names, ids, maps):
    """per-contig path: alnset + host serialisers + fzp_readmap"""
    from falcon_unzip_amd import _lib
    job = _lib.align_job_raw(eng, contigs, blob, off, read_ctg)
    job.run()
    b = job.to_batch()
    b.run(_lib.STAGE_ALL)
    res = b.results()
    out, recs = {}, []
    for c, ctg in enumerate(ids):
        aln, idx = job.alnset(c, names)
        qoff, qn = aln.qname_table()
        r = res[c]
        t = {"het_call/variant_pos": _lib.format_variant_pos(r.sites), "het_call/variant_map": _lib.format_variant_map(r.sites, r.vmap_qid),
             "het_call/q_id_map": _lib.format_q_id_map(aln), "g_atable/atable": _lib.format_atable(r.sites, r.arows),
             "get_phased_blocks/phased_variants": _lib.format_phased_variants(r.sites, r.pvars),
             "phased_reads": _lib.format_phased_reads(r.preads, ctg, qoff, qn)}
        rr, text = _lib.readmap(t["phased_reads"], maps[0], maps[1], maps[2], ctg, c)
        t["rid_to_phase.%s" % ctg] = text
        recs.append(rr)
        out[ctg] = t
    b.close()
    job.close()
    return out, np.concatenate(recs)


def _check_tree(root, exp, ids):
    for ctg in ids:
        for rel, data in exp[ctg].items():
            with open(os.path.join(root, ctg, rel), "rb") as f:
                got = f.read()
            assert got == data, (ctg, rel, len(got), len(data))


def test_device_text_equals_host_serialisers(eng):
    from falcon_unzip_amd import _lib
    for name in golden_util.cases()[:6]:
        case = golden_util.Case(name)
        aln = _lib.parse_sam(case.sam)
        b = eng.batch([aln], [case.ref_seq])
        b.run(_lib.STAGE_ALL)
        r = b.result(0)
        vm, vb = b.text(_lib.TEXT_VARIANT_MAP)
        at, ab = b.text(_lib.TEXT_ATABLE)
        assert vm == _lib.format_variant_map(r.sites, r.vmap_qid), name
        assert at == _lib.format_atable(r.sites, r.arows), name
        assert list(vb) == [0, len(vm)] and list(ab) == [0, len(at)]
        b.close()


def test_job_phase_write_equals_per_contig_path(eng, tmp_path):
    from falcon_unzip_amd import _lib
    contigs, blob, off, read_ctg, names, ids = _make_job()
    maps = _read_maps(names, read_ctg, ids)
    exp, exp_recs = _legacy(eng, contigs, blob, off, read_ctg, names, ids, maps)
    job = _lib.align_job_raw(eng, contigs, blob, off, read_ctg)
    stats, recs = job.phase_write(ids, names=names, out_dir=str(tmp_path / "a"), read_maps=maps, consensus=True)
    job.close()
    _check_tree(str(tmp_path / "a"), exp, ids)
    assert np.array_equal(recs, exp_recs) and len(recs) == len(names)
    assert stats["n_groups"] == 1 and stats["n_preads"] > 300 and stats["bytes_written"] > 100000
    for ctg in ids:
        assert os.path.getsize(os.path.join(str(tmp_path / "a"), ctg, "cns", "phased_blocks.fa")) > 1000


@pytest.mark.parametrize("lanes,group_bases", [(1, 0), (2, 1), (2, 1_500_000), (3, 900_000)])
def test_phase_contigs_groups_and_lanes(eng, tmp_path, lanes, group_bases):
    """host buffers in, files out: one group, one contig per group, two contigs per group -- on 1 to 3 lanes"""
    from falcon_unzip_amd import _lib
    contigs, blob, off, read_ctg, names, ids = _make_job()
    maps = _read_maps(names, read_ctg, ids)
    exp, exp_recs = _legacy(eng, contigs, blob, off, read_ctg, names, ids, maps)
    out = str(tmp_path / "o")
    stats, recs = _lib.phase_contigs(eng, contigs, blob, off, read_ctg, ids, names=names, out_dir=out, read_maps=maps, n_lanes=lanes, group_bases=group_bases)
    _check_tree(out, exp, ids)
    assert np.array_equal(recs, exp_recs)
    assert stats["n_reads"] == len(names) and stats["n_groups"] == (4 if group_bases == 1 else 1 if group_bases == 0 else stats["n_groups"])
    assert stats["dp_cells"] > 0


def test_phase_contigs_interleaved_reads_and_empty_contig(eng, tmp_path):
    """reads of the contigs interleaved in the input (gathered per group), a contig without reads, no read maps, no names"""
    from falcon_unzip_amd import _lib
    contigs, blob, off, read_ctg, names, ids = _make_job(n_ctg=3)
    perm = np.random.Generator(np.random.PCG64(5)).permutation(len(read_ctg))
    reads = [blob[off[i]:off[i + 1]] for i in perm]
    off2 = np.zeros(len(reads) + 1, np.int64)
    off2[1:] = np.cumsum([len(r) for r in reads])
    blob2, ctg2 = b"".join(reads), read_ctg[perm]
    contigs.append(b"ACGT" * 2000)
    ids.append("999999F")
    job = _lib.align_job_raw(eng, contigs, blob2, off2, ctg2)
    st1, _ = job.phase_write(ids, out_dir=str(tmp_path / "a"))
    job.close()
    st2, recs = _lib.phase_contigs(eng, contigs, blob2, off2, ctg2, ids, out_dir=str(tmp_path / "b"), n_lanes=2, group_bases=1)
    assert len(recs) == 0 and st2["n_groups"] == 4
    for ctg in ids:
        for rel in FILES:
            with open(os.path.join(str(tmp_path / "a"), ctg, rel), "rb") as f, open(os.path.join(str(tmp_path / "b"), ctg, rel), "rb") as g:
                assert f.read() == g.read(), (ctg, rel)
    assert os.path.getsize(os.path.join(str(tmp_path / "b"), "999999F", "phased_reads")) == 0
    with open(os.path.join(str(tmp_path / "b"), ids[0], "het_call", "q_id_map"), "rb") as f:
        assert f.readline().split()[1].startswith(b"read/")


def test_two_engines_in_one_process(oracle):
    """Two contexts on the same device, used alternately: each has its own block caches and staging; results of
    batches of both stay valid side by side."""
    from falcon_unzip_amd import _lib
    e1, e2 = _lib.Engine(0), _lib.Engine(0)
    c1, c2 = golden_util.Case(golden_util.cases()[0]), golden_util.Case(golden_util.cases()[1])
    a1, a2 = _lib.parse_sam(c1.sam), _lib.parse_sam(c2.sam)
    b1 = e1.batch([a1], [c1.ref_seq])
    b2 = e2.batch([a2], [c2.ref_seq])
    b1.run(_lib.STAGE_ALL)
    b2.run(_lib.STAGE_ALL)
    r1 = b1.results(copy=False)[0]
    r2 = b2.results(copy=False)[0]                        # second batch: must not disturb the first one's borrowed views
    b3 = e1.batch([a2], [c2.ref_seq])                     # a third batch on the first engine, run while b1's views are alive
    b3.run(_lib.STAGE_ALL)
    r3 = b3.results(copy=False)[0]
    c1.check("variant_pos", _lib.format_variant_pos(r1.sites))
    c1.check("atable", _lib.format_atable(r1.sites, r1.arows))
    c2.check("variant_pos", _lib.format_variant_pos(r2.sites))
    c2.check("atable", _lib.format_atable(r3.sites, r3.arows))
    for b in (b1, b2, b3):
        b.close()
    e2.close()
    b4 = e1.batch([a1], [c1.ref_seq])                     # the first engine keeps working after the second is gone
    b4.run(_lib.STAGE_ALL)
    c1.check("variant_map", _lib.format_variant_map(b4.result(0).sites, b4.result(0).vmap_qid))
    b4.close()
    e1.close()


def test_async_writes_are_complete_after_flush(eng, tmp_path):
    """FZP_PIPE_ASYNC_WRITES: the call returns with the writes queued; after fzp_pipe_flush the trees equal the synchronous ones"""
    from falcon_unzip_amd import _lib
    contigs, blob, off, read_ctg, names, ids = _make_job(n_ctg=3)
    maps = _read_maps(names, read_ctg, ids)
    job = _lib.align_job_raw(eng, contigs, blob, off, read_ctg)
    job.phase_write(ids, names=names, out_dir=str(tmp_path / "sync"), read_maps=maps)
    for k in range(3):                                     # several calls in flight share the writer threads
        st, recs = job.phase_write(ids, names=names, out_dir=str(tmp_path / ("async%d" % k)), read_maps=maps, async_writes=True)
    eng.pipe_flush()
    job.close()
    for k in range(3):
        for ctg in ids:
            for rel in FILES + ("rid_to_phase.%s" % ctg,):
                with open(os.path.join(str(tmp_path / "sync"), ctg, rel), "rb") as f, open(os.path.join(str(tmp_path / ("async%d" % k)), ctg, rel), "rb") as g:
                    assert f.read() == g.read(), (k, ctg, rel)
    assert st["bytes_written"] > 100000
    with pytest.raises(_lib.FzpError):                     # an unwritable target surfaces at the flush
        job2 = _lib.align_job_raw(eng, contigs, blob, off, read_ctg)
        job2.phase_write(ids, names=names, out_dir="/proc/no_such_dir/x", read_maps=maps, async_writes=True)
        eng.pipe_flush()


def test_a_rerun_into_an_existing_tree_leaves_the_files_of_a_first_run(eng, tmp_path, monkeypatch):
    """r6: a file that exists is overwritten in place and cut to its new length afterwards (no O_TRUNC: the old pages are reused).  A tree whose files are LONGER than the new
    ones (and hold other bytes), a tree whose files are shorter, and an untouched one all end up byte-identical (sync and async writes)."""
    from falcon_unzip_amd import _lib
    contigs, blob, off, read_ctg, names, ids = _make_job(n_ctg=2)
    maps = _read_maps(names, read_ctg, ids)
    job = _lib.align_job_raw(eng, contigs, blob, off, read_ctg)
    first = str(tmp_path / "first")
    job.phase_write(ids, names=names, out_dir=first, read_maps=maps, consensus=True)
    files = {}
    for ctg in ids:
        for d, _, fs in os.walk(os.path.join(first, ctg)):
            for f in fs:
                rel = os.path.relpath(os.path.join(d, f), first)
                files[rel] = open(os.path.join(first, rel), "rb").read()
    assert len(files) >= 2 * 8 and sum(map(len, files.values())) > 100000
    for tag, make in (("longer", lambda b: b"#" * (len(b) + 4097) + b"tail"), ("shorter", lambda b: b[: len(b) // 3]), ("same", lambda b: bytes(len(b)))):
        root = str(tmp_path / tag)
        for rel, data in files.items():
            os.makedirs(os.path.dirname(os.path.join(root, rel)), exist_ok=True)
            with open(os.path.join(root, rel), "wb") as f:
                f.write(make(data))
        for asyn in (False, True):
            job.phase_write(ids, names=names, out_dir=root, read_maps=maps, consensus=True, async_writes=asyn)
            eng.pipe_flush()
            for rel, data in files.items():
                assert open(os.path.join(root, rel), "rb").read() == data, (tag, asyn, rel)
    job.close()


def test_pipeline_edge_cases(eng, tmp_path):
    """no reads at all; a malformed read map (the reference raises while reading it, whatever the contig); an unwritable target"""
    from falcon_unzip_amd import _lib
    contigs, blob, off, read_ctg, names, ids = _make_job(n_ctg=2)
    # a job without reads: every file exists and is empty, no records
    st, recs = _lib.phase_contigs(eng, contigs, b"", np.zeros(1, np.int64), np.zeros(0, np.int32), ids, out_dir=str(tmp_path / "empty"))
    assert len(recs) == 0 and st["n_reads"] == 0
    for ctg in ids:
        for rel in FILES:
            assert os.path.getsize(os.path.join(str(tmp_path / "empty"), ctg, rel)) == 0
    # pread_to_contigs with a one-token row: IndexError in the reference at phasing_readmap.py:41 -> an error here, from the worker threads
    maps = _read_maps(names, read_ctg, ids)
    bad = (maps[0], maps[1], maps[2] + b"lonely\n")
    with pytest.raises(_lib.FzpError) as ei:
        _lib.phase_contigs(eng, contigs, blob, off, read_ctg, ids, names=names, out_dir=str(tmp_path / "bad"), read_maps=bad, n_lanes=2, group_bases=1)
    assert "short row" in str(ei.value)
    # pread id beyond pread_ids
    bad2 = (maps[0], b"only/10/0_1\n", maps[2])
    with pytest.raises(_lib.FzpError):
        _lib.phase_contigs(eng, contigs, blob, off, read_ctg, ids, names=names, out_dir=str(tmp_path / "bad2"), read_maps=bad2)
    with pytest.raises(_lib.FzpError):
        _lib.phase_contigs(eng, contigs, blob, off, read_ctg, ids, names=names, out_dir="/proc/nope/x")
    # the engine still works afterwards
    st, recs = _lib.phase_contigs(eng, contigs, blob, off, read_ctg, ids, names=names, out_dir=str(tmp_path / "ok"), read_maps=maps)
    assert len(recs) == len(names)


@pytest.mark.parametrize("async_writes", [False, True])
def test_bam_and_sentinels_from_the_same_pass(eng, tmp_path, async_writes):
    """FZP_PIPE_BAM | FZP_PIPE_SENTINELS: <ctg>/blasr/<ctg>_sorted.bam(.bai) are the bytes the per-contig route makes (alnset of every aligned
    read + fzp_format_bam), written from the one alignment pass of the call, group by group; the job_done files of the reference's two tasks
    (unzip.py:241,268) with their `.exit` twins (unzip.py:81,120) are there; a contig whose files cannot be written keeps the `.exit` files only."""
    from falcon_unzip_amd import _lib
    contigs, blob, off, read_ctg, names, ids = _make_job()
    maps = _read_maps(names, read_ctg, ids)
    exp, _ = _legacy(eng, contigs, blob, off, read_ctg, names, ids, maps)
    job = _lib.align_job_raw(eng, contigs, blob, off, read_ctg)
    job.run()
    summ = job.summaries()
    want = {}
    enc = [n.encode() for n in names]
    noff = np.zeros(len(enc) + 1, np.int64)
    noff[1:] = np.cumsum([len(e) for e in enc])
    for c, ctg in enumerate(ids):
        aln, idx = job.alnset(c, (noff, b"".join(enc)), all_records=True)
        want[ctg] = _lib.format_bam(aln, ctg, len(contigs[c]), (summ["strand"][idx] * 16).astype(np.int32))
    job.close()
    out = str(tmp_path / "o")
    st, recs = _lib.phase_contigs(eng, contigs, blob, off, read_ctg, ids, names=names, out_dir=out, read_maps=maps, n_lanes=2, group_bases=1_500_000, bam=True, sentinels=True,
                                  async_writes=async_writes)
    eng.pipe_flush()
    assert st["n_groups"] >= 2
    _check_tree(out, exp, ids)
    for ctg in ids:
        base = os.path.join(out, ctg)
        with open(os.path.join(base, "blasr", "%s_sorted.bam" % ctg), "rb") as f:
            assert f.read() == want[ctg][0], ctg
        with open(os.path.join(base, "blasr", "%s_sorted.bam.bai" % ctg), "rb") as f:
            assert f.read() == want[ctg][1], ctg
        for rel in ("blasr/aln_%s_done", "blasr/aln_%s_done.exit", "phasing/p_%s_done", "phasing/p_%s_done.exit"):
            assert os.path.getsize(os.path.join(base, rel % ctg)) == 0
    # sentinels without the BAM: only the phasing task's
    out2 = str(tmp_path / "p")
    _lib.phase_contigs(eng, contigs, blob, off, read_ctg, ids, names=names, out_dir=out2, read_maps=maps, sentinels=True, async_writes=async_writes)
    eng.pipe_flush()
    assert os.path.exists(os.path.join(out2, ids[0], "phasing", "p_%s_done" % ids[0])) and not os.path.exists(os.path.join(out2, ids[0], "blasr"))
    # a contig whose het_call directory cannot be made (a FILE sits in its place): its task fails, `.exit` only; the error names the directory
    out3 = str(tmp_path / "q")
    os.makedirs(os.path.join(out3, ids[1]))
    with open(os.path.join(out3, ids[1], "het_call"), "w") as f:
        f.write("in the way\n")
    with pytest.raises(_lib.FzpError) as ei:
        _lib.phase_contigs(eng, contigs, blob, off, read_ctg, ids, names=names, out_dir=out3, read_maps=maps, bam=True, sentinels=True, async_writes=async_writes)
        eng.pipe_flush()
    assert ids[1] in str(ei.value) and "Success" not in str(ei.value)
    assert os.path.exists(os.path.join(out3, ids[1], "phasing", "p_%s_done.exit" % ids[1])) and not os.path.exists(os.path.join(out3, ids[1], "phasing", "p_%s_done" % ids[1]))
    assert os.path.exists(os.path.join(out3, ids[1], "blasr", "aln_%s_done" % ids[1]))          # the BAM itself was fine


def test_device_fasta_reader_equals_the_host_reader_on_hostile_files(eng, tmp_path, monkeypatch):
    """r6: fzp_phase_contigs_files reads its files as they are and finds the records ON THE DEVICE (csrc/fzp_fasta.hip).  The same hostile directory the host reader is held
    against the Python reader on (tests/test_host_logic.py: CRLF, blank lines, wrapped and one-line records, text before the first header, no newline at the end, empty
    files, several / no records of the contig's name): record for record what the host reader makes of it -- contigs, reads, names, the contig each read belongs to."""
    from falcon_unzip_amd import _lib
    from tests.test_host_logic import _load_group, hostile_fasta_dir
    d = tmp_path / "reads"
    ids, ref_a = hostile_fasta_dir(d)
    exp_ctgs, exp_reads = _load_group(_lib, str(d), ids, 3)
    for threads, piece in ((0, None), (2, "4099"), (5, "61")):
        if piece:
            monkeypatch.setenv("FZP_FASTA_PIECE", piece)       # (the pieces the host threads pread: smaller than a line, than a header)
        ctgs, reads = _load_group(_lib, str(d), ids, threads, eng=eng)
        assert ctgs == exp_ctgs, threads
        assert len(reads) == len(exp_reads) > 2200 and reads == exp_reads, threads
    assert ctgs[0] == ref_a and ctgs[2] == b"" and ctgs[3] == b""
    monkeypatch.delenv("FZP_FASTA_PIECE")
    with pytest.raises(_lib.FzpError) as e:
        _load_group(_lib, str(d), ids + ["999999F"], eng=eng)
    assert "999999F" in str(e.value)
