"""Size-independent properties of the whole path at BASELINE-scale shapes (5 Mb contigs, 15 kb reads):
determinism, batch independence, structural invariants of every record type, and byte parity of the
phasing stages against the oracle given the aligner's records."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def eng():
    from falcon_unzip_amd import _lib
    e = _lib.Engine(0)
    yield e
    e.close()


def _make(n_ctg, L, n_reads, R, win, cfg=7):
    from falcon_unzip_amd import sim
    contigs, blobs, offs, rctg, base = [], [], [np.zeros(1, np.int64)], [], 0
    for c in range(n_ctg):
        rng = sim.rng_for(cfg, c)
        hap0, hap1, _ = sim.make_diploid(L, rng)
        lo = int(rng.integers(0, L - win + 1))
        codes, off, *_ = sim.simulate_raw_reads_bulk(hap0, hap1, n_reads, R, rng, lo=lo, hi=lo + win)
        contigs.append(sim.ACGT[hap0].tobytes())
        blobs.append(sim.ACGT[codes].tobytes())
        offs.append(off[1:] + base)
        base += int(off[-1])
        rctg.append(np.full(n_reads, c, np.int32))
    return contigs, b"".join(blobs), np.concatenate(offs), np.concatenate(rctg)


def _check_invariants(r, n_qid):
    s = r.sites
    assert np.all(np.diff(s["pos"]) > 0)
    assert np.all(s["total"] >= 10)
    assert np.all(4 * s["count"][:, 0].astype(np.int64) < 3 * s["total"]) and np.all(4 * s["count"][:, 1].astype(np.int64) > s["total"])
    assert np.all(s["count"][:, 0] >= s["count"][:, 1]) and np.all(s["count"][:, 1] >= s["count"][:, 2])
    rows = s["count"][:, 0].astype(np.int64) + s["count"][:, 1]
    assert np.array_equal(s["row_off"], np.concatenate(([0], np.cumsum(rows)[:-1])))
    assert len(r.vmap_qid) == rows.sum() and (len(r.vmap_qid) == 0 or (r.vmap_qid.min() >= 0 and r.vmap_qid.max() < n_qid))
    a = r.arows
    if len(a):
        assert np.all(a["site1"] < a["site2"])
        key = a["site1"].astype(np.int64) * (1 << 32) + a["site2"]
        assert np.all(np.diff(key) > 0)
        assert np.all(s["pos"][a["site2"]] - s["pos"][a["site1"]] <= 65536)
        assert np.all(a["n"].sum(axis=1) >= 6)
        assert np.bincount(a["site1"]).max() <= 501
    p = r.preads
    if len(p):
        key = p["q_id"].astype(np.int64) * (1 << 32) + p["block"]
        assert np.all(np.diff(key) > 0)
        assert np.all(np.abs(p["n0"] - p["n1"]) > 1)
        assert np.all((p["phase"] == 0) == (p["n0"] > p["n1"]))
    v = r.pvars
    if len(v):
        assert np.all(np.diff(v["block"]) >= 0) and v["block"].min() == 1
        assert np.all(np.bincount(v["block"])[1:] > 3)
        assert np.all(v["lscore"] >= 10) and np.all(v["rscore"] >= 10)


def test_cfg2_shape_one_contig_vs_oracle(eng, oracle):
    """One cfg2 contig at full size: 5 Mb contig, 2000 x 15 kb reads -> invariants + oracle parity of K2..K5."""
    from falcon_unzip_amd import _lib
    contigs, blob, off, rctg = _make(1, 5_000_000, 2000, 15000, 750_000)
    job = _lib.align_job_raw(eng, contigs, blob, off, rctg)
    job.run()
    sm = job.summaries()
    assert sm["aligned"].mean() > 0.995
    assert np.all(sm["cells"][sm["aligned"] == 1] > 0)
    aln, idx = job.alnset(0)
    assert np.all(np.diff(aln.rec_pos()) >= 0) and len(np.unique(aln.rec_qid())) == aln.n_rec
    b = job.to_batch()
    b.run(_lib.STAGE_ALL)
    r = b.results()[0]
    _check_invariants(r, aln.n_qid)
    assert len(r.sites) > 1000 and len(r.preads) > 1800
    sam = _lib.format_sam(aln, "c0")
    exp = oracle.phase_all(sam, contigs[0], "c0")
    off_q, names = aln.qname_table()
    assert _lib.format_variant_pos(r.sites) == exp["variant_pos"]
    assert _lib.format_variant_map(r.sites, r.vmap_qid) == exp["variant_map"]
    assert _lib.format_atable(r.sites, r.arows) == exp["atable"]
    assert _lib.format_phased_variants(r.sites, r.pvars) == exp["phased_variants"]
    assert _lib.format_phased_reads(r.preads, "c0", off_q, names) == exp["phased_reads"]
    b.close()
    job.close()


def test_determinism_and_batch_independence(eng):
    from falcon_unzip_amd import _lib
    contigs, blob, off, rctg = _make(3, 1_000_000, 500, 15000, 200_000, cfg=8)
    job = _lib.align_job_raw(eng, contigs, blob, off, rctg)
    job.run()
    s1 = job.summaries().copy()
    b1 = job.to_batch()
    b1.run(_lib.STAGE_ALL)
    r1 = b1.results()
    b1.close()
    job.run()                                   # same resident inputs, second pass
    assert np.array_equal(job.summaries(), s1)
    b2 = job.to_batch()
    b2.run(_lib.STAGE_ALL)
    r2 = b2.results()
    b2.close()
    for a, b in zip(r1, r2):
        for f in ("sites", "vmap_qid", "arows", "pvars", "preads"):
            assert np.array_equal(getattr(a, f), getattr(b, f)), f
    job.close()
    # every contig alone == that contig inside the batch
    for c in range(3):
        m = rctg == c
        lo, hi = off[:-1][m][0], off[1:][m][-1]
        o = np.concatenate((off[:-1][m], [hi])) - lo
        j1 = _lib.align_job_raw(eng, [contigs[c]], blob[lo:hi], o, np.zeros(int(m.sum()), np.int32))
        j1.run()
        assert np.array_equal(j1.summaries(), s1[m])
        bb = j1.to_batch()
        bb.run(_lib.STAGE_ALL)
        rs = bb.results()[0]
        for f in ("sites", "vmap_qid", "arows", "pvars", "preads"):
            assert np.array_equal(getattr(rs, f), getattr(r1[c], f)), (c, f)
        _check_invariants(rs, int(m.sum()))
        bb.close()
        j1.close()


def test_chunked_traceback_budget_matches(eng, monkeypatch):
    """A tiny trace-back budget forces many read chunks; results must not change."""
    from falcon_unzip_amd import _lib
    contigs, blob, off, rctg = _make(1, 300_000, 96, 8000, 100_000, cfg=9)
    job = _lib.align_job_raw(eng, contigs, blob, off, rctg)
    job.run()
    ref = job.summaries().copy()
    a_ref, _ = job.alnset(0)
    monkeypatch.setenv("FZP_SW_CHUNKS", "7")
    job.run()
    assert np.array_equal(job.summaries(), ref)
    a2, _ = job.alnset(0)
    assert _lib.format_sam(a2, "x") == _lib.format_sam(a_ref, "x")
    job.close()


def test_cfg5_slice_streams_through_groups(tmp_path):
    """BASELINE configs[4] at 1/10 scale: ~36 k reads x 15 kb against 13.6 Mb of contigs of 1-10 Mb at 40x, through
    fzp_phase_contigs: contig groups on two lanes, K1..K6, every file written.  Size-independent properties: every read aligned and
    accounted for in rid_to_phase, >= 99 % phased, the files of every contig present and consistent with the counts, and the same
    bytes whether the groups run on one lane or two."""
    import hashlib
    import os
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))
    import run_cfg5

    def digest(root):
        h = hashlib.sha256()
        n = 0
        for d, _, files in sorted(os.walk(root)):
            for f in sorted(files):
                with open(os.path.join(d, f), "rb") as fh:
                    h.update(os.path.relpath(os.path.join(d, f), root).encode())
                    h.update(fh.read())
                n += 1
        return h.hexdigest(), n
    res = {}
    for lanes in (2, 1):
        r = run_cfg5.run(scale=0.1, lanes=lanes, workers=1, out_root=str(tmp_path), keep=True)
        res[lanes] = (r, digest(r["out_dir"]))
    r, (dg, nfiles) = res[2]
    st = r["stats"]
    assert st["n_groups"] >= 2 and st["n_reads"] == st["n_aligned"] == r["r2p_records"] > 30000
    assert r["reads_phased"] >= 0.99 * st["n_reads"] and st["n_preads"] >= r["reads_phased"]
    assert nfiles == 8 * len([d for d in os.listdir(r["out_dir"])]) and r["peak_hbm_gb"] < 120
    assert st["n_sites"] > 20000 and st["n_pvars"] <= st["n_sites"] and st["n_rows"] > 30 * st["n_sites"]
    assert res[1][1] == (dg, nfiles)                       # lanes only change the schedule
    ctg = sorted(os.listdir(r["out_dir"]))[0]
    base = os.path.join(r["out_dir"], ctg)
    with open(os.path.join(base, "phased_reads")) as f:
        rows = [l.split() for l in f]
    assert all(x[1] == ctg and abs(int(x[4]) - int(x[5])) > 1 for x in rows)
    with open(os.path.join(base, "rid_to_phase.%s" % ctg)) as f:
        r2p = [l.split() for l in f]
    assert len(r2p) == len(set(x[0] for x in r2p)) and sum(x[2] != "-1" for x in r2p) == len(set(x[6] for x in rows))
    with open(os.path.join(base, "cns", "phased_blocks.fa")) as f:
        assert f.read().count(">") >= 2


def test_cfg2_workload_equals_oracles_on_every_read_and_contig(eng, oracle, record_property):
    """The bench's own workload, read for read: as many cfg2 contigs (5 Mb, 2 000 x 15 kb reads each) as the host's cores pay for in a
    few seconds -- all 20 on a 256-thread node -- aligned by the HIP path in one job and by the threaded CPU twin contig by contig:
    every summary field of every read equal; then K2..K5 of every contig against the oracle chain."""
    import os
    from falcon_unzip_amd import _lib
    from tests import oracle_lib
    cores = os.cpu_count() or 1
    n_ctg = max(1, min(20, cores // 12))
    record_property("cfg2_contigs_checked", n_ctg)
    try:                                                    # how many contigs this box paid for, where the round's records can see it
        rec = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out")
        os.makedirs(rec, exist_ok=True)
        with open(os.path.join(rec, "test_cfg2_contigs_checked.json"), "w") as f:
            f.write('{"host_threads": %d, "contigs_checked": %d, "reads_checked": %d}\n' % (cores, n_ctg, 2000 * n_ctg))
    except OSError:
        pass
    print("test_cfg2_workload: %d host threads -> %d of 20 contigs (%d reads) checked read for read" % (cores, n_ctg, 2000 * n_ctg))
    contigs, blob, off, rctg = _make(n_ctg, 5_000_000, 2000, 15000, 750_000, cfg=2)
    job = _lib.align_job_raw(eng, contigs, blob, off, rctg)
    job.run()
    got = job.summaries()
    got_hash = job.cigar_hashes()
    assert got["aligned"].mean() > 0.995
    fields = ("aligned", "strand", "pos", "ref_end", "q_start", "q_end", "score", "n_cigar", "cells", "n_columns", "n_match")
    n_checked = 0
    for c in range(n_ctg):
        idx = np.flatnonzero(rctg == c)
        reads = [blob[off[i]:off[i + 1]] for i in idx]
        exp, exp_cig = oracle_lib.align_reads(oracle, contigs[c], reads, n_threads=cores)
        for f in fields:
            assert np.array_equal(got[f][idx], exp[f]), (c, f, np.flatnonzero(got[f][idx] != exp[f])[:5])
        # ... and every CIGAR, gap placement included: 64-bit fingerprints of the device's words against the twin's (equal summaries do not prove equal gap placement)
        eh = _lib.cigar_hash_of_words(exp_cig)
        assert np.array_equal(got_hash[idx], eh), (c, "cigar", np.flatnonzero(got_hash[idx] != eh)[:5])
        assert (eh != 0).mean() > 0.99
        n_checked += len(idx)
    assert n_checked == 2000 * n_ctg
    # ... and the phasing chain of the same job, contig for contig, against the oracle chain fed the same records as SAM text
    from concurrent.futures import ThreadPoolExecutor
    b = job.to_batch()
    b.run(_lib.STAGE_ALL)
    res = b.results()

    alns = [job.alnset(c)[0] for c in range(n_ctg)]        # one context: fetched in turn; only the oracle runs threaded

    def one(c):
        aln = alns[c]
        name = "c%d" % c
        exp = oracle.phase_all(_lib.format_sam(aln, name), contigs[c], name)
        r = res[c]
        off_q, names = aln.qname_table()
        ok = (_lib.format_variant_pos(r.sites) == exp["variant_pos"]
              and _lib.format_variant_map(r.sites, r.vmap_qid) == exp["variant_map"]
              and _lib.format_atable(r.sites, r.arows) == exp["atable"]
              and _lib.format_phased_variants(r.sites, r.pvars) == exp["phased_variants"]
              and _lib.format_phased_reads(r.preads, name, off_q, names) == exp["phased_reads"])
        return ok, len(r.sites), len(r.preads)

    with ThreadPoolExecutor(max_workers=min(8, n_ctg)) as ex:
        outs = list(ex.map(one, range(n_ctg)))
    assert all(o[0] for o in outs), [c for c, o in enumerate(outs) if not o[0]]
    assert all(o[1] > 1000 and o[2] > 1800 for o in outs)
    b.close()
    job.close()


def test_upload_modes_give_the_same_job(eng, monkeypatch):
    """Big uploads (> 4 MiB) have three routes -- the runtime's pageable copy (default), the library's pinned threaded staging, an in-place
    page lock -- chosen per call by FZP_UPLOAD_MODE: same device bytes, so the same alignments, whichever is taken."""
    from falcon_unzip_amd import _lib
    contigs, blob, off, rctg = _make(2, 3_000_000, 600, 15000, 400_000, cfg=12)      # 9 MB of reads, 6 MB of contigs
    got = {}
    for mode in ("direct", "staged", "register"):
        monkeypatch.setenv("FZP_UPLOAD_MODE", mode)
        job = _lib.align_job_raw(eng, contigs, blob, off, rctg)
        job.run()
        got[mode] = job.summaries().copy()
        job.close()
    monkeypatch.delenv("FZP_UPLOAD_MODE")
    assert got["direct"]["aligned"].mean() > 0.99
    assert np.array_equal(got["direct"], got["staged"]) and np.array_equal(got["direct"], got["register"])


def test_fail_list_overflow_at_20000_reads(eng, monkeypatch):
    """The whole-masks retry at the size it exists for: 20 000 reads (ten cfg2 contigs) whose walks are told to accept +-2 lanes only (FZP_TB_WINDOW) -- far more than 8 192
    pieces leave the recorded lanes, the fail list overflows and the run is done again with whole masks for every piece (k_sw throughout).  Same summaries, same CIGARs
    (64-bit fingerprints of every read's words) as the default run, which test_cfg2_workload_equals_oracles_on_every_read_and_contig holds against the twin."""
    from falcon_unzip_amd import _lib
    contigs, blob, off, rctg = _make(10, 5_000_000, 2000, 15000, 750_000, cfg=2)
    job = _lib.align_job_raw(eng, contigs, blob, off, rctg)
    job.run()
    s0, h0 = job.summaries().copy(), job.cigar_hashes().copy()
    assert s0["aligned"].mean() > 0.995 and (h0 != 0).mean() > 0.995
    monkeypatch.setenv("FZP_SWB_64", "1")
    monkeypatch.setenv("FZP_TB_WINDOW", "2")
    monkeypatch.setenv("FZP_TB_NO_RETRY", "1")
    with pytest.raises(_lib.FzpError) as e:                     # it IS the overflow path: without the retry the run is refused
        job.run()
    assert "whole trace-back masks" in str(e.value)
    monkeypatch.delenv("FZP_TB_NO_RETRY")
    job.run()
    s1, h1 = job.summaries(), job.cigar_hashes()
    assert np.array_equal(s0, s1) and np.array_equal(h0, h1)
    # the same question asked late: fzp_job_phase_write's run leaves "did the fail list overflow?" to the fetch of fzp_align_to_batch, which then does the retry itself
    ids = ["%06dF" % c for c in range(10)]
    st1, _ = job.phase_write(ids)
    h2 = job.cigar_hashes().copy()
    monkeypatch.delenv("FZP_TB_WINDOW"); monkeypatch.delenv("FZP_SWB_64")
    st0, _ = job.phase_write(ids)
    keys = ("n_aligned", "n_rec", "n_sites", "n_rows", "n_arows", "n_pvars", "n_preads")
    assert np.array_equal(h0, h2) and {k: st1[k] for k in keys} == {k: st0[k] for k in keys} and st0["n_sites"] > 1000
    job.close()


def test_whole_mask_mode_in_chunks_stays_inside_its_budget(monkeypatch):
    """with whole masks for every piece (non-default scores, FZP_SW_NO_BITS, the retry after a fail-list overflow) a run cut into chunks sized each chunk's
    16-byte mask room by the WHOLE run's capacity -- 2 x (run steps x 16 B) where the budget said 2 x (chunk steps x 16 B).  3 000 reads of 15 kb (~100 M DP steps, 1.6 GB of
    whole masks) under a 1 GiB budget: several chunks, the same alignments as the one-chunk run, and the device memory the job took stays far below the whole run's masks twice."""
    from falcon_unzip_amd import _lib
    contigs, blob, off, rctg = _make(2, 3_000_000, 1500, 15000, 600_000, cfg=13)
    e1 = _lib.Engine(0)
    job = _lib.align_job_raw(e1, contigs, blob, off, rctg)
    job.run()
    s0, h0 = job.summaries().copy(), job.cigar_hashes().copy()
    steps = int(s0["cells"].sum()) // _lib.align_band()
    job.close(); e1.close()
    monkeypatch.setenv("FZP_SW_NO_BITS", "1")
    monkeypatch.setenv("FZP_TB_BUDGET_GB", "1")
    e2 = _lib.Engine(0)
    free0, _ = e2.mem_info()
    job = _lib.align_job_raw(e2, contigs, blob, off, rctg)
    job.run()
    s1, h1 = job.summaries(), job.cigar_hashes()
    free1, _ = e2.mem_info()
    assert np.array_equal(s0, s1) and np.array_equal(h0, h1)
    took = free0 - free1                                         # blocks the context keeps cached count as taken: this is the job's peak
    assert steps * 16 > 1.2e9                                    # the whole run's masks, once
    assert took < 1.5 * steps * 16, (took, steps * 16)           # measured 1.73 GB: two chunk buffers of 0.6 GB + the job's own 0.5 GB; before the fix 2 x 1.48 GB of masks + 0.5 GB of 8-byte room + the rest
    job.close(); e2.close()
