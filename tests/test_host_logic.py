"""CPU-side checks: host logic of the product (SAM parser, serializers, text parsers, readmap) and the
C-ABI surface.  No GPU compute is called here."""
import ctypes
import os
import re

import numpy as np
import pytest

from tests.golden_util import Case, cases

CASES = cases()
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def lib():
    import __graft_entry__ as g
    g.build()
    from falcon_unzip_amd import _lib
    return _lib


def test_library_exports_every_declared_symbol(lib):
    hdr = open(os.path.join(REPO, "include", "fzphase.h")).read()
    names = sorted(set(re.findall(r"\b(fzp_[a-z0-9_]+)\s*\(", hdr)))
    assert len(names) > 25
    so = ctypes.CDLL(lib.LIB_PATH)
    missing = [n for n in names if not hasattr(so, n)]
    assert not missing, missing


def test_no_device_fails_loudly(lib):
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    with pytest.raises(lib.FzpError) as ei:
        lib.Engine(0)
    assert ei.value.code == lib.FZP_ENODEVICE


@pytest.mark.parametrize("name", CASES)
def test_parse_sam_and_q_id_map(lib, name):
    c = Case(name)
    aln = lib.parse_sam(c.sam)
    c.check("q_id_map", lib.format_q_id_map(aln))
    pos = aln.rec_pos()
    assert np.all(np.diff(pos) >= 0)
    assert aln.last_pos == (int(pos[-1]) if len(pos) else -1)


@pytest.mark.parametrize("name", CASES)
def test_text_roundtrip(lib, name):
    from falcon_unzip_amd import textio
    c = Case(name)
    vm = c.expected("variant_map")
    if vm is None:
        pytest.skip("hash-only")
    sites, q = textio.parse_variant_map(vm)
    assert lib.format_variant_map(sites, q) == vm
    at = c.expected("atable")
    if at is not None:
        assert lib.format_atable(sites, textio.parse_atable(at, sites)) == at
    pv = c.expected("phased_variants")
    assert lib.format_phased_variants(sites, textio.parse_phased_variants(pv, sites)) == pv


@pytest.mark.parametrize("name", [n for n in CASES if Case(n).has("rid_to_phase")])
def test_readmap(lib, name):
    c = Case(name)
    rm = c.readmap_inputs()
    recs, text = lib.readmap(c.expected("phased_reads"), rm["rawread_ids"], rm["pread_ids"], rm["pread_to_contigs"], c.ctg_id, 7)
    c.check("rid_to_phase", text)
    assert len(recs) == text.count(b"\n")
    assert np.all(recs["ctg"] == 7)
    assert np.all(np.diff(recs["arid"]) > 0)


def _readmap_records(lib, rawread_ids, pread_ids, p2c, ctg_id, ctg_index, preads, name_off, names):
    C = ctypes
    so = lib.load()
    f = so.fzp_debug_readmap_records
    f.restype = C.c_int
    f.argtypes = [C.c_char_p, C.c_size_t, C.c_char_p, C.c_size_t, C.c_char_p, C.c_size_t, C.c_char_p, C.c_int32, C.c_void_p, C.c_int64, C.c_void_p, C.c_char_p, C.c_int32,
                  C.POINTER(C.c_void_p), C.POINTER(C.c_int64), C.POINTER(C.c_void_p), C.POINTER(C.c_size_t)]
    preads = np.ascontiguousarray(preads, dtype=lib.PREAD)
    name_off = np.ascontiguousarray(name_off, dtype=np.int64)
    rp, nr, tp, tn = C.c_void_p(), C.c_int64(), C.c_void_p(), C.c_size_t()
    rc = f(rawread_ids, len(rawread_ids), pread_ids, len(pread_ids), p2c, len(p2c), ctg_id.encode(), ctg_index, preads.ctypes.data, len(preads), name_off.ctypes.data, names,
           len(name_off) - 1, C.byref(rp), C.byref(nr), C.byref(tp), C.byref(tn))
    if rc != 0:
        raise lib.FzpError(rc, so.fzp_last_error().decode())
    recs = np.frombuffer(C.string_at(rp, nr.value * lib.R2P.itemsize), lib.R2P).copy()
    text = C.string_at(tp, tn.value)
    so.fzp_free(rp)
    so.fzp_free(tp)
    return recs, text


def test_read_map_from_records_equals_read_map_from_text(lib):
    """The read map fzp_job_phase_write makes per contig -- from phased-read RECORDS and the q_id name table, resolved in two halves (the rows of pread_to_contigs down to read
    names while the device is still phasing; the phases filled in afterwards) -- against fzp_readmap, the restatement of phasing_readmap.py:8-51 that works from the files' text
    and is pinned by the goldens: preads listed several times (the last row wins), rows of other ranks and of other contigs (also contigs whose id starts with this one's), reads
    that share a name, reads that are aligned but not phased, raw reads that are not among the aligned ones."""
    rng = np.random.Generator(np.random.PCG64(99))
    for trial in range(6):
        n_raw, n_q = 900, 400
        rawread_ids = b"".join(b"m/%d/0_%d\n" % (i % 700, 1000 + i % 700) for i in range(n_raw))      # names repeat from 700 on
        n_pread = 600
        pread_ids = b"".join(b"x/%d/0_1\n" % (10 * int(rng.integers(0, n_raw)) + int(rng.integers(0, 10))) for _ in range(n_pread))
        ctg = "000012F"
        rows = []
        for _ in range(1500):
            pid = int(rng.integers(0, n_pread))
            name = [ctg, ctg, ctg + "-001", "000013F", "0000"][int(rng.integers(0, 5))]
            rows.append(b"%09d %s 5 %d" % (pid, name.encode(), int(rng.integers(0, 2)) * int(rng.integers(0, 3))))
        p2c = b"\n".join(rows) + b"\n"
        # the aligned reads of the contig: names from the raw reads (some shared), q ids in some order; phased reads: a subset, some in two blocks
        qn = [b"m/%d/0_%d" % (i, 1000 + i) for i in rng.integers(0, 760, n_q)]
        name_off = np.zeros(n_q + 1, np.int64)
        name_off[1:] = np.cumsum([len(x) for x in qn])
        names = b"".join(qn)
        pr = []
        for q in range(n_q):
            for blk in sorted(set(int(x) for x in rng.integers(0, 6, int(rng.integers(0, 3))))):
                pr.append((q, blk, int(rng.integers(0, 2)), int(rng.integers(0, 9)), int(rng.integers(0, 9))))
        preads = np.array(pr, lib.PREAD)
        text_pr = lib.format_phased_reads(preads, ctg, name_off, names)
        exp_recs, exp_text = lib.readmap(text_pr, rawread_ids, pread_ids, p2c, ctg, 3)
        recs, text = _readmap_records(lib, rawread_ids, pread_ids, p2c, ctg, 3, preads, name_off, names)
        assert text == exp_text and len(text) > 3000, trial
        assert recs.tobytes() == exp_recs.tobytes()
        assert (recs["block"] == -1).any() and (recs["block"] >= 0).any()
    # the same messages for the same bad inputs
    for bad_p2c, bad_pid in ((b"000000005 000012F 5 x\n", pread_ids), (b"000000005 000012F\n", pread_ids), (b"000999999 000012F 5 0\n", pread_ids), (b"000000000 000012F 5 0\n", b"noslash\n")):
        with pytest.raises(lib.FzpError) as e1:
            lib.readmap(text_pr, rawread_ids, bad_pid, bad_p2c, ctg, 3)
        with pytest.raises(lib.FzpError) as e2:
            _readmap_records(lib, rawread_ids, bad_pid, bad_p2c, ctg, 3, preads, name_off, names)
        assert str(e1.value) == str(e2.value)


def test_parser_errors(lib):
    with pytest.raises(lib.FzpError) as ei:
        lib.parse_sam(b"r1\t0\tc\t1\t254\t*\t*\t0\t0\tACGT\t*\n")
    assert ei.value.code == lib.FZP_EZERODIV
    with pytest.raises(lib.FzpError) as ei:   # CIGAR longer than SEQ: IndexError in the reference
        lib.parse_sam(b"r1\t0\tc\t1\t254\t3000M\t*\t0\t0\tACGT\t*\n")
    assert ei.value.code == lib.FZP_EINVAL
    a = b"r1\t0\tc\t500\t254\t2500M\t*\t0\t0\t" + b"A" * 2500 + b"\t*\n"
    b = b"r2\t0\tc\t100\t254\t2500M\t*\t0\t0\t" + b"A" * 2500 + b"\t*\n"
    with pytest.raises(lib.FzpError) as ei:
        lib.parse_sam(a + b)
    assert ei.value.code == lib.FZP_EUNSORTED


def test_clip_filter_is_ieee_double(lib):
    """phasing.py:72: total=10000, skip=9000 -> 1.0 - 0.9 = 0.0999.. < 0.1 -> dropped (an integer test would keep it)."""
    rec = b"r1\t0\tc\t1\t254\t9000S1000M\t*\t0\t0\t" + b"A" * 10000 + b"\t*\n"
    aln = lib.parse_sam(rec)
    assert aln.n_rec == 0 and aln.n_qid == 1
    rec = b"r1\t0\tc\t1\t254\t8999S1001M\t*\t0\t0\t" + b"A" * 10000 + b"\t*\n"
    assert lib.parse_sam(rec).n_rec == 1


def test_contig_buffers_are_lent_not_copied():
    """The binding hands the library the contigs' own bytes (no copy): pointers and lengths for bytes, bytearray, numpy and empty contigs"""
    import ctypes as C
    import numpy as np
    from falcon_unzip_amd import _lib
    contigs = [b"ACGTACGT", bytearray(b"TTGCA"), np.frombuffer(b"GGGCCC", dtype=np.uint8), b""]
    keep, ptr, clen = _lib._contig_ptrs(contigs)
    assert list(clen) == [8, 5, 6, 0]
    for k, c in enumerate(contigs[:3]):
        assert C.string_at(ptr[k], clen[k]) == bytes(c)
    assert ptr[3]                                   # an empty contig still gets a valid address
    assert ptr[0] == C.cast(C.c_char_p(contigs[0]), C.c_void_p).value       # the bytes object's own buffer
    assert ptr[2] == contigs[2].ctypes.data


def test_unzip_config_hook_and_gpu_phasing_task(tmp_path):
    """falcon_unzip_amd/unzip_tasks.py: the [Unzip] keys and the one pypeflow task that stands in for unzip_all's per-contig loops (unzip.py:231-288).
    The task body only writes its script (as task_run_blasr / task_phasing do, unzip.py:61-133); what the script runs is covered on the GPU
    (tests/test_gpu_pipeline.py)."""
    import configparser
    from falcon_unzip_amd import _miniflow, unzip_tasks
    cfg = configparser.ConfigParser()
    cfg.read_string("[General]\njob_type = local\n[Unzip]\nsmrt_bin = /opt/smrt/bin\n")
    assert unzip_tasks.read_config(cfg)["phasing_backend"] == "blasr"                 # absent keys: the reference's own path
    cfg.read_string("[Unzip]\nphasing_backend = HIP\nphasing_gpus = 8\n")
    conf = unzip_tasks.read_config(cfg)
    assert conf["phasing_backend"] == "hip" and conf["phasing_gpus"] == 8
    cfg.read_string("[Unzip]\nphasing_backend = cuda\n")
    with pytest.raises(ValueError):
        unzip_tasks.read_config(cfg)
    unzip = tmp_path / "3-unzip"
    (unzip / "reads").mkdir(parents=True)
    for c in ("000000F", "000001F"):
        (unzip / "reads" / ("%s_ref.fa" % c)).write_text(">%s\nACGT\n" % c)
        (unzip / "reads" / ("%s_reads.fa" % c)).write_text(">r/1/0_4\nACGT\n")
    wf = _miniflow.PypeProcWatcherWorkflow()
    out = unzip_tasks.add_gpu_phasing_task(wf, conf, ["000000F", "000001F"], _miniflow.PypeTask, _miniflow.makePypeLocalFile, unzip_dir=str(unzip),
                                           read_map_dir=str(tmp_path / "read_maps"))
    assert str(out).endswith("3-unzip/1-hasm/rid-to-phase-all/rid_to_phase.all")     # what task_hasm waits for (unzip.py:285,293)
    task = wf._tasks[0]
    assert len(task.inputs) == 4 and set(task.outputs) == {"rid_to_phase_all", "job_done"}
    wf.refreshTargets()                                                                # runs the body: the script exists, nothing is executed
    script = open(task.generated_script_fn).read()
    done = str(task.outputs["job_done"])
    assert script.startswith("set -vex\ntrap 'touch %s.exit' EXIT\n" % done) and script.rstrip().endswith("touch %s" % done)
    assert "--nproc-per-node 8" in script and "fc_unzip_phase_gpu.py --unzip_dir %s --read_map_dir %s" % (unzip, tmp_path / "read_maps") in script


def _load_group(lib, reads_dir, ids, threads=0, eng=None):
    """fzp_debug_load_fasta_group -> (contigs, [(ctg index, name, seq)]); eng: through the device reader instead (fzp_debug_load_fasta_group_dev: tests/test_gpu_pipe.py)"""
    C = ctypes
    so = lib.load()
    f = so.fzp_debug_load_fasta_group_dev if eng is not None else so.fzp_debug_load_fasta_group
    f.restype = C.c_int
    P = C.POINTER
    f.argtypes = ([C.c_void_p] if eng is not None else []) + [C.c_char_p, P(C.c_char_p), C.c_int32, C.c_int32] + [P(C.c_void_p)] * 7 + [P(C.c_int64)]
    arr = (C.c_char_p * len(ids))(*[i.encode() for i in ids])
    out = [C.c_void_p() for _ in range(7)]
    n = C.c_int64()
    rc = f(*([eng._p] if eng is not None else []), reads_dir.encode(), arr, len(ids), threads, *[C.byref(o) for o in out], C.byref(n))
    if rc != 0:
        raise lib.FzpError(rc, so.fzp_last_error().decode())
    nr, nc = n.value, len(ids)
    ref_off = np.frombuffer(C.string_at(out[1], 8 * (nc + 1)), np.int64)
    ref = C.string_at(out[0], int(ref_off[-1]))
    off = np.frombuffer(C.string_at(out[3], 8 * (nr + 1)), np.int64)
    blob = C.string_at(out[2], int(off[-1]))
    noff = np.frombuffer(C.string_at(out[5], 8 * (nr + 1)), np.int64)
    names = C.string_at(out[4], int(noff[-1]))
    rctg = np.frombuffer(C.string_at(out[6], 4 * nr), np.int32) if nr else np.zeros(0, np.int32)
    for o in out:
        so.fzp_free(o)
    return [ref[ref_off[c]:ref_off[c + 1]] for c in range(nc)], [(int(rctg[r]), names[noff[r]:noff[r + 1]], blob[off[r]:off[r + 1]]) for r in range(nr)]


def hostile_fasta_dir(d):
    """<d>/<ctg>_{reads,ref}.fa of four contigs with everything a FASTA reader must not stumble over (see the test below) -> (ids, the first contig's true sequence)"""
    rng = np.random.Generator(np.random.PCG64(4242))
    acgt = np.frombuffer(b"ACGTacgtN", np.uint8)
    d.mkdir()
    ids = ["000000F", "000001F", "000002F", "000003F"]

    def seq(n):
        return acgt[rng.integers(0, len(acgt), n)].tobytes()

    def write_reads(path, n_rec, mean, style):
        with open(path, "wb") as f:
            if style == 1:
                f.write(b"stray line before any header\nACGT\n")
            for i in range(n_rec):
                s = seq(int(rng.integers(0, 2 * mean))) if i % 17 else b""
                nl = b"\r\n" if style == 2 and i % 2 else b"\n"
                f.write(b">r%d/%d/0_%d   some description > here" % (style, i, len(s)) + nl)
                w = (0, 60, 70, 1000)[(i + style) % 4]
                if w == 0:
                    f.write(s + nl)
                else:
                    for x in range(0, len(s), w):
                        f.write(b"  " * (i % 3 == 0) + s[x:x + w] + b" \t" * (i % 5 == 0) + nl)
                if i % 11 == 0:
                    f.write(nl)
            if style == 3:
                f.write(b">last_without_newline\nACGTACGT")
    write_reads(d / "000000F_reads.fa", 700, 9000, 0)          # ~6 MB: two pieces
    write_reads(d / "000001F_reads.fa", 1500, 4000, 2)         # CRLF
    write_reads(d / "000002F_reads.fa", 40, 300, 1)
    (d / "000003F_reads.fa").write_bytes(b"")                  # a contig without reads
    ref_a, ref_b = seq(30000), seq(100)
    (d / "000000F_ref.fa").write_bytes(b">other\n" + seq(500) + b"\n>000000F first\n" + ref_b + b"\n>000000F second wins\n" + b"\n".join(ref_a[x:x + 80] for x in range(0, len(ref_a), 80)) + b"\n")
    (d / "000001F_ref.fa").write_bytes(b">000001F\r\n" + ref_b + b"\r\n")
    (d / "000002F_ref.fa").write_bytes(b">000002F_not_it\n" + ref_b + b"\n")      # no record of that name: an empty contig
    (d / "000003F_ref.fa").write_bytes(b">000003F")                                # a header and nothing else
    write_reads(d / "000003F_reads.fa", 5, 200, 3)
    return ids, ref_a


def test_fasta_reader_equals_the_python_reader_on_hostile_files(lib, tmp_path, monkeypatch):
    """The library's FASTA reader (fzp_phase_contigs_files: files read in 4 MB pieces, records scanned and copied by several threads) against pipeline.read_fasta, the
    reader the in-memory path uses (falcon_kit's FastaReader as phasing.py:489-494 uses it): CRLF, blank lines, wrapped and unwrapped sequences, headers with descriptions,
    empty sequences, text before the first header, no newline at the end, an empty file, a reference file with several records (the LAST one named <ctg> is the contig) or
    none, and files several pieces long whose records straddle the piece boundaries."""
    from falcon_unzip_amd import pipeline
    d = tmp_path / "reads"
    ids, ref_a = hostile_fasta_dir(d)
    for threads, piece in ((1, None), (3, None), (0, None), (4, "4099"), (2, "61"), (5, "17")):
        if piece:
            monkeypatch.setenv("FZP_FASTA_PIECE", piece)       # pieces smaller than a line, than a header
        ctgs, reads = _load_group(lib, str(d), ids, threads)
        exp_reads = []
        for c, cid in enumerate(ids):
            ref = b""
            for nm, sq in pipeline.read_fasta(str(d / ("%s_ref.fa" % cid))):
                if nm.decode() == cid:
                    ref = sq
            assert ctgs[c] == ref, (threads, cid)
            exp_reads += [(c, nm, sq) for nm, sq in pipeline.read_fasta(str(d / ("%s_reads.fa" % cid)))]
        assert len(reads) == len(exp_reads) > 2200 and reads == exp_reads, threads
    assert ctgs[0] == ref_a and ctgs[2] == b"" and ctgs[3] == b""
    monkeypatch.delenv("FZP_FASTA_PIECE")
    with pytest.raises(lib.FzpError) as e:
        _load_group(lib, str(d), ids + ["999999F"])
    assert "999999F" in str(e.value)


def test_fasta_reader_uses_one_line_records_where_they_lie(lib, tmp_path, monkeypatch, capfd):
    """Files as falcon_kit's fetch_reads writes them (unzip.py:49-50: a header line, the sequence on one line): the reader hands the reads over as spans of the files' bytes
    (fzp_align_create_spans) instead of joining them; same records either way, also when the pieces are smaller than the lines, and one wrapped read anywhere sends the group
    through the joining path."""
    from falcon_unzip_amd import pipeline
    rng = np.random.Generator(np.random.PCG64(77))
    acgt = np.frombuffer(b"ACGT", np.uint8)
    d = tmp_path / "reads"
    d.mkdir()
    ids = ["000000F", "000001F", "000002F"]
    exp_reads, exp_ctgs = [], []
    for c, cid in enumerate(ids):
        ref = acgt[rng.integers(0, 4, 20000 + 1000 * c)].tobytes()
        exp_ctgs.append(ref)
        (d / ("%s_ref.fa" % cid)).write_bytes(b">%s\n%s\n" % (cid.encode(), ref))
        with open(d / ("%s_reads.fa" % cid), "wb") as f:
            for i in range(300):
                sq = acgt[rng.integers(0, 4, int(rng.integers(0 if i % 50 == 0 else 1, 3000)))].tobytes()
                nm = b"m/%d/%d_%d" % (c, i, len(sq))
                f.write(b">" + nm + b" RQ=0.8\n" + sq + (b"" if c == 2 and i == 299 else b"\n"))      # the last file ends without a newline
                exp_reads.append((c, nm, sq))
    monkeypatch.setenv("FZP_PIPE_TIMING", "1")
    for threads, piece, join in ((0, None, False), (3, "1000", False), (2, "7", False), (2, "1", False), (4, None, True), (3, "129", True)):
        monkeypatch.setenv("FZP_FASTA_PIECE", piece) if piece else monkeypatch.delenv("FZP_FASTA_PIECE", raising=False)
        monkeypatch.setenv("FZP_FASTA_JOIN", "1") if join else monkeypatch.delenv("FZP_FASTA_JOIN", raising=False)
        capfd.readouterr()
        ctgs, reads = _load_group(lib, str(d), ids, threads)
        err = capfd.readouterr().err
        assert ("reads used in place" in err) == (not join) and ("reads joined" in err) == join, err
        assert ctgs == exp_ctgs and reads == exp_reads, (threads, piece, join)
    monkeypatch.delenv("FZP_FASTA_JOIN", raising=False)
    monkeypatch.delenv("FZP_FASTA_PIECE", raising=False)
    with open(d / "000001F_reads.fa", "ab") as f:             # one wrapped read: the whole group is joined
        f.write(b">wrapped\nACGT\nACGT\n")
    capfd.readouterr()
    ctgs, reads = _load_group(lib, str(d), ids, 2)
    assert "reads joined" in capfd.readouterr().err
    at = 600
    assert reads[:at] == exp_reads[:at] and reads[at] == (1, b"wrapped", b"ACGTACGT") and reads[at + 1:] == exp_reads[at:]
    assert reads == [(c, nm, sq) for c, cid in enumerate(ids) for nm, sq in pipeline.read_fasta(str(d / ("%s_reads.fa" % cid)))]


def test_rid_to_phase_all_formatter(lib):
    """fzp_format_rid_to_phase_all == the rows phasing_readmap.py:47-51 prints ('%09d ctg block phase'), in the order given; a record of an unknown contig is an error"""
    rng = np.random.Generator(np.random.PCG64(5))
    n = 5000
    recs = np.zeros(n, lib.R2P)
    recs["arid"] = rng.integers(0, 10**9, n)
    recs["ctg"] = np.sort(rng.integers(0, 3, n))
    recs["block"] = rng.integers(-1, 40, n)
    recs["phase"] = rng.integers(0, 2, n)
    ids = ["000000F", "000001F_long_name", "x"]
    exp = "".join("%09d %s %d %d\n" % (r["arid"], ids[r["ctg"]], r["block"], r["phase"]) for r in recs).encode()
    assert lib.format_rid_to_phase_all(recs, ids) == exp
    assert lib.format_rid_to_phase_all(recs[:0], ids) == b""
    recs["ctg"][7] = 3
    with pytest.raises(lib.FzpError):
        lib.format_rid_to_phase_all(recs, ids)


def test_bench_scratch_choice_and_step_timeline(tmp_path):
    """bench.py takes a memory file system for its trees only when it has room (a container's default /dev/shm has 64 MB); tools/step_timeline.py finds a step's idle
    intervals in a kernel trace (here: a made-up one -- two steps of three launches with one 40 us hole)."""
    import subprocess
    import sys
    import bench
    assert bench.shm_with_room(1 << 62) is None
    assert bench.shm_with_room(1) in ("/dev/shm", None)
    rows = ['"Kind","Agent_Id","Queue_Id","Stream_Id","Thread_Id","Dispatch_Id","Kernel_Id","Kernel_Name","Correlation_Id","Start_Timestamp","End_Timestamp"']
    t = 1_000_000
    for step in range(3):
        base = t + step * 5_000_000
        for name, st, en in (("(anonymous namespace)::k_index_stage_anch(int)", 0, 100_000), ("void (anonymous namespace)::k_swb<true>(long)", 100_000, 2_000_000),
                             ("(anonymous namespace)::k_tb_walk_h(int)", 2_040_000, 3_000_000)):
            rows.append('"KERNEL_DISPATCH","Agent 2",1,1,1,1,1,"%s",1,%d,%d' % (name, base + st, base + en))
    p = tmp_path / "kt.csv"
    p.write_text("\n".join(rows) + "\n")
    out = subprocess.run([sys.executable, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools", "step_timeline.py"), str(p), "15", str(tmp_path / "l.txt")],
                         capture_output=True, text=True, check=True).stdout
    assert "3 launches" in out and "40.0  k_swb<true> -> k_tb_walk_h" in out
    assert "k_tb_walk_h" in (tmp_path / "l.txt").read_text()


def test_design_is_the_short_current_state_document_with_a_generated_kernel_table():
    """DESIGN.md <= 30 KB, its kernel table = what tools/design_kernel_table.py makes of the last profiles/r6?_* set; the log lives in HISTORY.md."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    assert os.path.getsize(os.path.join(root, "DESIGN.md")) <= 30 * 1024
    assert os.path.exists(os.path.join(root, "HISTORY.md"))
    r = subprocess.run([sys.executable, os.path.join(root, "tools", "design_kernel_table.py"), "--check"], capture_output=True, text=True)
    assert r.returncode == 0, r.stdout + r.stderr


def test_integer_formatting_of_the_serialisers(lib):
    """the serialisers write their numbers two digits per division (r6): every digit count, negative values, 2^31 - 1 and beyond (no GPU needed: host code of the library)"""
    from falcon_unzip_amd import _lib
    rng = np.random.default_rng(5)
    vals = [0, 9, 10, 99, 100, 999, 1000, 65535, 99999, 100000, 2147483646] + [int(x) for x in rng.integers(0, 2**31 - 2, 200)] + [int(10 ** k) for k in range(1, 10)] + [int(10 ** k - 1) for k in range(1, 10)]
    S = np.zeros(len(vals), _lib.SITE)
    for i, v in enumerate(vals):
        S[i]["pos"] = v
        S[i]["total"] = vals[-1 - i]
        S[i]["count"] = [v % 1000, -(v % 77), vals[(i * 7) % len(vals)], 0]
        S[i]["ref_base"] = 65
        S[i]["base"] = [65, 67, 71, 84]
    exp = "".join("%d A %d A %d C %d G %d T %d\n" % (v + 1, vals[-1 - i], v % 1000, -(v % 77), vals[(i * 7) % len(vals)], 0) for i, v in enumerate(vals))
    assert _lib.format_variant_pos(S).decode() == exp
