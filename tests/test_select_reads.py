"""falcon_unzip_amd/select_reads_from_bam.py against fixtures made by RUNNING the reference with a pysam stand-in
(tests/golden_select/make_golden_select.py): which reads land in which <ctg>.bam, in what order, under which header.  The BAM
reading / writing is the library's (fzp_bam_open / fzp_bam_write, host code: no device context is involved, so this runs in the
CPU suite)."""
import hashlib
import json
import os
import shutil
import subprocess
import sys

import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(HERE)
GOLD = os.path.join(HERE, "golden_select")


@pytest.mark.parametrize("case", ["s1", "s2"])
def test_matches_reference(case, tmp_path):
    from falcon_unzip_amd import _lib, select_reads_from_bam
    work = tmp_path / case
    shutil.copytree(os.path.join(GOLD, case), str(work))
    out = work / "out"
    out.mkdir()
    if case == "s1":
        subprocess.check_call([sys.executable, os.path.join(REPO, "scripts", "fc_select_reads_from_bam.py"), "--rawread-to-contigs", str(work / "rawread_to_contigs"),
                               "--rawread-ids", str(work / "rawread_ids"), "--sam-dir", str(out), str(work / "input_bam.fofn")], stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    else:
        select_reads_from_bam.select_reads_from_bam(str(work / "input_bam.fofn"), str(work / "rawread_to_contigs"), str(work / "rawread_ids"), str(out))
    with open(os.path.join(GOLD, case, "expected.json")) as f:
        exp = json.load(f)
    assert sorted(os.listdir(str(out))) == sorted(exp)
    for fn, e in exp.items():
        v = _lib.BamView((out / fn).read_bytes())
        assert v.header.decode() == e["header"], fn
        assert [n.decode() for n in v.names] == e["names"], fn
        assert hashlib.sha256(v.records).hexdigest() == e["records_sha256"], fn
        # and the file is real BAM for an independent reader (gzip members + magic)
        import gzip
        raw = gzip.decompress((out / fn).read_bytes())
        assert raw[:4] == b"BAM\x01" and (out / fn).read_bytes()[-28:] == bytes.fromhex("1f8b08040000000000ff0600424302001b0003000000000000000000")


def test_bam_view_round_trip_and_errors():
    from falcon_unzip_amd import _lib
    bam = open(os.path.join(GOLD, "s1", "movie0.subreads.bam"), "rb").read()
    v = _lib.BamView(bam)
    again = _lib.BamView(_lib.bam_write(v.header, v.n_ref, v.ref_block, [v.records]))
    assert again.header == v.header and again.names == v.names and again.records == v.records
    with pytest.raises(_lib.FzpError):
        _lib.BamView(bam[:200])
    with pytest.raises(_lib.FzpError):
        _lib.bam_write(v.header, 0, b"", [v.records[:-3]])


def _raw_record(name: bytes, seq_len: int, salt: int) -> bytes:
    import struct
    nm = name + b"\0"
    import random
    packed = bytes(b | 0x11 for b in random.Random(salt).randbytes((seq_len + 1) // 2))
    body = struct.pack("<iiBBHHHiiii", -1, -1, len(nm), 255, 4680, 0, 4, seq_len, -1, -1, 0) + nm + packed + b"\xff" * seq_len + b"RGZx\0"
    return struct.pack("<i", len(body)) + body


def test_streaming_route_many_blocks_and_long_records(tmp_path):
    """fzp_bam_route on inputs that span many BGZF blocks, with records longer than a block: what arrives per destination equals a
    whole-file selection through BamView, in input order; destinations without records get no file; errors name the file"""
    import random
    from falcon_unzip_amd import _lib
    rnd = random.Random(7)
    paths, allrecs = [], []
    for b in range(2):
        recs = [(b"mv%d/%d/0_9" % (b, i), _raw_record(b"mv%d/%d/0_9" % (b, i), 150_000 if i % 701 == 3 else rnd.randrange(40, 900), i)) for i in range(2500)]
        data = _lib.bam_write(b"@HD\tVN:1.5\n@RG\tID:r%d\n" % b, 0, b"", [b"".join(r[1] for r in recs)])
        assert len(data) > 5 * 65536
        p = tmp_path / ("in%d.bam" % b)
        p.write_bytes(data)
        paths.append(str(p))
        allrecs += recs
    names = [r[0] for r in allrecs if zlib_crc(r[0] + b'x') % 3 != 0]
    dest = [zlib_crc(n) % 4 for n in names]                 # destination 3 stays empty on purpose below
    dest = [d if d != 3 else 1 for d in dest]
    outs = [str(tmp_path / ("out%d.bam" % k)) for k in range(4)]
    hdr, n_ref, refs = _lib.bam_read_header(paths[0])
    assert hdr == b"@HD\tVN:1.5\n@RG\tID:r0\n" and n_ref == 0 and refs == b""
    counts, order = _lib.bam_route(paths, names, dest, outs, hdr, n_ref, refs)
    want = {k: [] for k in range(4)}
    where = dict(zip(names, dest))
    for nm, raw in allrecs:
        if nm in where:
            want[where[nm]].append((nm, raw))
    assert sorted(order) == [0, 1, 2] and not os.path.exists(outs[3]) and counts[3] == 0
    for k in range(3):
        v = _lib.BamView(open(outs[k], "rb").read())
        assert v.header == hdr and v.names == [w[0] for w in want[k]] and v.records == b"".join(w[1] for w in want[k]) and counts[k] == len(want[k])
    with pytest.raises(_lib.FzpError) as e:
        _lib.bam_route([str(tmp_path / "absent.bam")], names, dest, outs, hdr, n_ref, refs)
    assert e.value.code == _lib.FZP_EIO and "absent.bam" in str(e.value)
    bad = tmp_path / "cut.bam"
    bad.write_bytes(open(paths[0], "rb").read()[:100_000])
    with pytest.raises(_lib.FzpError) as e:
        _lib.bam_route([str(bad)], names, dest, outs, hdr, n_ref, refs)
    assert e.value.code == _lib.FZP_EINVAL and "cut.bam" in str(e.value)


def zlib_crc(b):
    import zlib
    return zlib.crc32(b)
