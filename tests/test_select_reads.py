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
