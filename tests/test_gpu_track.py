"""Raw-read tracker (fzp_track_reads): HIP path against the reference-generated goldens and the oracle (canonical order)."""
import os
import stat
import subprocess
import sys

import pytest

from tests import golden_ovlp_util as G
from tests import oracle_lib

pytestmark = pytest.mark.gpu
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def eng():
    from falcon_unzip_amd import _lib
    e = _lib.Engine(0)
    yield e
    e.close()


@pytest.mark.parametrize("name", G.track_cases())
def test_golden(eng, name):
    from falcon_unzip_amd import _lib
    c = G.load_track(name)
    out = _lib.track_reads(eng, c["files"], c["phased_reads"], c["read_to_contig_map"], c["rawread_ids"], c["params"]["min_len"], c["params"]["bestn"])
    assert out == c["expected"]


def test_vs_oracle_larger(eng, oracle):
    from falcon_unzip_amd import _lib
    sys.path.insert(0, os.path.join(REPO, "tests", "golden_ovlp"))
    import make_golden_track as M
    c = M.make_case(71, 5, 2500, 12, n_ctg=4, reads_per_ctg=500, ctg_len=300_000, mean_len=9000)
    enc = lambda k: c[k].encode()
    files = [f.encode() for f in c["files"]]
    exp = oracle_lib.track_reads(oracle, files, enc("phased_reads"), enc("read_to_contig_map"), enc("rawread_ids"), 2500, 12)
    out = _lib.track_reads(eng, files, enc("phased_reads"), enc("read_to_contig_map"), enc("rawread_ids"), 2500, 12)
    assert out == exp and out.count(b"\n") > 1500


def test_errors(eng):
    from falcon_unzip_amd import _lib
    c = G.load_track("t1_basic")
    args = (c["phased_reads"], c["read_to_contig_map"], c["rawread_ids"], 2500, 40)
    with pytest.raises(_lib.FzpError):               # every line is parsed before any test: a bad integer always raises
        _lib.track_reads(eng, [c["files"][0] + b"000000001 000000002 -5000 99.0 0 0 x 9000 0 0 5000 9000 overlap\n"], *args)
    with pytest.raises(_lib.FzpError):               # t beyond rawread_ids: IndexError at :50
        q = c["read_to_contig_map"].split()[1]       # an A-read that is in rid_to_ctg (the test at :46 comes first)
        _lib.track_reads(eng, [q + b" 000999999 -5000 99.0 0 0 5000 9000 0 0 5000 9000 overlap\n"], *args)
    assert _lib.track_reads(eng, [], *args) == b""
    assert _lib.track_reads(eng, c["files"], c["phased_reads"], c["read_to_contig_map"], c["rawread_ids"], 2500, 0) == b""   # bestn 0: heaps stay empty


def test_cli_dropin(tmp_path):
    c = G.load_track("t2_files_bestn3")
    bindir = tmp_path / "bin"
    bindir.mkdir()
    la = bindir / "LA4Falcon"
    la.write_text("#!/bin/sh\ncat \"$3\"\n")
    la.chmod(la.stat().st_mode | stat.S_IEXEC)
    for k, txt in enumerate(c["files"]):
        d = tmp_path / "0-rawreads" / ("m_%05d" % k)
        d.mkdir(parents=True)
        (d / ("raw_reads.%d.las" % (k + 1))).write_bytes(txt)
    for key in ("phased_reads", "read_to_contig_map", "rawread_ids"):
        (tmp_path / key).write_bytes(c[key])
    env = dict(os.environ, PATH=str(bindir) + os.pathsep + os.environ["PATH"], PYTHONPATH=REPO + os.pathsep + os.environ.get("PYTHONPATH", ""))
    subprocess.check_call([sys.executable, os.path.join(REPO, "scripts", "fc_rr_hctg_track.py"), "--phased-read-file", "phased_reads", "--read-to-contig-map",
                           "read_to_contig_map", "--rawread-ids", "rawread_ids", "--output", "rawread_to_contigs", "--min-len", str(c["params"]["min_len"]),
                           "--bestn", str(c["params"]["bestn"]), "--n-core", "2", "--silent"], cwd=str(tmp_path), env=env)
    assert (tmp_path / "rawread_to_contigs").read_bytes() == c["expected"]
