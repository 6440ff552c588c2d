"""Overlap filter (fzp_ovl_*): HIP path through the C-ABI against the reference-generated goldens and the oracle.
Byte-exact on the printed text; the ignore / contained sets are compared as sets of ids."""
import os
import stat
import subprocess
import sys

import numpy as np
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


def run_hip(eng, files, rid_map, params):
    from falcon_unzip_amd import _lib
    ovl = _lib.OvlSet(eng, files, rid_map)
    rows, ig, ct = _lib.ovl_filter(eng, ovl, params["max_diff"], params["max_cov"], params["min_cov"], params["min_len"], params["bestn"])
    out = ovl.format(rows)
    names = lambda a: sorted(ovl.id_name(i).decode() for i in a)
    res = out, names(ig), names(ct), ovl.n_lines, ovl.n_rows
    ovl.close()
    return res


@pytest.mark.parametrize("name", G.cases())
def test_golden(eng, name):
    c = G.load(name)
    out, ig, ct, n_lines, _ = run_hip(eng, c["files"], c["rid_map"], c["params"])
    assert n_lines == sum(f.count(b"\n") for f in c["files"])
    assert ig == c["ignore"]
    assert ct == c["contained"]
    assert out == c["expected"]


@pytest.mark.parametrize("seed,kw,params", [
    (11, dict(n_ctg=4, reads_per_ctg=400, ctg_len=400_000, mean_len=9000), dict(max_diff=60, max_cov=90, min_cov=1, min_len=2500, bestn=10)),
    (12, dict(n_ctg=2, reads_per_ctg=300, ctg_len=120_000, mean_len=10000, unphased_frac=0.6), dict(max_diff=40, max_cov=120, min_cov=2, min_len=3000, bestn=4)),
    (13, dict(n_ctg=1, reads_per_ctg=500, ctg_len=100_000, mean_len=7000, unphased_frac=0.1), dict(max_diff=10**6, max_cov=10**6, min_cov=0, min_len=0, bestn=2)),
])
def test_vs_oracle_larger(eng, oracle, seed, kw, params):
    from falcon_unzip_amd import sim_ovlp
    rng = sim_ovlp.rng_for(seed)
    reads = sim_ovlp.make_reads(rng, **kw)
    lines = sim_ovlp.overlap_lines(reads, rng, dup_frac=0.05)
    files = [f.encode() for f in sim_ovlp.split_files(lines, 4)]
    rid_map = sim_ovlp.rid_phase_map_text(reads, 0.05, rng).encode()
    exp, eig, ect = oracle_lib.ovlp_filter(oracle, files, rid_map, params)
    out, ig, ct, _, _ = run_hip(eng, files, rid_map, params)
    assert ig == sorted(eig) and ct == sorted(ect)
    assert out == exp
    assert len(exp) > 0


def test_errors(eng):
    from falcon_unzip_amd import _lib
    c = G.load("o3_quirks")
    with pytest.raises(_lib.FzpError):                    # `q_id, t_id = l[:2]` raises in the reference
        _lib.OvlSet(eng, [c["files"][0] + b"000000001\n"], c["rid_map"])
    with pytest.raises(_lib.FzpError):                    # map row with 3 fields: IndexError at :309
        _lib.OvlSet(eng, c["files"], c["rid_map"] + b"000000077 000000F 1\n")
    bad = b"000000001 000000002 -5000 abc 0 0 5000 9000 0 3000 8000 8000 overlap\n"
    ovl = _lib.OvlSet(eng, [bad], c["rid_map"])
    with pytest.raises(_lib.FzpError):                    # float('abc') on a line that passes the phase checks
        _lib.ovl_filter(eng, ovl, 10, 10, 1, 2500, 3)
    ovl.close()
    ok = b"000000001 000000003 -5000 abc\n"               # same block, other phase: the reference never parses it
    ovl = _lib.OvlSet(eng, [ok], c["rid_map"])
    rows, _, _ = _lib.ovl_filter(eng, ovl, 10, 10, 1, 2500, 3)
    assert len(rows) == 0
    ovl.close()
    ovl = _lib.OvlSet(eng, [], b"")                            # nothing at all
    rows, ig, ct = _lib.ovl_filter(eng, ovl, 10, 10, 1, 2500, 3)
    assert len(rows) == 0 and len(ig) == 0 and len(ct) == 0
    ovl.close()


def test_device_and_host_tokenisers_agree(eng, oracle, monkeypatch):
    """Keys shaped '%09d' -> K_tok on the device; FZP_OVL_HOST_TOKENISER forces the host path.  Odd but legal idt
    spellings (exponent, nan, inf, 20 digits) are left to strtod by the device tokeniser."""
    from falcon_unzip_amd import sim_ovlp
    rng = sim_ovlp.rng_for(21)
    reads = sim_ovlp.make_reads(rng, n_ctg=2, reads_per_ctg=150, ctg_len=120_000, mean_len=9000)
    lines = sim_ovlp.overlap_lines(reads, rng, dup_frac=0.03)
    odd = ["9.0e1", "nan", "inf", "89.99999999999999999999", "90.00000000000000000000", "+95.5", "095.5", "95.", "-0.0", "1e2", "0x5A", "9_5"]
    ok_rows = 0
    for k in range(0, len(lines), 7):
        t = lines[k].split()
        t[3] = odd[(k // 7) % (len(odd) - 2)]         # the last two are what float() rejects: kept out of lines that matter
        lines[k] = " ".join(t)
        ok_rows += 1
    lines.append("   ".join(lines[3].split()) + "  \t ")   # extra blanks and a tab: str.split() does not care
    files = [f.encode() for f in sim_ovlp.split_files(lines, 3)]
    rid_map = sim_ovlp.rid_phase_map_text(reads, 0.05, rng).encode()
    params = dict(max_diff=50, max_cov=80, min_cov=1, min_len=2500, bestn=6)
    exp, eig, ect = oracle_lib.ovlp_filter(oracle, files, rid_map, params)
    dev = run_hip(eng, files, rid_map, params)
    monkeypatch.setenv("FZP_OVL_HOST_TOKENISER", "1")
    host = run_hip(eng, files, rid_map, params)
    assert dev == host
    assert dev[0] == exp and dev[1] == sorted(eig) and dev[2] == sorted(ect)
    # a field float() rejects on a line that passes the phase checks is an error on both paths
    q = exp.decode().split("\n")[0].split()[:-2]        # a printed line: it passed every check
    monkeypatch.delenv("FZP_OVL_HOST_TOKENISER")
    from falcon_unzip_amd import _lib
    for bad_idt in ("0x5A", "9_5", ".", "abc"):
        q[3] = bad_idt
        ovl = _lib.OvlSet(eng, [(" ".join(q) + "\n").encode()], rid_map)
        with pytest.raises(_lib.FzpError):
            _lib.ovl_filter(eng, ovl, 50, 80, 1, 2500, 6)
        ovl.close()
        with pytest.raises(oracle_lib.OracleError):
            oracle_lib.ovlp_filter(oracle, [(" ".join(q) + "\n").encode()], rid_map, params)


def test_cli_dropin(tmp_path):
    """scripts/fc_ovlp_filter_with_phase.py with the reference's flags and a stand-in LA4Falcon (`cat $3`)."""
    c = G.load("o2_files")
    bindir = tmp_path / "bin"
    bindir.mkdir()
    la = bindir / "LA4Falcon"
    la.write_text("#!/bin/sh\ncat \"$3\"\n")
    la.chmod(la.stat().st_mode | stat.S_IEXEC)
    fns = []
    for k, txt in enumerate(c["files"]):
        fn = tmp_path / ("ovl.%d.las" % k)
        fn.write_bytes(txt)
        fns.append(str(fn))
    (tmp_path / "las.fofn").write_text("\n".join(fns) + "\n")
    (tmp_path / "rid_to_phase.all").write_bytes(c["rid_map"])
    p = c["params"]
    env = dict(os.environ, PATH=str(bindir) + os.pathsep + os.environ["PATH"], PYTHONPATH=REPO + os.pathsep + os.environ.get("PYTHONPATH", ""))
    out = subprocess.check_output([sys.executable, os.path.join(REPO, "scripts", "fc_ovlp_filter_with_phase.py"), "--fofn", str(tmp_path / "las.fofn"),
                                   "--db", "raw_reads.db", "--rid_phase_map", str(tmp_path / "rid_to_phase.all"), "--max_diff", str(p["max_diff"]),
                                   "--max_cov", str(p["max_cov"]), "--min_cov", str(p["min_cov"]), "--min_len", str(p["min_len"]), "--bestn", str(p["bestn"]),
                                   "--n_core", "2"], env=env)
    assert out == c["expected"]
