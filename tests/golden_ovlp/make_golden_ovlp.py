#!/usr/bin/env python3
"""Generate the overlap-filter golden vectors under tests/golden_ovlp/ by RUNNING THE REFERENCE.

Runs only in the build container (needs /root/reference).  falcon_unzip/ovlp_filter_with_phase.py is Python 2; it is
translated IN MEMORY with lib2to3 and run with two stand-ins injected into its namespace:
  * `sp`   -- `check_output("LA4Falcon -mo <db> <fn>")` returns the text of <fn> (the reference only reads the tool's
              text lines, ovlp_filter_with_phase.py:60,149,196), as `str` like Python 2 would have it;
  * `Pool` -- an in-process pool whose `imap` is an ordered map (the real one is ordered too) and which records what
              each stage returned, so the fixtures also pin the ignore / contained sets.
Only DATA is written: the synthetic inputs, the parameters and the reference's outputs.

Usage:  python tests/golden_ovlp/make_golden_ovlp.py [case ...]
"""
from __future__ import annotations

import contextlib
import gzip
import hashlib
import io
import json
import os
import sys
import types

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(os.path.dirname(HERE))
REF = "/root/reference/falcon_unzip/ovlp_filter_with_phase.py"
sys.path.insert(0, REPO)

from falcon_unzip_amd import sim_ovlp  # noqa: E402


def load_reference():
    from lib2to3 import refactor
    tool = refactor.RefactoringTool(refactor.get_fixers_from_package("lib2to3.fixes"))
    src = open(REF).read()
    code = str(tool.refactor_string(src + "\n", REF))
    mod = types.ModuleType("ref_ovlp_filter")
    exec(compile(code, "<reference ovlp_filter_with_phase.py, translated in memory>", "exec"), mod.__dict__)
    return mod


class FakeSP(object):
    @staticmethod
    def check_output(cmd):
        assert cmd[0] == "LA4Falcon" and cmd[1] == "-mo", cmd
        with open(cmd[3]) as f:
            return f.read()


class FakePool(object):
    log = {}

    def __init__(self, n):
        pass

    def imap(self, func, inputs):
        res = [func(x) for x in inputs]
        FakePool.log.setdefault(func.__name__, []).extend(res)
        return res


def run_reference(files, rid_map, params, workdir):
    """files: list of text; -> (stdout text, ignore list (per file, in order), contained set)"""
    mod = load_reference()
    mod.sp = FakeSP
    mod.Pool = FakePool
    FakePool.log = {}
    fns = []
    for k, txt in enumerate(files):
        fn = os.path.join(workdir, "ovl.%d.las" % k)
        with open(fn, "w") as f:
            f.write(txt)
        fns.append(fn)
    fofn = os.path.join(workdir, "las.fofn")
    with open(fofn, "w") as f:
        f.write("\n".join(fns) + "\n")
    mp = os.path.join(workdir, "rid_to_phase.all")
    with open(mp, "w") as f:
        f.write(rid_map)
    argv = ["fc_ovlp_filter_with_phase.py", "--fofn", fofn, "--db", "raw_reads.db", "--rid_phase_map", mp,
            "--max_diff", str(params["max_diff"]), "--max_cov", str(params["max_cov"]), "--min_cov", str(params["min_cov"]),
            "--min_len", str(params["min_len"]), "--bestn", str(params["bestn"]), "--n_core", "2"]
    buf = io.StringIO()
    with contextlib.redirect_stdout(buf):
        mod.main(argv)
    ignore = []
    for fn, lst in FakePool.log.get("filter_stage1", []):
        ignore.extend("None" if x is None else x for x in lst)
    contained = set()
    for fn, st in FakePool.log.get("filter_stage2", []):
        contained.update(st)
    return buf.getvalue(), ignore, sorted(contained)


# --------------------------------------------------------------------------- cases
def case_sim(seed, n_files, params, **kw):
    rng = sim_ovlp.rng_for(seed)
    drop = kw.pop("drop_frac", 0.05)
    gen = {k: kw.pop(k) for k in list(kw) if k in ("min_ovl", "noise_pairs", "low_idt_frac", "dup_frac", "odd_tag_frac")}
    reads = sim_ovlp.make_reads(rng, **kw)
    lines = sim_ovlp.overlap_lines(reads, rng, **gen)
    return sim_ovlp.split_files(lines, n_files), sim_ovlp.rid_phase_map_text(reads, drop, rng), params


def case_quirks():
    """Hand-written rows for the order- and text-dependent corners."""
    m = ["000000001 000000F 1 0", "000000002 000000F 1 0", "000000003 000000F 1 1", "000000004 000000F 2 0",
         "000000005 000000F -1 0", "000000006 000001F 1 0", "000000007 000000F 1 0", "000000008 000000F 1 0",
         "000000009 000000F -1 0", "000000010 000000F 1 0", "000000011 000000F 1 0", "000000012 000000F 1 0",
         "000000001 000000F 1 0",           # a later duplicate row overwrites
         "13 000000F 1 0",                  # a key that is not zero-padded
         "000000014 000000F 1 0", "000000015 000000F 1 0", "000000016 000000F 1 0"]
    rid_map = "\n".join(m) + "\n"

    def row(q, t, ln, idt, qs, qe, ql, ts, te, tl, tag, tstrand=0):
        return "%s %s %d %s 0 %d %d %d %d %d %d %d %s" % (q, t, -ln, idt, qs, qe, ql, tstrand, ts, te, tl, tag)
    A, B, C, D, E, F, G, H, I, J, K, L_ = ["%09d" % x for x in range(1, 13)]
    f0 = [
        row("000000099", A, 5000, "99.0", 0, 5000, 9000, 0, 5000, 8000, "overlap"),     # q not in the map: skipped before anything
        row(A, "000000098", 5000, "99.0", 0, 5000, 9000, 0, 5000, 8000, "overlap"),     # t not in the map
        row(A, F, 5000, "99.0", 0, 5000, 9000, 0, 5000, 8000, "overlap"),               # other contig
        row(A, C, 5000, "99.0", 0, 5000, 9000, 0, 5000, 8000, "overlap"),               # same block, other phase
        row(A, B, 5000, "99.0", 0, 5000, 9000, 3000, 8000, 8000, "overlap"),            # 5'
        row(A, D, 4000, "90", 5000, 9000, 9000, 0, 4000, 8000, "overlap"),              # 3', idt exactly 90 passes
        row(A, E, 4000, "89.99", 5000, 9000, 9000, 0, 4000, 8000, "overlap"),           # idt < 90
        row(A, G, 4100, "9e1", 4900, 9000, 9000, 0, 4100, 8000, "overlap"),             # float('9e1') == 90.0
        row(A, H, 3000, "95.5", 0, 3000, 9000, 0, 3000, 2499, "overlap"),               # t shorter than min_len
        row(A, I, 9000, "97.25", 0, 9000, 9000, 100, 9100, 12000, "contained"),         # both ends: 5' and 3' counted; contained
        row(B, A, 5000, "99.0", 3000, 8000, 8000, 0, 5000, 9000, "overlap"),
        row(B, "000000014", 3000, "98.0", 0, 3000, 8000, 5000, 8000, 8000, "overlap"),
        row("000000014", "000000015", 8000, "98.0", 0, 8000, 8000, 500, 8500, 9000, "contained"),   # 14 is contained
        row("000000014", "000000016", 2600, "98.0", 5400, 8000, 8000, 0, 2600, 2600, "contains"),    # ... and contains 16
        row(A, J, 2000, "96.0", 0, 2000, 9000, 7000, 9000, 9000, "overlap"),            # q = A again: a second group for A
        row(A, K, 2000, "96.0", 7000, 9000, 9000, 0, 2000, 2600, "overlap"),
    ]
    # ties: same (-inphase, -len, m_range); Python falls through to comparing the token lists
    f1 = [
        row(G, A, 3000, "97.0", 0, 3000, 7000, 6000, 9000, 9000, "overlap"),
        row(G, B, 3000, "96.0", 0, 3000, 7000, 5000, 8000, 8000, "overlap", 1),
        row(G, B, 3000, "96.0", 0, 3000, 7000, 5000, 8000, 8000, "overlap", 0),         # same pair twice, differs at token 8
        row(G, B, 3000, "100.0", 0, 3000, 7000, 5000, 8000, 8000, "overlap"),           # "100.0" < "96.0" as strings
        row(G, J, 3000, "96.0", 0, 3000, 7000, 6000, 9000, 9000, "overlap"),
        row(G, E, 3000, "96.0", 0, 3000, 7000, 6000, 9000, 9000, "overlap"),            # unphased partner: not in phase with G
        row(G, "13", 3100, "96.0", 0, 3100, 7000, 0, 3100, 9000, "overlap"),
        row(G, L_, 2900, "96.0", 4100, 7000, 7000, 0, 2900, 3500, "overlap"),           # m_range 600 <= 1000
        row(G, K, 2500, "96.0", 4500, 7000, 7000, 0, 2500, 2600, "overlap"),            # m_range 100
        row(G, H, 2400, "96.0", 4600, 7000, 7000, 0, 2400, 9000, "overlap"),
        row(G, D, 2300, "96.0", 4700, 7000, 7000, 0, 2300, 9000, "overlap"),
        row(G, I, 2200, "96.0", 4800, 7000, 7000, 0, 2200, 12000, "overlap"),
    ]
    f1.append(row(J, G, 3000, "97.0", 0, 3000, 9000, 4000, 7000, 7000, "overlap"))       # J: 5' only -> 3' count 0 < min_cov -> ignored
    f2 = []                                                                              # an empty dump
    f3 = [row("000000098", "000000097", 5000, "99.0", 0, 5000, 9000, 0, 5000, 8000, "overlap")]   # nothing passes
    files = ["".join(l + "\n" for l in f) for f in (f0, f1, f2, f3)]
    return files, rid_map, dict(max_diff=10, max_cov=10, min_cov=1, min_len=2500, bestn=3)


CASES = {
    "o1_basic": lambda: case_sim(1, 1, dict(max_diff=40, max_cov=60, min_cov=1, min_len=2500, bestn=10),
                                 n_ctg=2, reads_per_ctg=60, ctg_len=100_000, mean_len=8000),
    "o2_files": lambda: case_sim(2, 3, dict(max_diff=12, max_cov=30, min_cov=2, min_len=4000, bestn=3),
                                 n_ctg=3, reads_per_ctg=50, ctg_len=80_000, mean_len=8000, unphased_frac=0.5),
    "o3_quirks": case_quirks,
    "o4_mincov0": lambda: case_sim(4, 2, dict(max_diff=1000, max_cov=1000, min_cov=0, min_len=0, bestn=0),
                                   n_ctg=1, reads_per_ctg=70, ctg_len=80_000, mean_len=7000, drop_frac=0.3),
    "o5_dense": lambda: case_sim(5, 2, dict(max_diff=25, max_cov=35, min_cov=3, min_len=2500, bestn=5),
                                 n_ctg=1, reads_per_ctg=110, ctg_len=50_000, mean_len=9000, unphased_frac=0.2, dup_frac=0.1),
}


def sha(b):
    return hashlib.sha256(b).hexdigest()


def main():
    import tempfile
    names = sys.argv[1:] or list(CASES)
    manifest_fn = os.path.join(HERE, "manifest.json")
    manifest = json.load(open(manifest_fn)) if os.path.exists(manifest_fn) else {}
    for name in names:
        files, rid_map, params = CASES[name]()
        with tempfile.TemporaryDirectory() as wd:
            out, ignore, contained = run_reference(files, rid_map, params, wd)
        d = os.path.join(HERE, name)
        os.makedirs(d, exist_ok=True)
        for k, txt in enumerate(files):
            with gzip.GzipFile(os.path.join(d, "ovl.%d.txt.gz" % k), "wb", mtime=0) as f:
                f.write(txt.encode())
        with open(os.path.join(d, "rid_to_phase.all"), "w") as f:
            f.write(rid_map)
        with gzip.GzipFile(os.path.join(d, "expected.out.gz"), "wb", mtime=0) as f:
            f.write(out.encode())
        json.dump(dict(params=params, n_files=len(files), ignore=ignore, contained=contained), open(os.path.join(d, "case.json"), "w"), indent=1)
        manifest[name] = dict(n_files=len(files), n_lines=sum(t.count("\n") for t in files), n_out=out.count("\n"),
                              n_ignore=len(ignore), n_contained=len(contained), sha256_out=sha(out.encode()))
        print(name, manifest[name])
    json.dump(manifest, open(manifest_fn, "w"), indent=1, sort_keys=True)


if __name__ == "__main__":
    main()
