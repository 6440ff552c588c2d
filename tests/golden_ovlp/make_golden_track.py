#!/usr/bin/env python3
"""Golden vectors for the raw-read tracker (falcon_unzip/rr_hctg_track.py) by RUNNING THE REFERENCE.

Container only (needs /root/reference).  The module is translated in memory with lib2to3 and given stand-ins for the
two falcon_kit modules it imports: `falcon_kit.multiproc.Pool` (in-process, ordered imap) and `falcon_kit.util.io`
(`run_func`, `LOG`, `logstats`, and a reader context whose `readlines()` returns the text of the .las stand-in file --
the reference only reads `LA4Falcon -m` text lines, rr_hctg_track.py:26-29,38).  `run_track_reads` (:68-139) is called
directly with an explicit file list (the CLI globs 0-rawreads/m*/raw_reads.*.las, :148).

The output's line order is a Python dict order (:111) and so is the order of contigs with equal score (:125-126):
fixtures store the CANONICAL form -- lines sorted by (read, score, contig), ranks re-assigned in that order.
Only data is written.
"""
from __future__ import annotations

import gzip
import hashlib
import json
import os
import sys
import tempfile
import types

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(os.path.dirname(HERE))
REF = "/root/reference/falcon_unzip/rr_hctg_track.py"
sys.path.insert(0, REPO)

from falcon_unzip_amd import sim_ovlp  # noqa: E402


def canonical(text: str) -> str:
    rows = [l.split() for l in text.splitlines() if l.strip()]
    rows.sort(key=lambda r: (r[0], int(r[4]), r[1]))
    out, prev, rank = [], None, 0
    for r in rows:
        rank = rank + 1 if r[0] == prev else 0
        prev = r[0]
        out.append(" ".join([r[0], r[1], r[2], str(rank), r[4], r[5]]))
    return "".join(l + "\n" for l in out)


def load_reference():
    from lib2to3 import refactor
    tool = refactor.RefactoringTool(refactor.get_fixers_from_package("lib2to3.fixes"))
    code = str(tool.refactor_string(open(REF).read() + "\n", REF))

    class Reader(object):
        def __init__(self, cmd):
            self.fn = cmd.split()[-1]

        def __enter__(self):
            return self

        def __exit__(self, *a):
            return False

        def readlines(self):
            with open(self.fn) as f:
                return f.read().splitlines()

    io = types.ModuleType("falcon_kit.util.io")
    io.CapturedProcessReaderContext = Reader
    io.StreamedProcessReaderContext = Reader
    io.LOG = lambda *a, **k: None
    io.logstats = lambda *a, **k: None
    io.write_nothing = lambda *a, **k: None
    io.run_func = lambda args: args[0](*args[1:])
    mp = types.ModuleType("falcon_kit.multiproc")

    class Pool(object):
        def __init__(self, n=0):
            pass

        def imap(self, f, inputs):
            return [f(x) for x in inputs]

        def terminate(self):
            pass
    mp.Pool = Pool
    fk, fu = types.ModuleType("falcon_kit"), types.ModuleType("falcon_kit.util")
    sys.modules.update({"falcon_kit": fk, "falcon_kit.util": fu, "falcon_kit.util.io": io, "falcon_kit.multiproc": mp})
    fk.util, fk.multiproc, fu.io = fu, mp, io
    mod = types.ModuleType("ref_rr_hctg_track")
    exec(compile(code, "<reference rr_hctg_track.py, translated in memory>", "exec"), mod.__dict__)
    return mod, Pool


def make_case(seed, n_files, min_len, bestn, **kw):
    """-> dict of input texts"""
    rng = sim_ovlp.rng_for(seed)
    reads = sim_ovlp.make_reads(rng, **kw)
    lines = sim_ovlp.overlap_lines(reads, rng, dup_frac=0.03, min_ovl=600)
    files = sim_ovlp.split_files(lines, n_files)
    n = len(reads)
    oid = ["m%05d/%d/0_%d" % (seed, r["rid"], r["end"] - r["start"]) for r in reads]
    rawread_ids = "\n".join(oid) + "\n"                 # .split('\n') leaves a last empty entry: rid n maps to ''
    # phased reads: q_id ctg block phase n0 n1 QNAME  (phasing.py:478-480); a few reads phased in two blocks, some not at all
    pr = []
    for r in reads:
        if r["block"] >= 0:
            pr.append("%d %s %d %d %d %d %s" % (r["rid"], r["ctg"], r["block"], r["phase"], 5, 1, oid[r["rid"]]))
            if rng.random() < 0.05:
                pr.append("%d %s %d %d %d %d %s" % (r["rid"], r["ctg"], r["block"] + 1, 1 - r["phase"], 4, 1, oid[r["rid"]]))   # later row wins (:79)
    # read_to_contig_map: pid rid oid ctg; some reads map to two contigs, some to none
    rc = []
    for r in reads:
        u = rng.random()
        if u < 0.08:
            continue
        rc.append("%09d %09d %s %s" % (r["rid"] + 7, r["rid"], oid[r["rid"]], r["ctg"]))
        if u > 0.92:
            rc.append("%09d %09d %s %s" % (r["rid"] + 7, r["rid"], oid[r["rid"]], r["ctg"][:-1] + "R"))
    return dict(files=files, rawread_ids=rawread_ids, phased_reads="".join(l + "\n" for l in pr),
                read_to_contig_map="".join(l + "\n" for l in rc), params=dict(min_len=min_len, bestn=bestn))


CASES = {
    "t1_basic": lambda: make_case(61, 1, 2500, 40, n_ctg=2, reads_per_ctg=60, ctg_len=100_000, mean_len=8000),
    "t2_files_bestn3": lambda: make_case(62, 3, 4000, 3, n_ctg=2, reads_per_ctg=70, ctg_len=90_000, mean_len=8000, unphased_frac=0.5),
    "t3_dense": lambda: make_case(63, 2, 0, 10, n_ctg=1, reads_per_ctg=120, ctg_len=50_000, mean_len=9000, unphased_frac=0.1),
}


def run_reference(case, wd):
    mod, Pool = load_reference()
    fns = []
    for k, txt in enumerate(case["files"]):
        fn = os.path.join(wd, "raw_reads.%d.las" % k)
        open(fn, "w").write(txt)
        fns.append(fn)
    paths = {}
    for name in ("rawread_ids", "phased_reads", "read_to_contig_map"):
        paths[name] = os.path.join(wd, name)
        open(paths[name], "w").write(case[name])
    out = os.path.join(wd, "rawread_to_contigs")
    mod.run_track_reads(Pool(), paths["phased_reads"], paths["read_to_contig_map"], paths["rawread_ids"], fns, case["params"]["min_len"],
                        case["params"]["bestn"], "raw_reads.db", out)
    return open(out).read()


def main():
    names = sys.argv[1:] or list(CASES)
    mf = os.path.join(HERE, "manifest_track.json")
    manifest = json.load(open(mf)) if os.path.exists(mf) else {}
    for name in names:
        case = CASES[name]()
        with tempfile.TemporaryDirectory() as wd:
            raw = run_reference(case, wd)
        can = canonical(raw)
        d = os.path.join(HERE, name)
        os.makedirs(d, exist_ok=True)
        for k, txt in enumerate(case["files"]):
            with gzip.GzipFile(os.path.join(d, "ovl.%d.txt.gz" % k), "wb", mtime=0) as f:
                f.write(txt.encode())
        for key in ("rawread_ids", "phased_reads", "read_to_contig_map"):
            with gzip.GzipFile(os.path.join(d, key + ".gz"), "wb", mtime=0) as f:
                f.write(case[key].encode())
        with gzip.GzipFile(os.path.join(d, "expected.canonical.gz"), "wb", mtime=0) as f:
            f.write(can.encode())
        json.dump(dict(params=case["params"], n_files=len(case["files"])), open(os.path.join(d, "case.json"), "w"), indent=1)
        manifest[name] = dict(n_files=len(case["files"]), n_lines=sum(t.count("\n") for t in case["files"]), n_out=can.count("\n"),
                              sha256=hashlib.sha256(can.encode()).hexdigest())
        print(name, manifest[name])
    json.dump(manifest, open(mf, "w"), indent=1, sort_keys=True)


if __name__ == "__main__":
    main()
