"""Loader for tests/golden_ovlp/ (outputs of the reference's ovlp_filter_with_phase.py, see make_golden_ovlp.py)."""
from __future__ import annotations

import gzip
import json
import os

HERE = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden_ovlp")


def cases():
    return sorted(json.load(open(os.path.join(HERE, "manifest.json"))))


def load(name):
    d = os.path.join(HERE, name)
    meta = json.load(open(os.path.join(d, "case.json")))
    files = [gzip.open(os.path.join(d, "ovl.%d.txt.gz" % k)).read() for k in range(meta["n_files"])]
    rid_map = open(os.path.join(d, "rid_to_phase.all"), "rb").read()
    expected = gzip.open(os.path.join(d, "expected.out.gz")).read()
    ignore = sorted(set(x for x in meta["ignore"] if x != "None"))
    return dict(files=files, rid_map=rid_map, params=meta["params"], expected=expected, ignore=ignore, contained=sorted(meta["contained"]))


def track_cases():
    return sorted(json.load(open(os.path.join(HERE, "manifest_track.json"))))


def load_track(name):
    d = os.path.join(HERE, name)
    meta = json.load(open(os.path.join(d, "case.json")))
    rd = lambda fn: gzip.open(os.path.join(d, fn)).read()
    return dict(files=[rd("ovl.%d.txt.gz" % k) for k in range(meta["n_files"])], rawread_ids=rd("rawread_ids.gz"), phased_reads=rd("phased_reads.gz"),
                read_to_contig_map=rd("read_to_contig_map.gz"), params=meta["params"], expected=rd("expected.canonical.gz"))
