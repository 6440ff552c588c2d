"""Drop-in for `falcon_unzip.ovlp_filter_with_phase` (falcon_unzip/ovlp_filter_with_phase.py:279-354).

Same command line, same stdout.  The reference runs `LA4Falcon -mo <db> <las>` three times per file and filters
the text in Python with `--n_core` worker processes; here every dump is read once (`--n_core` threads only wait on
the LA4Falcon children), tokenised by libfzphase and filtered on the GPU (fzp_ovl_parse / fzp_ovl_filter /
fzp_ovl_format, include/fzphase.h).
"""
from __future__ import annotations

import argparse
import os
import shlex
import subprocess as sp
import sys
from multiprocessing.pool import ThreadPool

from . import _lib


def parse_args(argv):
    # flags, types, defaults and help strings as in the reference (:279-294)
    parser = argparse.ArgumentParser(description='a simple multi-processes LAS ovelap data filter')
    parser.add_argument('--n_core', type=int, default=4, help='number of processes used for generating consensus')
    parser.add_argument('--fofn', type=str, help='file contains the path of all LAS file to be processed in parallel')
    parser.add_argument('--db', type=str, help='read db file path')
    parser.add_argument('--max_diff', type=int, help="max difference of 5' and 3' coverage")
    parser.add_argument('--max_cov', type=int, help="max coverage of 5' or 3' coverage")
    parser.add_argument('--min_cov', type=int, help="min coverage of 5' or 3' coverage")
    parser.add_argument('--min_len', type=int, default=2500, help="min length of the reads")
    parser.add_argument('--bestn', type=int, default=10, help="output at least best n overlaps on 5' or 3' ends if possible")
    parser.add_argument('--rid_phase_map', type=str, help="the file that encode the relationship of the read id to phase blocks", required=True)
    return parser.parse_args(argv[1:])


def dump_las(db_fn, fn):
    """The text the reference's three stages iterate over (:60, :149, :196)."""
    return sp.check_output(shlex.split("LA4Falcon -mo %s %s" % (db_fn, fn)))


def filter_dumps(files, rid_map, max_diff, max_cov, min_cov, min_len=2500, bestn=10, device=None, eng=None):
    """files: list of bytes (one LA4Falcon -mo dump per .las file, fofn order) -> the bytes the reference prints."""
    own = eng is None
    if own:
        if device is None:
            device = int(os.environ.get("FZP_DEVICE", os.environ.get("LOCAL_RANK", "0")))
        eng = _lib.Engine(device)
    ovl = _lib.OvlSet(eng, files, rid_map)
    try:
        rows, _, _ = _lib.ovl_filter(eng, ovl, max_diff, max_cov, min_cov, min_len, bestn)
        return ovl.format(rows)
    finally:
        ovl.close()
        if own:
            eng.close()


def main(argv=sys.argv):
    args = parse_args(argv)
    for name in ("max_diff", "max_cov", "min_cov"):
        if getattr(args, name) is None:          # the reference compares ints with None (Python 2 ordering): refuse instead
            raise SystemExit("--%s is required" % name)
    with open(args.rid_phase_map, "rb") as f:
        rid_map = f.read()
    with open(args.fofn) as f:
        file_list = [fn for fn in f.read().split("\n") if len(fn) != 0]
    with ThreadPool(max(1, args.n_core)) as pool:
        dumps = pool.map(lambda fn: dump_las(args.db, fn), file_list)
    out = filter_dumps(dumps, rid_map, args.max_diff, args.max_cov, args.min_cov, args.min_len, args.bestn)
    sys.stdout.buffer.write(out)
    sys.stdout.flush()
