#!/usr/bin/env python3
"""Throughput of the overlap filter (fzp_ovl_*) on a synthetic LA4Falcon -mo dump, with the oracle port beside it.

usage: python tools/bench_ovlp.py [--reads-per-ctg N] [--contigs C] [--reps R]
Prints one JSON line: overlap lines/s for parse (host), filter (device) and the whole call, kernel ms, oracle lines/s.
"""
import argparse
import json
import os
import sys
import time

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--contigs", type=int, default=8)
    ap.add_argument("--reads-per-ctg", type=int, default=2500)
    ap.add_argument("--ctg-len", type=int, default=600_000)
    ap.add_argument("--reps", type=int, default=3)
    args = ap.parse_args()
    from falcon_unzip_amd import _lib, sim_ovlp
    from tests import oracle_lib
    rng = sim_ovlp.rng_for(99)
    reads = sim_ovlp.make_reads(rng, n_ctg=args.contigs, reads_per_ctg=args.reads_per_ctg, ctg_len=args.ctg_len, mean_len=9000)
    lines = sim_ovlp.overlap_lines(reads, rng, dup_frac=0.0)
    files = [f.encode() for f in sim_ovlp.split_files(lines, 8)]
    rid_map = sim_ovlp.rid_phase_map_text(reads, 0.05, rng).encode()
    params = dict(max_diff=100, max_cov=200, min_cov=1, min_len=2500, bestn=10)
    n_lines = len(lines)
    nbytes = sum(len(f) for f in files)
    eng = _lib.Engine(0)
    t_parse = t_filter = t_fmt = 0.0
    for rep in range(args.reps + 1):
        if rep == 1:
            t_parse = t_filter = t_fmt = 0.0
            eng.prof_reset(); eng.prof_enable(True)
        t0 = time.perf_counter()
        ovl = _lib.OvlSet(eng, files, rid_map)
        t1 = time.perf_counter()
        rows, ig, ct = _lib.ovl_filter(eng, ovl, **params)
        t2 = time.perf_counter()
        out = ovl.format(rows)
        t3 = time.perf_counter()
        n_rows = ovl.n_rows
        ovl.close()
        t_parse += t1 - t0; t_filter += t2 - t1; t_fmt += t3 - t2
    eng.prof_enable(False)
    prof = {k: round(v[0] / args.reps, 3) for k, v in eng.prof().items() if k.startswith("ovl_")}
    t0 = time.perf_counter()
    exp, _, _ = oracle_lib.ovlp_filter(oracle_lib.load(), files, rid_map, params)
    t_orc = time.perf_counter() - t0
    assert exp == out
    R = args.reps
    print(json.dumps({"lines": n_lines, "rows": n_rows, "bytes": nbytes, "selected": int(len(rows)),
                      "parse_ms": round(t_parse / R * 1e3, 2), "filter_ms": round(t_filter / R * 1e3, 2), "format_ms": round(t_fmt / R * 1e3, 2),
                      "lines_per_s_total": round(n_lines / ((t_parse + t_filter + t_fmt) / R)), "lines_per_s_filter": round(n_lines / (t_filter / R)),
                      "kernel_ms": prof, "oracle_ms": round(t_orc * 1e3, 2), "oracle_lines_per_s": round(n_lines / t_orc), "parity": "byte-identical"}))
    eng.close()


if __name__ == "__main__":
    main()
