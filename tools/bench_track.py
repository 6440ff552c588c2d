#!/usr/bin/env python3
"""Throughput of the raw-read tracker (fzp_track_reads) on a synthetic LA4Falcon -m dump, oracle port beside it.
usage: python tools/bench_track.py [--reads-per-ctg N] [--contigs C] [--reps R]  -> one JSON line"""
import argparse
import json
import os
import sys
import time

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
sys.path.insert(0, os.path.join(REPO, "tests", "golden_ovlp"))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--contigs", type=int, default=8)
    ap.add_argument("--reads-per-ctg", type=int, default=2500)
    ap.add_argument("--reps", type=int, default=3)
    args = ap.parse_args()
    from falcon_unzip_amd import _lib
    from tests import oracle_lib
    import make_golden_track as M
    c = M.make_case(98, 8, 2500, 40, n_ctg=args.contigs, reads_per_ctg=args.reads_per_ctg, ctg_len=600_000, mean_len=9000)
    files = [f.encode() for f in c["files"]]
    pr, rc, ri = c["phased_reads"].encode(), c["read_to_contig_map"].encode(), c["rawread_ids"].encode()
    n_lines = sum(f.count(b"\n") for f in files)
    eng = _lib.Engine(0)
    t_tot = 0.0
    for rep in range(args.reps + 1):
        if rep == 1:
            t_tot = 0.0
            eng.prof_reset(); eng.prof_enable(True)
        t0 = time.perf_counter()
        out = _lib.track_reads(eng, files, pr, rc, ri, 2500, 40)
        t_tot += time.perf_counter() - t0
    eng.prof_enable(False)
    prof = {k: round(v[0] / args.reps, 3) for k, v in eng.prof().items() if k.startswith(("trk_", "ovl_"))}
    t0 = time.perf_counter()
    exp = oracle_lib.track_reads(oracle_lib.load(), files, pr, rc, ri, 2500, 40)
    t_orc = time.perf_counter() - t0
    assert exp == out
    print(json.dumps({"lines": n_lines, "bytes": sum(len(f) for f in files), "out_rows": out.count(b"\n"), "call_ms": round(t_tot / args.reps * 1e3, 2),
                      "lines_per_s": round(n_lines / (t_tot / args.reps)), "kernel_ms": prof, "oracle_ms": round(t_orc * 1e3, 2),
                      "oracle_lines_per_s": round(n_lines / t_orc), "parity": "byte-identical (canonical order)"}))
    eng.close()


if __name__ == "__main__":
    main()
