#!/usr/bin/env python3
"""Summarise two rocprofv3 counter passes (FETCH_SIZE, WRITE_SIZE) into per-kernel HBM bytes.

usage: pmc_hbm_summary.py <fetch_counter_collection.csv> <write_counter_collection.csv> <out.csv> [<k1_sw json>]

Units: the counters report KB.  Per MI355X_MICROARCH.md (HBM section) FETCH_SIZE reports HALF the bytes of a WIDE COALESCED streaming read (16 B per lane) on gfx950 --
and only of those: a kernel that reads 4-8 B per lane, or at random (one 64-B request per 32-B bucket probe), is counted right.  The x2 is therefore applied only to the
kernels listed in WIDE_READERS (their dominant reads are dwordx4 / dwordx2 wave-contiguous loads); both columns are printed.  Rows are summed over a kernel's dispatches.
"""
import collections
import csv
import json
import sys


def kernel_key(name):
    """'void (anonymous namespace)::k_sw<true>(long, ...)' -> 'k_sw'"""
    k = name.replace("(anonymous namespace)::", "").split("(")[0].strip()
    if k.startswith("void "):
        k = k[5:]
    return k.split("<")[0]


# kernels whose reads are wide and wave-contiguous (16 B or 8 B per lane, consecutive lanes consecutive addresses): FETCH_SIZE x 2
WIDE_READERS = {"k_tb_walk", "k_tb_walk_h", "k_gather16", "k_gather", "k_index_build", "k_join", "k_apply_u32", "k_apply_u64", "k_tile_sums", "k_tile_sums_u64", "k_site_flag", "k_site_emit",
                "k_upper", "k_pack", "k_pack2", "k_revcomp", "__amd_rocclr_copyBuffer"}


def fetch_factor(k):
    return 2.0 if k in WIDE_READERS else 1.0


BIGGEST = {}      # (counter, kernel) -> the largest single dispatch's value: k_sw is launched for the forward extensions (the dominant launch) AND the short backward ones


def load(path, name):
    tot, calls, dur = collections.Counter(), collections.Counter(), collections.Counter()
    per = collections.defaultdict(float)
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] != name:
            continue
        k = kernel_key(r["Kernel_Name"])
        tot[k] += float(r["Counter_Value"])
        per[(k, r.get("Dispatch_Id", r.get("Correlation_Id", "")))] += float(r["Counter_Value"])      # (a dispatch's value comes as one row per XCD / dimension)
        calls[k] += 1
        dur[k] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6
    for (k, _), v in per.items():
        BIGGEST[(name, k)] = max(BIGGEST.get((name, k), 0.0), v)
    return tot, calls, dur


def main():
    f, fc, fd = load(sys.argv[1], "FETCH_SIZE")
    w, wc, wd = load(sys.argv[2], "WRITE_SIZE")
    rows = []
    for k in sorted(set(f) | set(w), key=lambda k: -(fetch_factor(k) * f[k] + w[k])):
        rows.append((k, max(fc[k], wc[k]), f[k], w[k], fd[k], (f[k] + w[k]) * 1024, (fetch_factor(k) * f[k] + w[k]) * 1024))
    with open(sys.argv[3], "w") as out:
        out.write("# rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes) -- python3 bench.py --steps 1 --warmup 0 --no-cpu-baseline --gen-workers 1\n")
        out.write("# units: KB as reported.  hbm_bytes_raw = (FETCH_SIZE + WRITE_SIZE)*1024; hbm_bytes = (c*FETCH_SIZE + WRITE_SIZE)*1024 with c = 2 for kernels whose reads are wide and\n")
        out.write("# wave-contiguous (MI355X_MICROARCH.md, HBM section: FETCH_SIZE reports half of those on gfx950), c = 1 for the others (column fetch_x)\n")
        out.write("kernel,calls,FETCH_SIZE_KB,WRITE_SIZE_KB,duration_ms_under_pmc,hbm_bytes_raw,fetch_x,hbm_bytes\n")
        for k, c, a, b_, d, raw, h in rows:
            out.write("%s,%d,%.0f,%.0f,%.3f,%.0f,%d,%.0f\n" % (k, c, a, b_, d, raw, int(fetch_factor(k)), h))
    if len(sys.argv) > 4:
        # the DP stage's dominant dispatch: the largest single one of its kernels (k_swb: the bit-sliced pieces; k_sw: narrow pieces, stragglers); their reads are the 2-bit
        # base streams, a few words per lane at a time: counted as reported
        best = None
        for k, c, a, b_, d, raw, h in rows:
            if k in ("k_sw", "k_swb", "k_swb2"):
                fa, wb = BIGGEST.get(("FETCH_SIZE", k), a / c), BIGGEST.get(("WRITE_SIZE", k), b_ / c)
                if best is None or fa + wb > best[1] + best[2]:
                    best = (k, fa, wb)
        if best:
            k, fa, wb = best
            json.dump({"kernel": k, "bytes_per_launch": (fa + wb) * 1024, "fetch_size_kb": fa, "write_size_kb": wb, "launch": "the largest dispatch (the chunk's bit-sliced pieces)",
                       "source": "%s (rocprofv3 --pmc, separate FETCH_SIZE and WRITE_SIZE passes; FETCH_SIZE as reported: the kernel reads a few words per lane at a time)" % sys.argv[3]},
                      open(sys.argv[4], "w"), indent=1)


if __name__ == "__main__":
    main()
