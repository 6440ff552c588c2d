#!/usr/bin/env python3
"""Writes DESIGN.md's kernel table (between the kernel-table markers) from the LAST profiles/r6<x>_* set.

Inputs, all committed under profiles/ and all made on an MI355X by tools/profile_round.sh:
  r6<x>_bench_kernel_stats.csv   rocprofv3 --kernel-trace --stats of `python3 bench.py --steps 2 --warmup 1 ...` (+ the instrumented pass behind the timed steps) (per-kernel calls, total ns)
  r6<x>_pmc_hbm_bytes.csv        the two --pmc passes (FETCH_SIZE / WRITE_SIZE) of one step, corrected as tools/pmc_hbm_summary.py says
  r6<x>_bench_line.json          the bench line of the same tree (HIP-event brackets, ms per step)
What a kernel does, what bounds it and its algorithmic bytes per unit are the catalogue below -- statements about the code, kept
here so that the numbers beside them can never be older than the profile set the table names.

  python tools/design_kernel_table.py            rewrite the table in DESIGN.md
  python tools/design_kernel_table.py --check    exit 1 if DESIGN.md's table is not what the last set gives (tests/test_host_logic.py)
"""
import csv
import glob
import json
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BEGIN, END = "<!-- kernel-table:begin (tools/design_kernel_table.py) -->", "<!-- kernel-table:end -->"

# (row name, kernels, what, bound, algorithmic bytes or operations per unit)
CATALOGUE = [
    ("K1 DP `k1_sw`", ["k_swb", "k_sw", "k_swb_units"],
     "adaptive banded DP of every extension piece in one launch; bit-sliced, a piece per lane, the band's 32 cells in the bits of a word (5b); `k_sw` (a wave per piece) beside it for pieces narrower than the band",
     "VALU issue (one persistent wave per SIMD)",
     "SURVEY 8d: 0.25 B/cell; written: 8 B per band step (the whole D and G masks); 1.37 wave64 VALU instructions per band step"),
    ("K1 walk `k1_traceback`", ["k_tb_walk_h", "k_tb_walk", "k_fail_plan"],
     "a lane per piece, 32 walkers per wave in step (one 8 KB run of the masks per wave and iteration, staged through LDS), 2-bit op stream out",
     "HBM (scattered 512-byte pieces, latency of the staged run)",
     "8 B per step below the path's end in, 0.25 B per step out"),
    ("K1 seeds `k1_seed`", ["k_seed", "k_chain"],
     "anchored k-mers looked up in 32-byte buckets, ordered hit list, vote bins, two windows; a wave per (read, window) chains the hits: anchor + waypoints",
     "HBM random access (`k_seed`), instruction issue (`k_chain`)",
     "32 B per probe (57 fetched: one 64-B request), 8 B per hit"),
    ("K1 path `k1_cigar`, `k1_join`", ["k_tb_cigar", "k_join", "k_pick_offsets"],
     "candidate pick, joined op stream (a wave per read); best-scoring stretch of the path (5c), run-length CIGAR, summary, identity gate, hand-off checkpoints",
     "latency / instruction issue (a wave per read)",
     "0.25 B per op + 0.25 B per base in; 4 B per run + 8 B per 256 ops out"),
    ("K1 index `k1_index`", ["k_index_stage_anch", "k_index_build"],
     "anchored k-mers of the contigs staged into 64 KB partitions, partition tables built in LDS and written as images",
     "HBM streaming", "8 B per selected k-mer staged, 64 KB per partition out"),
    ("K1 plans", ["k_slot_count", "k_slot_emit", "k_sort_hist", "k_sort_scatter", "k_route", "k_lists", "k_plan_final", "k_cand", "k_plan_keys", "k_plan_rank", "k_plan_emit", "k_rank_allpairs", "k_range_n"],
     "pieces as 32-byte slots, longest first, streams planned in launch order; records ranked by (POS, read)",
     "launch latency", "32 B per slot"),
    ("K2 `k2_*`", ["k_pileup_pk", "k_vmap_pk", "k_site_flag", "k_site_emit", "k_vmap_len", "k_vmap_put", "k_tile_tables", "k_copy_ref", "k_site_begin_blk", "k_vmap_row_begin"],
     "pileup from K1's packed records (a lane per 16-op word, LDS counters per tile), het call per position, variant_map rows",
     "memory latency / LDS atomics", "0.25 B per op + 0.25 B per base in, 20 B per live position out"),
    ("K3 `k3_*`", ["k_site_sets", "k_assoc", "k_assoc_compact", "k_arow_len", "k_arow_put", "k_arow_begin"],
     "per site the two alleles' sorted q_id sets; four intersections per site pair inside the 65 536 bp window",
     "latency (small)", "24 B per row out"),
    ("K4 `k4_*`", ["k_link_flag", "k_link_emit", "k_left_fill", "k_pj_init", "k_pj_resolve", "k_sweep", "k_extents", "k_segment", "k_pv_compact"],
     "links CSR, greedy start by pointer jumping, the ten sweeps 64 sites per step, extents, segmentation 256 sites per step",
     "latency (a contig is a few waves)", "16 B per link"),
    ("K5 `k5_*`", ["k_read_votes", "k_read_flag", "k_read_emit", "k_pread_begin"],
     "votes per (read, block), phase per read", "latency (small)", "12 B per set entry"),
    ("scans, fills, fetches", ["k_scan_small_u64", "k_apply_u32", "k_apply_sums_u32", "k_tile_sums", "k_tile_sums_u64", "k_apply_u64", "k_fill_regions", "k_fetch_post", "k_u32_to_i64_begin"],
     "ordered compaction (no output order rests on atomics); fills; count read-backs through mapped memory",
     "launch latency", "-"),
    ("runtime copies", ["__amd_rocclr_copyBuffer", "__amd_rocclr_fillBufferAligned"],
     "records and 30 MB of device-made text to pinned host memory", "PCIe", "-"),
    ("job set-up (outside the step)", ["k_pack", "k_revcomp", "k_upper"],
     "2-bit packing of reads and contigs, reverse complements, once per job", "HBM streaming", "1 B per base in, 0.25 out"),
]


def last_tag():
    tags = sorted({re.match(r"(r6[a-z])_", os.path.basename(p)).group(1) for p in glob.glob(os.path.join(ROOT, "profiles", "r6?_bench_kernel_stats.csv"))})
    if not tags:
        raise SystemExit("no profiles/r6?_bench_kernel_stats.csv")
    return tags[-1]


def base_name(n):
    n = re.sub(r"\(anonymous namespace\)::", "", n)
    n = re.sub(r"^void ", "", n)
    return re.match(r"[A-Za-z_0-9]+", n).group(0)


def table():
    tag = last_tag()
    P = lambda f: os.path.join(ROOT, "profiles", "%s_%s" % (tag, f))
    tot, calls = {}, {}
    for r in csv.DictReader(open(P("bench_kernel_stats.csv"))):
        b = base_name(r["Name"])
        tot[b] = tot.get(b, 0) + int(r["TotalDurationNs"])
        calls[b] = calls.get(b, 0) + int(r["Calls"])
    steps = calls.get("k_tb_cigar", 1)          # one launch per step
    hbm = {}
    if os.path.exists(P("pmc_hbm_bytes.csv")):
        rows = [l for l in open(P("pmc_hbm_bytes.csv")) if not l.startswith("#")]
        for r in csv.DictReader(rows):
            hbm[r["kernel"]] = (float(r["hbm_bytes"]), float(r["duration_ms_under_pmc"]))       # one step's launches together
    line = json.load(open(P("bench_line.json")))
    out = [BEGIN,
           "Profile set **`profiles/%s_*`** (`bench.py` line of the same tree: %.2f ms per step, %.2f M reads/s; kernel times = rocprofv3 `--kernel-trace --stats` totals / %d steps; HBM bytes = the two `--pmc` passes of one step, corrected per `tools/pmc_hbm_summary.py`)." % (
               tag, line["ms_per_step"], line["value"] / 1e6, steps),
           "",
           "| stage | kernels (ms per step each) | what | bound | algorithmic bytes per unit | ms/step | HBM moved per step | at |",
           "|---|---|---|---|---|---|---|---|"]
    seen, total = set(), 0.0
    for name, ks, what, bound, alg in CATALOGUE:
        parts, ms, by, dur = [], 0.0, 0.0, 0.0
        for k in ks:
            seen.add(k)
            if k in tot:
                per = tot[k] / 1e6 / steps if k not in ("k_pack", "k_revcomp", "k_upper") else tot[k] / 1e6
                ms += per
                if per >= 0.02:
                    parts.append("`%s` %.2f" % (k, per))
            if k in hbm:
                by += hbm[k][0]
                dur += hbm[k][1]
        if name.startswith("job set-up"):
            ms_s = "(%.2f once)" % ms
        else:
            ms_s = "**%.2f**" % ms
            total += ms
        out.append("| %s | %s | %s | %s | %s | %s | %s | %s |" % (
            name, ", ".join(parts) or "-", what, bound, alg, ms_s,
            ("%.2f GB" % (by / 1e9)) if by >= 5e6 else "-", ("%.1f TB/s" % (by / dur / 1e9)) if by >= 2e8 and dur > 0 else "-"))
    rest = sorted(((tot[k] / 1e6 / steps, k) for k in tot if k not in seen), reverse=True)
    if rest:
        out.append("| not in a row above | %s | | | | %.2f | | |" % (", ".join("`%s` %.3f" % (k, v) for v, k in rest[:8]), sum(v for v, _ in rest)))
        total += sum(v for v, _ in rest)
    out.append("| **all kernels of a step** | | | | | **%.2f** | | |" % total)
    out.append("")
    kt = json.load(open(P("kt_bench_line.json")))["ms_per_step"] if os.path.exists(P("kt_bench_line.json")) else None
    br = line.get("kernel_ms_per_step", {})
    out.append("Kernel times are the tracer's (a traced step takes %s ms of wall time, the untraced line %.2f ms); the untraced line's HIP-event brackets for the same stages: %s.  "
               "The time in which nothing runs on the GPU is listed by `tools/step_timeline.py` (section 8)." % (
                   ("%.1f" % kt) if kt else "more", line["ms_per_step"], ", ".join("`%s` %.2f" % (k, br[k]) for k in ("k1_sw", "k1_traceback", "k1_seed", "k1_cigar", "k1_index", "k2_pileup_count") if k in br)))
    out.append(END)
    return "\n".join(out)


def main():
    path = os.path.join(ROOT, "DESIGN.md")
    text = open(path).read()
    a, b = text.index(BEGIN), text.index(END) + len(END)
    new = text[:a] + table() + text[b:]
    if "--check" in sys.argv:
        if new != text:
            print("DESIGN.md's kernel table is not what profiles/%s_* gives: run tools/design_kernel_table.py" % last_tag())
            raise SystemExit(1)
        return
    open(path, "w").write(new)
    print("kernel table written from profiles/%s_*" % last_tag())


if __name__ == "__main__":
    main()
