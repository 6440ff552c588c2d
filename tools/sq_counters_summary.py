#!/usr/bin/env python3
"""rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_WAVES (one pass) -> per-kernel sums as JSON, and -- given the bench line of the
same workload -- profiles/k1_sw_counters.json: VALU / SALU / scalar-memory instructions of the two DP kernels (k_swb, k_sw) per 64-cell band step, which bench.py's roofline reads.
usage: sq_counters_summary.py <counter_collection.csv> <out.json> [<bench_line.json> <k1_sw_counters.json> <source label>]"""
import collections
import csv
import json
import sys

from pmc_hbm_summary import kernel_key


def main():
    tot = collections.defaultdict(lambda: collections.Counter())
    calls = collections.Counter()
    for r in csv.DictReader(open(sys.argv[1])):
        k = kernel_key(r["Kernel_Name"])
        tot[k][r["Counter_Name"]] += float(r["Counter_Value"])
        if r["Counter_Name"] == "SQ_WAVES":
            calls[k] += 1
    json.dump({k: dict(v, launches=calls[k]) for k, v in tot.items()}, open(sys.argv[2], "w"), indent=1)
    if len(sys.argv) > 4:
        line = json.load(open(sys.argv[3]))
        # band steps of ONE bench step: forward extensions of all candidates + the backward extensions (fzp_aln_summary.cells counts them all); the counter pass
        # ran exactly one step (bench.py --steps 1 --warmup 0), so the sums over every k_sw launch of the pass belong to these steps
        band = line.get("roofline", {}).get("band_cells", 64)      # fzalign v1.8: 32 cells per band step (v1.7: 64)
        steps = line["dp_cells_per_step"] / float(band)
        sw = tot["k_sw"] + tot["k_swb"]          # the two DP kernels: the bit-sliced one (a read per lane) and the wave-per-read one
        calls["k_sw"] += calls["k_swb"]
        json.dump({"kernel": "k_swb + k_sw", "band": band, "band_steps_per_bench_step": steps, "valu_k_swb": tot["k_swb"]["SQ_INSTS_VALU"], "valu_k_sw": tot["k_sw"]["SQ_INSTS_VALU"], "valu_per_step": round(sw["SQ_INSTS_VALU"] / steps, 3),
                   "salu_per_step": round(sw["SQ_INSTS_SALU"] / steps, 3), "smem_per_step": round(sw["SQ_INSTS_SMEM"] / steps, 3),
                   "launches_counted": calls["k_sw"], "source": sys.argv[5] if len(sys.argv) > 5 else sys.argv[2]}, open(sys.argv[4], "w"), indent=1)


if __name__ == "__main__":
    main()
