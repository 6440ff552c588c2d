#!/usr/bin/env python3
"""rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY  +  --pmc SQ_WAIT_ANY SQ_INSTS_LDS SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD (two passes)
-> per-kernel sums as JSON, with the ratios the DP stage's analysis reads: of a wave's cycles, how many it spent issuing VALU work, how many waiting for an instruction's
operands / on anything.  usage: sq_cycles_summary.py <pass a csv> <pass b csv> <out.json>"""
import collections
import csv
import json
import sys

from pmc_hbm_summary import kernel_key


def main():
    tot = collections.defaultdict(collections.Counter)
    for path in sys.argv[1:3]:
        for r in csv.DictReader(open(path)):
            tot[kernel_key(r["Kernel_Name"])][r["Counter_Name"]] += float(r["Counter_Value"])
    out = {}
    for k, v in tot.items():
        d = dict(v)
        wc = v.get("SQ_WAVE_CYCLES", 0.0)
        if wc:
            d["valu_active_per_wave_cycle"] = round(v.get("SQ_ACTIVE_INST_VALU", 0.0) / wc, 4)
            d["wait_inst_any_per_wave_cycle"] = round(v.get("SQ_WAIT_INST_ANY", 0.0) / wc, 4)
            d["wait_any_per_wave_cycle"] = round(v.get("SQ_WAIT_ANY", 0.0) / wc, 4)
        out[k] = d
    json.dump(out, open(sys.argv[3], "w"), indent=1)


if __name__ == "__main__":
    main()
