#!/usr/bin/env python3
"""The go / no-go table of a 32-cell band (fzalign v1.8; VERDICT r5 item 3's fallback, HISTORY.md section 14): the TWIN first, no kernel touched.  The scalar twin with
`band` = 64 and = 32 (orc_align_params.band), same reads, the quantities the GPU tests hold the aligner to:
    python3 tools/runs/band32_go_nogo.py > profiles/r6_band32_go_nogo.txt
CPU only (a few minutes on 8 cores)."""
import os
import sys

import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, REPO)
from falcon_unzip_amd import sim
from tests import oracle_lib


def rows(name, s, start, end, strand, lens):
    ok = s["aligned"] == 1
    inside = (s["q_end"] - s["q_start"])
    placed = (np.abs(s["pos"] - start) <= 64) & (np.abs(s["ref_end"] - end) <= 64)
    return {"set": name, "aligned": ok.mean(), "strand_ok": float(np.mean(s["strand"][ok] == strand[ok])), "placed_64": float(placed[ok].mean()),
            "bases_inside": float(inside[ok].sum() / lens.sum()), "reads_95_inside": float(np.mean(inside[ok] / lens[ok] >= 0.95)), "reads_999_inside": float(np.mean(inside[ok] / lens[ok] >= 0.999)),
            "score_sum": int(s["score"][ok].sum())}


def main():
    o64 = o32 = oracle_lib.load()
    sets = []
    rng = np.random.Generator(np.random.PCG64(31))
    hap0, hap1, _ = sim.make_diploid(600000, rng)
    reads = sim.simulate_reads(hap0, hap1, 300, 15000, rng, strand_mix=0.5)
    sets.append(("bench shape: 300 x 15 kb CLR, iid errors 13 %", hap0, [sim.codes_to_str(r.raw_seq_codes()).encode() for r in reads],
                 np.array([r.start for r in reads]), np.array([r.start + r.ref_span() for r in reads]), np.array([r.strand for r in reads])))
    rng = np.random.Generator(np.random.PCG64(32))
    hap0r, hap1r, _, spans = sim.make_repeat_diploid(1000000, rng, n_families=20, n_tandem=20)
    reads = sim.simulate_reads(hap0r, hap1r, 300, 15000, rng, strand_mix=0.5)
    sets.append(("repeat genome: 300 x 15 kb", hap0r, [sim.codes_to_str(r.raw_seq_codes()).encode() for r in reads],
                 np.array([r.start for r in reads]), np.array([r.start + r.ref_span() for r in reads]), np.array([r.strand for r in reads])))
    rng = np.random.Generator(np.random.PCG64(47))
    hap0s, hap1s, _ = sim.make_diploid(2_000_000, rng)
    codes, off, st, hp, sd, lens, bf, truth = sim.simulate_raw_reads_shaped(hap0s, hap1s, 600, rng, with_truth=True)
    blob = sim.ACGT[codes].tobytes()
    raw = [blob[off[i]:off[i + 1]] for i in range(600)]
    sets.append(("real shape: 600 reads, 3-60 kb, bursts of 0.3-1 kb at 30 %", hap0s, raw, None, None, sd, (off, st, lens, truth)))
    print("%-62s %5s %8s %8s %9s %9s %9s %9s %12s" % ("set", "band", "aligned", "placed64", "bases_in", ">=95%in", ">=99.9%in", "strand", "score sum"))
    for item in sets:
        name, hap, raw, start, end, strand = item[:6]
        ctg = sim.codes_to_str(hap).encode()
        res = {}
        for band, orc in ((64, o64), (32, o32)):
            s, _ = oracle_lib.align_reads(orc, ctg, raw, {"band": band}, n_threads=8)
            lens = np.array([len(x) for x in raw])
            if start is None:      # real shape: the true first aligned base / end from the simulator's truth table
                off, st, rl, truth = item[6]
                ok = s["aligned"] == 1
                stt = st + truth[off[:-1] + np.clip(s["q_start"], 0, lens - 1)]
                a, b = stt, st + rl
            else:
                a, b = start, end
            res[band] = (s, rows(name, s, a, b, strand, lens))
            r = res[band][1]
            print("%-62s %5d %8.4f %8.4f %9.5f %9.4f %9.4f %9.4f %12d" % (name, band, r["aligned"], r["placed_64"], r["bases_inside"], r["reads_95_inside"], r["reads_999_inside"], r["strand_ok"], r["score_sum"]))
        s64, s32 = res[64][0], res[32][0]
        both = (s64["aligned"] == 1) & (s32["aligned"] == 1)
        worse = both & (s32["score"] < s64["score"])
        print("    band 32 against 64: same score %d, lower %d (median loss %d, largest %d), higher %d; read bases lost from alignments: %d of %d" %
              (int((both & (s32["score"] == s64["score"])).sum()), int(worse.sum()), int(np.median((s64["score"] - s32["score"])[worse])) if worse.any() else 0,
               int((s64["score"] - s32["score"])[worse].max()) if worse.any() else 0, int((both & (s32["score"] > s64["score"])).sum()),
               int(((s64["q_end"] - s64["q_start"]) - (s32["q_end"] - s32["q_start"]))[both].clip(0).sum()), int(np.array([len(x) for x in raw]).sum())))


if __name__ == "__main__":
    main()
