#!/usr/bin/env python3
"""K6 accuracy against the simulator's true haplotypes, K1 -> K6 from raw reads on one GPU (parity of K6 is unpinned: this is its evidence).
  1. iid CLR errors at three coverages (fzcns v3, the default)
  2. hets that are 2-5 base indels, iid and homopolymer-biased errors: v1 / v2 / v3 side by side
  3. a homopolymer table: for every run of 1..8+ equal bases of the true haplotype inside a tig, is the run (with 8 bases of context on
     both sides) spelled exactly in the consensus?  -- iid and homopolymer-biased errors (indels 3x as likely inside runs)
usage (GPU box): python3 tools/cns_accuracy.py > profiles/r3_cns_accuracy.txt"""
import sys

import numpy as np

sys.path.insert(0, '.')
from falcon_unzip_amd import _lib, sim            # noqa: E402
from tests import cns_util                        # noqa: E402

eng = _lib.Engine(0)


def tig_errors(t, truths, maps=None):
    tot = err = 0
    for i, tig in enumerate(t.tigs):
        lo, hi = int(tig["lo"]), int(tig["hi"])
        s = t.sequence(i)
        ds = []
        for k, tr in enumerate(truths):
            a, b = (lo, hi + 1) if maps is None or maps[k] is None else (int(maps[k][lo]), int(maps[k][hi]) + 1)
            ds.append(cns_util.banded_edit_distance(s, tr[a:b]))
        tot += hi - lo + 1
        err += min(ds)
    return tot, err


def hp_table(t, truths, maps):
    """per run length: runs of the closer haplotype found verbatim (with 8 bases of context) / runs looked at"""
    ok, seen = np.zeros(10, np.int64), np.zeros(10, np.int64)
    for i, tig in enumerate(t.tigs):
        lo, hi = int(tig["lo"]), int(tig["hi"])
        s = t.sequence(i)
        best = None
        for k, tr in enumerate(truths):
            a, b = (lo, hi + 1) if maps[k] is None else (int(maps[k][lo]), int(maps[k][hi]) + 1)
            d = cns_util.banded_edit_distance(s, tr[a:b])
            if best is None or d < best[0]:
                best = (d, tr[a:b])
        tr = np.frombuffer(best[1], np.uint8)
        brk = np.flatnonzero(tr[1:] != tr[:-1]) + 1
        starts = np.concatenate(([0], brk))
        ends = np.concatenate((brk, [len(tr)]))
        for a, b in zip(starts, ends):
            if a < 8 or b + 8 > len(tr) or (a // 97) % 4:          # a quarter of the runs is plenty
                continue
            n = min(int(b - a), 9)
            seen[n] += 1
            ok[n] += s.find(best[1][a - 8:b + 8]) >= 0
    return ok, seen


print("== 1. iid CLR errors (sub 1 % / ins 8 % / del 4 %), SNP hets, 120 kb diploid contig, fzcns v3")
rng = np.random.Generator(np.random.PCG64(77))
L = 120000
hap0, hap1, het = sim.make_diploid(L, rng, het_rate=1.0 / 400)
for cov_reads in (400, 800, 1600):
    reads = sim.simulate_reads(hap0, hap1, cov_reads, 9000, rng, strand_mix=0.5)
    ctg = sim.codes_to_str(hap0).encode()
    raw = [sim.codes_to_str(r.raw_seq_codes()).encode() for r in reads]
    job = _lib.align_job(eng, [ctg], raw); job.run()
    b = job.to_batch(); b.run(_lib.STAGE_ALL); t = b.consensus()
    tot, err = tig_errors(t, [ctg, sim.codes_to_str(hap1).encode()])
    print("reads", cov_reads, "cov/hap ~%.0f" % (cov_reads * 9000 / L / 2), "tigs", len(t.tigs), "bases", tot, "errors", err, "identity %.4f%%" % (100 * (1 - err / max(1, tot))))
    t.close(); b.close(); job.close()

print("== 2. hets incl. 2-5 base insertions / deletions, 80 kb contig, ~30x per haplotype: errors of v1 / v2 / v3")
tables = {}
for label, hp_bias, seed in (("iid errors", 1.0, 81), ("iid errors, another genome", 1.0, 83), ("homopolymer-biased errors (3x)", 3.0, 82), ("homopolymer-biased errors (3x), another genome", 3.0, 84)):
    rng = np.random.Generator(np.random.PCG64(seed))
    hap0, hap1, map01, events = sim.make_diploid_indels(80000, rng)
    raw = [r[1] for r in sim.simulate_raw_reads_from(hap0, 270, 9000, rng, hp_bias=hp_bias, name_prefix="a")]
    raw += [r[1] for r in sim.simulate_raw_reads_from(hap1, 270, 9000, rng, hp_bias=hp_bias, name_prefix="b")]
    ctg = sim.codes_to_str(hap0).encode()
    truths, maps = [ctg, sim.codes_to_str(hap1).encode()], [None, map01]
    job = _lib.align_job(eng, [ctg], raw); job.run()
    b = job.to_batch(); b.run(_lib.STAGE_ALL)
    row = []
    for ver in (1, 2, 3):
        t = b.consensus(version=ver)
        tot, err = tig_errors(t, truths, maps)
        row.append(err)
        if ver == 3:
            tables[label] = hp_table(t, truths, maps)
        t.close()
    print("%-48s bases %d  errors v1 %d  v2 %d  v3 %d  (v3 identity %.4f%%)  het events: %s" % (label, tot, row[0], row[1], row[2], 100 * (1 - row[2] / max(1, tot)),
          {k: sum(1 for e in events if e[1] == k) for k in ("snp", "ins", "del")}))
    b.close(); job.close()

print("== 3. homopolymer table (fzcns v3): runs of the true haplotype spelled exactly, by run length")
for label, (ok, seen) in tables.items():
    print(label)
    print("   run length   " + "  ".join("%6s" % (str(n) if n < 9 else "9+") for n in range(1, 10)))
    print("   runs seen    " + "  ".join("%6d" % seen[n] for n in range(1, 10)))
    print("   exact        " + "  ".join("%6s" % ("%.1f%%" % (100.0 * ok[n] / seen[n]) if seen[n] else "-") for n in range(1, 10)))
eng.close()
