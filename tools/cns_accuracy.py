import numpy as np, sys
sys.path.insert(0, '.')
from falcon_unzip_amd import _lib, sim
from tests import cns_util
eng = _lib.Engine(0)
rng = np.random.Generator(np.random.PCG64(77))
L = 120000
hap0, hap1, het = sim.make_diploid(L, rng, het_rate=1.0 / 400)
for cov_reads in (400, 800, 1600):
    reads = sim.simulate_reads(hap0, hap1, cov_reads, 9000, rng, strand_mix=0.5)
    ctg = sim.codes_to_str(hap0).encode()
    raw = [sim.codes_to_str(r.raw_seq_codes()).encode() for r in reads]
    job = _lib.align_job(eng, [ctg], raw); job.run()
    b = job.to_batch(); b.run(_lib.STAGE_ALL); t = b.consensus()
    truth = [sim.codes_to_str(hap0).encode(), sim.codes_to_str(hap1).encode()]
    tot = err = 0
    for i, tig in enumerate(t.tigs):
        lo, hi = int(tig["lo"]), int(tig["hi"])
        d = min(cns_util.banded_edit_distance(t.sequence(i), tr[lo:hi + 1]) for tr in truth)
        tot += hi - lo + 1; err += d
    print("reads", cov_reads, "cov/hap ~%.0f" % (cov_reads * 9000 / L / 2), "tigs", len(t.tigs), "bases", tot, "errors", err, "identity %.4f%%" % (100 * (1 - err / max(1, tot))))
    t.close(); b.close(); job.close()
