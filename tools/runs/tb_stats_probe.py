import os, sys
import numpy as np
sys.path.insert(0, ".")
from falcon_unzip_amd import _lib, sim
rng = np.random.Generator(np.random.PCG64(49))
hap0, hap1, _ = sim.make_diploid(1_000_000, rng)
codes, off, *_ = sim.simulate_raw_reads_shaped(hap0, hap1, 500, rng)
ctg = sim.ACGT[hap0].tobytes(); blob = sim.ACGT[codes].tobytes()
eng = _lib.Engine(0)
for env in ({}, {"FZP_TB_GUESS_LANE": "1"}, {"FZP_TB_SERIAL": "1"}):
    os.environ.update(env)
    job = _lib.align_job_raw(eng, [ctg], blob, off, np.zeros(500, np.int32))
    job.run()
    s = job.summaries()
    print(env, job.tb_fallbacks(), int(s["aligned"].sum()), int((np.diff(off) > 18000).sum()), int(s["cells"].sum() // 64))
    job.close()
    for k in env: os.environ.pop(k)
