"""measurement aid: does the blocking-sync scheduling flag hold when torch brought the device up first (backend nccl does)?  Prints what hipSetDeviceFlags answered and the CPU time of the process against the wall time of a few alignment runs (a spinning launch thread shows as cpu ~ wall)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import bench
contigs, blob, off, rc = bench.make_inputs(2, list(range(6)), 2_000_000, lambda ci: 1500, 15000, 750_000, 1)
from falcon_unzip_amd import _lib
mode = "torch_first"      # (the other order -- HIP brought up through libfzphase BEFORE `import torch` -- leaves torch with "No HIP GPUs are available": measured, not offered)
import torch
torch.cuda.set_device(0)
x = torch.zeros(1024, device="cuda:0"); torch.cuda.synchronize()
eng = _lib.Engine(0)
job = _lib.align_job_raw(eng, contigs, blob, off, rc)
job.run(); eng.synchronize()
t0, c0 = time.perf_counter(), time.process_time()
for _ in range(10):
    job.run()
eng.synchronize()
print(mode, "hipSetDeviceFlags rc", _lib.sched_status(), "wall %.1f ms cpu %.1f ms per run" % ((time.perf_counter() - t0) * 100, (time.process_time() - c0) * 100))
