#!/usr/bin/env python3
"""Run tools/ubench/libh2d.so under the system HIP runtime (default) or under the one torch ships (H2D_TORCH_FIRST=1)."""
import ctypes
import os
import sys
if os.environ.get("H2D_TORCH_FIRST"):
    import torch
    torch.cuda.device_count()
lib = ctypes.CDLL(os.path.join(os.path.dirname(os.path.abspath(__file__)), "libh2d.so"))
sys.stdout.flush()
rc = lib.h2d_run()
with open("/proc/self/maps") as f:
    print(sorted({ln.split()[-1] for ln in f if "libamdhip64" in ln}))
sys.exit(rc)
