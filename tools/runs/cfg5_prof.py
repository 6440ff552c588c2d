"""rocprofv3 --kernel-trace --stats -- python3 tools/runs/cfg5_prof.py [ENV=VAL ...]: configs[4] at half scale, one GPU, env switches applied in-process"""
import os
import sys
for a in sys.argv[1:]:
    k, v = a.split("=", 1)
    os.environ[k] = v
sys.path.insert(0, "tools"); sys.path.insert(0, ".")
import run_cfg5
r = run_cfg5.run(scale=0.5, lanes=2, workers=1, consensus=False)
print(r["wall_s"], r["first_call_wall_s"], {k: r["stats"][k] for k in ("ms_upload", "ms_k1", "ms_phase", "ms_results", "ms_text", "n_groups")})
