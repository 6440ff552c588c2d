#!/usr/bin/env python3
"""configs[4] (tools/run_cfg5.py's job, generated once) under several k_swb schedules: FZP_SWB_WAVES / FZP_SWB_UNIT / FZP_SWB_HYST / FZP_SWB_DBG per run."""
import json
import os
import shutil
import sys
import tempfile
import time

REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, REPO)
sys.path.insert(0, os.path.join(REPO, "tools"))
import run_cfg5

def main():
    scale = float(sys.argv[1]) if len(sys.argv) > 1 else 1.0
    variants = sys.argv[2].split() if len(sys.argv) > 2 else ["1:1048576:0:0", "2:16:0:4"]
    contigs, blob, off, read_ctg, ids = run_cfg5.make_job(int(120e6 * scale), workers=8)
    print("generated", flush=True)
    from bench import make_names_and_maps      # (bench imports torch: not before the generator's worker processes are forked)
    from falcon_unzip_amd import _lib
    name_tab, maps = make_names_and_maps(read_ctg, off, ids, 0)
    eng = _lib.Engine(0)
    out_dir = tempfile.mkdtemp(prefix="fzp_cfg5ab_")
    for rep in range(2):
        for v in variants:
            w, u, h, d = (v.split(":") + ["0", "0"])[:4]
            os.environ.update(FZP_SWB_WAVES=w, FZP_SWB_UNIT=u, FZP_SWB_HYST=h, FZP_SWB_DBG=d)
            walls = []
            for k in range(3):
                t1 = time.perf_counter()
                stats, recs = _lib.phase_contigs(eng, contigs, blob, off, read_ctg, ids, names=name_tab, out_dir=os.path.join(out_dir, "run"), read_maps=maps, n_lanes=2, consensus=True)
                walls.append(time.perf_counter() - t1)
                shutil.rmtree(os.path.join(out_dir, "run"), ignore_errors=True)
            print(json.dumps({"variant": v, "rep": rep, "walls_s": [round(x, 4) for x in walls], "reads": int(len(read_ctg))}), flush=True)
    eng.close()
    shutil.rmtree(out_dir, ignore_errors=True)


if __name__ == "__main__":
    main()
