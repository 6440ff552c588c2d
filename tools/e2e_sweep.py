#!/usr/bin/env python3
"""End-to-end (host buffers in, files out) rate of fzp_phase_contigs on the bench's cfg2 workload for a few (contigs per group, lanes)
pairs.  Prints one JSON line per pair.  Usage: python3 tools/e2e_sweep.py [out.json]"""
import json
import os
import sys
import tempfile
import time

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
import bench  # noqa: E402


def main():
    n_ctg, L, R, RL, win = 20, 5_000_000, 2000, 15000, 750_000
    mine = list(range(n_ctg))
    contigs, blob, off, read_ctg = bench.make_inputs(2, mine, L, lambda ci: R, RL, win, max(1, min(8, os.cpu_count() or 1)))
    ids = ["%06dF" % ci for ci in mine]
    name_tab, maps = bench.make_names_and_maps(read_ctg, off, ids, 0)
    if os.environ.get("SWEEP_TORCH_FIRST"):
        import torch  # noqa: F401  (which HIP runtime the library binds to depends on what the process loaded first)
        torch.cuda.device_count()
    from falcon_unzip_amd import _lib
    eng = _lib.Engine(0)
    out_root = tempfile.mkdtemp(prefix="fzp_e2e_", dir=os.environ.get("TMPDIR", "/tmp"))
    rows = []
    for gc, lanes in ((20, 1), (10, 2)):
        gb = int(gc * R * RL * 1.06)
        ts = []
        for k in range(4):
            t0 = time.perf_counter()
            st, _ = _lib.phase_contigs(eng, contigs, blob, off, read_ctg, ids, names=name_tab, out_dir=os.path.join(out_root, "g%d_l%d_%d" % (gc, lanes, k)),
                                       read_maps=maps, ctg_index=mine, n_lanes=lanes, group_bases=gb, async_writes=True)
            ts.append(time.perf_counter() - t0)
        row = {"group_contigs": gc, "lanes": lanes, "groups": int(st["n_groups"]), "ms_best": round(min(ts[1:]) * 1e3, 2), "ms_all": [round(t * 1e3, 1) for t in ts],
               "reads_per_s": round(len(read_ctg) / min(ts[1:]), 1), "sections": {k: round(st[k], 1) for k in ("ms_upload", "ms_k1", "ms_phase", "ms_results", "ms_text")}}
        print(json.dumps(row), flush=True)
        rows.append(row)
    with open("/proc/self/maps") as f:
        print(sorted({ln.split()[-1] for ln in f if "libamdhip64" in ln or "libhsa-runtime" in ln}), flush=True)
    if len(sys.argv) > 1:
        with open(sys.argv[1], "w") as f:
            json.dump(rows, f, indent=1)


if __name__ == "__main__":
    main()
