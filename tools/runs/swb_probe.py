#!/usr/bin/env python3
"""timing probes of k_swb: FZP_SWB_DBG bit 0 = no mask stores, bit 1 = no stream refills (the masks of a normal run stay in the buffers, so the
trace-back of a probe run walks valid masks; its alignments are not used).  usage (GPU box): timeout 300 python3 tools/runs/swb_probe.py"""
import json
import os
import sys

REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, REPO)
import bench  # noqa: E402


def main():
    contigs, blob, off, read_ctg = bench.make_inputs(2, list(range(20)), 5_000_000, lambda ci: 2000, 15000, 750_000, 8)
    from falcon_unzip_amd import _lib
    eng = _lib.Engine(0)
    job = _lib.align_job_raw(eng, contigs, blob, off, read_ctg)
    out = {}
    for mode in sys.argv[1:] or ["0", "1", "2", "3", "0"]:
        os.environ.pop("FZP_SWB_DBG", None)
        os.environ.pop("FZP_SW_NO_BITS", None)
        os.environ.pop("FZP_SWB_INPUT_ORDER", None)
        job.run()
        eng.synchronize()
        os.environ.pop("FZP_SWB_INPUT_ORDER", None)
        if mode == "nobits":
            os.environ["FZP_SW_NO_BITS"] = "1"
        elif mode == "inorder":
            os.environ["FZP_SWB_INPUT_ORDER"] = "1"
        else:
            os.environ["FZP_SWB_DBG"] = mode
        eng.prof_reset()
        eng.prof_enable(True)
        for _ in range(3):
            job.run()
        eng.synchronize()
        eng.prof_enable(False)
        pr = eng.prof()
        out["dbg_" + mode] = {k: round(pr[k][0] / max(1, pr[k][1]), 3) for k in ("k1_sw", "k1_traceback") if k in pr}
        print(mode, out["dbg_" + mode], flush=True)
    os.environ.pop("FZP_SWB_DBG", None)
    job.close()
    eng.close()
    print(json.dumps(out))


if __name__ == "__main__":
    main()
