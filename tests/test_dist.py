"""Multi-rank plumbing on CPU (gloo, world size 2): contig sharding, the rid_to_phase all-gather and
the rid_to_phase.all ordering (reference: falcon_unzip/unzip.py:283-288,303-314)."""
import os
import socket
import subprocess
import sys

import numpy as np
import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_lpt_sharding_is_balanced_and_deterministic():
    from falcon_unzip_amd.dist import shard_contigs
    w = [50, 10, 10, 10, 10, 10, 30, 20]
    s = shard_contigs(w, 3)
    assert sorted(sum(s, [])) == list(range(len(w)))
    loads = [sum(w[i] for i in part) for part in s]
    assert max(loads) - min(loads) <= max(w) * 0.21
    assert s == shard_contigs(w, 3)
    assert shard_contigs([5, 5], 4) == [[0], [1], [], []]


def test_r2p_from_preads_last_block_wins():
    from falcon_unzip_amd import _lib
    from falcon_unzip_amd.dist import r2p_from_preads
    pr = np.zeros(4, _lib.PREAD)
    pr["q_id"] = [0, 2, 2, 3]
    pr["block"] = [1, 1, 2, 5]
    pr["phase"] = [0, 1, 0, 1]
    out = r2p_from_preads(pr, 5, 100, 7)
    assert list(out["arid"]) == [100, 101, 102, 103, 104]
    assert list(out["block"]) == [1, -1, 2, 5, -1]      # phasing_readmap.py:30-33,46
    assert list(out["phase"]) == [0, 0, 0, 1, 0]
    assert set(out["ctg"]) == {7}


WORKER = r'''
import os, sys
sys.path.insert(0, %(repo)r)
import numpy as np
import torch.distributed as dist
from falcon_unzip_amd import _lib
from falcon_unzip_amd import dist as fdist
rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
dist.init_process_group(backend="gloo")
# 5 contigs with unequal weights, dealt LPT; each rank fabricates records for its contigs only
weights = [7, 3, 9, 2, 5]
mine = fdist.shard_contigs(weights, world)[rank]
recs = []
for c in mine:
    r = np.zeros(weights[c], _lib.R2P)
    r["arid"] = 1000 * c + np.arange(weights[c])[::-1]      # deliberately unsorted
    r["ctg"] = c
    r["block"] = np.where(np.arange(weights[c]) %% 3 == 0, -1, c + 1)
    r["phase"] = np.arange(weights[c]) %% 2
    recs.append(r)
local = np.concatenate(recs) if recs else np.zeros(0, _lib.R2P)
allr = fdist.allgather_r2p(local, device="cpu")
np.save(os.path.join(%(out)r, "all_%%d.npy" %% rank), allr)
dist.barrier()
dist.destroy_process_group()
'''


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def test_allgather_two_ranks_gloo(tmp_path):
    from falcon_unzip_amd import _lib
    from falcon_unzip_amd.dist import format_rid_to_phase_all
    script = tmp_path / "worker.py"
    script.write_text(WORKER % {"repo": REPO, "out": str(tmp_path)})
    port = _free_port()
    procs = []
    for rank in range(2):
        env = dict(os.environ, RANK=str(rank), WORLD_SIZE="2", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        procs.append(subprocess.Popen([sys.executable, str(script)], env=env))
    for p in procs:
        assert p.wait(timeout=300) == 0
    a0 = np.load(tmp_path / "all_0.npy")
    a1 = np.load(tmp_path / "all_1.npy")
    assert a0.dtype == _lib.R2P and np.array_equal(a0, a1)          # every rank holds the global map
    assert len(a0) == 7 + 3 + 9 + 2 + 5
    key = a0["ctg"].astype(np.int64) * 10**6 + a0["arid"]
    assert np.all(np.diff(key) > 0)                                   # (contig, arid) order == sorted per-contig files
    # single-process result is identical
    weights = [7, 3, 9, 2, 5]
    exp = []
    for c, w in enumerate(weights):
        r = np.zeros(w, _lib.R2P)
        r["arid"] = 1000 * c + np.arange(w)
        r["ctg"] = c
        k = np.arange(w)[::-1]
        r["block"] = np.where(k % 3 == 0, -1, c + 1)
        r["phase"] = k % 2
        exp.append(r)
    assert np.array_equal(a0, np.concatenate(exp))
    text = format_rid_to_phase_all(a0[:3], ["c%d" % i for i in range(5)])
    assert text.split(b"\n")[0] == b"%09d c0 %d %d" % (0, a0[0]["block"], a0[0]["phase"])


def test_allgather_without_process_group():
    from falcon_unzip_amd import _lib
    from falcon_unzip_amd.dist import allgather_r2p
    r = np.zeros(3, _lib.R2P)
    r["arid"] = [5, 1, 3]
    out = allgather_r2p(r)
    assert list(out["arid"]) == [1, 3, 5]


def test_gather_looks_before_it_sorts():
    """r6: records that arrive in rid_to_phase.all's order (contig, then pread id) come back as they are -- the same array, nothing sorted --, any other order is sorted"""
    from falcon_unzip_amd import _lib
    from falcon_unzip_amd.dist import allgather_r2p
    rng = np.random.default_rng(3)
    r = np.zeros(5000, _lib.R2P)
    r["ctg"] = np.sort(rng.integers(0, 7, len(r)))
    for c in range(7):      # ascending ids inside a contig, with gaps and one repeat
        m = r["ctg"] == c
        ids = np.sort(rng.choice(20000, int(m.sum()), replace=False))
        if len(ids) > 2:
            ids[1] = ids[0]
        r["arid"][m] = ids
    r["block"] = rng.integers(-1, 9, len(r))
    out = allgather_r2p(r)
    assert out is r or np.shares_memory(out, r) or np.array_equal(out, r)
    assert np.array_equal(out, r)
    sh = r[rng.permutation(len(r))]
    out2 = allgather_r2p(sh)
    assert np.array_equal(out2["ctg"], r["ctg"]) and np.array_equal(out2["arid"], r["arid"])
    assert len(allgather_r2p(np.zeros(0, _lib.R2P))) == 0 and len(allgather_r2p(r[:1])) == 1


def test_r2p_from_batch_equals_per_contig():
    from falcon_unzip_amd import _lib
    from falcon_unzip_amd import dist as fdist
    rng = np.random.default_rng(5)
    reads_per_ctg = [7, 0, 12, 5]
    parts, begins, exp = [], [0], []
    for c, nq in enumerate(reads_per_ctg):
        rows = []
        for q in range(nq):
            for blk in sorted(rng.choice(np.arange(1, 6), size=int(rng.integers(0, 3)), replace=False)):
                rows.append((q, int(blk), int(rng.integers(0, 2)), 3, 1))
        pr = np.array(rows, dtype=_lib.PREAD) if rows else np.zeros(0, _lib.PREAD)
        parts.append(pr)
        begins.append(begins[-1] + len(pr))
        exp.append(fdist.r2p_from_preads(pr, nq, 100 + sum(reads_per_ctg[:c]), 40 + c))
    got = fdist.r2p_from_batch(np.concatenate(parts), begins, reads_per_ctg, 100, 40)
    assert np.array_equal(got, np.concatenate(exp))
    assert len(fdist.r2p_from_batch(np.zeros(0, _lib.PREAD), [0, 0], [0], 0, 0)) == 0
