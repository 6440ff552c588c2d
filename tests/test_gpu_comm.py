"""The C-ABI exchange step (fzp_comm_* / fzp_allgather_rid_to_phase, csrc/fzp_comm.hip): RCCL all-gather of rid_to_phase records.
World size 1 in-process; world size 2 as two processes that share the box's one GPU (RCCL may refuse two ranks on one device:
then that leg is skipped -- the N-GPU run is the driver's)."""
import os
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _records(rank, n):
    from falcon_unzip_amd import _lib
    rng = np.random.Generator(np.random.PCG64(500 + rank))
    r = np.zeros(n, _lib.R2P)
    r["arid"] = rng.permutation(n) + 1000 * rank
    r["ctg"] = rng.integers(0, 5, n) * 2 + rank          # ranks own interleaved contigs
    r["block"] = rng.integers(-1, 9, n)
    r["phase"] = rng.integers(0, 2, n)
    return r


def test_world_one():
    from falcon_unzip_amd import _lib, dist
    eng = _lib.Engine(0)
    comm = _lib.Comm(eng, 0, 1, _lib.comm_unique_id())
    for n in (0, 1, 777):
        loc = _records(0, n)
        got = comm.allgather_r2p(loc)
        assert np.array_equal(got, dist.allgather_r2p(loc))
    comm.close()
    eng.close()


def test_the_line_says_which_rccl_and_a_missing_peer_is_a_message_not_a_hang(monkeypatch):
    """What a run nobody watches needs: the file and version of the RCCL the library bound (a process that imported torch gets torch's copy under the same soname), and
    -- FZP_COMM_TIMEOUT_S -- a communicator whose peer never arrives ends as an error that says so."""
    from falcon_unzip_amd import _lib
    path, ver = _lib.comm_library()
    assert os.path.isabs(path) and "rccl" in os.path.basename(path).lower() and ver > 20000, (path, ver)
    monkeypatch.setenv("FZP_COMM_TIMEOUT_S", "3")
    eng = _lib.Engine(0)
    import time
    t0 = time.time()
    with pytest.raises(_lib.FzpError) as e:
        _lib.Comm(eng, 0, 2, _lib.comm_unique_id())           # rank 1 of 2 never comes
    assert "did not return within 3 s" in str(e.value) and "a peer never arrived" in str(e.value) and time.time() - t0 < 30
    eng.close()


CHILD = r"""
import sys, os, time, numpy as np
sys.path.insert(0, %r)
from falcon_unzip_amd import _lib
from tests.test_gpu_comm import _records
rank, path = int(sys.argv[1]), sys.argv[2]
eng = _lib.Engine(0)
if rank == 0:
    uid = _lib.comm_unique_id()
    with open(path + ".tmp", "wb") as f: f.write(uid)
    os.rename(path + ".tmp", path)
else:
    for _ in range(600):
        if os.path.exists(path): break
        time.sleep(0.05)
    uid = open(path, "rb").read()
comm = _lib.Comm(eng, rank, 2, uid)
got = comm.allgather_r2p(_records(rank, 300 + 211 * rank))
np.save(path + ".out%%d.npy" %% rank, got)
comm.close(); eng.close()
"""


def test_world_two_on_one_gpu(tmp_path):
    from falcon_unzip_amd import _lib
    path = str(tmp_path / "uid")
    env = dict(os.environ, PYTHONPATH=REPO, HSA_ENABLE_IPC_MODE_LEGACY="0")
    procs = [subprocess.Popen([sys.executable, "-c", CHILD % REPO, str(r), path], env=env, cwd=REPO, stdout=subprocess.PIPE, stderr=subprocess.STDOUT) for r in (0, 1)]
    outs = []
    for p in procs:
        try:
            o, _ = p.communicate(timeout=180)
        except subprocess.TimeoutExpired:
            for q in procs:
                q.kill()
            pytest.skip("RCCL did not complete a two-rank communicator on one device within 180 s")
        outs.append(o.decode(errors="replace"))
    if any(p.returncode != 0 for p in procs):
        text = "\n".join(outs)
        if "uplicate" in text or "invalid usage" in text.lower() or "ncclCommInitRank" in text:
            pytest.skip("RCCL refuses two ranks on one device here: " + text[-300:])
        raise AssertionError(text[-2000:])
    exp = np.concatenate([_records(0, 300), _records(1, 511)])
    exp = exp[np.lexsort((exp["arid"], exp["ctg"]))]
    for r in (0, 1):
        assert np.array_equal(np.load(path + ".out%d.npy" % r), exp)
