"""bench.py's N > 1 flow on a one-GPU box: `python bench.py --gpus 2` starts its own two ranks (children under
torch.distributed.run), which share GPU 0 (gloo for the exchange step; on a multi-GPU node the same code path runs over RCCL).
Checks the JSON contract, that both ranks' work is in `value`, and the strong_cfg3 leg (BASELINE configs[2]) of an N > 1 run."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SMALL = ["--contigs", "2", "--contig-len", "300000", "--reads-per-contig", "150", "--read-len", "8000", "--window", "120000", "--steps", "2", "--warmup", "1",
         "--no-cpu-baseline", "--gen-workers", "1", "--strong-leg-contigs", "5", "--strong-leg-contig-len", "200000"]


def run_bench(nproc, launcher=False):
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK")}
    env.update(FZP_BENCH_BACKEND="gloo", MASTER_ADDR="127.0.0.1")
    if not launcher:
        cmd = [sys.executable, os.path.join(REPO, "bench.py"), "--gpus", str(nproc)] + SMALL        # N > 1: bench.py starts its own ranks
    else:
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(nproc), "--master-addr", "127.0.0.1", "--master-port", "29741",
               os.path.join(REPO, "bench.py"), "--gpus", str(nproc)] + SMALL
    out = subprocess.check_output(cmd, env=env, cwd=REPO, stderr=subprocess.DEVNULL, timeout=600).decode()
    lines = [l for l in out.splitlines() if l.startswith("{")]
    assert len(lines) == 1, out
    return json.loads(lines[0])


def test_two_ranks_share_one_gpu():
    one = run_bench(1)
    two = run_bench(2)
    for d, n in ((one, 1), (two, 2)):
        assert d["n_gpus"] == n and d["steps"] == 2 and d["warmup"] == 1 and d["scaling"] == "weak" and d["higher_is_better"] is True
        assert d["unit"] == "reads/s" and d["vs_baseline"] is None and d["data"] == "synthetic" and "workload" in d["config"]
        rl = d["roofline"]
        assert rl["bound"] == "valu" and rl["launches"] >= 2 and rl["peak"] == 1228.8 and 0 < rl["frac"] < 1 and rl["unit"] == "G wave64-inst/s"
        assert rl["hbm_algorithmic"]["peak"] == 8000.0 and 0 < rl["hbm_algorithmic"]["frac"] < 1 and rl["hbm_moved"]["peak"] == 8000.0
        assert d["index_in_step"] is True and d["index_ms"] > 0 and d["ms_per_step_instrumented"] > 0 and d["kernel_ms_per_step"]["k1_sw"] > 0
        if n == 2:
            assert "error" not in d["from_files"] and d["from_files"]["n_gpus"] == 2 and d["value_from_files"] == d["from_files"]["reads_per_s"] > 0
        assert d["aligned_frac"] > 0.98
    assert one["stage_counts"]["r2p_records"] == 300 and two["stage_counts"]["r2p_records"] == 600      # the all-gather saw both shards
    assert two["config"]["reads_per_gpu"] == one["config"]["reads_per_gpu"] == 300 and two["config"]["reads_total"] == 600
    assert one["stage_counts"]["bytes_written"] > 10000 and one["end_to_end"]["reads_per_s"] > 0 and two["end_to_end"] is None
    assert len(two["rank_load"]) == 2
    assert one["value_end_to_end"] == one["end_to_end"]["reads_per_s"] and "strong_cfg3" not in one
    # gloo dry run: no RCCL communicator, and the line says so
    assert two["rccl_ranks"] == 0 and "torch.distributed" in two["gather"]
    sc = two["strong_cfg3"]
    assert "error" not in sc, sc
    assert sc["scaling"] == "strong" and sc["n_gpus"] == 2 and sc["reads_total"] == sc["r2p_records"] == sum(r["reads"] for r in sc["rank_load"]) > 0
    assert sum(r["contigs"] for r in sc["rank_load"]) == 5 and min(r["reads"] for r in sc["rank_load"]) > 0


def test_launcher_started_ranks_and_gpus_flag_must_agree():
    """the driver's form (torch.distributed.run around bench.py) gives the same line; a --gpus that contradicts WORLD_SIZE is refused"""
    two = run_bench(2, launcher=True)
    assert two["n_gpus"] == 2 and two["stage_counts"]["r2p_records"] == 600
    env = dict(os.environ, FZP_BENCH_BACKEND="gloo", WORLD_SIZE="2", RANK="0", LOCAL_RANK="0", MASTER_ADDR="127.0.0.1", MASTER_PORT="29749")
    p = subprocess.run([sys.executable, os.path.join(REPO, "bench.py"), "--gpus", "1"] + SMALL, env=env, cwd=REPO, capture_output=True, timeout=120)
    assert p.returncode != 0 and b"must agree" in p.stderr


def test_strong_scaling_mode_two_ranks():
    """--strong: a fixed set of uneven contigs dealt LPT over the ranks (BASELINE configs[2] shape); value = all reads / slowest rank"""
    import subprocess
    env = dict(os.environ, FZP_BENCH_BACKEND="gloo", MASTER_ADDR="127.0.0.1")
    args = ["--strong", "--contigs", "5", "--contig-len", "300000", "--reads-per-contig", "100", "--read-len", "8000", "--window", "120000", "--steps", "1", "--warmup", "1",
            "--no-cpu-baseline", "--gen-workers", "1"]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1", "--master-port", "29743",
           os.path.join(REPO, "bench.py"), "--gpus", "2"] + args
    out = subprocess.check_output(cmd, env=env, cwd=REPO, stderr=subprocess.DEVNULL, timeout=600).decode()
    d = json.loads([l for l in out.splitlines() if l.startswith("{")][0])
    assert d["scaling"] == "strong" and d["n_gpus"] == 2
    loads = [r["reads"] for r in d["rank_load"]]
    assert sum(loads) == d["config"]["reads_total"] == d["stage_counts"]["r2p_records"] and min(loads) > 0
    assert abs(loads[0] - loads[1]) <= 0.4 * sum(loads)


def test_strong_mode_with_an_empty_shard():
    """fewer contigs than ranks: the rank without a contig has no job but still takes part in every collective"""
    import subprocess
    env = dict(os.environ, FZP_BENCH_BACKEND="gloo", MASTER_ADDR="127.0.0.1")
    args = ["--strong", "--contigs", "1", "--contig-len", "300000", "--reads-per-contig", "100", "--read-len", "8000", "--window", "120000", "--steps", "1", "--warmup", "1",
            "--no-cpu-baseline", "--gen-workers", "1"]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1", "--master-port", "29747",
           os.path.join(REPO, "bench.py"), "--gpus", "2"] + args
    out = subprocess.check_output(cmd, env=env, cwd=REPO, stderr=subprocess.DEVNULL, timeout=600).decode()
    d = json.loads([l for l in out.splitlines() if l.startswith("{")][0])
    loads = sorted(r["reads"] for r in d["rank_load"])
    assert loads[0] == 0 and loads[1] == d["config"]["reads_total"] == d["stage_counts"]["r2p_records"] > 0
