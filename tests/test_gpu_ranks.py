"""The product's multi-rank driver and its input path: `scripts/fc_unzip_phase_gpu.py` under torch.distributed.run with two ranks
(gloo on a one-GPU box; on a multi-GPU node the same code runs the library's RCCL all-gather) must leave the very bytes the one-rank run leaves; the
library's own FASTA reader (fzp_phase_contigs_files) the very bytes the in-memory entry point (fzp_phase_contigs) writes; and BASELINE configs[2] / configs[4]
at their shapes on one GPU."""
import ctypes as C
import hashlib
import json
import os
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _tree_digest(root, skip=()):
    out = {}
    for d, _, files in sorted(os.walk(root)):
        for f in sorted(files):
            rel = os.path.relpath(os.path.join(d, f), root)
            if any(rel.startswith(s) for s in skip):
                continue
            with open(os.path.join(d, f), "rb") as fh:
                out[rel] = hashlib.sha256(fh.read()).hexdigest()
    return out


def _make_unzip_tree(root, n_ctg=5, seed=700):
    """a miniature 3-unzip/reads + read_maps: contigs of uneven size (so that LPT has something to do), reads named as the reference names them"""
    from falcon_unzip_amd import sim
    unzip = os.path.join(root, "3-unzip")
    os.makedirs(os.path.join(unzip, "reads"))
    ctgs = ["%06dF" % (7 * k % n_ctg) for k in range(n_ctg)]              # deliberately unsorted in ctg_list
    raw_names, p2c = [], []
    rid = 0
    for ctg in sorted(ctgs):
        k = int(ctg[:6])
        rng = np.random.Generator(np.random.PCG64(seed + k))
        L = 30000 + 9000 * k
        hap0, hap1, _ = sim.make_diploid(L, rng, het_rate=1.0 / 300)
        reads = sim.simulate_reads(hap0, hap1, 120 + 25 * k, 6000, rng, strand_mix=0.5, name_prefix="m%d" % k)
        sim.write_fasta(os.path.join(unzip, "reads", "%s_ref.fa" % ctg), [(ctg + " primary", sim.codes_to_str(hap0))], width=70)
        recs = []
        for r in reads:
            nm = "m%d/%d/0_%d" % (k, rid, len(r.raw_seq_codes()))
            recs.append((nm, sim.codes_to_str(r.raw_seq_codes())))
            raw_names.append(nm)
            p2c.append("%09d %s 6000 0 6000 1" % (rid, ctg))
            rid += 1
        sim.write_fasta(os.path.join(unzip, "reads", "%s_reads.fa" % ctg), recs)
    with open(os.path.join(unzip, "reads", "ctg_list"), "w") as f:
        f.write("".join(c + "\n" for c in ctgs))
    rmd = os.path.join(root, "read_maps")
    os.makedirs(os.path.join(rmd, "dump_rawread_ids"))
    os.makedirs(os.path.join(rmd, "dump_pread_ids"))
    with open(os.path.join(rmd, "dump_rawread_ids", "rawread_ids"), "w") as f:
        f.write("".join(n + "\n" for n in raw_names))
    with open(os.path.join(rmd, "dump_pread_ids", "pread_ids"), "w") as f:
        f.write("".join("pread/%d/0_6000\n" % (10 * i) for i in range(rid)))       # pread i <- raw read i (phasing_readmap.py:20-23)
    with open(os.path.join(rmd, "pread_to_contigs"), "w") as f:
        f.write("".join(l + "\n" for l in p2c))
    return unzip, rmd, sorted(ctgs), rid


def _run_driver(unzip, rmd, nproc, port):
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK")}
    env.update(FZP_BACKEND="gloo", MASTER_ADDR="127.0.0.1", PYTHONPATH=REPO + os.pathsep + env.get("PYTHONPATH", ""))
    script = os.path.join(REPO, "scripts", "fc_unzip_phase_gpu.py")
    if nproc == 1:
        cmd = [sys.executable, script, "--unzip_dir", unzip, "--read_map_dir", rmd]
    else:
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(nproc), "--master-addr", "127.0.0.1", "--master-port", str(port),
               script, "--unzip_dir", unzip, "--read_map_dir", rmd]
    subprocess.check_call(cmd, env=env, cwd=REPO, timeout=600)


def test_driver_two_ranks_write_what_one_rank_writes(tmp_path):
    """unzip.py:221-288,303-314 on one rank and on two: every per-contig file and rid_to_phase.all byte for byte (the contigs are dealt LPT over the ranks,
    each rank streams its own from the FASTA files, one exchange step; rank 0 writes rid_to_phase.all)."""
    import shutil
    a = tmp_path / "one"
    a.mkdir()
    unzip, rmd, ctgs, n_reads = _make_unzip_tree(str(a))
    b = tmp_path / "two"
    shutil.copytree(str(a), str(b))
    _run_driver(unzip, rmd, 1, 0)
    _run_driver(str(b / "3-unzip"), str(b / "read_maps"), 2, 29761)
    da, db = _tree_digest(unzip), _tree_digest(str(b / "3-unzip"))
    assert da == db and len(da) > 10 * len(ctgs)
    with open(os.path.join(unzip, "1-hasm", "rid-to-phase-all", "rid_to_phase.all")) as f:
        rows = [l.split() for l in f]
    assert len(rows) == n_reads and [r[1] for r in rows] == sorted(r[1] for r in rows)          # sorted-path order = ascending contig id (unzip.py:306-307)
    assert sum(r[2] != "-1" for r in rows) > 0.7 * n_reads
    cat = b"".join(open(os.path.join(unzip, "0-phasing", c, "rid_to_phase.%s" % c), "rb").read() for c in ctgs)
    assert open(os.path.join(unzip, "1-hasm", "rid-to-phase-all", "rid_to_phase.all"), "rb").read() == cat


def test_a_failing_rank_does_not_hang_the_exchange(tmp_path):
    """a rank whose phasing step raises used to leave before the gather, and its peers waited in it for ever.  Two ranks, one contig's reads file gone: both
    ranks end -- non-zero, within the time limit -- the failing one naming the file, the other one saying that a peer failed."""
    unzip, rmd, ctgs, n_reads = _make_unzip_tree(str(tmp_path))
    os.remove(os.path.join(unzip, "reads", "%s_reads.fa" % ctgs[0]))
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK")}
    env.update(FZP_BACKEND="gloo", MASTER_ADDR="127.0.0.1", PYTHONPATH=REPO + os.pathsep + env.get("PYTHONPATH", ""))
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1", "--master-port", "29763",
           os.path.join(REPO, "scripts", "fc_unzip_phase_gpu.py"), "--unzip_dir", unzip, "--read_map_dir", rmd]
    p = subprocess.run(cmd, env=env, cwd=REPO, capture_output=True, timeout=300)
    err = p.stderr.decode(errors="replace")
    assert p.returncode != 0
    assert "%s_reads.fa" % ctgs[0] in err and "another rank failed" in err, err[-3000:]
    assert not os.path.exists(os.path.join(unzip, "1-hasm", "rid-to-phase-all", "rid_to_phase.all"))


def test_files_entry_point_writes_what_the_memory_entry_point_writes(tmp_path):
    """fzp_phase_contigs_files (FASTA parsed by the library, a contig group ahead of the lanes) against fzp_phase_contigs fed by the Python reader: same files,
    same records; also with tiny groups on two lanes (several groups in flight) and with CRLF line ends / blank lines / lower case in the inputs."""
    from falcon_unzip_amd import _lib, pipeline
    unzip, rmd, ctgs, n_reads = _make_unzip_tree(str(tmp_path), n_ctg=6, seed=720)
    # make one reads file ugly: CRLF, a blank line, wrapped sequence lines, lower case
    p = os.path.join(unzip, "reads", "%s_reads.fa" % ctgs[2])
    recs = pipeline.read_fasta(p)
    with open(p, "wb") as f:
        for i, (nm, seq) in enumerate(recs):
            f.write(b">" + nm + b"  extra words\r\n")
            s = seq.lower() if i % 3 == 0 else seq
            for x in range(0, len(s), 61):
                f.write(s[x:x + 61] + (b"\r\n" if i % 2 else b"\n"))
            if i % 5 == 0:
                f.write(b"\n")
    eng = _lib.Engine(0)
    maps = pipeline.load_read_maps(rmd)
    jobs = pipeline.load_contig_jobs(unzip, ctgs)
    out_m = str(tmp_path / "mem")
    recs_m = None
    os.makedirs(out_m)
    contigs = [j[1] for j in jobs]
    names, blobs, read_ctg = [], [], []
    for c, (_, _, reads) in enumerate(jobs):
        for nm, seq in reads:
            names.append(nm); blobs.append(seq); read_ctg.append(c)
    offs = np.zeros(len(blobs) + 1, np.int64)
    offs[1:] = np.cumsum([len(b) for b in blobs])
    _, recs_m = _lib.phase_contigs(eng, contigs, b"".join(blobs), offs, np.array(read_ctg, np.int32), ctgs, names=names, out_dir=out_m, read_maps=maps,
                                   consensus=True, bam=True, sentinels=True)
    for tag, kw in (("files", {}), ("files_small_groups", {"n_lanes": 2, "group_bases": 400_000})):
        out_f = str(tmp_path / tag)
        os.makedirs(out_f)
        st, recs_f = _lib.phase_contigs_files(eng, os.path.join(unzip, "reads"), ctgs, out_dir=out_f, read_maps=maps, consensus=True, bam=True, sentinels=True, **kw)
        assert _tree_digest(out_f) == _tree_digest(out_m), tag
        assert np.array_equal(recs_f, recs_m) and st["n_reads"] == n_reads
        if kw:
            assert st["n_groups"] >= 3
    # a missing file is an error that names it
    with pytest.raises(_lib.FzpError) as e:
        _lib.phase_contigs_files(eng, os.path.join(unzip, "reads"), ctgs + ["999999F"], out_dir=str(tmp_path / "x"))
    assert "999999F_reads.fa" in str(e.value)
    eng.close()


def test_cfg3_shape_500_contigs_on_one_gpu(tmp_path, oracle):
    """BASELINE configs[2]'s job on ONE rank: 500 contigs x 750 kb with 0.5x..2x 2 000 reads of 15 kb each (what `bench.py --gpus N` deals LPT over N ranks
    as its strong_cfg3 leg), generated slab by slab into a preallocated buffer, through fzp_phase_contigs in one call.  Size-independent properties on all
    of it, and five contigs -- the lightest, the heaviest and three in between -- read for read against the twin aligner + the oracle chain."""
    sys.path.insert(0, REPO)
    import bench
    from falcon_unzip_amd import _lib
    from tests import oracle_lib
    nc, L, R = 500, 750_000, 15_000
    scale = float(os.environ.get("FZP_TEST_CFG3_READS", "2000"))
    u = np.random.Generator(np.random.PCG64(20263000)).random(nc)
    n_reads_c = (scale * (0.5 + 1.5 * u)).astype(np.int64)
    import multiprocessing as mp
    total = int(n_reads_c.sum())
    blob = bytearray(int(total * R * 1.06))
    off = np.zeros(total + 1, np.int64)
    read_ctg = np.zeros(total, np.int32)
    contigs = []
    at_r = at_b = 0
    with mp.get_context("spawn").Pool(min(16, bench.host_cores())) as pool:         # (spawn: this process has long initialised the GPU, a fork would take that state along)
        for c0 in range(0, nc, 50):
            res = pool.map(bench.gen_contig, [(3, ci, L, int(n_reads_c[ci]), R, L) for ci in range(c0, min(nc, c0 + 50))])
            for k, (ctg, rb, ro) in enumerate(res):
                contigs.append(ctg)
                n = len(ro) - 1
                blob[at_b:at_b + len(rb)] = rb
                off[at_r + 1:at_r + n + 1] = at_b + ro[1:]
                read_ctg[at_r:at_r + n] = c0 + k
                at_r += n; at_b += len(rb)
    assert at_r == total
    ids = ["%06dF" % c for c in range(nc)]
    name_tab, maps = bench.make_names_and_maps(read_ctg, off, ids, 0)
    eng = _lib.Engine(0)
    out_dir = str(tmp_path / "out")
    st, recs = _lib.phase_contigs(eng, contigs, (C.c_char * at_b).from_buffer(blob), off, read_ctg, ids, names=name_tab, out_dir=out_dir, read_maps=maps, async_writes=True)
    eng.pipe_flush()
    assert st["n_reads"] == total == len(recs) and st["n_aligned"] >= 0.999 * total and st["n_groups"] >= 2
    assert np.array_equal(recs["ctg"], np.sort(recs["ctg"])) and np.array_equal(np.bincount(recs["ctg"], minlength=nc), n_reads_c)
    assert (recs["block"] >= 0).mean() > 0.97
    assert len(os.listdir(out_dir)) == nc and all(len(os.listdir(os.path.join(out_dir, i))) >= 5 for i in ids[::37])
    # spot checks against the CPU chain
    order = np.argsort(n_reads_c)
    for ci in [int(order[0]), int(order[-1])] + [int(x) for x in order[[nc // 4, nc // 2, 3 * nc // 4]]]:
        m = np.flatnonzero(read_ctg == ci)
        reads = [bytes(blob[off[r]:off[r + 1]]) for r in m]
        job = _lib.align_job(eng, [contigs[ci]], reads)
        job.run()
        s = job.summaries()
        exp, exp_cig = oracle_lib.align_reads(oracle, contigs[ci], reads, n_threads=min(16, bench.host_cores()))
        for f in ("aligned", "strand", "pos", "ref_end", "q_start", "q_end", "score", "n_cigar", "cells", "n_columns", "n_match"):
            assert np.array_equal(s[f], exp[f]), (ci, f)
        assert np.array_equal(job.cigar_hashes(), _lib.cigar_hash_of_words(exp_cig)), (ci, "cigar")      # every CIGAR word, through a 64-bit fingerprint per read
        nm = [name_tab[1][name_tab[0][r]:name_tab[0][r + 1]].decode() for r in m]
        aln, _ = job.alnset(0, nm)
        ref = oracle.phase_all(_lib.format_sam(aln, ids[ci]), contigs[ci], ids[ci])
        for rel, key in (("het_call/variant_pos", "variant_pos"), ("het_call/variant_map", "variant_map"), ("g_atable/atable", "atable"),
                         ("get_phased_blocks/phased_variants", "phased_variants"), ("phased_reads", "phased_reads")):
            with open(os.path.join(out_dir, ids[ci], rel), "rb") as f:
                assert f.read() == ref[key], (ci, rel)
        job.close()
    eng.close()


def test_cfg5_full_size_on_one_gpu(tmp_path):
    """BASELINE configs[4] at FULL size on one GPU (120 Mb of contigs of 1-10 Mb, 40x of 15 kb reads: ~330 k reads, 4.9 Gb; K1..K6, every file): the invariants of the
    1/10 slice (tests/test_gpu_scale.py) at the real size, and the peak of the device's memory."""
    sys.path.insert(0, os.path.join(REPO, "tools"))
    import run_cfg5
    r = run_cfg5.run(scale=1.0, lanes=2, workers=8, out_root=str(tmp_path), keep=True)
    st = r["stats"]
    assert st["n_groups"] >= 2 and st["n_reads"] == r["r2p_records"] > 300000 and st["n_aligned"] >= 0.9995 * st["n_reads"]
    assert r["reads_phased"] >= 0.99 * st["n_reads"] and st["n_preads"] >= r["reads_phased"]
    ctgs = os.listdir(r["out_dir"])
    assert len(ctgs) >= 20 and all(sum(len(fs) for _, _, fs in os.walk(os.path.join(r["out_dir"], c))) == 8 for c in ctgs)
    assert r["peak_hbm_gb"] < 200 and st["n_sites"] > 200000 and st["n_pvars"] <= st["n_sites"]


def test_eight_rank_dry_run_of_the_bench():
    """what the round-end driver runs blind: `bench.py --gpus 8` -- here eight gloo ranks on one GPU, two small contigs each: ports, per-rank input generation, the
    strong_cfg3 leg, one JSON line with all eight ranks' work in it"""
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK")}
    env.update(FZP_BENCH_BACKEND="gloo", MASTER_ADDR="127.0.0.1")
    cmd = [sys.executable, os.path.join(REPO, "bench.py"), "--gpus", "8", "--contigs", "2", "--contig-len", "200000", "--reads-per-contig", "60", "--read-len", "6000", "--window", "100000",
           "--steps", "2", "--warmup", "1", "--no-cpu-baseline", "--gen-workers", "1", "--strong-leg-contigs", "12", "--strong-leg-contig-len", "120000"]
    out = subprocess.check_output(cmd, env=env, cwd=REPO, stderr=subprocess.DEVNULL, timeout=900).decode()
    lines = [l for l in out.splitlines() if l.startswith("{")]
    assert len(lines) == 1, out
    d = json.loads(lines[0])
    assert d["n_gpus"] == 8 and d["scaling"] == "weak" and len(d["rank_load"]) == 8 and d["stage_counts"]["r2p_records"] == 8 * 120
    assert d["config"]["reads_total"] == 960 and all(r["reads"] == 120 for r in d["rank_load"])
    assert "rccl_path" in d and "rccl_version" in d and d["gather_fallback"] is None and "error" not in d["from_files"] and d["from_files"]["n_gpus"] == 8
    # r6: the line says what every rank's step was made of, and how the N ranks compare with one of them running alone a moment earlier
    for r in d["rank_load"]:
        assert {"host_cpu_ms_per_step", "phase_write_ms", "gather_ms", "slowest_bracket", "cgroup"} <= set(r) and r["slowest_bracket"]["name"] and r["host_cpu_ms_per_step"] > 0
    se = d["scaling_estimate"]
    assert se["n_gpus"] == 8 and se["solo_ms_per_step"] > 0 and 0 < se["speedup_vs_one_gpu"] < 16 and 0 <= se["slowest_rank"] < 8
    sc = d["strong_cfg3"]
    assert "error" not in sc, sc
    assert sc["n_gpus"] == 8 and sum(r["contigs"] for r in sc["rank_load"]) == 12 and sc["reads_total"] == sc["r2p_records"]
