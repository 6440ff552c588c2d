"""The phasing section of `unzip_all` (falcon_unzip/unzip.py:221-288) as ONE pypeflow task for a GPU node, and the two `[Unzip]` keys that
select it.  The control plane stays the reference's: this module only supplies what a maintainer plugs into it.

    # falcon_unzip/unzip.py, in main() next to the other [Unzip] options (unzip.py:344-368):
    config.update(unzip_tasks.read_config(cfg))            # phasing_backend = hip | blasr (default), phasing_gpus = 8, sge_phasing_gpu = ...
    # ... and in unzip_all(), around the two per-contig loops (unzip.py:231-288):
    if config.get('phasing_backend') == 'hip':
        rid_to_phase_all = unzip_tasks.add_gpu_phasing_task(wf, config, ctg_ids, PypeTask, makePypeLocalFile)
    else:
        <the blasr loop, the phasing loop and get_rid_to_phase_all as they are>

The task follows the contract of `task_run_blasr` / `task_phasing` (unzip.py:61-133): its body writes a bash script (`set -vex`, `trap 'touch
{job_done}.exit' EXIT`, `touch {job_done}` at the end) and leaves its path in `self.generated_script_fn`; the script starts one process per GPU
(`scripts/fc_unzip_phase_gpu.py` under torch.distributed.run), which writes every file of every contig -- including the per-contig sentinels
`0-phasing/<ctg>/blasr/aln_<ctg>_done` and `0-phasing/<ctg>/phasing/p_<ctg>_done` (unzip.py:241,268), so a later run with the per-contig
tasks finds them done -- and `1-hasm/rid-to-phase-all/rid_to_phase.all` (unzip.py:285), the input of `task_hasm`."""
from __future__ import annotations

import os
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def read_config(cfg):
    """`cfg`: the ConfigParser of fc_unzip.cfg.  -> the keys this backend adds to unzip.py's config dict (absent keys keep the reference's path)"""
    out = {"phasing_backend": "blasr", "phasing_gpus": 1, "sge_phasing_gpu": " -pe smp 24 -q gpu "}
    if cfg.has_option("Unzip", "phasing_backend"):
        out["phasing_backend"] = cfg.get("Unzip", "phasing_backend").strip().lower()
    if cfg.has_option("Unzip", "phasing_gpus"):
        out["phasing_gpus"] = cfg.getint("Unzip", "phasing_gpus")
    if cfg.has_option("Unzip", "sge_phasing_gpu"):
        out["sge_phasing_gpu"] = cfg.get("Unzip", "sge_phasing_gpu")
    if out["phasing_backend"] not in ("blasr", "hip"):
        raise ValueError("[Unzip] phasing_backend must be 'blasr' or 'hip', not %r" % out["phasing_backend"])
    if out["phasing_gpus"] < 1:
        raise ValueError("[Unzip] phasing_gpus must be >= 1")
    return out


def _path(f):
    try:
        from pypeflow.simple_pwatcher_bridge import fn        # the real thing when it is installed
        return fn(f)
    except ImportError:
        return str(f)


def task_phase_gpu(self):
    """pypeflow task body (same shape as unzip.py:61-99): writes p_gpu.sh and hands it to the runner"""
    job_done = _path(self.job_done)
    wd = self.parameters["wd"]
    unzip_dir = self.parameters["unzip_dir"]
    read_map_dir = self.parameters["read_map_dir"]
    gpus = int(self.parameters["config"].get("phasing_gpus", 1))
    py = self.parameters.get("python", sys.executable)
    tool = os.path.join(REPO, "scripts", "fc_unzip_phase_gpu.py")
    if gpus > 1:
        launch = "%s -m torch.distributed.run --nnodes=1 --nproc-per-node %d --master-addr 127.0.0.1 --master-port %d %s" % (py, gpus, int(self.parameters.get("master_port", 29517)), tool)
    else:
        launch = "%s %s" % (py, tool)
    script_fn = os.path.join(wd, "p_gpu.sh")
    script = """\
set -vex
trap 'touch {job_done}.exit' EXIT
cd {wd}
hostname
date
export HSA_ENABLE_IPC_MODE_LEGACY=0
{launch} --unzip_dir {unzip_dir} --read_map_dir {read_map_dir}
date
touch {job_done}
""".format(**locals())
    os.makedirs(wd, exist_ok=True)
    with open(script_fn, "w") as f:
        f.write(script)
    self.generated_script_fn = script_fn


def add_gpu_phasing_task(wf, config, ctg_ids, PypeTask, makePypeLocalFile, unzip_dir="./3-unzip", read_map_dir="./2-asm-falcon/read_maps"):
    """One task instead of 2 x len(ctg_ids) + 1: inputs = every contig's two FASTA files (unzip.py:233-234), outputs = rid_to_phase.all
    (unzip.py:285) and the task's sentinel.  -> the rid_to_phase_all file handle `task_hasm` takes as its input (unzip.py:293)."""
    unzip_dir = os.path.abspath(unzip_dir)
    inputs = {}
    for ctg_id in ctg_ids:
        inputs["ref_%s" % ctg_id] = makePypeLocalFile(os.path.join(unzip_dir, "reads", "%s_ref.fa" % ctg_id))
        inputs["reads_%s" % ctg_id] = makePypeLocalFile(os.path.join(unzip_dir, "reads", "%s_reads.fa" % ctg_id))
    wd = os.path.join(unzip_dir, "0-phasing")
    rid_to_phase_all = makePypeLocalFile(os.path.join(unzip_dir, "1-hasm", "rid-to-phase-all", "rid_to_phase.all"))
    job_done = makePypeLocalFile(os.path.join(wd, "p_gpu_done"))
    parameters = {"job_uid": "ha-gpu", "wd": wd, "config": config, "unzip_dir": unzip_dir, "read_map_dir": os.path.abspath(read_map_dir),
                  "sge_option": config.get("sge_phasing_gpu", "")}
    task = PypeTask(inputs=inputs, outputs={"rid_to_phase_all": rid_to_phase_all, "job_done": job_done}, parameters=parameters)(task_phase_gpu)
    wf.addTask(task)
    return rid_to_phase_all
