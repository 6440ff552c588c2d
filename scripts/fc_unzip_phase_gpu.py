#!/usr/bin/env python
"""Phasing section of fc_unzip.py on GPUs: replaces the per-contig blasr + fc_phasing.py + fc_phasing_readmap.py
jobs and the rid_to_phase gather (falcon_unzip/unzip.py:221-288).  One process per GPU:

    python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 \
        scripts/fc_unzip_phase_gpu.py --unzip_dir ./3-unzip --read_map_dir ./2-asm-falcon/read_maps
"""
import argparse
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))


def main(argv=sys.argv):
    ap = argparse.ArgumentParser(description="GPU phasing of all contigs listed in <unzip_dir>/reads/ctg_list")
    ap.add_argument("--unzip_dir", default="./3-unzip")
    ap.add_argument("--read_map_dir", default=None, help="2-asm-falcon/read_maps (enables rid_to_phase.<ctg> and rid_to_phase.all)")
    args = ap.parse_args(argv[1:])
    world = int(os.environ.get("WORLD_SIZE", "1"))
    os.environ.setdefault("ROC_SIGNAL_POOL_SIZE", "4096")      # before any HIP runtime comes up in this process (under backend nccl torch's does first); include/fzphase.h
    if world > 1:
        import torch
        import torch.distributed as dist
        lr = int(os.environ.get("LOCAL_RANK", "0"))
        backend = os.environ.get("FZP_BACKEND", "nccl")
        if backend == "nccl":                                  # one process per GPU; the exchange step is the library's own RCCL all-gather (pipeline.run)
            torch.cuda.set_device(lr)
            dist.init_process_group(backend="nccl", device_id=torch.device("cuda", lr))
        else:                                                  # dry runs (gloo): the ranks share what devices there are
            dist.init_process_group(backend=backend)
    from falcon_unzip_amd import pipeline
    pipeline.run(args.unzip_dir, args.read_map_dir)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
