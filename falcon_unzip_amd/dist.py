"""Multi-GPU: contigs shard across ranks, ONE exchange step.

The reference runs one isolated job per contig (falcon_unzip/unzip.py:231-281) and then concatenates
every `rid_to_phase.<ctg>` into `rid_to_phase.all` (get_rid_to_phase_all, unzip.py:303-314).  Here:
one process per GPU, contigs dealt by longest-processing-time-first, every rank runs the per-contig
path on its shard with no communication, then a single all-gather of fixed 16-byte records
(arid, contig index, block, phase) assembles the global map on every rank (RCCL over xGMI with
backend "nccl"; "gloo" on CPU for tests).  Payload is tiny (16 B per pread), so the collective is
latency-bound; it is issued once per job, not per contig.
"""
from __future__ import annotations

import numpy as np

from ._lib import R2P


def shard_contigs(weights, world_size):
    """Greedy LPT: heaviest contig first onto the lightest rank.  -> list of index lists, one per rank.

    `weights` ~ sum of read bases per contig (proportional to DP cells).  Deterministic: ties go to the
    lower contig index / lower rank."""
    order = sorted(range(len(weights)), key=lambda i: (-weights[i], i))
    loads = [0] * world_size
    shards = [[] for _ in range(world_size)]
    for i in order:
        r = min(range(world_size), key=lambda k: (loads[k], k))
        shards[r].append(i)
        loads[r] += weights[i]
    for s in shards:
        s.sort()
    return shards


def r2p_from_preads(preads, n_reads, arid_base, ctg_index):
    """phasing_readmap.py:29-46 for a contig whose preads are its reads: read q -> (block, phase),
    last (= highest) block wins, unphased -> (-1, 0).  `preads` ascending (q_id, block)."""
    out = np.zeros(n_reads, R2P)
    out["arid"] = arid_base + np.arange(n_reads, dtype=np.int64)
    out["ctg"] = ctg_index
    out["block"] = -1
    out["phase"] = 0
    if len(preads):
        q = preads["q_id"]
        last = np.ones(len(q), bool)
        last[:-1] = q[1:] != q[:-1]
        out["block"][q[last]] = preads["block"][last]
        out["phase"][q[last]] = preads["phase"][last]
    return out


def r2p_from_batch(preads_all, pread_begin, reads_per_ctg, arid_base, ctg_index_base):
    """r2p_from_preads for every contig of a batch at once.  preads_all: the batch's phased_reads rows (contig-local
    q_ids, contigs concatenated, pread_begin[c] = first row of contig c); reads_per_ctg[c] reads get consecutive arids."""
    reads_per_ctg = np.asarray(reads_per_ctg, dtype=np.int64)
    qoff = np.concatenate(([0], np.cumsum(reads_per_ctg)))
    n = int(qoff[-1])
    out = np.zeros(n, R2P)
    out["arid"] = arid_base + np.arange(n, dtype=np.int64)
    out["ctg"] = ctg_index_base + np.repeat(np.arange(len(reads_per_ctg), dtype=np.int32), reads_per_ctg)
    out["block"] = -1
    if len(preads_all):
        q = preads_all["q_id"].astype(np.int64) + np.repeat(qoff[:-1], np.diff(np.asarray(pread_begin, dtype=np.int64)))
        last = np.ones(len(q), bool)
        last[:-1] = q[1:] != q[:-1]
        out["block"][q[last]] = preads_all["block"][last]
        out["phase"][q[last]] = preads_all["phase"][last]
    return out


def allgather_r2p(local, device=None):
    """All ranks contribute their shard's records; every rank gets all of them, ordered by
    (contig index, arid) -- the order of `rid_to_phase.all` (sorted per-contig paths, unzip.py:306-307).

    Works without an initialised process group (world size 1)."""
    import torch
    import torch.distributed as dist
    local = np.ascontiguousarray(local, dtype=R2P)
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        allr = local
    else:
        world = dist.get_world_size()
        dev = device if device is not None else ("cuda" if dist.get_backend() == "nccl" else "cpu")
        n_local = torch.tensor([len(local)], dtype=torch.int64, device=dev)
        counts = [torch.zeros(1, dtype=torch.int64, device=dev) for _ in range(world)]
        dist.all_gather(counts, n_local)
        counts = [int(c.item()) for c in counts]
        mx = max(counts + [1])
        buf = torch.zeros((mx, 4), dtype=torch.int32, device=dev)
        if len(local):
            buf[:len(local)] = torch.from_numpy(local.view(np.int32).reshape(-1, 4)).to(dev)
        parts = [torch.zeros((mx, 4), dtype=torch.int32, device=dev) for _ in range(world)]
        dist.all_gather(parts, buf)
        allr = np.concatenate([p[:c].cpu().numpy() for p, c in zip(parts, counts)]).reshape(-1, 4)
        allr = np.ascontiguousarray(allr).view(R2P).reshape(-1)
    if len(allr) > 1:      # one rank's records come out of the library in this order already: look before sorting
        c, a = allr["ctg"], allr["arid"]
        dc = np.diff(c)
        if bool(np.all((dc > 0) | ((dc == 0) & (np.diff(a) >= 0)))):
            return allr
    order = np.lexsort((allr["arid"], allr["ctg"]))
    return allr[order]


def format_rid_to_phase_all(records, ctg_ids):
    """`rid_to_phase.all` text: '%09d ctg block phase' rows (phasing_readmap.py:47-51), by the library's formatter (fzp_format_rid_to_phase_all)."""
    from . import _lib
    return _lib.format_rid_to_phase_all(records, ctg_ids)


def make_comm(eng, rank, world, coll_dev):
    """The library's own communicator for the exchange step (fzp_comm over RCCL, one process per GPU), brought up safely: ncclCommInitRank is itself
    collective, so every rank first proves locally that RCCL loads and its context binds (its own unique id is the probe), the ranks agree on that
    with an all_reduce(MIN) over the process group that started them, and only then does anybody enter the communicator's creation; a second
    agreement afterwards.  -> (Comm or None, why-not text or None); None means: use allgather_r2p over torch.distributed (said out loud by the caller)."""
    import torch
    import torch.distributed as tdist
    from . import _lib
    comm, why = None, ""
    try:
        my_id = _lib.comm_unique_id()
    except Exception as e:      # noqa: BLE001
        my_id, why = None, repr(e)
    flag = torch.tensor([1 if my_id is not None else 0], dtype=torch.int32, device=coll_dev)
    tdist.all_reduce(flag, op=tdist.ReduceOp.MIN)
    if int(flag.item()) == 1:
        box = [my_id if rank == 0 else None]
        tdist.broadcast_object_list(box, src=0)
        try:
            comm = _lib.Comm(eng, rank, world, box[0])
            if comm.ranks() != (rank, world):
                raise RuntimeError("communicator reports rank/size %r, expected %r" % (comm.ranks(), (rank, world)))
        except Exception as e:      # noqa: BLE001
            why = repr(e)
            if comm is not None:
                comm.close()
            comm = None
        flag = torch.tensor([1 if comm is not None else 0], dtype=torch.int32, device=coll_dev)
        tdist.all_reduce(flag, op=tdist.ReduceOp.MIN)
        if int(flag.item()) == 0 and comm is not None:
            comm.close()
            comm = None
    if comm is None:
        return None, "rank %d: RCCL C-ABI gather unavailable (%s); falling back to torch.distributed all_gather" % (rank, why or "another rank failed")
    return comm, None


def gather_r2p(local, comm=None):
    """The exchange step: fzp_allgather_rid_to_phase on the library's communicator when there is one, torch.distributed otherwise; either way every
    rank gets all records in (contig index, arid) order."""
    if comm is not None:
        return comm.allgather_r2p(local)
    return allgather_r2p(local)
