"""Multi-GPU sharding of the scan: one process per GPU, sequences partitioned
across ranks, no collective on the data path, one variable-length gather of hit
records to rank 0 at the end (the role mrnamotif's MT_RESULT messages play in
the reference, /root/reference/src/mrnamotif.c:733,910-917).

The backend is whatever torch.distributed was initialised with: "nccl" (RCCL
over xGMI) on GPUs, "gloo" in the CPU tests.
"""
from __future__ import annotations

from typing import List, Sequence, Tuple

import numpy as np


def partition_by_bases(lengths: Sequence[int], world: int) -> List[List[int]]:
    """Longest-first greedy partition of sequence indices over `world` ranks so
    that every rank gets about the same number of bases.  Within a rank the
    indices stay in database order, so a rank's hit stream is already in the
    reference's order for its own sequences."""
    loads = [0] * world
    parts: List[List[int]] = [[] for _ in range(world)]
    for i in sorted(range(len(lengths)), key=lambda i: (-lengths[i], i)):
        r = min(range(world), key=lambda r: (loads[r], r))
        parts[r].append(i)
        loads[r] += lengths[i]
    for p in parts:
        p.sort()
    return parts


def partition_ranges(lengths: Sequence[int], world: int, max_chunk: int = 0) -> List[List[Tuple[int, int, int]]]:
    """Like partition_by_bases(), but entries longer than `max_chunk` start
    positions (default: a world-th of the database) are cut into slices of
    start positions that different ranks search (the C ABI
    takes them as rma_db_create_ranges()).  Returns per rank a list of
    (entry index, lo, hi), sorted by entry and position; every rank that holds a
    slice holds the whole entry's text, so nothing overlaps and nothing is lost
    at the cuts."""
    total = sum(lengths)
    if max_chunk <= 0:
        max_chunk = max(1, -(-total // max(world, 1)))
    work: List[Tuple[int, int, int]] = []
    for i, n in enumerate(lengths):
        if n <= max_chunk:
            work.append((i, 0, max(n, 0)))
        else:
            k = -(-n // max_chunk)
            step = -(-n // k)
            for lo in range(0, n, step):
                work.append((i, lo, min(n, lo + step)))
    loads = [0] * world
    parts: List[List[Tuple[int, int, int]]] = [[] for _ in range(world)]
    for w in sorted(work, key=lambda w: (-(w[2] - w[1]), w[0], w[1])):
        r = min(range(world), key=lambda r: (loads[r], r))
        parts[r].append(w)
        loads[r] += w[2] - w[1]
    for p in parts:
        p.sort()
    return parts


def gather_hits(local_hits: np.ndarray, global_index: Sequence[int], stride: int, device=None,
                concat: bool = True):
    """Gather int32 hit records [n, stride] from every rank to rank 0.

    `global_index[i]` is the database-wide index of the rank's i-th sequence;
    word 0 of every record is rewritten to it before sending.  Rank 0 returns all
    records sorted by (seq, comp, szero, rank, order) -- the reference's output
    order over the whole database; other ranks return an empty array.  With
    concat=False rank 0 gets the per-rank arrays (each in order, in rank order) as a
    list instead: when the ranks hold consecutive runs of entries that list *is* the
    ordered hit stream and a consumer can walk it without the copy.
    """
    import torch
    import torch.distributed as dist

    world, rank = dist.get_world_size(), dist.get_rank()
    h = np.ascontiguousarray(local_hits, dtype=np.int32).reshape(-1, stride).copy()
    if h.shape[0]:
        h[:, 0] = np.asarray(global_index, dtype=np.int32)[h[:, 0]]
    dev = device if device is not None else torch.device("cpu")
    # one collective for the counts, one host sync
    n = torch.tensor([h.shape[0]], dtype=torch.int64, device=dev)
    all_n = torch.zeros(world, dtype=torch.int64, device=dev)
    dist.all_gather_into_tensor(all_n, n)
    counts = [int(c) for c in all_n.tolist()]
    if rank == 0:
        # post every receive, then wait: the transfers of the ranks overlap (xGMI is point to point)
        bufs, reqs = {}, []
        for r in range(1, world):
            if counts[r] == 0:
                continue
            bufs[r] = torch.empty((counts[r], stride), dtype=torch.int32, device=dev)
            reqs.append(dist.irecv(bufs[r], src=r))
        for q in reqs:
            q.wait()
        parts = [h] + [bufs[r].cpu().numpy() for r in sorted(bufs)]
        if not concat:
            return parts
        allh = np.concatenate(parts, axis=0) if len(parts) > 1 else h
        return sort_hits(allh)
    if h.shape[0]:
        dist.send(torch.from_numpy(h).to(dev), dst=0)
    return np.zeros((0, stride), dtype=np.int32) if concat else []


def sort_hits(allh: np.ndarray) -> np.ndarray:
    """Records by (seq, comp, szero, rank, order) = the reference's output order.  Ranks that
    hold consecutive runs of entries deliver an already ordered concatenation: checked first."""
    if allh.shape[0] < 2:
        return allh
    k = allh[:, :5].astype(np.uint64)
    a = (k[:, 0] << np.uint64(32)) | ((k[:, 1] & np.uint64(1)) << np.uint64(31)) | k[:, 2]
    b = (k[:, 3] << np.uint64(32)) | k[:, 4]
    if bool(np.all((a[1:] > a[:-1]) | ((a[1:] == a[:-1]) & (b[1:] >= b[:-1])))):
        return allh
    return allh[np.lexsort((b, a))]
