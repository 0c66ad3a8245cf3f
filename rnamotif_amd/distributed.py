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


def gather_hits(local_hits: np.ndarray, global_index: Sequence[int], stride: int, device=None) -> np.ndarray:
    """Gather int32 hit records [n, stride] from every rank to rank 0.

    `global_index[i]` is the database-wide index of the rank's i-th sequence;
    word 0 of every record is rewritten to it before sending.  Rank 0 returns all
    records sorted by (seq, comp, szero, rank, order) -- the reference's output
    order over the whole database; other ranks return an empty array.
    """
    import torch
    import torch.distributed as dist

    world, rank = dist.get_world_size(), dist.get_rank()
    h = np.ascontiguousarray(local_hits, dtype=np.int32).reshape(-1, stride).copy()
    if h.shape[0]:
        h[:, 0] = np.asarray(global_index, dtype=np.int32)[h[:, 0]]
    dev = device if device is not None else torch.device("cpu")
    n = torch.tensor([h.shape[0]], dtype=torch.int64, device=dev)
    counts = [torch.zeros(1, dtype=torch.int64, device=dev) for _ in range(world)]
    dist.all_gather(counts, n)
    counts = [int(c.item()) for c in counts]
    if rank == 0:
        parts = [h]
        for r in range(1, world):
            if counts[r] == 0:
                continue
            buf = torch.empty((counts[r], stride), dtype=torch.int32, device=dev)
            dist.recv(buf, src=r)
            parts.append(buf.cpu().numpy())
        allh = np.concatenate(parts, axis=0) if parts else h
        if allh.shape[0] > 1:
            order = np.lexsort(allh[:, :5].T[::-1])
            allh = allh[order]
        return allh
    if h.shape[0]:
        dist.send(torch.from_numpy(h).to(dev), dst=0)
    return np.zeros((0, stride), dtype=np.int32)
