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


def partition_slices(extents: Sequence[int], world: int, max_chunk: int = 0) -> List[List[Tuple[int, int, int]]]:
    """partition_ranges() for entries whose lengths are known as upper bounds only (their bytes in
    the file: what the ranks know before anybody has read the database): per rank a list of
    (entry, j, k) -- slice j of k of the entry's start positions, k = 1 for a whole entry -- sorted by
    entry.  A rank that holds a slice reads the whole entry and turns (j, k) into the positions
    [len * j // k, len * (j + 1) // k) once it knows the length; every rank does that alike."""
    total = sum(extents)
    if max_chunk <= 0:
        max_chunk = max(1, -(-total // max(world, 1)))
    work: List[Tuple[int, int, int, int]] = []          # (weight, entry, j, k)
    for i, n in enumerate(extents):
        k = 1 if n <= max_chunk else -(-n // max_chunk)
        for j in range(k):
            work.append((n // k, i, j, k))
    loads = [0] * world
    parts: List[List[Tuple[int, int, int]]] = [[] for _ in range(world)]
    for w in sorted(work, key=lambda w: (-w[0], w[1], w[2])):
        r = min(range(world), key=lambda r: (loads[r], r))
        parts[r].append((w[1], w[2], w[3]))
        loads[r] += w[0]
    for p in parts:
        p.sort()
    return parts


def all_ok(ok: bool, device=None) -> bool:
    """True on every rank iff every rank passes True: one MIN all-reduce.  A rank that fails
    before the gather (no scanner, no memory, a bad file) says so here instead of leaving the
    others waiting in a collective; every rank then leaves with an error of its own."""
    import torch
    import torch.distributed as dist
    if not dist.is_initialized() or dist.get_world_size() == 1:
        return ok
    t = torch.tensor([1 if ok else 0], dtype=torch.int32, device=device if device is not None else torch.device("cpu"))
    dist.all_reduce(t, op=dist.ReduceOp.MIN)
    return bool(int(t.item()))


def all_min(code: int, device=None) -> int:
    """The least of the ranks' codes on every rank (one MIN all-reduce): all_ok() with more than two answers."""
    import torch
    import torch.distributed as dist
    if not dist.is_initialized() or dist.get_world_size() == 1:
        return code
    t = torch.tensor([int(code)], dtype=torch.int32, device=device if device is not None else torch.device("cpu"))
    dist.all_reduce(t, op=dist.ReduceOp.MIN)
    return int(t.item())


_send_cache = {}


def _staging(device, words: int):
    """A device buffer for the records on their way out (reused: no allocation per step)."""
    import torch
    key = str(device)
    buf = _send_cache.get(key)
    if buf is None or buf.numel() < words:
        buf = torch.empty(max(words, 1 << 16), dtype=torch.int32, device=device)
        _send_cache[key] = buf
    return buf


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

    Two collectives per call: an all-gather of the counts (8 bytes per rank) and one gather of
    the records padded to the largest count -- KBs to a few MB, far below what a link carries in
    the time of a scan step; no send/receive per rank, one host synchronisation (the counts).
    """
    import torch
    import torch.distributed as dist

    world, rank = dist.get_world_size(), dist.get_rank()
    h = np.ascontiguousarray(local_hits, dtype=np.int32).reshape(-1, stride)
    dev = device if device is not None else torch.device("cpu")
    n = torch.tensor([h.shape[0]], dtype=torch.int64, device=dev)
    all_n = torch.zeros(world, dtype=torch.int64, device=dev)
    dist.all_gather_into_tensor(all_n, n)
    counts = [int(c) for c in all_n.tolist()]
    most = max(counts)
    if most == 0:
        empty = np.zeros((0, stride), dtype=np.int32)
        return (empty if concat else ([empty] * world if rank == 0 else []))
    # the records, entry numbers made database-wide, padded to the largest count, on the device
    mine = _staging(dev, most * stride)[: most * stride].view(most, stride)
    if h.shape[0]:
        src = torch.from_numpy(h)
        mine[: h.shape[0]].copy_(src, non_blocking=True)
        idx = torch.as_tensor(np.asarray(global_index, dtype=np.int32), device=dev)
        mine[: h.shape[0], 0] = idx[mine[: h.shape[0], 0].long()]
    if rank == 0:
        bufs = [torch.empty((most, stride), dtype=torch.int32, device=dev) for _ in range(world)]
        dist.gather(mine, gather_list=bufs, dst=0)
        parts = [bufs[r][: counts[r]].cpu().numpy() for r in range(world)]
        if not concat:
            return parts
        allh = np.concatenate(parts, axis=0)
        return sort_hits(allh)
    dist.gather(mine, gather_list=None, dst=0)
    return np.zeros((0, stride), dtype=np.int32) if concat else []


class NativeGather:
    """rma_gather_hits() of the C ABI (rnamotif_amd/csrc/rm_gather.cpp): the records of every rank's
    last scan travel from HBM to HBM over RCCL -- an all-gather of the counts and one grouped
    send/receive, no padding, no host hop -- and reach rank 0's host in one copy.  torch.distributed
    only hands rank 0's communicator id to the other ranks (one 128-byte broadcast at start-up)."""

    def __init__(self, rank: int, world: int, device_index: int, coll_device=None):
        import torch
        import torch.distributed as dist
        from . import Comm

        def bcast(buf: bytearray) -> None:
            t = torch.tensor(list(buf), dtype=torch.uint8, device=coll_device if coll_device is not None else torch.device("cpu"))
            dist.broadcast(t, src=0)
            buf[:] = bytes(t.cpu().tolist())

        self.comm = Comm(rank, world, device_index, broadcast=bcast)

    def gather(self, scanner, global_index: Sequence[int], root: int = 0):
        """(records on root -- rank by rank, each part in order -- else empty; counts per rank)"""
        return self.comm.gather(scanner, global_index, root)

    def comm_count(self) -> int:
        return self.comm.count()

    def close(self) -> None:
        self.comm.close()


def sort_hits(allh: np.ndarray) -> np.ndarray:
    """Records by (seq, comp, szero, rank, order) = the reference's output order: the library's
    rma_sort_hits() (host code, the same sort a single scan ends with)."""
    if allh.shape[0] < 2:
        return allh
    from . import sort_hits as _sort
    return _sort(allh)
