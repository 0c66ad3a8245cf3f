"""mrnamotif -- the rnamotif command line over all GPUs of a node.

    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        -m rnamotif_amd.mrnamotif -descr X.descr [rnamotif options] db.fastn [more files | packs]

One process per GPU.  Every rank reads the database with the library's own readers
(rma_pack_read: what rnamotif reads, -fmt and -N included), takes its share of the
entries -- entries longer than a share are cut into slices of start positions -- and
uploads only that (rma_db_create_packed_ranges); the candidate records travel to rank 0
in one variable-length gather over RCCL (rma_gather_hits of the C ABI: from HBM to HBM;
or torch.distributed's gather where that cannot be set up); rank 0 runs the score program
and prints, in the reference's order, exactly what `rnamotif` prints.  The
reference's own parallel driver hands whole files to MPI workers and collects
their text (/root/reference/src/mrnamotif.c:733,910-917).

Environment: RNAMOTIF_DIST_BACKEND (default nccl = RCCL; gloo: several ranks on one GPU, as the
tests run it), RNAMOTIF_DEVICE (GPU ordinal instead of LOCAL_RANK), RNAMOTIF_GATHER=torch,
RNAMOTIF_OUTPUT (rank 0 writes the hits to this file instead of stdout, which a launcher's or a
transport library's own messages may share).
"""
from __future__ import annotations

import os
import sys
from typing import List, Sequence, Tuple

import numpy as np

import rnamotif_amd as R
from rnamotif_amd.distributed import all_ok, gather_hits, partition_ranges, partition_slices, sort_hits

_VALUE_OPTS = {"-descr", "-xdescr", "-xdfname", "-N", "-fmt"}


def database_files(argv: Sequence[str]) -> List[str]:
    """The sequence files of an rnamotif command line (getargs.c:11-246)."""
    files, skip = [], False
    for a in argv:
        if skip:
            skip = False
        elif a in _VALUE_OPTS:
            skip = True
        elif not a.startswith("-"):
            files.append(a)
    return files


def _option(argv: Sequence[str], name: str, default: str = "") -> str:
    argv = list(argv)
    return argv[argv.index(name) + 1] if name in argv and argv.index(name) + 1 < len(argv) else default


def _read_database(argv: Sequence[str], files: Sequence[str]) -> "R.Pack":
    """The sequence files as rnamotif itself reads them -- -fmt fastn|pir|gb, -N, every quirk of
    FN_/PIR_/GB_fgetseq (dbutil.c:42,130,226): the library's readers (rma_pack_read), packed in
    memory.  Every rank reads the files; none keeps text."""
    n = _option(argv, "-N")
    return R.Pack.read(list(files), fmt=_option(argv, "-fmt"), maxslen=int(n) if n else 0)


def _device_index(local_rank: int) -> int:
    return int(os.environ.get("RNAMOTIF_DEVICE", local_rank))


def _read_share(argv: Sequence[str], files: Sequence[str], world: int, rank: int, dev):
    """What this rank reads and what it scans of it: (pack, [(entry, lo, hi)] with entry numbers of the
    whole database, the entries' numbers within pack).  With several ranks every rank but the first
    reads only its own entries (rma_database_index + rma_pack_read_entries: a gigabase parsed once per
    node, not once per rank); rank 0 reads them all, for it prints the hits of all.  Files that can only
    be read whole (pir, gb, entries with reader diagnostics, -N truncation) are read whole by everyone,
    as one rank alone does."""
    fmt, n = _option(argv, "-fmt"), _option(argv, "-N")
    pack = None
    if world > 1:
        ext = R.database_index(list(files), fmt=fmt)
        if all_ok(ext is not None, dev):
            parts = partition_slices(ext, world)[rank]
            ents = sorted(set(e for e, _, _ in parts))
            if rank == 0:
                pack = _read_database(argv, files)
                ok = pack.count == len(ext)
                local = {e: e for e in ents}
            else:
                pack = R.Pack.read_entries(list(files), ents, fmt=fmt, maxslen=int(n) if n else 0)
                ok = pack is not None
                local = {e: i for i, e in enumerate(ents)}
            if all_ok(ok, dev):
                mine = []
                for e, j, k in parts:
                    slen = R.lib().rma_pack_slen(pack._h, local[e])
                    mine.append((e, slen * j // k, slen * (j + 1) // k) if k > 1 else (e, 0, slen))
                return pack, mine, [local[e] for e, _, _ in parts]
            if rank != 0:
                pack = None
    pack = _read_database(argv, files) if world == 1 or rank != 0 or pack is None else pack
    mine = partition_ranges(pack.lengths(), world)[rank]
    return pack, mine, [i for i, _, _ in mine]


def _scan_shard(descr, pack, entries, ranges, local_rank: int, on_device: bool = False):
    """This rank's share on its GPU: its entries of the packed database, each with its range of
    start positions, straight into HBM (rma_db_create_packed_ranges + rma_scan).  on_device: the
    ordered records stay in HBM for the native gather and the scanner is returned with them."""
    if not entries and not on_device:
        return np.zeros((0, descr.hit_stride), np.int32)
    sc = R.Scanner(descr, device=_device_index(local_rank))
    db = sc.database_from_pack(pack, entries=entries, ranges=ranges)
    if not on_device:
        return sc.scan(db)
    sc.scan_begin(db)
    sc.scan_end_on_device()
    return sc, db


def _init_process_group(world: int, local_rank: int):
    """One process per GPU over RCCL; returns the device the collectives' tensors live on."""
    import datetime
    import torch
    import torch.distributed as dist
    if not torch.cuda.is_available():
        raise SystemExit("mrnamotif needs a GPU: the scan path has no CPU implementation")
    if world == 1:
        return None
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    backend = os.environ.get("RNAMOTIF_DIST_BACKEND", "nccl")
    idx = _device_index(local_rank)
    torch.cuda.set_device(idx)
    if backend != "nccl":
        dist.init_process_group(backend=backend, timeout=datetime.timedelta(minutes=10))
        return torch.device("cpu")
    dev = torch.device("cuda", idx)
    dist.init_process_group(backend="nccl", device_id=dev, timeout=datetime.timedelta(minutes=10))
    return dev


def _native_gather(world: int, rank: int, local_rank: int, dev):
    """The C ABI's gather when RCCL carries the job (every rank agrees), else None."""
    if world == 1 or dev is None or dev.type != "cuda" or os.environ.get("RNAMOTIF_GATHER", "native") != "native":
        return None
    from rnamotif_amd.distributed import NativeGather
    ng = None
    try:
        ng = NativeGather(rank, world, _device_index(local_rank), dev)
    except Exception as e:      # noqa: BLE001 -- the same exchange over torch.distributed then
        sys.stderr.write(f"mrnamotif (rank {rank}): native gather not available ({e})\n")
    return ng if all_ok(ng is not None, dev) else None


def run(argv: Sequence[str], out_path: str = "-") -> int:
    """argv: the rnamotif command line without the program name."""
    import torch.distributed as dist

    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    dev = _init_process_group(world, local_rank)
    rank = dist.get_rank() if world > 1 else 0

    # a rank that cannot do its part says so before the gather: nobody is left waiting
    failure, descr, pack, hits, mine, held = None, None, None, None, [], None
    native = _native_gather(world, rank, local_rank, dev)
    try:
        descr = R.Descriptor(list(argv))
        files = database_files(argv)
        if not files:
            raise R.RnamotifError("mrnamotif: no sequence file")
        if rank == 0 and "-descr" in argv:
            mx = "UNBND" if descr.maxlen == 0x7fffffff else str(descr.maxlen)
            sys.stderr.write(f"{_option(argv, '-descr')}: complete descr length: min/max = {descr.minlen}/{mx}\n")
        pack, mine, local = _read_share(argv, files, world, rank, dev)
        if native is not None:
            held = _scan_shard(descr, pack, local, [(lo, hi) for _, lo, hi in mine], local_rank, on_device=True)
        else:
            hits = _scan_shard(descr, pack, local, [(lo, hi) for _, lo, hi in mine], local_rank)
    except Exception as e:      # noqa: BLE001 -- reported below, on every rank
        failure = e
    if not all_ok(failure is None, dev):
        if failure is not None:
            sys.stderr.write(f"mrnamotif (rank {rank}): {failure}\n")
        if world > 1:
            dist.destroy_process_group()
        return 1
    if native is not None:
        # from HBM to HBM; the ranks' entries interleave (greedy partition), so rank 0 merges the parts
        hits, _ = native.gather(held[0], [i for i, _, _ in mine])
        hits = sort_hits(hits)
        native.close()
    elif world > 1:
        hits = gather_hits(hits, [i for i, _, _ in mine], descr.hit_stride, device=dev)
    elif hits.shape[0]:
        hits = hits.copy()
        hits[:, 0] = np.asarray([i for i, _, _ in mine], dtype=np.int32)[hits[:, 0]]
        hits = sort_hits(hits)
    if rank == 0:
        rp = R.Replay(descr, out_path)
        rp.pack(pack, hits)
        rp.close()
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    return 0


if __name__ == "__main__":
    sys.exit(run(sys.argv[1:], os.environ.get("RNAMOTIF_OUTPUT", "-")))
