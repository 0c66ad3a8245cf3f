"""mrnamotif -- the rnamotif command line over all GPUs of a node.

    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        -m rnamotif_amd.mrnamotif -descr X.descr [rnamotif options] db.fastn [more files | packs]

One process per GPU.  Every rank reads the database, takes its share of the
entries -- entries longer than a share are cut into slices of start positions
(rma_db_create_ranges) -- and scans it; the candidate records travel to rank 0
in one variable-length gather over RCCL; rank 0 runs the score program and
prints, in the reference's order, exactly what `rnamotif` prints.  The
reference's own parallel driver hands whole files to MPI workers and collects
their text (/root/reference/src/mrnamotif.c:733,910-917).
"""
from __future__ import annotations

import os
import sys
from typing import List, Sequence, Tuple

import numpy as np

import rnamotif_amd as R
from rnamotif_amd.distributed import gather_hits, partition_ranges, sort_hits

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


def _records(path: str) -> List[Tuple[bytes, bytes, bytes]]:
    with open(path, "rb") as f:
        magic = f.read(8)
    if magic == b"RMAPACK1":
        pk = R.Pack(path)
        return [pk.record(i) for i in range(pk.count)]
    return R.read_fasta(path)


def _scan_shard(descr, seqs, ranges, local_rank: int) -> np.ndarray:
    """This rank's share on its GPU (rma_db_create_ranges + rma_scan)."""
    if not seqs:
        return np.zeros((0, descr.hit_stride), np.int32)
    sc = R.Scanner(descr, device=local_rank)
    return sc.scan(sc.database(seqs, ranges=ranges))


def _init_process_group(world: int, local_rank: int):
    """One process per GPU over RCCL; returns the device the gather uses."""
    import torch
    import torch.distributed as dist
    if not torch.cuda.is_available():
        raise SystemExit("mrnamotif needs a GPU: the scan path has no CPU implementation")
    if world == 1:
        return None
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    dist.init_process_group(backend="nccl", device_id=dev)
    return dev


def run(argv: Sequence[str], out_path: str = "-") -> int:
    """argv: the rnamotif command line without the program name."""
    import torch.distributed as dist

    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    dev = _init_process_group(world, local_rank)
    rank = dist.get_rank() if world > 1 else 0

    if "-fmt" in argv and argv[list(argv).index("-fmt") + 1] != "fastn":
        raise SystemExit("mrnamotif reads fastn files and packed databases; convert other formats with rnamotif_pack")
    descr = R.Descriptor(list(argv))
    files = database_files(argv)
    if not files:
        raise SystemExit("mrnamotif: no sequence file")
    recs: List[Tuple[bytes, bytes, bytes]] = []
    for f in files:
        recs.extend(_records(f))
    if rank == 0 and "-descr" in argv:
        name = argv[list(argv).index("-descr") + 1]
        mx = "UNBND" if descr.maxlen == 0x7fffffff else str(descr.maxlen)
        sys.stderr.write(f"{name}: complete descr length: min/max = {descr.minlen}/{mx}\n")

    mine = partition_ranges([len(r[2]) for r in recs], world)[rank]
    seqs = [recs[i][2] for i, _, _ in mine]
    ranges = [(lo, hi) for _, lo, hi in mine]
    hits = _scan_shard(descr, seqs, ranges, local_rank)
    if world > 1:
        hits = gather_hits(hits, [i for i, _, _ in mine], descr.hit_stride, device=dev)
    elif hits.shape[0]:
        hits = hits.copy()
        hits[:, 0] = np.asarray([i for i, _, _ in mine], dtype=np.int32)[hits[:, 0]]
        hits = sort_hits(hits)
    if rank == 0:
        rp = R.Replay(descr, out_path)
        rp.batch([r[0] for r in recs], [r[1] for r in recs], [r[2] for r in recs], hits)
        rp.close()
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    return 0


if __name__ == "__main__":
    sys.exit(run(sys.argv[1:]))
