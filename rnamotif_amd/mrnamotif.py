"""mrnamotif -- the rnamotif command line over all GPUs of a node.

    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        -m rnamotif_amd.mrnamotif -descr X.descr [rnamotif options] db.fastn [more files | packs]

One process per GPU.  Every rank learns the entries of the sequence files without reading them
(rma_database_index), takes its share -- entries longer than a share are cut into slices of start
positions -- and reads, with the library's own readers (what rnamotif reads, -fmt and -N included), ITS
entries only, the first rank too: in rounds of RNAMOTIF_BATCH_BASES (64 Mbase), the entries of the next round
read while this one is uploaded (rma_db_create_packed_ranges), scanned and gathered.  The candidate records of
a round travel to rank 0 in one variable-length gather over RCCL (rma_gather_hits of the C ABI: from HBM to
HBM; or torch.distributed's gather where that cannot be set up); at the end rank 0 reads the text of the
entries that HAVE hits -- of no others -- runs the score program and prints, in the reference's order, exactly
what `rnamotif` prints.  Files whose entries cannot be told apart beforehand (pir, gb, reader diagnostics, -N
truncation) are read whole by every rank, as one rank alone reads them.  The reference's own parallel driver
hands whole files to MPI workers and collects their text (/root/reference/src/mrnamotif.c:733,910-917).

Environment: RNAMOTIF_DIST_BACKEND (default nccl = RCCL; gloo: several ranks on one GPU, as the
tests run it), RNAMOTIF_DEVICE (GPU ordinal instead of LOCAL_RANK), RNAMOTIF_GATHER=torch,
RNAMOTIF_OUTPUT (rank 0 writes the hits to this file instead of stdout, which a launcher's or a
transport library's own messages may share).
"""
from __future__ import annotations

import os
import sys
import time
from typing import List, Sequence, Tuple

import numpy as np

import rnamotif_amd as R
from rnamotif_amd.distributed import all_min, all_ok, gather_hits, partition_ranges, partition_slices, sort_hits

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


def _plan_share(argv: Sequence[str], files: Sequence[str], world: int, rank: int, dev, fit: bool):
    """What every rank is to read and scan, agreed by all: ("shard", ext, parts of every rank) when the entries
    of the files can be told apart without reading them (rma_database_index: FASTA text and packs) -- then a
    rank, the first one too, reads its own entries only --, else ("whole", None, None): pir, gb, entries with
    reader diagnostics, -N truncation -- everybody reads everything, as one rank alone does.  Every rank makes
    the same collective calls whatever happens to it (fit: nothing has failed on this rank so far)."""
    ext = None
    if world > 1 and fit:
        try:
            ext = R.database_index(list(files), fmt=_option(argv, "-fmt"))
        except Exception as e:      # noqa: BLE001 -- a file that cannot be indexed is read whole, where it says why
            sys.stderr.write(f"mrnamotif (rank {rank}): no index of the sequence files ({e}); every rank reads them whole\n")
            ext = None
    if world > 1 and all_ok(ext is not None, dev):
        return "shard", ext, partition_slices(ext, world)
    return "whole", None, None


def _rounds(parts_of_all, ext, batch_bases: int):
    """The shard of every rank in the same number of rounds -- the largest share's bases over batch_bases --, so
    that the gathers behind the rounds pair up: rank r's list of rounds, each a list of (entry, j, k)."""
    weight = lambda p: ext[p[0]] // p[2]        # noqa: E731
    most = max((sum(weight(p) for p in parts) for parts in parts_of_all), default=0)
    n_rounds = max(1, -(-most // max(batch_bases, 1)))
    out = []
    for parts in parts_of_all:
        total = sum(weight(p) for p in parts)
        per = max(1, -(-total // n_rounds))
        rounds, acc = [[] for _ in range(n_rounds)], 0
        for p in parts:
            rounds[min(n_rounds - 1, acc // per)].append(p)
            acc += weight(p)
        out.append(rounds)
    return out


def _read_entries(argv: Sequence[str], files: Sequence[str], parts):
    """The entries of one round, read and packed: (pack, [(entry, lo, hi)], entries' numbers within pack)."""
    fmt, n = _option(argv, "-fmt"), _option(argv, "-N")
    ents = sorted(set(e for e, _, _ in parts))
    if not ents:
        return None, [], []
    pack = R.Pack.read_entries(list(files), ents, fmt=fmt, maxslen=int(n) if n else 0)
    if pack is None:
        raise _NeedWhole()
    local = {e: i for i, e in enumerate(ents)}
    mine = []
    for e, j, k in parts:
        slen = R.lib().rma_pack_slen(pack._h, local[e])
        mine.append((e, slen * j // k, slen * (j + 1) // k) if k > 1 else (e, 0, slen))
    return pack, mine, [local[e] for e, _, _ in parts]


_scanners = {}


def _scanner(descr, local_rank: int):
    """One scanner per descriptor for the whole job (its program, hit buffers and streams are made once)."""
    sc = _scanners.get(id(descr))
    if sc is None:
        sc = _scanners[id(descr)] = R.Scanner(descr, device=_device_index(local_rank))
    return sc


def _scan_shard(descr, pack, entries, ranges, local_rank: int, on_device: bool = False):
    """One round of this rank's share on its GPU: its entries of the packed database, each with its range of
    start positions, straight into HBM (rma_db_create_packed_ranges + rma_scan).  on_device: the
    ordered records stay in HBM for the native gather; the scanner and the database are returned."""
    if not entries and not on_device:
        return np.zeros((0, descr.hit_stride), np.int32)
    sc = _scanner(descr, local_rank)
    if not entries:
        sc.forget_last()            # (a round without entries: nothing to send, whatever the round before left)
        return sc, None
    db = sc.database_from_pack(pack, entries=entries, ranges=ranges)
    if not on_device:
        h = sc.scan(db)
        db.close()
        return h
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


def _lap(rank: int, what: str, t0: float) -> None:
    if os.environ.get("RNAMOTIF_TIMING"):
        sys.stderr.write("[timing] rank %d %s %.1f ms\n" % (rank, what, (time.perf_counter() - t0) * 1e3))


class _NeedWhole(Exception):
    """An entry can only be read in the course of reading its whole file (reader diagnostics, -N)."""


_OK, _WHOLE, _FAIL = 2, 1, 0


def run(argv: Sequence[str], out_path: str = "-") -> int:
    """argv: the rnamotif command line without the program name."""
    import threading
    import torch.distributed as dist

    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    dev = _init_process_group(world, local_rank)
    rank = dist.get_rank() if world > 1 else 0
    batch_bases = int(os.environ.get("RNAMOTIF_BATCH_BASES", str(64 << 20)))

    # Every rank makes the same collective calls in the same order, whatever fails on it: a failure is kept,
    # said in the next agreement (all_ok / all_min), and then every rank leaves -- nobody waits in a collective
    # the others never join.
    st = {"failure": None}
    descr, files = None, []
    native = _native_gather(world, rank, local_rank, dev)

    def leave(code: int) -> int:
        if st["failure"] is not None:
            sys.stderr.write(f"mrnamotif (rank {rank}): {st['failure']}\n")
        if native is not None:
            native.close()
        if world > 1:
            dist.destroy_process_group()
        return code

    try:
        descr = R.Descriptor(list(argv))
        files = database_files(argv)
        if not files:
            raise R.RnamotifError("mrnamotif: no sequence file")
        if rank == 0 and "-descr" in argv:
            mx = "UNBND" if descr.maxlen == 0x7fffffff else str(descr.maxlen)
            sys.stderr.write(f"{_option(argv, '-descr')}: complete descr length: min/max = {descr.minlen}/{mx}\n")
    except Exception as e:      # noqa: BLE001 -- said below, on every rank
        st["failure"] = e
    mode, ext, parts_of_all = _plan_share(argv, files, world, rank, dev, st["failure"] is None)
    if not all_ok(st["failure"] is None, dev):
        return leave(1)

    def rounds_of(plan, whole_pack):
        """The rounds of this rank: plan[ i ] read when its turn comes (shard mode), or the one round over whole_pack."""
        if plan is None:
            mine = partition_ranges(whole_pack.lengths(), world)[rank]
            return [lambda: (whole_pack, mine, [i for i, _, _ in mine])]

        def reader(i):
            def read():
                t0 = time.perf_counter()
                try:
                    return _read_entries(argv, files, plan[i])
                finally:
                    _lap(rank, "read round %d: %d entries" % (i, len(plan[i])), t0)
            return read
        return [reader(i) for i in range(len(plan))]

    def scan_rounds(readers):
        """Read, scan and gather round by round; the entries of round i + 1 are read (a thread of their own: the readers
        leave the interpreter) while round i is uploaded, scanned and gathered.  (_OK, rank 0's parts) / (_WHOLE, None): some
        rank met an entry that can only be read with its whole file / (_FAIL, None)."""
        parts_hits, nxt = [], {}

        def prefetch(i):
            try:
                nxt[i] = readers[i]()
            except Exception as e:      # noqa: BLE001
                nxt[i] = e

        th = threading.Thread(target=prefetch, args=(0,))
        th.start()
        for i in range(len(readers)):
            th.join()
            got = nxt.pop(i)
            if i + 1 < len(readers):
                th = threading.Thread(target=prefetch, args=(i + 1,))
                th.start()
            code = _OK
            if isinstance(got, _NeedWhole):
                code, got = _WHOLE, (None, [], [])
            elif isinstance(got, Exception):
                st["failure"], code, got = got, _FAIL, (None, [], [])
            pack, mine, local = got
            hits, held = None, None
            t0 = time.perf_counter()
            if code == _OK:
                try:
                    if native is not None:
                        held = _scan_shard(descr, pack, local, [(lo, hi) for _, lo, hi in mine], local_rank, on_device=True)
                    else:
                        hits = _scan_shard(descr, pack, local, [(lo, hi) for _, lo, hi in mine], local_rank)
                except Exception as e:      # noqa: BLE001
                    st["failure"], code = e, _FAIL
            _lap(rank, "scan round %d" % i, t0)
            code = all_min(code, dev)
            if code != _OK:
                th.join()
                if held is not None and held[1] is not None:
                    held[1].close()
                return code, None
            t0 = time.perf_counter()
            index = [e for e, _, _ in mine]
            if native is not None:
                # from HBM to HBM; the ranks' entries interleave (greedy partition), so rank 0 merges the parts
                h, _ = native.gather(held[0], index)
                if held[1] is not None:
                    held[1].close()
            elif world > 1:
                h = gather_hits(hits, index, descr.hit_stride, device=dev, concat=False)
                h = np.concatenate(h, axis=0) if h else np.zeros((0, descr.hit_stride), np.int32)
            else:
                h = hits.copy()
                if h.shape[0]:
                    h[:, 0] = np.asarray(index, dtype=np.int32)[h[:, 0]]
            if rank == 0:
                parts_hits.append(h)
            _lap(rank, "gather round %d" % i, t0)
        return _OK, parts_hits

    status, parts_hits, whole_pack = _WHOLE, None, None
    if mode == "shard":
        status, parts_hits = scan_rounds(rounds_of(_rounds(parts_of_all, ext, batch_bases)[rank], None))
    if status == _WHOLE:
        # everybody reads everything and scans its ranges of it, as one rank alone does
        mode = "whole"
        t0 = time.perf_counter()
        try:
            whole_pack = _read_database(argv, files)
        except Exception as e:      # noqa: BLE001
            st["failure"] = e
        _lap(rank, "read %d entries (the whole database)" % (whole_pack.count if whole_pack else 0), t0)
        if not all_ok(st["failure"] is None, dev):
            return leave(1)
        status, parts_hits = scan_rounds(rounds_of(None, whole_pack))
    if status != _OK:
        if status == _WHOLE and st["failure"] is None:
            st["failure"] = R.RnamotifError("mrnamotif: the sequence files changed while they were read")
        return leave(1)

    rc = 0
    if rank == 0:
        try:
            hits = sort_hits(np.concatenate(parts_hits, axis=0)) if parts_hits else np.zeros((0, descr.hit_stride), np.int32)
            t0 = time.perf_counter()
            pack = whole_pack
            if mode == "shard":
                # what is printed needs the text of the entries that have hits, and of no others
                ents = [int(e) for e in np.unique(hits[:, 0])]
                if ents:
                    fmt, n = _option(argv, "-fmt"), _option(argv, "-N")
                    pack = R.Pack.read_entries(list(files), ents, fmt=fmt, maxslen=int(n) if n else 0)
                    if pack is None:
                        pack = _read_database(argv, files)      # (the records keep the entries' numbers in the whole database)
                    else:
                        hits = hits.copy()
                        hits[:, 0] = np.searchsorted(np.asarray(ents, dtype=np.int64), hits[:, 0]).astype(np.int32)
                _lap(rank, "read %d entries with hits" % len(ents), t0)
            rp = R.Replay(descr, out_path)
            if pack is not None:
                rp.pack(pack, hits)
            rp.close()
        except Exception as e:      # noqa: BLE001
            st["failure"], rc = e, 1
    if not all_ok(rc == 0, dev):
        return leave(1)
    if native is not None:
        native.close()
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    return 0


if __name__ == "__main__":
    sys.exit(run(sys.argv[1:], os.environ.get("RNAMOTIF_OUTPUT", "-")))
