"""rnamotif_amd -- MI355X-native scan path of rnamotif behind a C ABI.

Thin ctypes mirror of include/rnamotif_amd.h.  The work is done by
``librnamotif_amd.so`` (host front end in C++, search and efn kernels in HIP for
gfx950); this module only moves pointers.  There is no CPU implementation of the
scan in this package: if the library is missing or no GPU is usable the calls
fail loudly.

Reference boundary (see the header for the full table):
  rma_descr_compile  <- RM_init/yyparse/SE_link/RM_linkscore (rnamot.c:49-98)
  rma_scanner_create <- RM_fm_init (find_motif.c:109)
  rma_scan           <- RM_find_motif (find_motif.c:164), both strands
  rma_replay_*       <- RM_score + print_match (score.c:608, find_motif.c:1826)
"""
from __future__ import annotations

import ctypes as C
import os
from typing import Iterable, List, Optional, Sequence, Tuple

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("RNAMOTIF_AMD_LIB") or os.path.join(_HERE, "librnamotif_amd.so")    # (the variable: build variants, profiles/variants.sh)
EFNDATA_DIR = os.path.join(_HERE, "efndata")
CLI_PATH = os.path.join(_HERE, "bin", "rnamotif")

RMA_HIT_HDR = 5
_ERRLEN = 4096


class RnamotifError(RuntimeError):
    pass


_lib = None


def lib() -> C.CDLL:
    """Load librnamotif_amd.so (built in-tree by __graft_entry__.build())."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RnamotifError(
            f"{LIB_PATH} is missing: build it with `make -C rnamotif_amd/csrc` "
            "(hipcc --offload-arch=gfx950); there is no fallback implementation")
    L = C.CDLL(LIB_PATH)
    vp, i32p, i64p = C.c_void_p, C.POINTER(C.c_int32), C.POINTER(C.c_int64)
    cpp = C.POINTER(C.c_char_p)
    L.rma_version.restype = C.c_char_p
    L.rma_device_count.restype = C.c_int
    L.rma_descr_compile.argtypes = [C.c_int, cpp, C.POINTER(vp), C.c_char_p, C.c_size_t]
    L.rma_descr_free.argtypes = [vp]
    L.rma_descr_program.argtypes = [vp]
    L.rma_descr_program.restype = vp
    L.rma_descr_efndata.argtypes = [vp]
    L.rma_descr_efn2data.argtypes = [vp]
    L.rma_descr_efn2data.restype = vp
    L.rma_scanner_set_efn2data.argtypes = [vp, vp, C.c_char_p, C.c_size_t]
    L.rma_descr_efndata.restype = vp
    L.rma_descr_minlen.argtypes = [vp]
    L.rma_descr_maxlen.argtypes = [vp]
    L.rma_program_info.argtypes = [vp, i32p]
    L.rma_program_info.restype = None
    L.rma_scanner_create.argtypes = [vp, vp, C.c_int, C.POINTER(vp), C.c_char_p, C.c_size_t]
    L.rma_scanner_destroy.argtypes = [vp]
    L.rma_scanner_set_option.argtypes = [vp, C.c_char_p, C.c_int, C.c_char_p, C.c_size_t]
    L.rma_scanner_warmup.argtypes = [vp, C.c_char_p, C.c_size_t]
    L.rma_db_attach.argtypes = [vp, vp, C.c_char_p, C.c_size_t]
    L.rma_db_wait.argtypes = [vp, C.c_char_p, C.c_size_t]
    L.rma_db_create_packed_async.argtypes = [vp, vp, C.c_int32, C.c_int32, C.POINTER(vp), C.c_char_p, C.c_size_t]
    L.rma_pack_pin.argtypes = [vp, C.c_char_p, C.c_size_t]
    L.rma_scan_begin.argtypes = [vp, vp, C.c_char_p, C.c_size_t]
    L.rma_scan_end.argtypes = [vp, C.POINTER(i32p), i64p, C.c_char_p, C.c_size_t]
    L.rma_scan_end_on_device.argtypes = [vp, C.POINTER(vp), i64p, C.c_char_p, C.c_size_t]
    L.rma_comm_unique_id.argtypes = [C.c_char_p, C.c_char_p, C.c_size_t]
    L.rma_comm_create.argtypes = [C.c_char_p, C.c_int, C.c_int, C.c_int, C.POINTER(vp), C.c_char_p, C.c_size_t]
    L.rma_comm_destroy.argtypes = [vp]
    L.rma_comm_create_on.argtypes = [vp, C.c_int, C.c_int, C.c_int, C.POINTER(vp), C.c_char_p, C.c_size_t]
    L.rma_comm_count.argtypes = [vp, C.POINTER(C.c_int), C.c_char_p, C.c_size_t]
    L.rma_gather_hits.argtypes = [vp, vp, i32p, C.c_int32, C.c_int, C.POINTER(i32p), i64p, i64p, C.c_char_p, C.c_size_t]
    L.rma_db_create.argtypes = [vp, cpp, i32p, C.c_int32, C.POINTER(vp), C.c_char_p, C.c_size_t]
    L.rma_db_destroy.argtypes = [vp]
    L.rma_db_create_ranges.argtypes = [vp, cpp, i32p, i32p, i32p, C.c_int32, C.POINTER(vp), C.c_char_p, C.c_size_t]
    L.rma_db_create_packed.argtypes = [vp, vp, C.c_int32, C.c_int32, C.POINTER(vp), C.c_char_p, C.c_size_t]
    L.rma_pack_write.argtypes = [C.c_char_p, cpp, cpp, cpp, i32p, C.c_int32, C.c_char_p, C.c_size_t]
    L.rma_pack_read.argtypes = [cpp, C.c_int32, C.c_char_p, C.c_int32, C.c_int32, C.POINTER(vp), C.c_char_p, C.c_size_t]
    L.rma_database_index.argtypes = [cpp, C.c_int32, C.c_char_p, C.c_int32, C.POINTER(i64p), i32p, C.c_char_p, C.c_size_t]
    L.rma_pack_read_entries.argtypes = [cpp, C.c_int32, C.c_char_p, C.c_int32, C.c_int32, i32p, C.c_int32, C.POINTER(vp), C.c_char_p, C.c_size_t]
    L.rma_free.argtypes = [vp]
    L.rma_db_create_packed_ranges.argtypes = [vp, vp, i32p, i32p, i32p, C.c_int32, C.POINTER(vp), C.c_char_p, C.c_size_t]
    L.rma_replay_pack.argtypes = [vp, vp, C.c_int32, i32p, C.c_int64, i64p, C.c_char_p, C.c_size_t]
    L.rma_sort_hits.argtypes = [i32p, C.c_int64, C.c_int32, i32p, C.c_char_p, C.c_size_t]
    L.rma_pack_open.argtypes = [C.c_char_p, C.POINTER(vp), C.c_char_p, C.c_size_t]
    L.rma_pack_close.argtypes = [vp]
    L.rma_pack_count.argtypes = [vp]
    L.rma_pack_count.restype = C.c_int32
    L.rma_pack_bases.argtypes = [vp]
    L.rma_pack_bases.restype = C.c_int64
    L.rma_pack_sid.argtypes = [vp, C.c_int32]
    L.rma_pack_sid.restype = vp
    L.rma_pack_sdef.argtypes = [vp, C.c_int32]
    L.rma_pack_sdef.restype = vp
    L.rma_pack_slen.argtypes = [vp, C.c_int32]
    L.rma_pack_slen.restype = C.c_int32
    L.rma_pack_seq.argtypes = [vp, C.c_int32, C.c_char_p]
    L.rma_db_bases.argtypes = [vp]
    L.rma_db_bases.restype = C.c_int64
    L.rma_scan.argtypes = [vp, vp, C.POINTER(i32p), i64p, C.c_char_p, C.c_size_t]
    L.rma_scanner_last_kernel_ms.argtypes = [vp, C.POINTER(C.c_float), C.c_char_p, C.c_size_t]
    L.rma_scan_device.argtypes = [vp, vp, i64p, C.POINTER(C.c_float), C.POINTER(C.c_float),
                                  C.c_char_p, C.c_size_t]
    L.rma_replay_open.argtypes = [vp, C.c_char_p, C.POINTER(vp), C.c_char_p, C.c_size_t]
    L.rma_replay_batch.argtypes = [vp, cpp, cpp, cpp, i32p, C.c_int32, i32p, C.c_int64, i64p,
                                   C.c_char_p, C.c_size_t]
    L.rma_replay_close.argtypes = [vp, C.c_char_p, C.c_size_t]
    _lib = L
    return L


def _check(rc: int, err) -> None:
    if rc != 0:
        raise RnamotifError(err.value.decode("utf-8", "replace").rstrip())


def _cstr_array(items: Sequence[bytes]):
    arr = (C.c_char_p * max(len(items), 1))()
    for i, s in enumerate(items):
        arr[i] = s
    return arr


class Descriptor:
    """A compiled descriptor: ``Descriptor(["-descr", "trna.descr"])``.

    ``argv`` is the rnamotif command line without the program name.  For
    descriptors whose score section calls efn() the energy tables are read from
    ``efn_datadir`` / $EFNDATA as in the reference; $EFNDATA defaults to the
    tables shipped with the package.
    """

    def __init__(self, argv: Sequence[str]):
        os.environ.setdefault("EFNDATA", EFNDATA_DIR)
        L = lib()
        args = [b"rnamotif"] + [a.encode() for a in argv]
        arr = _cstr_array(args)
        h = C.c_void_p()
        err = C.create_string_buffer(_ERRLEN)
        _check(L.rma_descr_compile(len(args), arr, C.byref(h), err, _ERRLEN), err)
        self._h = h
        self.program = L.rma_descr_program(h)
        self.efndata = L.rma_descr_efndata(h)
        self.efn2data = L.rma_descr_efn2data(h)
        info = (C.c_int32 * 8)()
        L.rma_program_info(self.program, info)
        (self.n_elems, self.n_searches, self.hit_stride, self.ctx_off, self.efn_off,
         self.n_efn_sites, self.both_strands, self.windowsize) = list(info)
        self.minlen = L.rma_descr_minlen(h)
        self.maxlen = L.rma_descr_maxlen(h)

    def search_order(self) -> List[int]:
        """Element index heading each search level (rm_searches[k]->s_descr->s_index)."""
        hdr = (C.c_int32 * (4 + self.n_searches)).from_address(self.program)   # magic, size, n_elems, n_searches, searches[]
        return [int(x) for x in hdr[4:4 + self.n_searches]]

    def close(self) -> None:
        if self._h:
            lib().rma_descr_free(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class Database:
    """Sequences packed 2 bit + ambiguity mask, resident in HBM."""

    def __init__(self, scanner: "Scanner", seqs: Optional[Sequence[bytes]] = None, pack: Optional["Pack"] = None,
                 first: int = 0, count: Optional[int] = None, ranges: Optional[Sequence[Tuple[int, int]]] = None,
                 entries: Optional[Sequence[int]] = None, wait: bool = True):
        L = lib()
        self.scanner = scanner
        h = C.c_void_p()
        err = C.create_string_buffer(_ERRLEN)
        if pack is not None and entries is not None:
            # any entries of a packed database, each with a range of start positions (rma_db_create_packed_ranges)
            n = len(entries)
            assert ranges is None or len(ranges) == n
            self.n_seqs = n
            ent = (C.c_int32 * max(n, 1))(*[int(e) for e in entries])
            lo = (C.c_int32 * max(n, 1))(*[int(r[0]) for r in ranges]) if ranges is not None else None
            hi = (C.c_int32 * max(n, 1))(*[int(r[1]) for r in ranges]) if ranges is not None else None
            _check(L.rma_db_create_packed_ranges(scanner._h, pack._h, ent, lo, hi, n, C.byref(h), err, _ERRLEN), err)
        elif pack is not None:
            # entries [first, first+count) of a packed database, uploaded as they are
            count = pack.count - first if count is None else count
            self.n_seqs = count
            if wait:
                _check(L.rma_db_create_packed(scanner._h, pack._h, first, count, C.byref(h), err, _ERRLEN), err)
            else:
                # the copies run on the device's upload stream; the pack stays as it is until a scan has ended
                self._pack = pack
                _check(L.rma_db_create_packed_async(scanner._h, pack._h, first, count, C.byref(h), err, _ERRLEN), err)
        else:
            self.n_seqs = len(seqs)
            arr = _cstr_array(seqs)
            lens = (C.c_int32 * max(len(seqs), 1))(*[len(s) for s in seqs])
            if ranges is not None:
                # only start positions lo <= szero < hi of each strand of entry i (rma_db_create_ranges)
                lo = (C.c_int32 * max(len(seqs), 1))(*[int(r[0]) for r in ranges])
                hi = (C.c_int32 * max(len(seqs), 1))(*[int(r[1]) for r in ranges])
                _check(L.rma_db_create_ranges(scanner._h, arr, lens, lo, hi, len(seqs), C.byref(h), err, _ERRLEN), err)
            else:
                _check(L.rma_db_create(scanner._h, arr, lens, len(seqs), C.byref(h), err, _ERRLEN), err)
        self._h = h
        self.bases = L.rma_db_bases(h)

    def wait(self) -> None:
        """Until the upload is complete (wait=False databases)."""
        err = C.create_string_buffer(_ERRLEN)
        _check(lib().rma_db_wait(self._h, err, _ERRLEN), err)

    def close(self) -> None:
        if self._h:
            lib().rma_db_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class Scanner:
    """The motif program on one GPU (RM_fm_init + RM_find_motif)."""

    def __init__(self, descr: Descriptor, device: int = 0):
        L = lib()
        self.descr = descr
        h = C.c_void_p()
        err = C.create_string_buffer(_ERRLEN)
        _check(L.rma_scanner_create(descr.program, descr.efndata, device, C.byref(h), err, _ERRLEN), err)
        self._h = h
        if descr.efn2data:
            _check(L.rma_scanner_set_efn2data(h, descr.efn2data, err, _ERRLEN), err)

    def database(self, seqs: Sequence[bytes], ranges: Optional[Sequence[Tuple[int, int]]] = None) -> Database:
        return Database(self, seqs, ranges=ranges)

    def database_from_pack(self, pack: "Pack", first: int = 0, count: Optional[int] = None,
                           entries: Optional[Sequence[int]] = None,
                           ranges: Optional[Sequence[Tuple[int, int]]] = None, wait: bool = True) -> Database:
        return Database(self, pack=pack, first=first, count=count, entries=entries, ranges=ranges, wait=wait)

    def set_option(self, name: str, value: int) -> None:
        """A launch-shape / diagnostic switch between scans (rma_scanner_set_option); the RNAMOTIF_*
        environment is read once, when the scanner is created."""
        err = C.create_string_buffer(_ERRLEN)
        _check(lib().rma_scanner_set_option(self._h, name.encode(), int(value), err, _ERRLEN), err)

    def forget_last(self) -> None:
        """The last scan's records are no longer there for Comm.gather (a round in which this rank has no entries)."""
        self.set_option("forget_last", 1)

    def warmup(self) -> None:
        err = C.create_string_buffer(_ERRLEN)
        _check(lib().rma_scanner_warmup(self._h, err, _ERRLEN), err)

    def attach(self, db: Database) -> None:
        """Lay db out for this scanner ahead of its first scan of it (rma_db_attach)."""
        err = C.create_string_buffer(_ERRLEN)
        _check(lib().rma_db_attach(self._h, db._h, err, _ERRLEN), err)

    def scan_begin(self, db: Database) -> None:
        """The search kernel on its way (rma_scan_begin); scan_end() returns the records."""
        err = C.create_string_buffer(_ERRLEN)
        _check(lib().rma_scan_begin(self._h, db._h, err, _ERRLEN), err)

    def scan_end(self, copy: bool = True) -> np.ndarray:
        L = lib()
        hits = C.POINTER(C.c_int32)()
        n = C.c_int64()
        err = C.create_string_buffer(_ERRLEN)
        _check(L.rma_scan_end(self._h, C.byref(hits), C.byref(n), err, _ERRLEN), err)
        return self._records(hits, n.value, copy)

    def scan_end_on_device(self) -> int:
        """End the scan in flight leaving the ordered records in HBM (for Comm.gather); their number."""
        d = C.c_void_p()
        n = C.c_int64()
        err = C.create_string_buffer(_ERRLEN)
        _check(lib().rma_scan_end_on_device(self._h, C.byref(d), C.byref(n), err, _ERRLEN), err)
        return n.value

    def _records(self, hits, n: int, copy: bool) -> np.ndarray:
        stride = self.descr.hit_stride
        if n == 0:
            return np.zeros((0, stride), dtype=np.int32)
        a = np.ctypeslib.as_array(hits, shape=(n * stride,)).reshape(n, stride)
        return a.copy() if copy else a

    def scan(self, db: Database, copy: bool = True) -> np.ndarray:
        """All candidates of db in reference order: int32 array [n, hit_stride].  With
        copy=False the array is a view of the scanner's own buffer, valid until its next scan."""
        L = lib()
        hits = C.POINTER(C.c_int32)()
        n = C.c_int64()
        err = C.create_string_buffer(_ERRLEN)
        _check(L.rma_scan(self._h, db._h, C.byref(hits), C.byref(n), err, _ERRLEN), err)
        return self._records(hits, n.value, copy)

    def scan_device(self, db: Database) -> Tuple[int, float, float]:
        """Device part only: (candidates, search kernel ms, efn kernel ms)."""
        L = lib()
        n = C.c_int64()
        ms1, ms2 = C.c_float(), C.c_float()
        err = C.create_string_buffer(_ERRLEN)
        _check(L.rma_scan_device(self._h, db._h, C.byref(n), C.byref(ms1), C.byref(ms2), err, _ERRLEN), err)
        return n.value, ms1.value, ms2.value

    def last_kernel_ms(self) -> Tuple[float, float, float]:
        """(search kernel, drain kernel, efn kernel) of the last search in ms; 0 for a kernel that did not run."""
        ms = (C.c_float * 3)()
        err = C.create_string_buffer(_ERRLEN)
        _check(lib().rma_scanner_last_kernel_ms(self._h, ms, err, _ERRLEN), err)
        return ms[0], ms[1], ms[2]

    def close(self) -> None:
        if self._h:
            lib().rma_scanner_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class Replay:
    """Score program + hit printer over candidate records (host side)."""

    def __init__(self, descr: Descriptor, out_path: str = "-"):
        L = lib()
        h = C.c_void_p()
        err = C.create_string_buffer(_ERRLEN)
        _check(L.rma_replay_open(descr._h, out_path.encode(), C.byref(h), err, _ERRLEN), err)
        self._h = h

    def batch(self, sids: Sequence[bytes], sdefs: Sequence[bytes], seqs: Sequence[bytes],
              hits: np.ndarray) -> int:
        L = lib()
        hits = np.ascontiguousarray(hits, dtype=np.int32)
        lens = (C.c_int32 * max(len(seqs), 1))(*[len(s) for s in seqs])
        printed = C.c_int64()
        err = C.create_string_buffer(_ERRLEN)
        _check(L.rma_replay_batch(self._h, _cstr_array(sids), _cstr_array(sdefs), _cstr_array(seqs), lens,
                                  len(seqs), hits.ctypes.data_as(C.POINTER(C.c_int32)), hits.shape[0],
                                  C.byref(printed), err, _ERRLEN), err)
        return printed.value

    def pack(self, pack: "Pack", hits: np.ndarray, first: int = 0) -> int:
        """Candidates over a packed database (word 0 of a record: entry number minus first)."""
        L = lib()
        hits = np.ascontiguousarray(hits, dtype=np.int32)
        printed = C.c_int64()
        err = C.create_string_buffer(_ERRLEN)
        _check(L.rma_replay_pack(self._h, pack._h, first, hits.ctypes.data_as(C.POINTER(C.c_int32)), hits.shape[0],
                                 C.byref(printed), err, _ERRLEN), err)
        return printed.value

    def close(self) -> None:
        if self._h:
            err = C.create_string_buffer(_ERRLEN)
            rc = lib().rma_replay_close(self._h, err, _ERRLEN)
            self._h = None
            _check(rc, err)


class Comm:
    """The native gather of a multi-GPU search (rma_comm_*, rma_gather_hits): RCCL behind the C ABI.
    One per process.  `broadcast(buf: bytearray)` hands rank 0's 128-byte id to every rank -- with
    torch.distributed: a broadcast of a uint8 tensor (rnamotif_amd/distributed.py does that)."""

    def __init__(self, rank: int, world: int, device: int, broadcast=None):
        L = lib()
        ident = C.create_string_buffer(128)
        err = C.create_string_buffer(_ERRLEN)
        if world > 1:
            if rank == 0:
                _check(L.rma_comm_unique_id(ident, err, _ERRLEN), err)
            raw = bytearray(ident.raw)
            broadcast(raw)
            ident = C.create_string_buffer(bytes(raw), 128)
        h = C.c_void_p()
        _check(L.rma_comm_create(ident, rank, world, device, C.byref(h), err, _ERRLEN), err)
        self._h = h
        self.rank, self.world = rank, world

    def gather(self, scanner: Scanner, global_index: Sequence[int], root: int = 0):
        """The records of every rank's last scan (left in HBM) to `root`: (records [n, stride] on root --
        rank by rank, each part in order -- else empty, counts per rank)."""
        L = lib()
        idx = np.ascontiguousarray(np.asarray(global_index, dtype=np.int32))
        hits = C.POINTER(C.c_int32)()
        n = C.c_int64()
        counts = (C.c_int64 * self.world)()
        err = C.create_string_buffer(_ERRLEN)
        _check(L.rma_gather_hits(self._h, scanner._h, idx.ctypes.data_as(C.POINTER(C.c_int32)), len(idx), root,
                                 C.byref(hits), C.byref(n), counts, err, _ERRLEN), err)
        return scanner._records(hits, n.value, True), [int(c) for c in counts]

    def count(self) -> int:
        """The number of ranks the transport itself reports (RCCL: ncclCommCount)."""
        n = C.c_int()
        err = C.create_string_buffer(_ERRLEN)
        _check(lib().rma_comm_count(self._h, C.byref(n), err, _ERRLEN), err)
        return n.value

    def close(self) -> None:
        if self._h:
            lib().rma_comm_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def sort_hits(hits: np.ndarray) -> np.ndarray:
    """Hit records [n, stride] (from several scans, word 0 the database-wide entry number) in the
    reference's output order, order word renumbered: rma_sort_hits()."""
    h = np.ascontiguousarray(hits, dtype=np.int32)
    if h.ndim != 2:
        raise ValueError("hit records are [n, stride]")
    out = np.empty_like(h)
    err = C.create_string_buffer(512)
    i32p = C.POINTER(C.c_int32)
    _check(lib().rma_sort_hits(h.ctypes.data_as(i32p), h.shape[0], h.shape[1], out.ctypes.data_as(i32p), err, len(err)), err)
    return out


def read_fasta(path: str) -> List[Tuple[bytes, bytes, bytes]]:
    """(sid, sdef, seq) per record with the reference reader's normalisation
    (dbutil.c:42-128: every alpha character kept, lower case, u -> t).  Used by
    tests and the benchmark; the CLI has its own C++ reader."""
    import gzip
    op = gzip.open if path.endswith(".gz") else open
    recs: List[Tuple[bytes, bytes, bytes]] = []
    sid = sdef = None
    chunks: List[bytes] = []
    table = bytes(((c | 0x20) if (65 <= c <= 90) else c) for c in range(256)).replace(b"u", b"t")
    keep = bytes(c for c in range(256) if chr(c).isalpha() and c < 128)
    drop = bytes(c for c in range(256) if c not in keep)
    with op(path, "rb") as f:
        for line in f:
            if line.startswith(b">"):
                if sid is not None:
                    recs.append((sid, sdef, b"".join(chunks)))
                hdr = line[1:].strip().split(None, 1)
                sid = hdr[0] if hdr else b""
                sdef = hdr[1].rstrip() if len(hdr) > 1 else b""
                chunks = []
            else:
                chunks.append(line.translate(table, drop))
        if sid is not None:
            recs.append((sid, sdef, b"".join(chunks)))
    return recs


class Pack:
    """A packed database on disk (rma_pack_*): what the readers deliver, in the
    layout the scanner keeps in HBM."""

    def __init__(self, path: Optional[str] = None, _handle=None):
        L = lib()
        h = _handle if _handle is not None else C.c_void_p()
        if _handle is None:
            err = C.create_string_buffer(_ERRLEN)
            _check(L.rma_pack_open(path.encode(), C.byref(h), err, _ERRLEN), err)
        self._h = h
        self.count = int(L.rma_pack_count(h))
        self.bases = int(L.rma_pack_bases(h))

    @staticmethod
    def read(paths: Sequence[str], fmt: str = "", maxslen: int = 0, threads: int = 0) -> "Pack":
        """Sequence files (or packed databases) read the way rnamotif reads them, into memory
        (rma_pack_read): -fmt, -N and every quirk of the reference's readers included."""
        L = lib()
        h = C.c_void_p()
        err = C.create_string_buffer(_ERRLEN)
        arr = _cstr_array([p.encode() for p in paths])
        _check(L.rma_pack_read(arr, len(paths), fmt.encode() if fmt else None, maxslen, threads, C.byref(h), err, _ERRLEN), err)
        return Pack(_handle=h)

    @staticmethod
    def read_entries(paths: Sequence[str], entries: Sequence[int], fmt: str = "", maxslen: int = 0, threads: int = 0) -> Optional["Pack"]:
        """Only the entries with these numbers (ascending; numbered over all files as database_index()
        counts them), read and packed (rma_pack_read_entries); None when the files can only be read whole."""
        L = lib()
        h = C.c_void_p()
        err = C.create_string_buffer(_ERRLEN)
        arr = _cstr_array([p.encode() for p in paths])
        ent = (C.c_int32 * max(len(entries), 1))(*[int(e) for e in entries])
        rc = L.rma_pack_read_entries(arr, len(paths), fmt.encode() if fmt else None, maxslen, threads, ent, len(entries), C.byref(h), err, _ERRLEN)
        if rc == 2:
            return None
        _check(rc, err)
        return Pack(_handle=h)

    def pin(self) -> None:
        """Page-lock the packed words: uploads from this pack are DMA, asynchronous (rma_pack_pin)."""
        err = C.create_string_buffer(_ERRLEN)
        _check(lib().rma_pack_pin(self._h, err, _ERRLEN), err)

    def lengths(self) -> List[int]:
        L = lib()
        return [int(L.rma_pack_slen(self._h, i)) for i in range(self.count)]

    @staticmethod
    def write(path: str, records: Sequence[Tuple[bytes, bytes, bytes]]) -> None:
        """records: (sid, sdef, seq) as read_fasta() returns them."""
        L = lib()
        n = len(records)
        sids = _cstr_array([r[0] for r in records])
        sdefs = _cstr_array([r[1] for r in records])
        seqs = _cstr_array([r[2] for r in records])
        slens = (C.c_int32 * max(n, 1))(*[len(r[2]) for r in records])
        err = C.create_string_buffer(_ERRLEN)
        _check(L.rma_pack_write(path.encode(), sids, sdefs, seqs, slens, n, err, _ERRLEN), err)

    def record(self, i: int) -> Tuple[bytes, bytes, bytes]:
        L = lib()
        n = int(L.rma_pack_slen(self._h, i))
        if n < 0:
            raise IndexError(i)
        buf = C.create_string_buffer(n + 1)
        L.rma_pack_seq(self._h, i, buf)
        return (C.string_at(L.rma_pack_sid(self._h, i)), C.string_at(L.rma_pack_sdef(self._h, i)), buf.raw[:n])

    def close(self) -> None:
        if self._h:
            lib().rma_pack_close(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def database_index(paths: Sequence[str], fmt: str = "", threads: int = 0) -> Optional[List[int]]:
    """The entries of the sequence files, in order, each with an upper bound of its length -- without
    reading the database (rma_database_index); None when the files can only be read whole."""
    L = lib()
    ext = C.POINTER(C.c_int64)()
    n = C.c_int32()
    err = C.create_string_buffer(_ERRLEN)
    arr = _cstr_array([p.encode() for p in paths])
    rc = L.rma_database_index(arr, len(paths), fmt.encode() if fmt else None, threads, C.byref(ext), C.byref(n), err, _ERRLEN)
    if rc == 2:
        return None
    _check(rc, err)
    out = [int(ext[i]) for i in range(n.value)]
    L.rma_free(ext)
    return out


def synthetic_records(k: int, length: int = 1_000_000, seed: int = 20240601) -> List[bytes]:
    """BASELINE.md / SURVEY.md section 8d synthetic database: k records of
    iid uniform acgt from numpy default_rng(seed); the first 10 records of the
    default parameters are the survey's syn10M."""
    rng = np.random.default_rng(seed)
    lut = np.frombuffer(b"acgt", dtype=np.uint8)
    return [lut[rng.integers(0, 4, size=length)].tobytes() for _ in range(k)]


def write_synthetic_fasta(path: str, k: int, length: int = 1_000_000, seed: int = 20240601) -> str:
    """The synthetic database as a FASTA file in the layout SURVEY.md section 8d
    fixes (ids syn%04d, upper case, 50 columns); k=10 gives the survey's syn10M
    (md5 d33c2542e515346e1d0fdfc9edcc5658).  Returns the md5 of the file."""
    import hashlib
    h = hashlib.md5()
    with open(path, "wb") as f:
        for i, s in enumerate(synthetic_records(k, length, seed)):
            head = b">syn%04d synthetic uniform ACGT seed=%d len=%d\n" % (i, seed, length)
            a = np.frombuffer(s.upper(), dtype=np.uint8)
            full = (len(a) // 50) * 50
            body = np.concatenate([a[:full].reshape(-1, 50),
                                   np.full((full // 50, 1), 10, dtype=np.uint8)], axis=1).tobytes()
            if full < len(a):
                body += a[full:].tobytes() + b"\n"
            f.write(head)
            f.write(body)
            h.update(head)
            h.update(body)
    return h.hexdigest()
