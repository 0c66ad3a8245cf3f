"""ctypes access to oracle/liboracle.so -- TEST INFRASTRUCTURE.

Only tests, __graft_entry__.smoke() and bench.py's cpu_baseline leg import this.
"""
import ctypes as C
import os

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORACLE_LIB = os.path.join(ROOT, "oracle", "liboracle.so")


class RmoHits(C.Structure):
    _fields_ = [("data", C.POINTER(C.c_int32)), ("n", C.c_int64), ("cap", C.c_int64), ("stride", C.c_int)]


_lib = None


def lib():
    global _lib
    if _lib is None:
        L = C.CDLL(ORACLE_LIB)
        L.rmo_hits_init.argtypes = [C.POINTER(RmoHits), C.c_void_p]
        L.rmo_hits_free.argtypes = [C.POINTER(RmoHits)]
        L.rmo_scan.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_char_p, C.c_int, C.c_int, C.POINTER(RmoHits)]
        L.rmo_revcomp.argtypes = [C.c_char_p, C.c_int]
        L.rmo_load_efndata.argtypes = [C.c_char_p, C.c_void_p]
        _lib = L
    return _lib


def oracle_scan(descr, seqs, efn_dir=None):
    """Scalar CPU scan of seqs (list of bytes) with the compiled descriptor's
    program: int32 array [n, stride] in reference order (strand 0 then 1)."""
    L = lib()
    efn = None
    if descr.n_efn_sites:
        # the oracle reads the tables with its own loader (sizeof(rma_efndata_t) < 256 KiB)
        buf = C.create_string_buffer(256 * 1024)
        d = (efn_dir or os.environ.get("EFNDATA") or os.path.join(ROOT, "rnamotif_amd", "efndata")).encode()
        assert L.rmo_load_efndata(d, buf) == 1
        efn = C.cast(buf, C.c_void_p)
    # efn2() sites: the tables as the product's loader read them (checked against the
    # reference's efn2_drv in tests/test_efn2_oracle.py)
    L.rmo_set_efn2data.argtypes = [C.c_void_p]
    L.rmo_set_efn2data(getattr(descr, "efn2data", None))
    hits = RmoHits()
    L.rmo_hits_init(C.byref(hits), descr.program)
    for i, s in enumerate(seqs):
        b = C.create_string_buffer(s, len(s) + 1)
        assert L.rmo_scan(descr.program, efn, i, b, len(s), 0, C.byref(hits)) == 0
        if descr.both_strands:
            L.rmo_revcomp(b, len(s))
            assert L.rmo_scan(descr.program, efn, i, b, len(s), 1, C.byref(hits)) == 0
    n, stride = hits.n, hits.stride
    if n == 0:
        out = np.zeros((0, stride), dtype=np.int32)
    else:
        out = np.ctypeslib.as_array(hits.data, shape=(n * stride,)).reshape(n, stride).copy()
    L.rmo_hits_free(C.byref(hits))
    return out
