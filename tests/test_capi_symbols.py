"""CPU: the C-ABI library loads and exports every function include/rnamotif_amd.h
declares; the front-end half works without a GPU and the scanner half refuses
to run without one (no CPU fallback)."""
import ctypes
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared():
    text = open(os.path.join(ROOT, "include", "rnamotif_amd.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(rma_[a-z_0-9]+)\s*\(", text)))


def test_every_declared_symbol_is_exported(built):
    lib = ctypes.CDLL(built["lib"])
    names = _declared()
    assert len(names) >= 18
    for n in names:
        assert hasattr(lib, n), n


def test_front_end_without_gpu_and_loud_failure(built):
    import rnamotif_amd as R
    d = R.Descriptor(["-descr", os.path.join(ROOT, "tests", "golden", "descr", "trna.descr")])
    assert (d.n_elems, d.n_searches, d.minlen, d.maxlen, d.n_efn_sites) == (15, 11, 63, 95, 1)
    assert d.hit_stride == 5 + 4 * 15 + 4 + 1
    if R.lib().rma_device_count() == 0:
        with pytest.raises(R.RnamotifError, match="GPU only"):
            R.Scanner(d)


def test_descriptor_errors_are_reported_not_fatal(built, tmp_path):
    import rnamotif_amd as R
    bad = tmp_path / "bad.descr"
    bad.write_text("descr\n\th5( minlen=3 ) ss( len=4 )\n")          # h5 without h3
    with pytest.raises(R.RnamotifError, match="no matching"):
        R.Descriptor(["-descr", str(bad)])
    bad.write_text("descr\n\th5 ss( h3\n")
    with pytest.raises(R.RnamotifError, match="syntax error"):
        R.Descriptor(["-descr", str(bad)])
    with pytest.raises(R.RnamotifError):
        R.Descriptor(["-descr", str(tmp_path / "missing.descr")])


def test_author_corpus_compiles(built):
    """61 of the 62 descriptors of descr/ compile; ps.3 is a syntax error by
    design.  (Their scans are compared with the oracle on the GPU, tests/test_gpu_parity.py.)"""
    import rnamotif_amd as R
    d = os.path.join(ROOT, "tests", "golden", "descr")
    cwd = os.getcwd()
    os.chdir(d)
    try:
        bad = []
        names = sorted(f for f in os.listdir(".") if f.endswith(".descr"))
        for n in names:
            try:
                R.Descriptor(["-descr", n])
            except R.RnamotifError as e:
                bad.append((n, str(e)))
    finally:
        os.chdir(cwd)
    assert len(names) == 62
    assert [b[0] for b in bad] == ["ps.3.descr"] and "syntax error" in bad[0][1]
