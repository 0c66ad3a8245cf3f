"""CPU: the oracle's energy function against the reference's own efn_drv
(oracle/_ref/efn_drv, built from /root/reference/src/efn.c as it lies) on
random nested structures, and against the known answer of SURVEY.md 8c."""
import ctypes as C
import os
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EFN_DRV = os.path.join(ROOT, "oracle", "_ref", "efn_drv")
EFNDATA = os.path.join(ROOT, "rnamotif_amd", "efndata")


def _oracle_efn(lib, ed, seq, pairs):
    n = len(seq)
    code = {"a": 0, "c": 1, "g": 2, "u": 3, "t": 3}
    bc = (C.c_int * (n + 8))(*([code.get(ch, 4) for ch in seq] + [4] * 8))
    bp = (C.c_int * (n + 8))(*([-1] * (n + 8)))
    for i, j in pairs:
        bp[i], bp[j] = j, i
    lib.rmo_efn.argtypes = [C.c_void_p, C.POINTER(C.c_int), C.POINTER(C.c_int), C.c_int]
    return lib.rmo_efn(ed, bc, bp, n - 1)


def _ct(seq, pairs):
    n = len(seq)
    partner = [0] * n
    for i, j in pairs:
        partner[i], partner[j] = j + 1, i + 1
    lines = ["%5d test" % n]
    for i, ch in enumerate(seq):
        lines.append("%5d %s %5d %5d %5d %5d" % (i + 1, ch.upper(), i, i + 2, partner[i], i + 1))
    return "\n".join(lines) + "\n"


def _random_structure(rng, n):
    """Random nested helices of complementary (incl. G-U) bases with loops >= 3."""
    seq = [("acgu"[int(x)]) for x in rng.integers(0, 4, size=n)]
    pairs = []
    comp = {"a": "u", "u": "a", "c": "g", "g": "c"}

    def fill(lo, hi, depth):
        if hi - lo < 8 or depth > 4:
            return
        i = lo + int(rng.integers(0, 3))
        j = hi - int(rng.integers(0, 3))
        hl = int(rng.integers(2, 6))
        if j - i + 1 < 2 * hl + 3:
            return
        for k in range(hl):
            a = seq[i + k]
            seq[j - k] = comp[a] if rng.random() > 0.15 or a in "ac" else ("g" if a == "u" else "u")
            pairs.append((i + k, j - k))
        ilo, ihi = i + hl, j - hl
        if rng.random() < 0.4 and ihi - ilo > 20:       # multibranch / two stems
            mid = (ilo + ihi) // 2
            fill(ilo + int(rng.integers(0, 2)), mid, depth + 1)
            fill(mid + 1 + int(rng.integers(0, 2)), ihi, depth + 1)
        elif rng.random() < 0.7:
            fill(ilo + int(rng.integers(0, 3)), ihi - int(rng.integers(0, 3)), depth + 1)

    fill(int(rng.integers(0, 3)), n - 1 - int(rng.integers(0, 3)), 0)
    return "".join(seq), pairs


@pytest.fixture(scope="module")
def efn(built):
    lib = C.CDLL(built["oracle_lib"])
    buf = C.create_string_buffer(256 * 1024)
    lib.rmo_load_efndata.argtypes = [C.c_char_p, C.c_void_p]
    assert lib.rmo_load_efndata(EFNDATA.encode(), buf) == 1
    return lib, C.cast(buf, C.c_void_p)


def test_known_answer(efn):
    lib, ed = efn
    assert _oracle_efn(lib, ed, "gaaac", [(0, 4)]) == 570          # energy = 5.700


@pytest.mark.skipif(not os.path.exists(EFN_DRV), reason="oracle/_ref/efn_drv not built (no /root/reference)")
def test_against_reference_efn_drv(efn, tmp_path):
    lib, ed = efn
    rng = np.random.default_rng(5)
    env = dict(os.environ, EFNDATA=EFNDATA)
    checked = 0
    for t in range(300):
        seq, pairs = _random_structure(rng, int(rng.integers(12, 120)))
        if not pairs:
            continue
        f = tmp_path / "s.ct"
        f.write_text(_ct(seq, pairs))
        p = subprocess.run([EFN_DRV, str(f)], stdout=subprocess.PIPE, stderr=subprocess.PIPE, env=env, timeout=60)
        assert p.returncode == 0, p.stderr.decode()
        line = [l for l in p.stdout.decode().splitlines() if l.startswith("energy")][0]
        want = float(line.split("=")[1])
        got = _oracle_efn(lib, ed, seq, pairs)
        assert abs(0.01 * got - want) < 0.0051, (seq, pairs, got, want)
        checked += 1
    assert checked > 200
