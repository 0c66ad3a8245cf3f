"""CPU: efn2() -- the product's table loader (rma_efn2data_load) and the oracle's restatement
of RM_efn2 (oracle/rm_oracle_efn2.c) against the reference's own efn2_drv (oracle/_ref/efn2_drv,
built from /root/reference/src/efn2.c as it lies) on random nested structures: hairpins,
bulges, interior loops of every special size, multi-branch loops with coaxial stacking."""
import ctypes as C
import os
import re
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EFN2_DRV = os.path.join(ROOT, "oracle", "_ref", "efn2_drv")
EFNDATA = os.path.join(ROOT, "rnamotif_amd", "efndata")
EFN2_BYTES = 4 * 1024 * 1024          # > sizeof( rma_efn2data_t )


@pytest.fixture(scope="module")
def efn2(built):
    import rnamotif_amd as R
    L = R.lib()
    buf = C.create_string_buffer(EFN2_BYTES)
    err = C.create_string_buffer(4096)
    L.rma_efn2data_load.argtypes = [C.c_char_p, C.c_void_p, C.c_char_p, C.c_size_t]
    assert L.rma_efn2data_load(EFNDATA.encode(), buf, err, 4096) == 0, err.value
    ora = C.CDLL(built["oracle_lib"])
    ora.rmo_efn2.argtypes = [C.c_void_p, C.POINTER(C.c_int), C.POINTER(C.c_int), C.c_int, C.POINTER(C.c_int)]
    return ora, buf


def _oracle(efn2, seq, pairs):
    ora, buf = efn2
    n = len(seq)
    code = {"a": 0, "c": 1, "g": 2, "u": 3, "t": 3}
    bc = (C.c_int * (n + 8))(*([code.get(ch, 4) for ch in seq] + [4] * 8))
    bp = (C.c_int * (n + 8))(*([-1] * (n + 8)))
    for i, j in pairs:
        bp[i], bp[j] = j, i
    und = C.c_int(0)
    e = ora.rmo_efn2(buf, bc, bp, n - 1, C.byref(und))
    return e, und.value


def _ct(seq, pairs):
    n = len(seq)
    partner = [0] * n
    for i, j in pairs:
        partner[i], partner[j] = j + 1, i + 1
    lines = ["%5d test" % n]
    for i, ch in enumerate(seq):
        lines.append("%5d %s %5d %5d %5d %5d" % (i + 1, ch.upper(), i, i + 2, partner[i], i + 1))
    return "\n".join(lines) + "\n"


def _reference(cts):
    p = subprocess.run([EFN2_DRV], input="".join(cts).encode(), env=dict(os.environ, EFNDATA=EFNDATA),
                       stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=120)
    assert p.returncode == 0, p.stderr.decode()
    return [int(round(float(m) * 100)) for m in re.findall(r"dG =\s*(-?[0-9.]+)", p.stdout.decode())]


def _closed_structure(rng, n):
    """Random nested structure closed by its first and last base; loops of all sizes."""
    seq = [("acgu"[int(x)]) for x in rng.integers(0, 4, size=n)]
    pairs = []
    comp = {"a": "u", "u": "a", "c": "g", "g": "c"}

    def helix(i, j, hl):
        for k in range(hl):
            a = seq[i + k]
            seq[j - k] = comp[a] if rng.random() > 0.2 or a in "ac" else ("g" if a == "u" else "u")
            pairs.append((i + k, j - k))

    def fill(i, j, depth):
        # (i, j) will be a helix end to end
        hl = int(rng.integers(1, 6))
        if j - i + 1 < 2 * hl + 3:
            hl = max(1, (j - i + 1 - 3) // 2)
            if j - i + 1 < 2 * hl + 3:
                return False
        helix(i, j, hl)
        a, b = i + hl, j - hl               # interior
        if depth > 4 or b - a + 1 < 7:
            return True
        r = rng.random()
        if r < 0.35:                         # interior loop / bulge, sizes 0..4 each side
            l, rr = int(rng.integers(0, 5)), int(rng.integers(0, 5))
            if l + rr == 0:
                l = 1
            if b - a + 1 - l - rr >= 5:
                fill(a + l, b - rr, depth + 1)
        elif r < 0.7 and b - a + 1 >= 16:    # multi-branch, 2 or 3 branches, gaps 0..2
            nb = 2 if b - a + 1 < 30 else int(rng.integers(2, 4))
            gaps = [int(rng.integers(0, 3)) for _ in range(nb + 1)]
            room = b - a + 1 - sum(gaps)
            w = room // nb
            p = a
            for k in range(nb):
                p += gaps[k]
                q = p + w - 1 if k < nb - 1 else b - gaps[nb]
                if q - p + 1 >= 5:
                    fill(p, q, depth + 1)
                p = q + 1
        return True

    fill(0, n - 1, 0)
    return "".join(seq), pairs


def test_known_answers(efn2):
    # G A A A C with pair 1.5
    assert _oracle(efn2, "gaaac", [(0, 4)]) == (570, 0)
    assert _oracle(efn2, "ggggaaaacccc", [(0, 11), (1, 10), (2, 9), (3, 8)]) == (-540, 0)


@pytest.mark.skipif(not os.path.exists(EFN2_DRV), reason="oracle/_ref/efn2_drv not built (no /root/reference)")
def test_random_closed_structures_match_efn2_drv(efn2):
    rng = np.random.default_rng(20240602)
    cases = [_closed_structure(rng, int(rng.integers(9, 120))) for _ in range(400)]
    want = _reference([_ct(s, p) for s, p in cases])
    assert len(want) == len(cases)
    kinds = 0
    for (s, p), w in zip(cases, want):
        e, und = _oracle(efn2, s, p)
        assert und == 0
        if abs(e) < 90000:                  # %5.2f of the driver keeps two decimals of anything
            assert e == w, (s, p, e, w)
            kinds += 1
        else:
            assert abs(e - w) <= 1, (s, p, e, w)
    assert kinds > 300


@pytest.mark.skipif(not os.path.exists(EFN2_DRV), reason="oracle/_ref/efn2_drv not built (no /root/reference)")
def test_exterior_loop_where_defined_and_flagged_where_not(efn2):
    helix = [(1, 12), (2, 11), (3, 10), (4, 9)]
    e, und = _oracle(efn2, "aggggaaaacccca", helix)          # exterior helix starts at base 2: defined
    assert und == 0 and [e] == _reference([_ct("aggggaaaacccca", helix)])
    two = [(1, 9), (2, 8), (3, 7), (10, 18), (11, 17), (12, 16)]  # second helix flush against the first
    s = "agggaaacccgggaaaccca"
    e, und = _oracle(efn2, s, two)
    assert und == 0 and [e] == _reference([_ct(s, two)])
    # a helix further in: the reference reads rm_basepr[1] == -1 as a partner (efn2.c:1337-1345)
    e, und = _oracle(efn2, "aaggggaaaaccccaa", [(2, 13), (3, 12), (4, 11), (5, 10)])
    assert und == 1 and e == 9999999


@pytest.mark.skipif(not os.path.exists(EFN2_DRV), reason="oracle/_ref/efn2_drv not built (no /root/reference)")
def test_efn2_scores_of_a_search_equal_efn2_drv_on_the_hits(built, workdir):
    """End to end on the CPU: the score column efn2() produces for every hit of a hairpin search
    equals what the reference's pipeline `rnamotif | rm2ct | efn2_drv` gives for the same hits."""
    env = dict(os.environ, EFNDATA=EFNDATA, LC_ALL="C")
    descr = os.path.join(ROOT, "tests", "data", "hairpin.efn2.descr")
    p = subprocess.run([built["oracle_cli"], "-descr", descr, "-N", "400", "gbrna.111.0.fastn"], cwd=workdir, env=env,
                       stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=900)
    assert p.returncode == 0, p.stderr.decode()
    lines = [l for l in p.stdout.decode().split("\n") if l and l[0] not in "#>"]
    assert len(lines) > 2000
    scores = [int(round(float(l.split()[1]) * 100)) for l in lines]
    ct = subprocess.run([os.path.join(ROOT, "rnamotif_amd", "bin", "rm2ct")], input=p.stdout, stdout=subprocess.PIPE,
                        timeout=300).stdout
    q = subprocess.run([EFN2_DRV], input=ct, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=600)
    want = [int(round(float(m) * 100)) for m in re.findall(r"dG =\s*(-?[0-9.]+)", q.stdout.decode())]
    assert len(want) == len(scores)
    # %8.3f of a float holding 0.01*e against %5.2f of a double: equal to the cent
    assert scores == want
