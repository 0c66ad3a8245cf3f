"""seq= expressions the packed database cannot decide alone (round 4; refused until round 3): back references
`\\(..\\)..\\1`, and -- with iupac = 0 -- letters that are not acgt as literals or class members.  The scan tests what two
bits and a mask can tell (rma_regex_t::loose: a back reference as `.*`, such a letter as "any letter that is not acgt");
the host applies the expression itself to the text of every candidate it replays (Replayer::one_hit: chk_seq() of
find_motif.c:1810 through the restated step() / advance() of regexp.c:389-664, rm_regex.cpp).

CPU: the oracle-backed command line (the product's host front end around the oracle's scan) against an independent
statement of the same expressions in Python's `re`, start position by start position.  GPU: the product's command line
prints the same bytes."""
import os
import re
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

# (name, parms, element, python expression applied to the element's text, anchored)
CASES = [
    ("backref", "", 'ss(minlen=5,maxlen=7,seq="^\\(a[cg]\\)g\\1")', r"(a[cg])g\1", True),
    ("backref_floating", "", 'ss(minlen=6,maxlen=9,seq="\\(ac\\)g\\1")', r"(ac)g\1", False),
    ("backref_star", "", 'ss(minlen=3,maxlen=8,seq="^\\(ac\\)\\1*g")', r"(ac)(?:\1)*g", True),
    ("two_groups", "", 'ss(minlen=6,maxlen=8,seq="^\\(a\\)\\(c\\)..\\2\\1")', r"(a)(c)..\2\1", True),
    ("literal_n", "iupac = 0;", 'ss(minlen=4,maxlen=5,seq="^nnac")', r"nnac", True),
    ("literal_r_class", "iupac = 0;", 'ss(minlen=3,maxlen=4,seq="^[ar]n[^y]")', r"[ar]n[^y]", True),
    ("literal_mismatch", "iupac = 0;", 'ss(minlen=4,maxlen=4,seq="^nrac$",mismatch=1)', None, True),
]


def _database(path):
    rng = np.random.default_rng(77)
    lut = np.frombuffer(b"acgt", dtype=np.uint8)
    plants = [b"acgac", b"aggag", b"acacacg", b"acg", b"accgca", b"nnac", b"rnac", b"rna", b"anc", b"ynac", b"nracc", b"arac"]
    recs = []
    for k in range(24):
        s = bytearray(lut[rng.integers(0, 4, size=int(rng.integers(150, 400)))].tobytes())
        for i in rng.integers(0, len(s), size=6):
            s[i] = ord("nry"[int(rng.integers(0, 3))])
        for _ in range(10):
            p = plants[int(rng.integers(0, len(plants)))]
            at = int(rng.integers(0, len(s) - len(p)))
            s[at:at + len(p)] = p
        recs.append(bytes(s))
    with open(path, "wb") as f:
        for k, s in enumerate(recs):
            f.write(b">e%d\n" % k + (s.upper() if k % 5 == 0 else s) + b"\n")
    return [s.lower() for s in recs]


def _revcomp(s):
    tr = bytes((ord("n") if chr(c) not in "acgt" else c) for c in range(256)).translate(bytes.maketrans(b"acgt", b"tgca"))
    return s.translate(tr)[::-1]


def _hits(out):
    """(entry, strand, position field, length) of every hit line of rnamotif's output."""
    got = []
    for line in out.decode().splitlines():
        f = line.split()
        if len(f) >= 5 and f[0].startswith("e") and not line.startswith(">") and not line.startswith("#"):
            got.append((f[0], int(f[2]), int(f[3]), int(f[4])))
    return got


def _run(exe, cwd, text, db):
    with open(os.path.join(cwd, "x.descr"), "w") as f:
        f.write(text)
    env = dict(os.environ, EFNDATA=os.path.join(ROOT, "rnamotif_amd", "efndata"))
    return subprocess.run([exe, "-descr", "x.descr", db], cwd=cwd, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=600)


@pytest.mark.parametrize("name,parms,elem,pyre,anchored", CASES, ids=[c[0] for c in CASES])
def test_whole_expression_is_applied_at_replay(built, tmp_path, name, parms, elem, pyre, anchored):
    db = str(tmp_path / "db.fastn")
    seqs = _database(db)
    text = ("parms\n\t" + parms + "\n" if parms else "") + "descr\n\t" + elem + "\n"
    p = _run(built["oracle_cli"], str(tmp_path), text, db)
    assert p.returncode == 0, p.stderr.decode()
    got = _hits(p.stdout)
    lens = [int(x) for x in re.findall(r"(?:min|max)len=(\d+)", elem)]
    want = []
    for k, s in enumerate(seqs):
        for comp, t in ((0, s), (1, _revcomp(s))):
            for z in range(len(t)):
                for n in range(lens[1], lens[0] - 1, -1):      # (find_motif :273: the end position from the highest down)
                    if z + n > len(t):
                        continue
                    sub = t[z:z + n].decode()
                    if pyre is None:
                        # "^nrac$" with one mismatch: the letters compared one by one (mm_advance, mm_regexp.c:369)
                        ok = sum(a != b for a, b in zip(sub, "nrac")) <= 1
                    else:
                        ok = (re.match(pyre, sub) if anchored else re.search(pyre, sub)) is not None
                    if ok:
                        want.append(("e%d" % k, comp, len(t) - z if comp else z + 1, n))
    assert len(want) >= 5, "the database should hold some matches"
    assert got == want
    if "iupac" not in parms:
        assert b"mm_seqlen: 36?" in p.stderr or b"mm_seqlen: 37?" in p.stderr      # (mm_seqlen of a CBACK or CBACK | STAR opcode, mm_regexp.c:196)


def test_residual_refusals(built, tmp_path):
    """What is still refused, with its words: such an expression in a context element, and mismatches next to groups."""
    import rnamotif_amd as R
    for text, words in (('descr\n\tss(minlen=6,maxlen=6,seq="^\\(ac\\)g\\1",mismatch=1)\n', "mismatches"),
                        ('parms\n\tiupac = 0;\ndescr\n\tctx(minlen=2,maxlen=4,seq="nn")\n\tss(minlen=4,maxlen=6)\n', "context element")):
        (tmp_path / "r.descr").write_text(text)
        with pytest.raises(R.RnamotifError, match=words):
            R.Descriptor(["-context", "-descr", str(tmp_path / "r.descr")])


@pytest.mark.gpu
@pytest.mark.parametrize("name,parms,elem,pyre,anchored", CASES, ids=[c[0] for c in CASES])
def test_gpu_command_line_prints_the_same(built, tmp_path, name, parms, elem, pyre, anchored):
    db = str(tmp_path / "db.fastn")
    _database(db)
    # (the element inside a hairpin too: the helix' strands and the loop are tested in the kernels, the expression at replay)
    for text in (("parms\n\t" + parms + "\n" if parms else "") + "descr\n\t" + elem + "\n",
                 ("parms\n\t" + parms + "\n" if parms else "") + "descr\n\th5(minlen=2,maxlen=3)\n\t\t" + elem + "\n\th3\n"):
        want = _run(built["oracle_cli"], str(tmp_path), text, db)
        got = _run(built["cli"], str(tmp_path), text, db)
        assert want.returncode == 0 and got.returncode == 0, got.stderr.decode()
        assert got.stdout == want.stdout
        assert len(_hits(want.stdout)) >= 1
