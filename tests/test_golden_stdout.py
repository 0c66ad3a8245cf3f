"""CPU: the host front end (lexer, parser, descriptor compiler, score VM,
printer) + the scalar oracle reproduce the reference's outputs byte for byte.

Pins: md5 of raw stdout recorded from the reference's own objects
 and the reference's test/*.chk files (through the
reference's own rmfmt when oracle/_ref holds it)."""
import hashlib
import os
import subprocess

import pytest

import pins

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF_RMFMT = os.path.join(ROOT, "oracle", "_ref", "rmfmt")


_RUNS = {}      # (the md5 pins and the .chk comparisons look at the same 24 runs)


def _run(built, workdir, args):
    key = tuple(args)
    if key in _RUNS:
        return _RUNS[key]
    env = dict(os.environ, EFNDATA=os.path.join(ROOT, "rnamotif_amd", "efndata"),
               RMO_EFNDATA=os.path.join(ROOT, "rnamotif_amd", "efndata"))
    p = subprocess.run([built["oracle_cli"]] + args + ["gbrna.111.0.fastn"], cwd=workdir, env=env,
                       stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=1800)
    assert p.returncode == 0, p.stderr.decode()
    _RUNS[key] = (p.stdout, p.stderr)
    return _RUNS[key]


# pk1 (18 s) and getbest (7 s) dominate; everything together stays under a minute
@pytest.mark.parametrize("name", sorted(pins.SLACK))
def test_slack_md5(built, workdir, name):
    out, err = _run(built, workdir, ["-descr", name])
    nhits, md5 = pins.SLACK[name]
    assert sum(1 for l in out.split(b"\n") if l.startswith(b">")) == nhits
    assert hashlib.md5(out).hexdigest() == md5
    assert b"complete descr length: min/max" in err       # rnamot.c:89-97


@pytest.mark.parametrize("name", sorted(pins.STRICT))
def test_strict_md5(built, workdir, name):
    out, _ = _run(built, workdir, pins.STRICT_ARGS + ["-descr", name + ".strict.descr"])
    assert hashlib.md5(out).hexdigest() == pins.STRICT[name][1]


# all 12 slack and 12 strict targets of the reference's test/Makefile (pk1 with -sh: 25 s)
CHK = [(n, False) for n in ("nanlin", "pk1", "pk_j1+2", "qu+tr", "score.1", "score.2", "trna", "mp.ends", "efn",
                            "sprintf", "bulge", "getbest")]
CHK += [(n, True) for n, _ in CHK]


@pytest.mark.skipif(not os.path.exists(REF_RMFMT), reason="oracle/_ref/rmfmt not built (no /root/reference)")
@pytest.mark.parametrize("name,strict", CHK, ids=[n + (".strict" if s else "") for n, s in CHK])
def test_chk_files_through_reference_rmfmt(built, workdir, name, strict):
    """test/Makefile:34-37,139-245: rnamotif [-sh -context -Dctx_maxlen=5] -descr NAME[.strict].descr
    gbrna.111.0.fastn | rmfmt -l  must equal the reference's NAME[.strict].chk -- the vectors the
    reference itself holds for this path, through the reference's own rmfmt."""
    stem = name + (".strict" if strict else "")
    out, _ = _run(built, workdir, (pins.STRICT_ARGS if strict else []) + ["-descr", stem + ".descr"])
    p = subprocess.run([REF_RMFMT, "-l"], input=out, cwd=workdir, stdout=subprocess.PIPE, timeout=300,
                       env=dict(os.environ, LC_ALL="C"))
    want = open(os.path.join(ROOT, "tests", "golden", "test", stem + ".chk"), "rb").read()
    assert p.stdout == want
