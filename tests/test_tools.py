"""CPU: the post-processors of the hit stream -- rm2ct, rmfmt, rmprune -- against outputs
of the reference's own tools: committed fixtures (tests/golden/tools/*.ref, made by
make_tool_goldens.py with oracle/_ref) and, when oracle/_ref is present, the reference
binaries run here on a complete search output."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLD = os.path.join(ROOT, "tests", "golden", "tools")
BIN = os.path.join(ROOT, "rnamotif_amd", "bin")
REF = os.path.join(ROOT, "oracle", "_ref")
ENV = dict(os.environ, LC_ALL="C", EFNDATA=os.path.join(ROOT, "rnamotif_amd", "efndata"))

CASES = [("rm2ct", []), ("rm2ct", ["-t", "rnaviz"]), ("rmfmt", []), ("rmfmt", ["-l"]), ("rmfmt", ["-la"]),
         ("rmfmt", ["-a"]), ("rmfmt", ["-a", "-l"]), ("rmprune", [])]


def _run(exe, opts, data):
    p = subprocess.run([exe] + opts, input=data, env=ENV, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=300)
    return p.returncode, p.stdout


@pytest.mark.parametrize("name", ["trna", "pk1", "score.2"])
@pytest.mark.parametrize("tool,opts", CASES)
def test_tool_matches_reference_fixture(built, name, tool, opts):
    data = open(os.path.join(GOLD, name + ".rm.out"), "rb").read()
    tag = "".join(o.strip("-") for o in opts)
    want = open(os.path.join(GOLD, "%s.%s%s.ref" % (name, tool, "." + tag if tag else "")), "rb").read()
    rc, got = _run(os.path.join(BIN, tool), opts, data)
    assert rc == 0 and got == want


@pytest.mark.skipif(not os.path.exists(os.path.join(REF, "rmprune")), reason="oracle/_ref not built (no /root/reference)")
@pytest.mark.parametrize("name", ["trna.descr", "qu+tr.descr", "sprintf.descr"])
def test_tools_match_reference_binaries_on_full_output(built, workdir, name):
    p = subprocess.run([built["oracle_cli"], "-descr", name, "gbrna.111.0.fastn"], cwd=workdir, env=ENV,
                       stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=900)
    assert p.returncode == 0
    for tool, opts in CASES:
        rc_ref, want = _run(os.path.join(REF, tool), opts, p.stdout)
        rc, got = _run(os.path.join(BIN, tool), opts, p.stdout)
        assert (rc == 0) == (rc_ref == 0), (tool, opts)          # e.g. rm2ct refuses triple / quad helices
        assert got == want, (tool, opts)


def test_pipeline_of_the_reference_test_suite(built, workdir):
    """test/Makefile of the reference: `rnamotif -descr X db | rmfmt -l` must equal X.chk."""
    for name in ("nanlin", "score.1", "mp.ends"):
        p = subprocess.run([built["oracle_cli"], "-descr", name + ".descr", "gbrna.111.0.fastn"], cwd=workdir, env=ENV,
                           stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=900)
        rc, got = _run(os.path.join(BIN, "rmfmt"), ["-l"], p.stdout)
        assert got == open(os.path.join(ROOT, "tests", "golden", "test", name + ".chk"), "rb").read()
