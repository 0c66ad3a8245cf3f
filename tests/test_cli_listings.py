"""CPU: the command line program's listings on stderr (-v -s -d -h -p -c, rnamot.c:56-110,
dump.c:34).  No reference-made listing is available here (the reference's rnamotif needs
yacc/lex to build), so the checks are structural: the search order and the length columns
must agree with the compiled program the scanner receives through the C ABI."""
import os
import re
import subprocess

import pytest

import rnamotif_amd as R

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _cli(built, workdir, args):
    p = subprocess.run([built["oracle_cli"]] + args, cwd=workdir, stdout=subprocess.PIPE, stderr=subprocess.PIPE,
                       timeout=120)
    return p.returncode, p.stdout, p.stderr.decode()


def test_version_and_symbols(built, workdir):
    rc, out, err = _cli(built, workdir, ["-v"])
    assert rc == 0 and out == b"" and re.search(r": v\d", err)
    rc, out, err = _cli(built, workdir, ["-s"])
    assert rc == 0 and out == b""
    lines = err.split("\n")
    assert lines[0].startswith("PARMS:") and "global symbols." in lines[0]
    names = [l.split()[0] for l in lines[1:] if "(R" in l]
    assert names == sorted(names)                                   # in-order walk of the symbol tree
    assert "windowsize       (RO) = 6000" in err and 'wc               (RO) = { "a:u", "c:g", "g:c", "u:a" }' in err


@pytest.mark.parametrize("name", ["trna.descr", "pk1.descr", "qu+tr.descr", "pk_j1+2.descr"])
def test_hierarchy_agrees_with_program(built, workdir, name):
    rc, out, err = _cli(built, workdir, ["-c", "-h", "-descr", name])
    assert rc == 0 and out == b""
    cwd = os.getcwd()
    os.chdir(workdir)
    try:
        d = R.Descriptor(["-descr", name])
    finally:
        os.chdir(cwd)
    lines = err.split("\n")
    h = lines.index("desc# minl  maxl  mngl  mxgl  mnil  mxil start  stop  descr")
    rows = [l for l in lines[h + 1:h + 1 + d.n_elems]]
    assert sorted(int(r.split()[0]) for r in rows) == list(range(d.n_elems))
    depth = [l for l in lines if l.startswith("total search depth:")][0]
    assert int(depth.split(":")[1]) == d.n_searches
    s = lines.index("srch# desc# type  forward  backup")
    order = [int(l.split()[1]) for l in lines[s + 1:s + 1 + d.n_searches]]
    assert order == d.search_order()


def test_descr_listing(built, workdir):
    rc, out, err = _cli(built, workdir, ["-c", "-d", "-descr", "pk1.descr"])
    assert rc == 0
    assert "DESCR:   7 structure elements." in err and "SITES:    2 sites." in err
    assert "MAIN SCORE:" in err                                  # -d implies the score listing (rnamot.c:108)
    blk = err[err.index("descr[  0] = {"):err.index("descr[  1] = {")]
    assert "\ttype     = h5\n" in blk and "\tlen      = 3:6\n" in blk and "\tglen     = 22:46\n" in blk
    assert "\tseq      = '^tg'\n" in blk and "\tscopes   = [ 0, 2, 4, 6 ]\n" in blk
