"""CPU: database readers either side of the scan path -- FASTA from stdin and from
several files (DB_fnext, dbutil.c:12-40), PIR (PIR_fgetseq :130) and GenBank
(GB_fgetseq :226) flat files, -N truncation, show_progress (rnamot.c:170-174).
The same records through every format give the same hit lines."""
import os
import subprocess

import rnamotif_amd as R

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _cli(built, workdir, args, stdin=None):
    env = dict(os.environ, EFNDATA=os.path.join(ROOT, "rnamotif_amd", "efndata"))
    p = subprocess.run([built["oracle_cli"]] + args, cwd=workdir, env=env, input=stdin,
                       stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=600)
    assert p.returncode == 0, p.stderr.decode()
    return p.stdout, p.stderr


def _records(gbrna, n=400):
    return R.read_fasta(gbrna)[:n]


def _fasta(recs):
    return b"".join(b">" + sid + (b" " + sdef if sdef else b"") + b"\n" + seq + b"\n" for sid, sdef, seq in recs)


def test_stdin_and_multiple_files(built, workdir, gbrna, tmp_path):
    recs = _records(gbrna)
    whole = tmp_path / "all.fastn"
    whole.write_bytes(_fasta(recs))
    a, b = tmp_path / "a.fastn", tmp_path / "b.fastn"
    a.write_bytes(_fasta(recs[:150]))
    b.write_bytes(_fasta(recs[150:]))
    want, _ = _cli(built, workdir, ["-descr", "trna.descr", str(whole)])
    assert want.count(b"\n>") > 10
    got, _ = _cli(built, workdir, ["-descr", "trna.descr", str(a), str(b)])
    assert got == want
    got, _ = _cli(built, workdir, ["-descr", "trna.descr"], stdin=_fasta(recs))
    assert got == want


def test_pir_reader(built, workdir, gbrna, tmp_path):
    recs = _records(gbrna)
    fa = tmp_path / "r.fastn"
    fa.write_bytes(_fasta(recs))
    pir = tmp_path / "r.pir"
    # PIR: '>' id line, title line, sequence; '*' terminators and digits are not letters
    pir.write_bytes(b"".join(b">" + sid + b"\n" + (sdef or b" ") + b"\n" + seq.upper() + b"*\n"
                             for sid, sdef, seq in recs))
    want, _ = _cli(built, workdir, ["-descr", "trna.descr", str(fa)])
    got, _ = _cli(built, workdir, ["-descr", "trna.descr", "-fmt", "pir", str(pir)])
    # an empty FASTA definition prints nothing; the PIR stand-in title is one blank
    assert got.replace(b" \n", b"\n") == want.replace(b" \n", b"\n")


def test_genbank_reader(built, workdir, gbrna, tmp_path):
    recs = [r for r in _records(gbrna) if r[0].count(b"|") == 4]
    gb = tmp_path / "r.gb"
    out = []
    fasta_equiv = []
    for sid, sdef, seq in recs:
        _, gid, _, acc, locus = sid.split(b"|")
        lines = [b"LOCUS       " + locus + b"   %d bp    RNA" % len(seq),
                 b"DEFINITION  " + sdef + b".",
                 b"ACCESSION   " + acc,
                 b"VERSION     " + acc + b".1  GI:" + gid,
                 b"ORIGIN      "]
        for i in range(0, len(seq), 60):
            chunk = seq[i:i + 60]
            lines.append(b"%9d " % (i + 1) + b" ".join(chunk[j:j + 10] for j in range(0, len(chunk), 10)))
        lines.append(b"//")
        out.append(b"\n".join(lines) + b"\n")
        # what GB_fgetseq makes of it: the whole DEFINITION line is the definition
        fasta_equiv.append((sid, b"DEFINITION  " + sdef + b".", seq))
    gb.write_bytes(b"".join(out))
    fa = tmp_path / "r.fastn"
    fa.write_bytes(_fasta(fasta_equiv))
    want, _ = _cli(built, workdir, ["-descr", "trna.descr", str(fa)])
    got, _ = _cli(built, workdir, ["-descr", "trna.descr", "-fmt", "gb", str(gb)])
    assert want.count(b"\n>") > 5
    assert got == want


def test_maxslen_and_show_progress(built, workdir, gbrna, tmp_path):
    recs = _records(gbrna, 30)
    fa = tmp_path / "r.fastn"
    fa.write_bytes(_fasta(recs))
    cut = tmp_path / "cut.fastn"
    cut.write_bytes(_fasta([(sid, sdef, seq[:100]) for sid, sdef, seq in recs]))
    want, _ = _cli(built, workdir, ["-descr", "trna.descr", str(cut)])
    got, err = _cli(built, workdir, ["-descr", "trna.descr", "-N", "100", "-Dshow_progress=10", str(fa)])
    assert got == want
    assert err.count(b"truncated to 100") == sum(1 for r in recs if len(r[2]) > 100)
    prog = [l for l in err.split(b"\n") if b":      " in l and l.rstrip().endswith(tuple(r[0] for r in recs))]
    assert len(prog) == 3 and b"     10: " in prog[0] and prog[0].endswith(recs[9][0])


def test_batch_boundaries_do_not_show(built, workdir, gbrna):
    """The driver hands the scanner batches of entries (RNAMOTIF_BATCH_BASES); stateful score
    programs (getbest: HOLD/RELEASE across entries) must print the same whatever the batch size."""
    outs = []
    for bb in ("1000000000", "3000"):
        env = dict(os.environ, EFNDATA=os.path.join(ROOT, "rnamotif_amd", "efndata"), RNAMOTIF_BATCH_BASES=bb)
        p = subprocess.run([built["oracle_cli"], "-descr", "getbest.descr", "-N", "400", gbrna], cwd=workdir, env=env,
                           stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=900)
        assert p.returncode == 0, p.stderr.decode()
        outs.append(p.stdout)
    assert outs[0] == outs[1] and outs[0].count(b"\n>") > 100
