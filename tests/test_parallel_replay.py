"""CPU: the replay of candidates on several threads (rnamotif_amd/csrc/rm_driver.cpp ParallelReplayer,
ScoreVM::hit_independent in rm_score.cpp).  The oracle CLI runs the product's driver pipeline, so what
is tested here is the host code the GPU program runs: score programs whose runs cannot see each other
are replayed in parallel and print byte for byte what one thread prints; programs that carry anything
from one candidate to the next -- HOLD/RELEASE, counters, a value or a type left in a variable, a
variable END reads -- are recognised and replayed by one thread."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
MOTIF = "descr\n\th5(len=4)\n\t\tss(minlen=4,maxlen=6)\n\th3\n"

# (name, score section, replayed in parallel?)
PROGRAMS = [
    ("plain", "score\n\t{ SCORE = sprintf( '%5.2f', bits( h5[1], h3[3] ) ); }\n", True),
    ("reject", "score\n\t{ SCORE = length( ss[2] ); if( SCORE < 5 ) REJECT; }\n", True),
    ("loop", "score\n\t{ g = 0; n = length( ss[2] ); for( i = 1; i <= n; i++ ){ b = ss[2,i,1]; if( b == 'g' || b == 'c' ) g++; }\n"
             "\t  SCORE = 1.0 * g / n; if( SCORE < .5 ) REJECT; }\n", True),
    ("while", "score\n\t{ k = 3; t = 0; while( k > 0 ){ t += mispairs( h5[1] ) + k; k--; } SCORE = t; }\n", True),
    ("begin_const", "score\n\tBEGIN { lim = 5; }\n\t{ SCORE = length( ss[2] ); if( SCORE > lim ) REJECT; }\n", True),
    ("two_actions", "score\n\t{ SCORE = 0; }\n\tlength( ss[2] ) == 4 { SCORE = 4; }\n\tlength( ss[2] ) == 6 { REJECT; }\n", True),
    ("counter", "score\n\tBEGIN { n = 0; }\n\t{ n++; SCORE = n; }\n", False),
    ("counter2", "score\n\tBEGIN { n = 0; }\n\t{ n = n + 1; SCORE = n; }\n", False),
    ("plus_assign", "score\n\tBEGIN { n = 0; }\n\t{ n += length( ss[2] ); SCORE = n; }\n", False),
    ("carried_value", "score\n\tBEGIN { y = 0; }\n\t{ if( length( ss[2] ) == 4 ) y = y + 1; SCORE = length( ss[2] ); }\n", False),
    ("stale_value", "score\n\tBEGIN { y = 0; }\n\t{ if( length( ss[2] ) == 4 ) y = 7; SCORE = y; y = 1; }\n", False),
    ("score_sometimes", "score\n\t{ if( length( ss[2] ) == 4 ) SCORE = 1; }\n", False),
    ("type_latch", "score\n\t{ if( length( ss[2] ) == 4 ) v = 1; else v = 2.5; SCORE = v; }\n", False),
    # (SCORE itself typed by the branch taken, an explicit ACCEPT behind the join -- each worker would latch
    # the type from its own first hit: '16' / '1' where one thread prints '16.000' / '1.500' or the other way round)
    ("type_latch_score", "score\n\t{ n = length( ss[2] ); if( n > 4 ) SCORE = n * 2; else SCORE = 1.5; ACCEPT; }\n", False),
    ("type_latch_seq", "score\n\t{ v = 1; v = 2.5; SCORE = v; }\n", False),
    ("end_reads", "score\n\t{ last = length( ss[2] ); SCORE = last; }\n\tEND { last = last + 1; }\n", False),
    ("hold", "score\n\t{ SCORE = length( ss[2] ); if( SCORE == 4 ){ HOLD best; } else ACCEPT; }\n\tEND { RELEASE best; }\n", False),
]


@pytest.fixture(scope="module")
def small_db(gbrna, tmp_path_factory):
    d = tmp_path_factory.mktemp("preplay")
    with open(gbrna, "rb") as f:
        lines = f.readlines()[:6000]
    p = d / "small.fastn"
    p.write_bytes(b"".join(lines))
    return str(p)


def _run(built, cwd, args, threads, timing=True):
    env = dict(os.environ, EFNDATA=os.path.join(ROOT, "rnamotif_amd", "efndata"), RNAMOTIF_REPLAY_THREADS=str(threads),
               RNAMOTIF_BATCH_BASES="60000")
    if timing:
        env["RNAMOTIF_TIMING"] = "1"
    return subprocess.run([built["oracle_cli"]] + args, cwd=cwd, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=900)


@pytest.mark.parametrize("name,score,parallel", PROGRAMS, ids=[p[0] for p in PROGRAMS])
def test_threads_print_what_one_thread_prints(built, small_db, tmp_path, name, score, parallel):
    (tmp_path / "x.descr").write_text(MOTIF + score)
    one = _run(built, str(tmp_path), ["-descr", "x.descr", small_db], 1)
    assert one.returncode == 0, one.stderr.decode()[-2000:]
    assert one.stdout.count(b"\n>") > 200, "the motif should hit often enough for several chunks per batch"
    four = _run(built, str(tmp_path), ["-descr", "x.descr", small_db], 4)
    assert four.returncode == 0, four.stderr.decode()[-2000:]
    line = [l for l in four.stderr.decode().splitlines() if "replay on" in l][0]
    assert ("replay on 4 thread(s)" in line) == parallel, line
    assert four.stdout == one.stdout
    if parallel:
        seven = _run(built, str(tmp_path), ["-descr", "x.descr", small_db], 7)
        assert seven.stdout == one.stdout


def test_a_failing_candidate_ends_the_output_where_one_thread_ends_it(built, small_db, tmp_path):
    """substr() beyond the string fails for the first loop with cgcg in it: the serial loop prints
    the hits before that candidate, then the message; so does the parallel replay."""
    (tmp_path / "x.descr").write_text(MOTIF + "score\n\t{ if( ss[2] =~ 'cgcg' ) SCORE = substr( ss[2], 9, 1 ); else SCORE = 'x'; }\n")
    one = _run(built, str(tmp_path), ["-descr", "x.descr", small_db], 1, timing=False)
    four = _run(built, str(tmp_path), ["-descr", "x.descr", small_db], 4, timing=False)
    assert one.returncode == 1 and four.returncode == 1
    assert b"substr: bad posit" in one.stderr and b"substr: bad posit" in four.stderr
    assert four.stdout == one.stdout and one.stdout.count(b"\n>") >= 20


@pytest.mark.parametrize("name", ["score.1.descr", "score.2.descr", "sprintf.descr", "efn.descr", "getbest.descr"])
def test_reference_score_programs(built, workdir, small_db, name):
    one = _run(built, workdir, ["-descr", name, small_db], 1, timing=False)
    four = _run(built, workdir, ["-descr", name, small_db], 4, timing=False)
    assert one.returncode == 0 and four.returncode == 0
    assert four.stdout == one.stdout and len(one.stdout) > 100


def test_descriptor_that_can_be_read_once(built, small_db, tmp_path):
    """The workers of the parallel replay are compiled from the text the first descriptor was parsed from: a
    descriptor that comes through a pipe (-descr <(cat x.descr)) is read once."""
    import threading
    (tmp_path / "x.descr").write_text(MOTIF + PROGRAMS[0][1])
    want = _run(built, str(tmp_path), ["-descr", "x.descr", small_db], 4)
    assert want.returncode == 0
    fifo = tmp_path / "d.fifo"
    os.mkfifo(fifo)

    def feed():
        with open(fifo, "wb") as f:
            f.write((tmp_path / "x.descr").read_bytes())
    t = threading.Thread(target=feed)
    t.start()
    got = _run(built, str(tmp_path), ["-descr", "d.fifo", small_db], 4)
    t.join()
    assert got.returncode == 0, got.stderr.decode()[-2000:]
    assert "replay on 4 thread(s)" in got.stderr.decode()
    # (the header names the descriptor file: everything after it is the same)
    assert got.stdout.split(b"\n", 3)[3:] == want.stdout.split(b"\n", 3)[3:] and got.stdout.count(b"\n>") > 200
