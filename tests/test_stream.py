"""CPU: the parallel FASTA path of the command line program (rnamotif_amd/csrc/rm_stream.cpp: entries
parsed and packed by worker threads, text rebuilt per hit window) against the serial reader
(rm_fasta.cpp, the restatement of FN_fgetseq, /root/reference/src/dbutil.c:42-128) -- on the
reference's test database and on the irregular inputs where the parallel path has to hand over."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
H = os.path.join(ROOT, "rnamotif_amd", "csrc")
BIN = os.path.join(ROOT, "tests", "_build", "stream_check")


@pytest.fixture(scope="module")
def stream_check():
    os.makedirs(os.path.dirname(BIN), exist_ok=True)
    srcs = [os.path.join(ROOT, "tests", "hostsim", "stream_check.cpp")] + [os.path.join(H, f) for f in ("rm_fasta.cpp", "rm_pack.cpp", "rm_stream.cpp")]
    newest = max(os.path.getmtime(s) for s in srcs + [os.path.join(H, "rm_stream.h"), os.path.join(H, "rm_pack.h")])
    if not os.path.exists(BIN) or os.path.getmtime(BIN) < newest:
        subprocess.run(["g++", "-O2", "-std=c++17", "-pthread", "-I" + os.path.join(ROOT, "include"), "-I" + H, "-o", BIN] + srcs, check=True)
    return BIN


def test_reference_database(stream_check, gbrna):
    for threads, batch in ((1, 1 << 20), (6, 200_000), (3, 5_000)):
        p = subprocess.run([stream_check, gbrna, str(threads), str(batch)], stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=300)
        assert p.returncode == 0 and b"4067 entries identical (4067 through the parallel path)" in p.stdout, p.stdout + p.stderr


CASES = {
    "gt_in_header": b">a one > two >three\nACGT\nACGU\n>b\nNNRY\n",
    "no_trailing_nl": b">a x\nACGT\n>b y\nGGCC",
    "crlf": b">a x\r\nACGT\r\nAC\r\n>b\r\nTT\r\n",
    "unnamed_mid": b">a x\nACGT\n>\nGGGG\n>c z\nTTTT\n",
    "unnamed_blank": b">a x\nACGT\n>   \nGGGG\n>c z\nTTTT\n",
    "not_gt": b"ACGT\n>a\nACGT\n",
    "empty": b"",
    "only_gt": b">",
    "id_only_eof": b">a x\nACGT\n>b",
    "digits_in_seq": b">a x\n1 ACGT 5\n6 acgu 10\n>b\n*-.\n",
    "mid_line_gt": b">a x\nACGT>b y\nTTTT\n",
    "long_def": b">a " + b"d" * 25000 + b"\nACGT\n>b ok\nGG\n",
    "nul_def": b">a de\x00f\nACGT\n>b ok\nGG\n",
    "tabs": b">a\t\tdef here\nACGT\n> b  spaced\nGGA\n",
    "long_sid": b">" + b"s" * 150 + b" def\nACGT\n",
    "empty_seq": b">a x\n>b y\n>c z\nACGT\n",
    "def_spaces_only": b">a    \nACGT\n>b\t\nTT\n",
    "iupac": b">a x\nACGTURYKMSWBDHVNacgturykmswbdhvn\n>b\nXZ-*.\n",
}


@pytest.mark.parametrize("name", sorted(CASES))
def test_irregular_input(stream_check, tmp_path, name):
    """Same entries -- and the same hand-over to the serial reader's diagnostics -- whatever the
    threads, the batch size and the -N limit (4: every entry of 4 letters or more is truncated)."""
    path = tmp_path / (name + ".fa")
    path.write_bytes(CASES[name])
    for threads, maxslen in ((1, 30000001), (3, 30000001), (2, 4)):
        p = subprocess.run([stream_check, str(path), str(threads), "3", str(maxslen)], stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=60)
        assert p.returncode == 0 and b"entries identical" in p.stdout, (name, threads, maxslen, p.stdout, p.stderr[-300:])
