"""CPU, under AddressSanitizer: errors while batches are in flight in the command line driver's
pipeline (rnamotif_amd/csrc/rm_driver.cpp).  The scanner is a fake that keeps reading the pack it
was handed (tests/hostsim/driver_fault_check.cpp); a later file that is not a packed database, an
upload that fails, a scan that fails -- each must end with its message and exit code 1, and no
thread may touch a pack after the driver has let go of it."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
H = os.path.join(ROOT, "rnamotif_amd", "csrc")
BIN = os.path.join(ROOT, "tests", "_build", "driver_fault_check")
HOST = ("rm_regex.cpp", "rm_compile.cpp", "rm_parse.cpp", "rm_score.cpp", "rm_efndata.cpp", "rm_efn2data.cpp", "rm_fasta.cpp",
        "rm_driver.cpp", "rm_cli.cpp", "rm_dump.cpp", "rm_pack.cpp", "rm_stream.cpp")


@pytest.fixture(scope="module")
def fault_check():
    os.makedirs(os.path.dirname(BIN), exist_ok=True)
    srcs = [os.path.join(ROOT, "tests", "hostsim", "driver_fault_check.cpp")] + [os.path.join(H, f) for f in HOST]
    newest = max(os.path.getmtime(s) for s in srcs + [os.path.join(H, "rm_driver.h"), os.path.join(H, "rm_pack.h")])
    if not os.path.exists(BIN) or os.path.getmtime(BIN) < newest:
        subprocess.run(["g++", "-O1", "-g", "-std=c++17", "-pthread", "-fsanitize=address,undefined", "-fno-omit-frame-pointer",
                        "-I" + os.path.join(ROOT, "include"), "-I" + H, "-o", BIN] + srcs, check=True)
    return BIN


@pytest.fixture(scope="module")
def packs(built, gbrna, tmp_path_factory):
    d = tmp_path_factory.mktemp("faults")
    good = str(d / "db.rmpk")
    tool = os.path.join(ROOT, "rnamotif_amd", "bin", "rnamotif_pack")
    subprocess.run([tool, good, gbrna], check=True, stderr=subprocess.DEVNULL)
    data = open(good, "rb").read()
    bad = str(d / "bad.rmpk")
    open(bad, "wb").write(data[: len(data) // 3])          # the magic is there, the arrays are not
    huge = str(d / "huge.rmpk")
    import struct
    open(huge, "wb").write(data[:8] + struct.pack("<5q", 1 << 39, 1 << 39, 1 << 39, 1 << 39, 1 << 39) + data[48:4096])
    return good, bad, huge


def _run(fault_check, workdir, files, env=None):
    e = dict(os.environ, RNAMOTIF_BATCH_BASES="40000", ASAN_OPTIONS="detect_leaks=0", **(env or {}))
    return subprocess.run([fault_check, "-descr", "sprintf.descr"] + list(files), cwd=workdir, env=e,
                          stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=600)


def test_clean_run(fault_check, workdir, packs):
    p = _run(fault_check, workdir, [packs[0]])
    assert p.returncode == 0, p.stderr.decode()[-2000:]
    assert b"AddressSanitizer" not in p.stderr and b"runtime error" not in p.stderr
    assert b"uploads 5" in p.stderr or int(p.stderr.split(b"uploads ")[1].split(b",")[0]) > 5


@pytest.mark.parametrize("second", [1, 2])
def test_a_later_file_that_is_not_a_pack(fault_check, workdir, packs, second):
    """batches of db.rmpk are still being uploaded and scanned when the next file is refused"""
    p = _run(fault_check, workdir, [packs[0], packs[second]])
    err = p.stderr.decode()
    assert p.returncode == 1, err[-2000:]
    assert "is not a packed database of this build" in err
    assert "AddressSanitizer" not in err and "runtime error" not in err, err[-3000:]


@pytest.mark.parametrize("fault", ["upload:0", "upload:3", "scan:0", "scan:2", "scan:7"])
def test_a_failing_stage(fault_check, workdir, packs, fault):
    p = _run(fault_check, workdir, [packs[0], packs[0]], env={"FAULT": fault})
    err = p.stderr.decode()
    assert p.returncode == 1, err[-2000:]
    assert "scan failed: injected" in err
    assert "AddressSanitizer" not in err and "runtime error" not in err, err[-3000:]
    # the stages stop soon after the failure: not every batch of both files is scanned
    scans = int(err.split("scans ")[1].split()[0])
    assert scans < 2 * 57
