"""CPU: the device search state machine (rm_scan_core.h, the code the HIP kernel
runs per lane, including the pre-filter decomposition) compiled for the host and
compared with the oracle record by record on the reference's test database."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BIN = os.path.join(ROOT, "tests", "_build", "hostsim_check")
H = os.path.join(ROOT, "rnamotif_amd", "csrc")


@pytest.fixture(scope="module")
def hostsim():
    os.makedirs(os.path.dirname(BIN), exist_ok=True)
    srcs = [os.path.join(ROOT, "tests", "hostsim", "hostsim_check.cpp")]
    srcs += [os.path.join(H, f) for f in ("rm_regex.cpp", "rm_compile.cpp", "rm_parse.cpp", "rm_score.cpp",
                                          "rm_efndata.cpp", "rm_efn2data.cpp", "rm_fasta.cpp", "rm_driver.cpp", "rm_cli.cpp", "rm_dump.cpp", "rm_pack.cpp", "rm_stream.cpp",
                                          "rm_dev_program.cpp")]
    srcs += [os.path.join(ROOT, "oracle", f) for f in ("rm_oracle_scan.c", "rm_oracle_efn.c", "rm_oracle_efn2.c")]
    newest = max(os.path.getmtime(s) for s in srcs + [os.path.join(H, f) for f in ("rm_scan_core.h", "rm_efn_core.h", "rm_efn2_core.h", "rm_dev_program.h")])
    if not os.path.exists(BIN) or os.path.getmtime(BIN) < newest:
        subprocess.run(["g++", "-O2", "-std=c++17", "-pthread", "-I" + os.path.join(ROOT, "include"), "-I" + H,
                        "-I" + os.path.join(ROOT, "oracle"), "-o", BIN] + srcs + ["-lm"], check=True)
    return BIN


CASES = [(["-descr", "trna.descr"], 1351), (["-descr", "mp.ends.descr"], 580), (["-descr", "qu+tr.descr"], 9),
         (["-descr", "pk_j1+2.descr"], 32), (["-descr", "bulge.descr"], 686), (["-descr", "nanlin.descr"], 13),
         (["-sh", "-context", "-Dctx_maxlen=5", "-descr", "qu+tr.strict.descr"], 9),
         (["-sh", "-context", "-Dctx_maxlen=5", "-descr", "trna.strict.descr"], 184)]
# (pk1 with -sh takes 25 s here; it is pinned through the oracle in test_golden_stdout.py and
# compared on the GPU in test_gpu_parity.py)


@pytest.mark.parametrize("args,ncand", CASES, ids=[" ".join(c[0][-1:]) for c in CASES])
def test_state_machine_equals_oracle(hostsim, workdir, args, ncand):
    p = subprocess.run([hostsim] + args + ["gbrna.111.0.fastn"], cwd=workdir, stdout=subprocess.PIPE,
                       stderr=subprocess.PIPE, timeout=1800)
    assert p.returncode == 0, p.stdout.decode() + p.stderr.decode()
    assert (b"%d candidates, 0 mismatching strands" % ncand) in p.stdout


def test_efn2_device_core_equals_oracle(hostsim, workdir):
    """rm_efn2_core.h compiled for the host: the energy of every candidate of a cloverleaf
    search (multi-branch loop, coaxial stacking) equals the oracle's restatement of RM_efn2."""
    descr = os.path.join(ROOT, "tests", "data", "trna.efn2.descr")
    env = dict(os.environ, EFNDATA=os.path.join(ROOT, "rnamotif_amd", "efndata"))
    p = subprocess.run([hostsim, "-descr", descr, "gbrna.111.0.fastn"], cwd=workdir, env=env, stdout=subprocess.PIPE,
                       stderr=subprocess.PIPE, timeout=1800)
    assert p.returncode == 0, p.stdout.decode() + p.stderr.decode()
    assert b"1351 candidates, 0 mismatching strands (1351 efn2 energies compared)" in p.stdout


def test_efn_device_core_equals_oracle(hostsim, workdir, gbrna, tmp_path):
    """rm_efn_core.h compiled for the host, reading the tables as the kernel gets them (rma::efn_tables16), with and
    without the per-lane cache of codes and partners: efn() of every cloverleaf candidate, and of hairpins paired
    by the descriptor's own pair set (efn_usestdbp = 0, g:a pairs: RM_efn's "knot" returns, efn.c:1218,1262, drop
    what the call had added up -- round 4 found the device core keeping it)."""
    env = dict(os.environ, EFNDATA=os.path.join(ROOT, "rnamotif_amd", "efndata"))
    p = subprocess.run([hostsim, "-descr", "trna.efn.descr", "gbrna.111.0.fastn"], cwd=workdir, env=env, stdout=subprocess.PIPE,
                       stderr=subprocess.PIPE, timeout=1800)
    assert p.returncode == 0, p.stdout.decode() + p.stderr.decode()
    assert b"1351 candidates, 0 mismatching strands (1351 efn energies compared)" in p.stdout
    small = tmp_path / "small.fastn"
    with open(gbrna, "rb") as f:
        small.write_bytes(b"".join(f.readlines()[:1500]))
    descr = os.path.join(ROOT, "tests", "data", "hairpin.nostdbp.descr")
    p = subprocess.run([hostsim, "-descr", descr, str(small)], cwd=workdir, env=env, stdout=subprocess.PIPE,
                       stderr=subprocess.PIPE, timeout=1800)
    assert p.returncode == 0, p.stdout.decode() + p.stderr.decode()
    assert b" 0 mismatching strands" in p.stdout and b"efn energies compared" in p.stdout


def test_generated_descriptors(hostsim, tmp_path):
    """The generators of tests/test_gpu_parity.py on the CPU: the device state machine (lean and
    general paths), compiled for the host, against the oracle on 30 + 30 generated descriptors."""
    import numpy as np
    import rnamotif_amd as R
    import test_gpu_parity as T
    ran = 0
    for kind, gen, base, n in (("lean", T._random_descriptor, 1000, 30), ("general", T._random_general_descriptor, 5000, 30)):
        for seed in range(n):
            rng = np.random.default_rng(base + seed)
            path = tmp_path / ("%s_%d.descr" % (kind, seed))
            path.write_text(gen(rng))
            try:
                d = R.Descriptor(["-descr", str(path)])
            except R.RnamotifError:
                continue
            if d.maxlen > 160:
                continue
            s = T._planted_sequence(rng, 4_000)
            fa = tmp_path / "db.fastn"
            fa.write_bytes(b">a x\n" + s + b"\n>b y\n" + s[:301] + b"\n")
            p = subprocess.run([hostsim, "-descr", str(path), str(fa)], stdout=subprocess.PIPE, stderr=subprocess.PIPE,
                               timeout=120)
            assert p.returncode == 0 and b" 0 mismatching strands" in p.stdout, (kind, seed, p.stdout[-300:], p.stderr[-300:])
            ran += 1
    assert ran >= 40
