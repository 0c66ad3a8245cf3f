"""GPU: the sharded path on the real HIP scanner with more than one rank.

The pool gives one GPU per box, and RCCL refuses two ranks on one device, so these tests run world
size 2 over gloo with BOTH ranks on device 0 (RNAMOTIF_DIST_BACKEND=gloo, RNAMOTIF_DEVICE=0): every
piece of the multi-GPU path except the transport is the product's -- reading, range partition,
rma_db_create_packed_ranges, the HIP kernels, the gather, rank 0's merge and replay.  (The CPU
tests of tests/test_distributed_gloo.py put the oracle in the scanner's place; these do not.)
Each rank is a fresh child process of torch.distributed.run; nothing re-executes a process that
has touched the GPU.  Also here: the C ABI's native gather (rma_gather_hits) with a world of one.
"""
import json
import os
import socket
import subprocess
import sys

import numpy as np
import pytest

from test_gpu_parity import _descr

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _torchrun(args, cwd, env=None, timeout=1500, nproc=2):
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    e = dict(os.environ, OMP_NUM_THREADS="1", PYTHONPATH=ROOT, EFNDATA=os.path.join(ROOT, "rnamotif_amd", "efndata"),
             RNAMOTIF_DIST_BACKEND="gloo", RNAMOTIF_DEVICE="0", **(env or {}))
    return subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={nproc}",
                           "--master-addr", "127.0.0.1", "--master-port", str(port)] + args,
                          cwd=cwd, env=e, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=timeout)


@pytest.mark.parametrize("name", ["trna.efn.descr", "pk1.descr", "score.1.descr"])
def test_mrnamotif_two_ranks_equal_rnamotif_on_gbrna(built, workdir, gbrna, tmp_path, name):
    """mrnamotif with two ranks (HIP scanner on both) prints byte for byte what bin/rnamotif prints
    over the reference's test database."""
    env = dict(os.environ, EFNDATA=os.path.join(ROOT, "rnamotif_amd", "efndata"))
    want = subprocess.run([built["cli"], "-descr", name, "gbrna.111.0.fastn"], cwd=workdir, env=env,
                          stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=900)
    assert want.returncode == 0 and want.stdout.count(b"\n>") > 5
    out = tmp_path / "hits.txt"         # (gloo prints its own notices on stdout)
    p = _torchrun(["-m", "rnamotif_amd.mrnamotif", "-descr", name, "gbrna.111.0.fastn"], workdir, env={"RNAMOTIF_OUTPUT": str(out)})
    assert p.returncode == 0, p.stderr.decode()[-3000:]
    assert out.read_bytes() == want.stdout


def test_mrnamotif_two_ranks_equal_rnamotif_on_syn10m(built, workdir, tmp_path_factory, tmp_path):
    """... and over syn10M, whose ten 1 Mbase entries are cut into slices of start positions
    between the ranks (rma_db_create_packed_ranges); 630 hits, the reference's own number."""
    import rnamotif_amd as R
    import pins
    d = tmp_path_factory.getbasetemp() / "syn10M"
    d.mkdir(exist_ok=True)
    fa = d / "syn10M.fastn"
    if not fa.exists():
        assert R.write_synthetic_fasta(str(fa), 10) == "d33c2542e515346e1d0fdfc9edcc5658"
    env = dict(os.environ, EFNDATA=os.path.join(ROOT, "rnamotif_amd", "efndata"))
    want = subprocess.run([built["cli"], "-descr", "trna.efn.descr", str(fa)], cwd=workdir, env=env,
                          stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=900)
    assert want.returncode == 0
    assert want.stdout.count(b"\n>") + want.stdout.startswith(b">") == pins.SYN10M["trna.efn.descr"]
    out = tmp_path / "hits.txt"
    p = _torchrun(["-m", "rnamotif_amd.mrnamotif", "-descr", "trna.efn.descr", str(fa)], workdir, env={"RNAMOTIF_OUTPUT": str(out)})
    assert p.returncode == 0, p.stderr.decode()[-3000:]
    assert out.read_bytes() == want.stdout


def _bench_line(p):
    assert p.returncode == 0, p.stderr.decode()[-3000:]
    lines = [l for l in p.stdout.decode().splitlines() if l.startswith("{")]
    assert len(lines) == 1, p.stdout.decode()[-2000:]
    return json.loads(lines[0])


@pytest.mark.parametrize("descrs", [["descr/trna.descr"], ["test/qu+tr.descr", "test/mp.ends.descr"]])
def test_bench_strong_scaling_two_ranks(built, descrs):
    """bench.py's multi-rank path (BASELINE config 4 with one descriptor, config 5 with two) on the HIP
    scanner: 24 records divided between two ranks find what one rank finds in all 24, the line says
    strong scaling and counts every base once per descriptor."""
    dl = ",".join(os.path.join(ROOT, "tests", "golden", f) for f in descrs)
    common = ["--steps", "2", "--warmup", "1", "--cpu-bases", "0", "--descr", dl]
    one = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--records", "24"] + common,
                         stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=1500,
                         env=dict(os.environ, PYTHONPATH=ROOT))
    single = _bench_line(one)
    two = _bench_line(_torchrun([os.path.join(ROOT, "bench.py"), "--gpus", "2", "--total-records", "24", "--backend", "gloo"] + common, ROOT))
    assert two["n_gpus"] == 2 and two["scaling"] == "strong"
    assert two["config"]["total_bases"] == 24_000_000 * len(descrs) == single["config"]["total_bases"]
    assert two["config"]["candidates"] == single["config"]["candidates"] > 0
    assert "divided among the ranks" in two["config"]["workload"]
    assert two["value"] > 0


def test_bench_default_for_several_ranks_is_config_4(built):
    """--gpus N > 1 without further flags is BASELINE config 4: 1000 records divided among the ranks
    (checked on the argument handling alone: the line of a run with --total-records says what the
    default would have been)."""
    src = open(os.path.join(ROOT, "bench.py")).read()
    assert "args.total_records = 1000" in src and "world_env > 1 and args.total_records == 0 and not args.weak" in src


def test_native_gather_world_of_one(built, workdir):
    """rma_comm_create / rma_gather_hits with one rank (RCCL is not needed for that; two ranks on one
    device are refused by RCCL, so the exchange itself runs on the driver's multi-GPU node): the
    records the scan left in HBM come back relabelled with the database-wide entry numbers."""
    import rnamotif_amd as R
    d = _descr(workdir, "trna.efn.descr")
    seqs = R.synthetic_records(3)
    sc = R.Scanner(d)
    db = sc.database(seqs)
    want = sc.scan(db)
    assert want.shape[0] > 50
    comm = R.Comm(0, 1, 0)
    sc.scan_begin(db)
    n = sc.scan_end_on_device()
    assert n == want.shape[0]
    got, counts = comm.gather(sc, [40, 7, 19])
    assert counts == [n]
    relabelled = want.copy()
    relabelled[:, 0] = np.asarray([40, 7, 19], dtype=np.int32)[want[:, 0]]
    assert np.array_equal(got, relabelled)
    # a scan that finds nothing
    empty = sc.database([b"acgt" * 50])
    sc.scan_begin(empty)
    assert sc.scan_end_on_device() == 0
    got, counts = comm.gather(sc, [0])
    assert got.shape[0] == 0 and counts == [0]
    comm.close()
