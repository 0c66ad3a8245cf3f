"""CPU, world_size 2 over gloo: sequence sharding + variable length gather of
hit records (rnamotif_amd/distributed.py) give rank 0 exactly the records, in
exactly the order, of a single-process scan.  The scan itself is done by the
oracle here (no GPU in this container); on GPUs bench.py runs the same gather
over RCCL."""
import os
import socket
import subprocess
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

WORKER = r'''
import os, sys
sys.path.insert(0, sys.argv[1]); sys.path.insert(0, os.path.join(sys.argv[1], "tests"))
import numpy as np, torch, torch.distributed as dist
import rnamotif_amd as R
from rnamotif_amd.distributed import partition_by_bases, gather_hits
from oracle_binding import oracle_scan
dist.init_process_group(backend="gloo")
rank, world = dist.get_rank(), dist.get_world_size()
d = R.Descriptor(["-descr", os.path.join(sys.argv[1], "tests", "golden", "test", "sprintf.descr")])
rng = np.random.default_rng(11)
lut = np.frombuffer(b"acgt", dtype=np.uint8)
seqs = [lut[rng.integers(0, 4, size=int(n))].tobytes() for n in (30000, 5, 12000, 0, 26000, 9000, 41000)]
parts = partition_by_bases([len(s) for s in seqs], world)
mine = parts[rank]
local = oracle_scan(d, [seqs[i] for i in mine])
allh = gather_hits(local, mine, d.hit_stride)
hparts = gather_hits(local, mine, d.hit_stride, concat=False)     # the form bench.py takes
if rank == 0:
    from rnamotif_amd.distributed import sort_hits
    assert len(hparts) == world and np.array_equal(sort_hits(np.concatenate(hparts, axis=0)), allh)
    want = oracle_scan(d, seqs)
    assert sorted(sum(parts, [])) == list(range(len(seqs)))
    assert allh.shape == want.shape and np.array_equal(allh, want), (allh.shape, want.shape)
    np.save(sys.argv[2], allh)
else:
    assert allh.shape[0] == 0
dist.barrier()
dist.destroy_process_group()
'''


def test_shard_and_gather_world2(built, tmp_path):
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    script = tmp_path / "worker.py"
    script.write_text(WORKER)
    out = tmp_path / "hits.npy"
    p = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2",
                        "--master-addr", "127.0.0.1", "--master-port", str(port), str(script), ROOT, str(out)],
                       stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=600,
                       env=dict(os.environ, OMP_NUM_THREADS="1"))
    assert p.returncode == 0, p.stdout.decode()[-4000:]
    hits = np.load(out)
    assert hits.shape[0] > 0


def test_partition_balances_and_keeps_order():
    from rnamotif_amd.distributed import partition_by_bases
    lens = [1000, 10, 900, 50, 800, 5, 700]
    parts = partition_by_bases(lens, 3)
    assert sorted(sum(parts, [])) == list(range(len(lens)))
    loads = [sum(lens[i] for i in p) for p in parts]
    assert max(loads) - min(loads) <= max(lens)
    assert all(p == sorted(p) for p in parts)


def test_partition_ranges_covers_every_start_position_once():
    from rnamotif_amd.distributed import partition_ranges
    lens = [100_000, 10, 0, 35_000, 7]
    parts = partition_ranges(lens, 4)
    cover = {i: [] for i in range(len(lens))}
    for p in parts:
        assert p == sorted(p)
        for i, lo, hi in p:
            cover[i].append((lo, hi))
    for i, n in enumerate(lens):
        spans = sorted(cover[i])
        assert spans[0][0] == 0 and spans[-1][1] == n
        assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
    loads = [sum(hi - lo for _, lo, hi in p) for p in parts]
    assert max(loads) <= 1.3 * sum(lens) / 4


RANGE_WORKER = r'''
import os, sys
sys.path.insert(0, sys.argv[1]); sys.path.insert(0, os.path.join(sys.argv[1], "tests"))
import numpy as np, torch, torch.distributed as dist
import rnamotif_amd as R
from rnamotif_amd.distributed import partition_ranges, gather_hits
from oracle_binding import oracle_scan
dist.init_process_group(backend="gloo")
rank, world = dist.get_rank(), dist.get_world_size()
d = R.Descriptor(["-descr", os.path.join(sys.argv[1], "tests", "golden", "test", "sprintf.descr")])
rng = np.random.default_rng(5)
lut = np.frombuffer(b"acgt", dtype=np.uint8)
seqs = [lut[rng.integers(0, 4, size=int(n))].tobytes() for n in (90000, 700, 20000)]
mine = partition_ranges([len(s) for s in seqs], world, max_chunk=25000)[rank]
# the slice of the start positions this rank answers for (what rma_db_create_ranges() does on
# the device): here the oracle scans the entry and the records outside the slice are dropped
idx = sorted(set(w[0] for w in mine))
h = oracle_scan(d, [seqs[i] for i in idx])
keep = np.zeros(h.shape[0], dtype=bool)
for i, lo, hi in mine:
    keep |= (h[:, 0] == idx.index(i)) & (h[:, 2] >= lo) & (h[:, 2] < hi)
allh = gather_hits(h[keep], idx, d.hit_stride)
if rank == 0:
    want = oracle_scan(d, seqs)
    assert allh.shape == want.shape and np.array_equal(allh, want), (allh.shape, want.shape)
    assert want.shape[0] > 0
dist.barrier()
dist.destroy_process_group()
'''


def test_range_sharding_world2(built, tmp_path):
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    script = tmp_path / "worker.py"
    script.write_text(RANGE_WORKER)
    p = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2",
                        "--master-addr", "127.0.0.1", "--master-port", str(port), str(script), ROOT],
                       stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=600,
                       env=dict(os.environ, OMP_NUM_THREADS="1"))
    assert p.returncode == 0, p.stdout.decode()[-4000:]


MRNAMOTIF_WORKER = r'''
import os, sys
sys.path.insert(0, sys.argv[1]); sys.path.insert(0, os.path.join(sys.argv[1], "tests"))
import numpy as np
import torch.distributed as dist
from rnamotif_amd import mrnamotif
from oracle_binding import oracle_scan

# No GPU in this container: this TEST replaces the two device-bound pieces of the module
# (process group over RCCL, the scan of a shard) to exercise everything around them --
# reading, range partition, gather, ordering, replay.  The product has no such switch.
def cpu_group(world, local_rank):
    if world > 1:
        dist.init_process_group(backend="gloo")
    return None

def cpu_scan(descr, pack, entries, ranges, local_rank):
    h = oracle_scan(descr, [pack.record(i)[2] for i in entries])
    keep = np.zeros(h.shape[0], dtype=bool)
    for k, (lo, hi) in enumerate(ranges):
        keep |= (h[:, 0] == k) & (h[:, 2] >= lo) & (h[:, 2] < hi)
    return h[keep]

mrnamotif._init_process_group = cpu_group
mrnamotif._scan_shard = cpu_scan
os.chdir(sys.argv[2])
sys.exit(mrnamotif.run(["-descr", "sprintf.descr", sys.argv[3]], out_path=sys.argv[4]))
'''


def test_mrnamotif_world2_prints_what_rnamotif_prints(built, workdir, gbrna, tmp_path):
    """The multi-process command line (one rank per GPU on a node) over gloo with the oracle
    standing in for the device scan: rank 0 prints byte for byte what the single-process
    program prints, including one long entry cut between the ranks."""
    import rnamotif_amd as R
    recs = R.read_fasta(gbrna)[:300]
    big = (b"joined", b"entries 300..899 as one", b"".join(r[2] for r in R.read_fasta(gbrna)[300:900]))
    fa = tmp_path / "db.fastn"
    fa.write_bytes(b"".join(b">" + s + b" " + d + b"\n" + q + b"\n" for s, d, q in recs + [big]))
    env = dict(os.environ, OMP_NUM_THREADS="1", EFNDATA=os.path.join(ROOT, "rnamotif_amd", "efndata"))
    want = subprocess.run([built["oracle_cli"], "-descr", "sprintf.descr", str(fa)], cwd=workdir, env=env,
                          stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=600)
    assert want.returncode == 0 and want.stdout.count(b"\n>") > 20
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    script = tmp_path / "worker.py"
    script.write_text(MRNAMOTIF_WORKER)
    out = tmp_path / "out.txt"
    p = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2",
                        "--master-addr", "127.0.0.1", "--master-port", str(port), str(script), ROOT, workdir,
                        str(fa), str(out)],
                       stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=900, env=env)
    assert p.returncode == 0, p.stdout.decode()[-4000:]
    assert out.read_bytes() == want.stdout


MIXED_WORKER = r'''
import os, sys
sys.path.insert(0, sys.argv[1]); sys.path.insert(0, os.path.join(sys.argv[1], "tests"))
import numpy as np, torch, torch.distributed as dist
import rnamotif_amd as R
from rnamotif_amd.distributed import partition_by_bases, gather_hits, sort_hits
from oracle_binding import oracle_scan
dist.init_process_group(backend="gloo")
rank, world = dist.get_rank(), dist.get_world_size()
T = os.path.join(sys.argv[1], "tests", "golden", "test")
descrs = [R.Descriptor(["-descr", os.path.join(T, n)]) for n in ("sprintf.descr", "mp.ends.descr")]
rng = np.random.default_rng(23)
lut = np.frombuffer(b"acgt", dtype=np.uint8)
# lengths that make the greedy partition interleave the ranks' entries
seqs = [lut[rng.integers(0, 4, size=int(n))].tobytes() for n in (9000, 30000, 8000, 29000, 7000, 28000, 0, 6000)]
parts = partition_by_bases([len(s) for s in seqs], world)
mine = parts[rank]
assert any(a < b < c for a in parts[0] for b in parts[1] for c in parts[0]), parts    # interleaved
# one step of a mixed batch: every descriptor over the same shard, two gathers back to back
got = []
for d in descrs:
    local = oracle_scan(d, [seqs[i] for i in mine])
    got.append(gather_hits(local, mine, d.hit_stride, concat=False))
if rank == 0:
    for d, hparts in zip(descrs, got):
        assert len(hparts) == world
        want = oracle_scan(d, seqs)
        merged = sort_hits(np.concatenate(hparts, axis=0))
        assert want.shape[0] > 0 and merged.shape == want.shape and np.array_equal(merged, want)
        # every part is in order by itself
        for h in hparts:
            assert np.array_equal(sort_hits(h), h)
else:
    assert all(g == [] for g in got)
dist.barrier()
dist.destroy_process_group()
'''


def _torchrun(script, args, tmp_path, timeout=600, env=None):
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2",
                           "--master-addr", "127.0.0.1", "--master-port", str(port), str(script)] + args,
                          stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=timeout,
                          env=dict(os.environ, OMP_NUM_THREADS="1", **(env or {})))


def test_mixed_batch_interleaved_partitions_world2(built, tmp_path):
    """concat=False with entries of the ranks interleaved, two descriptors per step (BASELINE
    config 5's shape): the per-rank parts merge into exactly the single-process hit list."""
    script = tmp_path / "worker.py"
    script.write_text(MIXED_WORKER)
    p = _torchrun(script, [ROOT], tmp_path)
    assert p.returncode == 0, p.stdout.decode()[-4000:]


FAILING_WORKER = MRNAMOTIF_WORKER.replace('''mrnamotif._scan_shard = cpu_scan''', '''def failing_scan(descr, pack, entries, ranges, local_rank):
    if int(os.environ["RANK"]) == 1:
        raise RuntimeError("no scanner on this rank")
    return cpu_scan(descr, pack, entries, ranges, local_rank)

mrnamotif._scan_shard = failing_scan''')


def test_a_failing_rank_ends_the_job_on_every_rank(built, workdir, gbrna, tmp_path):
    """One rank cannot scan: no rank is left waiting in the gather; all leave with an error."""
    import rnamotif_amd as R
    fa = tmp_path / "db.fastn"
    fa.write_bytes(b"".join(b">" + s + b" " + d + b"\n" + q + b"\n" for s, d, q in R.read_fasta(gbrna)[:60]))
    script = tmp_path / "worker.py"
    script.write_text(FAILING_WORKER)
    p = _torchrun(script, [ROOT, workdir, str(fa), str(tmp_path / "out.txt")], tmp_path, timeout=300,
                  env={"EFNDATA": os.path.join(ROOT, "rnamotif_amd", "efndata")})
    assert p.returncode != 0
    assert b"no scanner on this rank" in p.stdout


def test_mrnamotif_reads_what_rnamotif_reads(built, workdir, gbrna, tmp_path):
    """The multi-process program reads its database with the library's readers: -N truncation, a
    '>' in the middle of a line, a PIR file -- rank 0 prints what the single-process program prints."""
    import rnamotif_amd as R
    recs = R.read_fasta(gbrna)[:200]
    fa = tmp_path / "db.fastn"
    body = b"".join(b">" + s + b" " + d + b"\n" + q + b"\n" for s, d, q in recs[:150])
    body += recs[150][2][:40] + b">midline entry\n" + recs[151][2] + b"\n"
    body += b"".join(b">" + s + b" " + d + b"\n" + q + b"\n" for s, d, q in recs[152:])
    fa.write_bytes(body)
    env = dict(os.environ, OMP_NUM_THREADS="1", EFNDATA=os.path.join(ROOT, "rnamotif_amd", "efndata"))
    script = tmp_path / "worker.py"
    for extra, path in ((["-N", "400"], fa),):
        want = subprocess.run([built["oracle_cli"], "-descr", "sprintf.descr"] + extra + [str(path)], cwd=workdir, env=env,
                              stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=600)
        assert want.returncode == 0 and want.stdout.count(b"\n>") > 10
        script.write_text(MRNAMOTIF_WORKER.replace('["-descr", "sprintf.descr", sys.argv[3]]',
                                                   '["-descr", "sprintf.descr"] + %r + [sys.argv[3]]' % (extra,)))
        out = tmp_path / "out.txt"
        p = _torchrun(script, [ROOT, workdir, str(path), str(out)], tmp_path, env={"EFNDATA": env["EFNDATA"]})
        assert p.returncode == 0, p.stdout.decode()[-4000:]
        assert out.read_bytes() == want.stdout


def test_mrnamotif_rounds_and_rank0_reads_what_it_prints(built, workdir, gbrna, tmp_path):
    """Shares in several rounds (the next round's entries read while this one is scanned and gathered), every rank --
    the first too -- reading its own entries only, and rank 0 at the end the entries that have hits: byte for byte
    what the single-process program prints; the laps say who read what."""
    import re
    import rnamotif_amd as R
    recs = R.read_fasta(gbrna)[:1500]
    fa = tmp_path / "db.fastn"
    fa.write_bytes(b"".join(b">" + s + b" " + d + b"\n" + q + b"\n" for s, d, q in recs))
    env = dict(os.environ, OMP_NUM_THREADS="1", EFNDATA=os.path.join(ROOT, "rnamotif_amd", "efndata"))
    want = subprocess.run([built["oracle_cli"], "-descr", "sprintf.descr", str(fa)], cwd=workdir, env=env,
                          stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=600)
    assert want.returncode == 0 and want.stdout.count(b"\n>") > 10
    script = tmp_path / "worker.py"
    script.write_text(MRNAMOTIF_WORKER)
    out = tmp_path / "out.txt"
    p = _torchrun(script, [ROOT, workdir, str(fa), str(out)], tmp_path,
                  env={"EFNDATA": env["EFNDATA"], "RNAMOTIF_BATCH_BASES": "120000", "RNAMOTIF_TIMING": "1"})
    assert p.returncode == 0, p.stdout.decode()[-4000:]
    assert out.read_bytes() == want.stdout
    log = p.stdout.decode()
    per_rank = {r: [int(m) for m in re.findall(r"\[timing\] rank %d read round \d+: (\d+) entries" % r, log)] for r in (0, 1)}
    assert len(per_rank[0]) == len(per_rank[1]) >= 3                 # the same number of rounds on every rank
    assert sum(per_rank[0]) + sum(per_rank[1]) == len(recs)            # every entry read once, by the rank that scans it
    assert abs(sum(per_rank[0]) - sum(per_rank[1])) < len(recs) // 3   # (rank 0 reads its share, not the database)
    with_hits = int(re.search(r"rank 0 read (\d+) entries with hits", log).group(1))
    assert 0 < with_hits < len(recs)                                   # (entries with candidates: the score program rejects most)
    assert "the whole database" not in log
