#!/usr/bin/env python3
"""bench.py -- headline benchmark of the scan path (BASELINE.json).

One step = one pass of the hot path over one batch of synthetic database that is in HBM when the timed
region starts: search kernel over every start position of both strands (+ the drain kernel behind it), efn
kernel over every candidate, ordering and copy back of the hit records.  (Round 4: until round 3 `value` was
SURVEY.md section 8d's step, with the upload of the packed batch -- 0.375 B/base from page-locked host memory on
the device's upload stream, under the kernels of the step before -- inside it; that PCIe-inclusive rate is now
the `h2d_inclusive` leg, and `value` under --h2d.)

Workload at N=1: descr/trna.descr (4-stem cloverleaf, bits()+efn() score) over the 100 Mbase
synthetic FASTA of BASELINE.md (100 records x 1 Mbase, iid uniform acgt, numpy default_rng(20240601))
-- BASELINE config 2.  `value` is that, over exactly --steps steps.

Next to it, on rank 0 at N=1 with the default workload (none of it inside the timed K steps):
  h2d_inclusive    the same step with the upload of its batch inside
  real_db          the reference's own test database x 44 (100 Mbase in 179 k short entries): trna / pk1 / mp.ends,
                   kernel ms as short entries and as the same bases in long entries, candidates cross-checked
  sustained        the headline step repeated for at least a second
  north_star_1gbase  the north star's own size: trna.descr over 1 Gbase on one GPU, >= 1 s timed
  cli_end_to_end   the whole command line program (bin/rnamotif -descr trna.descr) over 1 Gbase from a
                   text file and from a packed database: search time and whole-process time
  roofline         HBM roofline of the search kernel from HIP events taken here; its two passes
                   apart (pre-filter alone, option dbg=1); HBM traffic and the VALU-issue
                   secondary roofline from the committed rocprofv3 PMC summary -- only if that
                   summary was made from the kernel sources this run uses (hash), else null
  cpu_baseline     the scalar oracle on a bounded sample, one core (and all host cores)

With --gpus N > 1 (launched by torch.distributed.run, one rank per GPU) the default is BASELINE
config 4: 1000 records (1 Gbase) divided among the ranks, strong scaling; --weak gives every rank
its own --records Mbase instead.  --descr qu+tr.descr,mp.ends.descr is config 5 (both descriptors
over one upload of the shard, their kernels side by side on two streams).  The hit records travel
to rank 0 inside the timed region -- the path's only exchange: rma_gather_hits() of the C ABI
(RCCL all-gather of the counts + grouped send/receive, device to device), or, if that cannot be
set up, the same over torch.distributed.

Prints ONE JSON line on rank 0.
"""
import argparse
import hashlib
import json
import os
import subprocess
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

ALGO_BYTES_PER_BASE = 0.375      # 2 bit code + 1 bit ambiguity mask, read once for both strands
HBM_PEAK_GBPS = 8000.0           # MI355X_MICROARCH.md: 8.0 TB/s spec
# MI355X_MICROARCH.md: a wave64 VALU instruction issues over 2 cycles on a SIMD-32 once two or more
# waves share the SIMD: 256 CUs x 4 SIMDs x 2.4 GHz / 2
VALU_PEAK_NOMINAL = 256 * 4 * 2.4e9 / 2 / 1e9
KERNEL_SOURCES = ("rm_scan_kernel.h", "rm_scan_core.h", "rm_dev_program.h", "rm_dev_program.cpp", "rm_efn_core.h", "rm_kernels.h")
PROFILE_ROUND = "r04"
PROFILE = os.path.join(ROOT, "profiles", PROFILE_ROUND + "_trna")     # _pmc_summary.csv, _meta.json (profiles/collect.sh + summarize.py)
SEED = 20240601


def kernel_hash():
    h = hashlib.sha256()
    for f in KERNEL_SOURCES:
        with open(os.path.join(ROOT, "rnamotif_amd", "csrc", f), "rb") as fh:
            h.update(fh.read())
    return h.hexdigest()[:16]


def synthetic_slice(first: int, count: int, length: int):
    """Records [first, first+count) of the synthetic stream (numpy default_rng(20240601), record k =
    the k-th draw of `length` integers).  A record of even length takes length/2 steps of the PCG64
    stream (two 32-bit draws per step), so a rank jumps to its first record instead of generating
    everything before it."""
    import numpy as np
    lut = np.frombuffer(b"acgt", dtype=np.uint8)
    bg = np.random.PCG64(SEED)
    rng = np.random.Generator(bg)
    skip = 0
    if length % 2 == 0:
        bg.advance(first * (length // 2))
    else:
        skip = first
    out = []
    for k in range(skip + count):
        v = rng.integers(0, 4, size=length)
        if k >= skip:
            out.append(lut[v].tobytes())
    return out


def profile_counters(kernels=("rma_search_kernel", "rma_drain_kernel")):
    """Per-launch means of the committed PMC passes over the default workload -- or None when the
    summary was made from other kernel sources than the ones this run uses."""
    import csv
    meta, summ = PROFILE + "_meta.json", PROFILE + "_pmc_summary.csv"
    if not (os.path.exists(meta) and os.path.exists(summ)):
        return None
    if json.load(open(meta)).get("kernel_hash") != kernel_hash():
        return None
    # (one launch of each per scan: the search kernel and the drain kernel that walks what it left in its list)
    c = {}
    for r in csv.DictReader(open(summ)):
        if any(k in r["kernel"] for k in kernels):
            c[r["counter"]] = c.get(r["counter"], 0.0) + float(r["mean_per_dispatch"])
    return c or None


def cpu_model():
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def cpu_baseline(descr, seqs, budget_bases):
    """The scalar CPU oracle (kind 'port': byte-identical to the reference on its golden tests, and
    within a few percent of its speed where both ran) on a bounded sample of the same workload, one thread."""
    from oracle_binding import oracle_scan
    sample, got = [], 0
    for s in seqs:
        if got >= budget_bases:
            break
        take = s[: budget_bases - got]
        sample.append(take)
        got += len(take)
    t0 = time.perf_counter()
    hits = oracle_scan(descr, sample)
    dt = time.perf_counter() - t0
    return {"value": round(got / dt / 1e6, 4), "unit": "Mbases/s", "cores": 1, "kind": "port", "cpu": cpu_model(),
            "sample": f"first {got} bases of the same database, both strands, {hits.shape[0]} candidates, {dt:.1f} s",
            "seconds": round(dt, 2)}


def _cpu_worker(job):
    """One host core's share of the all-cores CPU baseline (spawned before the GPU is touched)."""
    descr_path, sample = job
    import rnamotif_amd as R
    from oracle_binding import oracle_scan
    os.environ.setdefault("EFNDATA", R.EFNDATA_DIR)
    d = R.Descriptor(["-descr", descr_path])
    t0 = time.perf_counter()
    hits = oracle_scan(d, [sample])
    return len(sample), hits.shape[0], time.perf_counter() - t0


def cpu_baseline_all_cores(descr_path, seqs, bases_per_core):
    """The reference's own way to use more cores is one process per database file (mrnamotif); the
    same here: every host core scans its own slice of the database."""
    import multiprocessing as mp
    from oracle_pool import host_cores
    # every core this process may use: the affinity mask, cut to the cgroup's CPU quota (a GPU box of this pool shows 256
    # cores of its two EPYC 9575F and grants the job 16 of them, /sys/fs/cgroup/cpu.max)
    cores = min(host_cores(), len(seqs))
    jobs = [(descr_path, seqs[k][:bases_per_core]) for k in range(cores)]
    t0 = time.perf_counter()
    with mp.get_context("spawn").Pool(cores) as pool:
        res = pool.map(_cpu_worker, jobs)
    wall = time.perf_counter() - t0
    busy = max(r[2] for r in res)
    total = sum(r[0] for r in res)
    return {"value": round(total / busy / 1e6, 4), "unit": "Mbases/s", "cores": cores, "kind": "port", "cpu": cpu_model(),
            "host_cores_visible": len(os.sched_getaffinity(0)), "host_cores_granted": host_cores(),
            "sample": f"{cores} processes x first {bases_per_core} bases of records 0..{cores - 1}, both strands, "
                      f"{sum(r[1] for r in res)} candidates, slowest process {busy:.1f} s (wall {wall:.1f} s with start-up)"}


def cli_end_to_end(descr_path, seqs, tmp, threads_env=None):
    """The whole command line program over `seqs` written as a FASTA text file and as a packed
    database: per input the best of two runs -- `search` = from the scanner's creation to the last
    line printed (reading or loading, packing, upload, kernels, score program, printing; the laps
    RNAMOTIF_TIMING prints), `process` = the whole program by this process's clock (HIP start-up,
    descriptor compilation and exit included)."""
    import numpy as np
    import rnamotif_amd as R
    fa, pk = os.path.join(tmp, "syn.fastn"), os.path.join(tmp, "syn.rmpk")
    with open(fa, "wb") as f:
        for i, s in enumerate(seqs):
            f.write(b">syn%04d synthetic uniform ACGT seed=%d len=%d\n" % (i, SEED, len(s)))
            a = np.frombuffer(s, dtype=np.uint8)
            full = (len(a) // 50) * 50
            f.write(np.concatenate([a[:full].reshape(-1, 50), np.full((full // 50, 1), 10, dtype=np.uint8)], axis=1).tobytes())
            if full < len(a):
                f.write(a[full:].tobytes() + b"\n")
    R.Pack.write(pk, [(b"syn%04d" % i, b"synthetic uniform ACGT seed=%d len=%d" % (SEED, len(s)), s) for i, s in enumerate(seqs)])
    bases = sum(len(s) for s in seqs)
    env = dict(os.environ, RNAMOTIF_TIMING="1", EFNDATA=R.EFNDATA_DIR)
    env.pop("RNAMOTIF_NO_WARMUP", None)
    out = {"bases": bases, "command": "rnamotif_amd/bin/rnamotif -descr " + os.path.basename(descr_path) + " <file>",
           "what": "search = scanner created .. last line printed; process = the whole program, HIP start-up included; best of two runs"}
    ref_stdout = None
    for what, path in (("text", fa), ("pack", pk)):
        best = None
        for _ in range(2):
            t0 = time.perf_counter()
            p = subprocess.run([R.CLI_PATH, "-descr", descr_path, path], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE)
            wall = time.perf_counter() - t0
            if p.returncode != 0:
                out[what] = {"error": p.stderr.decode("utf-8", "replace")[-300:]}
                best = None
                break
            laps = {}
            for line in p.stderr.decode("utf-8", "replace").splitlines():
                if line.startswith("[timing]") and " at " in line and line.rstrip().endswith("ms"):
                    name, at = line[len("[timing]"):].rsplit(" at ", 1)
                    laps[name.strip()] = float(at.split()[0])
            search_ms = laps.get("search done", float("nan")) - laps.get("scanner created", float("nan"))
            hits = p.stdout.count(b"\n>")
            if ref_stdout is None:
                ref_stdout = hashlib.md5(p.stdout).hexdigest()
            same = hashlib.md5(p.stdout).hexdigest() == ref_stdout
            cur = {"search_ms": round(search_ms, 1), "search_mbases_per_s": round(bases / search_ms / 1e3, 1),
                   "process_ms": round(wall * 1e3, 1), "process_mbases_per_s": round(bases / wall / 1e6, 1),
                   "hits_printed": hits, "stdout_identical_to_first_run": same}
            if best is None or cur["search_ms"] < best["search_ms"]:
                best = cur
        if best is not None:
            out[what] = best
    for f in (fa, pk):
        try:
            os.unlink(f)
        except OSError:
            pass
    return out


REAL_DB = os.path.join(ROOT, "tests", "golden", "test", "gbrna.111.0.fastn.gz")
REAL_DB_COPIES = 44
# candidates (before the score program) of one copy of the reference's test database: the oracle's counts, which
# tests/test_gpu_parity.py::test_hit_records_equal_oracle and tests/test_hostsim.py hold the GPU records to
REAL_DB_CANDIDATES = {"trna.descr": 1351, "mp.ends.descr": 580, "pk1.descr": 193}


def real_db_leg(dev_index, names=("trna.descr", "pk1.descr", "mp.ends.descr")):
    """The reference's own test database (test/gbrna.111.0.fastn: GenBank RNA entries, 4 067 of them, 560 bases on average,
    2.7 % n) 44 times over = 100 Mbase in 179 k SHORT entries -- what real RNA databases look like, and the filters'
    hard case next to iid uniform sequence (tRNA and rRNA genes cluster the pre-filter's survivors).  Per descriptor:
    kernel ms and Mbases/s over the entries as they are, over the same bases joined into 44 long entries, and the
    candidates, which must be 44 times one copy's (= the oracle's count of one copy)."""
    import numpy as np
    import rnamotif_amd as R
    recs = R.read_fasta(REAL_DB)
    one = [r[2] for r in recs]
    short = one * REAL_DB_COPIES
    joined = b"".join(one)
    long_ = [joined] * REAL_DB_COPIES
    bases = sum(len(s) for s in short)
    out = {"database": "tests/golden/test/gbrna.111.0.fastn x %d" % REAL_DB_COPIES, "entries": len(short), "bases": bases,
           "mean_entry_bases": round(bases / len(short), 1), "descriptors": {}}
    for name in names:
        path = os.path.join(ROOT, "tests", "golden", "descr" if name == "trna.descr" else "test", name)
        d = R.Descriptor(["-descr", path])
        sc = R.Scanner(d, device=dev_index)
        res = {}
        n_one = sc.scan(sc.database(one), copy=False).shape[0]
        for what, seqs in (("short_entries", short), ("long_entries", long_)):
            db = sc.database(seqs)
            sc.scan_device(db)
            ms, n = [], 0
            for _ in range(5):
                n, s_ms, _e = sc.scan_device(db)
                k = sc.last_kernel_ms()
                ms.append(k[0] + k[1])
            db.close()
            res[what] = {"kernel_ms": round(float(np.mean(ms)), 3), "mbases_per_s": round(bases / (float(np.mean(ms)) * 1e-3) / 1e6, 1), "candidates": int(n)}
        res["short_over_long"] = round(res["short_entries"]["kernel_ms"] / res["long_entries"]["kernel_ms"], 3)
        res["candidates_one_copy"] = int(n_one)
        res["candidates_one_copy_oracle"] = REAL_DB_CANDIDATES.get(name)
        res["candidates_ok"] = bool(res["short_entries"]["candidates"] == REAL_DB_COPIES * n_one and
                                    (REAL_DB_CANDIDATES.get(name) is None or REAL_DB_CANDIDATES[name] == n_one))
        out["descriptors"][name] = res
        sc.close()
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--records", type=int, default=100, help="1 Mbase records per GPU (default 100 = 100 Mbase; with --gpus N > 1 only under --weak)")
    ap.add_argument("--total-records", type=int, default=0,
                    help="strong scaling: this many records divided among the ranks (default with --gpus N > 1: 1000 = BASELINE config 4)")
    ap.add_argument("--weak", action="store_true", help="with --gpus N > 1: every rank scans its own --records records")
    ap.add_argument("--record-len", type=int, default=1_000_000)
    ap.add_argument("--descr", default=os.path.join(ROOT, "tests", "golden", "descr", "trna.descr"),
                    help="descriptor file; a comma separated list = mixed batch (every descriptor over the same database)")
    ap.add_argument("--cpu-bases", type=int, default=12_000_000, help="sample size of the CPU baseline (0 = skip it and the other extras)")
    ap.add_argument("--north-star-records", type=int, default=1000, help="records of the 1-GPU north star run and of the command line leg (0 = skip)")
    ap.add_argument("--resident", action="store_true", help="(kept for old command lines: the database staying in HBM is the default since round 4)")
    ap.add_argument("--h2d", action="store_true", help="every step uploads its batch (SURVEY.md 8d's step): then `value` is the PCIe-inclusive rate")
    ap.add_argument("--no-real-db", action="store_true", help="skip the real_db leg")
    ap.add_argument("--serial", action="store_true", help="one scanner per descriptor: every step is ended before the next begins")
    ap.add_argument("--gather", choices=("native", "torch"), default="native", help="N > 1: rma_gather_hits (RCCL behind the C ABI) or torch.distributed")
    ap.add_argument("--backend", default=os.environ.get("RNAMOTIF_DIST_BACKEND", "nccl"),
                    help="torch.distributed backend for N > 1 (nccl = RCCL; gloo: tests with several ranks on one GPU)")
    args = ap.parse_args()
    # `value` is measured with the inputs already in HBM when the timed region starts (the round's measurement rule); the
    # PCIe-inclusive step of SURVEY.md 8d stands beside it as `h2d_inclusive` (and is `value` only under --h2d)
    args.resident = not args.h2d

    world_env = int(os.environ.get("WORLD_SIZE", "1"))
    rank_env = int(os.environ.get("RANK", "0"))
    if world_env > 1 and args.total_records == 0 and not args.weak:
        args.total_records = 1000           # BASELINE config 4 / 5: 1 Gbase divided among the ranks
    default_workload = (args.descr == ap.get_default("descr") and args.records == 100 and
                        args.record_len == 1_000_000 and args.total_records == 0 and args.resident)
    extras = world_env == 1 and args.cpu_bases > 0 and "," not in args.descr
    strong = args.total_records > 0
    if strong:
        per = -(-args.total_records // world_env)
        first_rec = min(args.total_records, rank_env * per)
        n_rec = max(0, min(args.total_records, first_rec + per) - first_rec)
    else:
        first_rec, n_rec = rank_env * args.records, args.records

    seqs_all = None
    cpu_all = None
    if extras:
        # host-only work first: worker processes are spawned before this process touches the GPU
        n_gen = max(n_rec, args.north_star_records if default_workload else 0)
        seqs_all = synthetic_slice(0, n_gen, args.record_len)
        cpu_all = cpu_baseline_all_cores(args.descr, seqs_all, min(args.cpu_bases // 2, args.record_len))

    import numpy as np
    import torch
    import rnamotif_amd as R

    world, rank = world_env, rank_env
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    # (tests run several ranks on one GPU: RNAMOTIF_DEVICE names it)
    dev_index = int(os.environ.get("RNAMOTIF_DEVICE", local_rank))
    dist = None
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the scan path has no CPU implementation")
    dev = torch.device("cuda", dev_index)
    if world > 1:
        import datetime
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        torch.cuda.set_device(dev_index)
        if args.backend == "nccl":
            dist.init_process_group(backend="nccl", device_id=dev, timeout=datetime.timedelta(minutes=10))
        else:
            dist.init_process_group(backend=args.backend, timeout=datetime.timedelta(minutes=10))
    coll_dev = dev if (world > 1 and args.backend == "nccl") else torch.device("cpu")

    os.environ.setdefault("EFNDATA", R.EFNDATA_DIR)
    descr_files = args.descr.split(",")
    descrs = [R.Descriptor(["-descr", f]) for f in descr_files]
    descr = descrs[0]
    seqs = seqs_all[:n_rec] if seqs_all is not None else synthetic_slice(first_rec, n_rec, args.record_len)
    scs = [R.Scanner(d, device=dev_index) for d in descrs]
    sc = scs[0]
    tmpdir = tempfile.mkdtemp(prefix="rnamotif_bench_")
    # the rank's share as a packed database in page-locked host memory: what a step uploads
    pkpath = os.path.join(tmpdir, "shard%d.rmpk" % rank)
    R.Pack.write(pkpath, [(b"syn%04d" % (first_rec + i), b"", s) for i, s in enumerate(seqs)])
    pack = R.Pack(pkpath)
    os.unlink(pkpath)
    pack.pin()

    from rnamotif_amd.distributed import gather_hits as torch_gather, NativeGather
    my_index = [first_rec + i for i in range(n_rec)]   # entry numbers within the whole job
    native = None
    gather_kind = "none (1 rank)"
    if world > 1:
        gather_kind = "torch.distributed " + args.backend
        if args.gather == "native" and args.backend == "nccl":
            try:
                native = NativeGather(rank, world, dev_index, coll_dev)
                gather_kind = "rma_gather_hits (RCCL all-gather of counts + grouped send/recv, device to device)"
            except Exception as e:      # noqa: BLE001 -- the same exchange over torch.distributed then
                sys.stderr.write(f"bench.py (rank {rank}): native gather not available ({e}); using torch.distributed\n")
                native = None
        # every rank must take the same path
        from rnamotif_amd.distributed import all_ok
        if not all_ok(native is not None, coll_dev):
            native = None
            if args.gather == "native" and args.backend == "nccl":
                gather_kind = "torch.distributed " + args.backend + " (native gather refused on some rank)"

    def new_db(wait):
        return sc.database_from_pack(pack, wait=wait)

    # Two scanners per descriptor, taking the steps in turns: the kernels of step i + 1 are launched before step i is
    # ended, so the drain kernel, the energies, the ordering and the copy back of step i run under the search kernel of
    # step i + 1 (each scanner has its own stream, hit buffer and lists; rma_scan_begin / rma_scan_end of the C ABI).
    # A step is still one pass of the whole path over one batch, and every step begun inside the timed region is ended
    # inside it; --serial ends every step before the next begins (the rate is in the `serial_steps` leg either way).
    sets = [scs] if args.serial else [scs, [R.Scanner(d, device=dev_index) for d in descrs]]
    pipe = {"i": 0, "pending": None}

    def scan_begin(db, k=0):
        """Every descriptor's search kernel over db on its way, side by side on the scanners' streams."""
        for s_ in sets[k]:
            s_.scan_begin(db)

    def scan_end(k=0):
        """Energies, ordering, the records of all descriptors on rank 0; their number."""
        n = 0
        for d_, s_ in zip(descrs, sets[k]):
            if world == 1:
                n += s_.scan_end(copy=False).shape[0]
            elif native is not None:
                s_.scan_end_on_device()
                h, _ = native.gather(s_, my_index)
                n += h.shape[0]
            else:
                # every rank holds a consecutive run of entries: the per-rank arrays in rank order are the
                # ordered hit stream of the whole job, left as they arrive (no concatenation on rank 0)
                n += sum(part.shape[0] for part in torch_gather(s_.scan_end(copy=False), my_index, d_.hit_stride,
                                                                device=coll_dev, concat=False))
        return n

    state = {"cur": new_db(True)}

    def flush():
        """The step still in flight is ended; its candidates."""
        n = 0
        if pipe["pending"] is not None:
            k, old_db = pipe["pending"]
            n = scan_end(k)
            if old_db is not None:
                old_db.close()
            pipe["pending"] = None
        return n

    def step_h2d():
        """The kernels of this batch are launched, then the next batch's upload is put on the upload stream
        (its host side runs under the kernels), then the batch before this one is finished; its block of HBM
        goes back to the pool."""
        k = pipe["i"] % len(sets)
        pipe["i"] += 1
        scan_begin(state["cur"], k)
        nxt = new_db(False)
        n = flush() if len(sets) > 1 else 0
        pipe["pending"] = (k, state["cur"])
        state["cur"] = nxt
        return flush() if len(sets) == 1 else n

    def step_resident():
        k = pipe["i"] % len(sets)
        pipe["i"] += 1
        scan_begin(state["cur"], k)
        n = flush() if len(sets) > 1 else 0
        pipe["pending"] = (k, None)
        return flush() if len(sets) == 1 else n

    step = step_resident if args.resident else step_h2d

    for _ in range(args.warmup):
        step()
    flush()

    def fence():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    fence()
    t0 = time.perf_counter()
    total_hits = 0
    for _ in range(args.steps):
        step()
    total_hits = flush()                # (the last step is ended inside the timed region)
    state["cur"].wait()                 # (the upload the last step started is part of it)
    fence()
    dt = time.perf_counter() - t0
    db = state["cur"]
    if world > 1:
        t = torch.tensor([dt], dtype=torch.float64, device=coll_dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
        tb = torch.tensor([db.bases], dtype=torch.int64, device=coll_dev)
        dist.all_reduce(tb, op=dist.ReduceOp.SUM)
        job_bases = int(tb.item())
    else:
        job_bases = db.bases

    # kernel time of the dominant kernel, HIP events on the scanner's own stream; then the pre-filter alone
    parts = {"rma_search_kernel": [], "rma_drain_kernel": []}

    def kernel_ms(reps=5, keep_parts=False):
        s_tot, e_tot = [], []
        for _ in range(reps):
            tot = [0.0, 0.0]
            p0 = p1 = 0.0
            for sc_ in scs:
                _, s_ms, e_ms = sc_.scan_device(db)
                tot[0] += s_ms
                tot[1] += e_ms
                k_ms = sc_.last_kernel_ms()
                p0 += k_ms[0]
                p1 += k_ms[1]
            s_tot.append(tot[0])
            e_tot.append(tot[1])
            if keep_parts:
                parts["rma_search_kernel"].append(p0)
                parts["rma_drain_kernel"].append(p1)
        return float(np.mean(s_tot)), float(np.mean(e_tot))

    search_ms, efn_ms = kernel_ms(keep_parts=True)
    drain_ms = float(np.mean(parts["rma_drain_kernel"]))
    pass_a_ms = None
    if rank == 0 and world == 1 and args.cpu_bases > 0:       # (not under the profiler: profiles/collect.sh passes --cpu-bases 0,
        # so that every launch of the kernel it counts is a whole one)
        for s_ in scs:
            s_.set_option("dbg", 1)
        devnull = os.open(os.devnull, os.O_WRONLY)
        saved = os.dup(2)
        os.dup2(devnull, 2)             # (the diagnostic lines of dbg launches are not this run's output)
        try:
            pass_a_ms, _ = kernel_ms(3)
        finally:
            os.dup2(saved, 2)
            os.close(devnull)
            os.close(saved)
        for s_ in scs:
            s_.set_option("dbg", 0)

    # what RCCL itself says the job is (ncclCommCount of the native gather's communicator), and every rank's kernel time
    rccl_ranks = native.comm_count() if native is not None else None
    per_rank_kernel_ms = None
    if world > 1:
        t = torch.zeros(world, dtype=torch.float64, device=coll_dev)
        t[rank] = search_ms
        dist.all_reduce(t, op=dist.ReduceOp.SUM)
        per_rank_kernel_ms = [round(float(x), 3) for x in t.tolist()]
    out = None
    if rank == 0:
        bases_per_gpu = db.bases
        total_bases = job_bases * len(descrs)          # mixed batch: every descriptor scans every base
        ms_per_step = dt / args.steps * 1e3
        value = total_bases / (dt / args.steps) / 1e6
        algo_bytes = ALGO_BYTES_PER_BASE * bases_per_gpu * len(descrs)
        achieved = algo_bytes / (search_ms * 1e-3) / 1e9
        names = "+".join(os.path.basename(f) for f in descr_files)
        counters = profile_counters() if default_workload and world == 1 else None
        traffic = None
        secondary = None
        if counters and "FETCH_SIZE" in counters and "WRITE_SIZE" in counters:
            # KiB; FETCH_SIZE tallies 128-byte requests at 64 bytes on gfx950 -> doubled (MI355X_MICROARCH.md, HBM)
            traffic = int(2 * counters["FETCH_SIZE"] * 1024 + counters["WRITE_SIZE"] * 1024)
        if counters and "SQ_INSTS_VALU" in counters:
            ach = counters["SQ_INSTS_VALU"] / (search_ms * 1e-3) / 1e9
            measured_peak = None
            vp = os.path.join(ROOT, "profiles", "r02_valu_peak.json")
            if os.path.exists(vp):
                measured_peak = json.load(open(vp)).get("waves_per_simd_4")
            secondary = {"bound": "valu-issue", "achieved": round(ach, 1), "peak": round(VALU_PEAK_NOMINAL, 1),
                         "unit": "G wave64-instr/s", "frac": round(ach / VALU_PEAK_NOMINAL, 4),
                         "peak_measured": measured_peak,
                         "frac_of_measured": round(ach / measured_peak, 4) if measured_peak else None,
                         "formula": f"SQ_INSTS_VALU per launch (profiles/{PROFILE_ROUND}_trna_pmc_summary.csv) / kernel_ms of this run; "
                                    "peak = 256 CUs x 4 SIMDs x 2.4 GHz / 2 cycles; peak_measured = profiles/valu_peak.hip at 4 waves per SIMD",
                         "salu_per_valu": round(counters.get("SQ_INSTS_SALU", 0) / counters["SQ_INSTS_VALU"], 3)}
            if "SQ_THREAD_CYCLES_VALU" in counters and "SQ_ACTIVE_INST_VALU" in counters:
                secondary["active_lanes_of_64"] = round(counters["SQ_THREAD_CYCLES_VALU"] / counters["SQ_ACTIVE_INST_VALU"], 1)
        where = ("database resident in HBM (no upload in the step)" if args.resident else
                 "every step uploads its batch (0.375 B/base from page-locked host memory, upload stream) under the kernels of the step before")
        out = {
            "metric": "Mbases scanned/sec (whole node) + hits/sec, " + names,
            "value": round(value, 3),
            "unit": "Mbases/s",
            "hits_per_s": round(total_hits / (dt / args.steps), 2),
            "n_gpus": world,
            "rccl_ranks": rccl_ranks,
            "per_rank_kernel_ms": per_rank_kernel_ms,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": round(ms_per_step, 3),
            "higher_is_better": True,
            "scaling": "strong" if strong else "weak",
            "vs_baseline": None,
            "dtype": "u8",
            "data": "synthetic",
            "config": {
                "workload": f"{names} ({'mixed batch, ' if len(descrs) > 1 else ''}{'+'.join(str(d.n_elems) for d in descrs)} elements) over "
                            + (f"{args.total_records} x {args.record_len} base synthetic records divided among the ranks" if strong else
                               f"{args.records} x {args.record_len} base synthetic records per GPU")
                            + f" (iid uniform acgt, numpy default_rng({SEED})), both strands; " + where,
                "step": "SURVEY.md 8d: H2D of the packed batch + search + efn + ordering + D2H of the hits" if not args.resident else
                        "search kernel (+ drain kernel) + efn kernel + ordering + D2H of the hit records; the packed database is in HBM when the timed region starts"
                        + ("" if args.serial else "; two scanners take the steps in turns (step i's drain kernel, energies, ordering and copy back run under step i + 1's search kernel)"),
                "bases_per_gpu": bases_per_gpu,
                "total_bases": total_bases,
                "candidates": total_hits,
                "parallelism": f"{world} rank(s), sequences sharded, gather of hit records: {gather_kind}" if world > 1 else "1 GPU",
            },
            "roofline": {
                "bound": "hbm",
                "kernel": "rma_search_kernel + rma_drain_kernel" if drain_ms > 0 else "rma_search_kernel",
                "kernel_parts_ms": {k: round(float(np.mean(v)), 3) for k, v in parts.items()},
                "achieved": round(achieved, 4),
                "peak": HBM_PEAK_GBPS,
                "unit": "GB/s",
                "frac": round(achieved / HBM_PEAK_GBPS, 8),
                "traffic": traffic,
                "traffic_note": "bytes per launch, 2 x FETCH_SIZE + WRITE_SIZE of the committed PMC passes over this workload "
                                f"(profiles/{PROFILE_ROUND}_trna_pmc_summary.csv); null when that summary was made from other kernel sources",
                "algorithmic_bytes": int(algo_bytes),
                "algorithmic_bytes_per_base": ALGO_BYTES_PER_BASE,
                "kernel_ms": round(search_ms, 3),
                "pass_a_ms": round(pass_a_ms, 3) if pass_a_ms is not None else None,
                "pass_b_ms": round(search_ms - pass_a_ms, 3) if pass_a_ms is not None else None,
                "efn_kernel_ms": round(efn_ms, 3),
                "kernel_hash": kernel_hash(),
                "secondary": secondary,
                "note": "the search is integer/LDS work with dependent accesses, not HBM bound (SURVEY.md 8d); kernel_ms = the search "
                        "kernel and, where the descriptor has one, the drain kernel launched behind it (the walks of the items the "
                        "search kernel's filters let through), HIP events on the scanner's stream; pass A = "
                        "pre-filter (decode, bit rows, first-pairs test, queue), pass B = the search proper; "
                        "kernel-only rate = %.1f Mbases/s" % (bases_per_gpu * len(descrs) / (search_ms * 1e-3) / 1e6),
            },
        }

    if rank == 0 and extras:
        def timed(fn, min_steps, min_s):
            fn()
            flush()
            n, t0 = 0, time.perf_counter()
            while n < min_steps or time.perf_counter() - t0 < min_s:
                fn()
                n += 1
            flush()
            state["cur"].wait()
            torch.cuda.synchronize()
            return n, time.perf_counter() - t0

        # ---- the headline step for at least a second
        n_sus, ds = timed(step, 1, 1.2)
        out["sustained"] = {"steps": n_sus, "seconds": round(ds, 3), "value": round(db.bases * n_sus / ds / 1e6, 3), "unit": "Mbases/s"}
        # ---- the two kinds of step side by side: upload inside and database resident
        n_h, dh = timed(step_h2d, 20, 0.5)
        n_r, dr = timed(step_resident, 20, 0.5)
        out["h2d_inclusive"] = {"value": round(db.bases * n_h / dh / 1e6, 3), "unit": "Mbases/s", "ms_per_step": round(dh / n_h * 1e3, 3), "steps": n_h,
                                "what": "SURVEY.md 8d's step (= `value`'s): upload of the packed batch under the kernels of the step before, search + efn "
                                        "kernels, ordering, copy back of the hits"}
        out["resident"] = {"value": round(db.bases * n_r / dr / 1e6, 3), "unit": "Mbases/s", "ms_per_step": round(dr / n_r * 1e3, 3), "steps": n_r,
                           "what": "the same scan over a database that stays in HBM"}
        out["h2d_over_resident"] = round(out["h2d_inclusive"]["value"] / out["resident"]["value"], 3)
        if len(sets) > 1:
            # ... and one step ended before the next begins (one scanner, as until round 3)
            both = sets[:]
            del sets[1:]
            n_s, d_s = timed(step_resident, 20, 0.5)
            out["serial_steps"] = {"value": round(db.bases * n_s / d_s / 1e6, 3), "unit": "Mbases/s", "ms_per_step": round(d_s / n_s * 1e3, 3), "steps": n_s,
                                   "what": "database resident, every step ended before the next begins: search kernel, drain kernel, energies, ordering and "
                                           "copy back one after the other (`value` has two scanners take the steps in turns)"}
            sets[:] = both

        # ---- the north star's own size on one GPU, and the whole command line over it
        if default_workload and args.north_star_records > n_rec and seqs_all is not None and len(seqs_all) >= args.north_star_records:
            big = seqs_all[: args.north_star_records]
            state["cur"].close()
            bdb = sc.database(big)
            for _ in range(2):
                sc.scan(bdb, copy=False)
            n_big, hits_big, t0 = 0, 0, time.perf_counter()
            while n_big < 5 or time.perf_counter() - t0 < 1.0:
                hits_big = sc.scan(bdb, copy=False).shape[0]
                n_big += 1
            torch.cuda.synchronize()
            dbg = time.perf_counter() - t0
            _, big_ms, big_efn = sc.scan_device(bdb)
            out["north_star_1gbase"] = {"value": round(bdb.bases * n_big / dbg / 1e6, 3), "unit": "Mbases/s", "bases": bdb.bases,
                                        "steps": n_big, "seconds": round(dbg, 3), "ms_per_step": round(dbg / n_big * 1e3, 3),
                                        "candidates": hits_big, "kernel_ms": round(big_ms, 3), "efn_kernel_ms": round(big_efn, 3),
                                        "what": "descr/trna.descr over 1000 x 1 Mbase synthetic records on one GPU, database resident in HBM"}
            bdb.close()
            state["cur"] = new_db(True)
            try:
                out["cli_end_to_end"] = cli_end_to_end(args.descr, big, tmpdir)
            except Exception as e:      # noqa: BLE001 -- a leg of its own: the headline stands without it
                out["cli_end_to_end"] = {"error": str(e)[:300]}
        out["cpu_baseline"] = cpu_baseline(descr, seqs, args.cpu_bases)
        out["gpu_over_cpu"] = round(out["value"] / out["cpu_baseline"]["value"], 1)
        out["cpu_baseline_all_cores"] = cpu_all
        if default_workload and not args.no_real_db:
            state["cur"].close()
            try:
                out["real_db"] = real_db_leg(dev_index)
            except Exception as e:      # noqa: BLE001 -- a leg of its own
                out["real_db"] = {"error": str(e)[:300]}
            state["cur"] = new_db(True)
    elif rank == 0:
        out["cpu_baseline"] = None
    if rank == 0:
        print(json.dumps(out), flush=True)
    state["cur"].close()
    try:
        os.rmdir(tmpdir)
    except OSError:
        pass
    if native is not None:
        native.close()
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
