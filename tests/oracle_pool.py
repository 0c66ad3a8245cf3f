"""The CPU oracle over many records at once, one process per host core -- TEST INFRASTRUCTURE.

Only tests and bench.py's cpu_baseline legs import this.  The oracle is scalar C (oracle/rm_oracle_scan.c);
the reference's own way to use more cores is one process per database file (mrnamotif.c), and this is the
same: every worker is a fresh interpreter (multiprocessing "spawn": a child process started with its own
program, not a copy of a parent that may hold the GPU), compiles the descriptor for itself, makes the
records it is asked for from the BASELINE stream (numpy default_rng(20240601); PCG64.advance jumps to record
k) and scans them one after the other.  At BASELINE's full size (100 records of 1 Mbase) trna.descr is
85 core-seconds of oracle, pk1.descr some 500: seconds on the cores of a GPU box.
"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SEED = 20240601


def host_cores():
    """Cores this process may use: its affinity mask, cut to the cgroup's CPU quota if there is one."""
    n = len(os.sched_getaffinity(0))
    try:
        with open("/sys/fs/cgroup/cpu.max") as f:
            quota, period = f.read().split()[:2]
        if quota != "max":
            n = max(1, min(n, int(int(quota) / int(period) + 0.5)))
    except (OSError, ValueError):
        pass
    return n


def synthetic_record(k, length=1_000_000):
    """Record k of the synthetic stream (the k-th draw of `length` integers of default_rng(20240601))."""
    import numpy as np
    lut = np.frombuffer(b"acgt", dtype=np.uint8)
    bg = np.random.PCG64(SEED)
    rng = np.random.Generator(bg)
    if length % 2 == 0:
        bg.advance(k * (length // 2))
        return lut[rng.integers(0, 4, size=length)].tobytes()
    s = None
    for _ in range(k + 1):
        s = lut[rng.integers(0, 4, size=length)].tobytes()
    return s


def _worker(job):
    descr_args, cwd, records, length, limit = job
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import rnamotif_amd as R
    from oracle_binding import oracle_scan
    os.environ.setdefault("EFNDATA", R.EFNDATA_DIR)
    if cwd:
        os.chdir(cwd)
    d = R.Descriptor(list(descr_args))
    out = []
    busy = 0.0
    for k in records:
        s = synthetic_record(k, length)
        if limit:
            s = s[:limit]
        t0 = time.perf_counter()
        h = oracle_scan(d, [s])
        busy += time.perf_counter() - t0
        out.append((k, h.shape, h.tobytes()))
    return out, busy


def oracle_records(descr_args, records, cwd=None, length=1_000_000, procs=None, limit=0):
    """The oracle's candidates of each of `records` (numbers within the synthetic stream), scanned as a
    database of that one record: {record: int32 array [n, stride], entry number 0}.  Returns
    (that dict, {"procs", "wall_s", "busy_s" (the slowest worker's time in the oracle), "cpu_s" (all workers')}).
    limit: only the first so many bases of every record."""
    import multiprocessing as mp
    import numpy as np
    records = list(records)
    procs = max(1, min(procs or host_cores(), len(records)))
    # records dealt round-robin: every worker gets the same number, give or take one
    jobs = [(list(descr_args), cwd, records[i::procs], length, limit) for i in range(procs)]
    t0 = time.perf_counter()
    with mp.get_context("spawn").Pool(procs) as pool:
        res = pool.map(_worker, jobs)
    wall = time.perf_counter() - t0
    out = {}
    for part, _ in res:
        for k, shape, raw in part:
            out[k] = np.frombuffer(raw, dtype=np.int32).reshape(shape).copy()
    return out, {"procs": procs, "wall_s": wall, "busy_s": max(b for _, b in res), "cpu_s": sum(b for _, b in res)}


def concat_records(per_record, records, stride):
    """The per-record candidates as the candidates of one database holding `records` in this order."""
    import numpy as np
    parts = []
    for i, k in enumerate(records):
        h = per_record[k].copy()
        h[:, 0] = i
        parts.append(h)
    return np.concatenate(parts) if parts else np.zeros((0, stride), np.int32)
