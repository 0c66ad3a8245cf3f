#!/usr/bin/env python3
"""bench.py -- headline benchmark of the scan path (BASELINE.json).

One step = one pass of the hot path over one synthetic database that is already
resident in HBM: search kernel over every start position of both strands, efn
kernel over every candidate, copy back and ordering of the hit records
(rma_scan of the C ABI).  Workload at N=1: descr/trna.descr (4-stem cloverleaf,
bits()+efn() score) over the 100 Mbase synthetic FASTA of BASELINE.md
(100 records x 1 Mbase, iid uniform acgt, numpy default_rng(20240601)).

With --gpus N > 1 (launched by torch.distributed.run, one rank per GPU) every
rank scans its own 100 Mbase of the same synthetic stream (records
[100*rank, 100*rank+100)), i.e. weak scaling; the hit records are gathered to
rank 0 over RCCL inside the timed region, which is the path's only exchange.

Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

ALGO_BYTES_PER_BASE = 0.375      # 2 bit code + 1 bit ambiguity mask, read once for both strands
HBM_PEAK_GBPS = 8000.0           # MI355X_MICROARCH.md: 8.0 TB/s spec


def synthetic_slice(first: int, count: int, length: int):
    """Records [first, first+count) of the synthetic stream (seed 20240601)."""
    import numpy as np
    rng = np.random.default_rng(20240601)
    lut = np.frombuffer(b"acgt", dtype=np.uint8)
    out = []
    for k in range(first + count):
        v = rng.integers(0, 4, size=length)
        if k >= first:
            out.append(lut[v].tobytes())
    return out


def profiled_traffic(kernel="rma_search_kernel"):
    """HBM bytes per launch of the dominant kernel from the committed rocprofv3
    PMC passes (profiles/r01_final7_pmc_summary.csv: FETCH_SIZE and WRITE_SIZE in
    separate passes over this same default workload, in KiB).  Corrected as
    MI355X_MICROARCH.md prescribes for gfx950: FETCH_SIZE tallies 128-byte
    requests at 64 bytes -> doubled; WRITE_SIZE is exact."""
    import csv
    path = os.path.join(ROOT, "profiles", "r01_final7_pmc_summary.csv")
    if not os.path.exists(path):
        return None
    kb = {}
    for r in csv.DictReader(open(path)):
        if kernel in r["kernel"] and r["counter"] in ("FETCH_SIZE", "WRITE_SIZE"):
            kb[r["counter"]] = float(r["mean_per_dispatch"])
    if len(kb) != 2:
        return None
    return int(2 * kb["FETCH_SIZE"] * 1024 + kb["WRITE_SIZE"] * 1024)


def profiled_issue(kernel_ms, kernel="rma_search_kernel"):
    """The bound that does apply: instruction issue.  VALU wave-instructions per launch from
    the committed PMC pass over this workload (SQ_INSTS_VALU) divided by the kernel time
    measured now, against 256 CUs x 4 SIMDs x 2.4 GHz / 4 cycles per 64-wide VALU instruction;
    lanes = average active lanes per VALU instruction (SQ_THREAD_CYCLES_VALU / SQ_ACTIVE_INST_VALU)."""
    import csv
    path = os.path.join(ROOT, "profiles", "r01_final7_pmc_summary.csv")
    if not os.path.exists(path):
        return None
    c = {}
    for r in csv.DictReader(open(path)):
        if kernel in r["kernel"]:
            c[r["counter"]] = float(r["mean_per_dispatch"])
    if "SQ_INSTS_VALU" not in c:
        return None
    peak = 256 * 4 * 2.4e9 / 4 / 1e9
    ach = c["SQ_INSTS_VALU"] / (kernel_ms * 1e-3) / 1e9
    out = {"bound": "valu-issue", "achieved": round(ach, 1), "peak": round(peak, 1), "unit": "G wave-instr/s",
           "frac": round(ach / peak, 4), "salu_per_valu": round(c.get("SQ_INSTS_SALU", 0) / c["SQ_INSTS_VALU"], 3)}
    if "SQ_THREAD_CYCLES_VALU" in c and "SQ_ACTIVE_INST_VALU" in c:
        out["active_lanes_of_64"] = round(c["SQ_THREAD_CYCLES_VALU"] / c["SQ_ACTIVE_INST_VALU"], 1)
    return out


def cpu_model():
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def cpu_baseline(descr, seqs, budget_bases):
    """The scalar CPU oracle (kind 'port': byte-identical to the reference on
    its golden tests, and within a few percent of its speed here) on a bounded
    sample of the same workload, one thread."""
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
    """The reference's own way to use more cores is one process per database file
    (mrnamotif); the same here: every host core scans its own slice of the database."""
    import multiprocessing as mp
    cores = min(len(os.sched_getaffinity(0)), 16, len(seqs))
    jobs = [(descr_path, seqs[k][:bases_per_core]) for k in range(cores)]
    t0 = time.perf_counter()
    with mp.get_context("spawn").Pool(cores) as pool:
        res = pool.map(_cpu_worker, jobs)
    wall = time.perf_counter() - t0
    busy = max(r[2] for r in res)
    total = sum(r[0] for r in res)
    return {"value": round(total / busy / 1e6, 4), "unit": "Mbases/s", "cores": cores, "kind": "port", "cpu": cpu_model(),
            "sample": f"{cores} processes x first {bases_per_core} bases of records 0..{cores - 1}, both strands, "
                      f"{sum(r[1] for r in res)} candidates, slowest process {busy:.1f} s (wall {wall:.1f} s with start-up)"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--records", type=int, default=100, help="1 Mbase records per GPU (default 100 = 100 Mbase)")
    ap.add_argument("--record-len", type=int, default=1_000_000)
    ap.add_argument("--descr", default=os.path.join(ROOT, "tests", "golden", "descr", "trna.descr"),
                    help="descriptor file; a comma separated list = mixed batch (every descriptor over the same database)")
    ap.add_argument("--cpu-bases", type=int, default=12_000_000, help="sample size of the CPU baseline (0 = skip)")
    args = ap.parse_args()

    seqs = None
    cpu_all = None
    world_env = int(os.environ.get("WORLD_SIZE", "1"))
    if world_env == 1 and args.cpu_bases > 0 and "," not in args.descr:
        # host-only work first: worker processes are spawned before this process touches the GPU
        seqs = synthetic_slice(0, args.records, args.record_len)
        cpu_all = cpu_baseline_all_cores(args.descr, seqs, min(args.cpu_bases // 2, args.record_len))

    import numpy as np
    import torch
    import rnamotif_amd as R

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        torch.cuda.set_device(local_rank)
        dist.init_process_group(backend="nccl", device_id=torch.device("cuda", local_rank))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the scan path has no CPU implementation")
    dev = torch.device("cuda", local_rank)

    os.environ.setdefault("EFNDATA", R.EFNDATA_DIR)
    default_workload = (args.descr == ap.get_default("descr") and args.records == 100 and
                        args.record_len == 1_000_000)
    descr_files = args.descr.split(",")
    descrs = [R.Descriptor(["-descr", f]) for f in descr_files]
    descr = descrs[0]
    if seqs is None:
        seqs = synthetic_slice(rank * args.records, args.records, args.record_len)
    scs = [R.Scanner(d, device=local_rank) for d in descrs]
    dbs = [s.database(seqs) for s in scs]
    sc, db = scs[0], dbs[0]

    from rnamotif_amd.distributed import gather_hits as gather_to_rank0
    my_index = [rank * args.records + i for i in range(args.records)]   # entry numbers within the whole job

    def gather_hits(h, stride):
        """Variable length gather of hit records to rank 0 over RCCL (rnamotif_amd/distributed.py,
        the same function the world-size-2 gloo test runs)."""
        if world == 1:
            return [h]
        # every rank holds a consecutive run of entries: the per-rank arrays in rank order are the
        # ordered hit stream of the whole job, left as they arrive (no concatenation on rank 0)
        return gather_to_rank0(h, my_index, stride, device=dev, concat=False)

    def step():
        n = 0
        for d_, sc_, db_ in zip(descrs, scs, dbs):
            n += sum(part.shape[0] for part in gather_hits(sc_.scan(db_, copy=False), d_.hit_stride))
        return n

    for _ in range(args.warmup):
        step()

    def fence():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    fence()
    t0 = time.perf_counter()
    total_hits = 0
    for _ in range(args.steps):
        total_hits = step()
    fence()
    dt = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([dt], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())

    # kernel time of the dominant kernel, HIP events on the scanner's own stream
    kms = []
    for _ in range(max(1, min(args.steps, 3))):
        tot = [0.0, 0.0]
        for sc_, db_ in zip(scs, dbs):
            n_cand, s_ms, e_ms = sc_.scan_device(db_)
            tot[0] += s_ms
            tot[1] += e_ms
        kms.append(tuple(tot))
    search_ms = float(np.mean([k[0] for k in kms]))
    efn_ms = float(np.mean([k[1] for k in kms]))

    if rank == 0:
        bases_per_gpu = db.bases
        total_bases = bases_per_gpu * world * len(descrs)   # mixed batch: every descriptor scans every base
        ms_per_step = dt / args.steps * 1e3
        value = total_bases / (dt / args.steps) / 1e6
        achieved = ALGO_BYTES_PER_BASE * bases_per_gpu * len(descrs) / (search_ms * 1e-3) / 1e9
        names = "+".join(os.path.basename(f) for f in descr_files)
        out = {
            "metric": "Mbases scanned/sec (whole node) + hits/sec, " + names,
            "value": round(value, 3),
            "unit": "Mbases/s",
            "hits_per_s": round(total_hits / (dt / args.steps), 2),
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": round(ms_per_step, 3),
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "u8",
            "data": "synthetic",
            "config": {
                "workload": f"{names} ({'mixed batch, ' if len(descrs) > 1 else ''}{'+'.join(str(d.n_elems) for d in descrs)} elements) over "
                            f"{args.records} x {args.record_len} base synthetic records per GPU "
                            f"(iid uniform acgt, numpy default_rng(20240601)), both strands",
                "bases_per_gpu": bases_per_gpu,
                "total_bases": total_bases,
                "candidates": total_hits,
                "parallelism": f"{world} rank(s), sequences sharded, RCCL gather of hit records" if world > 1 else "1 GPU",
            },
            "roofline": {
                "bound": "hbm",
                "kernel": "rma_search_kernel",
                "achieved": round(achieved, 4),
                "peak": HBM_PEAK_GBPS,
                "unit": "GB/s",
                "frac": round(achieved / HBM_PEAK_GBPS, 8),
                "traffic": profiled_traffic() if default_workload and world == 1 else None,
                "traffic_note": "bytes per launch from the committed PMC passes of this workload "
                                "(profiles/r01_final7_pmc_summary.csv), 2 x FETCH_SIZE + WRITE_SIZE",
                "algorithmic_bytes": int(ALGO_BYTES_PER_BASE * bases_per_gpu * len(descrs)),
                "kernel_ms": round(search_ms, 3),
                "efn_kernel_ms": round(efn_ms, 3),
                "algorithmic_bytes_per_base": ALGO_BYTES_PER_BASE,
                "secondary": profiled_issue(search_ms) if default_workload and world == 1 else None,
                "note": "the search is integer/LDS issue bound, not HBM bound (SURVEY.md 8d); "
                        "kernel-only rate = %.1f Mbases/s" % (bases_per_gpu * len(descrs) / (search_ms * 1e-3) / 1e6),
            },
        }
        if world == 1 and args.cpu_bases > 0:
            out["cpu_baseline"] = cpu_baseline(descr, seqs, args.cpu_bases)
            out["gpu_over_cpu"] = round(value / out["cpu_baseline"]["value"], 1)
            out["cpu_baseline_all_cores"] = cpu_all
        else:
            out["cpu_baseline"] = None
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
