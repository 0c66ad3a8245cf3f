"""Two scanners taking the steps in turns (bench.py's step): ms per step by the search kernel's workgroups per CU and the drain
kernel's waves per CU -- does step i's drain kernel run beside step i + 1's search kernel when both leave each other room?
python profiles/overlap_try.py"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import rnamotif_amd as R  # noqa: E402

os.environ.setdefault("EFNDATA", R.EFNDATA_DIR)
d = R.Descriptor(["-descr", os.path.join(ROOT, "tests", "golden", "descr", "trna.descr")])
seqs = R.synthetic_records(100)
scs = [R.Scanner(d), R.Scanner(d)]
db = scs[0].database(seqs)
for s in scs:
    s.attach(db)


def run(n):
    pend = None
    for i in range(n):
        s = scs[i & 1]
        s.scan_begin(db)
        if pend is not None:
            pend.scan_end(copy=False)
        pend = s
    return pend.scan_end(copy=False).shape[0]


for wgs, dw in ((0, 0), (0, 1), (0, 2), (0, 3), (0, 4), (0, 6), (0, 8), (0, 0), (0, 4), (0, 2)):
    for s in scs:
        s.set_option("search_wgs", wgs)
        s.set_option("drain_waves", dw)
    run(6)
    t0 = time.perf_counter()
    n = 200
    cand = run(n)
    dt = (time.perf_counter() - t0) / n * 1e3
    scs[0].scan_device(db)
    k = scs[0].last_kernel_ms()
    print(f"search_wgs {wgs} drain_waves {dw}: {dt:.3f} ms a step in turns; one scan's kernels {k[0]:.3f} + {k[1]:.3f} ms; {cand} candidates", flush=True)
