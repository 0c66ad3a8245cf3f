"""Scanners taking the steps in turns, two, three or four of them (rma_scan_begin() of step i before rma_scan_end() of step
i - depth + 1): ms per step.  python profiles/depth_try.py [descriptor]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import rnamotif_amd as R  # noqa: E402

os.environ.setdefault("EFNDATA", R.EFNDATA_DIR)
name = sys.argv[1] if len(sys.argv) > 1 else "descr/trna.descr"
d = R.Descriptor(["-descr", os.path.join(ROOT, "tests", "golden", name)])
seqs = R.synthetic_records(100)
scs = [R.Scanner(d) for _ in range(4)]
db = scs[0].database(seqs)
for s in scs:
    s.attach(db)


def run(n, depth):
    pend = []
    last = None
    for i in range(n):
        s = scs[i % depth]
        s.scan_begin(db)
        pend.append(s)
        if len(pend) == depth:
            last = pend.pop(0).scan_end(copy=False)
    while pend:
        last = pend.pop(0).scan_end(copy=False)
    return last.shape[0]


for depth in (1, 2, 3, 4, 2, 3):
    run(8, depth)
    t0 = time.perf_counter()
    n = 240
    cand = run(n, depth)
    dt = (time.perf_counter() - t0) / n * 1e3
    print(f"{name} depth {depth}: {dt:.3f} ms a step; {cand} candidates", flush=True)
