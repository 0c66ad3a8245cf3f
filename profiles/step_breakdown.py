"""Where a step of bench.py goes: the laps of rma_scan() (RNAMOTIF_TIMING) and the wall time of the
Python call around it, 100 x 1 Mbase against trna.descr.  python profiles/step_breakdown.py"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import rnamotif_amd as R  # noqa: E402

os.environ.setdefault("EFNDATA", R.EFNDATA_DIR)
n_rec = int(sys.argv[1]) if len(sys.argv) > 1 else 100
d = R.Descriptor(["-descr", os.path.join(ROOT, "tests", "golden", "descr", "trna.descr")])
sc = R.Scanner(d)
db = sc.database(R.synthetic_records(n_rec))
for _ in range(3):
    sc.scan(db, copy=False)
t = []
for _ in range(20):
    t0 = time.perf_counter()
    h = sc.scan(db, copy=False)
    t.append(time.perf_counter() - t0)
print("python call: median %.3f ms, min %.3f ms, %d records" % (np.median(t) * 1e3, min(t) * 1e3, h.shape[0]))
_, k1, k2 = sc.scan_device(db)
print("kernels by events: search %.3f ms, efn %.3f ms" % (k1, k2))
t = []
for _ in range(20):
    t0 = time.perf_counter()
    sc.scan_device(db)
    t.append(time.perf_counter() - t0)
print("device part (rma_scan_device): median %.3f ms" % (np.median(t) * 1e3))
sc.set_option("timing", 1)
sys.stderr.flush()
for _ in range(3):
    sc.scan(db, copy=False)
    sys.stderr.write("--\n")
