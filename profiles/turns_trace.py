"""Two scanners taking the steps in turns (bench.py's step) for a kernel trace: rocprofv3 --kernel-trace -- python3 profiles/turns_trace.py"""
import os
import sys

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
pend = None
for i in range(60):
    s = scs[i & 1]
    s.scan_begin(db)
    if pend is not None:
        pend.scan_end(copy=False)
    pend = s
pend.scan_end(copy=False)
