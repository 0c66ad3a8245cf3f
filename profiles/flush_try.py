"""The pooled lean instance that walks nothing (RMK_LEAN_FLUSH, option `flush`) against the one that walks its own items:
kernel ms, ms per step with two scanners in turns, and the records of both, which must be the same bytes.
python profiles/flush_try.py [descriptor ...]"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import rnamotif_amd as R  # noqa: E402

os.environ.setdefault("EFNDATA", R.EFNDATA_DIR)
names = sys.argv[1:] or ["descr/trna.descr"]
seqs = R.synthetic_records(100)
for name in names:
    d = R.Descriptor(["-descr", os.path.join(ROOT, "tests", "golden", name)])
    ref = None
    for flush in (0, 1):
        scs = [R.Scanner(d), R.Scanner(d)]
        for s in scs:
            s.set_option("flush", flush)
        db = scs[0].database(seqs)
        for s in scs:
            s.attach(db)
        hits = scs[0].scan(db)
        if ref is None:
            ref = hits
        same = hits.shape == ref.shape and bool(np.array_equal(hits, ref))

        def run(n):
            pend = None
            for i in range(n):
                s = scs[i & 1]
                s.scan_begin(db)
                if pend is not None:
                    pend.scan_end(copy=False)
                pend = s
            return pend.scan_end(copy=False).shape[0]

        run(6)
        t0 = time.perf_counter()
        n = 200
        run(n)
        dt = (time.perf_counter() - t0) / n * 1e3
        ks = []
        for _ in range(5):
            scs[0].scan_device(db)
            ks.append(scs[0].last_kernel_ms())
        k = np.mean(np.array(ks), axis=0)
        print(f"{name} flush {flush}: {dt:.3f} ms a step in turns; kernels {k[0]:.3f} + {k[1]:.3f} ms; {hits.shape[0]} candidates, same records {same}", flush=True)
        for s in scs:
            s.close()
