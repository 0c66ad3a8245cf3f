"""Not a test: where groups of small tiles stop paying -- 40 Mbase of random sequence cut into
entries of one length, one tile per workgroup pass against groups (RNAMOTIF_SHORT=0/1).
Round 1: groups win up to entries of about 4000 bases (db_upload's SHORT_ENTRY_MEAN)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, rnamotif_amd as R
os.environ.setdefault("EFNDATA", R.EFNDATA_DIR)
rng = np.random.default_rng(5)
lut = np.frombuffer(b"acgt", dtype=np.uint8)
big = lut[rng.integers(0, 4, size=40_000_000)].tobytes()
for L in (300, 1000, 2000, 3000, 5000, 10000):
    seqs = [big[i:i + L] for i in range(0, len(big) - L, L)]
    for descr in ("descr/trna.descr", "test/mp.ends.descr"):
        d = R.Descriptor(["-descr", os.path.join(ROOT, "tests/golden", descr)])
        out = []
        for mode in ("0", "1"):
            os.environ["RNAMOTIF_SHORT"] = mode
            sc = R.Scanner(d, device=0)
            db = sc.database(seqs)
            sc.scan_device(db)
            out.append(min(sc.scan_device(db)[1] for _ in range(4)))
        print("entry length %5d %-20s one tile %.2f ms, grouped %.2f ms" % (L, descr, out[0], out[1]), flush=True)
