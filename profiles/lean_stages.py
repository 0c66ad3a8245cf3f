"""Where the lean search kernel's time goes, by cutting it short (RNAMOTIF_DBG ablation bits; the
cut versions find nothing): decode + rows only (65536), + look-ahead chain (131072), + pre-filter
and queue (1: nothing is taken from the queue), + stem-loop tests of the queued items (2048: the
pool is thrown away), whole search.  python profiles/lean_stages.py [descr ...]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import rnamotif_amd as R  # noqa: E402

os.environ.setdefault("EFNDATA", R.EFNDATA_DIR)
names = sys.argv[1:] or ["trna.descr"]
seqs = R.synthetic_records(100)
os.chdir(os.path.join(ROOT, "tests", "golden", "test"))
for name in names:
    d = R.Descriptor(["-descr", name])
    sc = R.Scanner(d)
    db = sc.database(seqs)
    sc.scan_device(db)
    out = []
    for what, bits in (("decode+rows", 65536), ("+chain", 131072), ("+pre-filter", 1), ("+stem-loop tests", 2048), ("whole", 0)):
        sc.set_option("dbg", bits)
        out.append("%s %.3f" % (what, min(sc.scan_device(db)[1] for _ in range(5))))
    sc.set_option("dbg", 0)
    print("==", name, "(ms, cumulative):", ", ".join(out), flush=True)
    db.close()
    sc.close()
