"""The search kernel that walks nothing (RMK_LEAN_FLUSH), cut short stage by stage (RNAMOTIF_DBG ablation bits): decode + rows
(65536), + look-ahead chain (131072), + pre-filter and queue (1: nothing is taken from the queue), whole (pass A' and the
list).  Search kernel ms, the drain kernel's beside it.  python profiles/flush_stages.py [descr ...]"""
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
    for what, bits in (("decode+rows", 65536), ("+chain", 131072), ("+pre-filter", 1), ("whole", 0)):
        sc.set_option("dbg", bits)
        ks = []
        for _ in range(5):
            sc.scan_device(db)
            ks.append(sc.last_kernel_ms())
        out.append("%s %.3f (+ %.3f)" % (what, min(k[0] for k in ks), min(k[1] for k in ks)))
    sc.set_option("dbg", 0)
    print("==", name, "(search kernel ms, cumulative; drain kernel):", ", ".join(out), flush=True)
    sc.set_option("dbg", 32)
    sc.scan_device(db)
    sc.set_option("dbg", 0)
    db.close()
    sc.close()
