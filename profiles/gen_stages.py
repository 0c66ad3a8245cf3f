"""A general instance cut short stage by stage (RNAMOTIF_DBG ablation bits): decode + rows (65536), + the strand filter's vectors
of a leading 4-plex (131072), + pre-filter and queue (1: nothing is taken from the queue), whole.  python profiles/gen_stages.py [descr ...]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import rnamotif_amd as R  # noqa: E402

os.environ.setdefault("EFNDATA", R.EFNDATA_DIR)
names = sys.argv[1:] or ["qu+tr.descr", "pk1.descr"]
seqs = R.synthetic_records(100)
os.chdir(os.path.join(ROOT, "tests", "golden", "test"))
for name in names:
    d = R.Descriptor(["-descr", name])
    sc = R.Scanner(d)
    db = sc.database(seqs)
    sc.scan_device(db)
    out = []
    for what, bits in (("decode+rows", 65536), ("+vectors", 131072), ("+pre-filter", 1), ("whole", 0)):
        sc.set_option("dbg", bits)
        ks = []
        for _ in range(4):
            sc.scan_device(db)
            ks.append(sc.last_kernel_ms()[0])
        out.append("%s %.3f" % (what, min(ks)))
    sc.set_option("dbg", 0)
    print("==", name, "(search kernel ms, cumulative):", ", ".join(out), flush=True)
    db.close()
    sc.close()
