"""Search kernel time, queued items and candidates of the general-instance descriptors over the bench
database.  python profiles/gen_counts.py [descr ...]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import rnamotif_amd as R  # noqa: E402

os.environ.setdefault("EFNDATA", R.EFNDATA_DIR)
names = sys.argv[1:] or ["pk1.descr", "qu+tr.descr", "pk_j1+2.descr"]
seqs = R.synthetic_records(100)
os.chdir(os.path.join(ROOT, "tests", "golden", "test"))
for name in names:
    d = R.Descriptor(["-descr", name])
    sc = R.Scanner(d)
    db = sc.database(seqs)
    sc.scan_device(db)
    print("==", name, "search %.3f ms" % min(sc.scan_device(db)[1] for _ in range(3)), flush=True)
    sc.set_option("dbg", 34)
    sc.scan_device(db)
    sys.stderr.flush()
    db.close()
    sc.close()
