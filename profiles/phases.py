"""Wave-cycle shares of the search kernel's phases (RNAMOTIF_DBG=32 counters) for a descriptor over
the bench database.  python profiles/phases.py [descr ...]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import rnamotif_amd as R  # noqa: E402

os.environ.setdefault("EFNDATA", R.EFNDATA_DIR)
names = sys.argv[1:] or ["trna.descr"]
HERE = os.getcwd()
seqs = R.synthetic_records(100)
os.chdir(os.path.join(ROOT, "tests", "golden", "test"))
for name in names:
    d = R.Descriptor(["-descr", name])
    sc = R.Scanner(d)
    db = sc.database(seqs)
    sc.scan_device(db)
    best = lambda: min(sc.scan_device(db)[1] for _ in range(5))
    print("==", name, "search %.3f ms" % best(), flush=True)
    sc.set_option("dbg", 2097152)
    print("   items whole in the drain kernel's list: %.3f ms" % best(), flush=True)
    sc.set_option("dbg", 0)
    sc.set_option("drain", 0)
    print("   no drain kernel: %.3f ms" % best(), flush=True)
    sc.set_option("drain", 1)
    sc.set_option("dbg", 2 + 1048576)
    sc.scan_device(db)
    sys.stderr.flush()
    sc.set_option("dbg", 34)
    sc.scan_device(db)
    sys.stderr.flush()
    sc.set_option("dbg", 0)
    db.close()
    sc.close()
