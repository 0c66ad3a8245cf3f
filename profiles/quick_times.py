"""Kernel ms (search + drain, HIP events on the scanner's stream) of the BASELINE descriptors over the 100 Mbase synthetic
database, and of the lean ones over the reference's test database x 44 as short entries: the numbers a kernel change is
judged by, in one short run.  python profiles/quick_times.py [--real] [descr ...]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import rnamotif_amd as R  # noqa: E402

os.environ.setdefault("EFNDATA", R.EFNDATA_DIR)
args = [a for a in sys.argv[1:] if not a.startswith("--")]
names = args or ["trna.descr", "mp.ends.descr", "ire.descr", "pk1.descr", "qu+tr.descr"]


def path(n):
    return os.path.join(ROOT, "tests", "golden", "descr" if n == "trna.descr" else "test", n)


def times(sc, db, reps=7):
    sc.scan_device(db)
    parts, n = [], 0
    for _ in range(reps):
        n, _s, e_ms = sc.scan_device(db)
        k = sc.last_kernel_ms()
        parts.append((k[0], k[1], e_ms))
    a = np.asarray(parts)
    return n, a.mean(axis=0), a.min(axis=0)


seqs = R.synthetic_records(100)
for n in names:
    d = R.Descriptor(["-descr", path(n)])
    sc = R.Scanner(d)
    db = sc.database(seqs)
    cand, mean, best = times(sc, db)
    print("%-14s synthetic 100 Mbase: search %.3f + drain %.3f = %.3f ms (best %.3f), efn %.3f, %d candidates" %
          (n, mean[0], mean[1], mean[0] + mean[1], best[0] + best[1], mean[2], cand), flush=True)
    db.close()
    sc.close()
if "--real" in sys.argv:
    one = [r[2] for r in R.read_fasta(os.path.join(ROOT, "tests", "golden", "test", "gbrna.111.0.fastn.gz"))]
    short, long_ = one * 44, [b"".join(one)] * 44
    for n in names:
        d = R.Descriptor(["-descr", path(n)])
        sc = R.Scanner(d)
        for what, s in (("short", short), ("long", long_)):
            db = sc.database(s)
            cand, mean, best = times(sc, db, 5)
            print("%-14s gbrna x 44 %-5s entries: search %.3f + drain %.3f = %.3f ms, %d candidates" % (n, what, mean[0], mean[1], mean[0] + mean[1], cand), flush=True)
            db.close()
        sc.close()
