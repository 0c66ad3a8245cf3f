"""Search kernel time of every lean descriptor of the reference's test set over the bench database
(100 x 1 Mbase), tile-by-tile pass B (RNAMOTIF_POOL=0) against the pooled instance (=1).
python profiles/pool_matrix.py [descr ...]"""
import glob
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import rnamotif_amd as R  # noqa: E402

os.environ.setdefault("EFNDATA", R.EFNDATA_DIR)
G = os.path.join(ROOT, "tests", "golden")
# (descriptors that return tens of millions of candidates from random sequence are left out)
names = sys.argv[1:] or ["descr/trna.descr", "test/ire.descr", "test/bulge.descr", "test/mp.ends.descr", "test/score.1.descr",
                         "test/nanlin.descr", "test/sprintf.descr", "test/getbest.descr", "test/efn.descr", "test/ire.1.descr", "test/score.2.descr",
                         "descr/ch.descr", "descr/ch2.descr", "descr/ch2.mm.descr", "descr/ends.descr", "descr/hlx.gf.if.descr"]
seqs = R.synthetic_records(100)
for path in names:
    os.chdir(os.path.join(G, os.path.dirname(path)))
    name = os.path.basename(path)
    try:
        d = R.Descriptor(["-descr", name])
        sc = R.Scanner(d)
    except R.RnamotifError as e:
        print("%-22s refused: %s" % (name, str(e)[:60]))
        continue
    db = sc.database(seqs)
    res = []
    for pool in ("0", "1"):
        sc.set_option("pool", int(pool))
        sc.scan_device(db)
        n, ms, _ = min((sc.scan_device(db) for _ in range(3)), key=lambda x: x[1])
        res.append((n, ms))
    print("%-22s %9d candidates  tile by tile %8.3f ms  pooled %8.3f ms  %+5.1f%%" % (name, res[0][0], res[0][1], res[1][1], 100 * (res[1][1] / res[0][1] - 1)), flush=True)
    assert res[0][0] == res[1][0]
    db.close()
    sc.close()
