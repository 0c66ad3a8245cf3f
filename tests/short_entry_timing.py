"""Not a test: search kernel time on a database of short entries (the reference's test database
x 20: 81 140 entries, 45.3 Mbase), one tile per workgroup pass against groups of small tiles,
and the same bases as 20 long entries.  Run on the GPU box from the repository root:

    python tests/short_entry_timing.py [descr ...]      (paths relative to tests/golden)
"""
import gzip, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import rnamotif_amd as R
os.environ.setdefault("EFNDATA", R.EFNDATA_DIR)
data = gzip.open(os.path.join(ROOT, "tests/golden/test/gbrna.111.0.fastn.gz"), "rb").read()
open("/tmp/gb20.fastn", "wb").write(data * 20)
short = [r[2] for r in R.read_fasta("/tmp/gb20.fastn")]
long_ = [b"".join(short[: len(short) // 20])] * 20


def run(d, seqs, mode):
    if mode is None:
        os.environ.pop("RNAMOTIF_SHORT", None)
    else:
        os.environ["RNAMOTIF_SHORT"] = mode
    sc = R.Scanner(d, device=0)
    db = sc.database(seqs)
    h = sc.scan(db)
    return h, min(sc.scan_device(db)[1] for _ in range(5))


for descr in sys.argv[1:] or ["descr/trna.descr", "test/mp.ends.descr", "test/bulge.descr", "test/ire.descr",
                                "test/pk1.descr", "test/qu+tr.descr", "test/nanlin.descr"]:
    d = R.Descriptor(["-descr", os.path.join(ROOT, "tests/golden", descr)])
    h0, t0 = run(d, short, "0")
    h1, t1 = run(d, short, None)
    _, t2 = run(d, long_, None)
    print("%-22s %d candidates: one tile %.2f ms, grouped %.2f ms (same records: %s), as 20 long entries %.2f ms"
          % (descr, h0.shape[0], t0, t1, h0.shape == h1.shape and np.array_equal(h0, h1), t2), flush=True)
