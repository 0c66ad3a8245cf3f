"""The drain kernel cut short (RNAMOTIF_DBG bits): items taken and dropped (268435456), no hand-overs (4194304), whole.
python profiles/drain_ablate.py"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import rnamotif_amd as R  # noqa: E402

os.environ.setdefault("EFNDATA", R.EFNDATA_DIR)
d = R.Descriptor(["-descr", os.path.join(ROOT, "tests", "golden", "descr", "trna.descr")])
one = [r[2] for r in R.read_fasta(os.path.join(ROOT, "tests", "golden", "test", "gbrna.111.0.fastn.gz"))]
for what, seqs in (("synthetic", R.synthetic_records(100)), ("gbrna x 44", one * 44)):
    sc = R.Scanner(d)
    db = sc.database(seqs)
    sc.scan_device(db)
    out = []
    for name, bits in (("items taken and dropped", 268435456), ("no hand-overs", 4194304), ("whole", 0)):
        sc.set_option("dbg", bits)
        ks = []
        for _ in range(5):
            n = sc.scan_device(db)[0]
            ks.append(sc.last_kernel_ms()[1])
        out.append("%s %.3f ms (%d candidates)" % (name, min(ks), n))
    sc.set_option("dbg", 0)
    print("==", what, "drain kernel:", ", ".join(out), flush=True)
    sc.set_option("dbg", 536870912)
    sc.scan_device(db)
    sc.set_option("dbg", 0)
    sys.stderr.flush()
    db.close()
    sc.close()
