"""The drain kernel's own counters (RNAMOTIF_DBG bit 32) over the synthetic 100 Mbase and the reference's database x 44.
python profiles/drain_stats.py"""
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
    sc.scan_device(db)
    k = sc.last_kernel_ms()
    print(f"== {what}: search {k[0]:.3f} + drain {k[1]:.3f} ms", flush=True)
    sys.stderr.flush()
    sc.set_option("dbg", 32)
    sc.scan_device(db)
    sc.set_option("dbg", 0)
    sys.stderr.flush()
    db.close()
    sc.close()
