import os, sys
ROOT = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
sys.path.insert(0, ROOT)
import rnamotif_amd as R
os.environ.setdefault("EFNDATA", R.EFNDATA_DIR)
seqs = R.synthetic_records(100)
d = R.Descriptor(["-descr", os.path.join(ROOT, "tests", "golden", "descr", "trna.descr")])
sc = R.Scanner(d)
db = sc.database(seqs)
sc.scan_device(db)
for dw in (0, 2, 4, 6, 8, 10, 12, 16):
    sc.set_option("drain_waves", dw)
    for refill in (48, 16, 4):
        sc.set_option("pool_refill", refill)
        sc.scan_device(db)
        best = (9, 9)
        for _ in range(5):
            sc.scan_device(db)
            k = sc.last_kernel_ms()
            best = min(best, (k[0] + k[1], k[1]))
        print("drain_waves %2d refill %2d: search+drain %.3f ms, drain %.3f" % (dw, refill, best[0], best[1]), flush=True)
