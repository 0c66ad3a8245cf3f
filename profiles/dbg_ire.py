import os, sys
import numpy as np
ROOT = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import rnamotif_amd as R
from oracle_binding import oracle_scan
os.environ.setdefault("EFNDATA", R.EFNDATA_DIR)
os.chdir(os.path.join(ROOT, "tests", "golden", "test"))
rng = np.random.default_rng(11)
lut = np.frombuffer(b"acgt", dtype=np.uint8)
seqs = [lut[rng.integers(0, 4, size=n)].tobytes() for n in (300_001, 4097, 2048, 2047, 131_071, 33)]
d = R.Descriptor(["-descr", "ire.descr"])
want = oracle_scan(d, seqs)
for dbg in (0, 4194304, 2097152):
    sc = R.Scanner(d)
    sc.set_option("dbg", dbg)
    got = sc.scan(sc.database(seqs))
    print("dbg", dbg, got.shape, want.shape, "equal" if got.shape == want.shape and np.array_equal(got, want) else "DIFFER")
    if got.shape == want.shape and not np.array_equal(got, want):
        bad = np.where((got != want).any(axis=1))[0]
        print(" rows", bad[:10])
        for i in bad[:4]:
            print("  got ", got[i][:40].tolist()); print("  want", want[i][:40].tolist())
    sc.set_option("dbg", 0)
