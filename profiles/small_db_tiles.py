"""General instances over SMALL databases: tiles sized so that the database makes as many as the device holds workgroups
(layout_for) against the scanner's usual tile (RNAMOTIF_TILE forces it).  python profiles/small_db_tiles.py"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import rnamotif_amd as R  # noqa: E402

os.environ.setdefault("EFNDATA", R.EFNDATA_DIR)
one = [r[2] for r in R.read_fasta(os.path.join(ROOT, "tests", "golden", "test", "gbrna.111.0.fastn.gz"))]
long_one = [b"".join(one)]
syn = R.synthetic_records(1)
for name in ("pk1.descr", "qu+tr.descr"):
    d = R.Descriptor(["-descr", os.path.join(ROOT, "tests", "golden", "test", name)])
    for what, seqs in (("gbrna, 2.26 Mbase in 4067 entries", one), ("the same as one entry", long_one), ("1 Mbase synthetic", syn), ("100 kbase synthetic", [syn[0][:100_000]])):
        out = []
        for tile in ("0", "2560"):
            os.environ["RNAMOTIF_TILE"] = tile
            sc = R.Scanner(d)
            db = sc.database(seqs)
            sc.scan_device(db)
            ks = []
            for _ in range(5):
                n = sc.scan_device(db)[0]
                ks.append(sc.last_kernel_ms()[0])
            out.append("%s %.3f ms (%d)" % ("sized to the device" if tile == "0" else "tiles of 2560", min(ks), n))
            sc.close()
        os.environ.pop("RNAMOTIF_TILE", None)
        print(name, what + ":", ", ".join(out), flush=True)
