#!/usr/bin/env python3
"""profiles/ablate.py -- kernel time of the search kernel per descriptor under the
RNAMOTIF_DBG ablation switches: whole kernel, pass A only
(bit 1: pass B skipped), and the number of queued items (bit 2).

  python profiles/ablate.py [--records 100] descr [descr ...]

Prints one JSON line per descriptor.  The switches change no output; they are read by
rma_scan_device() at every launch.
"""
import argparse
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--records", type=int, default=100)
    ap.add_argument("--record-len", type=int, default=1_000_000)
    ap.add_argument("--reps", type=int, default=3)
    ap.add_argument("--env", action="append", default=[], help="NAME=VALUE set before the scanner is made")
    ap.add_argument("--dbg-or", type=int, default=0, help="RNAMOTIF_DBG bits set in every measurement (64 no rows, 128 no level voting, 256 no split)")
    ap.add_argument("--quick", action="store_true", help="whole kernel only")
    ap.add_argument("descr", nargs="+")
    args = ap.parse_args()
    for kv in args.env:
        k, v = kv.split("=", 1)
        os.environ[k] = v
    import rnamotif_amd as R
    os.environ.setdefault("EFNDATA", R.EFNDATA_DIR)
    seqs = R.synthetic_records(args.records, args.record_len)
    for f in args.descr:
        path = f if os.path.exists(f) else os.path.join(ROOT, "tests", "golden", "test", f)
        if not os.path.exists(path):
            path = os.path.join(ROOT, "tests", "golden", "descr", f)
        d = R.Descriptor(["-descr", path])
        sc = R.Scanner(d, device=0)
        db = sc.database(seqs)
        out = {"descr": os.path.basename(path), "bases": db.bases, "env": args.env}
        for name, dbg in (("all", None), ("pass_a_only", "1")):
            if dbg is not None and args.quick:
                continue
            sc.set_option("dbg", int(dbg or 0) | args.dbg_or)
            sc.scan_device(db)
            ms = []
            for _ in range(args.reps):
                n, s_ms, e_ms = sc.scan_device(db)
                ms.append(s_ms)
            out[name + "_ms"] = round(min(ms), 3)
            if dbg is None:
                out["candidates"] = n
                out["efn_ms"] = round(e_ms, 3)
        if not args.quick:
            sc.set_option("dbg", 34 | args.dbg_or)     # 2: count queued items, 32: wave cycles per phase
            sys.stderr.flush()
            sc.scan_device(db)      # prints "[dbg] queued items" on stderr
        sc.set_option("dbg", 0)
        out["dbg_or"] = args.dbg_or
        out["gbases_per_s"] = round(db.bases / out["all_ms"] / 1e6, 2)
        print(json.dumps(out), flush=True)
        db.close()
        sc.close()


if __name__ == "__main__":
    main()
