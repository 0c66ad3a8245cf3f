"""From a rocprofv3 --kernel-trace csv: the kernels' intervals in time order -- which ran side by side.
python3 profiles/overlap_trace.py <dir with *_kernel_trace.csv> [first] [count]"""
import csv
import glob
import sys

d = sys.argv[1]
first = int(sys.argv[2]) if len(sys.argv) > 2 else 400
count = int(sys.argv[3]) if len(sys.argv) > 3 else 24
f = sorted(glob.glob(d + "/**/*kernel_trace.csv", recursive=True))[0]
rows = list(csv.DictReader(open(f)))
ev = []
for r in rows:
    n = r["Kernel_Name"]
    short = "S" if "rma_search" in n else "D" if "rma_drain" in n else "E" if "rma_efn" in n else "o"
    ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), short, r.get("Queue_Id", "?")))
ev.sort()
t0 = ev[first][0]
for s, e, k, q in ev[first:first + count]:
    print(f"{k} queue {q}: {(s - t0) / 1e3:9.1f} .. {(e - t0) / 1e3:9.1f} us  ({(e - s) / 1e3:7.1f})")
