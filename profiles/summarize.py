#!/usr/bin/env python3
"""Condense a gpurun_out/prof_* directory (rocprofv3 csv output) into the small
summaries committed under profiles/: kernel_stats + per-kernel PMC means.

usage: summarize.py gpurun_out/prof_r1 profiles/r01_label
"""
import collections
import csv
import glob
import os
import shutil
import sys

src, dst = sys.argv[1], sys.argv[2]
# what the profiled binary was built from: bench.py quotes the counters only for the same sources
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import json
from bench import kernel_hash
with open(dst + "_meta.json", "w") as out:
    json.dump({"kernel_hash": kernel_hash(), "source": src}, out)
    out.write("\n")
def newest(pattern):
    """(gpurun merges every call's files into gpurun_out/: a directory collected twice holds both runs' files)"""
    files = sorted(glob.glob(pattern), key=os.path.getmtime)
    return files[-1:]


for f in newest(os.path.join(src, "trace", "*", "*_kernel_stats.csv")):
    shutil.copy(f, dst + "_kernel_stats.csv")
with open(dst + "_pmc_summary.csv", "w") as out:
    w = csv.writer(out)
    w.writerow(["pass", "kernel", "counter", "dispatches", "mean_per_dispatch"])
    for d in sorted(glob.glob(os.path.join(src, "pmc_*"))):
        for f in newest(os.path.join(d, "*", "*_counter_collection.csv")):
            agg = collections.defaultdict(list)
            for r in csv.DictReader(open(f)):
                agg[(r["Kernel_Name"].split("(")[0], r["Counter_Name"])].append(float(r["Counter_Value"]))
            for (k, c), v in sorted(agg.items()):
                if k.startswith("__amd"):
                    continue
                w.writerow([os.path.basename(d), k, c, len(v), "%.6g" % (sum(v) / len(v))])
for f in glob.glob(os.path.join(src, "bench_*.log")):
    for line in open(f, errors="replace"):
        if line.startswith('{"metric"'):
            with open(dst + "_" + os.path.basename(f).replace(".log", ".json"), "w") as o:
                o.write(line)
