#!/bin/bash
# profiles/passb_pmc.sh -- VALU instructions and active lanes of the search kernel with and without
# pass B (RNAMOTIF_DBG=1 skips it): what pass B issues and how full its waves are.
set -u
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
O=$R/gpurun_out/passb
rm -rf "$O"; mkdir -p "$O"
cd /tmp && export TMPDIR=/tmp
B="$R/bench.py --steps 1 --warmup 0 --cpu-bases 0 --north-star-records 0 $*"
for v in 0 1; do
	if [ $v = 1 ]; then export RNAMOTIF_DBG=1; fi
	timeout 300 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES -d $O/a$v --output-format csv -- python3 $B > $O/a$v.log 2>&1
	timeout 300 rocprofv3 --kernel-trace --pmc SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_ACTIVE_INST_SCA -d $O/b$v --output-format csv -- python3 $B > $O/b$v.log 2>&1
done
python3 - <<PY
import csv, glob, collections
for v in "01":
    tot = collections.Counter(); n = 0
    for d in ("a", "b"):
        for f in glob.glob("$O/%s%s/**/*counter_collection.csv" % (d, v), recursive=True):
            for row in csv.DictReader(open(f)):
                if "rma_search_kernel" in row["Kernel_Name"]:
                    tot[row["Counter_Name"]] += float(row["Counter_Value"])
                    if row["Counter_Name"] == "SQ_INSTS_VALU": n += 1
    print("DBG=%s launches %d" % (v, n), {k: round(x / max(n, 1) / 1e6, 2) for k, x in sorted(tot.items())})
PY
