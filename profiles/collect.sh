#!/bin/bash
# profiles/collect.sh LABEL [bench.py flags] -- on the GPU box: the default bench.py workload (or the
# one the flags name, e.g. --descr tests/golden/test/pk1.descr) under rocprofv3,
# one kernel-trace + stats run and separate PMC passes (never combined with other trace
# domains), written under gpurun_out/prof_LABEL; profiles/summarize.py condenses them.
#   gpurun --timeout 900 -- 'bash profiles/collect.sh r1x'
#   python3 profiles/summarize.py gpurun_out/prof_r1x profiles/r01_x
set -u
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
O=$R/gpurun_out/prof_$1
shift
rm -rf "$O"; mkdir -p "$O"
cd /tmp && export TMPDIR=/tmp
B="$R/bench.py --steps 1 --warmup 0 --cpu-bases 0 $*"
python3 $R/bench.py --cpu-bases 0 $* > $O/bench_plain.log 2>&1
timeout 300 rocprofv3 --kernel-trace --stats -d $O/trace --output-format csv -- python3 $B > $O/bench_trace.log 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $O/pmc_fetch --output-format csv -- python3 $B > $O/bench_fetch.log 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $O/pmc_write --output-format csv -- python3 $B > $O/bench_write.log 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc SQ_BUSY_CYCLES SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVES SQ_WAVE_CYCLES \
	-d $O/pmc_sq --output-format csv -- python3 $B > $O/bench_sq.log 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_INSTS_SMEM SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_ANY \
	-d $O/pmc_sq2 --output-format csv -- python3 $B > $O/bench_sq2.log 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VALU SQ_INST_CYCLES_SALU SQ_THREAD_CYCLES_VALU SQ_WAIT_INST_LDS \
	-d $O/pmc_sq3 --output-format csv -- python3 $B > $O/bench_sq3.log 2>&1
# (round 4: what the stall breakdown of profiles/stalls.py is made from)
timeout 300 rocprofv3 --kernel-trace --pmc SQ_WAIT_ANY SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_VALU SQ_INSTS_BRANCH SQ_INST_LEVEL_LDS SQ_INST_LEVEL_VMEM SQ_LEVEL_WAVES \
	-d $O/pmc_sq4 --output-format csv -- python3 $B > $O/bench_sq4.log 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc SQ_INSTS_LDS_LOAD SQ_INSTS_LDS_STORE SQ_IFETCH SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_FLAT SQ_LDS_ADDR_CONFLICT SQ_LDS_IDX_ACTIVE SQ_CYCLES \
	-d $O/pmc_sq5 --output-format csv -- python3 $B > $O/bench_sq5.log 2>&1
grep -h '^{"metric"' $O/bench_plain.log | cut -c1-400
ls $O
