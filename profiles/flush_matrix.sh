#!/bin/bash
# profiles/flush_matrix.sh -- on the GPU box: the instance that walks nothing at 4, 5, 6 workgroups a CU (FLUSH_WAVES_PER_SIMD),
# each built there, timed by profiles/flush_try.py.   gpurun --timeout 1500 -- 'bash profiles/flush_matrix.sh'
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd $R/rnamotif_amd/csrc
for N in ${WAVES:-4 5 6}; do
	rm -f build/rm_scanner.o build/rm_scan_inst_lean_flush.o build/rm_scan_inst_lean_concat_flush.o build/rm_scan_inst_lean_drain.o
	make -j16 CXXFLAGS="-O3 -std=c++17 -fPIC -Wall -pthread -I../../include -I. -DFLUSH_WAVES_PER_SIMD=$N $EXTRA" > /tmp/make_$N.log 2>&1 || { tail -5 /tmp/make_$N.log; exit 1; }
	echo "== FLUSH_WAVES_PER_SIMD $N"
	RNAMOTIF_TILE=${TILE:-0} RNAMOTIF_DBG=${DBG:-0} python3 $R/profiles/flush_try.py "$@" 2>&1 | grep -v "^\[dbg\] w" | tail -${TAIL:-6}
done
