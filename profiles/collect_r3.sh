#!/bin/bash
# profiles/collect_r3.sh -- everything profiles/r03_* is made from, in one call on the GPU box:
#   gpurun --timeout 1500 -- 'bash profiles/collect_r3.sh'
# then, here:  for l in trna pk1 qutr; do python3 profiles/summarize.py gpurun_out/prof_r3$l profiles/r03_$l; done
#              cp gpurun_out/r3/*.json gpurun_out/r3/*.txt profiles/   (as r03_*)
set -u
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
T=$R/tests/golden/test
mkdir -p $R/gpurun_out/r3
bash $R/profiles/collect.sh r3trna
bash $R/profiles/collect.sh r3pk1 --descr $T/pk1.descr
bash $R/profiles/collect.sh r3qutr --descr $T/qu+tr.descr
cd /tmp
for c in trna:$R/tests/golden/descr/trna.descr pk1:$T/pk1.descr qutr:$T/qu+tr.descr mpends:$T/mp.ends.descr ire:$T/ire.1.descr pkj12:$T/pk_j1+2.descr mixed:$T/qu+tr.descr,$T/mp.ends.descr; do
	python3 $R/bench.py --steps 20 --warmup 3 --cpu-bases 0 --descr ${c#*:} 2> /dev/null | grep '^{"metric"' > $R/gpurun_out/r3/cfg_${c%%:*}_100M.json
done
python3 $R/profiles/step_breakdown.py > $R/gpurun_out/r3/step_breakdown.txt 2>&1
python3 $R/profiles/lean_stages.py trna.descr mp.ends.descr ire.1.descr 2>&1 | grep '^==' > $R/gpurun_out/r3/lean_stages.txt
python3 $R/profiles/gen_counts.py > $R/gpurun_out/r3/gen_counts.txt 2>&1
python3 $R/profiles/phases.py trna.descr > $R/gpurun_out/r3/trna_phases.txt 2>&1
python3 $R/bench.py 2> $R/gpurun_out/r3/bench_final.err | grep '^{"metric"' > $R/gpurun_out/r3/bench_final.json
ls -la $R/gpurun_out/r3
