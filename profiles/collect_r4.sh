#!/bin/bash
# profiles/collect_r4.sh -- everything profiles/r04_* is made from, in one call on the GPU box:
#   gpurun --timeout 2400 -- 'bash profiles/collect_r4.sh'
# then, here:  for l in trna pk1 qutr; do python3 profiles/summarize.py gpurun_out/prof_r4$l profiles/r04_$l; done
#              python3 profiles/stalls.py profiles/r04_trna > profiles/r04_trna_stalls.txt
#              cp gpurun_out/r4c/* profiles/   (as r04_*)
set -u
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
T=$R/tests/golden/test
mkdir -p $R/gpurun_out/r4c
bash $R/profiles/collect.sh r4trna
bash $R/profiles/collect.sh r4pk1 --descr $T/pk1.descr
bash $R/profiles/collect.sh r4qutr --descr $T/qu+tr.descr
cd /tmp
for c in trna:$R/tests/golden/descr/trna.descr pk1:$T/pk1.descr qutr:$T/qu+tr.descr mpends:$T/mp.ends.descr ire:$T/ire.1.descr pkj12:$T/pk_j1+2.descr mixed:$T/qu+tr.descr,$T/mp.ends.descr; do
	python3 $R/bench.py --steps 20 --warmup 3 --cpu-bases 0 --descr ${c#*:} 2> /dev/null | grep '^{"metric"' > $R/gpurun_out/r4c/cfg_${c%%:*}_100M.json
done
# BASELINE config 5 at its own size on one GPU: both descriptors over one upload of the gigabase
python3 $R/bench.py --steps 5 --warmup 1 --cpu-bases 0 --records 1000 --descr $T/qu+tr.descr,$T/mp.ends.descr 2> /dev/null | grep '^{"metric"' > $R/gpurun_out/r4c/cfg_mixed_1G.json
python3 $R/profiles/step_breakdown.py > $R/gpurun_out/r4c/step_breakdown.txt 2>&1
python3 $R/profiles/lean_stages.py trna.descr mp.ends.descr ire.1.descr 2>&1 | grep '^==' > $R/gpurun_out/r4c/lean_stages.txt
python3 $R/profiles/flush_stages.py 2> /dev/null | grep '^==' > $R/gpurun_out/r4c/flush_stages_now.txt
python3 $R/profiles/drain_stats.py 2>&1 | grep -v 'wave cycles: ' > $R/gpurun_out/r4c/drain_stats.txt
python3 $R/profiles/gen_counts.py > $R/gpurun_out/r4c/gen_counts.txt 2>&1
python3 $R/profiles/phases.py trna.descr > $R/gpurun_out/r4c/trna_phases.txt 2>&1
python3 $R/profiles/quick_times.py --real > $R/gpurun_out/r4c/quick_times.txt 2>&1
python3 $R/bench.py 2> $R/gpurun_out/r4c/bench_final.err | grep '^{"metric"' > $R/gpurun_out/r4c/bench_final.json
ls -la $R/gpurun_out/r4c
