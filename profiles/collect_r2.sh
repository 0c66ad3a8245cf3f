#!/bin/bash
# profiles/collect_r2.sh -- on the GPU box: everything profiles/r02_* is made from.
#   gpurun --timeout 1800 -- 'bash profiles/collect_r2.sh'
# then here:  for w in trna pk1 qutr; do python3 profiles/summarize.py gpurun_out/prof_r2_$w profiles/r02_$w; done
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
T=$R/tests/golden/test
bash $R/profiles/collect.sh r2_trna
bash $R/profiles/collect.sh r2_pk1 --descr $T/pk1.descr
bash $R/profiles/collect.sh r2_qutr --descr $T/qu+tr.descr
mkdir -p $R/gpurun_out/r2_cfg
cd $R
python3 bench.py --steps 20 --warmup 5 > gpurun_out/r2_cfg/trna_100M.json 2> gpurun_out/r2_cfg/trna_100M.err
python3 bench.py --steps 20 --warmup 5 --cpu-bases 1 --north-star-records 0 --descr $T/pk1.descr > gpurun_out/r2_cfg/pk1_100M.json 2>/dev/null
python3 bench.py --steps 20 --warmup 5 --cpu-bases 1 --north-star-records 0 --descr $T/qu+tr.descr,$T/mp.ends.descr > gpurun_out/r2_cfg/mixed_100M.json 2>/dev/null
python3 bench.py --steps 20 --warmup 5 --cpu-bases 1 --north-star-records 0 --descr $T/ire.descr > gpurun_out/r2_cfg/ire_100M.json 2>/dev/null
hipcc --offload-arch=gfx950 -O3 -o /tmp/valu_peak profiles/valu_peak.hip && /tmp/valu_peak > gpurun_out/r2_cfg/valu_peak.json
tail -c 600 gpurun_out/r2_cfg/*.json
