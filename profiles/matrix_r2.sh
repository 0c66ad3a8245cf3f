# profiles/matrix_r2.sh -- on the GPU box: run-time switches of the general instance against each other on
# one device (devices differ by 10 % and more): RNAMOTIF_DBG bits 128 no level voting, 256 no continuations,
# 512 no chaining of levels within a step; RNAMOTIF_BUDGET iterations per step
for dbg in 0 512 128 640; do for bud in 32 64 128; do
    echo "== dbg=$dbg budget=$bud"
    python profiles/ablate.py --quick --reps 2 --dbg-or $dbg --env RNAMOTIF_BUDGET=$bud pk1.descr qu+tr.descr pk_j1+2.descr 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    d = json.loads(l); print('   ', d['descr'], d['all_ms'])"
done; done
