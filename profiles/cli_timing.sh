set -e
cd /tmp
python3 -c "
import sys; sys.path.insert(0, '$GRAFT_REPO_ROOT')
import rnamotif_amd as R
R.write_synthetic_fasta('/tmp/syn200M.fastn', 200)
"
B=$GRAFT_REPO_ROOT/rnamotif_amd/bin
$B/rnamotif_pack /tmp/syn200M.rmpk /tmp/syn200M.fastn
export EFNDATA=$GRAFT_REPO_ROOT/rnamotif_amd/efndata
D=$GRAFT_REPO_ROOT/tests/golden/descr/trna.descr
for i in 1 2; do
echo "== text"; ( time RNAMOTIF_TIMING=1 $B/rnamotif -descr $D /tmp/syn200M.fastn > /tmp/o1.txt ) 2>&1 | grep "timing\|real"
echo "== text, 64M batches"; ( time RNAMOTIF_BATCH_BASES=64000000 RNAMOTIF_TIMING=1 $B/rnamotif -descr $D /tmp/syn200M.fastn > /tmp/o1b.txt ) 2>&1 | grep "timing\|real"
echo "== pack"; ( time RNAMOTIF_TIMING=1 $B/rnamotif -descr $D /tmp/syn200M.rmpk > /tmp/o2.txt ) 2>&1 | grep "timing\|real"
echo "== pack, 64M batches"; ( time RNAMOTIF_BATCH_BASES=64000000 RNAMOTIF_TIMING=1 $B/rnamotif -descr $D /tmp/syn200M.rmpk > /tmp/o2b.txt ) 2>&1 | grep "timing\|real"
done
md5sum /tmp/o1.txt /tmp/o1b.txt /tmp/o2.txt /tmp/o2b.txt; nproc
