# profiles/cli_timeline.sh [records] -- raw RNAMOTIF_TIMING laps of the command line program over a
# synthetic database (default 1000 x 1 Mbase), from text and from a pack: one file per run under gpurun_out/.
set -e
N=${1:-1000}
B=$GRAFT_REPO_ROOT/rnamotif_amd/bin
export EFNDATA=$GRAFT_REPO_ROOT/rnamotif_amd/efndata
D=$GRAFT_REPO_ROOT/tests/golden/descr/trna.descr
python3 -c "
import sys; sys.path.insert(0, '$GRAFT_REPO_ROOT')
import rnamotif_amd as R
R.write_synthetic_fasta('/tmp/syn$N.fastn', $N)
"
$B/rnamotif_pack /tmp/syn$N.rmpk /tmp/syn$N.fastn
for i in 1 2; do
for what in fastn rmpk; do
	RNAMOTIF_TIMING=1 $B/rnamotif -descr $D /tmp/syn$N.$what > /tmp/o_$what.txt 2> $GRAFT_REPO_ROOT/gpurun_out/timeline_${what}_$i.txt
done
done
md5sum /tmp/o_fastn.txt /tmp/o_rmpk.txt
rm -f /tmp/syn$N.fastn /tmp/syn$N.rmpk
