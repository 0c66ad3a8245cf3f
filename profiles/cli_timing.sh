# profiles/cli_timing.sh [records ...] -- the whole command line program on synthetic databases of
# the given numbers of 1 Mbase records (default 200 and 1000), from text and from a pack, with the
# laps RNAMOTIF_TIMING prints; "search" = scanner created .. search done.
set -e
cd /tmp
B=$GRAFT_REPO_ROOT/rnamotif_amd/bin
export EFNDATA=$GRAFT_REPO_ROOT/rnamotif_amd/efndata
D=$GRAFT_REPO_ROOT/tests/golden/descr/trna.descr
for N in ${*:-200 1000}; do
python3 -c "
import sys; sys.path.insert(0, '$GRAFT_REPO_ROOT')
import rnamotif_amd as R
R.write_synthetic_fasta('/tmp/syn$N.fastn', $N)
"
$B/rnamotif_pack /tmp/syn$N.rmpk /tmp/syn$N.fastn
for i in 1 2; do
for what in fastn rmpk; do
	echo "== $N Mbase, $what, run $i"
	( time RNAMOTIF_BATCH_BASES=${BB:-64000000} RNAMOTIF_TIMING=1 $B/rnamotif -descr $D /tmp/syn$N.$what > /tmp/o_$what.txt ) 2> /tmp/err.txt
	grep "real\|scanner created\|search done\|pack opened\|replay on" /tmp/err.txt
	python3 - <<PY
import re
t = open('/tmp/err.txt').read()
at = {m.group(1).strip(): float(m.group(2)) for m in re.finditer(r'\[timing\] ([a-z ]+?) +at +([0-9.]+) ms', t)}
s = (at['search done'] - at['scanner created']) * 1e-3
scans = [float(x) for x in re.findall(r'scan of \d+ entries: ([0-9.]+) ms', t)]
reads = [float(x) for x in re.findall(r'read (?:and packed|from the pack): ([0-9.]+) ms', t)]
reps = [float(x) for x in re.findall(r'replay of \d+ candidates: ([0-9.]+) ms', t)]
med = lambda v: sorted(v)[len(v) // 2] if v else 0
print('   scans', ' '.join('%.0f' % x for x in scans)); print('   reads', ' '.join('%.0f' % x for x in reads)); print('   replays', ' '.join('%.0f' % x for x in reps))
print('   search %.1f ms = %.2f Gbases/s; per batch (median): read(+pack) %.1f ms, scan %.1f ms, replay %.1f ms; %d batches' % (s * 1e3, $N * 1e-3 / s, med(reads), med(scans), med(reps), len(scans)))
PY
done
done
md5sum /tmp/o_fastn.txt /tmp/o_rmpk.txt
rm -f /tmp/syn$N.fastn /tmp/syn$N.rmpk
done
nproc
