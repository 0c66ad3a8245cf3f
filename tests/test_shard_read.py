"""CPU: a rank's share of a database without reading the rest (rma_database_index,
rma_pack_read_entries; rnamotif_amd/csrc/rm_capi.cpp, rm_stream.cpp read_entries, rm_pack.cpp
ensure_range) -- the entries a rank picks are what the whole-database reader delivers for them, from
text files and packed databases side by side, and files that can only be read whole say so."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_entries_of_text_and_pack_equal_the_whole_read(built, gbrna, tmp_path):
    import rnamotif_amd as R
    recs = R.read_fasta(gbrna)
    a = tmp_path / "a.fastn"
    a.write_bytes(b"".join(b">" + s + b" " + d + b"\n" + q + b"\n" for s, d, q in recs[:700]))
    bpk = tmp_path / "b.rmpk"
    R.Pack.write(str(bpk), recs[700:1500])
    c = tmp_path / "c.fastn"
    c.write_bytes(b"".join(b">" + s + b" " + d + b"\n" + q[:60] + b"\n" + q[60:] + b"\n" for s, d, q in recs[1500:1900]))
    files = [str(a), str(bpk), str(c)]
    ext = R.database_index(files)
    assert ext is not None and len(ext) == 1900
    whole = R.Pack.read(files)
    assert whole.count == 1900
    lens = whole.lengths()
    assert all(e >= n for e, n in zip(ext, lens)), "an extent bounds the entry's letters from above"
    assert ext[700:1500] == lens[700:1500], "a packed database knows the lengths"
    picks = [0, 1, 2, 350, 699, 700, 701, 702, 1100, 1499, 1500, 1501, 1899]
    part = R.Pack.read_entries(files, picks, threads=3)
    assert part is not None and part.count == len(picks)
    for k, i in enumerate(picks):
        assert part.record(k) == whole.record(i), i
    # a share of every entry: two ranks' picks are the database
    for rank in range(2):
        mine = list(range(rank, 1900, 2))
        part = R.Pack.read_entries(files, mine)
        assert part.bases == sum(lens[i] for i in mine)
        assert part.record(len(mine) - 1) == whole.record(mine[-1])
    assert R.Pack.read_entries(files, []).count == 0


def test_files_that_can_only_be_read_whole(built, gbrna, tmp_path):
    import rnamotif_amd as R
    recs = R.read_fasta(gbrna)[:50]
    plain = tmp_path / "p.fastn"
    plain.write_bytes(b"".join(b">" + s + b" " + d + b"\n" + q + b"\n" for s, d, q in recs))
    assert R.database_index([str(plain)], fmt="pir") is None          # the serial readers only
    assert R.Pack.read_entries([str(plain)], [1, 2], fmt="gb") is None
    odd = tmp_path / "o.fastn"
    odd.write_bytes(b">a one\nACGT\n>\nGGGG\n>c three\nTTTT\n")     # an unnamed entry: the serial reader's business
    assert R.database_index([str(odd)]) == [12, 7, 14]
    assert R.Pack.read_entries([str(odd)], [0, 2]) is not None
    assert R.Pack.read_entries([str(odd)], [1]) is None
    assert R.Pack.read_entries([str(plain)], [3, 4], maxslen=20) is None     # -N truncation


def test_partition_slices_covers_every_entry_and_slice_once():
    from rnamotif_amd.distributed import partition_slices
    ext = [100_000, 10, 0, 35_000, 7, 220_000]
    parts = partition_slices(ext, 4)
    seen = {}
    for p in parts:
        assert p == sorted(p)
        for e, j, k in p:
            seen.setdefault(e, []).append((j, k))
    for e, n in enumerate(ext):
        ks = {k for _, k in seen[e]}
        assert len(ks) == 1
        k = ks.pop()
        assert sorted(j for j, _ in seen[e]) == list(range(k))
    loads = [sum(ext[e] // k for e, _, k in p) for p in parts]
    assert max(loads) <= 1.3 * sum(ext) / 4
