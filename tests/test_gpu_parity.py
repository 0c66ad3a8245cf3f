"""GPU parity: the HIP scanner, called through the C ABI, against the CPU
oracle and the reference's pinned outputs.  Run on the MI355X box: -m gpu."""
import hashlib
import os
import subprocess
import sys
import tempfile

import numpy as np
import pytest

import pins

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run_cli(built, workdir, args):
    env = dict(os.environ, EFNDATA=os.path.join(ROOT, "rnamotif_amd", "efndata"))
    p = subprocess.run([built["cli"]] + args + ["gbrna.111.0.fastn"], cwd=workdir, env=env,
                       stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=1800)
    assert p.returncode == 0, p.stderr.decode()
    return p.stdout


@pytest.mark.parametrize("name", sorted(pins.SLACK))
def test_cli_stdout_matches_reference_pin(built, workdir, name):
    """`rnamotif -descr X gbrna.111.0.fastn` on the GPU: byte-identical stdout."""
    out = _run_cli(built, workdir, ["-descr", name])
    nhits, md5 = pins.SLACK[name]
    assert out.count(b"\n>") + (1 if out.startswith(b">") else 0) == nhits
    assert hashlib.md5(out).hexdigest() == md5


@pytest.mark.parametrize("name", sorted(pins.STRICT))
def test_cli_strict_stdout_matches_reference_pin(built, workdir, name):
    out = _run_cli(built, workdir, pins.STRICT_ARGS + ["-descr", name + ".strict.descr"])
    nhits, md5 = pins.STRICT[name]
    assert hashlib.md5(out).hexdigest() == md5


@pytest.mark.parametrize("name", ["trna.efn.descr", "pk1.descr", "qu+tr.descr", "mp.ends.descr",
                                  "efn.descr", "getbest.descr", "pk_j1+2.descr"])
def test_hit_records_equal_oracle(built, workdir, gbrna, name):
    """Every candidate record (offsets, lengths, mispairs, mismatches, contexts,
    efn energies) equals the oracle's, in the same order: bit exact."""
    import rnamotif_amd as R
    from oracle_binding import oracle_scan
    cwd = os.getcwd()
    os.chdir(workdir)
    try:
        d = R.Descriptor(["-descr", name])
    finally:
        os.chdir(cwd)
    recs = R.read_fasta(gbrna)
    seqs = [r[2] for r in recs]
    sc = R.Scanner(d)
    db = sc.database(seqs)
    got = sc.scan(db)
    want = oracle_scan(d, seqs)
    assert got.shape == want.shape
    assert np.array_equal(got, want)


def test_ambiguity_and_short_sequences(built, workdir):
    """Ragged input: empty, shorter than the motif, all-N, N-rich and mixed-case
    IUPAC records; both strands."""
    import rnamotif_amd as R
    from oracle_binding import oracle_scan
    rng = np.random.default_rng(7)
    lut = np.frombuffer(b"acgt", dtype=np.uint8)
    seqs = [b"", b"acgu".replace(b"u", b"t"), b"n" * 500]
    for n in (19, 20, 21, 63, 64, 95, 96, 300, 5000, 70000):
        s = bytearray(lut[rng.integers(0, 4, size=n)].tobytes())
        for i in rng.integers(0, n, size=max(1, n // 37)):
            s[i] = ord("nryswkm"[int(rng.integers(0, 7))])
        seqs.append(bytes(s))
    cwd = os.getcwd()
    os.chdir(workdir)
    try:
        for name in ("mp.ends.descr", "trna.efn.descr", "pk1.descr"):
            d = R.Descriptor(["-descr", name])
            sc = R.Scanner(d)
            got = sc.scan(sc.database(seqs))
            want = oracle_scan(d, seqs)
            assert np.array_equal(got, want), name
    finally:
        os.chdir(cwd)


def test_syn10m_hit_counts(built, workdir):
    """The synthetic database of BASELINE.md: candidates on the first 10 Mbase
    equal the oracle's count; the reference reported 630 hits for trna."""
    import rnamotif_amd as R
    seqs = R.synthetic_records(10)
    cwd = os.getcwd()
    os.chdir(workdir)
    try:
        d = R.Descriptor(["-descr", "trna.efn.descr"])
    finally:
        os.chdir(cwd)
    sc = R.Scanner(d)
    hits = sc.scan(sc.database(seqs))
    assert hits.shape[0] == pins.SYN10M["trna.efn.descr"]
    # size independent properties: order, bounds, contiguity of the elements
    key = hits[:, :5]
    order = np.lexsort(key.T[::-1])
    assert np.array_equal(order, np.arange(len(hits)))
    off = hits[:, 5::4][:, :d.n_elems]
    ln = hits[:, 6::4][:, :d.n_elems]
    assert np.all(off[:, 1:] == off[:, :-1] + ln[:, :-1])
    total = ln.sum(axis=1)
    assert np.all((total >= d.minlen) & (total <= d.maxlen))
    assert np.all(off[:, 0] == hits[:, 2])


def _descr(workdir, name, extra=()):
    import rnamotif_amd as R
    cwd = os.getcwd()
    os.chdir(workdir)
    try:
        return R.Descriptor(list(extra) + ["-descr", name])
    finally:
        os.chdir(cwd)


class _env:
    """Environment variables for the duration of a with block (launch-shape overrides)."""

    def __init__(self, **kw):
        self.kw = kw

    def __enter__(self):
        self.old = {k: os.environ.get(k) for k in self.kw}
        for k, v in self.kw.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = str(v)

    def __exit__(self, *a):
        for k, v in self.old.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v


SYN_DESCR = ["trna.descr", "pk1.descr", "pk_j1+2.descr", "qu+tr.descr", "ire.descr", "mp.ends.descr",
             "bulge.descr", "nanlin.descr", "score.1.descr", "sprintf.descr"]


@pytest.mark.parametrize("name", SYN_DESCR)
def test_synthetic_records_equal_oracle(built, workdir, name):
    """Random sequence (no biology: the pre-filters see their worst case mix of
    near misses) -- records bit-identical to the oracle, record boundaries that
    are not multiples of the tile or of 32 bases."""
    import rnamotif_amd as R
    from oracle_binding import oracle_scan
    rng = np.random.default_rng(11)
    lut = np.frombuffer(b"acgt", dtype=np.uint8)
    seqs = [lut[rng.integers(0, 4, size=n)].tobytes() for n in (300_001, 4097, 2048, 2047, 131_071, 33)]
    d = _descr(workdir, name)
    sc = R.Scanner(d)
    got = sc.scan(sc.database(seqs))
    want = oracle_scan(d, seqs)
    assert got.shape == want.shape
    assert np.array_equal(got, want)


@pytest.mark.parametrize("dbg", [4, 8, 16, 28])
@pytest.mark.parametrize("name", ["trna.descr", "pk1.descr", "qu+tr.descr"])
def test_search_paths_agree(built, workdir, name, dbg):
    """The kernel's optional stages (4: bit-parallel pre-filter off, 8: literal
    filter off, 16: LDS-record search off -> general state machine) are output
    neutral: every combination gives the oracle's records."""
    import rnamotif_amd as R
    from oracle_binding import oracle_scan
    rng = np.random.default_rng(12)
    lut = np.frombuffer(b"acgt", dtype=np.uint8)
    seqs = [lut[rng.integers(0, 4, size=n)].tobytes() for n in (150_000, 5000)]
    d = _descr(workdir, name)
    want = oracle_scan(d, seqs)
    sc = R.Scanner(d)
    db = sc.database(seqs)
    sc.set_option("dbg", dbg)          # (the environment is read once, at the scanner's creation)
    got = sc.scan(db)
    assert np.array_equal(got, want)


@pytest.mark.parametrize("name", ["trna.descr", "ire.descr", "mp.ends.descr", "bulge.descr", "score.1.descr", "sprintf.descr"])
def test_pooled_and_tile_by_tile_pass_b_agree(built, workdir, gbrna, name):
    """Lean descriptors whose window fits a lane's column of LDS are searched by the pooled instance
    (survivors of many tiles together, windows rebuilt from the packed database); RNAMOTIF_POOL=0 is
    pass B tile by tile.  Same records either way -- with small pools (sessions between most tiles),
    N-rich entries, entries shorter than the window, slices of start positions."""
    import rnamotif_amd as R
    from oracle_binding import oracle_scan
    rng = np.random.default_rng(31)
    lut = np.frombuffer(b"acgtn", dtype=np.uint8)
    seqs = [lut[rng.integers(0, 4, size=n)].tobytes() for n in (200_003, 61, 7, 40_000)]
    seqs.append(lut[rng.choice(5, size=30_000, p=[.22, .22, .22, .22, .12])].tobytes())
    seqs += [r[2] for r in R.read_fasta(gbrna)[:400]]
    d = _descr(workdir, name)
    want = oracle_scan(d, seqs)
    sc = R.Scanner(d)
    db = sc.database(seqs)
    # drain 0: every workgroup walks its own items; dbg 2097152: the drain kernel's items stay whole (no pieces);
    # 4194304: no subtrees handed to idle lanes; 8388608: everything a workgroup holds at the end goes to the list
    # glist: items of the drain kernel's list (7: it overflows at once and the workgroups that find no room walk their own items)
    base = {"pool": -1, "pool_min": 1024, "pool_refill": 48, "drain": 1, "dbg": 0, "glist": 0}
    for opts in ({}, {"pool": 0}, {"pool_min": 8, "pool_refill": 1}, {"pool_min": 100000}, {"drain": 0}, {"dbg": 2097152},
                 {"dbg": 4194304}, {"dbg": 8388608}, {"dbg": 8388608 + 4194304 + 2097152}, {"dbg": 8388608, "pool_refill": 1},
                 {"dbg": 8388608, "glist": 7}, {"glist": 1}, {"dbg": 8388608, "glist": 300}):
        for k, v in dict(base, **opts).items():
            sc.set_option(k, v)
        got = sc.scan(db)
        assert got.shape == want.shape and np.array_equal(got, want), opts
    # slices of start positions (what a rank of a multi-GPU job holds)
    sl = sc.database(seqs, ranges=[(0, 100_000), (10, 61), (0, 7), (20_000, 40_000), (5, 29_000)] + [(0, len(s)) for s in seqs[5:]])
    ref = None
    for opts in ({"pool": 0}, {}):
        for k, v in dict(base, **opts).items():
            sc.set_option(k, v)
        got = sc.scan(sl)
        if ref is None:
            ref = got
        assert np.array_equal(got, ref)
    assert 0 < ref.shape[0] <= want.shape[0] or want.shape[0] == 0


@pytest.mark.parametrize("name", ["trna.descr", "bulge.descr", "ire.descr", "mp.ends.descr"])
def test_search_kernel_that_walks_nothing(built, workdir, gbrna, name):
    """RMK_LEAN_FLUSH (round 4; option `flush`: by default where the descriptor has a look-ahead chain): every survivor of
    pass A' goes to the drain kernel's list, the search kernel keeps no pool and walks nothing.  What has no room -- a tile's
    items beyond queue and spill area, the list's items beyond its end -- is reported and the scan repeated with room for
    it (search_finish), where the other pooled instance searches in place.  Records equal to the oracle's with the
    instance forced on and off, a list of one item, a queue of 64 and no spill area, small tiles, and the energy kernel
    in both its forms."""
    import rnamotif_amd as R
    from oracle_binding import oracle_scan
    rng = np.random.default_rng(47)
    lut = np.frombuffer(b"acgtn", dtype=np.uint8)
    seqs = [lut[rng.integers(0, 4, size=n)].tobytes() for n in (300_007, 61, 7, 40_000)]
    seqs.append(lut[rng.choice(5, size=30_000, p=[.22, .22, .22, .22, .12])].tobytes())
    seqs += [b"".join(r[2] for r in R.read_fasta(gbrna)[:300])]        # (one long entry of real RNA: clustered survivors)
    d = _descr(workdir, name)
    want = oracle_scan(d, seqs)
    assert want.shape[0] > 0 or name == "ire.descr"
    for env in ({}, {"RNAMOTIF_QCAP": "64", "RNAMOTIF_SPILL": "0"}, {"RNAMOTIF_TILE": "512", "RNAMOTIF_QCAP": "64", "RNAMOTIF_SPILL": "16"}):
        with _env(RNAMOTIF_SHORT="0", **env):
            sc = R.Scanner(d)
            db = sc.database(seqs)
            for opts in ({"flush": 1}, {"flush": 0}, {"flush": -1}, {"flush": 1, "glist": 1}, {"flush": 1, "efn_light": 0},
                         {"flush": 0, "efn_light": 1}, {"flush": 1, "dbg": 2097152}, {"flush": 1, "dbg": 4194304}):
                for k, v in dict({"flush": -1, "glist": 0, "efn_light": -1, "dbg": 0}, **opts).items():
                    sc.set_option(k, v)
                got = sc.scan(db)
                assert got.shape == want.shape and np.array_equal(got, want), (env, opts)
                # (a second scan: the areas the first one had to grow are there now)
                assert np.array_equal(sc.scan(db), want), (env, opts)
            db.close()
            sc.close()


def test_tile_sizes_agree(built, workdir):
    """Tile size is a launch parameter only: 256..8192 start positions per
    workgroup give identical records."""
    import rnamotif_amd as R
    rng = np.random.default_rng(13)
    lut = np.frombuffer(b"acgt", dtype=np.uint8)
    seqs = [lut[rng.integers(0, 4, size=n)].tobytes() for n in (400_000, 777)]
    d = _descr(workdir, "trna.descr")
    res = []
    old = os.environ.get("RNAMOTIF_TILE")
    try:
        for t in (256, 1024, 2048, 8192):
            os.environ["RNAMOTIF_TILE"] = str(t)
            sc = R.Scanner(d)
            res.append(sc.scan(sc.database(seqs)))
    finally:
        if old is None:
            os.environ.pop("RNAMOTIF_TILE", None)
        else:
            os.environ["RNAMOTIF_TILE"] = old
    for r in res[1:]:
        assert np.array_equal(res[0], r)


@pytest.mark.parametrize("name", ["trna.descr", "mp.ends.descr", "bulge.descr", "ire.descr", "score.1.descr", "efn.descr",
                                  "pk1.descr", "qu+tr.descr", "pk_j1+2.descr", "nanlin.descr"])
def test_grouped_tiles_agree(built, workdir, gbrna, name):
    """Databases of short entries are searched in groups of small tiles that share one work
    queue (rma_search_kernel<.., G>), or -- descriptors the pooled lean instance takes -- in tiles over the
    concatenation of the entries (Layout::concat); the choice is a launch shape only.  The reference's test
    database (4000 entries of 560 bases on average) and random entries of awkward lengths, one
    tile per pass against grouped, with an ample queue and with one that overflows."""
    import rnamotif_amd as R
    from oracle_binding import oracle_scan
    d = _descr(workdir, name)
    rng = np.random.default_rng(14)
    lut = np.frombuffer(b"acgtn", dtype=np.uint8)
    lens = [0, 1, 33, 70, 71, 72, 73, 200, 511, 512, 513, 767, 768, 769, 1023, 1024, 1025, 1100, 2047, 2049, 40_000]
    syn = [lut[rng.choice(5, size=n, p=[0.25, 0.25, 0.25, 0.24, 0.01])].tobytes() for n in lens * 4]
    for seqs, check_oracle in (([r[2] for r in R.read_fasta(gbrna)], False), (syn, True)):
        res = []
        # ("2": tiles over the concatenation of the entries -- round 4, the pooled lean instance; the queue of 64 with a spill
        # area of 0 / 32 items sends items through the search in place, from the tile's bytes in their entry's coordinates)
        # (RNAMOTIF_FLUSH: the pooled instance that walks nothing, forced off and on -- by default descriptors with a look-ahead
        # chain take it; what its queue and spill area do not hold makes the scan repeat with room for it)
        for short, qcap, spill, flush in (("0", None, None, None), ("1", None, None, None), ("1", 64, None, None), ("1", 64, 0, None),
                                          ("0", 64, 32, None), ("2", None, None, None), ("2", 64, 32, None), ("2", 64, 0, None),
                                          ("2", None, None, "0"), ("2", 64, 0, "0"), ("2", None, None, "1"), ("2", 64, 0, "1"), ("0", 64, 0, "1")):
            with _env(RNAMOTIF_SHORT=short, RNAMOTIF_QCAP=qcap, RNAMOTIF_SPILL=spill, RNAMOTIF_FLUSH=flush):
                sc = R.Scanner(d)
                res.append(sc.scan(sc.database(seqs)))
        assert res[0].shape[0] > 0 or not check_oracle or name in ("ire.descr", "pk_j1+2.descr", "qu+tr.descr", "nanlin.descr")
        for r in res[1:]:
            assert r.shape == res[0].shape and np.array_equal(r, res[0])
        if check_oracle:
            want = oracle_scan(d, seqs)
            assert want.shape == res[0].shape and np.array_equal(want, res[0])


@pytest.mark.parametrize("seed", [52490, 50964, 53200, 50360])
def test_concatenation_tiles_items_in_the_padding(built, tmp_path, seed):
    """Found by tests/fuzz_campaign.py `concat` (round 4): a general instance over tiles that lie over the concatenation of the
    entries pops its items 64 at a time and brings each to its entry; starts in the padding between two entries belong to
    none, and they stand one after the other in the queue -- a round whose items ALL fell out ended the tile's search
    (general_pass_b: `no lane has work` is not `no lane can get any`).  The campaign's descriptors and databases, as they were."""
    import rnamotif_amd as R
    from oracle_binding import oracle_scan
    rng = np.random.default_rng(1000 + seed)
    text = _random_descriptor(rng)
    path = tmp_path / "f.descr"
    path.write_text(text)
    d = R.Descriptor(["-descr", str(path)])
    s = _planted_sequence(rng, 6000)
    cuts = np.sort(rng.integers(0, len(s), size=int(rng.integers(5, 40))))
    seqs = [s[a:b] for a, b in zip(np.r_[0, cuts], np.r_[cuts, len(s)])] + [s[:1030], s[:1024 + d.maxlen], b""]
    want = oracle_scan(d, seqs)
    assert want.shape[0] > 0
    for short in ("2", "0"):
        with _env(RNAMOTIF_SHORT=short):
            sc = R.Scanner(d)
            got = sc.scan(sc.database(seqs))
        assert got.shape == want.shape and np.array_equal(got, want), (short, text)


SYN10M_FIRST_TRNA = (b"syn0000         1.981  -12.300 0   59767   82 cgagcc tt ctt taca gag a catg acggaac catg "
                     b"caatccggcaccggagtgaga aggct gttgggc agtct ggcttg catg")


@pytest.mark.parametrize("name", sorted(pins.SYN10M))
def test_syn10m_cli_hit_counts(built, workdir, tmp_path_factory, name):
    """The reference's own numbers on syn10M (measured with the
    unmodified reference sources): hits printed by the command line program on
    the same file -- trna 630, pk1 1039, qu+tr 155, mp.ends 36, ire 3."""
    import rnamotif_amd as R
    d = tmp_path_factory.getbasetemp() / "syn10M"
    d.mkdir(exist_ok=True)
    fa = d / "syn10M.fastn"
    if not fa.exists():
        assert R.write_synthetic_fasta(str(fa), 10) == "d33c2542e515346e1d0fdfc9edcc5658"
    env = dict(os.environ, EFNDATA=os.path.join(ROOT, "rnamotif_amd", "efndata"))
    p = subprocess.run([built["cli"], "-descr", name, str(fa)], cwd=workdir, env=env,
                       stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=1800)
    assert p.returncode == 0, p.stderr.decode()
    lines = p.stdout.split(b"\n")
    assert sum(1 for l in lines if l.startswith(b">")) == pins.SYN10M[name]
    if name == "trna.efn.descr":
        first = [l for l in lines if l.startswith(b"syn")][0]
        assert b" ".join(first.split()) == b" ".join(SYN10M_FIRST_TRNA.split())


def _sorted_unique(h):
    key = h[:, :5]
    order = np.lexsort(key.T[::-1])
    assert np.array_equal(order, np.arange(len(order))), "records are not in (seq, comp, szero, rank, order) order"
    assert len(np.unique(key, axis=0)) == len(key)


@pytest.mark.parametrize("name", ["trna.efn.descr", "mp.ends.descr", "pk1.descr"])
def test_full_size_properties(built, workdir, name):
    """BASELINE.json's headline size (100 records of 1 Mbase, the bench workload; pk1.descr is
    config 3), where the oracle would take minutes: properties that hold at any size.  (1) Shards
    add up: the two halves of the database scanned separately are the whole scan.  (2) Strands
    mirror: the reverse complement of every record gives the same candidates with the strand flag
    flipped (offsets in a record are strand local).  (3) Sorted, no duplicates."""
    import rnamotif_amd as R
    d = _descr(workdir, name)
    seqs = R.synthetic_records(100)
    sc = R.Scanner(d)
    whole = sc.scan(sc.database(seqs))
    assert whole.shape[0] > 100
    _sorted_unique(whole)
    # (1)
    a = sc.scan(sc.database(seqs[:50]))
    b = sc.scan(sc.database(seqs[50:]))
    b[:, 0] += 50
    assert np.array_equal(np.concatenate([a, b]), whole)
    # (2)
    tr = bytes.maketrans(b"acgt", b"tgca")
    rc = [s.translate(tr)[::-1] for s in seqs]
    del seqs
    mirror = sc.scan(sc.database(rc))
    mirror[:, 1] ^= 1
    mirror = mirror[np.lexsort(mirror[:, :5].T[::-1])]
    assert np.array_equal(mirror, whole)


@pytest.mark.parametrize("name", ["trna.efn.descr", "pk1.descr", "qu+tr.descr", "mp.ends.descr", "ire.descr"])
def test_full_size_records_equal_oracle(built, workdir, name):
    """BASELINE.json's configs at their own size -- 100 records of 1 Mbase, the bench workload -- against the
    oracle over the WHOLE database, every start position of both strands (RM_find_motif's loop,
    find_motif.c:164-207), not a sample: the records of the HIP scan equal the oracle's bit for bit, energies
    included.  The oracle runs one process per host core, a record each (tests/oracle_pool.py); the bench line's
    candidate counts (trna 5 837, pk1 10 400, qu+tr 1 037) are pinned here."""
    import rnamotif_amd as R
    from oracle_pool import concat_records, oracle_records
    d = _descr(workdir, name)
    records = list(range(100))
    want_by_rec, info = oracle_records(["-descr", name], records, cwd=workdir)
    want = concat_records(want_by_rec, records, d.hit_stride)
    seqs = R.synthetic_records(100)
    sc = R.Scanner(d)
    got = sc.scan(sc.database(seqs))
    print(f"{name}: {got.shape[0]} candidates; oracle {info['cpu_s']:.0f} core-seconds on {info['procs']} processes, wall {info['wall_s']:.1f} s")
    assert got.shape == want.shape, (got.shape, want.shape)
    assert np.array_equal(got, want)
    expected = {"trna.efn.descr": 5837, "pk1.descr": 10400, "qu+tr.descr": 1037}
    if name in expected:
        assert got.shape[0] == expected[name]


def test_full_size_mixed_batch_on_one_database(built, workdir):
    """BASELINE config 5's single-GPU form: qu+tr.descr and mp.ends.descr over ONE upload of the 100
    Mbase database (a database belongs to a device, not to a descriptor), their kernels side by
    side on two streams.  Same records as each descriptor alone on a database of its own; shards
    add up; strands mirror; sorted, no duplicates."""
    import rnamotif_amd as R
    names = ("qu+tr.descr", "mp.ends.descr")
    ds = [_descr(workdir, n) for n in names]
    seqs = R.synthetic_records(100)
    scs = [R.Scanner(d) for d in ds]
    shared = scs[0].database(seqs)
    for sc in scs:
        sc.attach(shared)
    for sc in scs:
        sc.scan_begin(shared)
    together = [sc.scan_end() for sc in scs]
    tr = bytes.maketrans(b"acgt", b"tgca")
    for d, sc, got in zip(ds, scs, together):
        assert got.shape[0] > 100
        _sorted_unique(got)
        own = R.Scanner(d)
        alone = own.scan(own.database(seqs))
        assert np.array_equal(alone, got)
        a = own.scan(own.database(seqs[:37]))
        b = own.scan(own.database(seqs[37:]))
        b[:, 0] += 37
        assert np.array_equal(np.concatenate([a, b]), got)
    rc = scs[0].database([s.translate(tr)[::-1] for s in seqs])
    for sc, got in zip(scs, together):
        mirror = sc.scan(rc)
        mirror[:, 1] ^= 1
        mirror = mirror[np.lexsort(mirror[:, :5].T[::-1])]
        assert np.array_equal(mirror, got)


def test_one_gbase_is_the_sum_of_its_slices(built, workdir):
    """The north star's size -- trna.descr over 1000 records of 1 Mbase on one GPU, the figure
    bench.py reports as north_star_1gbase: its records are the ten 100-record slices' records, one
    after the other (the first slice is the headline workload), sorted, no duplicates; a hundred records
    drawn across the slices equal the oracle.  Then BASELINE config 5 over the same gigabase in HBM."""
    import rnamotif_amd as R
    sys.path.insert(0, ROOT)
    from bench import synthetic_slice
    d = _descr(workdir, "trna.efn.descr")
    sc = R.Scanner(d)
    parts = []
    with tempfile.TemporaryDirectory() as tmp:
        pk = os.path.join(tmp, "g.rmpk")
        recs = []
        for k in range(10):
            seqs = synthetic_slice(100 * k, 100, 1_000_000)
            if k == 0:
                assert seqs[:3] == R.synthetic_records(3)          # (the jump into the stream lands where the stream is)
            h = sc.scan(sc.database(seqs))
            h[:, 0] += 100 * k
            parts.append(h)
            recs += [(b"syn%04d" % (100 * k + i), b"", s) for i, s in enumerate(seqs)]
        del seqs
        R.Pack.write(pk, recs)
        del recs
        pack = R.Pack(pk)
    big = sc.database_from_pack(pack)
    whole = sc.scan(big)
    assert pack.bases == 1_000_000_000 and whole.shape[0] > 50_000
    _sorted_unique(whole)
    assert np.array_equal(np.concatenate(parts), whole)
    # ... and a hundred of its records, ten from every slice, are the oracle's bit for bit
    from oracle_pool import oracle_records
    sample = sorted(100 * k + (7 * k + 10 * j + 3) % 100 for k in range(10) for j in range(10))
    assert len(set(sample)) == 100
    want, info = oracle_records(["-descr", "trna.efn.descr"], sample, cwd=workdir)
    n_cmp = 0
    for k in sample:
        got_k = whole[whole[:, 0] == k].copy()
        got_k[:, 0] = 0
        assert got_k.shape == want[k].shape and np.array_equal(got_k, want[k]), k
        n_cmp += got_k.shape[0]
    assert n_cmp > 5000
    # BASELINE config 5 at its own size on one GPU: qu+tr.descr and mp.ends.descr over the SAME upload of the
    # gigabase, kernels side by side; equal to each descriptor's ten slices, and fifty records each to the oracle
    names = ("qu+tr.descr", "mp.ends.descr")
    ds = [_descr(workdir, n) for n in names]
    scs = [R.Scanner(dd) for dd in ds]
    for s_ in scs:
        s_.attach(big)
    for s_ in scs:
        s_.scan_begin(big)
    together = [s_.scan_end() for s_ in scs]
    big.close()
    sample5 = sample[::2]
    for n, dd, s_, got in zip(names, ds, scs, together):
        _sorted_unique(got)
        assert got.shape[0] > 1000
        sl = []
        for k in range(10):
            h = s_.scan(s_.database(synthetic_slice(100 * k, 100, 1_000_000)))
            h[:, 0] += 100 * k
            sl.append(h)
        assert np.array_equal(np.concatenate(sl), got), n
        want5, _ = oracle_records(["-descr", n], sample5, cwd=workdir)
        for k in sample5:
            got_k = got[got[:, 0] == k].copy()
            got_k[:, 0] = 0
            assert got_k.shape == want5[k].shape and np.array_equal(got_k, want5[k]), (n, k)


def test_start_position_ranges(built, workdir):
    """rma_db_create_ranges: slices of an entry's start positions searched separately (as
    different GPUs would) add up to the whole entry's records, cut anywhere."""
    import rnamotif_amd as R
    rng = np.random.default_rng(21)
    lut = np.frombuffer(b"acgt", dtype=np.uint8)
    seqs = [lut[rng.integers(0, 4, size=n)].tobytes() for n in (300_000, 50, 4_100)]
    for name in ("trna.descr", "pk1.descr"):
        d = _descr(workdir, name)
        sc = R.Scanner(d)
        whole = sc.scan(sc.database(seqs))
        cuts = [0, 1, 2047, 2048, 2049, 77_777, 150_000, 299_950, 300_000]
        parts = []
        for lo, hi in zip(cuts, cuts[1:]):
            parts.append(sc.scan(sc.database(seqs, ranges=[(lo, hi), (lo, hi), (lo, hi)])))
        allh = np.concatenate(parts, axis=0)
        allh = allh[np.lexsort(allh[:, :5].T[::-1])]
        assert whole.shape[0] > 0 and np.array_equal(allh, whole), name
        empty = sc.scan(sc.database(seqs, ranges=[(5, 5), (0, 0), (4_100, 9_000)]))
        assert empty.shape[0] == 0


def test_mrnamotif_single_rank_equals_cli(built, workdir, gbrna, tmp_path):
    """python -m rnamotif_amd.mrnamotif (the multi-GPU command line, here with one rank):
    same bytes on stdout as bin/rnamotif."""
    import sys
    env = dict(os.environ, EFNDATA=os.path.join(ROOT, "rnamotif_amd", "efndata"), PYTHONPATH=ROOT)
    want = _run_cli(built, workdir, ["-descr", "trna.efn.descr"])
    p = subprocess.run([sys.executable, "-m", "rnamotif_amd.mrnamotif", "-descr", "trna.efn.descr", "gbrna.111.0.fastn"],
                       cwd=workdir, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=900)
    assert p.returncode == 0, p.stderr.decode()[-2000:]
    assert p.stdout == want
    assert b"complete descr length: min/max = 63/95" in p.stderr


@pytest.mark.parametrize("name", ["trna.efn2.descr", "hairpin.efn2.descr", "hairpin.nostdbp.descr"])
def test_efn2_sites_equal_oracle(built, workdir, gbrna, name):
    """efn2() on the device (rm_efn2_core.h: coaxial stacking, 1x1/2x1/2x2 interior loops from
    global tables) beside efn(): every record, energies included, equals the oracle's.
    hairpin.nostdbp.descr: efn_usestdbp = 0 -- both functions pair the bases by the helix' own pair set, one
    with g:a in it (setbp, score.c:3245; rm_efn_core.h rme_setup): half the records' energies change."""
    import rnamotif_amd as R
    from oracle_binding import oracle_scan
    d = R.Descriptor(["-descr", os.path.join(ROOT, "tests", "data", name)])
    assert d.efn2data
    recs = R.read_fasta(gbrna)
    seqs = [r[2] for r in recs][:4067 if name.startswith("trna") else 60 if "nostdbp" in name else 400]
    sc = R.Scanner(d)
    got = sc.scan(sc.database(seqs))
    want = oracle_scan(d, seqs)
    assert want.shape[0] > 1000
    assert got.shape == want.shape and np.array_equal(got, want)
    e2 = got[:, d.efn_off]
    if "nostdbp" in name:
        # the switch matters: with the standard pairs the same candidates get other energies
        text = open(os.path.join(ROOT, "tests", "data", name)).read().replace("efn_usestdbp = 0", "efn_usestdbp = 1")
        with tempfile.TemporaryDirectory() as tmp:
            open(os.path.join(tmp, "std.descr"), "w").write(text)
            d1 = R.Descriptor(["-descr", os.path.join(tmp, "std.descr")])
        sc1 = R.Scanner(d1)
        std = sc1.scan(sc1.database(seqs))
        assert std.shape == got.shape and np.array_equal(std[:, :d.efn_off], got[:, :d.efn_off])
        assert np.array_equal(std, oracle_scan(d1, seqs))
        assert (std[:, d.efn_off:] != got[:, d.efn_off:]).any(axis=1).mean() > 0.2
    else:
        assert np.all(np.abs(e2) < 100000)          # closed structures: always defined


def test_helices_of_64_to_127_base_pairs(built, tmp_path):
    """Helices longer than 63 base pairs (refused until round 3): the general instance whose sets of helix lengths are two
    words (rm_scan_core.h rmd_lset_t, RMD_KIND_WIDE; the reference keeps a helix' candidates in h3[ 101 ], find_motif.c:406).
    Planted hairpins of 63 to 127 base pairs in random sequence, with and without a mispair: records equal the oracle's."""
    import rnamotif_amd as R
    from oracle_binding import oracle_scan
    rng = np.random.default_rng(5)
    lut = np.frombuffer(b"acgt", dtype=np.uint8)
    comp = bytes.maketrans(b"acgt", b"tgca")
    rnd = lambda n: lut[rng.integers(0, 4, size=n)].tobytes()        # noqa: E731
    seqs = []
    for hl, loop in ((70, 5), (64, 4), (100, 6), (63, 3), (90, 8), (127, 4)):
        stem = rnd(hl)
        seqs.append(rnd(30) + stem + rnd(loop) + stem.translate(comp)[::-1] + rnd(25))
    seqs.append(rnd(3000))
    for text, least in (("descr\n\th5(minlen=20,maxlen=110,mispair=1)\n\t\tss(minlen=3,maxlen=8)\n\th3\n", 1000),
                        ("parms\n\twc += gu;\ndescr\n\th5(minlen=60,maxlen=127)\n\t\tss(minlen=3,maxlen=8)\n\th3\n", 500),
                        # (a pseudoknot whose first helix may be that long)
                        ("descr\n\th5(tag='a',minlen=30,maxlen=80)\n\t\tss(minlen=1,maxlen=3)\n\th5(tag='b',minlen=3,maxlen=4)\n"
                         "\t\tss(minlen=0,maxlen=3)\n\th3(tag='a')\n\t\tss(minlen=1,maxlen=30)\n\th3(tag='b')\n", 500)):
        (tmp_path / "wide.descr").write_text(text)
        d = R.Descriptor(["-descr", str(tmp_path / "wide.descr")])
        sc = R.Scanner(d)
        got = sc.scan(sc.database(seqs))
        want = oracle_scan(d, seqs)
        assert want.shape[0] >= least
        assert got.shape == want.shape and np.array_equal(got, want)


def test_energy_calls_over_many_helices(built, tmp_path):
    """efn() and efn2() over seventeen hairpins side by side (refused until round 3: "at most 15"): the energy kernel's
    instance with stacks for fifty helices (rm_efn_core.h rme_ctx_t<.., BIG>, rmd_program_t::efn_big); records, energies
    included, equal the oracle's -- RM_efn / RM_efn2 of the reference have no such bound (efn.c:1162, efn2.c:1103)."""
    import rnamotif_amd as R
    from oracle_binding import oracle_scan
    text = ("parms\n\twc += gu;\ndescr\n" +
            "".join("\th5(tag='h%d',minlen=2,maxlen=3)\n\t\tss(len=3)\n\th3(tag='h%d')\n\tss(minlen=1,maxlen=2)\n" % (i, i) for i in range(17)) +
            "score\n\t{ e2 = efn2( h5['h0'], h3['h16'] ); SCORE = efn( h5['h0'], h3['h16'] ); }\n")
    (tmp_path / "many.descr").write_text(text)
    d = R.Descriptor(["-descr", str(tmp_path / "many.descr")])
    rng = np.random.default_rng(31)
    lut = np.frombuffer(b"acgt", dtype=np.uint8)
    flank = lambda n: lut[rng.integers(0, 4, size=n)].tobytes()                      # noqa: E731
    units = [b"gcaaagca", b"ggcttagtca", b"gtaaaaca", b"cgagacgaa", b"ggaaauca".replace(b"u", b"t")]
    seqs = []
    for k in range(4):
        body = b"".join(units[int(i)] for i in rng.integers(0, len(units), size=17))
        seqs.append(flank(40 + 7 * k) + body + flank(30))
    sc = R.Scanner(d)
    got = sc.scan(sc.database(seqs))
    want = oracle_scan(d, seqs)
    assert want.shape[0] >= 4
    assert got.shape == want.shape and np.array_equal(got, want)
    assert np.all(np.abs(got[:, d.efn_off + 1]) < 16000)          # efn() of seventeen hairpins side by side: defined


def test_cli_efn2_equals_oracle_cli(built, workdir):
    """The command line program with an efn2() score: same bytes as the oracle-backed program
    (whose efn2 column is checked against the reference's efn2_drv in tests/test_efn2_oracle.py)."""
    env = dict(os.environ, EFNDATA=os.path.join(ROOT, "rnamotif_amd", "efndata"))
    descr = os.path.join(ROOT, "tests", "data", "trna.efn2.descr")
    outs = []
    for exe in (built["cli"], built["oracle_cli"]):
        p = subprocess.run([exe, "-descr", descr, "gbrna.111.0.fastn"], cwd=workdir, env=env,
                           stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=1800)
        assert p.returncode == 0, p.stderr.decode()
        outs.append(p.stdout)
    assert outs[0] == outs[1] and outs[0].count(b"\n>") == 1351


def _corpus():
    d = os.path.join(ROOT, "tests", "golden", "descr")
    return sorted(f for f in os.listdir(d) if f.endswith(".descr"))


@pytest.mark.parametrize("name", _corpus())
def test_descriptor_corpus_equals_oracle(built, gbrna, name):
    """Every descriptor of the authors' descr/ corpus (parallel helices, triplexes, 4-plexes,
    pseudoknots of all shapes, mismatches, pairfrac, sites, score programs): candidate records
    on real and on random sequence equal the oracle's, or the descriptor is refused with a
    message (syntax error by design: ps.3; beyond a documented device limit)."""
    import rnamotif_amd as R
    from oracle_binding import oracle_scan
    path = os.path.join(ROOT, "tests", "golden", "descr", name)
    cwd = os.getcwd()
    os.chdir(os.path.dirname(path))
    try:
        try:
            d = R.Descriptor(["-descr", name])
        except R.RnamotifError as e:
            assert name in ("ps.3.descr", "rs.bad.descr", "rs.bad.2.descr") or "efn" in str(e), (name, str(e))
            return
    finally:
        os.chdir(cwd)
    try:
        sc = R.Scanner(d)
    except R.RnamotifError as e:
        pytest.skip("refused by the device build: " + str(e))
    rng = np.random.default_rng(31)
    lut = np.frombuffer(b"acgt", dtype=np.uint8)
    recs = R.read_fasta(gbrna)
    if d.maxlen > 1500:
        # unbounded interiors (ss without maxlen under the 6000 base window): the search is
        # quadratic in the window for the reference too -- minutes per kilobase -- so short entries
        seqs = [r[2][:160] for r in recs[:6]] + [lut[rng.integers(0, 4, size=200)].tobytes()]
    else:
        seqs = [r[2] for r in recs[:150]] + [lut[rng.integers(0, 4, size=20_000)].tobytes()]
    got = sc.scan(sc.database(seqs))
    want = oracle_scan(d, seqs)
    assert got.shape == want.shape, name
    assert np.array_equal(got, want), name


def test_n_rich_synthetic_equals_oracle(built, workdir):
    """the synthetic database with 2.7 % N (gbrna-like) -- the ambiguity
    mask path at scale, both strands."""
    import rnamotif_amd as R
    from oracle_binding import oracle_scan
    rng = np.random.default_rng(27)
    lut = np.frombuffer(b"acgtn", dtype=np.uint8)
    seqs = []
    for n in (1_500_000, 65_537):
        v = rng.integers(0, 4, size=n)
        v[rng.random(n) < 0.027] = 4
        seqs.append(lut[v].tobytes())
    for name in ("trna.descr", "mp.ends.descr"):
        d = _descr(workdir, name)
        sc = R.Scanner(d)
        got = sc.scan(sc.database(seqs))
        want = oracle_scan(d, seqs)
        assert got.shape == want.shape and np.array_equal(got, want), name


def test_descriptors_of_many_elements(built, tmp_path):
    """Beyond the 32 elements of round 1, up to the reference's own 100 (compile.c:49): a chain of
    twelve hairpins (48 elements, 25 search levels: general instance) and one of 33 single bases
    between two helices (lean descriptor, too many levels for the lean records) against the oracle."""
    import rnamotif_amd as R
    from oracle_binding import oracle_scan
    rng = np.random.default_rng(77)
    s = _planted_sequence(rng, 30_000)
    texts = ["parms\n\twc += gu;\ndescr\n" + "".join("\th5(minlen=2,maxlen=3)\n\t\tss(minlen=3,maxlen=4)\n\th3\n\tss(minlen=0,maxlen=2)\n" for _ in range(12)),
             "descr\n\th5(minlen=3,maxlen=4)\n" + "".join("\t\tss(len=1)\n" for _ in range(33)) + "\th3\n\tss(minlen=1,maxlen=3)\n\th5(minlen=3,maxlen=3)\n\t\tss(minlen=3,maxlen=5)\n\th3\n"]
    for k, text in enumerate(texts):
        path = tmp_path / ("many%d.descr" % k)
        path.write_text(text)
        d = R.Descriptor(["-descr", str(path)])
        assert d.n_elems > 32
        sc = R.Scanner(d)
        got = sc.scan(sc.database([s, s[:500]]))
        want = oracle_scan(d, [s, s[:500]])
        assert got.shape == want.shape and np.array_equal(got, want), text
        assert k == 0 or want.shape[0] > 0


def test_hit_dense_energy_sites(built, workdir, gbrna, tmp_path):
    """Many candidates per base: efn.descr (a tetraloop hairpin scored with efn()) and a hairpin
    without sequence constraint scored with efn() and efn2(), over the reference's test database
    five times (11.3 Mbase of rRNA and tRNA genes) -- every record, energies included, equal to the
    oracle's; the time of the energy kernel is printed (one workgroup of 256 lanes per CU, tables
    staged once per workgroup, candidates in a grid-stride loop)."""
    import rnamotif_amd as R
    from oracle_binding import oracle_scan
    seqs = [r[2] for r in R.read_fasta(gbrna)] * 5
    dense = tmp_path / "dense.descr"
    dense.write_text("parms\n\twc += gu;\ndescr\n\th5( minlen=5, maxlen=9, tag='a' )\n\t\tss( minlen=3, maxlen=8 )\n\th3( tag='a' )\n"
                     "score\n\t{ SCORE = efn( h5['a'], h3['a'] ) + efn2( h5['a'], h3['a'] ); }\n")
    for path in (os.path.join(workdir, "efn.descr"), str(dense)):
        d = R.Descriptor(["-descr", path])
        sc = R.Scanner(d)
        db = sc.database(seqs)
        got = sc.scan(db)
        want = oracle_scan(d, seqs)
        assert got.shape == want.shape and np.array_equal(got, want), path
        n, s_ms, e_ms = sc.scan_device(db)
        n, s_ms, e_ms = sc.scan_device(db)
        print("\n%s: %d candidates over %d bases (%.0f per Mbase), search kernel %.2f ms, energy kernel %.2f ms = %.2f us per candidate"
              % (os.path.basename(path), n, db.bases, n / (db.bases / 1e6), s_ms, e_ms, 1e3 * e_ms / max(n, 1)))
        assert n == want.shape[0] and n > 2000


def test_one_long_entry(built, workdir):
    """A single 12 Mbase entry (coordinates far beyond 16 bits, thousands of tiles): the
    whole-entry scan equals the union of three start-position slices of it, and every record
    is self-consistent."""
    import rnamotif_amd as R
    rng = np.random.default_rng(28)
    lut = np.frombuffer(b"acgt", dtype=np.uint8)
    seq = lut[rng.integers(0, 4, size=12_000_000)].tobytes()
    d = _descr(workdir, "trna.descr")
    sc = R.Scanner(d)
    whole = sc.scan(sc.database([seq]))
    parts = [sc.scan(sc.database([seq], ranges=[(lo, hi)])) for lo, hi in ((0, 4_000_001), (4_000_001, 11_999_000), (11_999_000, 12_000_000))]
    allh = np.concatenate(parts, axis=0)
    allh = allh[np.lexsort(allh[:, :5].T[::-1])]
    assert whole.shape[0] > 300 and np.array_equal(allh, whole)
    off = whole[:, 5::4][:, :d.n_elems]
    ln = whole[:, 6::4][:, :d.n_elems]
    assert np.all(off[:, 1:] == off[:, :-1] + ln[:, :-1]) and np.all(off[:, 0] == whole[:, 2])
    assert whole[:, 2].max() > 11_000_000 and np.all(off[:, -1] + ln[:, -1] <= len(seq))


def _random_descriptor(rng):
    """A random nested descriptor of ss and Watson-Crick helices (the lean path's domain) with
    random lengths, mispair / pairfrac / ends settings, pair sets and seq= constraints."""
    lines = []

    def ss(indent, force=False):
        r = rng.random()
        if r < 0.3:
            n = int(rng.integers(1, 7))
            spec = "len=%d" % n
            if rng.random() < 0.3:
                k = int(rng.integers(0, n))
                spec += ', seq="^%s%s"' % ("." * k, "acgt"[int(rng.integers(0, 4))])
        else:
            lo = int(rng.integers(0 if not force else 1, 5))
            hi = lo + int(rng.integers(0, 9))
            if hi == 0:
                hi = 1
            spec = "minlen=%d,maxlen=%d" % (lo, hi)
            if rng.random() < 0.15:
                spec += ', seq="%s"' % "".join("acgtn"[int(x)] for x in rng.integers(0, 5, size=int(rng.integers(1, 3))))
        lines.append("\t" * indent + "ss(%s)" % spec)

    def helix(indent, depth):
        lo = int(rng.integers(2, 6))
        hi = lo + int(rng.integers(0, 4))
        spec = "minlen=%d,maxlen=%d" % (lo, hi)
        r = rng.random()
        if r < 0.25:
            spec += ",mispair=%d" % int(rng.integers(1, 3))
        elif r < 0.35:
            spec += ",pairfrac=%.2f" % (0.6 + 0.35 * rng.random())
        if rng.random() < 0.25:
            spec += ",ends='%s'" % ["mm", "pm", "mp", "pp"][int(rng.integers(0, 4))]
        if rng.random() < 0.15:
            spec += ',pair+=gu'
        if rng.random() < 0.1:
            spec += ',seq="^%s"' % "acgt"[int(rng.integers(0, 4))]
        lines.append("\t" * indent + "h5(%s)" % spec)
        interior(indent + 1, depth + 1)
        lines.append("\t" * indent + "h3")

    def interior(indent, depth):
        n_hlx = 0 if depth >= 3 else int(rng.choice([0, 1, 1, 2, 3] if depth < 2 else [0, 0, 1]))
        if n_hlx == 0:
            ss(indent, force=True)
            return
        if rng.random() < 0.7:
            ss(indent)
        for k in range(n_hlx):
            helix(indent, depth)
            if k < n_hlx - 1 or rng.random() < 0.7:
                ss(indent)

    if rng.random() < 0.3:
        ss(1)
    helix(1, 0)
    if rng.random() < 0.5:
        ss(1)
    parms = "parms\n\twc += gu;\n" if rng.random() < 0.5 else ""
    return parms + "descr\n" + "\n".join(lines) + "\n"


@pytest.mark.parametrize("seed", range(160))
def test_random_descriptors_equal_oracle(built, tmp_path, seed):
    """Differential test over generated descriptors: every pruning rule, tile choice and queue
    path of the lean instance (and the general one where the generator's ss(minlen=0) or windows
    push a descriptor there) against the oracle, records bit for bit."""
    import rnamotif_amd as R
    from oracle_binding import oracle_scan
    rng = np.random.default_rng(1000 + seed)
    text = _random_descriptor(rng)
    path = tmp_path / "rand.descr"
    path.write_text(text)
    try:
        d = R.Descriptor(["-descr", str(path)])
    except R.RnamotifError:
        pytest.skip("generated descriptor does not compile (length constraints)")
    if d.maxlen > 600:
        pytest.skip("window too large for a quick differential run")
    try:
        sc = R.Scanner(d)
    except R.RnamotifError as e:
        pytest.skip("refused by the device build: " + str(e))
    lut = np.frombuffer(b"acgtn", dtype=np.uint8)
    # biased composition and planted inverted repeats so that helices actually form
    n = 12_000
    v = rng.choice(5, size=n, p=[0.2, 0.3, 0.3, 0.19, 0.01])
    s = bytearray(lut[v].tobytes())
    comp = {ord("a"): ord("t"), ord("c"): ord("g"), ord("g"): ord("c"), ord("t"): ord("a"), ord("n"): ord("n")}
    for _ in range(150):
        a = int(rng.integers(0, n - 80))
        k = int(rng.integers(4, 10))
        gap = int(rng.integers(3, 50))
        if a + 2 * k + gap < n:
            s[a + k + gap:a + 2 * k + gap] = bytes(comp[c] for c in reversed(s[a:a + k]))
    seqs = [bytes(s), bytes(s[:257])]
    want = oracle_scan(d, seqs)
    if want.shape[0] > 400_000:
        pytest.skip("too many candidates for a quick run")
    got = sc.scan(sc.database(seqs))
    assert got.shape == want.shape, text
    assert np.array_equal(got, want), text
    # the same entries cut into pieces, as a database of short entries (groups of small tiles)
    pieces = [bytes(s[a:a + int(ln)]) for a, ln in zip(range(0, n - 900, 700), rng.integers(0, 900, size=64))]
    want = oracle_scan(d, pieces)
    with _env(RNAMOTIF_SHORT="1", RNAMOTIF_QCAP=(64 if seed % 2 else None)):
        sc2 = R.Scanner(d)             # (the launch-shape variables are read when a scanner is created)
        got = sc2.scan(sc2.database(pieces))
    assert got.shape == want.shape, text
    assert np.array_equal(got, want), text


def _random_general_descriptor(rng):
    """Random descriptors for the general instance: pseudoknots, parallel helices, triplexes,
    4-plexes and hairpins side by side, with mispairs, mismatches, ends and sites."""
    lines, sites = [], []
    tag = [0]

    def ss(indent, lo=0, hi=6):
        a = int(rng.integers(lo, hi))
        b = a + int(rng.integers(0, 5))
        if b == 0:
            b = 1
        spec = "minlen=%d,maxlen=%d" % (a, b)
        if rng.random() < 0.2:
            spec = "len=%d" % max(1, b)
        if rng.random() < 0.2:
            pat = "".join("acgt"[int(x)] for x in rng.integers(0, 4, size=int(rng.integers(1, 3))))
            form = rng.random()     # plain, ^anchored, anchored$, ^both$ (the early tests of pinned windows)
            if form >= 0.4:
                pat = ("^" if form < 0.6 or form >= 0.8 else "") + pat + ("$" if form >= 0.6 else "")
            spec += ', seq="%s"' % pat
            if rng.random() < 0.25:
                spec += ",mismatch=1"
        lines.append("\t" * indent + "ss(%s)" % spec)

    def hspec(lo_min=2):
        lo = int(rng.integers(lo_min, 5))
        hi = lo + int(rng.integers(0, 3))
        spec = "minlen=%d,maxlen=%d" % (lo, hi)
        r = rng.random()
        if r < 0.3:
            spec += ",mispair=1"
        elif r < 0.4:
            spec += ",pairfrac=%.2f" % (0.6 + 0.3 * rng.random())
        if rng.random() < 0.2:
            spec += ",ends='%s'" % ["mm", "pm", "mp", "pp"][int(rng.integers(0, 4))]
        return spec

    def hairpin(indent):
        lines.append("\t" * indent + "h5(%s)" % hspec())
        ss(indent + 1, 3, 6)
        lines.append("\t" * indent + "h3")

    def pknot(indent):
        tag[0] += 2
        a, b = tag[0] - 1, tag[0]
        lines.append("\t" * indent + "h5(tag='%d',%s)" % (a, hspec()))
        ss(indent + 1)
        lines.append("\t" * indent + "h5(tag='%d',%s)" % (b, hspec()))
        ss(indent + 1, 1, 4)
        lines.append("\t" * indent + "h3(tag='%d')" % a)
        ss(indent + 1)
        lines.append("\t" * indent + "h3(tag='%d')" % b)
        if rng.random() < 0.4:
            sites.append("h5(tag='%d',pos=1):h3(tag='%d',pos=$) in { 'g:c', 'c:g', 'a:t', 't:a' }" % (a, a))

    def phlx(indent):
        lines.append("\t" * indent + "p5(%s)" % hspec())
        ss(indent + 1, 2, 6)
        lines.append("\t" * indent + "p3")

    def triplex(indent):
        tag[0] += 1
        lines.append("\t" * indent + "t1(tag='%d',%s)" % (tag[0], hspec(3)))
        ss(indent + 1, 3, 6)
        lines.append("\t" * indent + "t2(tag='%d')" % tag[0])
        ss(indent + 1, 3, 6)
        lines.append("\t" * indent + "t3(tag='%d')" % tag[0])

    def quad(indent):
        tag[0] += 1
        lines.append("\t" * indent + "q1(tag='%d',minlen=2,maxlen=3%s)" % (tag[0], ",mispair=1" if rng.random() < 0.5 else ""))
        for k in (2, 3, 4):
            ss(indent + 1, 2, 5)
            lines.append("\t" * indent + "q%d(tag='%d')" % (k, tag[0]))

    units = [hairpin, pknot, pknot, phlx, triplex, quad]
    if rng.random() < 0.4:
        ss(1, 1, 4)
    for k in range(int(rng.integers(1, 3))):
        units[int(rng.integers(0, len(units)))](1)
        if rng.random() < 0.6:
            ss(1, 0, 4)
    text = ("parms\n\twc += gu;\n" if rng.random() < 0.5 else "") + "descr\n" + "\n".join(lines) + "\n"
    if sites:
        text += "sites\n\t" + "\n\t".join(sites) + "\n"
    return text


def _planted_sequence(rng, n):
    lut = np.frombuffer(b"acgtn", dtype=np.uint8)
    v = rng.choice(5, size=n, p=[0.2, 0.3, 0.3, 0.19, 0.01])
    s = bytearray(lut[v].tobytes())
    comp = {ord("a"): ord("t"), ord("c"): ord("g"), ord("g"): ord("c"), ord("t"): ord("a"), ord("n"): ord("n")}
    for _ in range(n // 80):
        a = int(rng.integers(0, n - 80))
        k = int(rng.integers(4, 10))
        gap = int(rng.integers(3, 50))
        if a + 2 * k + gap < n:
            s[a + k + gap:a + 2 * k + gap] = bytes(comp[c] for c in reversed(s[a:a + k]))
    for _ in range(n // 60):            # runs of g for the 4-plexes, of a / t for the triplexes
        a = int(rng.integers(0, n - 12))
        k = int(rng.integers(3, 9))
        s[a:a + k] = bytes([ord("ggat"[int(rng.integers(0, 4))])]) * k
    return bytes(s)


@pytest.mark.parametrize("strict", [False, True])
@pytest.mark.parametrize("seed", range(40))
def test_random_general_descriptors_equal_oracle(built, tmp_path, seed, strict):
    import rnamotif_amd as R
    from oracle_binding import oracle_scan
    rng = np.random.default_rng(5000 + seed)
    text = _random_general_descriptor(rng)
    path = tmp_path / "rand.descr"
    path.write_text(text)
    argv = (["-sh", "-context", "-Dctx_maxlen=4"] if strict else []) + ["-descr", str(path)]
    try:
        d = R.Descriptor(argv)
    except R.RnamotifError:
        pytest.skip("generated descriptor does not compile")
    if d.maxlen > 160:
        pytest.skip("window too large for a quick differential run")
    try:
        sc = R.Scanner(d)
    except R.RnamotifError as e:
        pytest.skip("refused by the device build: " + str(e))
    s = _planted_sequence(rng, 6_000)
    seqs = [s, s[:301]]
    want = oracle_scan(d, seqs)
    if want.shape[0] > 300_000:
        pytest.skip("too many candidates for a quick run")
    got = sc.scan(sc.database(seqs))
    assert got.shape == want.shape, text
    assert np.array_equal(got, want), text
    # cut into short entries (lean descriptors the generator happens to produce: groups of small
    # tiles; the others: many ragged tiles), every other seed with a queue that overflows
    pieces = [s[a:a + int(ln)] for a, ln in zip(range(0, 5000, 450), rng.integers(0, 700, size=64))]
    want = oracle_scan(d, pieces)
    with _env(RNAMOTIF_SHORT="1", RNAMOTIF_QCAP=(64 if seed % 2 else None)):
        sc2 = R.Scanner(d)             # (the launch-shape variables are read when a scanner is created)
        got = sc2.scan(sc2.database(pieces))
    assert got.shape == want.shape, text
    assert np.array_equal(got, want), text


@pytest.mark.parametrize("gen", ["lean", "general"])
@pytest.mark.parametrize("seed", range(40))
def test_random_descriptors_under_stress_settings(built, tmp_path, seed, gen):
    """The same generated descriptors with tiles of 256 positions and a work queue of 64 entries:
    almost every tile overflows its queue, so the in-place search of the pre-filter, the tile
    edges and the ragged last tiles carry the load.  Entries of awkward lengths ride along."""
    import rnamotif_amd as R
    from oracle_binding import oracle_scan
    rng = np.random.default_rng((1000 if gen == "lean" else 5000) + seed)
    text = _random_descriptor(rng) if gen == "lean" else _random_general_descriptor(rng)
    path = tmp_path / "rand.descr"
    path.write_text(text)
    try:
        d = R.Descriptor(["-descr", str(path)])
    except R.RnamotifError:
        pytest.skip("generated descriptor does not compile")
    if d.maxlen > 160:
        pytest.skip("window too large for a quick differential run")
    s = _planted_sequence(rng, 5_000)
    seqs = [s, b"", s[:max(d.minlen - 1, 0)], s[:d.minlen], s[:d.maxlen], s[:d.maxlen + 1], s[100:100 + 255], s[7:7 + 513]]
    want = oracle_scan(d, seqs)
    if want.shape[0] > 300_000:
        pytest.skip("too many candidates for a quick run")
    old = {k: os.environ.get(k) for k in ("RNAMOTIF_TILE", "RNAMOTIF_QCAP", "RNAMOTIF_SPILL")}
    os.environ["RNAMOTIF_TILE"] = "256"
    os.environ["RNAMOTIF_QCAP"] = "64"
    # what does not fit the queue spills to HBM, and what does not fit there is searched in place:
    # no spill area at all, or one of 16 items
    os.environ["RNAMOTIF_SPILL"] = "0" if seed % 2 else "16"
    try:
        try:
            sc = R.Scanner(d)
        except R.RnamotifError as e:
            pytest.skip("refused by the device build: " + str(e))
        got = sc.scan(sc.database(seqs))
    finally:
        for k, v in old.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v
    assert got.shape == want.shape, text
    assert np.array_equal(got, want), text


@pytest.mark.parametrize("variant", ["anchored", "floating", "mismatch", "helix", "ranges"])
def test_long_seq_expressions(built, tmp_path, variant):
    """seq= expressions of 64 to 127 positions (the device's position automaton takes two words a set
    of states since round 3; round 2 refused them): planted occurrences, exact and with mismatches,
    anchored and not, on a single strand and on a helix strand, with repeat ranges -- records equal to the oracle's."""
    import rnamotif_amd as R
    from oracle_binding import oracle_scan
    rng = np.random.default_rng(404)
    lut = np.frombuffer(b"acgt", dtype=np.uint8)
    word = lut[rng.integers(0, 4, size=80)].tobytes().decode()
    if variant == "anchored":
        text = 'descr\n\tss(minlen=80,maxlen=90,seq="^%s")\n' % word
    elif variant == "floating":
        text = 'descr\n\tss(minlen=85,maxlen=100,seq="%s")\n' % word
    elif variant == "mismatch":
        text = 'descr\n\tss(len=80,seq="^%s$",mismatch=3)\n' % word
    elif variant == "ranges":
        text = 'descr\n\tss(minlen=72,maxlen=100,seq="^%sn\\{2,30\\}%s$")\n' % (word[:40], word[40:70])
    else:
        stem = lut[rng.integers(0, 4, size=12)].tobytes().decode()
        text = 'descr\n\tss(minlen=66,maxlen=70,seq="^%s")\n\th5(len=6)\n\t\tss(minlen=4,maxlen=8)\n\th3\n' % word[:66]
    path = tmp_path / "long.descr"
    path.write_text(text)
    d = R.Descriptor(["-descr", str(path)])
    comp = bytes.maketrans(b"acgt", b"tgca")
    seqs = []
    for k in range(6):
        bg = bytearray(lut[rng.integers(0, 4, size=3000 + 17 * k)].tobytes())
        for pos in (100, 1500, 2800):
            w = bytearray(word.encode())
            if variant == "mismatch":
                for j in rng.choice(80, size=k % 5, replace=False):
                    w[j] = ord("acgt"[(b"acgt".index(w[j]) + 1) % 4])
            if variant == "ranges":
                w = bytearray(word[:40].encode()) + bytearray(lut[rng.integers(0, 4, size=2 + 5 * k)].tobytes()) + bytearray(word[40:70].encode())
            if variant == "helix":
                # the word, then a hairpin that the helix can take
                hp = lut[rng.integers(0, 4, size=6)].tobytes()
                w = bytearray(word[:66].encode()) + bytearray(b"a" * (k % 3)) + bytearray(hp + b"gaaaa" + hp.translate(comp)[::-1])
            if pos + len(w) < len(bg):
                bg[pos:pos + len(w)] = w
        s_ = bytes(bg)
        seqs.append(s_ if k % 2 == 0 else s_.translate(comp)[::-1])
    want = oracle_scan(d, seqs)
    assert want.shape[0] > 0
    sc = R.Scanner(d)
    got = sc.scan(sc.database(seqs))
    assert got.shape == want.shape and np.array_equal(got, want)


@pytest.mark.parametrize("seed", range(40))
def test_random_lean_descriptors_through_the_drain_kernel(built, tmp_path, seed):
    """Generated ss / helix descriptors over planted sequence with everything the search kernel's filters let
    through walked by the drain kernel (dbg 8388608) -- in pieces and whole, with and without subtrees handed
    to idle lanes -- and by the workgroups themselves (drain 0): the same records, bit for bit, as the oracle's.
    (The order words of the candidates are numbers of the walk's choices there, counts here: both sorts
    renumber them.)"""
    import rnamotif_amd as R
    from oracle_binding import oracle_scan
    rng = np.random.default_rng(7000 + seed)
    text = _random_descriptor(rng)
    path = tmp_path / "rand.descr"
    path.write_text(text)
    try:
        d = R.Descriptor(["-descr", str(path)])
    except R.RnamotifError:
        pytest.skip("generated descriptor does not compile")
    if d.maxlen > 200:
        pytest.skip("window too large for a quick differential run")
    s = _planted_sequence(rng, 20_000)
    seqs = [s, b"", s[:d.minlen], s[:d.maxlen + 1], s[100:100 + 2047], _planted_sequence(rng, 3_000)]
    want = oracle_scan(d, seqs)
    if want.shape[0] > 300_000:
        pytest.skip("too many candidates for a quick run")
    try:
        sc = R.Scanner(d)
    except R.RnamotifError as e:
        pytest.skip("refused by the device build: " + str(e))
    db = sc.database(seqs)
    for opts in ({"dbg": 8388608}, {"dbg": 8388608 + 2097152}, {"dbg": 8388608 + 4194304}, {"drain": 0, "dbg": 0}):
        for k, v in dict({"drain": 1}, **opts).items():
            sc.set_option(k, v)
        got = sc.scan(db)
        assert got.shape == want.shape, (opts, text)
        assert np.array_equal(got, want), (opts, text)


@pytest.mark.parametrize("seed", range(60))
def test_random_descriptors_with_energy_sites(built, tmp_path, seed):
    """Generated nested descriptors scored with efn() and efn2() over their outermost helix:
    both energies of every candidate (hairpins, bulges, interior and multi-branch loops as the
    generator nests them) equal the oracle's."""
    import rnamotif_amd as R
    from oracle_binding import oracle_scan
    rng = np.random.default_rng(1000 + seed)
    text = _random_descriptor(rng)
    elems = [l for l in text.split("descr\n")[1].split("\n") if l.strip()]
    tops = [i + 1 for i, l in enumerate(elems) if l.startswith("\th") and not l.startswith("\t\t")]
    a, b = tops[0], tops[-1]
    text += "score\n\t{ SCORE = sprintf( '%%8.3f %%8.3f', efn( h5[%d], h3[%d] ), efn2( h5[%d], h3[%d] ) ); }\n" % (a, b, a, b)
    path = tmp_path / "rand.descr"
    path.write_text(text)
    try:
        d = R.Descriptor(["-descr", str(path)])
    except R.RnamotifError:
        pytest.skip("generated descriptor does not compile")
    if d.maxlen > 300:
        pytest.skip("window too large for a quick differential run")
    assert d.n_efn_sites == 2 and d.efn2data
    sc = R.Scanner(d)
    s = _planted_sequence(rng, 8_000)
    want = oracle_scan(d, [s])
    if want.shape[0] > 200_000:
        pytest.skip("too many candidates for a quick run")
    got = sc.scan(sc.database([s]))
    assert got.shape == want.shape, text
    assert np.array_equal(got, want), text


def test_sixteen_sites_and_energy_calls(built, tmp_path, gbrna):
    """The boundary's own limits -- 16 sites, 16 efn()/efn2() calls in the score section (8 each until
    round 2) -- against the oracle: a cloverleaf-like descriptor with 12 sites and 10 energy calls."""
    import rnamotif_amd as R
    from oracle_binding import oracle_scan
    text = ("parms\n\twc += gu;\ndescr\n"
            "\th5(tag='a',minlen=5,maxlen=7)\n\t\tss(minlen=2,maxlen=4)\n"
            "\t\th5(tag='b',minlen=3,maxlen=4)\n\t\t\tss(minlen=4,maxlen=9)\n\t\th3(tag='b')\n"
            "\t\tss(minlen=1,maxlen=3)\n"
            "\t\th5(tag='c',minlen=3,maxlen=5)\n\t\t\tss(minlen=5,maxlen=8)\n\t\th3(tag='c')\n"
            "\t\tss(minlen=2,maxlen=6)\n\th3(tag='a')\n"
            "sites\n" +
            "".join("\th5(tag='%s',pos=%d):h3(tag='%s',pos=$-%d) in { 'a:u', 'u:a', 'g:c', 'c:g', 'g:u', 'u:g' }\n" % (t, p, t, p - 1)
                    for t, n in (("a", 5), ("b", 3), ("c", 3)) for p in range(1, n + 1)) +
            "\th5(tag='a',pos=1):h3(tag='a',pos=$) in { 'g:c', 'c:g', 'a:u', 'u:a' }\n"
            "score\n\t{ SCORE = sprintf( '" + " ".join(["%6.2f"] * 10) + "', " +
            ", ".join(["efn( h5['a'], h3['a'] )", "efn2( h5['a'], h3['a'] )", "efn( h5['b'], h3['b'] )", "efn2( h5['b'], h3['b'] )",
                       "efn( h5['c'], h3['c'] )", "efn2( h5['c'], h3['c'] )", "efn( h5['b'], h3['c'] )", "efn2( h5['b'], h3['c'] )",
                       "efn( h5['a'], h3['c'] )", "efn2( h5['b'], h3['a'] )"]) + " ); }\n")
    p = tmp_path / "many.descr"
    p.write_text(text)
    d = R.Descriptor(["-descr", str(p)])
    assert d.n_efn_sites == 10
    seqs = [r[2] for r in R.read_fasta(gbrna)[:1200]]
    want = oracle_scan(d, seqs)
    sc = R.Scanner(d)
    got = sc.scan(sc.database(seqs))
    assert want.shape[0] > 20 and got.shape == want.shape and np.array_equal(got, want)


Q1_VARIANTS = [
    # (4-plex attributes, what follows)
    ("minlen=3, maxlen=5, mispair=1", "\tss( minlen=3, maxlen=10 )\n\tt1( tag='2', minlen=4, maxlen=7, mispair=1 )\n\t\tss( minlen=3, maxlen=10 )\n\tt2( tag='2' )\n\t\tss( minlen=3, maxlen=10 )\n\tt3( tag='2' )\n"),
    ("minlen=2, maxlen=4, mispair=0", "\tt1( tag='2', minlen=3, maxlen=4, mispair=0 )\n\t\tss( minlen=2, maxlen=6 )\n\tt2( tag='2' )\n\t\tss( minlen=2, maxlen=6 )\n\tt3( tag='2' )\n"),
    ("minlen=3, maxlen=3, mispair=2, ends='mm'", "\tss( minlen=0, maxlen=4 )\n\tt1( tag='2', minlen=3, maxlen=5, mispair=2, ends='mm' )\n\t\tss( minlen=1, maxlen=5 )\n\tt2( tag='2' )\n\t\tss( minlen=1, maxlen=5 )\n\tt3( tag='2' )\n"),
    ("minlen=2, maxlen=3, mispair=1", "\tss( minlen=2, maxlen=5 )\n\th5( minlen=3, maxlen=4 )\n\t\tss( minlen=3, maxlen=6 )\n\th3\n"),
    ("minlen=3, maxlen=4, mispair=1", ""),
]


@pytest.mark.parametrize("variant", range(len(Q1_VARIANTS)))
def test_leading_4plex_strand_filter(built, tmp_path, variant):
    """A 4-plex at the head of the search list takes the pre-filter's strand filter (rmd_q1filter_t),
    with the look-ahead along a triplex that follows it where there is one: same records as the oracle
    on G-rich sequence (where 4-plexes and triplexes do occur), for several shapes of the two."""
    import rnamotif_amd as R
    from oracle_binding import oracle_scan
    attrs, after = Q1_VARIANTS[variant]
    text = ("descr\n\tq1( tag='1', %s )\n\t\tss( minlen=2, maxlen=6 )\n\tq2( tag='1' )\n\t\tss( minlen=2, maxlen=6 )\n"
            "\tq3( tag='1' )\n\t\tss( minlen=2, maxlen=6 )\n\tq4( tag='1' )\n" % attrs) + after
    p = tmp_path / "q.descr"
    p.write_text(text)
    d = R.Descriptor(["-descr", str(p)])
    rng = np.random.default_rng(40 + variant)
    lut = np.frombuffer(b"acgtn", dtype=np.uint8)
    seqs = [lut[rng.choice(5, size=n, p=pr)].tobytes()
            for n, pr in ((60_000, [.15, .1, .55, .2, 0]), (20_003, [.3, .1, .3, .3, 0]), (9_000, [.1, .1, .5, .28, .02]), (70, [.1, .1, .6, .2, 0]))]
    want = oracle_scan(d, seqs)
    sc = R.Scanner(d)
    got = sc.scan(sc.database(seqs))
    assert want.shape[0] > 0 or variant == 1, "the sequence should hold some of these"
    assert got.shape == want.shape and np.array_equal(got, want)
    # ... and with the filter switched off (dbg bits 8192, 16384: the launch shape stays)
    for bits in (8192, 16384):
        sc.set_option("dbg", bits)
        again = sc.scan(sc.database(seqs))
        assert np.array_equal(again, want)
