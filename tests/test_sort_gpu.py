"""GPU: the ordering of the hit stream on the device (rnamotif_amd/csrc/rm_hitsort_dev.hip: packed
64-bit keys, radix sort, gather + renumbering kernel) against the host's (rm_hitsort.h,
RNAMOTIF_HOSTSORT=1) and the oracle -- including records whose order word outgrows the key's field
(the scan then falls back to the host sort, and widens the field for the scans after it)."""
import os

import numpy as np
import pytest

from test_gpu_parity import _descr, _env

pytestmark = pytest.mark.gpu


def _random(seed, sizes):
    rng = np.random.default_rng(seed)
    lut = np.frombuffer(b"acgt", dtype=np.uint8)
    return [lut[rng.integers(0, 4, size=n)].tobytes() for n in sizes]


@pytest.mark.parametrize("name", ["trna.descr", "pk1.descr", "qu+tr.descr", "mp.ends.descr", "score.1.descr"])
def test_device_order_equals_host_order(built, workdir, gbrna, name):
    import rnamotif_amd as R
    seqs = _random(21, (400_000, 9, 70_001, 1)) + [r[2] for r in R.read_fasta(gbrna)[:1500]]
    d = _descr(workdir, name)
    sc = R.Scanner(d)
    db = sc.database(seqs)
    dev = sc.scan(db)
    sc.set_option("host_sort", 1)
    host = sc.scan(db)
    sc.set_option("host_sort", 0)
    assert np.array_equal(dev, host)
    assert dev.shape[0] > 1 or name == "qu+tr.descr"      # (nothing of that shape in these entries)
    again = sc.scan(db)
    assert np.array_equal(again, host)


def test_order_words_beyond_the_key_field(built, tmp_path):
    """three single strands in a row: hundreds of candidates per (start, end) -- order words above 255"""
    import rnamotif_amd as R
    from oracle_binding import oracle_scan
    p = tmp_path / "dense.descr"
    p.write_text("descr\n h5(len=3) ss(minlen=1,maxlen=30) ss(minlen=1,maxlen=30) ss(minlen=1,maxlen=30) h3\n")
    d = R.Descriptor(["-descr", str(p)])
    seqs = _random(22, (1500, 700))
    want = oracle_scan(d, seqs)
    assert want[:, 4].max() > 255
    sc = R.Scanner(d)
    db = sc.database(seqs)
    first = sc.scan(db)          # falls back to the host sort, widens the field
    second = sc.scan(db)         # on the device with the wide field
    sc.set_option("host_sort", 1)
    host = sc.scan(db)
    assert np.array_equal(first, want) and np.array_equal(second, want) and np.array_equal(host, want)
