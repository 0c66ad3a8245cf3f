"""CPU: rma_sort_hits() (rnamotif_amd/csrc/rm_hitsort.h) -- the order every scan's records leave in:
by (entry, strand, start, rank of the outer alternative, order), ties in buffer order, the order word
renumbered 0, 1, ... within (entry, strand, start, rank) as the reference's walk counts its
candidates (find_motif.c:164-215, 370-411).  Against numpy's stable lexsort on the same records."""
import numpy as np
import pytest

import rnamotif_amd as R


def _expect(h):
    if h.shape[0] == 0:
        return h.copy()
    k = h[:, :5].astype(np.int64)
    idx = np.lexsort((np.arange(h.shape[0]), k[:, 4], k[:, 3], k[:, 2], k[:, 1] & 1, k[:, 0]))
    out = h[idx].copy()
    key = np.stack([out[:, 0], out[:, 1] & 1, out[:, 2], out[:, 3]], axis=1)
    new = np.ones(out.shape[0], dtype=bool)
    new[1:] = np.any(key[1:] != key[:-1], axis=1)
    start = np.maximum.accumulate(np.where(new, np.arange(out.shape[0]), 0))
    out[:, 4] = np.arange(out.shape[0]) - start
    return out


def _records(rng, n, stride, n_seq, max_pos, max_rank, max_order):
    h = rng.integers(0, 1 << 30, size=(n, stride), dtype=np.int64).astype(np.int32)
    h[:, 0] = rng.integers(0, n_seq, n)
    h[:, 1] = rng.integers(0, 2, n)
    h[:, 2] = rng.integers(0, max_pos, n)
    h[:, 3] = rng.integers(0, max_rank, n)
    h[:, 4] = rng.integers(0, max_order, n)
    return h


@pytest.mark.parametrize("n,stride,n_seq,max_pos,max_rank,max_order", [
    (0, 9, 1, 1, 1, 1), (1, 5, 3, 10, 2, 2), (700, 69, 20, 50, 3, 4),       # comparison sort below 1024 records
    (5000, 9, 100, 1000, 33, 8), (5000, 9, 1, 40, 2, 3),                      # many ties
    (200000, 7, 3000, 1 << 20, 40, 1 << 12),
    (30000, 6, (1 << 31) - 1, (1 << 31) - 1, 65536, (1 << 31) - 1),           # every bit of the keys in use
])
def test_order_and_renumbering(built, n, stride, n_seq, max_pos, max_rank, max_order):
    rng = np.random.default_rng(n * 31 + stride)
    h = _records(rng, n, stride, n_seq, max_pos, max_rank, max_order)
    got = R.sort_hits(h)
    assert np.array_equal(got, _expect(h))


def test_sorted_input_and_slices_of_ranks(built):
    rng = np.random.default_rng(5)
    h = _expect(_records(rng, 40000, 11, 500, 100000, 6, 1))
    assert np.array_equal(R.sort_hits(h), h)
    # what rank 0 of a multi-GPU job holds: every rank's records in order, ranks interleaved by entry
    parts = [h[(h[:, 0] % 4) == r] for r in range(4)]
    assert np.array_equal(R.sort_hits(np.concatenate(parts)), h)


def test_refuses_records_without_header(built):
    with pytest.raises(R.RnamotifError):
        R.sort_hits(np.zeros((3, 4), dtype=np.int32))
