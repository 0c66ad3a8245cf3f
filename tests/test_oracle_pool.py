"""CPU: the process-parallel oracle of the full-size GPU tests (tests/oracle_pool.py) gives what the
oracle gives in one process, and its records are the BASELINE stream's."""
import os

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_records_jump_into_the_stream():
    import rnamotif_amd as R
    from oracle_pool import synthetic_record
    want = R.synthetic_records(4, length=30_000)
    for k in (0, 1, 3):
        assert synthetic_record(k, 30_000) == want[k]
    want = R.synthetic_records(3, length=1001)
    assert synthetic_record(2, 1001) == want[2]


def test_pool_equals_one_process(built, workdir):
    import rnamotif_amd as R
    from oracle_binding import oracle_scan
    from oracle_pool import concat_records, host_cores, oracle_records
    assert host_cores() >= 1
    args = ["-descr", "sprintf.descr"]
    cwd = os.getcwd()
    os.chdir(workdir)
    try:
        d = R.Descriptor(args)
    finally:
        os.chdir(cwd)
    records = [5, 0, 2, 9, 1]
    got, info = oracle_records(args, records, cwd=workdir, length=40_000, procs=3)
    assert info["procs"] == 3 and sorted(got) == sorted(records)
    seqs = R.synthetic_records(10, length=40_000)
    want = oracle_scan(d, [seqs[k] for k in records])
    assert want.shape[0] > 20
    assert np.array_equal(concat_records(got, records, d.hit_stride), want)
