"""GPU (one device, one process): rma_gather_hits() of the C ABI driven through a transport of the test's own
(rma_comm_create_on): this process is rank 0 of a world of two, the transport plays rank 1.  What RCCL would do
between two GPUs cannot run on a box with one; everything around its calls can -- the count exchange, the root's
buffers, the grouped receive, and what happens when a call fails in the middle of the group (ADVICE / VERDICT
round 3: a failure between ncclGroupStart and ncclGroupEnd left the group open)."""
import ctypes as C
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

AG = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p)
SR = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_size_t, C.c_int, C.c_void_p)
GR = C.CFUNCTYPE(C.c_int, C.c_void_p)
ES = C.CFUNCTYPE(C.c_char_p, C.c_void_p, C.c_int)
CC = C.CFUNCTYPE(C.c_int, C.c_void_p, C.POINTER(C.c_int))


class Transport(C.Structure):
    _fields_ = [("all_gather", AG), ("send", SR), ("recv", SR), ("group_start", GR), ("group_end", GR),
                ("error_string", ES), ("comm_count", CC), ("ctx", C.c_void_p)]


class FakePeer:
    """Rank 1 of a world of two, as a transport: its records, its flag, and which call fails next."""

    def __init__(self):
        self.hip = C.CDLL("libamdhip64.so")
        self.hip.hipMemcpy.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int]
        self.hip.hipStreamSynchronize.argtypes = [C.c_void_p]
        self.records = np.zeros((0, 1), np.int32)
        self.unfit = 0
        self.fail = None                   # "recv" | "group_start" | "all_gather"
        self.calls = {"all_gather": 0, "recv": 0, "send": 0, "group_start": 0, "group_end": 0}
        self.t = Transport(AG(self._all_gather), SR(self._send), SR(self._recv), GR(self._gs), GR(self._ge),
                           ES(lambda ctx, code: b"injected failure"), CC(self._count), None)

    def _all_gather(self, ctx, send, recv, n, stream):
        self.calls["all_gather"] += 1
        if self.fail == "all_gather":
            return 7
        assert self.hip.hipStreamSynchronize(stream) == 0
        mine = (C.c_longlong * n)()
        assert self.hip.hipMemcpy(mine, send, 8 * n, 2) == 0                    # device to host
        both = (C.c_longlong * (2 * n))(*list(mine), *([self.records.shape[0], self.unfit] + [0] * n)[:n])
        assert self.hip.hipMemcpy(recv, both, 16 * n, 1) == 0                   # host to device
        return 0

    def _send(self, ctx, buf, n, peer, stream):
        self.calls["send"] += 1
        return 0

    def _recv(self, ctx, buf, n, peer, stream):
        self.calls["recv"] += 1
        if self.fail == "recv":
            return 5
        assert peer == 1 and n == self.records.size
        assert self.hip.hipStreamSynchronize(stream) == 0
        assert self.hip.hipMemcpy(buf, self.records.ctypes.data, 4 * n, 1) == 0
        return 0

    def _gs(self, ctx):
        self.calls["group_start"] += 1
        return 3 if self.fail == "group_start" else 0

    def _ge(self, ctx):
        self.calls["group_end"] += 1
        return 0

    def _count(self, ctx, out):
        out[0] = 2
        return 0


def test_exchange_through_a_transport_of_the_tests_own(built, workdir):
    import rnamotif_amd as R
    os.environ.setdefault("EFNDATA", R.EFNDATA_DIR)
    L = R.lib()
    d = R.Descriptor(["-descr", os.path.join(ROOT, "tests", "golden", "test", "sprintf.descr")])
    seqs = R.synthetic_records(1, length=60_000)
    sc = R.Scanner(d)
    db = sc.database(seqs)
    want = sc.scan(db)
    assert want.shape[0] > 50
    peer = FakePeer()
    theirs = want.copy()
    theirs[:, 0] = 9                       # (the peer's records carry their database-wide entry number already)
    peer.records = np.ascontiguousarray(theirs)
    h = C.c_void_p()
    err = C.create_string_buffer(1024)
    assert L.rma_comm_create_on(C.byref(peer.t), 0, 2, 0, C.byref(h), err, 1024) == 0, err.value
    comm = R.Comm.__new__(R.Comm)
    comm._h, comm.rank, comm.world = h, 0, 2
    assert comm.count() == 2

    def scan_on_device():
        sc.scan_begin(db)
        return sc.scan_end_on_device()

    # 1. the exchange: this rank's records relabelled (entry 0 -> 4), the peer's behind them
    assert scan_on_device() == want.shape[0]
    got, counts = comm.gather(sc, [4])
    mine = want.copy()
    mine[:, 0] = 4
    assert counts == [want.shape[0], theirs.shape[0]]
    assert np.array_equal(got, np.concatenate([mine, theirs]))
    assert peer.calls["group_start"] == peer.calls["group_end"] == 1 and peer.calls["recv"] == 1
    assert peer.calls["all_gather"] == 2          # (the first gather: the root's buffers had to be made, and every rank heard that they were)

    # 2. a second gather of the SAME scan: the records are relabelled once (entry 4 is not looked up again)
    got2, _ = comm.gather(sc, [4])
    assert np.array_equal(got2, got)
    assert peer.calls["all_gather"] == 3          # (no growth: one exchange of counts)

    # 3. a receive fails inside the group: the error comes back and the group is closed
    scan_on_device()
    peer.fail = "recv"
    with pytest.raises(R.RnamotifError, match="receive: injected failure"):
        comm.gather(sc, [4])
    assert peer.calls["group_start"] == peer.calls["group_end"] == 3
    peer.fail = None
    got3, _ = comm.gather(sc, [4])             # ... and the communicator works on (the same scan: already relabelled)
    assert np.array_equal(got3, got)
    assert peer.calls["group_start"] == peer.calls["group_end"] == 4

    # 4. the group cannot be opened: nothing is posted, nothing is left open
    peer.fail = "group_start"
    with pytest.raises(R.RnamotifError, match="group start"):
        comm.gather(sc, [4])
    assert peer.calls["group_start"] == 5 and peer.calls["group_end"] == 4 and peer.calls["recv"] == 4
    peer.fail = None

    # 5. a rank that cannot take part says so in the count exchange: nobody enters the exchange
    peer.unfit = 1
    with pytest.raises(R.RnamotifError, match="rank 1 cannot take part"):
        comm.gather(sc, [4])
    assert peer.calls["group_start"] == 5
    peer.unfit = 0

    # 6. this rank's records are on the host only (ordering on the host, copied back): it says so, and so the peer hears
    sc.set_option("host_sort", 1)
    host = sc.scan(db)
    assert np.array_equal(host, want)
    with pytest.raises(R.RnamotifError, match="ordered on the host"):
        comm.gather(sc, [4])
    assert peer.calls["group_start"] == 5
    sc.set_option("host_sort", 0)

    # 7. more records than the root's buffers hold: it allocates, and says that it could, before anybody sends
    big = np.ascontiguousarray(np.tile(theirs, (8, 1)))
    peer.records = big
    scan_on_device()
    n_ag = peer.calls["all_gather"]
    got7, counts7 = comm.gather(sc, [4])
    assert peer.calls["all_gather"] == n_ag + 2 and counts7 == [want.shape[0], big.shape[0]]
    assert np.array_equal(got7, np.concatenate([mine, big]))
    comm.close()
