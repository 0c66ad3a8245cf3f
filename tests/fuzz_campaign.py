"""Not a test: a longer differential campaign over the descriptor generators of
tests/test_gpu_parity.py (GPU scanner against the oracle, a third of the cases with small tiles
and a small work queue).  Run on the GPU box from the repository root:

    python tests/fuzz_campaign.py lean 160 3000 ; python tests/fuzz_campaign.py general 40 3000
    python tests/fuzz_campaign.py grouped 160 3000     (lean descriptors, database cut into short
                                                        entries, groups of small tiles forced)
    python tests/fuzz_campaign.py concat 160 3000      (... tiles over the concatenation of the entries forced)
    python tests/fuzz_campaign.py gconcat 160 3000     (general descriptors over such tiles)
    RNAMOTIF_FLUSH=1 python tests/fuzz_campaign.py lean 160 3000     (the search kernel that walks nothing, whatever the descriptor)
    FUZZ_LOWC=1 python tests/fuzz_campaign.py lean 160 3000          (low-complexity sequence: clustered survivors, overflowing queues)

Round 1: 6400 lean, 5700 general and 3566 grouped descriptors, no mismatch.
Round 3 (drain kernel, order words from the walk's choices, start positions by words): 1173 lean (seeds
3000-4190), 1673 lean with everything through the drain kernel's list (RNAMOTIF_DBG=8388608, seeds
9000-10698), 2482 general (3000-6000), 1106 grouped (3000-4121): no mismatch.  And on the round's final build
(walk that skips back over fixed single strands, merged chain passes, start positions by words in the general
instances too): 1788 lean (20000-21821), 1751 lean through the list (30000-31796), 4943 general (6000-12000),
404 grouped (9000-9407): no mismatch.  After the groups got the word-wise start positions: 2386 grouped
(15000-17447) and 518 lean (40000-40524): no mismatch.  General instances at three waves per SIMD: 3987 general
(12000-16796): no mismatch.
Round 4 (the search kernel that walks nothing, tickets for four tiles, tiles over the concatenation of the entries): 3250
lean with RNAMOTIF_FLUSH=1 (50000-53323) and 798 lean (60000-60804): no mismatch; 3341 concat with RNAMOTIF_FLUSH=1
(50000-53421): ELEVEN mismatches, all general descriptors -- general_pass_b ended a tile's search when a round's 64 popped
items all lay in the padding between entries (fixed; tests/test_gpu_parity.py::test_concatenation_tiles_items_in_the_padding
keeps four of them); 3603 general (20000-24340), 2102 grouped (20000-22137): no mismatch.  After the fix: 3116 + 2886
concat (50000-53189 with RNAMOTIF_FLUSH=1, 70000-72950), 2337 general (30000-32836) and 3263 gconcat (20000-23940): no mismatch.  Low-complexity sequence
(FUZZ_LOWC=1, RNAMOTIF_FLUSH=1): 266 lean (80000-80296), no mismatch -- and slow in places: seed 80277, a general
descriptor over runs of repeats, takes the oracle ten seconds and the device eighty (profiles/lowc_case.py).
With tiles sized to the device for small databases (general instances: 256 positions here): 8186 general (50000-60000) and
6685 gconcat (30000-38045): no mismatch -- at 41 descriptors a second where it was 15."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, rnamotif_amd as R
import test_gpu_parity as T
from oracle_binding import oracle_scan
kind, lo, hi = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
bad = ran = 0
t0 = time.time()
for seed in range(lo, hi):
    lean = kind in ("lean", "grouped", "concat")
    rng = np.random.default_rng((1000 if lean else 5000) + seed)
    text = T._random_descriptor(rng) if lean else T._random_general_descriptor(rng)
    open("/tmp/f.descr", "w").write(text)
    argv = ["-descr", "/tmp/f.descr"]
    if not lean and seed % 2:
        argv = ["-sh", "-context", "-Dctx_maxlen=4"] + argv
    try:
        d = R.Descriptor(argv)
    except R.RnamotifError:
        continue
    if d.maxlen > (400 if lean else 160):
        continue
    s = T._planted_sequence(rng, 6000)
    if os.environ.get("FUZZ_LOWC"):
        # low-complexity sequence: runs of short repeats with a few point changes between planted stretches -- the filters'
        # survivors come in clusters, queues and lists overflow
        lut = np.frombuffer(b"acgt", dtype=np.uint8)
        parts, n = [], 0
        while n < 6000:
            if rng.random() < 0.3:
                a = int(rng.integers(0, 5800)); piece = s[a:a + int(rng.integers(50, 200))]
            else:
                unit = lut[rng.integers(0, 4, size=int(rng.integers(1, 7)))].tobytes()
                piece = bytearray(unit * int(rng.integers(5, 120)))
                for _ in range(len(piece) // 25):
                    piece[int(rng.integers(0, len(piece)))] = int(lut[rng.integers(0, 4)])
                piece = bytes(piece)
            parts.append(piece); n += len(piece)
        s = b"".join(parts)[:6000]
    seqs = [s, s[:301], s[:d.maxlen], s[:d.minlen]]
    if kind in ("grouped", "concat", "gconcat"):
        cuts = np.sort(rng.integers(0, len(s), size=int(rng.integers(5, 40))))
        seqs = [s[a:b] for a, b in zip(np.r_[0, cuts], np.r_[cuts, len(s)])] + [s[:1030], s[:1024 + d.maxlen], b""]
        os.environ["RNAMOTIF_SHORT"] = "1" if kind == "grouped" else "2"      # (concat: tiles over the concatenation of the entries)
    want = oracle_scan(d, seqs)
    if want.shape[0] > 300000:
        continue
    if seed % 3 == 0:
        os.environ["RNAMOTIF_TILE"] = "512"; os.environ["RNAMOTIF_QCAP"] = "128"
        if kind in ("grouped", "concat", "gconcat"):
            os.environ.pop("RNAMOTIF_TILE")     # (a forced tile size switches the groups off)
            os.environ["RNAMOTIF_QCAP"] = "64"
        # the queue's spill area: the default, or one of 16 items (then the rest is searched in place)
        if seed % 2:
            os.environ["RNAMOTIF_SPILL"] = "16"
        else:
            os.environ.pop("RNAMOTIF_SPILL", None)
    else:
        os.environ.pop("RNAMOTIF_TILE", None); os.environ.pop("RNAMOTIF_QCAP", None); os.environ.pop("RNAMOTIF_SPILL", None)
    try:
        sc = R.Scanner(d)
    except R.RnamotifError:
        continue
    if os.environ.get("FUZZ_TRACE"):       # (the case at hand, for a run that does not come back)
        open(os.path.join(ROOT, "gpurun_out", "fuzz_last_%s.descr" % kind), "w").write("# seed %d, %d candidates wanted\n" % (seed, want.shape[0]) + text)
    got = sc.scan(sc.database(seqs))
    ran += 1
    if got.shape != want.shape or not np.array_equal(got, want):
        bad += 1
        print("MISMATCH", kind, seed, got.shape, want.shape, flush=True)
        os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
        open(os.path.join(ROOT, "gpurun_out", "fuzz_bad_%s_%d.descr" % (kind, seed)), "w").write(text)
    if time.time() - t0 > float(os.environ.get("FUZZ_SECONDS", "240")):
        print("time budget reached at seed", seed, flush=True)
        break
print(kind, "ran", ran, "bad", bad, "in %.0f s" % (time.time() - t0), flush=True)
