"""The slow low-complexity case of round 4's fuzz campaign (tests/fuzz_campaign.py, FUZZ_LOWC=1, seed 80277): kernel times and
the drain kernel's own counters.  python profiles/lowc_case.py"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np  # noqa: E402
import rnamotif_amd as R  # noqa: E402
import test_gpu_parity as T  # noqa: E402

os.environ.setdefault("EFNDATA", R.EFNDATA_DIR)
lut = np.frombuffer(b"acgt", dtype=np.uint8)
seed = 80277
rng = np.random.default_rng(1000 + seed)
text = T._random_descriptor(rng)
open("/tmp/f.descr", "w").write(text)
d = R.Descriptor(["-descr", "/tmp/f.descr"])
s = T._planted_sequence(rng, 6000)
parts, n = [], 0
while n < 6000:
    if rng.random() < 0.3:
        a = int(rng.integers(0, 5800))
        piece = s[a:a + int(rng.integers(50, 200))]
    else:
        unit = lut[rng.integers(0, 4, size=int(rng.integers(1, 7)))].tobytes()
        piece = bytearray(unit * int(rng.integers(5, 120)))
        for _ in range(len(piece) // 25):
            piece[int(rng.integers(0, len(piece)))] = int(lut[rng.integers(0, 4)])
        piece = bytes(piece)
    parts.append(piece)
    n += len(piece)
s = b"".join(parts)[:6000]
seqs = [s]
sc = R.Scanner(d)
db = sc.database(seqs)
sc.set_option("dbg", 2 + 32)
t0 = time.time()
n = sc.scan_device(db)[0]
dt = time.time() - t0
k = sc.last_kernel_ms()
sc.set_option("dbg", 0)
print("seed %d: %d candidates, %.1f s; search kernel %.1f ms, drain kernel %.1f ms" % (seed, n, dt, k[0], k[1]), flush=True)
