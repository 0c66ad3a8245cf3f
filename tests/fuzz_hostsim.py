"""Not a test: a CPU differential campaign of the device search state machines
(rnamotif_amd/csrc/rm_scan_core.h compiled for the host, tests/hostsim) against the oracle over
the descriptor generators of tests/test_gpu_parity.py.

    python tests/fuzz_hostsim.py general 0 2000 [workers]
    python tests/fuzz_hostsim.py lean-as-general 0 2000    (lean descriptors through the general path)

Needs tests/_build/hostsim_check (built by tests/test_hostsim.py).

Round 3: `lean 0 1500` -- 1471 generated lean descriptors, each candidate's order word (the number of the walk's
choices, as the kernels store it) checked to grow along the walk: no mismatch, no order word out of place."""
import os
import subprocess
import sys
import tempfile
from concurrent.futures import ProcessPoolExecutor

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
BIN = os.path.join(ROOT, "tests", "_build", "hostsim_check")


def one(job):
    kind, seed = job
    import numpy as np
    import rnamotif_amd as R
    import test_gpu_parity as T
    lean = kind != "general"
    rng = np.random.default_rng((1000 if lean else 5000) + seed)
    text = T._random_descriptor(rng) if lean else T._random_general_descriptor(rng)
    with tempfile.TemporaryDirectory() as tmp:
        path = os.path.join(tmp, "f.descr")
        open(path, "w").write(text)
        argv = ["-descr", path]
        if not lean and seed % 2:
            argv = ["-sh", "-context", "-Dctx_maxlen=4"] + argv
        try:
            d = R.Descriptor(argv)
        except R.RnamotifError:
            return seed, "skip", ""
        if d.maxlen > (400 if lean else 160):
            return seed, "skip", ""
        s = T._planted_sequence(rng, 6000)
        fa = os.path.join(tmp, "db.fastn")
        with open(fa, "wb") as f:
            for i, q in enumerate((s, s[:301], s[:d.maxlen], s[:d.minlen])):
                f.write(b">s%d x\n" % i + q + b"\n")
        env = dict(os.environ)
        if kind == "lean-as-general":
            env["HOSTSIM_NOLEAN"] = "1"
        env["HOSTSIM_BUDGET"] = str((4, 5, 7, 32)[seed % 4])    # steps that pause after that many iterations
        try:
            p = subprocess.run([BIN] + argv + [fa], stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=600, env=env)
        except subprocess.TimeoutExpired:
            return seed, "timeout", text
        if p.returncode == 2:
            return seed, "skip", ""     # outside the device limits
        if p.returncode != 0 or b" 0 mismatching strands" not in p.stdout:
            return seed, "BAD", text + "\n" + " ".join(argv) + "\n" + p.stdout.decode()[-400:] + p.stderr.decode()[-800:]
        return seed, "ok", ""


def main():
    kind, lo, hi = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
    workers = int(sys.argv[4]) if len(sys.argv) > 4 else 7
    ran = bad = 0
    with ProcessPoolExecutor(workers) as ex:
        for seed, what, text in ex.map(one, [(kind, s) for s in range(lo, hi)], chunksize=4):
            if what == "ok":
                ran += 1
            elif what != "skip":
                bad += 1
                print(what, kind, seed, flush=True)
                os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
                open(os.path.join(ROOT, "gpurun_out", "hostsim_bad_%s_%d.txt" % (kind, seed)), "w").write(text)
    print(kind, "ran", ran, "bad", bad, flush=True)


if __name__ == "__main__":
    main()
