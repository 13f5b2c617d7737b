"""Randomized differential soak of the isolated product against the exact convolution (run on a
GPU box: python -u tools/soak_polymul.py).  Operand magnitudes beyond the path's digits are
included on purpose: they show where an f64 FFT stops determining the result."""
import os, sys, time
ROOT = os.getcwd()
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import _oracle as orc, tfhe_amd
rng = np.random.default_rng(11)
eng = tfhe_amd.Engine(0)
t0 = time.time(); bad = 0; trials = 0
while time.time() - t0 < 150:
    B = int(rng.choice([1, 2, 3, 5, 8, 9, 2047, 2048, 2049, 2051, 4097, 6000]))
    half = int(rng.choice([64, 512, 1 << 15, 1 << 20]))
    d = rng.integers(-half, half, (B, 1024), dtype=np.int64).astype(np.int32)
    kind = int(rng.integers(0, 3))
    if kind == 0: s = rng.integers(-2**31, 2**31, (B, 1024), dtype=np.int64).astype(np.int32)
    elif kind == 1: s = rng.integers(-2**20, 2**20, (B, 1024), dtype=np.int64).astype(np.int32)
    else:
        s = np.zeros((B, 1024), np.int32); s[:, rng.integers(0, 1024, 8)] = np.int32(2**31 - 1)
    out = eng.polymul(d, s)
    acc0 = rng.integers(-2**31, 2**31, (B, 1024), dtype=np.int64).astype(np.int32)
    acc = eng.polymul(d, s, accumulate_into=acc0)
    for i in set([0, B - 1] + list(rng.integers(0, B, 3))):
        w, _ = orc.poly_multiplier(d[i], s[i], orc.MUL_EXACT)
        ok = (out[i] == w).all() and (acc[i] == (acc0[i].astype(np.int64) + w[:1024]).astype(np.int32)).all()
        if not ok:
            bad += 1; print("MISMATCH", B, half, kind, i, flush=True)
    trials += 1
    if trials % 10 == 0: print("trials", trials, "bad", bad, flush=True)
print("DONE trials", trials, "mismatches", bad)
