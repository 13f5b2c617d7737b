"""Randomized differential soak of the gate path against the CPU oracle (run on a GPU box:
  python -u tools/soak_gates.py [seed] [seconds]).  Not part of the test suite."""
import os, sys, time
ROOT = os.getcwd()
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import _oracle as orc, tfhe_amd
SHAPES = {"l2": (2, 10), "l3": (3, 7)}
rng = np.random.default_rng(int(sys.argv[1]) if len(sys.argv) > 1 else 1)
eng = tfhe_amd.Engine(0)
t0 = time.time(); bad = 0; trials = 0
while time.time() - t0 < float(sys.argv[2]) if len(sys.argv) > 2 else 120:
    tag = ("l2", "l3")[int(rng.integers(0, 2))]
    l, bg = SHAPES[tag]
    n = int(rng.integers(1, 30)); B = int(rng.integers(1, 70))
    if rng.integers(0, 5) == 0:      # more ciphertexts than compute units: several wavefronts per workgroup, ragged rounds
        n = int(rng.integers(1, 9)); B = int(rng.integers(257, 4300))
    P = orc.small_params(n=n, l=l, bg_bit=bg)
    seed = int(rng.integers(1, 1 << 30))
    K = orc.generate_synthetic_keys(P, seed)
    eng.load_keys(tfhe_amd.make_params(P.n, l=l, bg_bit=bg, n_ks=P.n_ks), K["bk"], K["ks_a"], K["ks_b"], K["ks_var"])
    mode = ("auto", "batch", "batch64", "split")[int(rng.integers(0, 4))]
    if mode == "split" and B > 1100: mode = "batch64"
    eng.set_blind_rotate_mode(mode)
    mk = lambda s, shape: orc.uniform_i32(seed + s, shape)
    ca, cb, cc = (mk(1, (B, n)), mk(2, (B,))), (mk(3, (B, n)), mk(4, (B,))), (mk(5, (B, n)), mk(6, (B,)))
    # force special rotation amounts into some coefficients: abar = 0, -1024 (x = 2^31 rounds to -1024), 1023
    if rng.integers(0, 2):
        ca[0][:, 0] = 0; cb[0][:, 0] = 0
        if n > 1: ca[0][:, 1] = np.int32(-2**31); cb[0][:, 1] = 0
    g = orc.GATES[int(rng.integers(0, 11))]
    oa, ob, var = eng.gate(g, ca, cb, cc if g == "mux" else None, want_var=True)
    for i in rng.choice(B, size=min(B, 6), replace=False):
        wa, wb, wv = orc.gate(P, g, K, (ca[0][i], ca[1][i], 0.0), (cb[0][i], cb[1][i], 0.0), (cc[0][i], cc[1][i], 0.0), orc.MUL_EXACT)
        if not ((oa[i] == wa).all() and ob[i] == wb and var[i] == wv):
            bad += 1; print("MISMATCH", tag, n, B, g, mode, seed, i, flush=True)
    trials += 1
    if trials % 20 == 0: print("trials", trials, "bad", bad, flush=True)
print("DONE trials", trials, "mismatches", bad)
