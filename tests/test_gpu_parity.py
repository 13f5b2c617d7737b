"""GPU parity: every stage of the bootstrapped path, called through the C ABI
(libtfhe_hip.so), against the CPU oracle on the same seeded inputs.

Bar: bit-exact int32 results.  The oracle runs in exact-integer mode, i.e. it
returns what the reference's f64 FFT returns whenever that FFT's error is below
0.5 (measured margin on these inputs: ~0.015, see test_oracle_kat.py).
"""
import numpy as np
import pytest

import _oracle as orc

pytestmark = pytest.mark.gpu

SHAPES = {"l2": (2, 10), "l3": (3, 7)}


@pytest.fixture(scope="module", params=["batch", "batch64", "split"])
def eng(request):
    """Every test runs against all three mappings of the blind rotation (one ciphertext per
    wavefront with one or with two wavefronts per SIMD / one per workgroup); results must be
    identical."""
    import tfhe_amd
    e = tfhe_amd.Engine(0)
    e.set_blind_rotate_mode(request.param)
    yield e
    e.close()


def _hip_params(P):
    import tfhe_amd
    return tfhe_amd.make_params(P.n, P.N, P.k, P.l, P.bg_bit, P.ks_t, P.ks_base_bit, P.n_ks)


def _load(eng, P, K):
    eng.load_keys(_hip_params(P), K["bk"], K["ks_a"], K["ks_b"], K.get("ks_var"))


# ---- isolated polynomial product ------------------------------------------------

@pytest.mark.parametrize("half_bg,B", [(512, 37), (64, 8), (512, 1), (512, 2)])
def test_polymul_full_and_accumulate(eng, half_bg, B):
    d = orc.range_i32(100 + B, -half_bg, half_bg, (B, 1024))
    s = orc.uniform_i32(200 + B, (B, 1024))
    got = eng.polymul(d, s)
    assert got.shape == (B, 2047)
    for i in range(B):
        want, _ = orc.poly_multiplier(d[i], s[i], orc.MUL_EXACT)
        assert (got[i] == want).all(), i
    r0 = orc.uniform_i32(300 + B, (B, 1024))
    got = eng.polymul(d, s, accumulate_into=r0)
    for i in range(B):
        want, _ = orc.mul_r(r0[i], d[i], s[i], orc.MUL_EXACT)
        assert (got[i] == want).all(), i


def test_polymul_saturation_and_edges(eng):
    B = 6
    d = np.zeros((B, 1024), np.int32)
    s = np.zeros((B, 1024), np.int32)
    # exact coefficients straddling +-2^31 and +-(2^31 - 1)
    d[0, 0], d[0, 1], s[0, 0], s[0, 1], s[0, 2] = 1, 1, 2**31 - 1, 1, 5
    d[1, 0], d[1, 1], s[1, 0], s[1, 1], s[1, 2] = 1, 1, -2**31, -1, -5
    d[2, 0], s[2, 0] = 1, 2**31 - 1
    d[3, 0], s[3, 0] = -1, -2**31            # +2^31 saturates to 2^31-1
    d[4, 1023], s[4, 1023] = -512, -2**31    # top coefficient of the 2047-long product
    d[5, :] = -512
    s[5, :] = 3                               # all-negative ramp, no saturation
    got = eng.polymul(d, s)
    for i in range(B):
        want, _ = orc.poly_multiplier(d[i], s[i], orc.MUL_EXACT)
        assert (got[i] == want).all(), i
    # zeros in, zeros out (reference KAT /root/reference/src/numerics.rs:508-515)
    z = eng.polymul(np.zeros((3, 1024), np.int32), np.zeros((3, 1024), np.int32))
    assert z.shape == (3, 2047) and not z.any()


# ---- external product, blind rotate -----------------------------------------------

@pytest.mark.parametrize("tag", ["l2", "l3"])
def test_extprod(eng, tag):
    l, bg = SHAPES[tag]
    P = orc.small_params(n=4, l=l, bg_bit=bg)
    K = orc.generate_synthetic_keys(P, 7)
    _load(eng, P, K)
    B = 9
    t = orc.uniform_i32(17, (B, 3, 1024))
    t[0] = 0                      # digits of zero are not all zero (level >= 1 gives -Bg/2)
    for idx in (0, 3):
        got = eng.extprod(t, idx)
        for b in range(B):
            want, _, _ = orc.extern_mul(P, t[b], K["bk"][idx], None, orc.MUL_EXACT)
            assert (got[b] == want).all(), (idx, b)


@pytest.mark.parametrize("tag", ["l2", "l3"])
def test_blind_rotate_hook(eng, tag):
    l, bg = SHAPES[tag]
    P = orc.small_params(n=7, l=l, bg_bit=bg)
    K = orc.generate_synthetic_keys(P, 8)
    _load(eng, P, K)
    B = 5
    acc = orc.uniform_i32(21, (B, 3, 1024))
    bara = orc.range_i32(22, -1024, 1024, (B, P.n))
    bara[0, :] = [1, -1, 0, 1023, -1024, -1023, 512]
    bara[1, :] = 0                                   # every iteration skipped
    got = eng.blind_rotate(acc, bara)
    for b in range(B):
        want, _, _ = orc.blind_rotate(P, acc[b], bara[b], K["bk"], None, orc.MUL_EXACT)
        assert (got[b] == want).all(), b


# ---- bootstrap, key switch, gates on non-degenerate keys ----------------------------

@pytest.mark.parametrize("tag", ["l2", "l3"])
def test_bootstrap_small_n(eng, tag):
    l, bg = SHAPES[tag]
    P = orc.small_params(n=8, l=l, bg_bit=bg)
    K = orc.generate_synthetic_keys(P, 9)
    _load(eng, P, K)
    B = 6
    a = orc.uniform_i32(31, (B, P.n))
    b = orc.uniform_i32(32, (B,))
    b[0], b[1] = -(1 << 20), -(1 << 20) - 1           # barb = 0 and barb = -1
    mu = 1 << 29
    ea, eb = eng.bootstrap_noks(mu, a, b)
    oa, ob, var = eng.bootstrap(mu, a, b, want_var=True)
    for i in range(B):
        wa, wb, _ = orc.bootstrap_noks(P, K["bk"], mu, a[i], b[i], None, orc.MUL_EXACT)
        assert (ea[i] == wa).all() and eb[i] == wb, i
        wa, wb, wv, _ = orc.bootstrap(P, K["bk"], K["ks_a"], K["ks_b"], mu, a[i], b[i], None,
                                      K["ks_var"], orc.MUL_EXACT)
        assert (oa[i] == wa).all() and ob[i] == wb, i
        assert var[i] == wv, i


@pytest.mark.parametrize("n,n_ks", [(20, 20), (500, 500), (630, 630), (13, 1024)])
def test_keyswitch(eng, n, n_ks):
    P = orc.small_params(n=n)
    P.n_ks = n_ks
    P.l, P.bg_bit = 2, 10
    K = orc.generate_synthetic_keys(P, 40 + n)
    K["ks_a"][:, :, 0] = 0                            # the reference's h = 0 entries
    K["ks_b"][:, :, 0] = 0
    K["ks_var"][:, :, 0] = 0
    # a tiny bootstrapping key: only the key-switch key matters here
    eng.load_keys(_hip_params(P), K["bk"], K["ks_a"], K["ks_b"], K["ks_var"])
    B = 5
    a = orc.uniform_i32(51, (B, 1024))
    a[0, :3] = [2**31 - 1, -2**31, (2**31 - 1) - (1 << 15)]   # aibar wraps
    a[1, :] = 0
    b = orc.uniform_i32(52, (B,))
    oa, ob, var = eng.keyswitch(a, b, want_var=True)
    for i in range(B):
        wa, wb, wv = orc.key_switch(P, K["ks_a"], K["ks_b"], a[i], b[i], K["ks_var"])
        assert (oa[i] == wa).all() and ob[i] == wb, i
        assert var[i] == wv, i


@pytest.mark.parametrize("n,ks_t", [(20, 8), (630, 8), (33, 5), (16, 11)])
def test_keyswitch_large_batch(n, ks_t):
    """Batches of at least 8 ciphertexts per compute unit take the tiled key switch (one
    pass over the key for 8 ciphertexts), for the crate's t = 8 and for other t; a ragged
    batch must equal the per-ciphertext kernel (same inputs in small chunks) and the oracle."""
    import tfhe_amd
    P = orc.small_params(n=n)
    P.l, P.bg_bit = 2, 10
    P.ks_t = ks_t
    K = orc.generate_synthetic_keys(P, 140 + n)
    B = 2051
    a = orc.uniform_i32(141, (B, 1024))
    b = orc.uniform_i32(142, (B,))
    a[7, :] = 0
    e = tfhe_amd.Engine(0)
    try:
        e.load_keys(_hip_params(P), K["bk"], K["ks_a"], K["ks_b"], K["ks_var"])
        oa, ob, var = e.keyswitch(a, b, want_var=True)
        for lo in range(0, B, 700):
            ca, cb, cv = e.keyswitch(a[lo:lo + 700], b[lo:lo + 700], want_var=True)
            assert (oa[lo:lo + 700] == ca).all() and (ob[lo:lo + 700] == cb).all() and (var[lo:lo + 700] == cv).all()
        # mux form (two inputs and the constant) on the tiled path
        ca = (orc.uniform_i32(143, (B, P.n)), orc.uniform_i32(144, (B,)))
        cb = (orc.uniform_i32(145, (B, P.n)), orc.uniform_i32(146, (B,)))
        cc = (orc.uniform_i32(147, (B, P.n)), orc.uniform_i32(148, (B,)))
        ma, mb, mv = e.gate("mux", ca, cb, cc, want_var=True)
    finally:
        e.close()
    for i in (0, 7, B - 1):
        wa, wb, wv = orc.key_switch(P, K["ks_a"], K["ks_b"], a[i], b[i], K["ks_var"])
        assert (oa[i] == wa).all() and ob[i] == wb and var[i] == wv, i
        wa, wb, wv = orc.gate(P, "mux", K, (ca[0][i], ca[1][i], 0.0), (cb[0][i], cb[1][i], 0.0),
                              (cc[0][i], cc[1][i], 0.0), orc.MUL_EXACT)
        assert (ma[i] == wa).all() and mb[i] == wb and mv[i] == wv, i


@pytest.mark.parametrize("n", [20, 630])
def test_keyswitch_largest_batches(n):
    """From 64 ciphertexts per compute unit on (t = 8) the key switch runs in its 32-ciphertext form,
    every wavefront selecting the key rows through an LDS area of its own.  A ragged batch of that
    size must equal the same inputs switched in chunks that take the other kernels, and the
    oracle; so must the mux form (two inputs and the constant)."""
    import tfhe_amd
    P = orc.small_params(n=n)
    P.l, P.bg_bit = 2, 10
    K = orc.generate_synthetic_keys(P, 150 + n)
    B = 16384 + 37
    a = orc.uniform_i32(151, (B, 1024))
    b = orc.uniform_i32(152, (B,))
    a[7, :] = 0
    a[B - 1, :] = -1
    e = tfhe_amd.Engine(0)
    try:
        e.load_keys(_hip_params(P), K["bk"], K["ks_a"], K["ks_b"], K["ks_var"])
        oa, ob, var = e.keyswitch(a, b, want_var=True)
        for lo in (0, 2051, 9000, B - 1500):
            ca, cb, cv = e.keyswitch(a[lo:lo + 1500], b[lo:lo + 1500], want_var=True)
            assert (oa[lo:lo + 1500] == ca).all() and (ob[lo:lo + 1500] == cb).all() and (var[lo:lo + 1500] == cv).all()
        if n == 20:
            ca = (orc.uniform_i32(153, (B, P.n)), orc.uniform_i32(154, (B,)))
            cb = (orc.uniform_i32(155, (B, P.n)), orc.uniform_i32(156, (B,)))
            cc = (orc.uniform_i32(157, (B, P.n)), orc.uniform_i32(158, (B,)))
            ma, mb, mv = e.gate("mux", ca, cb, cc, want_var=True)
    finally:
        e.close()
    for i in (0, 7, 31, 32, 16383, 16384, B - 1):
        wa, wb, wv = orc.key_switch(P, K["ks_a"], K["ks_b"], a[i], b[i], K["ks_var"])
        assert (oa[i] == wa).all() and ob[i] == wb and var[i] == wv, i
        if n == 20:
            wa, wb, wv = orc.gate(P, "mux", K, (ca[0][i], ca[1][i], 0.0), (cb[0][i], cb[1][i], 0.0),
                                  (cc[0][i], cc[1][i], 0.0), orc.MUL_EXACT)
            assert (ma[i] == wa).all() and mb[i] == wb and mv[i] == wv, i


@pytest.mark.parametrize("tag", ["l2", "l3"])
def test_gates_small_n(eng, tag):
    l, bg = SHAPES[tag]
    P = orc.small_params(n=6, l=l, bg_bit=bg)
    K = orc.generate_synthetic_keys(P, 60)
    _load(eng, P, K)
    B = 3
    ca = (orc.uniform_i32(61, (B, P.n)), orc.uniform_i32(62, (B,)))
    cb = (orc.uniform_i32(63, (B, P.n)), orc.uniform_i32(64, (B,)))
    cc = (orc.uniform_i32(65, (B, P.n)), orc.uniform_i32(66, (B,)))
    for g in orc.GATES[:11]:
        third = cc if g == "mux" else None
        oa, ob, var = eng.gate(g, ca, cb, third, want_var=True)
        for i in range(B):
            sa = (ca[0][i], ca[1][i], 0.0)
            sb = (cb[0][i], cb[1][i], 0.0)
            sc = (cc[0][i], cc[1][i], 0.0)
            wa, wb, wv = orc.gate(P, g, K, sa, sb, sc, orc.MUL_EXACT)
            assert (oa[i] == wa).all() and ob[i] == wb, (g, i)
            assert var[i] == wv, (g, i)


# ---- circuits::add, level-batched ---------------------------------------------------

def _oracle_add(P, K, a, b, mode):
    """full_adder / add of /root/reference/src/circuits/circuits.rs:83-120 as a
    composition of oracle gates; a, b: lists of (a, b, var) samples, bit 0 first."""
    carry = (np.zeros(P.n, np.int32), np.int32(-(1 << 29)), 0.0)      # constant(false)
    sums = []
    for x, y in zip(a, b):
        a_b_xor = orc.gate(P, "xor", K, x, y, None, mode)
        s_ = orc.gate(P, "xor", K, a_b_xor, carry, None, mode)
        t = orc.gate(P, "and", K, carry, a_b_xor, None, mode)
        g = orc.gate(P, "and", K, x, y, None, mode)
        carry = orc.gate(P, "or", K, t, g, None, mode)
        sums.append(s_)
    return sums, carry


@pytest.mark.parametrize("tag,bits,count", [("l2", 4, 5), ("l3", 3, 2), ("l2", 0, 3), ("l2", 1, 1)])
def test_circuit_add(eng, tag, bits, count):
    l, bg = SHAPES[tag]
    P = orc.small_params(n=6, l=l, bg_bit=bg)
    K = orc.generate_synthetic_keys(P, 80)
    _load(eng, P, K)
    A = (orc.uniform_i32(81, (bits, count, P.n)), orc.uniform_i32(82, (bits, count)))
    Bv = (orc.uniform_i32(83, (bits, count, P.n)), orc.uniform_i32(84, (bits, count)))
    (sa, sb), (ca, cb), sv, cv = eng.circuit_add(A, Bv, want_var=True)
    for c in range(count):
        a = [(A[0][i, c], A[1][i, c], 0.0) for i in range(bits)]
        b = [(Bv[0][i, c], Bv[1][i, c], 0.0) for i in range(bits)]
        sums, carry = _oracle_add(P, K, a, b, orc.MUL_EXACT)
        for i in range(bits):
            assert (sa[i, c] == sums[i][0]).all() and sb[i, c] == sums[i][1] and sv[i, c] == sums[i][2], (c, i)
        assert (ca[c] == carry[0]).all() and cb[c] == carry[1] and cv[c] == carry[2], c


# ---- full-size parameter sets -------------------------------------------------------

@pytest.fixture(scope="module")
def ref_keys80():
    P = orc.params(80)
    return P, orc.generate_keys(P, 0xC0FFEE)


def test_nand_on_reference_style_keys(eng, ref_keys80):
    """Keys generated the way the reference generates them: the external
    product is identically zero, the result is the noiseless trivial sample
    +-mu (sign of b + 2^20), and the decrypted bit must match the oracle's."""
    P, K = ref_keys80
    _load(eng, P, K)
    B = 8
    bits = [(i & 1, (i >> 1) & 1) for i in range(B)]
    ca = [orc.encrypt(P, K["lwe_key"], x, 500 + i) for i, (x, _) in enumerate(bits)]
    cb = [orc.encrypt(P, K["lwe_key"], y, 600 + i) for i, (_, y) in enumerate(bits)]
    A = (np.stack([c[0] for c in ca]), np.array([c[1] for c in ca], np.int32))
    Bm = (np.stack([c[0] for c in cb]), np.array([c[1] for c in cb], np.int32))
    oa, ob, var = eng.gate("nand", A, Bm, want_var=True)
    mu = 1 << 29
    for i in range(B):
        wa, wb, wv = orc.gate(P, "nand", K, ca[i], cb[i], None, orc.MUL_FFT)
        assert (oa[i] == wa).all() and ob[i] == wb and var[i] == wv, i
        assert not oa[i].any() and abs(int(ob[i])) == mu
        assert orc.decrypt(oa[i], ob[i], K["lwe_key"]) == orc.decrypt(wa, wb, K["lwe_key"])


@pytest.mark.parametrize("level,B", [(80, 2), (128, 1)])
def test_bootstrap_full_size_synthetic_keys(eng, level, B):
    P = orc.params(level)
    K = orc.generate_synthetic_keys(P, 70 + level)
    _load(eng, P, K)
    a = orc.uniform_i32(71, (B, P.n))
    b = orc.uniform_i32(72, (B,))
    mu = 1 << 29
    oa, ob, var = eng.bootstrap(mu, a, b, want_var=True)
    for i in range(B):
        wa, wb, wv, _ = orc.bootstrap(P, K["bk"], K["ks_a"], K["ks_b"], mu, a[i], b[i], None,
                                      K["ks_var"], orc.MUL_EXACT)
        assert (oa[i] == wa).all() and ob[i] == wb and var[i] == wv, i


@pytest.mark.parametrize("level,B", [(80, 700), (80, 900), (80, 2049), (80, 2648), (80, 3300), (128, 1030), (128, 2700)])
def test_blind_rotate_mappings_agree_at_full_size(level, B):
    """Size-independent property at the bench's parameters: the two throughput mappings and
    the low-latency mapping return the same bits on a batch that spans several
    rounds of persistent workgroups (and the automatic choice equals all of them).  The sizes
    walk the automatic mode's plans on a GPU of 256 compute units: 700 gates run in three rounds
    of the low-latency mapping, 900 with one wavefront per SIMD, 1 030 as 1 024 of those and
    6 in the low-latency mapping; after a full round of 2 048 with two wavefronts per SIMD a
    remainder of 1, 600 or 652 goes to the low-latency mapping and one of 1 252 is divided
    between the one-wavefront-per-SIMD mapping (1 024) and the low-latency one (228)."""
    import tfhe_amd
    P = orc.params(level)
    K = orc.generate_synthetic_keys(P, 90 + level)
    ca = (orc.uniform_i32(91, (B, P.n)), orc.uniform_i32(92, (B,)))
    cb = (orc.uniform_i32(93, (B, P.n)), orc.uniform_i32(94, (B,)))
    out = {}
    e = tfhe_amd.Engine(0)
    try:
        _load(e, P, K)
        for mode in ("batch", "batch64", "split", "auto"):
            e.set_blind_rotate_mode(mode)
            out[mode] = e.gate("xor", ca, cb, want_var=True)
    finally:
        e.close()
    for mode in ("batch64", "split", "auto"):
        for x, y in zip(out["batch"], out[mode]):
            assert (x == y).all(), mode
    # and two ciphertexts of the batch against the oracle
    for i in (0, B - 1):
        wa, wb, wv = orc.gate(P, "xor", K, (ca[0][i], ca[1][i], 0.0), (cb[0][i], cb[1][i], 0.0), None,
                              orc.MUL_EXACT)
        assert (out["batch"][0][i] == wa).all() and out["batch"][1][i] == wb and out["batch"][2][i] == wv


def test_random_small_configurations(eng):
    """Seeded sweep over ragged shapes: n from 1, batches from 1 (including sizes that leave
    wavefront slots of the last workgroup empty), both decomposition shapes, every gate kind."""
    rng = np.random.default_rng(20240607)
    for trial in range(12):
        tag = ("l2", "l3")[trial % 2]
        l, bg = SHAPES[tag]
        n = int(rng.integers(1, 14))
        B = int(rng.integers(1, 11))
        P = orc.small_params(n=n, l=l, bg_bit=bg)
        K = orc.generate_synthetic_keys(P, 300 + trial)
        _load(eng, P, K)
        ca = (orc.uniform_i32(400 + trial, (B, n)), orc.uniform_i32(500 + trial, (B,)))
        cb = (orc.uniform_i32(600 + trial, (B, n)), orc.uniform_i32(700 + trial, (B,)))
        cc = (orc.uniform_i32(800 + trial, (B, n)), orc.uniform_i32(900 + trial, (B,)))
        g = orc.GATES[trial % 11]
        oa, ob, var = eng.gate(g, ca, cb, cc if g == "mux" else None, want_var=True)
        for i in range(B):
            wa, wb, wv = orc.gate(P, g, K, (ca[0][i], ca[1][i], 0.0), (cb[0][i], cb[1][i], 0.0),
                                  (cc[0][i], cc[1][i], 0.0), orc.MUL_EXACT)
            assert (oa[i] == wa).all() and ob[i] == wb and var[i] == wv, (trial, tag, n, B, g, i)


def test_full_batch_properties():
    """The bench's batch size (16 384 gates, Bit80): properties that need no oracle.
    Symmetric gates do not care about operand order (their prologue is a wrapping sum),
    repeating a ciphertext inside the batch repeats its result, and a second run of the
    same batch returns the same bits; one ciphertext is checked against the oracle."""
    import tfhe_amd
    P = orc.params(80)
    K = orc.generate_synthetic_keys(P, 170)
    B = 16384
    ca = (orc.uniform_i32(171, (B, P.n)), orc.uniform_i32(172, (B,)))
    cb = (orc.uniform_i32(173, (B, P.n)), orc.uniform_i32(174, (B,)))
    for arr_a, arr_b in ((ca[0], cb[0]), (ca[1], cb[1])):      # ciphertext B-1 repeats ciphertext 5
        arr_a[B - 1] = arr_a[5]
        arr_b[B - 1] = arr_b[5]
    e = tfhe_amd.Engine(0)
    try:
        _load(e, P, K)
        r1 = e.gate("nand", ca, cb, want_var=True)
        r2 = e.gate("nand", cb, ca, want_var=True)
        r3 = e.gate("nand", ca, cb, want_var=True)
    finally:
        e.close()
    for x, y, z in zip(r1, r2, r3):
        assert (x == y).all() and (x == z).all()
    assert (r1[0][B - 1] == r1[0][5]).all() and r1[1][B - 1] == r1[1][5] and r1[2][B - 1] == r1[2][5]
    i = B // 2 + 1
    wa, wb, wv = orc.gate(P, "nand", K, (ca[0][i], ca[1][i], 0.0), (cb[0][i], cb[1][i], 0.0), None, orc.MUL_EXACT)
    assert (r1[0][i] == wa).all() and r1[1][i] == wb and r1[2][i] == wv


def test_errors_are_reported_not_thrown(eng):
    import ctypes as C
    import tfhe_amd
    bad = tfhe_amd.make_params(8, N=512)
    rc = eng.lib.tfhe_hip_load_keys(eng.ctx, C.byref(bad), None, None, None, None)
    assert rc == -1
    assert b"N=1024" in eng.lib.tfhe_hip_last_error(eng.ctx)
    with pytest.raises(tfhe_amd.TfheHipError):
        eng._check(rc)


def test_abi_edge_cases():
    """Empty batches, missing keys, NULL pointers and bad selectors come back as
    error codes with a message (the crate panics; nothing may cross the C ABI)."""
    import ctypes as C
    import tfhe_amd
    e = tfhe_amd.Engine(0)
    try:
        lib, ctx = e.lib, e.ctx
        # no keys yet: state error
        assert lib.tfhe_hip_gate(ctx, 0, None, None, None, None, None, None, None, None, None, 1) == -2
        assert b"no keys" in lib.tfhe_hip_last_error(ctx)
        assert lib.tfhe_hip_set_blind_rotate_mode(ctx, 7) == -1
        P = orc.small_params(n=4)
        K = orc.generate_synthetic_keys(P, 3)
        _load(e, P, K)
        # empty batch: success, pointers untouched
        assert lib.tfhe_hip_gate(ctx, 0, None, None, None, None, None, None, None, None, None, 0) == 0
        assert lib.tfhe_hip_bootstrap(ctx, 1 << 29, None, None, None, None, None, 0) == 0
        assert lib.tfhe_hip_circuit_add(ctx, None, None, None, None, None, None, None, None, None, None, 0, 8) == 0
        # NULL operands, unknown gate, mux without its third operand, negative width
        assert lib.tfhe_hip_gate(ctx, 0, None, None, None, None, None, None, None, None, None, 1) == -1
        a = np.zeros((1, P.n), np.int32)
        b = np.zeros(1, np.int32)
        pa, pb = a.ctypes.data_as(C.c_void_p), b.ctypes.data_as(C.c_void_p)
        assert lib.tfhe_hip_gate(ctx, 99, pa, pb, pa, pb, None, None, pa, pb, None, 1) == -1
        assert b"unknown gate" in lib.tfhe_hip_last_error(ctx)
        assert lib.tfhe_hip_gate(ctx, 10, pa, pb, pa, pb, None, None, pa, pb, None, 1) == -1
        assert lib.tfhe_hip_circuit_add(ctx, pa, pb, pa, pb, pa, pb, pa, pb, None, None, 1, -1) == -1
        assert lib.tfhe_hip_extprod(ctx, pa, 99, pa, 1) == -1
        # the context still works after the errors
        oa, ob = e.gate("nand", (a, b), (a, b))
        wa, wb, _ = orc.gate(P, "nand", K, (a[0], b[0], 0.0), (a[0], b[0], 0.0))
        assert (oa[0] == wa).all() and ob[0] == wb
    finally:
        e.close()
