"""GPU parity at the sizes the benches run at, and of the host layer: persistent
loops that take several trips, key blobs moved between contexts, the Bit128
parameter set at the full batch, 32-bit adders, the rounding margin measured in
the kernels, device-resident sample handles, the key file and multi-GPU
contexts.  Everything goes through the C ABI (libtfhe_hip.so); expected values
come from the CPU oracle on the same seeded inputs.

Oracle modes: MUL_EXACT is the exact integer convolution (ground truth);
MUL_FFT is the oracle's own f64 FFT, which returns the same integers whenever
its rounding margin stays below 0.5 (asserted where it is used for speed).
"""
import os
import subprocess
import sys

import numpy as np
import pytest

import _oracle as orc

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _hip_params(P):
    import tfhe_amd
    return tfhe_amd.make_params(P.n, P.N, P.k, P.l, P.bg_bit, P.ks_t, P.ks_base_bit, P.n_ks)


def _load(eng, P, K):
    eng.load_keys(_hip_params(P), K["bk"], K["ks_a"], K["ks_b"], K.get("ks_var"))


@pytest.fixture(scope="module")
def eng():
    import tfhe_amd
    e = tfhe_amd.Engine(0)
    yield e
    e.close()


# ---- isolated product: persistent loop, config 2 of the baseline ---------------------

def test_polymul_ragged_batch_takes_several_trips(eng):
    """4 099 products = 2 050 wavefront pairs on at most 1 024 wavefront slots: every slot
    takes two or three trips of the persistent loop and the last pair is half empty.  Every
    product is checked, in the full and in the accumulate form."""
    B = 4099
    d = orc.range_i32(1100, -512, 512, (B, 1024))
    s = orc.uniform_i32(1101, (B, 1024))
    got = eng.polymul(d, s)
    r0 = orc.uniform_i32(1102, (B, 1024))
    acc = eng.polymul(d, s, accumulate_into=r0)
    for i in range(B):
        want, _ = orc.poly_multiplier(d[i], s[i], orc.MUL_EXACT)
        assert (got[i] == want).all(), i
        assert (acc[i] == (r0[i].astype(np.int64) + want[:1024]).astype(np.int32)).all(), i


@pytest.mark.parametrize("half_bg", [512, 64])
def test_polymul_65536(eng, half_bg):
    """The bench's batch (65 536 products): a seeded sample of 512 products against the exact
    oracle, plus two properties over the whole batch: the first N coefficients of the full form
    equal the accumulate form on a zero accumulator, and repeated inputs repeat their outputs."""
    B = 65536
    d = orc.range_i32(1200 + half_bg, -half_bg, half_bg, (B, 1024))
    s = orc.uniform_i32(1201 + half_bg, (B, 1024))
    d[B - 1], s[B - 1] = d[17], s[17]
    got = eng.polymul(d, s)
    acc = eng.polymul(d, s, accumulate_into=np.zeros((B, 1024), np.int32))
    assert (got[:, :1024] == acc).all()
    assert (got[B - 1] == got[17]).all()
    rng = np.random.default_rng(7)
    for i in rng.choice(B, size=512, replace=False):
        want, _ = orc.poly_multiplier(d[i], s[i], orc.MUL_EXACT)
        assert (got[i] == want).all(), i


# ---- rounding margin, measured in the kernels ----------------------------------------

@pytest.mark.parametrize("half_bg", [512, 64])
def test_polymul_rounding_margin(eng, half_bg):
    """|x - rint(x)| before rounding, from the product kernel's own FFT, on digit x uniform
    inputs: far below the 0.5 at which two correct f64 FFTs could disagree."""
    B = 256
    d = orc.range_i32(1300 + half_bg, -half_bg, half_bg, (B, 1024))
    s = orc.uniform_i32(1301 + half_bg, (B, 1024))
    m = eng.polymul_margin(d, s)
    assert m.shape == (B, 2)
    # An FFT's error is absolute (set by the largest values in the transform, about 2^45 here,
    # times 2^-53 and a few units), so small coefficients see the same margin as large ones.
    assert m[:, 0].max() < 0.25 and m[:, 1].max() < 0.05, m.max(axis=0)
    print(f"\npolymul margin, |d| < {half_bg}: all {m[:, 0].max():.3g}, deciding {m[:, 1].max():.3g}")


def test_polymul_all_max_stress_reports_margin(eng):
    """All-maximum operands: |x| reaches 2^50, where one unit in the last place of an f64 is
    0.25, so the margin over all coefficients is expected to reach the guard; those
    coefficients saturate whatever their fraction is.  Reported, and the results must still
    equal the exact oracle wherever the deciding margin holds."""
    pats = []
    for dv, sv in ((-512, -2**31), (511, 2**31 - 1), (-512, 2**31 - 1), (511, -2**31)):
        pats.append((np.full(1024, dv, np.int32), np.full(1024, sv, np.int32)))
    alt = np.where(np.arange(1024) % 2 == 0, 511, -512).astype(np.int32)
    pats.append((alt, np.full(1024, 2**31 - 1, np.int32)))
    pats.append((alt, np.where(np.arange(1024) % 2 == 0, 2**31 - 1, -2**31).astype(np.int32)))
    d = np.stack([p[0] for p in pats])
    s = np.stack([p[1] for p in pats])
    m = eng.polymul_margin(d, s)
    got = eng.polymul(d, s)
    mismatches = 0
    for i in range(len(pats)):
        want, _ = orc.poly_multiplier(d[i], s[i], orc.MUL_EXACT)
        mismatches += int((got[i] != want).sum())
    print(f"\nall-max stress: margin over all coefficients {m[:, 0].max():.3g}, over deciding "
          f"coefficients {m[:, 1].max():.3g}, mismatches vs exact oracle {mismatches}")
    if m[:, 1].max() < 0.25:
        assert mismatches == 0


@pytest.mark.parametrize("level", [80, 128])
def test_extprod_rounding_margin_full_size_keys(eng, level):
    """The same margin inside the external product of the blind rotation (level_product), on
    uniform accumulators and uniform keys of the real parameter sets."""
    P = orc.params(level)
    P.n = P.n_ks = 3
    K = orc.generate_synthetic_keys(P, 1400 + level)
    _load(eng, P, K)
    B = 64
    t = orc.uniform_i32(1401, (B, 3, 1024))
    out, m = eng.extprod(t, 1, want_margin=True)
    assert m[:, 0].max() < 0.25 and m[:, 1].max() < 0.05, m.max(axis=0)
    print(f"\nextprod margin Bit{level}: all {m[:, 0].max():.3g}, deciding {m[:, 1].max():.3g}")
    for b in (0, B - 1):
        want, _, _ = orc.extern_mul(P, t[b], K["bk"][1], None, orc.MUL_EXACT)
        assert (out[b] == want).all(), b


# ---- blind-rotate hook: any i32 rotation amount ---------------------------------------

@pytest.mark.parametrize("mode", ["batch", "batch64", "split"])
@pytest.mark.parametrize("tag", [(2, 10), (3, 7)])
def test_blind_rotate_hook_any_power(mode, tag):
    """mul_by_monomial reduces its power with (power / N) mod 2 and power % N
    (/root/reference/src/polynomial.rs:289-291); the hook takes caller-supplied powers, so
    amounts in (N, 2N), below -N and far outside are checked against the oracle."""
    import tfhe_amd
    l, bg = tag
    P = orc.small_params(n=7, l=l, bg_bit=bg)
    K = orc.generate_synthetic_keys(P, 1500)
    bara = np.array([[1025, 2047, 1536, -1025, -2047, -2048, 2048],
                     [3000, -3000, 4101, -4101, 2**31 - 1, -2**31, 1024],
                     [-1024, 1023, -1, 1, 0, 3072, -3072]], np.int32)
    acc = orc.uniform_i32(1501, (3, 3, 1024))
    e = tfhe_amd.Engine(0)
    try:
        e.set_blind_rotate_mode(mode)
        _load(e, P, K)
        got = e.blind_rotate(acc, bara)
    finally:
        e.close()
    for b in range(3):
        want, _, _ = orc.blind_rotate(P, acc[b], bara[b], K["bk"], None, orc.MUL_EXACT)
        assert (got[b] == want).all(), b


# ---- key blobs between contexts (what the multi-process bench does over RCCL) ----------

@pytest.mark.parametrize("level", [80, 128])
def test_key_blob_round_trip_between_contexts(level):
    """Context A loads and transforms the keys, its device blobs are exported, context B
    imports them: NAND and mux outputs and variances of B equal A's bit for bit."""
    import torch
    import tfhe_amd
    P = orc.params(level)
    K = orc.generate_synthetic_keys(P, 1600 + level)
    hp = _hip_params(P)
    B = 6
    ca = (orc.uniform_i32(1601, (B, P.n)), orc.uniform_i32(1602, (B,)))
    cb = (orc.uniform_i32(1603, (B, P.n)), orc.uniform_i32(1604, (B,)))
    cc = (orc.uniform_i32(1605, (B, P.n)), orc.uniform_i32(1606, (B,)))
    a, b = tfhe_amd.Engine(0), tfhe_amd.Engine(0)
    try:
        a.load_keys(hp, K["bk"], K["ks_a"], K["ks_b"], K["ks_var"])
        bk_bytes, ks_bytes = a.key_blob_sizes()
        dev = torch.device("cuda", 0)
        bk_blob = torch.empty(bk_bytes, dtype=torch.uint8, device=dev)
        ks_blob = torch.empty(ks_bytes, dtype=torch.uint8, device=dev)
        a.export_keys_dev(bk_blob, ks_blob)
        b.import_keys_dev(hp, bk_blob, ks_blob)
        del bk_blob, ks_blob
        for g, third in (("nand", None), ("mux", cc)):
            ra = a.gate(g, ca, cb, third, want_var=True)
            rb = b.gate(g, ca, cb, third, want_var=True)
            for x, y in zip(ra, rb):
                assert (x == y).all(), g
        # and B's NAND against the oracle
        wa, wb, wv = orc.gate(P, "nand", K, (ca[0][0], ca[1][0], 0.0), (cb[0][0], cb[1][0], 0.0), None,
                              orc.MUL_FFT)
        rb = b.gate("nand", ca, cb, want_var=True)
        assert (rb[0][0] == wa).all() and rb[1][0] == wb and rb[2][0] == wv
    finally:
        a.close()
        b.close()


# ---- Bit128 (the crate's default parameters) at the full batch ---------------------------

def test_bit128_full_batch(eng):
    """16 384 NAND gates on Bit128 (n = 630, l = 3, Bg = 2^7; Parameters::default,
    /root/reference/src/lwe.rs:235-241): operand order and repetition properties over the whole
    batch, five ciphertexts against the oracle (one with the exact integer product)."""
    P = orc.params(128)
    K = orc.generate_synthetic_keys(P, 1700)
    _load(eng, P, K)
    B = 16384
    ca = (orc.uniform_i32(1701, (B, P.n)), orc.uniform_i32(1702, (B,)))
    cb = (orc.uniform_i32(1703, (B, P.n)), orc.uniform_i32(1704, (B,)))
    for x, y in ((ca[0], cb[0]), (ca[1], cb[1])):
        x[B - 1], y[B - 1] = x[9], y[9]
    r1 = eng.gate("nand", ca, cb, want_var=True)
    r2 = eng.gate("nand", cb, ca, want_var=True)
    for x, y in zip(r1, r2):
        assert (x == y).all()
    assert (r1[0][B - 1] == r1[0][9]).all() and r1[1][B - 1] == r1[1][9] and r1[2][B - 1] == r1[2][9]
    for k, i in enumerate((0, 1023, 1024, B // 2 + 3, B - 2)):
        wa, wb, wv = orc.gate(P, "nand", K, (ca[0][i], ca[1][i], 0.0), (cb[0][i], cb[1][i], 0.0), None,
                              orc.MUL_EXACT if k == 0 else orc.MUL_FFT)
        assert (r1[0][i] == wa).all() and r1[1][i] == wb and r1[2][i] == wv, i


def test_nand_on_reference_style_keys_bit128(eng):
    """Keys generated the way the crate generates them, Bit128: noiseless +-mu, the decrypted
    bit equals the oracle's."""
    P = orc.params(128)
    K = orc.generate_keys(P, 0xC0FFEE + 128)
    _load(eng, P, K)
    B = 8
    bits = [(i & 1, (i >> 1) & 1) for i in range(B)]
    ca = [orc.encrypt(P, K["lwe_key"], x, 1800 + i) for i, (x, _) in enumerate(bits)]
    cb = [orc.encrypt(P, K["lwe_key"], y, 1900 + i) for i, (_, y) in enumerate(bits)]
    A = (np.stack([c[0] for c in ca]), np.array([c[1] for c in ca], np.int32))
    Bm = (np.stack([c[0] for c in cb]), np.array([c[1] for c in cb], np.int32))
    oa, ob, var = eng.gate("nand", A, Bm, want_var=True)
    for i in range(B):
        wa, wb, wv = orc.gate(P, "nand", K, ca[i], cb[i], None, orc.MUL_FFT)
        assert (oa[i] == wa).all() and ob[i] == wb and var[i] == wv, i
        assert not oa[i].any() and abs(int(ob[i])) == 1 << 29
        assert orc.decrypt(oa[i], ob[i], K["lwe_key"]) == orc.decrypt(wa, wb, K["lwe_key"])


# ---- 32-bit adders at a batch that fills the GPU -----------------------------------------

def _oracle_add(P, K, a, b, mode, pool):
    """full_adder / add of /root/reference/src/circuits/circuits.rs:83-120 composed of oracle
    gates; independent gates run on the pool's threads (the oracle is a C library)."""
    axb = list(pool.map(lambda xy: orc.gate(P, "xor", K, xy[0], xy[1], None, mode), zip(a, b)))
    g = list(pool.map(lambda xy: orc.gate(P, "and", K, xy[0], xy[1], None, mode), zip(a, b)))
    carry = (np.zeros(P.n, np.int32), np.int32(-(1 << 29)), 0.0)      # constant(false)
    sums = []
    for i in range(len(a)):
        fs = pool.submit(orc.gate, P, "xor", K, axb[i], carry, None, mode)
        t = orc.gate(P, "and", K, carry, axb[i], None, mode)
        carry = orc.gate(P, "or", K, t, g[i], None, mode)
        sums.append(fs)
    return [f.result() for f in sums], carry


def test_circuit_add_32_bits_600_instances(eng):
    """circuits::add (/root/reference/src/circuits/circuits.rs:83-120) on 600 pairs of 32-bit
    operands, Bit80: 96 000 gates in 65 dependency levels; the carry-independent level alone
    is 38 400 ciphertexts in one blind rotation.  Two instances are checked end to end against
    the same circuit composed of oracle gates, sums, carry and variances."""
    P = orc.params(80)
    K = orc.generate_synthetic_keys(P, 2000)
    _load(eng, P, K)
    bits, count = 32, 600
    A = (orc.uniform_i32(2001, (bits, count, P.n)), orc.uniform_i32(2002, (bits, count)))
    Bv = (orc.uniform_i32(2003, (bits, count, P.n)), orc.uniform_i32(2004, (bits, count)))
    (sa, sb), (ca, cb), sv, cv = eng.circuit_add(A, Bv, want_var=True)
    from concurrent.futures import ThreadPoolExecutor

    def check(c, pool):
        a = [(A[0][i, c], A[1][i, c], 0.0) for i in range(bits)]
        b = [(Bv[0][i, c], Bv[1][i, c], 0.0) for i in range(bits)]
        sums, carry = _oracle_add(P, K, a, b, orc.MUL_FFT, pool)
        for i in range(bits):
            assert (sa[i, c] == sums[i][0]).all() and sb[i, c] == sums[i][1] and sv[i, c] == sums[i][2], (c, i)
        assert (ca[c] == carry[0]).all() and cb[c] == carry[1] and cv[c] == carry[2], c

    with ThreadPoolExecutor(6) as gates, ThreadPoolExecutor(2) as instances:
        for f in [instances.submit(check, c, gates) for c in (0, count - 1)]:
            f.result()


# ---- device-resident sample handles --------------------------------------------------------

def _chain_host(e, ca, cb, cc):
    x = e.gate("nand", ca, cb, want_var=True)
    y = e.gate("xor", (x[0], x[1]), cc, want_var=True)
    return e.gate("mux", (y[0], y[1]), ca, ( -x[0], -x[1]), want_var=True), y


def _chain_handles(e, ca, cb, cc):
    B = ca[1].size
    ha, hb, hc = e.samples_from(*ca), e.samples_from(*cb), e.samples_from(*cc)
    x, nx, y, z, one = (e.samples(B) for _ in range(5))
    e.samples_gate("nand", ha, hb, x)
    e.samples_not(x, nx)
    e.samples_gate("xor", x, hc, y)
    e.samples_gate("mux", y, ha, z, cc=nx)
    e.samples_constant(True, one)
    out = z.download(want_var=True), y.download(want_var=True), one.download(want_var=True), nx.download(want_var=True)
    for h in (ha, hb, hc, x, nx, y, z, one):
        h.free()
    return out


def test_sample_handles_chain_equals_host_calls(eng):
    """nand -> not / xor -> mux on handles that never leave the GPU equals the same chain
    through the host-pointer calls, masks, bodies and variances."""
    P = orc.small_params(n=9, l=2, bg_bit=10)
    K = orc.generate_synthetic_keys(P, 2100)
    _load(eng, P, K)
    B = 11
    ca = (orc.uniform_i32(2101, (B, P.n)), orc.uniform_i32(2102, (B,)))
    cb = (orc.uniform_i32(2103, (B, P.n)), orc.uniform_i32(2104, (B,)))
    cc = (orc.uniform_i32(2105, (B, P.n)), orc.uniform_i32(2106, (B,)))
    (z_want, y_want) = _chain_host(eng, ca, cb, cc)
    z, y, one, nx = _chain_handles(eng, ca, cb, cc)
    for got, want in ((z, z_want), (y, y_want)):
        for g, w in zip(got, want):
            assert (g == w).all()
    assert not one[0].any() and (one[1] == 1 << 29).all() and not one[2].any()
    x = eng.gate("nand", ca, cb, want_var=True)
    assert (nx[0] == -x[0]).all() and (nx[1] == -x[1]).all() and (nx[2] == x[2]).all()
    # the handle chain against the oracle composed gate by gate (src/gates.rs:23-41,83-99,122-124,239-268)
    for i in (0, B - 1):
        sa, sb, sc = ((c[0][i], c[1][i], 0.0) for c in (ca, cb, cc))
        ox = orc.gate(P, "nand", K, sa, sb, None, orc.MUL_EXACT)
        oy = orc.gate(P, "xor", K, ox, sc, None, orc.MUL_EXACT)
        onx = (-ox[0], (-ox[1] + 2 ** 31) % 2 ** 32 - 2 ** 31, ox[2])      # gates::not, wrapping
        oz = orc.gate(P, "mux", K, oy, sa, onx, orc.MUL_EXACT)
        assert (y[0][i] == oy[0]).all() and y[1][i] == oy[1] and y[2][i] == oy[2], i
        assert (z[0][i] == oz[0]).all() and z[1][i] == oz[1] and z[2][i] == oz[2], i
    # errors: size mismatch, foreign handle
    import tfhe_amd
    h1, h2 = eng.samples(3), eng.samples(4)
    with pytest.raises(tfhe_amd.TfheHipError):
        eng.samples_gate("nand", h1, h1, h2)
    h1.free()
    h2.free()


# ---- key file ---------------------------------------------------------------------------------

def test_key_file_loads_like_load_keys(eng):
    import tfhe_amd
    P = orc.small_params(n=10, l=3, bg_bit=7)
    K = orc.generate_synthetic_keys(P, 2200)
    hp = _hip_params(P)
    B = 4
    ca = (orc.uniform_i32(2201, (B, P.n)), orc.uniform_i32(2202, (B,)))
    cb = (orc.uniform_i32(2203, (B, P.n)), orc.uniform_i32(2204, (B,)))
    eng.load_keys(hp, K["bk"], K["ks_a"], K["ks_b"], K["ks_var"])
    want = eng.gate("xnor", ca, cb, want_var=True)
    image = eng.pack_key_file(hp, K["bk"], K["ks_a"], K["ks_b"], K["ks_var"])
    e2 = tfhe_amd.Engine(0)
    try:
        e2.load_key_file(image)
        assert e2.params.n == 10 and e2.params.l == 3
        got = e2.gate("xnor", ca, cb, want_var=True)
        for g, w in zip(got, want):
            assert (g == w).all()
        # the context that only ever saw the key file against the oracle on the original keys
        for i in range(B):
            wa, wb, wv = orc.gate(P, "xnor", K, (ca[0][i], ca[1][i], 0.0), (cb[0][i], cb[1][i], 0.0), None,
                                  orc.MUL_EXACT)
            assert (got[0][i] == wa).all() and got[1][i] == wb and got[2][i] == wv, i
        # an unaligned copy of the image loads too
        raw = np.empty(image.size + 1, np.uint8)
        raw[1:] = image
        e2.load_key_file(raw[1:])
        got = e2.gate("xnor", ca, cb, want_var=True)
        for g, w in zip(got, want):
            assert (g == w).all()
        bad = image.copy()
        bad[0] ^= 0xFF
        with pytest.raises(tfhe_amd.TfheHipError, match="magic"):
            e2.load_key_file(bad)
        with pytest.raises(tfhe_amd.TfheHipError, match="size"):
            e2.load_key_file(image[:-8])
    finally:
        e2.close()


# ---- multi-GPU context (two workers on the one GPU of the test box) ---------------------------

def test_multi_context_equals_single_context(eng):
    """tfhe_hip_create_multi with devices = {0, 0}: keys replicated between the workers,
    every batch split contiguously and run at the same time; results equal the single-device
    context bit for bit.  Ragged and tiny batches leave a worker with less or nothing."""
    import tfhe_amd
    P = orc.small_params(n=12, l=2, bg_bit=10)
    K = orc.generate_synthetic_keys(P, 2300)
    hp = _hip_params(P)
    eng.load_keys(hp, K["bk"], K["ks_a"], K["ks_b"], K["ks_var"])
    m = tfhe_amd.Engine([0, 0])
    try:
        m.load_keys(hp, K["bk"], K["ks_a"], K["ks_b"], K["ks_var"])
        for B in (1, 2, 7, 64):
            ca = (orc.uniform_i32(2301 + B, (B, P.n)), orc.uniform_i32(2302 + B, (B,)))
            cb = (orc.uniform_i32(2303 + B, (B, P.n)), orc.uniform_i32(2304 + B, (B,)))
            cc = (orc.uniform_i32(2305 + B, (B, P.n)), orc.uniform_i32(2306 + B, (B,)))
            for g, third in (("nand", None), ("mux", cc), ("xor", None)):
                for x, y in zip(eng.gate(g, ca, cb, third, want_var=True), m.gate(g, ca, cb, third, want_var=True)):
                    assert (x == y).all(), (B, g)
            for x, y in zip(eng.bootstrap(1 << 29, *ca, want_var=True), m.bootstrap(1 << 29, *ca, want_var=True)):
                assert (x == y).all(), B
            ea = eng.bootstrap_noks(-(1 << 29), *ca)
            for x, y in zip(ea, m.bootstrap_noks(-(1 << 29), *ca)):
                assert (x == y).all(), B
            for x, y in zip(eng.keyswitch(*ea, want_var=True), m.keyswitch(*ea, want_var=True)):
                assert (x == y).all(), B
        # circuits shard by instance
        bits, count = 3, 5
        A = (orc.uniform_i32(2311, (bits, count, P.n)), orc.uniform_i32(2312, (bits, count)))
        Bv = (orc.uniform_i32(2313, (bits, count, P.n)), orc.uniform_i32(2314, (bits, count)))
        r1, r2 = eng.circuit_add(A, Bv, want_var=True), m.circuit_add(A, Bv, want_var=True)
        assert (r1[0][0] == r2[0][0]).all() and (r1[0][1] == r2[0][1]).all()
        assert (r1[1][0] == r2[1][0]).all() and (r1[1][1] == r2[1][1]).all()
        assert (r1[2] == r2[2]).all() and (r1[3] == r2[3]).all()
        # products and handles
        d = orc.range_i32(2321, -512, 512, (9, 1024))
        s = orc.uniform_i32(2322, (9, 1024))
        assert (eng.polymul(d, s) == m.polymul(d, s)).all()
        B = 7
        ca = (orc.uniform_i32(2331, (B, P.n)), orc.uniform_i32(2332, (B,)))
        cb = (orc.uniform_i32(2333, (B, P.n)), orc.uniform_i32(2334, (B,)))
        cc = (orc.uniform_i32(2335, (B, P.n)), orc.uniform_i32(2336, (B,)))
        for got, want in zip(_chain_handles(m, ca, cb, cc), _chain_handles(eng, ca, cb, cc)):
            for g, w in zip(got, want):
                assert (g == w).all()
        # what a multi-GPU context refuses
        with pytest.raises(tfhe_amd.TfheHipError, match="single-device"):
            m._check(m.lib.tfhe_hip_export_keys_dev(m.ctx, None, None))
    finally:
        m.close()


def test_multi_context_rccl_replication_in_a_fresh_process():
    """The RCCL leg of the key replication (ncclCommInitAll + ncclBroadcast, bound with
    dlopen) is what distinct GPUs use; forced here on a one-GPU communicator so that the
    binding, the call signatures and the stream handling run on hardware."""
    code = (
        "import os, sys\n"
        f"sys.path.insert(0, {ROOT!r}); sys.path.insert(0, {os.path.join(ROOT, 'tests')!r})\n"
        "os.environ['TFHE_HIP_MULTI_FORCE_RCCL'] = '1'\n"
        "import numpy as np, _oracle as orc, tfhe_amd\n"
        "P = orc.small_params(n=5); K = orc.generate_synthetic_keys(P, 2400)\n"
        "hp = tfhe_amd.make_params(P.n, n_ks=P.n_ks)\n"
        "m = tfhe_amd.Engine([0, 0]); m.load_keys(hp, K['bk'], K['ks_a'], K['ks_b'], K['ks_var'])\n"
        "print('MSG transport', m.key_transport())\n"
        "ca = (orc.uniform_i32(1, (3, P.n)), orc.uniform_i32(2, (3,)))\n"
        "oa, ob = m.gate('nand', ca, ca)\n"
        "for i in range(3):\n"
        "    wa, wb, _ = orc.gate(P, 'nand', K, (ca[0][i], ca[1][i], 0.0), (ca[0][i], ca[1][i], 0.0))\n"
        "    assert (oa[i] == wa).all() and ob[i] == wb\n"
        "m.close(); print('DONE')\n")
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=240)
    assert r.returncode == 0 and "DONE" in r.stdout, r.stdout + r.stderr
    assert "MSG transport rccl" in r.stdout, r.stdout + r.stderr


def test_multi_context_peer_copy_replication_in_a_fresh_process():
    """The fallback leg of the key replication (blobs allocated on the other device, then
    hipMemcpyPeerAsync from the first one), which distinct GPUs take when RCCL is unavailable
    or fails: the two workers on GPU 0 are treated as sitting on GPUs of their own and RCCL
    is switched off, so that the allocation-only path and the peer copies run on hardware.
    A sample handle that outlives its context is then freed without touching it."""
    code = (
        "import os, sys\n"
        f"sys.path.insert(0, {ROOT!r}); sys.path.insert(0, {os.path.join(ROOT, 'tests')!r})\n"
        "os.environ['TFHE_HIP_MULTI_DISTINCT'] = '1'; os.environ['TFHE_HIP_MULTI_NO_RCCL'] = '1'\n"
        "import numpy as np, _oracle as orc, tfhe_amd\n"
        "P = orc.small_params(n=5); K = orc.generate_synthetic_keys(P, 2400)\n"
        "hp = tfhe_amd.make_params(P.n, n_ks=P.n_ks)\n"
        "m = tfhe_amd.Engine([0, 0]); print('MSG before', m.key_transport())\n"
        "m.load_keys(hp, K['bk'], K['ks_a'], K['ks_b'], K['ks_var'])\n"
        "print('MSG transport', m.key_transport())\n"
        "ca = (orc.uniform_i32(1, (4, P.n)), orc.uniform_i32(2, (4,)))\n"
        "oa, ob = m.gate('nand', ca, ca)\n"
        "for i in range(4):\n"
        "    wa, wb, _ = orc.gate(P, 'nand', K, (ca[0][i], ca[1][i], 0.0), (ca[0][i], ca[1][i], 0.0))\n"
        "    assert (oa[i] == wa).all() and ob[i] == wb\n"
        "h = m.samples_from(*ca)\n"
        "m.close(); h.free(); print('DONE')\n")
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=240)
    assert r.returncode == 0 and "DONE" in r.stdout, r.stdout + r.stderr
    assert "MSG before none" in r.stdout and "MSG transport peer" in r.stdout, r.stdout + r.stderr


def test_bench_two_ranks_on_one_gpu():
    """bench.py as the driver launches it for N = 2 (torch.distributed.run, one process per
    rank), rehearsed on one GPU: both ranks on GPU 0 over gloo.  Exercises rank 0's key load,
    the export -> broadcast -> import of the device blobs on the other rank, sharding with
    shard_bounds, the max-over-ranks timing and the check of a timed output against the oracle.
    (Throughput of such a run means nothing.)"""
    import json
    env = dict(os.environ, TFHE_BENCH_REHEARSE="1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
           "--master-addr", "127.0.0.1", "--master-port", "29541", os.path.join(ROOT, "bench.py"),
           "--gpus", "2", "--total-gates", "3001", "--batch", "1024", "--steps", "1", "--warmup", "0",
           "--no-latency", "--no-bit128", "--no-cpu-baseline"]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=280, env=env, cwd=ROOT)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    line = [x for x in r.stdout.splitlines() if x.startswith("{")][-1]
    res = json.loads(line)
    assert res["n_gpus"] == 2 and res["scaling"] == "strong" and res["value"] > 0
    assert res["output_check"]["matches_oracle"] is True
    assert res["roofline"]["launches"] == 2        # rank 0's shard: 1501 gates in launches of 1024


# ---- circuits::compare and circuits::eq, level-batched -----------------------------------------

def _oracle_compare(P, K, a, b, mode):
    """compare / le of /root/reference/src/circuits/circuits.rs:33-47 composed of oracle gates."""
    carry = (np.zeros(P.n, np.int32), np.int32(1 << 29), 0.0)         # constant(true)
    for x, y in zip(a, b):
        cond = orc.gate(P, "xor", K, x, y, None, mode)
        carry = orc.gate(P, "mux", K, cond, y, carry, mode)
    return carry


def _oracle_eq(P, K, a, b, mode):
    """eq / eq_bit of /root/reference/src/circuits/circuits.rs:50-70 composed of oracle gates."""
    one = (np.zeros(P.n, np.int32), np.int32(1 << 29), 0.0)
    bits = [orc.gate(P, "xor", K, orc.gate(P, "xor", K, x, y, None, mode), one, None, mode) for x, y in zip(a, b)]
    res = one
    for bit in bits:
        res = orc.gate(P, "and", K, res, bit, None, mode)
    return res


@pytest.mark.parametrize("tag,bits,count", [((2, 10), 4, 5), ((3, 7), 3, 2), ((2, 10), 1, 1), ((2, 10), 0, 3)])
def test_circuit_compare_and_eq(eng, tag, bits, count):
    import tfhe_amd
    l, bg = tag
    P = orc.small_params(n=6, l=l, bg_bit=bg)
    K = orc.generate_synthetic_keys(P, 2500)
    _load(eng, P, K)
    A = (orc.uniform_i32(2501, (bits, count, P.n)), orc.uniform_i32(2502, (bits, count)))
    Bv = (orc.uniform_i32(2503, (bits, count, P.n)), orc.uniform_i32(2504, (bits, count)))
    m = tfhe_amd.Engine([0, 0])
    try:
        m.load_keys(_hip_params(P), K["bk"], K["ks_a"], K["ks_b"], K["ks_var"])
        ca, cb, cv = eng.circuit_compare(A, Bv, want_var=True)
        for x, y in zip((ca, cb, cv), m.circuit_compare(A, Bv, want_var=True)):
            assert (x == y).all()
        if bits >= 1:
            ea, eb, ev = eng.circuit_eq(A, Bv, want_var=True)
            for x, y in zip((ea, eb, ev), m.circuit_eq(A, Bv, want_var=True)):
                assert (x == y).all()
        else:
            with pytest.raises(tfhe_amd.TfheHipError, match="at least one bit"):
                eng.circuit_eq(A, Bv)
    finally:
        m.close()
    for c in range(count):
        a = [(A[0][i, c], A[1][i, c], 0.0) for i in range(bits)]
        b = [(Bv[0][i, c], Bv[1][i, c], 0.0) for i in range(bits)]
        w = _oracle_compare(P, K, a, b, orc.MUL_EXACT)
        assert (ca[c] == w[0]).all() and cb[c] == w[1] and cv[c] == w[2], c
        if bits >= 1:
            w = _oracle_eq(P, K, a, b, orc.MUL_EXACT)
            assert (ea[c] == w[0]).all() and eb[c] == w[1] and ev[c] == w[2], c
