"""Pins the CPU oracle: the reference's own known-answer tests, hand-derived
values from the reference source, and the independently written Python model
in tests/golden/.  No GPU."""
import os

import numpy as np
import pytest

import _oracle as orc

GOLD = np.load(os.path.join(os.path.dirname(__file__), "golden", "golden_small.npz"))


# ---- the reference's own KATs ------------------------------------------------

@pytest.mark.parametrize("mode", [orc.MUL_EXACT, orc.MUL_FFT, orc.MUL_FFT_PLAN_PER_CALL])
def test_poly_multiplier_kat(mode):
    # /root/reference/src/numerics.rs:499-506
    out, _ = orc.poly_multiplier([10, 20, 30], [1, 2, 3], mode)
    assert out.tolist() == [10, 40, 100, 120, 90]


@pytest.mark.parametrize("mode", [orc.MUL_EXACT, orc.MUL_FFT])
def test_poly_multiplier_degree(mode):
    # /root/reference/src/numerics.rs:508-515
    out, _ = orc.poly_multiplier(np.zeros(1024, np.int32), np.zeros(1024, np.int32), mode)
    assert len(out) == 2047 and not out.any()


def test_mul_by_monomial_kat():
    # /root/reference/src/polynomial.rs:333-364
    p = np.arange(10, dtype=np.int32)
    neg = lambda r: [-x for x in r]
    assert orc.mul_by_monomial(p, 4).tolist() == neg(range(6, 10)) + list(range(0, 6))
    assert orc.mul_by_monomial(p, 10).tolist() == neg(range(0, 10))
    assert orc.mul_by_monomial(p, 14).tolist() == list(range(6, 10)) + neg(range(0, 6))
    assert orc.mul_by_monomial(p, 20).tolist() == list(range(0, 10))
    assert orc.mul_by_monomial(p, 24).tolist() == neg(range(6, 10)) + list(range(0, 6))


def test_sub_different_sizes_kat():
    # /root/reference/src/polynomial.rs:319-331
    assert orc.poly_sub([1, 2, 3, 4], [1, 1, 2, 3, 4]).tolist() == [-1, 0, 0, 0, 0]
    assert orc.poly_sub([1, 1, 2, 3, 4], [1, 2, 3, 4]).tolist() == [1, 0, 0, 0, 0]
    assert orc.poly_sub([1, 2, 3, 4], [1, 2, 3, 4]).tolist() == [0, 0, 0, 0]


def test_float_to_torus_saturates():
    # /root/reference/src/numerics.rs:56-58; the wrap-around expectations of the
    # ignored test at :407-415 do not hold for a saturating cast.
    assert orc.lib.orc_f64_to_torus32(0.0) == 0
    assert orc.lib.orc_f64_to_torus32(0.25) == 1 << 30
    assert orc.lib.orc_f64_to_torus32(0.5) == (1 << 31) - 1
    assert orc.lib.orc_f64_to_torus32(-0.5) == -(1 << 31)
    assert orc.lib.orc_f64_to_torus32(-0.25) == -(1 << 30)


def test_encode_decode():
    # /root/reference/src/numerics.rs:117-130
    assert orc.encode_message(1, 8) == 1 << 29
    assert orc.encode_message(-1, 8) == -(1 << 29)
    assert orc.encode_message(1, 4) == 1 << 30
    assert orc.decode_message(-(1 << 31), 2048) == -1024
    assert orc.decode_message((1 << 31) - 1, 2048) == -1024   # wrapping add
    assert orc.decode_message((1 << 31) - (1 << 20) - 1, 2048) == 1023
    got = [orc.decode_message(int(x), 2048) for x in GOLD["decode_in"]]
    assert got == GOLD["decode_out"].tolist()


# ---- values derived by hand from the reference source -------------------------

def test_decomposition_hand_values():
    # /root/reference/src/tgsw.rs:50-52 gives offset 0x80000000 for every base.
    assert orc.decompose([0], 2, 10)[:, 0].tolist() == [0, -512]
    assert orc.decompose([0], 3, 7)[:, 0].tolist() == [0, -64, -64]
    assert orc.decompose([0x12345678], 2, 10)[:, 0].tolist() == [72, 325]


def test_key_switch_digit_recomposition():
    # the identity asserted at /root/reference/src/lwe.rs:852-859 (t=14, base 4)
    rng = np.random.default_rng(5)
    t, bb = 14, 2
    prec_offset = 1 << (32 - (1 + bb * t))
    prec_mask = (-(1 << (32 - bb * t))) & 0xFFFFFFFF
    for a in rng.integers(-2**31, 2**31, size=200):
        aibar = (int(a) + prec_offset) & 0xFFFFFFFF & prec_mask
        dec = 0
        for j in range(t):
            aij = (aibar >> (32 - (j + 1) * bb)) & 3
            dec += aij << (32 - (j + 1) * bb)
        assert dec == aibar


# ---- independent Python model (tests/golden/make_golden.py) --------------------

def test_golden_mul_by_monomial():
    p = GOLD["mbm_in"]
    for w, want in zip(GOLD["mbm_powers"], GOLD["mbm_out"]):
        assert orc.mul_by_monomial(p, int(w)).tolist() == want.tolist(), int(w)


def test_golden_decomposition():
    assert (orc.decompose(GOLD["dec_in"], 2, 10) == GOLD["dec_out_l2_bg10"]).all()
    assert (orc.decompose(GOLD["dec_in"], 3, 7) == GOLD["dec_out_l3_bg7"]).all()


@pytest.mark.parametrize("mode", [orc.MUL_EXACT, orc.MUL_FFT])
def test_golden_mul_r(mode):
    for d, s, r, want in zip(GOLD["mulr_d"], GOLD["mulr_s"], GOLD["mulr_r"], GOLD["mulr_out"]):
        got, frac = orc.mul_r(r, d, s, mode)
        assert got.tolist() == want.tolist()
        assert frac < 0.25


def _small(tag):
    l, bg = {"l2": (2, 10), "l3": (3, 7)}[tag]
    P = orc.small_params(n=6, l=l, bg_bit=bg)
    P.N = 32
    return P


@pytest.mark.parametrize("tag", ["l2", "l3"])
@pytest.mark.parametrize("mode", [orc.MUL_EXACT, orc.MUL_FFT])
def test_golden_extern_mul_blind_rotate_bootstrap(tag, mode):
    P = _small(tag)
    bk = np.ascontiguousarray(GOLD[f"ext_{tag}_bk"])
    acc = GOLD[f"ext_{tag}_acc"]
    got, _, frac = orc.extern_mul(P, acc, bk[0], None, mode)
    want = GOLD[f"ext_{tag}_out"]
    assert (got[:2] == want[:2]).all() and not got[2].any() and not want[2].any()
    assert frac < 0.25
    got, _, _ = orc.blind_rotate(P, acc, GOLD[f"br_{tag}_bara"], bk, None, mode)
    assert (got == GOLD[f"br_{tag}_out"]).all()
    oa, ob, _ = orc.bootstrap_noks(P, bk, 1 << 29, GOLD[f"bs_{tag}_a"], int(GOLD[f"bs_{tag}_b"][0]),
                                   None, mode)
    assert (oa == GOLD[f"bs_{tag}_out_a"]).all() and ob == int(GOLD[f"bs_{tag}_out_b"][0])


def test_golden_key_switch():
    P = orc.small_params(n=6)
    P.N = 32
    oa, ob, _ = orc.key_switch(P, np.ascontiguousarray(GOLD["ks_a"]),
                               np.ascontiguousarray(GOLD["ks_b"]), GOLD["ks_in_a"],
                               int(GOLD["ks_in_b"][0]))
    assert (oa == GOLD["ks_out_a"]).all() and ob == int(GOLD["ks_out_b"][0])


# ---- exact vs FFT at full size -------------------------------------------------

@pytest.mark.parametrize("half_bg", [512, 64])
def test_exact_matches_fft_n1024(half_bg):
    d = orc.range_i32(11, -half_bg, half_bg, (1024,))
    s = orc.uniform_i32(12, (1024,))
    e, _ = orc.poly_multiplier(d, s, orc.MUL_EXACT)
    f, frac = orc.poly_multiplier(d, s, orc.MUL_FFT)
    assert (e == f).all()
    assert frac < 0.25
    # the 2047-long key polynomial of tlwe.rs:112-134 only contributes its first N entries
    s_long = np.concatenate([s, orc.uniform_i32(13, (1023,))])
    r0 = orc.uniform_i32(14, (1024,))
    a, _ = orc.mul_r(r0, d, s, orc.MUL_EXACT)
    b, _ = orc.mul_r(r0, d, s_long, orc.MUL_FFT)
    assert (a == b).all()


# ---- gates ------------------------------------------------------------------

TRUTH = {  # /root/reference/tests/bootstrapping_tests.rs:136-198
    "nand": lambda a, b: not (a and b), "or": lambda a, b: a or b, "and": lambda a, b: a and b,
    "xor": lambda a, b: a != b, "xnor": lambda a, b: a == b, "nor": lambda a, b: not (a or b),
    "andny": lambda a, b: (not a) and b, "andyn": lambda a, b: a and (not b),
    "orny": lambda a, b: (not a) or b, "oryn": lambda a, b: a or (not b),
}


@pytest.fixture(scope="module")
def keys80():
    P = orc.params(80)
    return P, orc.generate_keys(P, 0xC0FFEE)


def test_gate_truth_tables_without_bootstrap(keys80):
    """What the reference's integration tests check: CI runs them with the
    `bootstrapping` feature off, i.e. linear prologue + decrypt."""
    P, K = keys80
    seed = 100
    for name, fn in TRUTH.items():
        for va in (False, True):
            for vb in (False, True):
                seed += 2
                ca = orc.encrypt(P, K["lwe_key"], va, seed)
                cb = orc.encrypt(P, K["lwe_key"], vb, seed + 1)
                xa, xb, xv, _ = orc.gate_prologue(name, ca, cb)
                assert orc.decrypt(xa, xb, K["lwe_key"]) == bool(fn(va, vb)), (name, va, vb)


def test_encrypt_decrypt_roundtrip(keys80):
    # /root/reference/tests/bootstrapping_tests.rs:13-25
    P, K = keys80
    for m in (False, True):
        a, b, var = orc.encrypt(P, K["lwe_key"], m, 7 + m)
        assert orc.decrypt(a, b, K["lwe_key"]) == m
        assert var == P.ks_stdev ** 2


def test_reference_key_structure(keys80):
    """Layout produced by LweBootstrappingKey::create on the reference's code
    (/root/reference/src/tgsw.rs:118-167, src/tlwe.rs:112-134)."""
    P, K = keys80
    bk = K["bk"]
    assert not bk[:, 1:].any()                                   # rows >= 1 stay zero
    s = K["lwe_key"]
    assert set(np.unique(s)) <= {0, 1}
    # second polynomial of a row-0 sample: [s*l, 0 .. 0, g0]
    b = bk[:, 0, :, 1, :]
    assert (b[:, :, 0] == (s * P.l)[:, None]).all()
    assert not b[:, :, 1:P.N - 1].any()
    assert not K["ks_a"][:, :, 0].any() and not K["ks_b"][:, :, 0].any()


@pytest.mark.parametrize("mode", [orc.MUL_FFT])
def test_bootstrap_closed_form_on_reference_keys(keys80, mode):
    """With reference-generated keys the external product is identically zero,
    so tfhe_bootstrap returns a noiseless trivial sample +-mu whose sign is
    the sign of (b + 2^20) (bootstrapping.rs:50-99, polynomial.rs:281-313)."""
    P, K = keys80
    mu = orc.encode_message(1, 8)
    for seed, m in ((31, True), (32, False)):
        a, b, _ = orc.encrypt(P, K["lwe_key"], m, seed)
        oa, ob, var, frac = orc.bootstrap(P, K["bk"], K["ks_a"], K["ks_b"], mu, a, b,
                                          K["bk_var"], K["ks_var"], mode)
        assert not oa.any()
        barb = orc.decode_message(b, 2 * P.N)
        # coefficient 0 of X^{(-barb) mod 2N} * [mu]*N is +mu iff barb >= 0
        assert ob == (mu if barb >= 0 else -mu)
        assert var == 0.0
        assert frac == 0.0
