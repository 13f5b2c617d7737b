"""The numpy model of the kernels' transforms (tests/wave_model.py) against exact integer
convolutions: pins the twisted-transform identity, the lane/register layout and the key-block
layout the kernels rely on.  CPU only."""
import numpy as np

import wave_model as wm


def _exact(a, b):
    return np.convolve(np.asarray(a, dtype=object), np.asarray(b, dtype=object))


def test_twisted_product_is_the_linear_convolution():
    rng = np.random.default_rng(5)
    d = rng.integers(-512, 512, 1024)
    s = rng.integers(-2**31, 2**31, 1024)
    lo, hi = wm.inverse_twisted(wm.forward_twisted(d) * wm.forward_twisted(s) / 1024.0)
    want = _exact(d, s)
    assert np.max(np.abs(lo - np.rint(lo))) < 0.05 and np.max(np.abs(hi - np.rint(hi))) < 0.05
    assert all(int(np.rint(lo[i])) == want[i] for i in range(1024))
    assert all(int(np.rint(hi[i])) == want[1024 + i] for i in range(1023)) and abs(hi[1023]) < 0.05


def test_four_step_layout_matches_a_plain_fft():
    rng = np.random.default_rng(6)
    x = rng.standard_normal(1024) + 1j * rng.standard_normal(1024)
    X = wm.fft1024_half(wm.to_lanes(x))
    assert np.allclose(wm.to_poly(X), np.fft.fft(x), atol=1e-9)           # [lane kb][reg ka] = X[kb + 32 ka]
    assert np.allclose(wm.to_poly(wm.fft1024_half(X, inverse=True)), x * 1024, atol=1e-8)


def test_level_product_feeds_the_right_result_polynomials():
    rng = np.random.default_rng(7)
    d = [rng.integers(-64, 64, 1024) for _ in range(2)]
    K = [[rng.integers(-2**31, 2**31, 1024) for _ in range(2)] for _ in range(2)]
    own, cross = wm.level_product(d, wm.key_blocks(K))
    for h in range(2):
        w_own = _exact(d[h], K[h][h])[:1024]
        w_cross = _exact(d[h], K[h][1 - h])[:1024]
        assert all(int(np.rint(own[h][i])) == w_own[i] for i in range(1024))
        assert all(int(np.rint(cross[h][i])) == w_cross[i] for i in range(1024))


def test_scalar_form_of_the_twist_equals_the_table_form():
    rng = np.random.default_rng(8)
    d = rng.integers(-512, 512, 1024)
    s = rng.integers(-2**31, 2**31, 1024)
    D, S = wm.forward_twisted(d), wm.forward_twisted(s)
    assert np.allclose(wm.forward_twisted_scalar(d), D, rtol=0, atol=1e-6)
    assert np.allclose(wm.forward_twisted_scalar(s), S, rtol=0, atol=1e-6 * 2.0**21)
    lo, hi = wm.inverse_twisted_scalar(wm.forward_twisted_scalar(d) * wm.forward_twisted_scalar(s) / 1024.0)
    want = _exact(d, s)
    assert np.max(np.abs(lo - np.rint(lo))) < 0.05 and np.max(np.abs(hi - np.rint(hi))) < 0.05
    assert all(int(np.rint(lo[i])) == want[i] for i in range(1024))
    assert all(int(np.rint(hi[i])) == want[1024 + i] for i in range(1023))
