"""The numpy model of the one-wavefront-per-transform layout (tests/wave64_model.py), whose
data movement wave_fft64.hpp follows step by step: index maps, the permlane exchanges and the
product of real polynomials through the twisted transforms."""
import numpy as np

import wave64_model as m


def test_forward_is_the_dft_in_the_spectrum_layout():
    rng = np.random.default_rng(1)
    x = rng.standard_normal(1024) + 1j * rng.standard_normal(1024)
    F = m.forward(m.to_regs(x))
    assert np.abs(F - np.fft.fft(x)[m.spectrum_index()]).max() < 1e-11
    assert np.abs(m.inverse(F).reshape(-1) / 1024 - x).max() < 1e-13


def test_lane_register_exchange_is_an_involution_that_swaps_index_bits():
    v = np.arange(16 * 64, dtype=np.float64).reshape(16, 64)          # value = 64 reg + lane
    w = m.lanes_to_regs(v)
    assert (m.regs_to_lanes(w) == v).all()
    reg, lane = np.divmod(w.astype(int), 64)
    r, l = np.arange(16)[:, None], np.arange(64)[None, :]
    # register bits (3, 2) now hold the former lane bits (5, 4) and the other way round
    assert ((reg >> 2) == (l >> 4)).all() and ((lane >> 4) == (r >> 2)).all()
    assert ((reg & 3) == (r & 3)).all() and ((lane & 15) == (l & 15)).all()


def test_twisted_product_is_the_linear_convolution():
    rng = np.random.default_rng(2)
    a = rng.integers(-512, 512, 1024)
    b = rng.integers(-2 ** 31, 2 ** 31, 1024)
    lo, hi = m.inverse_twisted(m.forward_twisted(a) * m.forward_twisted(b) / 1024.0)
    c = np.convolve(a.astype(object), b.astype(object))
    c = np.array(list(c) + [0], dtype=np.float64)
    assert np.abs(lo - c[:1024]).max() < 0.25 and np.abs(hi - c[1024:]).max() < 0.25


def test_twiddle_after_the_exchange_equals_twiddle_before_it():
    """The kernel applies T1 after the lane/register exchange, inside the first half of the
    4-point pass, reading the table at the permuted index: same transform."""
    rng = np.random.default_rng(3)
    x = rng.standard_normal(1024) + 1j * rng.standard_normal(1024)
    v = m.to_regs(x)
    assert np.abs(m.forward_t1_fused(v) - m.forward(v)).max() < 1e-10
    assert np.abs(m.forward_t1_fused(v, m.T1_TWISTED) - m.forward(v, m.T1_TWISTED)).max() < 1e-10
