"""numpy model of the 64-lane transforms (tfhe_amd/csrc/wave_fft64.hpp): one WAVEFRONT owns one
1024-point transform, element n = lane + 64 r in lane `lane`, register r (16 registers).

1024 = 16 (registers) x 4 (lane bits 5, 4) x 16 (lane bits 3..0).  With n = 64 r + 16 a + ql and
k = kr + 16 c + 64 kq,  w1024^(nk) = w16^(r kr) . w1024^((16a+ql) kr) . w4^(a c) . w64^(ql c) . w16^(ql kq):

  1. 16-point DFT over the registers                     r  -> kr
  2. twiddle T1[kr][lane] = w1024^(lane kr)              (table; the lane part of the twist folds in)
  3. lane bits (5, 4) <-> register bits (3, 2)           v_permlane32_swap, v_permlane16_swap
  4. 4-point DFT over the registers that now hold a      a  -> c
  5. twiddle T2 = w64^(ql c)                             (three factors per lane)
  6. 16 x 16 transpose inside every row of 16 lanes      (through LDS): registers <-> ql
  7. 16-point DFT over the registers                     ql -> kq

The inverse runs the mirror image with conjugated factors.  Arrays are [reg][lane]; every step
moves data exactly as the kernel does, so the spectrum layout the key transform writes and the
inverse reads is the model's too.
"""
import numpy as np

LANES, REGS, N = 64, 16, 1024
_lane = np.arange(LANES)
_reg = np.arange(REGS)
T1 = np.exp(-2j * np.pi * np.outer(_reg, _lane) / 1024)                  # [kr][lane]
TWIST_LANE = np.exp(-2j * np.pi * _lane / 4096)                          # w4096^lane
TWIST_REG = np.exp(-2j * np.pi * _reg / 64)                              # w4096^(64 r) = w64^r
T1_TWISTED = T1 * TWIST_LANE[None, :]                                    # what the kernel's table holds
T2 = np.exp(-2j * np.pi * np.outer(np.arange(4), _lane & 15) / 64)       # [c][lane] = w64^(ql c)


def permlane32_swap(a, b):
    """v_permlane32_swap_b32 a, b: lanes 32..63 of a <-> lanes 0..31 of b."""
    a, b = a.copy(), b.copy()
    t = a[32:].copy()
    a[32:] = b[:32]
    b[:32] = t
    return a, b


def permlane16_swap(a, b):
    """v_permlane16_swap_b32 a, b: odd rows (of 16 lanes) of a <-> even rows of b."""
    a, b = a.copy(), b.copy()
    for row in (0, 2):
        t = a[16 * (row + 1):16 * (row + 2)].copy()
        a[16 * (row + 1):16 * (row + 2)] = b[16 * row:16 * (row + 1)]
        b[16 * row:16 * (row + 1)] = t
    return a, b


def lanes_to_regs(v):
    """Step 3: register bit 3 <-> lane bit 5, register bit 2 <-> lane bit 4."""
    v = v.copy()
    for r in range(16):
        if not r & 8:
            v[r], v[r | 8] = permlane32_swap(v[r], v[r | 8])
    for r in range(16):
        if not r & 4:
            v[r], v[r | 4] = permlane16_swap(v[r], v[r | 4])
    return v


def regs_to_lanes(v):
    """The inverse of lanes_to_regs (each swap is its own inverse; the two kinds commute)."""
    return lanes_to_regs(v)


def dft4_over_high_bits(v, inverse):
    """4-point DFT over register bits (3, 2) = a -> c, for every value of bits (1, 0)."""
    x = v.reshape(4, 4, LANES)
    w = np.exp((2j if inverse else -2j) * np.pi * np.outer(np.arange(4), np.arange(4)) / 4)
    return np.einsum("ca,ajl->cjl", w, x).reshape(16, LANES)


def row_transpose(v):
    """Step 6: inside every row of 16 lanes, register index <-> lane index (low 4 bits)."""
    x = v.reshape(16, 4, 16)                     # [reg][row][ql]
    return x.transpose(2, 1, 0).reshape(16, LANES).copy()


def dft16(v, inverse):
    return np.fft.ifft(v, axis=0) * 16.0 if inverse else np.fft.fft(v, axis=0)


def to_regs(x):
    """flat[1024] -> [reg][lane] with x[lane + 64 reg]."""
    return np.asarray(x).reshape(REGS, LANES).copy()


def forward(v, table=T1):
    v = dft16(v, False) * table
    v = dft4_over_high_bits(lanes_to_regs(v), False)
    v = v * np.repeat(T2, 4, axis=0)             # register (c, j) takes w64^(ql c)
    return dft16(row_transpose(v), False)


def t1_after_exchange(table=T1):
    """T1 as the kernel reads it when the twiddle is applied AFTER step 3: the element now in lane
    (a | ql), register y came from lane (y3 y2 | ql), register (a1 a0 y1 y0) = kr."""
    y = _reg[:, None]
    lane = _lane[None, :]
    kr = ((lane >> 4) << 2) | (y & 3)
    src_lane = ((y >> 2) << 4) | (lane & 15)
    return table[kr, src_lane]


def forward_t1_fused(v, table=T1):
    """forward() with T1 inside the first half of the 4-point pass, as the kernel computes it:
    a' = wa a + wb b, b' = wa a - wb b on the register pairs (y, y | 8), then the second half."""
    v = lanes_to_regs(dft16(v, False)) * t1_after_exchange(table)
    s, d = v[:8] + v[8:], v[:8] - v[8:]          # rows 0..3: (x0 +- x2)[j], rows 4..7: (x1 +- x3)[j]
    out = np.empty_like(v)
    out[0:4], out[8:12] = s[:4] + s[4:], s[:4] - s[4:]
    out[4:8], out[12:16] = d[:4] - 1j * d[4:], d[:4] + 1j * d[4:]
    out = out * np.repeat(T2, 4, axis=0)
    return dft16(row_transpose(out), False)


def inverse(V, table=T1):
    v = row_transpose(dft16(V, True))
    v = v * np.repeat(T2, 4, axis=0).conj()
    v = regs_to_lanes(dft4_over_high_bits(v, True))
    return dft16(v * table.conj(), True)


def spectrum_index():
    """k held by [reg][lane] after forward(): lane = (kr3 kr2 | c1 c0 kr1 kr0), reg = kq."""
    lane = _lane[None, :]
    kr = ((lane >> 4) << 2) | (lane & 3)
    c = (lane >> 2) & 3
    return kr + 16 * c + 64 * _reg[:, None]


def forward_twisted(a):
    """Real polynomial a[1024] -> its values at w4096^(4k+1) in the spectrum layout."""
    return forward(to_regs(np.asarray(a, dtype=np.float64)) * TWIST_REG[:, None], T1_TWISTED)


def inverse_twisted(C):
    """Spectrum of a real polynomial c of degree < 2048 -> 1024 * (c[0:1024], c[1024:2048])."""
    y = inverse(C, T1_TWISTED) * TWIST_REG.conj()[:, None]
    return y.real.reshape(-1), (-y.imag).reshape(-1)
