"""numpy model of the register/lane layout and of the transforms used by the HIP kernels.

A half-wave (32 lanes) owns one 1024-point complex transform: element n = q + 32*r of a
sequence lives in lane q, register r, and the 1024-point DFT is the four-step 32x32
decomposition (32-point DFT over registers, twiddle by w1024^(q*kb), 32x32 transpose between
lanes and registers, 32-point DFT over registers).

Products of real polynomials use the TWISTED transform: A(k) = sum_n a[n] w4096^n w1024^(nk),
the value of a(x) at z_k = w4096^(4k+1).  These 1024 points and their conjugates are the roots
of z^2048 = -i, so they determine a real polynomial of degree < 2048: with C = A*B the
unnormalised inverse transform gives y[n] = 1024 (c[n] - i c[n+1024]) w4096^n, i.e. the low
half of the linear convolution is Re(y conj(w4096^n))/1024 and the high half -Im(...)/1024.

The model mirrors the kernels' data movement step by step (arrays indexed [lane][reg]) so that
index maps and the key layout can be checked on the CPU.
"""
import numpy as np

N = 1024
Q = 32

_q = np.arange(Q)
_r = np.arange(Q)
T_E = np.exp(-2j * np.pi * np.outer(_q, _r) / 1024)                         # [q][kb] = w1024^(q*kb)
T_Q = np.exp(-2j * np.pi * (_q[:, None] + 32 * _r[None, :]) / 4096)         # [q][r] = w4096^(q+32r)
# Scalar form of the twist: w4096^(q+32r) = w4096^q w128^r.  The register part is 32 factors
# common to all lanes; the lane part is folded into the inter-pass twiddles, one table per
# direction (the lane index of the input is q before the forward's transpose and n1 after the
# inverse's).
W128 = np.exp(-2j * np.pi * _r / 128)                                       # [r]
T_F = T_E * np.exp(-2j * np.pi * _q / 4096)[:, None]                        # [q][kb] = w1024^(q kb) w4096^q
T_I = T_E * np.exp(-2j * np.pi * _r / 4096)[None, :]                        # [kb][n1] = w1024^(n1 kb) w4096^n1


def fft1024_half(x, inverse=False, table=None):
    """x: [32 lanes][32 regs] complex, element q+32r at [q][r].  Same layout out.
    table: inter-pass twiddles, [lane][reg] at the point of use (default w1024^(q kb))."""
    dft32 = (lambda a: np.fft.ifft(a, axis=1) * 32.0) if inverse else (lambda a: np.fft.fft(a, axis=1))
    te = (T_E if table is None else table)
    te = te.conj() if inverse else te
    y = dft32(x)                # step A: per lane DFT over registers -> [q][kb]
    y = y * te                  # inter-pass twiddle
    y = y.T.copy()              # transpose: lane kb, reg n1=q
    return dft32(y)             # step B: -> [kb][ka] = X[kb + 32 ka]


def to_lanes(a):
    """flat[1024] -> [q][r] with a[q + 32 r]."""
    return np.asarray(a).reshape(Q, Q).T


def to_poly(x):
    """[32][32] layout -> flat[1024]."""
    return x.T.reshape(-1)


def forward_twisted(a):
    """Real polynomial a[1024] -> its 1024 values, [lane][reg] = A(lane + 32 reg)."""
    return fft1024_half(to_lanes(np.asarray(a, dtype=np.float64)) * T_Q)


def inverse_twisted(C):
    """Spectrum of a real polynomial c of degree < 2048 -> (c[0:1024], c[1024:2048]) * 1024."""
    y = fft1024_half(C, inverse=True) * T_Q.conj()
    return to_poly(y.real), to_poly(-y.imag)


def forward_twisted_scalar(a):
    """forward_twisted with the twist split into w128^r (per register) and the table T_F."""
    return fft1024_half(to_lanes(np.asarray(a, dtype=np.float64)) * W128[None, :], table=T_F)


def inverse_twisted_scalar(C):
    """inverse_twisted with the lane part of the untwist in the table T_I, w128^r afterwards."""
    y = fft1024_half(C, inverse=True, table=T_I) * W128.conj()[None, :]
    return to_poly(y.real), to_poly(-y.imag)


def key_blocks(K):
    """K[j][u]: the key polynomial digit polynomial j multiplies for result polynomial u.
    Returns the two blocks of a level as [64 lanes][32 regs]: block 0 holds K_{h,h} in half-wave
    h, block 1 K_{h,1-h}; the 1/1024 of the inverse transform is folded in."""
    blk = np.empty((2, 64, Q), dtype=np.complex128)
    for h in range(2):
        blk[0, 32 * h:32 * h + 32] = forward_twisted(K[h][h]) / 1024.0
        blk[1, 32 * h:32 * h + 32] = forward_twisted(K[h][1 - h]) / 1024.0
    return blk


def level_product(d, blk):
    """d[h]: digit polynomial h.  Returns what each half-wave produces: own[h] = first N
    coefficients of d_h * K_{h,h} (result polynomial h) and cross[h] = those of d_h * K_{h,1-h}
    (result polynomial 1 - h, handed to the other half-wave), before rounding."""
    own, cross = [], []
    for h in range(2):
        D = forward_twisted(d[h])
        own.append(inverse_twisted(D * blk[0, 32 * h:32 * h + 32])[0])
        cross.append(inverse_twisted(D * blk[1, 32 * h:32 * h + 32])[0])
    return own, cross
