"""numpy model of the register/lane layout used by the HIP blind-rotate kernel.

One 64-lane wavefront computes a zero-padded 2048-point transform as two
1024-point transforms, one per half-wave: lanes 0-31 (half E) produce the even
bins, lanes 32-63 (half O) the odd bins.  Inside a half, element n = q + 32*r of
a 1024-point sequence lives in lane q, register r, and a 1024-point DFT is the
four-step 32x32 decomposition: 32-point DFT over registers, twiddle by
w1024^(q*kb), 32x32 transpose between lanes and registers, 32-point DFT over
registers.  Input and output use the same layout.

The model mirrors the kernel's data movement step by step (arrays indexed
[lane][reg]) so that index maps can be checked on the CPU.
"""
import numpy as np

N = 1024
Q = 32
LANES = 64

_q = np.arange(Q)
_r = np.arange(Q)
T_E = np.exp(-2j * np.pi * np.outer(_q, _r) / 1024)          # [q][kb] = w1024^(q*kb)
TWIST = np.exp(-2j * np.pi * (_q[:, None] + 32 * _r[None, :]) / 2048)   # [q][r] = w2048^(q+32r)


def fft1024_half(x, inverse=False):
    """x: [32 lanes][32 regs] complex, element q+32r at [q][r].  Same layout out."""
    dft32 = (lambda a: np.fft.ifft(a, axis=1) * 32.0) if inverse else (lambda a: np.fft.fft(a, axis=1))
    te = T_E.conj() if inverse else T_E
    y = dft32(x)                # step A: per lane DFT over registers -> [q][kb]
    y = y * te                  # inter-pass twiddle
    y = y.T.copy()              # transpose: lane kb, reg n1=q
    return dft32(y)             # step B: -> [kb][ka] = X[kb + 32 ka]


def forward2048(c):
    """c: complex[1024] (zero-padded to 2048 implicitly).  Returns [64][32]:
    lane (h,q) reg r holds C[2*(q+32r)+h]."""
    x = c.reshape(Q, Q).T       # [q][r] = c[q + 32 r]
    out = np.empty((LANES, Q), dtype=np.complex128)
    out[:32] = fft1024_half(x)
    out[32:] = fft1024_half(x * TWIST)
    return out


def mirror(C):
    """Returns Ct with Ct[lane][r] = conj(C at bin (-k mod 2048))."""
    Ct = np.empty_like(C)
    for h in range(2):
        for q in range(Q):
            for r in range(Q):
                if h == 0:
                    if q == 0:
                        q2, r2 = 0, (32 - r) % 32
                    else:
                        q2, r2 = 32 - q, 31 - r
                else:
                    q2, r2 = 31 - q, 31 - r
                Ct[h * 32 + q, r] = np.conj(C[h * 32 + q2, r2])
    return Ct


def inverse2048_first_half(Z):
    """Z: [64][32] in the forward's layout.  Returns complex x[t], t<1024, as
    [32][32] (lane q reg r = t = q+32r), unnormalised."""
    E = fft1024_half(Z[:32], inverse=True)
    O = fft1024_half(Z[32:], inverse=True) * TWIST.conj()
    return E + O


def bk_spectrum(s0, s1):
    """Key polynomial pair (u=0, u=1) -> [64][32] spectrum with all scaling folded in."""
    return forward2048(s0.astype(np.float64) + 1j * s1.astype(np.float64)) / 4096.0


def to_poly(x):
    """[32][32] layout -> flat[1024]."""
    return x.T.reshape(-1)


def mul_pair(d0, d1, B0, B1):
    """Products of digit polys d0, d1 with key spectra B0, B1.  Returns four
    real arrays: conv(d0,s00), conv(d0,s01), conv(d1,s10), conv(d1,s11), first N."""
    C = forward2048(d0.astype(np.float64) + 1j * d1.astype(np.float64))
    Ct = mirror(C)
    D0 = C + Ct                 # the 1/2 is folded into the key spectrum
    D1 = -1j * (C - Ct)
    x0 = to_poly(inverse2048_first_half(D0 * B0))
    x1 = to_poly(inverse2048_first_half(D1 * B1))
    return x0.real, x0.imag, x1.real, x1.imag
