"""Model of PackedSample (tfhe_amd/csrc/tfhe_hip.hip): the decomposed sample of the two-wavefront
blind rotation keeps, per coefficient, only the top l * bg_bit bits of its two polynomials, packed
into one word plus a share of a remainder word.  Checked here: every digit of every level of both
polynomials, extracted from the packed form as the kernel does, equals the reference's digit
((c + 0x80000000) >> decal) & (Bg - 1)) - Bg / 2 (/root/reference/src/tgsw.rs:299-328), for both
parameter shapes and for extreme bit patterns."""
import numpy as np
import pytest

M32 = 0xFFFFFFFF


def tt_mask(l, bg_bit):
    m = 0
    for p in range(1, l):
        m |= 1 << (31 - p * bg_bit)
    return m


def sbfe(x, off, width):
    v = (x >> off) & ((1 << width) - 1)
    return v - (1 << width) if v >> (width - 1) else v


def ubfe(x, off, width):
    return (x >> off) & ((1 << width) - 1)


class Packed:
    def __init__(self, l, bg_bit):
        self.keep = l * bg_bit
        self.rem_bits = 2 * self.keep - 32
        assert self.keep < 32 and 0 < self.rem_bits <= 16
        self.per = 32 // self.rem_bits
        self.word = [0] * 16
        self.rem = [0] * ((16 + self.per - 1) // self.per)

    def put(self, r, x0, x1):
        self.word[r] = (x0 & (M32 << (32 - self.keep)) & M32) | (x1 >> self.keep)
        rest = ubfe(x1, 32 - self.keep, self.rem_bits)
        if r % self.per == 0:
            self.rem[r // self.per] = rest
        else:
            self.rem[r // self.per] |= rest << ((r % self.per) * self.rem_bits)

    def get(self, h, r):
        if h == 0:
            return self.word[r]
        return ((self.word[r] << self.keep) & M32) | (ubfe(self.rem[r // self.per], (r % self.per) * self.rem_bits,
                                                            self.rem_bits) << (32 - self.keep))


def reference_digit(c, p, bg_bit):
    decal = 32 - (p + 1) * bg_bit
    return (((c + 0x80000000) & M32) >> decal & ((1 << bg_bit) - 1)) - (1 << (bg_bit - 1))


@pytest.mark.parametrize("l,bg_bit", [(2, 10), (3, 7)])
def test_digits_from_the_packed_sample(l, bg_bit):
    rng = np.random.default_rng(l * 100 + bg_bit)
    special = [0, 1, M32, 0x80000000, 0x7FFFFFFF, 0x00000FFF, 0xFFFFF000, 0x00100000, 0x000FFFFF]
    for trial in range(200):
        t = rng.integers(0, 1 << 32, size=(2, 16), dtype=np.uint64)
        if trial < len(special):
            t[:, :] = special[trial]
        pk = Packed(l, bg_bit)
        mask = tt_mask(l, bg_bit)
        for r in range(16):
            pk.put(r, int(t[0, r]) ^ mask, int(t[1, r]) ^ mask)
        assert all(w <= M32 for w in pk.word + pk.rem)
        for p in range(l):
            decal = 32 - (p + 1) * bg_bit
            for h in range(2):
                for r in range(16):
                    assert sbfe(pk.get(h, r), decal, bg_bit) == reference_digit(int(t[h, r]), p, bg_bit), (p, h, r)


def test_register_count():
    assert len(Packed(2, 10).word) + len(Packed(2, 10).rem) == 20
    assert len(Packed(3, 7).word) + len(Packed(3, 7).rem) == 22
