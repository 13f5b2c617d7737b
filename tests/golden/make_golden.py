#!/usr/bin/env python3
"""Generates tests/golden/golden_small.npz.

An independent restatement of the reference's bootstrapped path in plain
Python integers (no shared code with oracle/oracle.cpp), written from the
semantics of the reference's source:

  mul_by_monomial   /root/reference/src/polynomial.rs:281-313
  decomposition     /root/reference/src/tgsw.rs:38-68,299-328
  mul_r             /root/reference/src/numerics.rs:138-155,185-261
  external product  /root/reference/src/tgsw.rs:229-264, src/tlwe.rs:149-157
  blind rotate      /root/reference/src/bootstrapping.rs:109-146
  extract           /root/reference/src/tlwe.rs:159-180
  key switch        /root/reference/src/lwe.rs:601-648
  bootstrap         /root/reference/src/bootstrapping.rs:25-99

The reference is a Rust crate that cannot be built in this image, so these
vectors come from this second restatement, not from running the reference.
The fixture pins the C++ oracle against an independently written model on
small shapes (N=32); the reference's own known-answer tests are checked
separately in tests/test_oracle_kat.py.

Run:  python tests/golden/make_golden.py
"""
import os
import random

import numpy as np

M32 = 1 << 32


def wrap(x):
    x &= M32 - 1
    return x - M32 if x >= 1 << 31 else x


def sat(x):
    return max(-(1 << 31), min((1 << 31) - 1, x))


def rust_div(a, b):  # truncating division
    q = abs(a) // abs(b)
    return q if (a >= 0) == (b >= 0) else -q


def rust_rem(a, b):
    return a - b * rust_div(a, b)


def modulo(a, m):
    r = rust_rem(a, m)
    return r + abs(m) if r < 0 else r


def mbm(p, power):
    n = len(p)
    if power == 0:
        return list(p)
    cycle = modulo(rust_div(power, n), 2) == 1
    power = rust_rem(power, n)
    out = [0] * n
    for j in range(0, power):
        v = p[modulo(n - power + j, n)]
        out[modulo(j, n)] = wrap(-v) if not cycle else v
    for j in range(power, n):
        v = p[modulo(j - power, n)]
        out[modulo(j, n)] = wrap(-v) if cycle else v
    return out


def decode_message(phase, space):
    log2 = space.bit_length() - 1
    return wrap(phase + (1 << (32 - log2 - 1))) >> (32 - log2)


def decompose(poly, l, bg_bit):
    half = 1 << (bg_bit - 1)
    offset = ((1 << (32 - bg_bit)) * half) & (M32 - 1)   # the sum runs over one term only
    out = []
    for p in range(l):
        decal = 32 - (p + 1) * bg_bit
        out.append([((wrap(c + offset) >> decal) & ((1 << bg_bit) - 1)) - half for c in poly])
    return out


def mul_r(result, d, s):
    n = len(d)
    out = []
    for t in range(n):
        acc = sum(d[u] * s[t - u] for u in range(t + 1) if t - u < len(s))
        out.append(wrap(result[t] + sat(acc)))
    return out


def extern_mul(acc, bk_i, k, l, bg_bit):
    """acc: [k+2][N]; bk_i[p][j][u] polys.  Returns [k+2][N]."""
    n = len(acc[0])
    res = [[0] * n for _ in range(k + 2)]
    dec = [decompose(acc[j], l, bg_bit) for j in range(k + 1)]
    for p in range(l):
        for j in range(k + 1):
            for u in range(k + 1):
                res[u] = mul_r(res[u], dec[j][p], bk_i[p][j][u])
    return res


def blind_rotate(acc, bara, bk, k, l, bg_bit):
    prev = [[0] * len(a) for a in acc]
    cur = [list(a) for a in acc]
    for i, ai in enumerate(bara):
        if ai == 0:
            continue
        t = []
        for a in range(k + 2):
            s = [wrap(x + y) for x, y in zip(prev[a], cur[a])]
            r = mbm(s, ai)
            t.append([wrap(x - y) for x, y in zip(r, cur[a])])
        ext = extern_mul(t, bk[i], k, l, bg_bit)
        new = [[wrap(x + y) for x, y in zip(cur[a], ext[a])] for a in range(k + 2)]
        prev, cur = cur, new
    return cur


def extract(acc, k):
    n = len(acc[0])
    out = []
    for i in range(k):
        out.append(acc[i][0])
        out.extend(wrap(-acc[i][n - j]) for j in range(1, n))
    return out, acc[k + 1][0]


def key_switch(in_a, in_b, ks_a, ks_b, n_ks, t, bb, n):
    out = [0] * n
    b = in_b
    for i in range(n_ks):
        aibar = wrap(in_a[i] + (1 << (32 - (1 + bb * t))))
        for j in range(t):
            d = (aibar >> (32 - (j + 1) * bb)) & ((1 << bb) - 1)
            if d != 0:
                out = [wrap(x - y) for x, y in zip(out, ks_a[i][j][d])]
                b = wrap(b - ks_b[i][j][d])
    return out, b


def bootstrap_noks(a, b, mu, bk, N, k, l, bg_bit):
    barb = decode_message(b, 2 * N)
    bara = [decode_message(x, 2 * N) for x in a]
    tv = [mu] * N
    if barb != 0:
        tv = mbm(tv, modulo(-barb, 2 * N))
    acc = [[0] * N for _ in range(k + 1)] + [tv]
    acc = blind_rotate(acc, bara, bk, k, l, bg_bit)
    return extract(acc, k)


def main():
    rnd = random.Random(20240611)

    def ri32():
        return rnd.randrange(-(1 << 31), 1 << 31)

    g = {}
    # --- mul_by_monomial, n = 16, every power a bootstrap can produce (scaled): [-16, 31]
    p16 = [ri32() for _ in range(16)]
    p16[3] = -(1 << 31)  # negation of i32::MIN wraps
    g["mbm_in"] = np.array(p16, dtype=np.int32)
    g["mbm_powers"] = np.arange(-16, 32, dtype=np.int32)
    g["mbm_out"] = np.array([mbm(p16, int(w)) for w in range(-16, 32)], dtype=np.int32)

    # --- decode_message at the corners
    ph = [0, 1, -1, (1 << 31) - 1, -(1 << 31), (1 << 20) - 1, -(1 << 20), 1 << 20, 1 << 29,
          -(1 << 29)] + [ri32() for _ in range(54)]
    g["decode_in"] = np.array(ph, dtype=np.int32)
    g["decode_out"] = np.array([decode_message(x, 2048) for x in ph], dtype=np.int32)

    # --- decomposition
    vals = [0, 0x12345678, -1, (1 << 31) - 1, -(1 << 31), 1 << 21, (1 << 21) - 1] + \
           [ri32() for _ in range(57)]
    g["dec_in"] = np.array(vals, dtype=np.int32)
    g["dec_out_l2_bg10"] = np.array(decompose(vals, 2, 10), dtype=np.int32)
    g["dec_out_l3_bg7"] = np.array(decompose(vals, 3, 7), dtype=np.int32)

    # --- mul_r on N = 32 (digit x torus), including saturation straddles
    N = 32
    cases_d, cases_s, cases_r, cases_o = [], [], [], []
    for c in range(8):
        d = [rnd.randrange(-512, 512) for _ in range(N)]
        s = [ri32() for _ in range(N)]
        if c >= 4:  # sparse: exact coefficients near +-2^31 without overflow of every term
            d = [0] * N
            s = [0] * N
            d[0], d[1] = 1, 1
            s[0] = [(1 << 31) - 1, -(1 << 31), (1 << 31) - 2, -(1 << 31) + 1][c - 4]
            s[1] = [1, -1, 1, -1][c - 4]
            s[2] = [5, -5, 0, 0][c - 4]
        r = [ri32() for _ in range(N)]
        cases_d.append(d), cases_s.append(s), cases_r.append(r)
        cases_o.append(mul_r(r, d, s))
    g["mulr_d"] = np.array(cases_d, dtype=np.int32)
    g["mulr_s"] = np.array(cases_s, dtype=np.int32)
    g["mulr_r"] = np.array(cases_r, dtype=np.int32)
    g["mulr_out"] = np.array(cases_o, dtype=np.int32)

    # --- external product, blind rotate, extract, bootstrap on N = 32, n = 6
    for tag, (l, bg_bit) in {"l2": (2, 10), "l3": (3, 7)}.items():
        k, n = 1, 6
        bk = [[[[[ri32() for _ in range(N)] for _ in range(k + 1)] for _ in range(k + 1)]
               for _ in range(l)] for _ in range(n)]
        acc = [[ri32() for _ in range(N)] for _ in range(k + 2)]
        g[f"ext_{tag}_bk"] = np.array(bk, dtype=np.int32)
        g[f"ext_{tag}_acc"] = np.array(acc, dtype=np.int32)
        g[f"ext_{tag}_out"] = np.array(extern_mul(acc, bk[0], k, l, bg_bit), dtype=np.int32)
        bara = [5, -3, 0, 2 * N - 1 - N, -N, N - 1]   # positive, negative, skip, -N and N-1 corners
        g[f"br_{tag}_bara"] = np.array(bara, dtype=np.int32)
        g[f"br_{tag}_out"] = np.array(blind_rotate(acc, bara, bk, k, l, bg_bit), dtype=np.int32)
        # bootstrap without key switch: message space 2N = 64 -> shift 26
        a = [ri32() for _ in range(n)]
        b = ri32()
        mu = 1 << 29
        oa, ob = bootstrap_noks(a, b, mu, bk, N, k, l, bg_bit)
        g[f"bs_{tag}_a"] = np.array(a, dtype=np.int32)
        g[f"bs_{tag}_b"] = np.array([b], dtype=np.int32)
        g[f"bs_{tag}_out_a"] = np.array(oa, dtype=np.int32)
        g[f"bs_{tag}_out_b"] = np.array([ob], dtype=np.int32)

    # --- key switch: N*k = 32 input coefficients, n = n_ks = 6, t = 8, base 4
    n, t, bb = 6, 8, 2
    ks_a = [[[[ri32() for _ in range(n)] for _ in range(4)] for _ in range(t)] for _ in range(n)]
    ks_b = [[[ri32() for _ in range(4)] for _ in range(t)] for _ in range(n)]
    for i in range(n):
        for j in range(t):
            ks_a[i][j][0] = [0] * n
            ks_b[i][j][0] = 0
    in_a = [ri32() for _ in range(32)]
    in_a[0] = (1 << 31) - 1          # aibar wraps
    in_b = ri32()
    oa, ob = key_switch(in_a, in_b, ks_a, ks_b, n, t, bb, n)
    g["ks_a"] = np.array(ks_a, dtype=np.int32)
    g["ks_b"] = np.array(ks_b, dtype=np.int32)
    g["ks_in_a"] = np.array(in_a, dtype=np.int32)
    g["ks_in_b"] = np.array([in_b], dtype=np.int32)
    g["ks_out_a"] = np.array(oa, dtype=np.int32)
    g["ks_out_b"] = np.array([ob], dtype=np.int32)

    out = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden_small.npz")
    np.savez_compressed(out, **g)
    print("wrote", out, sum(v.nbytes for v in g.values()), "bytes")


if __name__ == "__main__":
    main()
