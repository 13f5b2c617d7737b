"""Key generation on the GPU (tfhe_hip_generate_keys) against the reference's procedure
(/root/reference/src/lwe.rs:294-309,474-493,530-558, src/tgsw.rs:81-176, src/tlwe.rs:43-64,
112-134).  The reference draws every random value from an unseedable thread_rng, so its
keys cannot be reproduced bit for bit by anyone; parity here is
  * the keystream: ChaCha20 (RFC 8439 section 2.3.2 known answer, then the device against a
    Python restatement),
  * every uniform part of the keys recomputed on the CPU from the documented streams, bit for bit,
  * the structure the reference's code produces (the oracle's key-structure assertions),
  * the Gaussian parts by what they must satisfy (recentred mean, standard deviation, the
    phase of every key-switch sample under the secret key),
  * gates on the generated keys: equal to the oracle on the exported keys, and the closed form
    the reference's keys imply (noiseless +-mu, decryptable)."""
import numpy as np
import pytest

import _oracle as orc

pytestmark = pytest.mark.gpu

SEED = bytes(range(32))


def _rotl(v, c):
    return ((v << c) & 0xFFFFFFFF) | (v >> (32 - c))


def chacha20_block(key_words, counter, n1, n2, n3):
    """RFC 8439 section 2.3: state = constants, key, counter, three nonce words."""
    st = [0x61707865, 0x3320646E, 0x79622D32, 0x6B206574] + list(key_words) + [counter, n1, n2, n3]
    x = list(st)

    def qr(a, b, c, d):
        x[a] = (x[a] + x[b]) & 0xFFFFFFFF; x[d] = _rotl(x[d] ^ x[a], 16)
        x[c] = (x[c] + x[d]) & 0xFFFFFFFF; x[b] = _rotl(x[b] ^ x[c], 12)
        x[a] = (x[a] + x[b]) & 0xFFFFFFFF; x[d] = _rotl(x[d] ^ x[a], 8)
        x[c] = (x[c] + x[d]) & 0xFFFFFFFF; x[b] = _rotl(x[b] ^ x[c], 7)

    for _ in range(10):
        qr(0, 4, 8, 12); qr(1, 5, 9, 13); qr(2, 6, 10, 14); qr(3, 7, 11, 15)
        qr(0, 5, 10, 15); qr(1, 6, 11, 12); qr(2, 7, 8, 13); qr(3, 4, 9, 14)
    return [(a + b) & 0xFFFFFFFF for a, b in zip(x, st)]


def stream_words(seed, purpose, stream, nwords):
    kw = np.frombuffer(seed, dtype="<u4").tolist()
    out = []
    for blk in range((nwords + 15) // 16):
        out += chacha20_block(kw, blk, purpose, stream & 0xFFFFFFFF, stream >> 32)
    return np.array(out[:nwords], dtype=np.uint32)


@pytest.fixture(scope="module")
def eng():
    import tfhe_amd
    e = tfhe_amd.Engine(0)
    yield e
    e.close()


def test_chacha20_known_answer_and_device_keystream(eng):
    # RFC 8439 section 2.3.2: key 00..1f, counter 1, nonce 00:00:00:09 00:00:00:4a 00:00:00:00
    want = [0xE4E7F110, 0x15593BD1, 0x1FDD0F50, 0xC47120A3, 0xC7F4D1C7, 0x0368C033, 0x9AAA2204, 0x4E6CD4C3,
            0x466482D2, 0x09AA9F07, 0x05D7C214, 0xA2028BD9, 0xD19C12B5, 0xB94E16DE, 0xE883D0CB, 0x4E3C50A2]
    kw = np.frombuffer(SEED, dtype="<u4").tolist()
    assert chacha20_block(kw, 1, 0x09000000, 0x4A000000, 0) == want
    got = eng.keygen_stream(SEED, 0x09000000, 0x4A000000, 1, 1)
    assert got[0].tolist() == want
    # a longer run of another stream, device against the Python restatement
    seed = bytes((7 * i + 3) % 256 for i in range(32))
    got = eng.keygen_stream(seed, 5, (3 << 32) | 77, 0, 40)
    kw = np.frombuffer(seed, dtype="<u4").tolist()
    for blk in (0, 1, 39):
        assert got[blk].tolist() == chacha20_block(kw, blk, 5, 77, 3)


@pytest.fixture(scope="module")
def generated80(eng):
    import tfhe_amd
    P = orc.params(80)
    hp = tfhe_amd.make_params(P.n, P.N, P.k, P.l, P.bg_bit, P.ks_t, P.ks_base_bit, P.n_ks)
    K = eng.generate_keys(hp, P.ks_stdev, P.bk_stdev, SEED)
    return P, hp, K


def test_generated_keys_structure_and_uniform_parts(generated80):
    P, hp, K = generated80
    n, N, l = P.n, P.N, P.l
    s = K["lwe_key"]
    assert set(np.unique(s)) <= {0, 1} and 0.4 < s.mean() < 0.6
    assert set(np.unique(K["tlwe_key"])) <= {0, 1}
    # secret keys: top bit of the words of streams (1, 0) and (2, 0)
    assert (s == (stream_words(SEED, 1, 0, n) >> 31)).all()
    assert (K["tlwe_key"].reshape(-1) == (stream_words(SEED, 2, 0, 2 * N) >> 31)).all()
    bk = K["bk"]
    # the oracle's key-structure assertions (tests/test_oracle_kat.py::test_reference_key_structure)
    assert not bk[:, 1:].any()                                    # rows >= 1 stay zero
    b = bk[:, 0, :, 1, :]
    assert (b[:, :, 0] == (s * l)[:, None]).all()                 # body: [s*l, 0 .. 0, g0]
    assert not b[:, :, 1:N - 1].any()
    assert not K["ks_a"][:, :, 0].any() and not K["ks_b"][:, :, 0].any() and not K["ks_var"][:, :, 0].any()
    assert (K["ks_var"][:, :, 1:] == P.ks_stdev ** 2).all()
    # mask polynomials: the keystream of (5, i*2 + col), coefficient 0 += s_i * l
    for i, col in ((0, 0), (3, 1), (n - 1, 1)):
        w = stream_words(SEED, 5, i * 2 + col, N).astype(np.int64)
        w[0] += int(s[i]) * l
        assert (bk[i, 0, col, 0].astype(np.uint32) == (w & 0xFFFFFFFF).astype(np.uint32)).all(), (i, col)
    # key-switch masks: the keystream of (4, row)
    for (i, j, h) in ((0, 0, 1), (7, 3, 2), (n - 1, 7, 3)):
        row = (i * P.ks_t + j) * 4 + h
        assert (K["ks_a"][i, j, h].astype(np.uint32) == stream_words(SEED, 4, row, n)).all(), (i, j, h)
    # the Gaussian coefficient of the bodies: n*2 draws of standard deviation bk_stdev
    g0 = b[:, :, N - 1].astype(np.float64) / 2.0 ** 32
    assert abs(g0.std() / P.bk_stdev - 1.0) < 0.12 and abs(g0.mean()) < 4 * P.bk_stdev / np.sqrt(g0.size)


def test_generated_key_switch_samples_decrypt_to_their_messages(generated80):
    """ks[i][j][h] encrypts key'[i] * h * 2^(32 - 2 (j+1)) under the LWE key (src/lwe.rs:543-557):
    the phase b - <a, s> minus the message is the recentred Gaussian noise."""
    P, hp, K = generated80
    s = K["lwe_key"].astype(np.int64)
    ext = K["tlwe_key"][0].astype(np.int64)                      # LweKey::extract: first polynomial
    a = K["ks_a"].astype(np.int64)
    dot = (a * s).sum(axis=-1)
    i = np.arange(P.n_ks)[:, None, None]
    j = np.arange(P.ks_t)[None, :, None]
    h = np.arange(4)[None, None, :]
    msg = ext[:P.n_ks][:, None, None] * h * (1 << (32 - 2 * (j + 1))) + 0 * i
    err = (K["ks_b"].astype(np.int64) - dot - msg)
    err = ((err + 2 ** 31) % 2 ** 32 - 2 ** 31)[:, :, 1:].astype(np.float64) / 2.0 ** 32
    assert abs(err.std() / P.ks_stdev - 1.0) < 0.05
    assert np.abs(err).max() < 7 * P.ks_stdev
    # recentred: the mean is zero up to the truncation of f64_to_torus_32 (at most 2^-32 each)
    assert abs(err.mean()) < 2.0 ** -32


def test_gates_on_generated_keys(eng, generated80):
    """Bit for bit equal to the oracle on the exported keys, and the closed form keys of this
    structure imply: a noiseless trivial sample +-mu whose decryption matches the oracle's."""
    P, hp, K = generated80
    B = 8
    bits = [(i & 1, (i >> 1) & 1) for i in range(B)]
    ca = [orc.encrypt(P, K["lwe_key"], x, 4100 + i) for i, (x, _) in enumerate(bits)]
    cb = [orc.encrypt(P, K["lwe_key"], y, 4200 + i) for i, (_, y) in enumerate(bits)]
    A = (np.stack([c[0] for c in ca]), np.array([c[1] for c in ca], np.int32))
    Bm = (np.stack([c[0] for c in cb]), np.array([c[1] for c in cb], np.int32))
    Ko = {k: np.ascontiguousarray(v) for k, v in K.items()}
    for g in ("nand", "xor"):
        oa, ob, var = eng.gate(g, A, Bm, want_var=True)
        for i in range(B):
            wa, wb, wv = orc.gate(P, g, Ko, ca[i], cb[i], None, orc.MUL_FFT)
            assert (oa[i] == wa).all() and ob[i] == wb and var[i] == wv, (g, i)
            assert not oa[i].any() and abs(int(ob[i])) == 1 << 29
            assert orc.decrypt(oa[i], ob[i], K["lwe_key"]) == orc.decrypt(wa, wb, K["lwe_key"])


def test_seed_reproduces_keys_and_key_file_round_trip(generated80):
    import tfhe_amd
    P, hp, K = generated80
    e2 = tfhe_amd.Engine([0, 0])          # multi-GPU context: generated on the first worker, replicated
    e3 = tfhe_amd.Engine(0)
    try:
        K2 = e2.generate_keys(hp, P.ks_stdev, P.bk_stdev, SEED)
        for name in K:
            assert (K[name] == K2[name]).all(), name
        K3 = e3.generate_keys(hp, P.ks_stdev, P.bk_stdev, bytes(32), export=False)
        assert sorted(K3) == ["lwe_key", "tlwe_key"] and (K3["lwe_key"] != K["lwe_key"]).any()
        # exported keys -> key file -> another context: same gate results as the generating one
        e3.load_key_file(e3.pack_key_file(hp, K["bk"], K["ks_a"], K["ks_b"], K["ks_var"]))
        a = (orc.uniform_i32(4301, (5, P.n)), orc.uniform_i32(4302, (5,)))
        b = (orc.uniform_i32(4303, (5, P.n)), orc.uniform_i32(4304, (5,)))
        r2, r3 = e2.gate("nand", a, b, want_var=True), e3.gate("nand", a, b, want_var=True)
        for x, y in zip(r2, r3):
            assert (x == y).all()
        # and both equal the oracle on the exported keys (first hand, not only each other)
        Ko = {k: np.ascontiguousarray(v) for k, v in K.items()}
        for i in (0, 4):
            wa, wb, wv = orc.gate(P, "nand", Ko, (a[0][i], a[1][i], 0.0), (b[0][i], b[1][i], 0.0), None, orc.MUL_FFT)
            assert (r3[0][i] == wa).all() and r3[1][i] == wb and r3[2][i] == wv, i
    finally:
        e2.close()
        e3.close()


def test_generate_keys_bit128_small_batch(eng):
    """Bit128 (l = 3): structure of the extra zero rows and the gadget value 3 * s."""
    import tfhe_amd
    P = orc.params(128)
    hp = tfhe_amd.make_params(P.n, P.N, P.k, P.l, P.bg_bit, P.ks_t, P.ks_base_bit, P.n_ks)
    K = eng.generate_keys(hp, P.ks_stdev, P.bk_stdev, SEED)
    assert K["bk"].shape == (630, 3, 2, 2, 1024) and not K["bk"][:, 1:].any()
    assert (K["bk"][:, 0, :, 1, 0] == (K["lwe_key"] * 3)[:, None]).all()
    a, b, _ = orc.encrypt(P, K["lwe_key"], True, 4400)
    oa, ob = eng.gate("and", (a[None], np.array([b], np.int32)), (a[None], np.array([b], np.int32)))
    Ko = {k: np.ascontiguousarray(v) for k, v in K.items()}
    wa, wb, _ = orc.gate(P, "and", Ko, (a, b, 0.0), (a, b, 0.0), None, orc.MUL_FFT)
    assert (oa[0] == wa).all() and ob[0] == wb
