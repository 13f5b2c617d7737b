// textbook.cpp — CPU oracle of the TEXTBOOK-CORRECT variant of the bootstrapped gate.
//
// TEST INFRASTRUCTURE ONLY (see oracle.h): the checker of libtfhe_hip_textbook.so.
//
// The reference crate's bootstrapping does not implement TFHE (its own docs say so,
// src/lib.rs:6): with its key structure a bootstrapped gate returns +-mu by the sign of the
// input's body alone.  This file restates the same functions with the deviations from TFHE
// corrected, one by one, at the lines where the crate has them:
//   * gadget values h[p] = 2^(32 - (p+1) bg_bit) for every level and the decomposition offset
//     summed over all l levels                       (src/tgsw.rs:43-52 seed only level 0)
//   * every row of a TGSW sample is a TLWE encryption of zero; message * h[p] goes to
//     polynomial j of row (j, p)                     (src/tgsw.rs:118-127 encrypts row 0 only,
//                                                     :129-167 adds to every polynomial)
//   * a TLWE sample has k mask polynomials and a body b = e + sum a_j (x) s_j, (x) the negacyclic
//     product mod X^N + 1                            (src/tlwe.rs:86 allocates k + 2 polynomials,
//                                                     :112-134 multiplies the key with zeros;
//                                                     src/numerics.rs:185-261 is a plain
//                                                     convolution with saturation)
//   * the external product decomposes and updates all k + 1 polynomials, body included
//                                                    (src/tgsw.rs:229-297 skip the last one)
//   * blind rotation is acc <- acc + bk_i [.] (X^abar acc - acc), abar, bbar in [0, 2N)
//                                                    (src/bootstrapping.rs:109-146 mixes in the
//                                                     previous accumulator; :50-55 and
//                                                     src/numerics.rs:126-130 yield [-N, N))
//   * the key-switch key has N k rows                (src/lwe.rs:458-461 builds n)
//   * NOR bootstraps with MU = +1/8                  (src/gates.rs:136 uses -1/8: inverted output)
// Everything else (gate weights src/gates.rs, sample extraction src/tlwe.rs:159-180, the key
// switch loop src/lwe.rs:601-648, encode/decode) is the crate's.  All arithmetic is exact:
// products are wrapping 32-bit integer convolutions, so this oracle has no rounding at all and
// the GPU path must agree with it bit for bit.
#include <cstring>
#include <vector>

#include "oracle.h"

namespace {

inline int32_t wadd(int32_t a, int32_t b) { return (int32_t)((uint32_t)a + (uint32_t)b); }
inline int32_t wsub(int32_t a, int32_t b) { return (int32_t)((uint32_t)a - (uint32_t)b); }
inline int32_t wmul(int32_t a, int32_t b) { return (int32_t)((uint32_t)a * (uint32_t)b); }
inline int32_t wneg(int32_t a) { return (int32_t)(0u - (uint32_t)a); }

// result += d (x) s mod (X^N + 1), wrapping
__attribute__((optimize("O3"))) void negacyclic_mac(uint32_t* __restrict__ result, const int32_t* __restrict__ d,
                                                    const int32_t* __restrict__ s, int N) {
  std::vector<uint32_t> full(2 * (size_t)N, 0u);
  for (int u = 0; u < N; u++) {
    const uint32_t du = (uint32_t)d[u];
    if (du == 0) continue;
    uint32_t* f = full.data() + u;
    for (int v = 0; v < N; v++) f[v] += du * (uint32_t)s[v];
  }
  for (int t = 0; t < N; t++) result[t] += full[t] - full[t + N];
}

// X^p * poly mod (X^N + 1), p in [0, 2N)   (src/polynomial.rs:281-313, its p >= 0 branches)
void rotate(const int32_t* p, int N, int power, int32_t* out) {
  const bool flip = power >= N;
  const int w = flip ? power - N : power;
  for (int j = 0; j < N; j++) {
    const int32_t v = j < w ? wneg(p[N - w + j]) : p[j - w];
    out[j] = flip ? wneg(v) : v;
  }
}

// modSwitchFromTorus32(x, 2N): the nearest multiple of 1/2N, in [0, 2N)
inline int32_t mod_switch(int32_t x, int N) {
  int log2n = 0;
  while ((1 << log2n) < 2 * N) log2n++;
  return (int32_t)(((uint32_t)x + (1u << (31 - log2n))) >> (32 - log2n));
}

struct Rng {          // splitmix64-seeded xoshiro256**, as oracle.cpp
  uint64_t s[4];
  explicit Rng(uint64_t seed) {
    uint64_t z = seed;
    for (int i = 0; i < 4; i++) {
      z += 0x9E3779B97F4A7C15ull;
      uint64_t x = z;
      x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull;
      x = (x ^ (x >> 27)) * 0x94D049BB133111EBull;
      s[i] = x ^ (x >> 31);
    }
  }
  static uint64_t rotl(uint64_t x, int k) { return (x << k) | (x >> (64 - k)); }
  uint64_t next() {
    uint64_t r = rotl(s[1] * 5, 7) * 9, t = s[1] << 17;
    s[2] ^= s[0]; s[3] ^= s[1]; s[1] ^= s[2]; s[0] ^= s[3];
    s[2] ^= t; s[3] = rotl(s[3], 45);
    return r;
  }
  int32_t i32() { return (int32_t)(next() >> 32); }
  int bit() { return (int)(next() >> 63); }
  double unit() { return (double)(next() >> 11) * (1.0 / 9007199254740992.0); }
  double normal(double sigma);
};

}  // namespace

#include <cmath>
double Rng::normal(double sigma) {
  double u1 = unit(), u2 = unit();
  if (u1 < 1e-300) u1 = 1e-300;
  return sigma * std::sqrt(-2.0 * std::log(u1)) * std::cos(6.283185307179586476925 * u2);
}

extern "C" {

// Decomposition with the offset over all levels (src/tgsw.rs:299-328 with :50-52 corrected).
// out is [l][N].
void orc_tb_decompose(const int32_t* poly, int32_t N, int32_t l, int32_t bg_bit, int32_t* out) {
  const int32_t bg = 1 << bg_bit, half = bg / 2, mask = bg - 1;
  uint32_t offset = 0;
  for (int p = 0; p < l; p++) offset += (uint32_t)half << (32 - (p + 1) * bg_bit);
  for (int j = 0; j < N; j++) {
    const uint32_t buf = (uint32_t)poly[j] + offset;
    for (int p = 0; p < l; p++) out[(size_t)p * N + j] = (int32_t)((buf >> (32 - (p + 1) * bg_bit)) & mask) - half;
  }
}

// acc [k+1][N] (mask polynomials, then the body); bk_i [l][k+1][k+1][N]: row (level p,
// component j), polynomial u; result [k+1][N] = sum_{j,p} D_{j,p} (x) bk_i[p][j][u]
// (src/tgsw.rs:229-264 with every polynomial decomposed and written).
void orc_tb_extern_product(const orc_params* P, const int32_t* acc, const int32_t* bk_i, int32_t* result) {
  const int N = P->N, k = P->k, l = P->l;
  std::vector<int32_t> dec((size_t)l * N);
  std::memset(result, 0, sizeof(int32_t) * (size_t)(k + 1) * N);
  for (int j = 0; j <= k; j++) {
    orc_tb_decompose(acc + (size_t)j * N, N, l, P->bg_bit, dec.data());
    for (int p = 0; p < l; p++)
      for (int u = 0; u <= k; u++)
        negacyclic_mac((uint32_t*)result + (size_t)u * N, dec.data() + (size_t)p * N,
                       bk_i + ((((size_t)p * (k + 1)) + j) * (k + 1) + u) * N, N);
  }
}

// acc <- acc + bk_i [.] (X^bara_i acc - acc) for every bara_i != 0, bara_i in [0, 2N)
// (src/bootstrapping.rs:109-146 without the previous-accumulator term).
void orc_tb_blind_rotate(const orc_params* P, int32_t* acc, const int32_t* bara, const int32_t* bk) {
  const int N = P->N, k = P->k;
  const size_t stride = (size_t)P->l * (k + 1) * (k + 1) * N, words = (size_t)(k + 1) * N;
  std::vector<int32_t> t(words), r(words);
  for (int i = 0; i < P->n; i++) {
    if (bara[i] == 0) continue;
    for (int j = 0; j <= k; j++) {
      rotate(acc + (size_t)j * N, N, bara[i], t.data() + (size_t)j * N);
      for (int x = 0; x < N; x++) t[(size_t)j * N + x] = wsub(t[(size_t)j * N + x], acc[(size_t)j * N + x]);
    }
    orc_tb_extern_product(P, t.data(), bk + (size_t)i * stride, r.data());
    for (size_t x = 0; x < words; x++) acc[x] = wadd(acc[x], r[x]);
  }
}

// tfhe_bootstrap_without_key_switching (src/bootstrapping.rs:38-99) with the unsigned mod
// switch; out_a has N k entries.
void orc_tb_bootstrap_noks(const orc_params* P, const int32_t* bk, int32_t mu, const int32_t* a, int32_t b,
                           int32_t* out_a, int32_t* out_b) {
  const int N = P->N, k = P->k;
  const int32_t barb = mod_switch(b, N);
  std::vector<int32_t> bara(P->n), acc((size_t)(k + 1) * N, 0), tv(N, mu);
  for (int i = 0; i < P->n; i++) bara[i] = mod_switch(a[i], N);
  // acc = trivial(X^(2N - barb) * [mu]*N), bootstrapping.rs:89-96
  rotate(tv.data(), N, (2 * N - barb) % (2 * N), acc.data() + (size_t)k * N);
  orc_tb_blind_rotate(P, acc.data(), bara.data(), bk);
  // extract_lwe, src/tlwe.rs:159-180
  for (int i = 0; i < k; i++) {
    out_a[(size_t)i * N] = acc[(size_t)i * N];
    for (int j = 1; j < N; j++) out_a[(size_t)i * N + j] = wneg(acc[(size_t)i * N + N - j]);
  }
  *out_b = acc[(size_t)k * N];
}

// tfhe_bootstrap (src/bootstrapping.rs:25-29); the key switch is the crate's loop over
// P->n_ks = N k rows (orc_key_switch).
void orc_tb_bootstrap(const orc_params* P, const int32_t* bk, const int32_t* ks_a, const int32_t* ks_b,
                      const double* ks_var, int32_t mu, const int32_t* a, int32_t b, int32_t* out_a,
                      int32_t* out_b, double* out_var) {
  std::vector<int32_t> ea((size_t)P->N * P->k);
  int32_t eb;
  orc_tb_bootstrap_noks(P, bk, mu, a, b, ea.data(), &eb);
  orc_key_switch(P, ks_a, ks_b, ks_var, ea.data(), eb, out_a, out_b, out_var);
}

// The gates of src/gates.rs:23-268 on the corrected bootstrap.
int orc_tb_gate(const orc_params* P, int gate, const int32_t* bk, const int32_t* ks_a, const int32_t* ks_b,
                const double* ks_var, const int32_t* ca_a, int32_t ca_b, const int32_t* cb_a, int32_t cb_b,
                const int32_t* cc_a, int32_t cc_b, int32_t* out_a, int32_t* out_b, double* out_var) {
  const int n = P->n;
  if (gate == ORC_GATE_MUX) {                        // gates.rs:239-268
    const int NK = P->N * P->k;
    std::vector<int32_t> x(n), u1(NK), u2(NK), s(NK);
    int32_t xb, u1b, u2b, mu;
    double xv;
    orc_gate_prologue(ORC_GATE_AND, n, ca_a, ca_b, 0.0, cb_a, cb_b, 0.0, x.data(), &xb, &xv, &mu);
    orc_tb_bootstrap_noks(P, bk, mu, x.data(), xb, u1.data(), &u1b);
    orc_gate_prologue(ORC_GATE_ANDNY, n, ca_a, ca_b, 0.0, cc_a, cc_b, 0.0, x.data(), &xb, &xv, &mu);
    orc_tb_bootstrap_noks(P, bk, mu, x.data(), xb, u2.data(), &u2b);
    for (int i = 0; i < NK; i++) s[i] = wadd(u1[i], u2[i]);
    const int32_t sb = wadd(wadd(orc_encode_message(1, 8), u1b), u2b);
    orc_key_switch(P, ks_a, ks_b, ks_var, s.data(), sb, out_a, out_b, out_var);
    return 0;
  }
  std::vector<int32_t> x(n);
  int32_t xb, mu;
  double xv;
  if (orc_gate_prologue(gate, n, ca_a, ca_b, 0.0, cb_a, cb_b, 0.0, x.data(), &xb, &xv, &mu)) return -1;
  if (gate == ORC_GATE_NOR) mu = orc_encode_message(1, 8);      // gates.rs:136 corrected
  orc_tb_bootstrap(P, bk, ks_a, ks_b, ks_var, mu, x.data(), xb, out_a, out_b, out_var);
  return 0;
}

// generate_keys (src/encryption.rs:39-47) with the corrected structures.  Buffers:
//   lwe_key [n]; tlwe_key [k][N]; bk [n][l][k+1][k+1][N];
//   ks_a [N k][t][base][n]; ks_b, ks_var [N k][t][base]   (P->n_ks must be N k).
void orc_tb_generate_keys(const orc_params* P, uint64_t seed, int32_t* lwe_key, int32_t* tlwe_key, int32_t* bk,
                          int32_t* ks_a, int32_t* ks_b, double* ks_var) {
  const int n = P->n, N = P->N, k = P->k, l = P->l, t = P->ks_t, bb = P->ks_base_bit, base = 1 << bb;
  Rng rng(seed);
  for (int i = 0; i < n; i++) lwe_key[i] = rng.bit();                  // lwe.rs:294-309
  for (int i = 0; i < k * N; i++) tlwe_key[i] = rng.bit();             // tlwe.rs:43-64
  // key-switch key over the N k extracted key bits (lwe.rs:530-558 with lwe.rs:461 corrected)
  {
    const double alpha = P->ks_stdev;
    const int size_ks = P->n_ks * t * (base - 1);
    std::vector<double> noise(size_ks);
    double sum = 0.0;
    for (int i = 0; i < size_ks; i++) { noise[i] = rng.normal(alpha); sum += noise[i]; }
    const double error = sum / (double)size_ks;
    int index = 0;
    for (int i = 0; i < P->n_ks; i++)
      for (int j = 0; j < t; j++) {
        const size_t z = ((size_t)i * t + j) * base;
        std::memset(ks_a + z * n, 0, sizeof(int32_t) * n);
        ks_b[z] = 0;
        ks_var[z] = 0.0;
        for (int h = 1; h < base; h++) {
          const size_t idx = z + h;
          const int32_t message = wmul(tlwe_key[i] * h, (int32_t)(1u << (32 - (j + 1) * bb)));
          int32_t b = wadd(message, orc_f64_to_torus32(noise[index++] - error));
          for (int x = 0; x < n; x++) {
            const int32_t a = rng.i32();
            ks_a[idx * n + x] = a;
            b = wadd(b, wmul(a, lwe_key[x]));
          }
          ks_b[idx] = b;
          ks_var[idx] = alpha * alpha;
        }
      }
  }
  // bk[i] = TGSW encryption of lwe_key[i]: every row (p, j) a TLWE encryption of zero, plus
  // lwe_key[i] * h[p] on coefficient 0 of polynomial j
  const double alpha = P->bk_stdev;
  std::vector<uint32_t> body(N);
  for (int i = 0; i < n; i++)
    for (int p = 0; p < l; p++)
      for (int j = 0; j <= k; j++) {
        int32_t* row = bk + ((((size_t)i * l + p) * (k + 1)) + j) * (k + 1) * N;
        for (int x = 0; x < N; x++) body[x] = (uint32_t)orc_f64_to_torus32(rng.normal(alpha));
        for (int u = 0; u < k; u++) {
          for (int x = 0; x < N; x++) row[(size_t)u * N + x] = rng.i32();
          negacyclic_mac(body.data(), tlwe_key + (size_t)u * N, row + (size_t)u * N, N);
        }
        for (int x = 0; x < N; x++) row[(size_t)k * N + x] = (int32_t)body[x];
        const int32_t h = (int32_t)(1u << (32 - (p + 1) * P->bg_bit));
        row[(size_t)j * N] = wadd(row[(size_t)j * N], wmul(lwe_key[i], h));
      }
}

}  // extern "C"
