// oracle.cpp — CPU restatement of the reference's bootstrapped-gate path.
// TEST INFRASTRUCTURE ONLY (see oracle.h).  Build with -fwrapv: wrapping
// signed arithmetic is the specification (release-mode Rust).
//
// Every function names the reference lines it follows; /root/reference is the
// IsakSundeSingh/tfhe crate.

#include "oracle.h"

#include <algorithm>
#include <cmath>
#include <complex>
#include <cstring>
#include <vector>

namespace {

using cplx = std::complex<double>;
typedef __int128 i128;

inline int32_t wadd(int32_t a, int32_t b) { return (int32_t)((uint32_t)a + (uint32_t)b); }
inline int32_t wsub(int32_t a, int32_t b) { return (int32_t)((uint32_t)a - (uint32_t)b); }
inline int32_t wmul(int32_t a, int32_t b) { return (int32_t)((uint32_t)a * (uint32_t)b); }
inline int32_t wneg(int32_t a) { return (int32_t)(0u - (uint32_t)a); }

// Rust `x as i32` for f64: saturating, NaN -> 0.
inline int32_t sat_cast(double x) {
  if (x != x) return 0;
  if (x >= 2147483647.0) return INT32_MAX;
  if (x <= -2147483648.0) return INT32_MIN;
  return (int32_t)x;
}

// numerics.rs:29-38  Modulo::modulo
inline int32_t modulo(int32_t a, int32_t m) {
  int32_t r = a % m;
  return r < 0 ? r + std::abs(m) : r;
}

// ---------------------------------------------------------------- RNG
struct Rng {
  uint64_t s[4];
  explicit Rng(uint64_t seed) {
    uint64_t z = seed;
    for (int i = 0; i < 4; i++) {  // splitmix64
      z += 0x9E3779B97F4A7C15ull;
      uint64_t x = z;
      x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull;
      x = (x ^ (x >> 27)) * 0x94D049BB133111EBull;
      s[i] = x ^ (x >> 31);
    }
  }
  static uint64_t rotl(uint64_t x, int k) { return (x << k) | (x >> (64 - k)); }
  uint64_t next() {  // xoshiro256**
    uint64_t r = rotl(s[1] * 5, 7) * 9, t = s[1] << 17;
    s[2] ^= s[0]; s[3] ^= s[1]; s[1] ^= s[2]; s[0] ^= s[3];
    s[2] ^= t; s[3] = rotl(s[3], 45);
    return r;
  }
  int32_t i32() { return (int32_t)(next() >> 32); }
  int bit() { return (int)(next() >> 63); }
  double unit() { return (double)(next() >> 11) * (1.0 / 9007199254740992.0); }
  double normal(double sigma) {
    double u1 = unit(), u2 = unit();
    if (u1 < 1e-300) u1 = 1e-300;
    return sigma * std::sqrt(-2.0 * std::log(u1)) * std::cos(6.283185307179586476925 * u2);
  }
};

// numerics.rs:42-52
inline int32_t gaussian32(Rng& r, int32_t message, double sigma) {
  return wadd(message, orc_f64_to_torus32(r.normal(sigma)));
}

// ---------------------------------------------------------------- FFT
// Iterative radix-2 complex FFT (any correct f64 FFT serves: the rounded
// result is independent of the FFT's internals while its error is < 0.5).
struct Plan {
  int n = 0;
  std::vector<cplx> w;      // w[j] = exp(-2*pi*i*j/n), j < n/2
  std::vector<int> rev;
  void build(int len) {
    n = len;
    w.resize(len / 2);
    for (int j = 0; j < len / 2; j++) {
      double ang = -6.283185307179586476925 * (double)j / (double)len;
      w[j] = cplx(std::cos(ang), std::sin(ang));
    }
    rev.resize(len);
    int bits = 0;
    while ((1 << bits) < len) bits++;
    for (int i = 0; i < len; i++) {
      int r = 0;
      for (int b = 0; b < bits; b++) r |= ((i >> b) & 1) << (bits - 1 - b);
      rev[i] = r;
    }
  }
  void run(cplx* x, bool inverse) const {
    for (int i = 0; i < n; i++)
      if (i < rev[i]) std::swap(x[i], x[rev[i]]);
    for (int half = 1; half < n; half <<= 1) {
      int step = n / (2 * half);
      for (int base = 0; base < n; base += 2 * half)
        for (int j = 0; j < half; j++) {
          cplx tw = w[j * step];
          if (inverse) tw = std::conj(tw);
          cplx u = x[base + j], v = x[base + j + half] * tw;
          x[base + j] = u + v;
          x[base + j + half] = u - v;
        }
    }
  }
};

const Plan& cached_plan(int len) {
  static thread_local Plan plans[32];
  int bits = 0;
  while ((1 << bits) < len) bits++;
  if (plans[bits].n != len) plans[bits].build(len);
  return plans[bits];
}

int next_pow2(int v) {
  int p = 1;
  while (p < v) p <<= 1;
  return p;
}

// numerics.rs:185-261, FFT variant.
void poly_multiplier_fft(const int32_t* a, int la, const int32_t* b, int lb, int32_t* out,
                         bool plan_per_call, double* max_frac) {
  int degree = (la - 1) + (lb - 1);
  int plen = next_pow2(la + lb - 1);           // :206-214
  std::vector<cplx> p(plen, cplx(0, 0)), q(plen, cplx(0, 0));
  for (int i = 0; i < la; i++) p[i] = cplx((double)a[la - 1 - i], 0.0);   // .rev(), :198-204
  for (int i = 0; i < lb; i++) q[i] = cplx((double)b[lb - 1 - i], 0.0);   // :216-222
  Plan local;
  const Plan* plan;
  if (plan_per_call) { local.build(plen); plan = &local; }                // :235-236 fresh planner
  else plan = &cached_plan(plen);
  plan->run(p.data(), false);                                             // :237-238
  plan->run(q.data(), false);
  double sp = std::sqrt((double)plen);
  std::vector<cplx> r(plen);
  for (int i = 0; i < plen; i++) r[i] = (p[i] / sp) * (q[i] / sp);         // :243-247
  Plan local2;
  if (plan_per_call) { local2.build(plen); plan = &local2; }              // :250 second fresh planner
  plan->run(r.data(), true);                                              // :251 (unnormalised)
  double mf = 0.0;
  for (int i = 0; i <= degree; i++) {                                     // :253-258 take, round, cast, rev
    double x = r[i].real();
    double rx = std::round(x);
    double f = std::fabs(x - rx);
    if (f > mf) mf = f;
    out[degree - i] = sat_cast(rx);
  }
  if (max_frac && mf > *max_frac) *max_frac = mf;
}

inline int32_t sat_i128(i128 v) {
  if (v > (i128)INT32_MAX) return INT32_MAX;
  if (v < (i128)INT32_MIN) return INT32_MIN;
  return (int32_t)v;
}

// Exact value of the product the FFT approximates; first `nout` coefficients.
void poly_multiplier_exact(const int32_t* a, int la, const int32_t* b, int lb, int32_t* out,
                           int nout) {
  int64_t ma = 0, mb = 0;
  for (int i = 0; i < la; i++) ma = std::max<int64_t>(ma, std::llabs((long long)a[i]));
  for (int i = 0; i < lb; i++) mb = std::max<int64_t>(mb, std::llabs((long long)b[i]));
  bool fits64 = ((i128)ma * mb * std::min(la, lb)) < ((i128)1 << 62);
  for (int t = 0; t < nout; t++) {
    int ulo = std::max(0, t - (lb - 1)), uhi = std::min(t, la - 1);
    if (fits64) {
      int64_t acc = 0;
      for (int u = ulo; u <= uhi; u++) acc += (int64_t)a[u] * (int64_t)b[t - u];
      out[t] = sat_i128((i128)acc);
    } else {
      i128 acc = 0;
      for (int u = ulo; u <= uhi; u++) acc += (i128)a[u] * (i128)b[t - u];
      out[t] = sat_i128(acc);
    }
  }
}

inline size_t bk_sample_stride(const orc_params* P) {
  return (size_t)P->l * (P->k + 1) * (P->k + 1) * P->N;
}

}  // namespace

extern "C" {

int orc_params_for(int level, orc_params* o) {
  // lwe.rs:179-232
  o->N = 1024; o->k = 1; o->ks_t = 8; o->ks_base_bit = 2;
  if (level == 80) {
    o->n = 500; o->l = 2; o->bg_bit = 10; o->ks_stdev = 2.44e-5; o->bk_stdev = 7.18e-9;
  } else if (level == 128) {
    o->n = 630; o->l = 3; o->bg_bit = 7;
    o->ks_stdev = std::pow(2.0, -15.0); o->bk_stdev = std::pow(2.0, -15.0);
  } else {
    return -1;
  }
  o->n_ks = o->n;  // lwe.rs:461: the key-switch key is built with the in/out n
  return 0;
}

int32_t orc_encode_message(int32_t mu, int32_t space) {
  int log2 = __builtin_ctz((unsigned)space);   // numerics.rs:117-120
  return (int32_t)((uint32_t)mu << (32 - log2));
}

int32_t orc_decode_message(int32_t phase, int32_t space) {
  int log2 = __builtin_ctz((unsigned)space);   // numerics.rs:126-130
  return wadd(phase, (int32_t)(1u << (32 - log2 - 1))) >> (32 - log2);
}

int32_t orc_f64_to_torus32(double d) { return sat_cast(std::trunc(d * 4294967296.0)); }

void orc_mul_by_monomial(const int32_t* p, int32_t n, int32_t power, int32_t* out) {
  // polynomial.rs:281-313, statement for statement.
  if (power == 0) { std::memcpy(out, p, sizeof(int32_t) * n); return; }
  bool cycle = modulo(power / n, 2) == 1;
  power = power % n;
  std::vector<int32_t> coefs(n, 0);
  bool shift_first = !cycle, shift_last = cycle;
  for (int32_t j = 0; j < power; j++) {
    int32_t v = p[modulo(n - power + j, n)];
    coefs[modulo(j, n)] = shift_first ? wneg(v) : v;
  }
  for (int32_t j = power; j < n; j++) {
    int32_t v = p[modulo(j - power, n)];
    coefs[modulo(j, n)] = shift_last ? wneg(v) : v;
  }
  std::memcpy(out, coefs.data(), sizeof(int32_t) * n);
}

static void padded_op(const int32_t* a, int32_t la, const int32_t* b, int32_t lb, int32_t* out,
                      bool sub) {
  // polynomial.rs:66-90: the shorter operand is left-padded with zeros.
  int32_t len = std::max(la, lb);
  for (int32_t i = 0; i < len; i++) {
    int32_t ia = i - (len - la), ib = i - (len - lb);
    int32_t va = ia >= 0 ? a[ia] : 0, vb = ib >= 0 ? b[ib] : 0;
    out[i] = sub ? wsub(va, vb) : wadd(va, vb);
  }
}
void orc_poly_add(const int32_t* a, int32_t la, const int32_t* b, int32_t lb, int32_t* out) {
  padded_op(a, la, b, lb, out, false);   // polynomial.rs:118-146
}
void orc_poly_sub(const int32_t* a, int32_t la, const int32_t* b, int32_t lb, int32_t* out) {
  padded_op(a, la, b, lb, out, true);    // polynomial.rs:147-173
}

void orc_poly_multiplier(const int32_t* a, int32_t la, const int32_t* b, int32_t lb, int32_t* out,
                         int mode, double* max_frac) {
  if (mode == ORC_MUL_EXACT) poly_multiplier_exact(a, la, b, lb, out, la + lb - 1);
  else poly_multiplier_fft(a, la, b, lb, out, mode == ORC_MUL_FFT_PLAN_PER_CALL, max_frac);
}

void orc_torus_polynomial_mul_r(int32_t* result, const int32_t* p1, int32_t la, const int32_t* p2,
                                int32_t lb, int mode, double* max_frac) {
  // numerics.rs:138-155: product, keep the first len(poly1) entries, wrapping add.
  std::vector<int32_t> prod((size_t)la + lb - 1);
  if (mode == ORC_MUL_EXACT) poly_multiplier_exact(p1, la, p2, lb, prod.data(), la);
  else poly_multiplier_fft(p1, la, p2, lb, prod.data(), mode == ORC_MUL_FFT_PLAN_PER_CALL, max_frac);
  for (int32_t t = 0; t < la; t++) result[t] = wadd(result[t], prod[t]);
}

void orc_decompose(const int32_t* poly, int32_t N, int32_t l, int32_t bg_bit, int32_t* out) {
  // tgsw.rs:40-52: half_bg, mask, and the offset summed over 0..1 only.
  int32_t bg = 1 << bg_bit, half_bg = bg >> 1;
  uint32_t mask_mod = (uint32_t)(bg - 1);
  uint32_t temp = 1u << (32 - bg_bit);
  uint32_t offset = temp * (uint32_t)half_bg;
  // tgsw.rs:299-328
  for (int32_t p = 0; p < l; p++) {
    int32_t decal = 32 - (p + 1) * bg_bit;
    for (int32_t j = 0; j < N; j++) {
      int32_t buf = wadd(poly[j], (int32_t)offset);
      int32_t temp1 = (buf >> decal) & (int32_t)mask_mod;
      out[(size_t)p * N + j] = temp1 - half_bg;
    }
  }
}

void orc_extern_mul(const orc_params* P, const int32_t* accum, const int32_t* bk_i,
                    const double* bk_i_var, int32_t* result, double* result_var, int mode,
                    double* max_frac) {
  const int N = P->N, k = P->k, l = P->l;
  // tgsw.rs:236-244: result has as many (zero) polynomials as accum, variance 0.
  std::memset(result, 0, sizeof(int32_t) * (size_t)(k + 2) * N);
  double var = 0.0;
  // tgsw.rs:280-297: dec[j][p] = level p of accum.a[j], j in 0..=k (never a[k+1]).
  std::vector<int32_t> dec((size_t)(k + 1) * l * N);
  for (int j = 0; j <= k; j++)
    orc_decompose(accum + (size_t)j * N, N, l, P->bg_bit, dec.data() + (size_t)j * l * N);
  // tgsw.rs:246-263: transpose, then pair (p, j) with all_sample[p][j] in flatten order.
  for (int p = 0; p < l; p++)
    for (int j = 0; j <= k; j++) {
      const int32_t* d = dec.data() + ((size_t)j * l + p) * N;
      const int32_t* sample = bk_i + ((size_t)p * (k + 1) + j) * (k + 1) * N;
      // tlwe.rs:149-157 add_mul_r_
      for (int u = 0; u <= k; u++)
        orc_torus_polynomial_mul_r(result + (size_t)u * N, d, N, sample + (size_t)u * N, N, mode,
                                   max_frac);
      double ns = 0.0;  // polynomial.rs:47-53
      for (int x = 0; x < N; x++) ns += (double)(d[x] * d[x]);
      if (bk_i_var) var += ns * bk_i_var[p * (k + 1) + j];
    }
  if (result_var) *result_var = var;
}

void orc_blind_rotate(const orc_params* P, int32_t* acc, double* acc_var, const int32_t* bara,
                      const int32_t* bk, const double* bk_var, int mode, double* max_frac) {
  const int N = P->N, k = P->k, l = P->l, polys = k + 2;
  const size_t sz = (size_t)polys * N;
  // bootstrapping.rs:116-118: temp = cleared copy, temp2 = accum.
  std::vector<int32_t> prev(sz, 0), cur(acc, acc + sz), t(sz), ext(sz), nxt(sz);
  double prev_var = 0.0, cur_var = acc_var ? *acc_var : 0.0;
  for (int i = 0; i < P->n; i++) {
    int32_t barai = bara[i];
    if (barai == 0) continue;                         // :122-126
    // :136-146 tfhe_mux_rotate(result = prev, accum = cur)
    for (int a = 0; a < polys; a++) {
      std::vector<int32_t> sum(N), rot(N);
      for (int x = 0; x < N; x++) sum[x] = wadd(prev[(size_t)a * N + x], cur[(size_t)a * N + x]);
      orc_mul_by_monomial(sum.data(), N, barai, rot.data());      // tlwe.rs:216-226
      for (int x = 0; x < N; x++) t[(size_t)a * N + x] = wsub(rot[x], cur[(size_t)a * N + x]);
    }
    double ext_var = 0.0;
    orc_extern_mul(P, t.data(), bk + (size_t)i * bk_sample_stride(P),
                   bk_var ? bk_var + (size_t)i * l * (k + 1) : nullptr, ext.data(), &ext_var, mode,
                   max_frac);
    for (size_t x = 0; x < sz; x++) nxt[x] = wadd(cur[x], ext[x]);
    double nxt_var = cur_var + ext_var;               // tlwe.rs:193
    // :128-129 temp = new; swap(temp, temp2)
    prev = cur; prev_var = cur_var;
    cur = nxt; cur_var = nxt_var;
  }
  (void)prev_var;
  std::memcpy(acc, cur.data(), sizeof(int32_t) * sz);
  if (acc_var) *acc_var = cur_var;
}

void orc_extract_lwe(const orc_params* P, const int32_t* acc, int32_t* out_a, int32_t* out_b) {
  const int N = P->N, k = P->k;   // tlwe.rs:159-180
  for (int i = 0; i < k; i++) {
    out_a[i * N] = acc[(size_t)i * N];
    for (int j = 1; j < N; j++) out_a[i * N + j] = wneg(acc[(size_t)i * N + (N - j)]);
  }
  *out_b = acc[(size_t)(k + 1) * N];  // last polynomial of k+2
}

void orc_bootstrap_noks(const orc_params* P, const int32_t* bk, const double* bk_var, int32_t mu,
                        const int32_t* a, int32_t b, int32_t* out_a, int32_t* out_b,
                        double* out_var, int mode, double* max_frac) {
  const int N = P->N, k = P->k, n = P->n, n2 = 2 * N;
  // bootstrapping.rs:50-55
  int32_t barb = orc_decode_message(b, n2);
  std::vector<int32_t> bara(n);
  for (int i = 0; i < n; i++) bara[i] = orc_decode_message(a[i], n2);
  std::vector<int32_t> tv(N, mu), tvb(N);            // :58-62
  // :89-94
  if (barb != 0) orc_mul_by_monomial(tv.data(), N, modulo(-barb, n2), tvb.data());
  else tvb = tv;
  // tlwe.rs:99-109: k+2 polynomials, mu in the last one.
  std::vector<int32_t> acc((size_t)(k + 2) * N, 0);
  std::memcpy(acc.data() + (size_t)(k + 1) * N, tvb.data(), sizeof(int32_t) * N);
  double var = 0.0;
  orc_blind_rotate(P, acc.data(), &var, bara.data(), bk, bk_var, mode, max_frac);
  orc_extract_lwe(P, acc.data(), out_a, out_b);
  if (out_var) *out_var = 0.0;   // tlwe.rs:165: LweSample::new, variance never copied
}

void orc_key_switch(const orc_params* P, const int32_t* ks_a, const int32_t* ks_b,
                    const double* ks_var, const int32_t* in_a, int32_t in_b, int32_t* out_a,
                    int32_t* out_b, double* var_io) {
  const int n = P->n, t = P->ks_t, bb = P->ks_base_bit, base = 1 << bb;
  // lwe.rs:601-603: r = trivial(sample.b), variance 0
  for (int x = 0; x < n; x++) out_a[x] = 0;
  int32_t rb = in_b;
  double var = 0.0;
  // lwe.rs:624-648
  int32_t prec_offset = 1 << (32 - (1 + bb * t));
  int32_t mask = base - 1;
  for (int i = 0; i < P->n_ks; i++) {
    int32_t aibar = wadd(in_a[i], prec_offset);
    for (int j = 0; j < t; j++) {
      int32_t aij = (aibar >> (32 - (j + 1) * bb)) & mask;
      if (aij != 0) {
        size_t idx = ((size_t)i * t + j) * base + aij;
        const int32_t* row = ks_a + idx * n;
        for (int x = 0; x < n; x++) out_a[x] = wsub(out_a[x], row[x]);
        rb = wsub(rb, ks_b[idx]);
        if (ks_var) var += ks_var[idx];
      }
    }
  }
  *out_b = rb;
  if (var_io) *var_io = var;
}

void orc_bootstrap(const orc_params* P, const int32_t* bk, const double* bk_var,
                   const int32_t* ks_a, const int32_t* ks_b, const double* ks_var, int32_t mu,
                   const int32_t* a, int32_t b, int32_t* out_a, int32_t* out_b, double* out_var,
                   int mode, double* max_frac) {
  std::vector<int32_t> ea((size_t)P->N * P->k);
  int32_t eb;
  double v = 0.0;
  orc_bootstrap_noks(P, bk, bk_var, mu, a, b, ea.data(), &eb, &v, mode, max_frac);
  orc_key_switch(P, ks_a, ks_b, ks_var, ea.data(), eb, out_a, out_b, out_var);
}

int orc_gate_prologue(int gate, int32_t n, const int32_t* ca_a, int32_t ca_b, double ca_var,
                      const int32_t* cb_a, int32_t cb_b, double cb_var, int32_t* x_a, int32_t* x_b,
                      double* x_var, int32_t* mu) {
  const int32_t e8 = orc_encode_message(1, 8), m8 = orc_encode_message(-1, 8);
  const int32_t e4 = orc_encode_message(1, 4), m4 = orc_encode_message(-1, 4);
  int32_t c, wa, wb, m = e8;
  switch (gate) {                                   // gates.rs
    case ORC_GATE_NAND:  c = e8; wa = -1; wb = -1; break;            // :23-41
    case ORC_GATE_OR:    c = e8; wa = +1; wb = +1; break;            // :44-61
    case ORC_GATE_AND:   c = m8; wa = +1; wb = +1; break;            // :64-80
    case ORC_GATE_XOR:   c = e4; wa = +2; wb = +2; break;            // :83-99
    case ORC_GATE_XNOR:  c = m4; wa = -2; wb = -2; break;            // :102-119
    case ORC_GATE_NOR:   c = m8; wa = -1; wb = -1; m = m8; break;    // :127-143
    case ORC_GATE_ANDNY: c = m8; wa = -1; wb = +1; break;            // :147-162
    case ORC_GATE_ANDYN: c = m8; wa = +1; wb = -1; break;            // :166-182
    case ORC_GATE_ORNY:  c = e8; wa = -1; wb = +1; break;            // :186-202
    case ORC_GATE_ORYN:  c = e8; wa = +1; wb = -1; break;            // :206-222
    default: return -1;
  }
  // (trivial(c) op |wa|*ca) op |wb|*cb; 2*x is wrapping_mul and scales the
  // variance by 4 (lwe.rs:113-128); add and sub both sum variances.
  int32_t sa = std::abs(wa), sb = std::abs(wb);
  for (int i = 0; i < n; i++) {
    int32_t va = wmul(ca_a[i], sa), vb = wmul(cb_a[i], sb);
    int32_t r = wa > 0 ? wadd(0, va) : wsub(0, va);
    x_a[i] = wb > 0 ? wadd(r, vb) : wsub(r, vb);
  }
  int32_t va = wmul(ca_b, sa), vb = wmul(cb_b, sb);
  int32_t r = wa > 0 ? wadd(c, va) : wsub(c, va);
  *x_b = wb > 0 ? wadd(r, vb) : wsub(r, vb);
  if (x_var) *x_var = (0.0 + (double)(sa * sa) * ca_var) + (double)(sb * sb) * cb_var;
  *mu = m;
  return 0;
}

int orc_gate(const orc_params* P, int gate, const int32_t* bk, const double* bk_var,
             const int32_t* ks_a, const int32_t* ks_b, const double* ks_var, const int32_t* ca_a,
             int32_t ca_b, double ca_var, const int32_t* cb_a, int32_t cb_b, double cb_var,
             const int32_t* cc_a, int32_t cc_b, double cc_var, int32_t* out_a, int32_t* out_b,
             double* out_var, int mode, double* max_frac) {
  const int n = P->n;
  if (gate == ORC_GATE_NOT) {            // gates.rs:122-124, lwe.rs:138-151
    for (int i = 0; i < n; i++) out_a[i] = wneg(ca_a[i]);
    *out_b = wneg(ca_b);
    if (out_var) *out_var = ca_var;
    return 0;
  }
  if (gate == ORC_GATE_MUX) {            // gates.rs:239-268
    const int NK = P->N * P->k;
    std::vector<int32_t> x(n), u1(NK), u2(NK), s(NK);
    int32_t xb, u1b, u2b, mu;
    double xv;
    orc_gate_prologue(ORC_GATE_AND, n, ca_a, ca_b, ca_var, cb_a, cb_b, cb_var, x.data(), &xb, &xv, &mu);
    orc_bootstrap_noks(P, bk, bk_var, mu, x.data(), xb, u1.data(), &u1b, nullptr, mode, max_frac);
    orc_gate_prologue(ORC_GATE_ANDNY, n, ca_a, ca_b, ca_var, cc_a, cc_b, cc_var, x.data(), &xb, &xv, &mu);
    orc_bootstrap_noks(P, bk, bk_var, mu, x.data(), xb, u2.data(), &u2b, nullptr, mode, max_frac);
    for (int i = 0; i < NK; i++) s[i] = wadd(wadd(0, u1[i]), u2[i]);
    int32_t sb = wadd(wadd(orc_encode_message(1, 8), u1b), u2b);
    orc_key_switch(P, ks_a, ks_b, ks_var, s.data(), sb, out_a, out_b, out_var);
    return 0;
  }
  std::vector<int32_t> x(n);
  int32_t xb, mu;
  double xv;
  if (orc_gate_prologue(gate, n, ca_a, ca_b, ca_var, cb_a, cb_b, cb_var, x.data(), &xb, &xv, &mu))
    return -1;
  orc_bootstrap(P, bk, bk_var, ks_a, ks_b, ks_var, mu, x.data(), xb, out_a, out_b, out_var, mode,
                max_frac);
  return 0;
}

void orc_generate_keys(const orc_params* P, uint64_t seed, int32_t* lwe_key, int32_t* tlwe_key,
                       int32_t* bk, double* bk_var, int32_t* ks_a, int32_t* ks_b, double* ks_var) {
  const int n = P->n, N = P->N, k = P->k, l = P->l, t = P->ks_t, bb = P->ks_base_bit;
  const int base = 1 << bb;
  Rng rng(seed);
  // encryption.rs:40, lwe.rs:294-309
  for (int i = 0; i < n; i++) lwe_key[i] = rng.bit();
  // encryption.rs:41, tgsw.rs:81-89, tlwe.rs:43-64: k+1 binary polynomials
  for (int i = 0; i < (k + 1) * N; i++) tlwe_key[i] = rng.bit();
  // lwe.rs:484, 369-386: extracted key = first k polynomials, N*k bits
  const int32_t* extracted = tlwe_key;
  // lwe.rs:485, 530-558: key-switch key with n_ks = n rows (lwe.rs:461)
  {
    const double alpha = P->ks_stdev;
    const int size_ks = P->n_ks * t * (base - 1);
    std::vector<double> noise(size_ks);
    double sum = 0.0;
    for (int i = 0; i < size_ks; i++) { noise[i] = rng.normal(alpha); sum += noise[i]; }
    double error = sum / (double)size_ks;
    for (int i = 0; i < size_ks; i++) noise[i] -= error;
    int index = 0;
    for (int i = 0; i < P->n_ks; i++)
      for (int j = 0; j < t; j++) {
        size_t z = ((size_t)i * t + j) * base;          // h = 0: all-zero sample, variance 0
        std::memset(ks_a + z * n, 0, sizeof(int32_t) * n);
        ks_b[z] = 0;
        ks_var[z] = 0.0;
        for (int h = 1; h < base; h++) {
          size_t idx = z + h;
          int32_t message = wmul(extracted[i] * h, (int32_t)(1u << (32 - (j + 1) * bb)));
          // lwe.rs:334-358 encrypt_with_external_noise
          int32_t b = wadd(message, orc_f64_to_torus32(noise[index]));
          int32_t dot = 0;
          for (int x = 0; x < n; x++) {
            int32_t a = rng.i32();
            ks_a[idx * n + x] = a;
            dot = wadd(dot, wmul(a, lwe_key[x]));
          }
          ks_b[idx] = wadd(b, dot);
          ks_var[idx] = alpha * alpha;
          index++;
        }
      }
  }
  // lwe.rs:487-490: bk[i] = TGSW encryption of key_in.key[i]
  const double alpha = P->bk_stdev;
  const int full = 2 * N - 1;
  for (int i = 0; i < n; i++) {
    int32_t* s = bk + (size_t)i * bk_sample_stride(P);
    std::memset(s, 0, sizeof(int32_t) * bk_sample_stride(P));   // tgsw.rs:108-116
    for (int p = 0; p < l; p++)
      for (int j = 0; j <= k; j++) bk_var[((size_t)i * l + p) * (k + 1) + j] = 0.0;
    // tgsw.rs:118-127: encrypt_zero on row 0 only
    for (int j = 0; j <= k; j++) {
      int32_t* smp = s + ((size_t)0 * (k + 1) + j) * (k + 1) * N;
      // tlwe.rs:112-134
      for (int u = 0; u < k; u++)
        for (int x = 0; x < N; x++) smp[(size_t)u * N + x] = rng.i32();     // a_part uniform
      std::vector<int32_t> last(N);
      for (int x = 0; x < N; x++) last[x] = gaussian32(rng, 0, alpha);       // a_last_part
      // poly_sum: products of the key with the sample's OLD (all-zero)
      // polynomials: 2N-1 zeros; the left-padded add gives a 2N-1 polynomial
      // whose first N-1 entries are zero.
      std::vector<int32_t> zero(N, 0), prod(full), sum(N, 0), tmp;
      int sum_len = N;
      for (int m = 0; m <= k; m++) {
        orc_poly_multiplier(tlwe_key + (size_t)m * N, N, zero.data(), N, prod.data(), ORC_MUL_EXACT, nullptr);
        tmp.assign(std::max(sum_len, full), 0);
        orc_poly_add(sum.data(), sum_len, prod.data(), full, tmp.data());
        sum = tmp; sum_len = (int)tmp.size();
      }
      std::vector<int32_t> blast(std::max(N, sum_len));
      orc_poly_add(last.data(), N, sum.data(), sum_len, blast.data());
      // only the first N coefficients are ever read by the path (numerics.rs:152)
      for (int x = 0; x < N; x++) smp[(size_t)k * N + x] = blast[x];
      bk_var[((size_t)i * l + 0) * (k + 1) + j] = alpha * alpha;
    }
    // tgsw.rs:129-167 add_mu_int_h with h = [l, 0, ..] (tgsw.rs:43-48):
    // coefficient 0 of every polynomial of every sample of row p gets
    // message*h[p]; only h[0] = l is non-zero.
    for (int p = 0; p < l; p++) {
      int32_t hp = (p == 0) ? l : 0;
      for (int j = 0; j <= k; j++)
        for (int u = 0; u <= k; u++) {
          int32_t* c0 = s + (((size_t)p * (k + 1) + j) * (k + 1) + u) * N;
          *c0 = wadd(*c0, wmul(lwe_key[i], hp));
        }
    }
  }
}

void orc_generate_synthetic_keys(const orc_params* P, uint64_t seed, int32_t* bk, double* bk_var,
                                 int32_t* ks_a, int32_t* ks_b, double* ks_var) {
  Rng rng(seed);
  const int base = 1 << P->ks_base_bit;
  size_t nbk = (size_t)P->n * bk_sample_stride(P);
  for (size_t i = 0; i < nbk; i++) bk[i] = rng.i32();
  size_t nv = (size_t)P->n * P->l * (P->k + 1);
  for (size_t i = 0; i < nv; i++) bk_var[i] = rng.unit() * 1e-9;
  size_t nks = (size_t)P->n_ks * P->ks_t * base;
  for (size_t i = 0; i < nks * P->n; i++) ks_a[i] = rng.i32();
  for (size_t i = 0; i < nks; i++) { ks_b[i] = rng.i32(); ks_var[i] = rng.unit() * 1e-9; }
}

void orc_encrypt(const orc_params* P, const int32_t* lwe_key, int message, uint64_t seed,
                 int32_t* a, int32_t* b, double* var) {
  Rng rng(seed);
  const int32_t e8 = orc_encode_message(1, 8);
  int32_t mu = message ? e8 : wneg(e8);           // encryption.rs:50-57
  double alpha = P->ks_stdev;                     // in_out_params.alpha_min
  int32_t bb = gaussian32(rng, mu, alpha);        // lwe.rs:319
  int32_t dot = 0;
  for (int i = 0; i < P->n; i++) {
    a[i] = rng.i32();
    dot = wadd(dot, wmul(a[i], lwe_key[i]));
  }
  *b = wadd(bb, dot);
  if (var) *var = alpha * alpha;
}

int32_t orc_lwe_phase(int32_t n, const int32_t* a, int32_t b, const int32_t* lwe_key) {
  int32_t axs = 0;                                 // lwe.rs:392-406
  for (int i = 0; i < n; i++) axs = wadd(axs, wmul(a[i], lwe_key[i]));
  return wsub(b, axs);
}

int orc_decrypt(int32_t n, const int32_t* a, int32_t b, const int32_t* lwe_key) {
  return orc_lwe_phase(n, a, b, lwe_key) > 0;      // encryption.rs:60-62
}

void orc_fill_uniform_i32(uint64_t seed, int32_t* out, size_t count) {
  Rng rng(seed);
  for (size_t i = 0; i < count; i++) out[i] = rng.i32();
}

void orc_fill_range_i32(uint64_t seed, int32_t lo, int32_t hi, int32_t* out, size_t count) {
  Rng rng(seed);
  uint64_t span = (uint64_t)((int64_t)hi - (int64_t)lo);
  for (size_t i = 0; i < count; i++) out[i] = (int32_t)((int64_t)lo + (int64_t)(rng.next() % span));
}

}  // extern "C"
