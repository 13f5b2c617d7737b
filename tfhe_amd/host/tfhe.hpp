// tfhe.hpp — host-side C++ mirror of the reference crate's public API for the
// bootstrapped-gate path, on top of the C ABI of include/tfhe_hip.h.
//
// Same names, argument meaning and error behaviour as the crate
// (IsakSundeSingh/tfhe; citations are file:line there):
//   Parameters::with / default        src/lwe.rs:163-241
//   generate_keys, encrypt, decrypt   src/encryption.rs:39-62
//   constant, nand, or, and, xor, xnor, not, nor, andny, andyn, orny, oryn, mux
//                                     src/gates.rs:12-268
//   tfhe_bootstrap                    src/bootstrapping.rs:25-29
// Key generation, encryption and decryption run on the host, as in the crate;
// every bootstrapped gate is one call into libtfhe_hip.so.  The crate panics on
// malformed input; here the same conditions throw std::runtime_error.
//
// Header-only; link with -ltfhe_hip.
#pragma once

#include <cmath>
#include <cstdint>
#include <memory>
#include <random>
#include <stdexcept>
#include <string>
#include <utility>
#include <vector>

#include "../../include/tfhe_hip.h"

namespace tfhe {

using Torus32 = int32_t;   // src/numerics.rs:15

enum class SecurityLevel { Bit80, Bit128 };   // src/encryption.rs:17-22

// src/numerics.rs:117-120
constexpr Torus32 encode_message(int32_t mu, int32_t message_space) {
  int log2 = 0;
  while (((message_space >> log2) & 1) == 0) log2++;
  return (Torus32)((uint32_t)mu << (32 - log2));
}

struct LweParams {            // src/lwe.rs:410-428
  int32_t n;
  double alpha_min, alpha_max;
};
struct TLweParameters {       // src/tlwe.rs:9-32
  int32_t n, k;
  double alpha_min, alpha_max;
  LweParams extracted_lweparams;
};
struct TGswParams {           // src/tgsw.rs:9-68
  int32_t l, bg_bit;
  TLweParameters tlwe_params;
};

struct Parameters {           // src/lwe.rs:155-241
  int32_t ks_t, ks_base_bit;
  LweParams in_out_params;
  TGswParams tgsw_params;

  static Parameters with(SecurityLevel level) {
    const int32_t N = 1024, K = 1;
    const double max_stdev = 0.012467;
    Parameters p;
    p.ks_base_bit = 2;
    p.ks_t = 8;
    int32_t n, l, bg_bit;
    double ks_stdev, bk_stdev;
    if (level == SecurityLevel::Bit80) {
      n = 500; l = 2; bg_bit = 10; ks_stdev = 2.44e-5; bk_stdev = 7.18e-9;
    } else {
      n = 630; l = 3; bg_bit = 7; ks_stdev = std::pow(2.0, -15.0); bk_stdev = std::pow(2.0, -15.0);
    }
    p.in_out_params = LweParams{n, ks_stdev, max_stdev};
    TLweParameters acc{N, K, bk_stdev, max_stdev, LweParams{N * K, bk_stdev, max_stdev}};
    p.tgsw_params = TGswParams{l, bg_bit, acc};
    return p;
  }
  // Parameters::default() is the 128-bit set (src/lwe.rs:235-241).
  static Parameters default_() { return with(SecurityLevel::Bit128); }
};

struct LweSample {            // src/lwe.rs:20-27
  std::vector<Torus32> coefficients;
  Torus32 b = 0;
  double current_variance = 0.0;
};

struct SecretKey {            // src/lwe.rs:264-274
  Parameters params;
  std::vector<int32_t> lwe_key;
};

namespace detail {

inline int32_t wadd(int32_t a, int32_t b) { return (int32_t)((uint32_t)a + (uint32_t)b); }
inline int32_t wsub(int32_t a, int32_t b) { return (int32_t)((uint32_t)a - (uint32_t)b); }
inline int32_t wmul(int32_t a, int32_t b) { return (int32_t)((uint32_t)a * (uint32_t)b); }

// src/numerics.rs:56-58 (Rust's saturating `as i32`)
inline Torus32 f64_to_torus_32(double d) {
  double t = std::trunc(d * 4294967296.0);
  if (t != t) return 0;
  if (t >= 2147483647.0) return INT32_MAX;
  if (t <= -2147483648.0) return INT32_MIN;
  return (Torus32)t;
}

struct Rng {
  std::mt19937_64 gen;
  explicit Rng(uint64_t seed) : gen(seed) {}
  int32_t torus() { return (int32_t)(gen() >> 32); }
  int32_t bit() { return (int32_t)(gen() >> 63); }
  double normal(double sigma) { return std::normal_distribution<double>(0.0, sigma)(gen); }
};

inline Rng& thread_rng() {      // the crate draws everything from rand::thread_rng()
  thread_local Rng rng(std::random_device{}());
  return rng;
}

// Owns the tfhe_hip context of a CloudKey.
struct Device {
  tfhe_hip_ctx* ctx = nullptr;
  explicit Device(int device) {
    if (tfhe_hip_create(&ctx, device) != 0)
      throw std::runtime_error("tfhe_hip_create failed: no usable GPU " + std::to_string(device));
  }
  // one context over several GPUs: batches are split over them, keys replicated once
  explicit Device(const std::vector<int>& devices) {
    if (tfhe_hip_create_multi(&ctx, devices.data(), (int)devices.size()) != 0)
      throw std::runtime_error("tfhe_hip_create_multi failed");
  }
  ~Device() { tfhe_hip_destroy(ctx); }
  Device(const Device&) = delete;
  Device& operator=(const Device&) = delete;
  void check(int rc) const {
    if (rc != 0) throw std::runtime_error(std::string("tfhe_hip: ") + tfhe_hip_last_error(ctx));
  }
};

}  // namespace detail

// LweKeySwitchKey (src/lwe.rs:495-514) with ks[i][j][h] flattened the way tfhe_hip_load_keys
// takes it.
struct LweKeySwitchKey {
  int32_t n = 0, t = 0, base_bit = 0;   // rows (the in/out n, src/lwe.rs:461), digits, log2(base)
  std::vector<int32_t> a;               // [n][t][base][n_out]  ks[i][j][h].coefficients
  std::vector<int32_t> b;               // [n][t][base]         ks[i][j][h].b
  std::vector<double> var;              // [n][t][base]         ks[i][j][h].current_variance
};

// LweBootstrappingKey (src/lwe.rs:432-446): the parameters the path reads, the TGSW samples
// flattened the way tfhe_hip_load_keys takes them, the key-switch key, and the GPU context
// that holds the uploaded key.  The context hangs here, not on CloudKey, because
// tfhe_bootstrap(bk: &LweBootstrappingKey, ..) (src/bootstrapping.rs:25) sees only this.
struct LweBootstrappingKey {
  LweParams in_out_params{};
  TGswParams bk_params{};
  std::vector<int32_t> bk;      // [n][l][k+1][k+1][N]
  LweKeySwitchKey ks;
  std::shared_ptr<detail::Device> device;

  tfhe_hip_params hip_params() const {
    const auto& g = bk_params;
    return tfhe_hip_params{in_out_params.n, g.tlwe_params.n, g.tlwe_params.k, g.l, g.bg_bit,
                           ks.t, ks.base_bit, ks.n};
  }
  // Uploads the key material to GPU `device_index` (done by generate_keys).
  void upload(int device_index = 0) {
    device = std::make_shared<detail::Device>(device_index);
    load();
  }
  // Uploads it to several GPUs: every batched call is then split over them.
  void upload(const std::vector<int>& devices) {
    device = std::make_shared<detail::Device>(devices);
    load();
  }
  const detail::Device& on_device() const {
    if (!device) throw std::runtime_error("LweBootstrappingKey has no device context");
    return *device;
  }

 private:
  void load() {
    tfhe_hip_params hp = hip_params();
    device->check(tfhe_hip_load_keys(device->ctx, &hp, bk.data(), ks.a.data(), ks.b.data(), ks.var.data()));
  }
};

// CloudKey (src/lwe.rs:247-257): the parameters and the bootstrapping key.
struct CloudKey {
  Parameters params;
  LweBootstrappingKey bk;

  static CloudKey with_params(const Parameters& params) {     // CloudKey::new + LweBootstrappingKey::new
    CloudKey ck;
    ck.params = params;
    ck.bk.in_out_params = params.in_out_params;
    ck.bk.bk_params = params.tgsw_params;
    ck.bk.ks.n = params.in_out_params.n;                      // src/lwe.rs:461
    ck.bk.ks.t = params.ks_t;
    ck.bk.ks.base_bit = params.ks_base_bit;
    return ck;
  }
  tfhe_hip_params hip_params() const { return bk.hip_params(); }
  void upload(int device_index = 0) { bk.upload(device_index); }
  void upload(const std::vector<int>& devices) { bk.upload(devices); }
  // The key file of tfhe_hip_key_file_pack (the crate derives Serialize for CloudKey,
  // src/lwe.rs:247-251, without fixing a byte format).
  std::vector<unsigned char> to_bytes() const {
    tfhe_hip_params hp = hip_params();
    size_t bytes = 0;
    if (tfhe_hip_key_file_size(&hp, 1, &bytes) != 0) throw std::runtime_error("tfhe_hip_key_file_size failed");
    std::vector<unsigned char> out(bytes);
    if (tfhe_hip_key_file_pack(&hp, bk.bk.data(), bk.ks.a.data(), bk.ks.b.data(), bk.ks.var.data(), out.data(),
                               bytes) != 0)
      throw std::runtime_error("tfhe_hip_key_file_pack failed");
    return out;
  }
  // A CloudKey for gates only (the flat key arrays stay empty) from a key file image.
  static CloudKey from_bytes(const Parameters& params, const std::vector<unsigned char>& image,
                             int device_index = 0) {
    CloudKey ck = with_params(params);
    ck.bk.device = std::make_shared<detail::Device>(device_index);
    ck.bk.device->check(tfhe_hip_load_keys_blob(ck.bk.device->ctx, image.data(), image.size()));
    return ck;
  }
};

// src/encryption.rs:39-47.  Restates LweKey::generate (src/lwe.rs:294-309),
// TGswKey::generate (src/tgsw.rs:81-89), LweBootstrappingKey::create
// (src/lwe.rs:474-492), LweKeySwitchKey::create (src/lwe.rs:530-558),
// TGswKey::encrypt (src/tgsw.rs:91-94,118-167) and TLweSample::encrypt_zero
// (src/tlwe.rs:112-134), including their observable quirks: only row 0 of each
// TGSW sample is encrypted, h = [l, 0, ..] and the second polynomial of a row-0
// sample is [s*l, 0, .., 0, g_0, ...] of which the first N coefficients matter.
inline std::pair<SecretKey, CloudKey> generate_keys(const Parameters& params, int device_index = 0) {
  using namespace detail;
  Rng& rng = thread_rng();
  const int32_t n = params.in_out_params.n, N = params.tgsw_params.tlwe_params.n;
  const int32_t k = params.tgsw_params.tlwe_params.k, l = params.tgsw_params.l;
  const int32_t t = params.ks_t, bb = params.ks_base_bit, base = 1 << bb;

  SecretKey sk{params, std::vector<int32_t>(n)};
  for (auto& v : sk.lwe_key) v = rng.bit();
  std::vector<int32_t> tlwe_key((size_t)(k + 1) * N);
  for (auto& v : tlwe_key) v = rng.bit();

  CloudKey ck = CloudKey::with_params(params);
  // key-switch key: rows for the first n coefficients of the extracted key (src/lwe.rs:461)
  const size_t rows = (size_t)n * t * base;
  ck.bk.ks.a.assign(rows * n, 0);
  ck.bk.ks.b.assign(rows, 0);
  ck.bk.ks.var.assign(rows, 0.0);
  {
    const double alpha = params.in_out_params.alpha_min;
    const int size_ks = n * t * (base - 1);
    std::vector<double> noise(size_ks);
    double sum = 0.0;
    for (auto& x : noise) { x = rng.normal(alpha); sum += x; }
    const double error = sum / (double)size_ks;
    int index = 0;
    for (int i = 0; i < n; i++)
      for (int j = 0; j < t; j++)
        for (int h = 1; h < base; h++) {
          const size_t idx = ((size_t)i * t + j) * base + h;
          const Torus32 message = wmul(tlwe_key[i] * h, (int32_t)(1u << (32 - (j + 1) * bb)));
          Torus32 b = wadd(message, f64_to_torus_32(noise[index++] - error));
          for (int x = 0; x < n; x++) {
            const int32_t a = rng.torus();
            ck.bk.ks.a[idx * n + x] = a;
            b = wadd(b, wmul(a, sk.lwe_key[x]));
          }
          ck.bk.ks.b[idx] = b;
          ck.bk.ks.var[idx] = alpha * alpha;
        }
  }
  // bootstrapping key
  const double alpha = params.tgsw_params.tlwe_params.alpha_min;
  ck.bk.bk.assign((size_t)n * l * (k + 1) * (k + 1) * N, 0);
  for (int i = 0; i < n; i++) {
    int32_t* s = ck.bk.bk.data() + (size_t)i * l * (k + 1) * (k + 1) * N;
    for (int j = 0; j <= k; j++) {                    // row 0 only (src/tgsw.rs:118-127)
      int32_t* smp = s + (size_t)j * (k + 1) * N;
      for (int u = 0; u < k; u++)
        for (int x = 0; x < N; x++) smp[(size_t)u * N + x] = rng.torus();
      // a_last_part + poly_sum is left-padded to 2N-1 entries: N-1 zeros, then the
      // gaussian terms; only index N-1 of those lands in the first N coefficients.
      std::vector<Torus32> g(N);
      for (auto& x : g) x = wadd(0, f64_to_torus_32(rng.normal(alpha)));
      smp[(size_t)k * N + (N - 1)] = g[0];
    }
    for (int j = 0; j <= k; j++)                      // add_mu_int_h with h[0] = l (src/tgsw.rs:43-48,129-167)
      for (int u = 0; u <= k; u++) {
        int32_t* c0 = s + ((size_t)j * (k + 1) + u) * N;
        *c0 = wadd(*c0, wmul(sk.lwe_key[i], l));
      }
  }
  ck.upload(device_index);
  return {std::move(sk), std::move(ck)};
}

// generate_keys with the randomness drawn on the GPU (tfhe_hip_generate_keys): the same key
// structure, every value a function of `seed` (32 bytes, the ChaCha20 key), so a seed
// reproduces its keys.  As with the crate's own generate_keys, the cloud key this returns
// carries the LWE secret key bits in the clear (coefficient 0 of the row-0 polynomials is
// key bit * l plus nothing, src/tgsw.rs:129-167 applied to an all-zero body): a parity and
// benchmark fixture, not key material to protect data with.
inline std::pair<SecretKey, CloudKey> generate_keys_gpu(const Parameters& params, const uint8_t seed[32],
                                                        int device_index = 0) {
  const int32_t n = params.in_out_params.n, N = params.tgsw_params.tlwe_params.n;
  const int32_t k = params.tgsw_params.tlwe_params.k, l = params.tgsw_params.l;
  const size_t rows = (size_t)n * params.ks_t * ((size_t)1 << params.ks_base_bit);
  SecretKey sk{params, std::vector<int32_t>(n)};
  std::vector<int32_t> tlwe_key((size_t)(k + 1) * N);
  CloudKey ck = CloudKey::with_params(params);
  ck.bk.bk.resize((size_t)n * l * (k + 1) * (k + 1) * N);
  ck.bk.ks.a.resize(rows * n);
  ck.bk.ks.b.resize(rows);
  ck.bk.ks.var.resize(rows);
  ck.bk.device = std::make_shared<detail::Device>(device_index);
  tfhe_hip_params hp = ck.hip_params();
  ck.bk.device->check(tfhe_hip_generate_keys(ck.bk.device->ctx, &hp, params.in_out_params.alpha_min,
                                             params.tgsw_params.tlwe_params.alpha_min, seed, sk.lwe_key.data(),
                                             tlwe_key.data(), ck.bk.bk.data(), ck.bk.ks.a.data(),
                                             ck.bk.ks.b.data(), ck.bk.ks.var.data()));
  return {std::move(sk), std::move(ck)};
}

// src/encryption.rs:50-57, src/lwe.rs:315-331
inline LweSample encrypt(bool message, const SecretKey& key) {
  using namespace detail;
  Rng& rng = thread_rng();
  const Torus32 e8 = encode_message(1, 8);
  const Torus32 mu = message ? e8 : -e8;
  const double alpha = key.params.in_out_params.alpha_min;
  LweSample s;
  s.coefficients.resize(key.lwe_key.size());
  s.b = wadd(mu, f64_to_torus_32(rng.normal(alpha)));
  for (size_t i = 0; i < s.coefficients.size(); i++) {
    s.coefficients[i] = rng.torus();
    s.b = wadd(s.b, wmul(s.coefficients[i], key.lwe_key[i]));
  }
  s.current_variance = alpha * alpha;
  return s;
}

// src/encryption.rs:60-62, src/lwe.rs:392-406
inline bool decrypt(const LweSample& sample, const SecretKey& key) {
  using namespace detail;
  if (sample.coefficients.size() != key.lwe_key.size())
    throw std::runtime_error("decrypt: sample and key sizes differ");   // zip_eq panics in the crate
  int32_t axs = 0;
  for (size_t i = 0; i < key.lwe_key.size(); i++) axs = wadd(axs, wmul(sample.coefficients[i], key.lwe_key[i]));
  return wsub(sample.b, axs) > 0;
}

// src/gates.rs:12-20
inline LweSample constant(bool value, const CloudKey& bk) {
  LweSample s;
  s.coefficients.assign(bk.params.in_out_params.n, 0);
  s.b = value ? encode_message(1, 8) : -encode_message(1, 8);
  return s;
}

// src/gates.rs:122-124, src/lwe.rs:138-151
inline LweSample not_(const LweSample& ca, const CloudKey&) {
  LweSample s = ca;
  for (auto& c : s.coefficients) c = (int32_t)(0u - (uint32_t)c);
  s.b = (int32_t)(0u - (uint32_t)s.b);
  return s;
}

namespace detail {

// Runs one gate id on batches (all samples must have n coefficients).
inline std::vector<LweSample> gate_batch(int gate, const std::vector<LweSample>& a,
                                         const std::vector<LweSample>& b,
                                         const std::vector<LweSample>* c, const CloudKey& bk) {
  const detail::Device& dev = bk.bk.on_device();
  const size_t B = a.size(), n = bk.params.in_out_params.n;
  if (b.size() != B || (c && c->size() != B)) throw std::runtime_error("gate: batch sizes differ");
  auto pack = [&](const std::vector<LweSample>& v, std::vector<int32_t>& m, std::vector<int32_t>& body) {
    m.resize(B * n);
    body.resize(B);
    for (size_t i = 0; i < B; i++) {
      if (v[i].coefficients.size() != n) throw std::runtime_error("gate: sample size differs from n");
      std::copy(v[i].coefficients.begin(), v[i].coefficients.end(), m.begin() + i * n);
      body[i] = v[i].b;
    }
  };
  std::vector<int32_t> aa, ab, ba, bb, ca, cb, oa(B * n), ob(B);
  std::vector<double> var(B);
  pack(a, aa, ab);
  pack(b, ba, bb);
  if (c) pack(*c, ca, cb);
  dev.check(tfhe_hip_gate(dev.ctx, gate, aa.data(), ab.data(), ba.data(), bb.data(),
                          c ? ca.data() : nullptr, c ? cb.data() : nullptr, oa.data(), ob.data(),
                          var.data(), B));
  std::vector<LweSample> out(B);
  for (size_t i = 0; i < B; i++) {
    out[i].coefficients.assign(oa.begin() + i * n, oa.begin() + (i + 1) * n);
    out[i].b = ob[i];
    out[i].current_variance = var[i];
  }
  return out;
}

inline LweSample gate1(int gate, const LweSample& a, const LweSample& b, const CloudKey& bk) {
  return gate_batch(gate, {a}, {b}, nullptr, bk)[0];
}

}  // namespace detail

// The ten bootstrapped binary gates (src/gates.rs:23-222) and mux (src/gates.rs:239-268).
#define TFHE_DEFINE_GATE(NAME, ID)                                                             \
  inline LweSample NAME(const LweSample& ca, const LweSample& cb, const CloudKey& bk) {        \
    return detail::gate1(ID, ca, cb, bk);                                                      \
  }                                                                                            \
  inline std::vector<LweSample> NAME(const std::vector<LweSample>& ca,                         \
                                     const std::vector<LweSample>& cb, const CloudKey& bk) {  \
    return detail::gate_batch(ID, ca, cb, nullptr, bk);                                        \
  }
TFHE_DEFINE_GATE(nand, TFHE_HIP_NAND)
TFHE_DEFINE_GATE(or_, TFHE_HIP_OR)
TFHE_DEFINE_GATE(and_, TFHE_HIP_AND)
TFHE_DEFINE_GATE(xor_, TFHE_HIP_XOR)
TFHE_DEFINE_GATE(xnor, TFHE_HIP_XNOR)
TFHE_DEFINE_GATE(nor, TFHE_HIP_NOR)
TFHE_DEFINE_GATE(andny, TFHE_HIP_ANDNY)
TFHE_DEFINE_GATE(andyn, TFHE_HIP_ANDYN)
TFHE_DEFINE_GATE(orny, TFHE_HIP_ORNY)
TFHE_DEFINE_GATE(oryn, TFHE_HIP_ORYN)
#undef TFHE_DEFINE_GATE

inline LweSample mux(const LweSample& a, const LweSample& b, const LweSample& c, const CloudKey& bk) {
  std::vector<LweSample> cc{c};
  return detail::gate_batch(TFHE_HIP_MUX, {a}, {b}, &cc, bk)[0];
}

// src/bootstrapping.rs:25-29: tfhe_bootstrap(bk: &LweBootstrappingKey, mu, x); the crate's
// callers pass &cloud_key.bk (src/gates.rs:35, benches/bootstrapping.rs:21).
inline LweSample tfhe_bootstrap(const LweBootstrappingKey& bk, Torus32 mu, const LweSample& x) {
  const detail::Device& dev = bk.on_device();
  const size_t n = bk.in_out_params.n;
  if (x.coefficients.size() != n) throw std::runtime_error("tfhe_bootstrap: sample size differs from n");
  LweSample out;
  out.coefficients.resize(n);
  dev.check(tfhe_hip_bootstrap(dev.ctx, mu, x.coefficients.data(), &x.b, out.coefficients.data(), &out.b,
                               &out.current_variance, 1));
  return out;
}

// src/circuits/circuits.rs.  Same names, argument order and panics (as
// std::runtime_error) as the crate; every gate runs on the GPU.  `add` also
// comes in a many-instances form that runs level by level on the device.
namespace circuits {

namespace detail {
// Bit-slices many operands: out_a [bits][count][n], out_b [bits][count].
inline void bit_slice(const std::vector<std::vector<LweSample>>& x, size_t bits, size_t n, const char* what,
                      std::vector<int32_t>& out_a, std::vector<int32_t>& out_b) {
  const size_t count = x.size();
  out_a.resize(bits * count * n);
  out_b.resize(bits * count);
  for (size_t c = 0; c < count; c++) {
    if (x[c].size() != bits) throw std::runtime_error(std::string(what) + ": operands differ in length");
    for (size_t i = 0; i < bits; i++) {
      if (x[c][i].coefficients.size() != n) throw std::runtime_error(std::string(what) + ": sample size differs from n");
      std::copy(x[c][i].coefficients.begin(), x[c][i].coefficients.end(), out_a.begin() + (i * count + c) * n);
      out_b[i * count + c] = x[c][i].b;
    }
  }
}

// compare / eq on many operand pairs through the level-batched device executors.
template <class Fn>
inline std::vector<LweSample> reduce(Fn fn, const char* what, const std::vector<std::vector<LweSample>>& a,
                                     const std::vector<std::vector<LweSample>>& b, const CloudKey& key) {
  const ::tfhe::detail::Device& dev = key.bk.on_device();
  if (a.size() != b.size()) throw std::runtime_error(std::string(what) + ": numbers of operands differ");
  const size_t count = a.size(), n = key.params.in_out_params.n;
  if (count == 0) return {};
  const size_t bits = a[0].size();
  std::vector<int32_t> aa, ab, ba, bb;
  bit_slice(a, bits, n, what, aa, ab);
  bit_slice(b, bits, n, what, ba, bb);
  std::vector<int32_t> oa(count * n), ob(count);
  std::vector<double> ov(count);
  dev.check(fn(dev.ctx, aa.data(), ab.data(), ba.data(), bb.data(), oa.data(), ob.data(),
                       ov.data(), count, (int)bits));
  std::vector<LweSample> out(count);
  for (size_t c = 0; c < count; c++) {
    out[c].coefficients.assign(oa.begin() + c * n, oa.begin() + (c + 1) * n);
    out[c].b = ob[c];
    out[c].current_variance = ov[c];
  }
  return out;
}
}  // namespace detail

// `a <= b` (circuits.rs:33-47) for many operand pairs at once: the xor of every bit of every
// pair is one launch, then one mux level per bit.
inline std::vector<LweSample> compare(const std::vector<std::vector<LweSample>>& a,
                                      const std::vector<std::vector<LweSample>>& b, const CloudKey& key) {
  return detail::reduce(tfhe_hip_circuit_compare, "compare", a, b, key);
}

// `a <= b` on bit vectors (circuits.rs:33-41).
inline LweSample compare(const std::vector<LweSample>& a, const std::vector<LweSample>& b,
                         const CloudKey& key) {
  if (a.size() != b.size()) throw std::runtime_error("compare: operands differ in length");
  return compare(std::vector<std::vector<LweSample>>{a}, std::vector<std::vector<LweSample>>{b}, key)[0];
}

// circuits.rs:65-69
inline LweSample eq_bit(const LweSample& a, const LweSample& b, const CloudKey& key) {
  const LweSample res = xor_(a, b, key);
  return xor_(res, constant(true, key), key);
}

// Bitwise equality (circuits.rs:50-62) for many operand pairs at once.
inline std::vector<LweSample> eq(const std::vector<std::vector<LweSample>>& a,
                                 const std::vector<std::vector<LweSample>>& b, const CloudKey& key) {
  if (!a.empty() && a[0].empty()) throw std::runtime_error("eq: empty operand");
  return detail::reduce(tfhe_hip_circuit_eq, "eq", a, b, key);
}

// circuits.rs:50-62
inline LweSample eq(const std::vector<LweSample>& a, const std::vector<LweSample>& b, const CloudKey& key) {
  if (a.empty()) throw std::runtime_error("eq: empty operand");
  if (a.size() != b.size()) throw std::runtime_error("eq: operands differ in length");
  return eq(std::vector<std::vector<LweSample>>{a}, std::vector<std::vector<LweSample>>{b}, key)[0];
}

// circuits.rs:77-79
inline std::pair<LweSample, LweSample> half_adder(const LweSample& a, const LweSample& b,
                                                  const CloudKey& key) {
  return {xor_(a, b, key), and_(a, b, key)};
}

// circuits.rs:83-97
inline std::pair<LweSample, LweSample> full_adder(const LweSample& a, const LweSample& b,
                                                  const LweSample& carry_in, const CloudKey& key) {
  const LweSample a_b_xor = xor_(a, b, key);
  const LweSample sum = xor_(a_b_xor, carry_in, key);
  const LweSample carry_in_and_a_xor_b = and_(carry_in, a_b_xor, key);
  const LweSample a_and_b = and_(a, b, key);
  return {sum, or_(carry_in_and_a_xor_b, a_and_b, key)};
}

// Ripple-carry adders of many independent operand pairs at once
// (circuits.rs:105-120 per pair): a[i], b[i] are the little-endian bit vectors
// of pair i, all of the same length.  Returns (sum bits, carry) per pair.
inline std::vector<std::pair<std::vector<LweSample>, LweSample>> add(
    const std::vector<std::vector<LweSample>>& a, const std::vector<std::vector<LweSample>>& b,
    const CloudKey& key) {
  const ::tfhe::detail::Device& dev = key.bk.on_device();
  if (a.size() != b.size()) throw std::runtime_error("add: numbers of operands differ");
  const size_t count = a.size(), n = key.params.in_out_params.n;
  if (count == 0) return {};
  const size_t bits = a[0].size();
  std::vector<int32_t> aa(bits * count * n), ab(bits * count), ba(aa.size()), bb(ab.size());
  for (size_t c = 0; c < count; c++) {
    if (a[c].size() != bits || b[c].size() != bits)
      throw std::runtime_error("Cannot add two numbers with different number of bits!");
    for (size_t i = 0; i < bits; i++) {
      if (a[c][i].coefficients.size() != n || b[c][i].coefficients.size() != n)
        throw std::runtime_error("add: sample size differs from n");
      std::copy(a[c][i].coefficients.begin(), a[c][i].coefficients.end(), aa.begin() + (i * count + c) * n);
      std::copy(b[c][i].coefficients.begin(), b[c][i].coefficients.end(), ba.begin() + (i * count + c) * n);
      ab[i * count + c] = a[c][i].b;
      bb[i * count + c] = b[c][i].b;
    }
  }
  std::vector<int32_t> sa(aa.size()), sb(ab.size()), ca(count * n), cb(count);
  std::vector<double> sv(ab.size()), cv(count);
  dev.check(tfhe_hip_circuit_add(dev.ctx, aa.data(), ab.data(), ba.data(), bb.data(),
                                         sa.data(), sb.data(), ca.data(), cb.data(), sv.data(), cv.data(),
                                         count, (int)bits));
  std::vector<std::pair<std::vector<LweSample>, LweSample>> out(count);
  for (size_t c = 0; c < count; c++) {
    out[c].first.resize(bits);
    for (size_t i = 0; i < bits; i++) {
      LweSample& s = out[c].first[i];
      s.coefficients.assign(sa.begin() + (i * count + c) * n, sa.begin() + (i * count + c + 1) * n);
      s.b = sb[i * count + c];
      s.current_variance = sv[i * count + c];
    }
    out[c].second.coefficients.assign(ca.begin() + c * n, ca.begin() + (c + 1) * n);
    out[c].second.b = cb[c];
    out[c].second.current_variance = bits ? cv[c] : 0.0;
  }
  return out;
}

// circuits.rs:105-120
inline std::pair<std::vector<LweSample>, LweSample> add(const std::vector<LweSample>& a,
                                                        const std::vector<LweSample>& b,
                                                        const CloudKey& key) {
  if (a.size() != b.size()) throw std::runtime_error("Cannot add two numbers with different number of bits!");
  return add(std::vector<std::vector<LweSample>>{a}, std::vector<std::vector<LweSample>>{b}, key)[0];
}

}  // namespace circuits

}  // namespace tfhe
