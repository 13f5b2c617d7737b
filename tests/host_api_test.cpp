// host_api_test.cpp — the C++ host API (tfhe_amd/host/tfhe.hpp, product code)
// against the CPU oracle (oracle/, the checker), bit for bit, on keys made by
// the host API's own generate_keys.  Run by tests/test_gpu_host_api.py on a GPU.
#include <cstdio>
#include <cstdlib>

#include "../oracle/oracle.h"
#include "../tfhe_amd/host/tfhe.hpp"

static int failures = 0;
#define CHECK(cond, ...)                                  \
  do {                                                    \
    if (!(cond)) {                                        \
      std::printf("FAIL %s:%d: ", __FILE__, __LINE__);    \
      std::printf(__VA_ARGS__);                           \
      std::printf("\n");                                  \
      failures++;                                         \
    }                                                     \
  } while (0)

int main() {
  using namespace tfhe;
  const Parameters params = Parameters::with(SecurityLevel::Bit80);
  auto [sk, ck] = generate_keys(params);

  orc_params P;
  orc_params_for(80, &P);
  const tfhe_hip_params hp = ck.hip_params();
  CHECK(hp.n == P.n && hp.N == P.N && hp.k == P.k && hp.l == P.l && hp.bg_bit == P.bg_bit &&
            hp.ks_t == P.ks_t && hp.ks_base_bit == P.ks_base_bit && hp.n_ks == P.n_ks,
        "parameter sets differ");

  // key structure the crate's generate_keys produces
  const size_t sample = (size_t)P.l * 4 * P.N;
  for (int i = 0; i < P.n; i++) {
    const int32_t* s = ck.bk.bk.data() + i * sample;
    for (size_t x = 2 * 2 * (size_t)P.N; x < sample; x++) CHECK(s[x] == 0, "row >= 1 not zero");
    CHECK(s[1 * P.N] == sk.lwe_key[i] * P.l, "b-poly coefficient 0");
    CHECK(s[3 * P.N] == sk.lwe_key[i] * P.l, "b-poly coefficient 0 (col 1)");
  }

  // encrypt / decrypt round trip (tests/bootstrapping_tests.rs:13-25)
  for (int m = 0; m < 2; m++) CHECK(decrypt(encrypt(m != 0, sk), sk) == (m != 0), "round trip %d", m);
  CHECK(decrypt(constant(true, ck), sk) && !decrypt(constant(false, ck), sk), "constant");

  struct G { const char* name; int orc; LweSample (*fn)(const LweSample&, const LweSample&, const CloudKey&); };
  const G gates[] = {
      {"nand", ORC_GATE_NAND, nand}, {"or", ORC_GATE_OR, or_}, {"and", ORC_GATE_AND, and_},
      {"xor", ORC_GATE_XOR, xor_}, {"xnor", ORC_GATE_XNOR, xnor}, {"nor", ORC_GATE_NOR, nor},
      {"andny", ORC_GATE_ANDNY, andny}, {"andyn", ORC_GATE_ANDYN, andyn},
      {"orny", ORC_GATE_ORNY, orny}, {"oryn", ORC_GATE_ORYN, oryn}};
  std::vector<int32_t> oa(P.n);
  int32_t ob;
  double ov;
  for (const G& g : gates)
    for (int bits = 0; bits < 4; bits++) {
      const LweSample a = encrypt(bits & 1, sk), b = encrypt(bits & 2, sk);
      const LweSample got = g.fn(a, b, ck);
      orc_gate(&P, g.orc, ck.bk.bk.data(), nullptr, ck.bk.ks.a.data(), ck.bk.ks.b.data(), ck.bk.ks.var.data(),
               a.coefficients.data(), a.b, a.current_variance, b.coefficients.data(), b.b,
               b.current_variance, a.coefficients.data(), 0, 0.0, oa.data(), &ob, &ov, ORC_MUL_FFT,
               nullptr);
      CHECK(got.coefficients == oa && got.b == ob && got.current_variance == ov, "%s(%d) differs", g.name, bits);
      CHECK(decrypt(got, sk) == (orc_decrypt(P.n, oa.data(), ob, sk.lwe_key.data()) != 0), "%s(%d) bit", g.name, bits);
    }
  {
    const LweSample a = encrypt(true, sk), b = encrypt(false, sk), c = encrypt(true, sk);
    const LweSample got = mux(a, b, c, ck);
    orc_gate(&P, ORC_GATE_MUX, ck.bk.bk.data(), nullptr, ck.bk.ks.a.data(), ck.bk.ks.b.data(), ck.bk.ks.var.data(),
             a.coefficients.data(), a.b, 0.0, b.coefficients.data(), b.b, 0.0, c.coefficients.data(),
             c.b, 0.0, oa.data(), &ob, &ov, ORC_MUL_FFT, nullptr);
    CHECK(got.coefficients == oa && got.b == ob && got.current_variance == ov, "mux differs");
    const LweSample bs = tfhe_bootstrap(ck.bk, encode_message(1, 8), a);
    orc_bootstrap(&P, ck.bk.bk.data(), nullptr, ck.bk.ks.a.data(), ck.bk.ks.b.data(), ck.bk.ks.var.data(),
                  encode_message(1, 8), a.coefficients.data(), a.b, oa.data(), &ob, &ov, ORC_MUL_FFT,
                  nullptr);
    CHECK(bs.coefficients == oa && bs.b == ob && bs.current_variance == ov, "tfhe_bootstrap differs");
    const LweSample na = not_(a, ck);
    CHECK(!decrypt(na, sk), "not");
  }
  // batch form
  {
    std::vector<LweSample> va, vb;
    for (int i = 0; i < 5; i++) { va.push_back(encrypt(i & 1, sk)); vb.push_back(encrypt(i & 2, sk)); }
    const auto out = nand(va, vb, ck);
    for (int i = 0; i < 5; i++) {
      const LweSample one = nand(va[i], vb[i], ck);
      CHECK(out[i].coefficients == one.coefficients && out[i].b == one.b, "batch nand %d", i);
    }
  }
  // circuits: the device executor of add against full_adder composed gate by gate
  {
    const int bits = 2;
    std::vector<LweSample> a, b;
    for (int i = 0; i < bits; i++) { a.push_back(encrypt(i & 1, sk)); b.push_back(encrypt(true, sk)); }
    const auto [sum, carry] = circuits::add(a, b, ck);
    LweSample c = constant(false, ck);
    for (int i = 0; i < bits; i++) {
      const auto [s1, c1] = circuits::full_adder(a[i], b[i], c, ck);
      CHECK(sum[i].coefficients == s1.coefficients && sum[i].b == s1.b &&
                sum[i].current_variance == s1.current_variance, "add: sum bit %d", i);
      c = c1;
    }
    CHECK(carry.coefficients == c.coefficients && carry.b == c.b && carry.current_variance == c.current_variance,
          "add: carry");
    const LweSample e = circuits::eq(a, b, ck);
    LweSample res = constant(true, ck);
    for (int i = 0; i < bits; i++) res = and_(res, circuits::eq_bit(a[i], b[i], ck), ck);
    CHECK(e.coefficients == res.coefficients && e.b == res.b, "eq");
    const LweSample le = circuits::compare(a, b, ck);
    LweSample rim = constant(true, ck);
    for (int i = 0; i < bits; i++) rim = mux(xor_(a[i], b[i], ck), b[i], rim, ck);
    CHECK(le.coefficients == rim.coefficients && le.b == rim.b, "compare");
    bool threw = false;
    try { (void)circuits::add(a, std::vector<LweSample>{b[0]}, ck); } catch (const std::runtime_error&) { threw = true; }
    CHECK(threw, "add: length mismatch not reported");
  }
  // error behaviour: malformed input is reported (the crate panics)
  {
    bool threw = false;
    LweSample bad = encrypt(true, sk);
    bad.coefficients.pop_back();
    try { (void)nand(bad, bad, ck); } catch (const std::runtime_error&) { threw = true; }
    CHECK(threw, "size mismatch not reported");
  }
  // keys drawn on the GPU: same structure, reproducible from the seed, gates match the oracle
  // run on the exported key material; key file and two-worker context give the same results
  // (bit-exact parity is the check: with the key structure the crate generates a bootstrapped
  // gate does not decrypt to its truth table, which is why the crate's CI runs with the
  // `bootstrapping` feature off, tests/bootstrapping_tests.rs:136-198)
  {
    uint8_t seed[32];
    for (int i = 0; i < 32; i++) seed[i] = (uint8_t)(7 * i + 3);
    auto [gsk, gck] = generate_keys_gpu(params, seed);
    auto [gsk2, gck2] = generate_keys_gpu(params, seed);
    CHECK(gsk.lwe_key == gsk2.lwe_key && gck.bk.bk == gck2.bk.bk && gck.bk.ks.a == gck2.bk.ks.a && gck.bk.ks.b == gck2.bk.ks.b,
          "generate_keys_gpu: a seed does not reproduce its keys");
    for (int i = 0; i < P.n; i++) {
      const int32_t* s = gck.bk.bk.data() + i * sample;
      for (size_t x = 2 * 2 * (size_t)P.N; x < sample; x++) CHECK(s[x] == 0, "gpu keys: row >= 1 not zero");
      CHECK(s[1 * P.N] == gsk.lwe_key[i] * P.l, "gpu keys: b-poly coefficient 0");
    }
    CloudKey from_file = CloudKey::from_bytes(params, gck.to_bytes());
    CloudKey two = gck;                // same key material on a context of two workers (one GPU twice)
    two.upload(std::vector<int>{0, 0});
    std::vector<LweSample> va, vb;
    for (int bits = 0; bits < 4; bits++) { va.push_back(encrypt(bits & 1, gsk)); vb.push_back(encrypt(bits & 2, gsk)); }
    const std::vector<LweSample> many = nand(va, vb, two);
    for (int bits = 0; bits < 4; bits++) {
      const LweSample &a = va[bits], &b = vb[bits];
      const LweSample got = nand(a, b, gck);
      orc_gate(&P, ORC_GATE_NAND, gck.bk.bk.data(), nullptr, gck.bk.ks.a.data(), gck.bk.ks.b.data(), gck.bk.ks.var.data(),
               a.coefficients.data(), a.b, a.current_variance, b.coefficients.data(), b.b,
               b.current_variance, a.coefficients.data(), 0, 0.0, oa.data(), &ob, &ov, ORC_MUL_FFT, nullptr);
      CHECK(got.coefficients == oa && got.b == ob && got.current_variance == ov, "gpu keys: nand vs oracle %d", bits);
      const LweSample f = nand(a, b, from_file);
      CHECK(f.coefficients == got.coefficients && f.b == got.b, "key file: nand %d", bits);
      CHECK(many[bits].coefficients == got.coefficients && many[bits].b == got.b, "two workers: nand %d", bits);
    }
  }
  std::printf(failures ? "host_api_test: %d FAILURES\n" : "host_api_test: ok\n", failures);
  return failures ? 1 : 0;
}
