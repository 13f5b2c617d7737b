// cpu_baseline.cpp — times the oracle's restatement of tfhe_bootstrap
// (benches/bootstrapping.rs:10-25: one bootstrap of a fresh encryption of
// `true`) on host cores.  TEST/BENCH INFRASTRUCTURE ONLY.
//
// The timed function follows the reference's cost structure: per polynomial
// product three zero-padded complex f64 FFTs and a fresh plan (twiddles) per
// call (src/numerics.rs:235-251), no caching of key spectra.
//
// usage: cpu_baseline <level 80|128> <threads> <gates_per_thread>
// prints one JSON line.
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <thread>
#include <vector>

#include "oracle.h"

int main(int argc, char** argv) {
  int level = argc > 1 ? atoi(argv[1]) : 80;
  int threads = argc > 2 ? atoi(argv[2]) : 1;
  int per_thread = argc > 3 ? atoi(argv[3]) : 1;
  orc_params P;
  if (orc_params_for(level, &P)) return 2;
  const int base = 1 << P.ks_base_bit;
  size_t nbk = (size_t)P.n * P.l * (P.k + 1) * (P.k + 1) * P.N;
  size_t nks = (size_t)P.n_ks * P.ks_t * base;
  std::vector<int32_t> lwe(P.n), tlwe((size_t)(P.k + 1) * P.N), bk(nbk), ks_a(nks * P.n), ks_b(nks);
  std::vector<double> bk_var((size_t)P.n * P.l * (P.k + 1)), ks_var(nks);
  orc_generate_keys(&P, 0xC0FFEEull, lwe.data(), tlwe.data(), bk.data(), bk_var.data(),
                    ks_a.data(), ks_b.data(), ks_var.data());
  std::vector<int> ok(threads, 0);
  auto t0 = std::chrono::steady_clock::now();
  std::vector<std::thread> pool;
  for (int t = 0; t < threads; t++)
    pool.emplace_back([&, t]() {
      std::vector<int32_t> a(P.n), oa(P.n);
      int32_t b, ob;
      double v;
      int good = 0;
      for (int g = 0; g < per_thread; g++) {
        orc_encrypt(&P, lwe.data(), 1, 1000 + 97 * t + g, a.data(), &b, &v);
        orc_bootstrap(&P, bk.data(), bk_var.data(), ks_a.data(), ks_b.data(), ks_var.data(),
                      orc_encode_message(1, 8), a.data(), b, oa.data(), &ob, &v,
                      ORC_MUL_FFT_PLAN_PER_CALL, nullptr);
        good += (ob == orc_encode_message(1, 8) || ob == -orc_encode_message(1, 8));
      }
      ok[t] = good;
    });
  for (auto& th : pool) th.join();
  double sec = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
  int total = threads * per_thread, good = 0;
  for (int g : ok) good += g;
  printf("{\"level\": %d, \"threads\": %d, \"gates\": %d, \"seconds\": %.6f, "
         "\"gates_per_sec\": %.6f, \"closed_form_ok\": %d}\n",
         level, threads, total, sec, total / sec, good);
  return good == total ? 0 : 1;
}
