/*
 * oracle.h — CPU restatement of the reference's bootstrapped-gate path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing in the shipped library (tfhe_amd/) may
 * include, link or call this.  It exists to check the HIP path and to time a
 * CPU baseline.
 *
 * The reference (IsakSundeSingh/tfhe, a Rust crate) cannot be compiled here
 * (no rustc/cargo), so this file restates its algorithm line by line; every
 * function cites the reference file:line it follows.  All integer arithmetic is
 * release-mode Rust: wrapping i32, arithmetic >>, saturating f64->i32 casts.
 *
 * Pinning: checked against the reference's own known-answer tests
 * (src/numerics.rs:499-515, src/polynomial.rs:320-364) and against an
 * independent numpy restatement (tests/golden/).  The reference has no test
 * that pins blind rotation, decomposition, external product, extraction or
 * key-switch values, so for those stages parity is pinned by source reading
 * only ("parity unpinned" by reference fixtures).
 *
 * The polynomial product lives in a third-party dependency of the reference,
 * rustfft ^5.0.0 (Cargo.toml:15, not vendored).  The product it computes is an
 * exactly-integer convolution rounded to the nearest integer, so any correct
 * f64 FFT with error < 0.5 gives the same i32; the oracle therefore offers an
 * exact-integer convolution (ground truth) and an f64 FFT restatement of
 * src/numerics.rs:185-261 that also reports the rounding margin.
 */
#ifndef TFHE_ORACLE_H
#define TFHE_ORACLE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct {
  int32_t n;           /* LWE dimension (in/out key)            src/lwe.rs:190,211 */
  int32_t N;           /* polynomial size                       src/lwe.rs:180 */
  int32_t k;           /* TLWE mask polynomials                 src/lwe.rs:181 */
  int32_t l;           /* TGSW decomposition length             src/lwe.rs:191,212 */
  int32_t bg_bit;      /* log2 of decomposition base            src/lwe.rs:192,213 */
  int32_t ks_t;        /* key-switch decomposition length       src/lwe.rs:186 */
  int32_t ks_base_bit; /* key-switch log2 base                  src/lwe.rs:185 */
  int32_t n_ks;        /* rows of the key-switch key (= n)      src/lwe.rs:461 */
  double ks_stdev;     /* in/out alpha_min                      src/lwe.rs:195,216 */
  double bk_stdev;     /* accumulator alpha_min                 src/lwe.rs:198,219 */
} orc_params;

enum { ORC_MUL_EXACT = 0, ORC_MUL_FFT = 1, ORC_MUL_FFT_PLAN_PER_CALL = 2 };

enum {
  ORC_GATE_NAND = 0, ORC_GATE_OR, ORC_GATE_AND, ORC_GATE_XOR, ORC_GATE_XNOR,
  ORC_GATE_NOR, ORC_GATE_ANDNY, ORC_GATE_ANDYN, ORC_GATE_ORNY, ORC_GATE_ORYN,
  ORC_GATE_MUX, ORC_GATE_NOT
};

/* Parameters::with(SecurityLevel)  src/lwe.rs:179-232.  level = 80 or 128. */
int orc_params_for(int level, orc_params* out);

/* src/numerics.rs:117-130 */
int32_t orc_encode_message(int32_t mu, int32_t message_space);
int32_t orc_decode_message(int32_t phase, int32_t message_space);
/* src/numerics.rs:56-58 (saturating cast) */
int32_t orc_f64_to_torus32(double d);

/* src/polynomial.rs:281-313 */
void orc_mul_by_monomial(const int32_t* p, int32_t n, int32_t power, int32_t* out);
/* src/polynomial.rs:66-90,118-173 (left-padded wrapping add / sub);
 * out has max(la, lb) entries. */
void orc_poly_add(const int32_t* a, int32_t la, const int32_t* b, int32_t lb, int32_t* out);
void orc_poly_sub(const int32_t* a, int32_t la, const int32_t* b, int32_t lb, int32_t* out);

/* src/numerics.rs:185-261: out has la+lb-1 entries.  mode ORC_MUL_EXACT uses
 * exact integer convolution then the same saturating cast; the FFT modes
 * restate the zero-padded complex f64 FFT and store in *max_frac (may be
 * NULL) the largest |x - round(x)| seen before rounding. */
void orc_poly_multiplier(const int32_t* a, int32_t la, const int32_t* b, int32_t lb,
                         int32_t* out, int mode, double* max_frac);
/* src/numerics.rs:138-155: result[t] += product[t] for t < la (wrapping). */
void orc_torus_polynomial_mul_r(int32_t* result, const int32_t* p1, int32_t la,
                                const int32_t* p2, int32_t lb, int mode, double* max_frac);

/* src/tgsw.rs:299-328 with the offset of src/tgsw.rs:50-52.  out is [l][N]. */
void orc_decompose(const int32_t* poly, int32_t N, int32_t l, int32_t bg_bit, int32_t* out);

/* src/tgsw.rs:229-264.  accum is [k+2][N]; bk_i is one TGSW sample laid out
 * [l][k+1][k+1][N] (row, column, polynomial; first N coefficients of each);
 * bk_i_var is [l][k+1] or NULL; result is [k+2][N]; *result_var (may be NULL)
 * receives the variance of src/tlwe.rs:156. */
void orc_extern_mul(const orc_params* P, const int32_t* accum, const int32_t* bk_i,
                    const double* bk_i_var, int32_t* result, double* result_var,
                    int mode, double* max_frac);

/* src/bootstrapping.rs:109-146.  acc is [k+2][N], updated in place.
 * bara has n entries. bk is [n][l][k+1][k+1][N]; bk_var [n][l][k+1] or NULL. */
void orc_blind_rotate(const orc_params* P, int32_t* acc, double* acc_var, const int32_t* bara,
                      const int32_t* bk, const double* bk_var, int mode, double* max_frac);

/* src/tlwe.rs:159-180.  acc is [k+2][N]; out_a has N*k entries. */
void orc_extract_lwe(const orc_params* P, const int32_t* acc, int32_t* out_a, int32_t* out_b);

/* src/bootstrapping.rs:38-99: out_a has N*k entries. */
void orc_bootstrap_noks(const orc_params* P, const int32_t* bk, const double* bk_var, int32_t mu,
                        const int32_t* a, int32_t b, int32_t* out_a, int32_t* out_b,
                        double* out_var, int mode, double* max_frac);

/* src/lwe.rs:601-648.  ks_a is [n_ks][t][base][n], ks_b is [n_ks][t][base],
 * ks_var likewise or NULL.  in_a needs at least n_ks entries.  *var_io (may be
 * NULL) is the running variance. */
void orc_key_switch(const orc_params* P, const int32_t* ks_a, const int32_t* ks_b,
                    const double* ks_var, const int32_t* in_a, int32_t in_b,
                    int32_t* out_a, int32_t* out_b, double* var_io);

/* src/bootstrapping.rs:25-29 */
void orc_bootstrap(const orc_params* P, const int32_t* bk, const double* bk_var,
                   const int32_t* ks_a, const int32_t* ks_b, const double* ks_var, int32_t mu,
                   const int32_t* a, int32_t b, int32_t* out_a, int32_t* out_b, double* out_var,
                   int mode, double* max_frac);

/* src/gates.rs:23-222: the linear prologue only (what the crate computes with
 * the `bootstrapping` feature off).  Inputs/outputs are n-coefficient samples.
 * Returns MU for the bootstrap in *mu. */
int orc_gate_prologue(int gate, int32_t n, const int32_t* ca_a, int32_t ca_b, double ca_var,
                      const int32_t* cb_a, int32_t cb_b, double cb_var, int32_t* x_a,
                      int32_t* x_b, double* x_var, int32_t* mu);

/* src/gates.rs:12-268 with the `bootstrapping` feature on.  cc_* only for MUX. */
int orc_gate(const orc_params* P, int gate, const int32_t* bk, const double* bk_var,
             const int32_t* ks_a, const int32_t* ks_b, const double* ks_var,
             const int32_t* ca_a, int32_t ca_b, double ca_var, const int32_t* cb_a, int32_t cb_b,
             double cb_var, const int32_t* cc_a, int32_t cc_b, double cc_var, int32_t* out_a,
             int32_t* out_b, double* out_var, int mode, double* max_frac);

/* ---- input generation (seeded; the reference uses thread_rng and cannot be
 * reproduced bit for bit, so only the distributions and structure match) ---- */

/* src/encryption.rs:39-47 and everything it calls (src/lwe.rs:294-309,474-558,
 * src/tgsw.rs:81-176, src/tlwe.rs:43-64,112-134).  Buffers:
 *   lwe_key [n]; tlwe_key [(k+1)][N]; bk [n][l][k+1][k+1][N]; bk_var [n][l][k+1];
 *   ks_a [n_ks][t][base][n]; ks_b [n_ks][t][base]; ks_var [n_ks][t][base]. */
void orc_generate_keys(const orc_params* P, uint64_t seed, int32_t* lwe_key, int32_t* tlwe_key,
                       int32_t* bk, double* bk_var, int32_t* ks_a, int32_t* ks_b, double* ks_var);
/* Same shapes, every entry uniform i32 (variances uniform in [0,1e-9)): keys
 * on which no stage of the path degenerates to zeros. */
void orc_generate_synthetic_keys(const orc_params* P, uint64_t seed, int32_t* bk, double* bk_var,
                                 int32_t* ks_a, int32_t* ks_b, double* ks_var);
/* src/encryption.rs:50-57, src/lwe.rs:315-331 */
void orc_encrypt(const orc_params* P, const int32_t* lwe_key, int message, uint64_t seed,
                 int32_t* a, int32_t* b, double* var);
/* src/lwe.rs:392-406 */
int32_t orc_lwe_phase(int32_t n, const int32_t* a, int32_t b, const int32_t* lwe_key);
/* src/encryption.rs:60-62 */
int orc_decrypt(int32_t n, const int32_t* a, int32_t b, const int32_t* lwe_key);

/* ---- the textbook-correct variant (textbook.cpp; checker of libtfhe_hip_textbook.so) ----
 * The same path with the crate's deviations from TFHE corrected (listed at the top of
 * textbook.cpp): acc and bk rows have k+1 polynomials (mask, body), products are negacyclic
 * and exact (wrapping integers), rotation amounts are in [0, 2N), P->n_ks = N k. */
void orc_tb_decompose(const int32_t* poly, int32_t N, int32_t l, int32_t bg_bit, int32_t* out);
void orc_tb_extern_product(const orc_params* P, const int32_t* acc, const int32_t* bk_i, int32_t* result);
void orc_tb_blind_rotate(const orc_params* P, int32_t* acc, const int32_t* bara, const int32_t* bk);
void orc_tb_bootstrap_noks(const orc_params* P, const int32_t* bk, int32_t mu, const int32_t* a, int32_t b,
                           int32_t* out_a, int32_t* out_b);
void orc_tb_bootstrap(const orc_params* P, const int32_t* bk, const int32_t* ks_a, const int32_t* ks_b,
                      const double* ks_var, int32_t mu, const int32_t* a, int32_t b, int32_t* out_a,
                      int32_t* out_b, double* out_var);
int orc_tb_gate(const orc_params* P, int gate, const int32_t* bk, const int32_t* ks_a, const int32_t* ks_b,
                const double* ks_var, const int32_t* ca_a, int32_t ca_b, const int32_t* cb_a, int32_t cb_b,
                const int32_t* cc_a, int32_t cc_b, int32_t* out_a, int32_t* out_b, double* out_var);
/* lwe_key [n]; tlwe_key [k][N]; bk [n][l][k+1][k+1][N]; ks_* with N k rows. */
void orc_tb_generate_keys(const orc_params* P, uint64_t seed, int32_t* lwe_key, int32_t* tlwe_key, int32_t* bk,
                          int32_t* ks_a, int32_t* ks_b, double* ks_var);

/* Deterministic fills used by tests and benches. */
void orc_fill_uniform_i32(uint64_t seed, int32_t* out, size_t count);
void orc_fill_range_i32(uint64_t seed, int32_t lo, int32_t hi /*exclusive*/, int32_t* out, size_t count);

#ifdef __cplusplus
}
#endif
#endif
