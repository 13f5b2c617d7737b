/*
 * tfhe_hip.h — C ABI of libtfhe_hip.so: the bootstrapped-gate path of the
 * TFHE crate IsakSundeSingh/tfhe on AMD MI355X (gfx950).
 *
 * The reference crate has no FFI of its own (it is `#![forbid(unsafe_code)]`,
 * src/lib.rs:25); these entry points are what a thin `extern "C"` module of
 * that crate would bind in place of the Rust functions cited on each one.
 * Citations are file:line in the reference crate.
 *
 * Conventions
 *  - Every function returns 0 on success and a negative code on failure; the
 *    message is available from tfhe_hip_last_error().  Nothing throws or aborts.
 *  - Torus32 values are int32_t with wrapping arithmetic (src/numerics.rs:15).
 *  - An LWE sample is its mask `a` (n int32) and body `b` (one int32); batches
 *    are row-major [B][n] and [B] (src/lwe.rs:20-27).
 *  - Entry points without a `_dev` suffix take HOST pointers, copy in and out,
 *    and return when the result is in the caller's buffers.  The `_dev`
 *    variants take DEVICE pointers on the context's GPU, enqueue on the
 *    context's stream and return without synchronising.
 *  - A context is bound to one GPU and is not re-entrant: one call at a time.
 *  - There is no CPU fallback: without a usable GPU tfhe_hip_create fails.
 */
#ifndef TFHE_HIP_H
#define TFHE_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct tfhe_hip_ctx tfhe_hip_ctx;

/* The integers of Parameters / TGswParams / LweKeySwitchKey that the path
 * reads (src/lwe.rs:155-161, src/tgsw.rs:9-36, src/lwe.rs:496-506).
 * Supported: N = 1024, k = 1, l in {2, 3}, bg_bit*l <= 32, ks_base_bit = 2,
 * n_ks <= N*k.  n_ks is the row count of the key-switch key, which the
 * reference builds with the in/out dimension n (src/lwe.rs:461). */
typedef struct {
  int32_t n, N, k, l, bg_bit, ks_t, ks_base_bit, n_ks;
} tfhe_hip_params;

enum {
  TFHE_HIP_OK = 0,
  TFHE_HIP_ERR_ARG = -1,      /* bad argument or unsupported parameter set */
  TFHE_HIP_ERR_STATE = -2,    /* keys not loaded */
  TFHE_HIP_ERR_RUNTIME = -3   /* a HIP call failed */
};

/* Gate ids; the weights and constants are those of src/gates.rs:23-268. */
enum {
  TFHE_HIP_NAND = 0, TFHE_HIP_OR, TFHE_HIP_AND, TFHE_HIP_XOR, TFHE_HIP_XNOR, TFHE_HIP_NOR,
  TFHE_HIP_ANDNY, TFHE_HIP_ANDYN, TFHE_HIP_ORNY, TFHE_HIP_ORYN, TFHE_HIP_MUX
};

/* Creates a context on GPU `device`.  Every entry point makes that device the
 * calling thread's current HIP device and leaves it current.  A context is not
 * re-entrant: one call at a time per context. */
int tfhe_hip_create(tfhe_hip_ctx** out, int device);
void tfhe_hip_destroy(tfhe_hip_ctx* ctx);
const char* tfhe_hip_last_error(tfhe_hip_ctx* ctx);

/* Work is enqueued on `hip_stream` (a hipStream_t; NULL = the default stream). */
int tfhe_hip_set_stream(tfhe_hip_ctx* ctx, void* hip_stream);
/* Waits for the context's stream. */
int tfhe_hip_synchronize(tfhe_hip_ctx* ctx);

/* How tfhe_blind_rotate (src/bootstrapping.rs:109-134) is mapped to the GPU.
 * BATCH: one ciphertext per wavefront, persistent workgroups (throughput form).
 * SPLIT: one ciphertext per workgroup, one wavefront per (decomposition level,
 *        key column) pair of tgsw_extern_mul_to_tlwe (src/tgsw.rs:229-264); a
 *        third of the latency, 4/3 of the arithmetic.
 * AUTO (default): SPLIT while the batch has at most two ciphertexts per
 *        compute unit, BATCH above.  Results are identical in every mode. */
enum { TFHE_HIP_BR_AUTO = 0, TFHE_HIP_BR_BATCH = 1, TFHE_HIP_BR_SPLIT = 2 };
int tfhe_hip_set_blind_rotate_mode(tfhe_hip_ctx* ctx, int mode);

/* Per-kernel timing with HIP events on the context's stream.  While enabled,
 * every launch of the gate-prologue, blind-rotate and key-switch kernels is
 * bracketed by two events.  tfhe_hip_profile_read waits for the stream and
 * returns, for those three kernels in that order, the summed duration in
 * milliseconds and the launch count since the last read, then forgets them.
 * (No counterpart in the reference: measurement support for benches.) */
int tfhe_hip_profile_enable(tfhe_hip_ctx* ctx, int on);
int tfhe_hip_profile_read(tfhe_hip_ctx* ctx, double ms_sum[3], int launches[3]);

/* Replaces the key material a CloudKey carries (LweBootstrappingKey,
 * src/lwe.rs:432-446).  Host pointers:
 *   bk     [n][l][k+1][k+1][N]  bk[i].all_sample[row][col].a[poly], first N
 *                               coefficients of each polynomial (only those
 *                               can reach the result, src/numerics.rs:152)
 *   ks_a   [n_ks][t][base][n]   ks.ks[i][j][h].coefficients
 *   ks_b   [n_ks][t][base]      ks.ks[i][j][h].b
 *   ks_var [n_ks][t][base]      ks.ks[i][j][h].current_variance, or NULL (zeros)
 * Transforms the bootstrapping key to its FFT-domain form on the GPU. */
int tfhe_hip_load_keys(tfhe_hip_ctx* ctx, const tfhe_hip_params* params, const int32_t* bk,
                       const int32_t* ks_a, const int32_t* ks_b, const double* ks_var);

/* Sizes in bytes of the two device-side key blobs (FFT-domain bootstrapping
 * key; key-switch rows followed by their variances). */
int tfhe_hip_key_blob_sizes(const tfhe_hip_params* params, size_t* bk_bytes, size_t* ks_bytes);
/* Copies the loaded blobs into caller-owned DEVICE buffers (e.g. to broadcast
 * them to other ranks), and the reverse: adopts blobs produced by export on
 * another context with the same parameters. */
int tfhe_hip_export_keys_dev(tfhe_hip_ctx* ctx, void* bk_blob_dev, void* ks_blob_dev);
int tfhe_hip_import_keys_dev(tfhe_hip_ctx* ctx, const tfhe_hip_params* params,
                             const void* bk_blob_dev, const void* ks_blob_dev);

/* tfhe_bootstrap (src/bootstrapping.rs:25-29): out = KeySwitch(BlindRotate(x)).
 * a [B][n], b [B] -> out_a [B][n], out_b [B]; var_out [B] or NULL receives
 * LweSample.current_variance of each result. */
int tfhe_hip_bootstrap(tfhe_hip_ctx* ctx, int32_t mu, const int32_t* a, const int32_t* b,
                       int32_t* out_a, int32_t* out_b, double* var_out, size_t B);
int tfhe_hip_bootstrap_dev(tfhe_hip_ctx* ctx, int32_t mu, const int32_t* a, const int32_t* b,
                           int32_t* out_a, int32_t* out_b, double* var_out, size_t B);

/* tfhe_bootstrap_without_key_switching (src/bootstrapping.rs:38-65):
 * out_a [B][N*k], out_b [B]. */
int tfhe_hip_bootstrap_noks(tfhe_hip_ctx* ctx, int32_t mu, const int32_t* a, const int32_t* b,
                            int32_t* out_a, int32_t* out_b, size_t B);
int tfhe_hip_bootstrap_noks_dev(tfhe_hip_ctx* ctx, int32_t mu, const int32_t* a, const int32_t* b,
                                int32_t* out_a, int32_t* out_b, size_t B);

/* lwe_key_switch (src/lwe.rs:601-648): a [B][N*k], b [B] -> out_a [B][n], out_b [B]. */
int tfhe_hip_keyswitch(tfhe_hip_ctx* ctx, const int32_t* a, const int32_t* b, int32_t* out_a,
                       int32_t* out_b, double* var_out, size_t B);
int tfhe_hip_keyswitch_dev(tfhe_hip_ctx* ctx, const int32_t* a, const int32_t* b, int32_t* out_a,
                           int32_t* out_b, double* var_out, size_t B);

/* The bootstrapped gates nand/or/and/xor/xnor/nor/andny/andyn/orny/oryn
 * (src/gates.rs:23-222) and mux (src/gates.rs:239-268; cc_* are its third
 * operand, NULL otherwise).  All samples [B][n] / [B]. */
int tfhe_hip_gate(tfhe_hip_ctx* ctx, int gate, const int32_t* ca_a, const int32_t* ca_b,
                  const int32_t* cb_a, const int32_t* cb_b, const int32_t* cc_a,
                  const int32_t* cc_b, int32_t* out_a, int32_t* out_b, double* var_out, size_t B);
int tfhe_hip_gate_dev(tfhe_hip_ctx* ctx, int gate, const int32_t* ca_a, const int32_t* ca_b,
                      const int32_t* cb_a, const int32_t* cb_b, const int32_t* cc_a,
                      const int32_t* cc_b, int32_t* out_a, int32_t* out_b, double* var_out,
                      size_t B);

/* circuits::add (src/circuits/circuits.rs:105-120, full_adder :83-97), the
 * ripple-carry adder, on `count` independent pairs of `bits`-bit operands, run
 * level by level so that every launch holds the gates of all instances that are
 * ready: xor(a, b) and and(a, b) of all bits first, then per bit
 * {xor(x, carry), and(carry, x)} as one batch and or(t, g) as the next.
 * Intermediate ciphertexts never leave the GPU.
 * Operands and sums are bit-sliced: a_a, b_a, sum_a [bits][count][n]; a_b, b_b,
 * sum_b [bits][count] (bit 0 = least significant, the first element of the
 * reference's slices); carry_a [count][n], carry_b [count].  sum_var
 * [bits][count] and carry_var [count] may be NULL.  bits = 0 returns
 * carry = constant(false), as the reference's fold does. */
int tfhe_hip_circuit_add(tfhe_hip_ctx* ctx, const int32_t* a_a, const int32_t* a_b,
                         const int32_t* b_a, const int32_t* b_b, int32_t* sum_a, int32_t* sum_b,
                         int32_t* carry_a, int32_t* carry_b, double* sum_var, double* carry_var,
                         size_t count, int bits);
int tfhe_hip_circuit_add_dev(tfhe_hip_ctx* ctx, const int32_t* a_a, const int32_t* a_b,
                             const int32_t* b_a, const int32_t* b_b, int32_t* sum_a, int32_t* sum_b,
                             int32_t* carry_a, int32_t* carry_b, double* sum_var, double* carry_var,
                             size_t count, int bits);

/* poly_multiplier / torus_polynomial_mul_r (src/numerics.rs:138-155,185-261)
 * on B independent pairs of N-coefficient polynomials d [B][N], s [B][N].
 * accumulate == 0: out [B][2N-1] = the full saturated product;
 * accumulate != 0: out [B][N] += first N coefficients (wrapping). */
int tfhe_hip_polymul(tfhe_hip_ctx* ctx, const int32_t* d, const int32_t* s, int32_t* out,
                     int accumulate, size_t B);
int tfhe_hip_polymul_dev(tfhe_hip_ctx* ctx, const int32_t* d, const int32_t* s, int32_t* out,
                         int accumulate, size_t B);

/* tgsw_extern_mul_to_tlwe (src/tgsw.rs:229-264) of B accumulators
 * t [B][k+2][N] with the loaded bk[bk_index]; out [B][k+2][N]. */
int tfhe_hip_extprod(tfhe_hip_ctx* ctx, const int32_t* t, int bk_index, int32_t* out, size_t B);

/* tfhe_blind_rotate (src/bootstrapping.rs:109-134) of B accumulators
 * acc [B][k+2][N] (updated in place) by bara [B][n] with the loaded key. */
int tfhe_hip_blind_rotate(tfhe_hip_ctx* ctx, int32_t* acc, const int32_t* bara, size_t B);

#ifdef __cplusplus
}
#endif
#endif
