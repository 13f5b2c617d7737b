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
 *  - A context is bound to one GPU (tfhe_hip_create) or to a list of GPUs
 *    (tfhe_hip_create_multi) and is not re-entrant: one call at a time.
 *  - Host-pointer entry points stage through buffers the context keeps (a device
 *    buffer and a pinned host buffer, grown on demand, freed by tfhe_hip_destroy):
 *    after the first call of a given size they allocate nothing.
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

/* Which build of the library this is.
 * "reference": libtfhe_hip.so, the drop-in for the crate's bootstrapped path; its results are the
 *   crate's, rule for rule (what this header documents).
 * "textbook": libtfhe_hip_textbook.so, built from the same sources with -DTFHE_TEXTBOOK: the
 *   same entry points with the crate's deviations from TFHE corrected, so that bootstrapped
 *   gates decrypt to their truth tables.  Differences a caller sees: keys have the textbook
 *   structure (every TGSW row an encryption of zero with k mask polynomials and a body, gadget
 *   2^(32 - (p+1) bg_bit) on every level; src/tgsw.rs:43-52,118-176, src/tlwe.rs:86,112-134 are
 *   where the crate departs), bk [n][l][k+1][k+1][N] is [i][level][component][polynomial],
 *   n_ks must be N k (src/lwe.rs:458-461 builds n rows), products are negacyclic and exact
 *   (src/numerics.rs:138-155,185-261 computes a saturated plain convolution), rotation amounts
 *   are in [0, 2N) (src/bootstrapping.rs:50-55), NOR bootstraps with MU = +1/8
 *   (src/gates.rs:136 uses -1/8, which inverts it).  tfhe_hip_extprod*, tfhe_hip_blind_rotate* (the
 *   parity hooks of the reference's intermediate functions) are not part of it, and the
 *   blind-rotate mode is ignored (one mapping).
 * No counterpart in the reference. */
const char* tfhe_hip_variant(void);

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
/* A context over `ndev` GPUs (devices[i] may repeat: each entry is one worker
 * with a stream of its own).  The reference's callers are one single-threaded
 * process (src/gates.rs:35, src/bootstrapping.rs:25-29, benches/bootstrapping.rs:21);
 * through this context that one process drives every listed GPU:
 *   - tfhe_hip_load_keys / tfhe_hip_load_keys_blob transform the keys on
 *     devices[0] and replicate the device blobs once: an RCCL broadcast
 *     (ncclCommInitAll + ncclBroadcast over xGMI) between distinct GPUs,
 *     peer-to-peer copies if RCCL is unavailable or fails, a device-to-device
 *     copy between workers that share a GPU (tfhe_hip_key_transport says which);
 *   - every host-pointer entry point and every tfhe_hip_samples_* call splits its
 *     batch contiguously over the workers (tfhe_hip_circuit_add by instance, so
 *     that all dependency levels of an instance stay on one GPU), runs the
 *     shards at the same time and joins them; there is no data-path collective;
 *   - the parity hooks (tfhe_hip_extprod*, tfhe_hip_blind_rotate,
 *     tfhe_hip_polymul_margin) run on devices[0];
 *   - the `_dev` entry points, tfhe_hip_set_stream and the key export/import
 *     need a single-device context and fail with TFHE_HIP_ERR_ARG here.
 * Results are bit-identical to those of a single-device context. */
int tfhe_hip_create_multi(tfhe_hip_ctx** out, const int* devices, int ndev);
/* How the loaded keys reached the other GPUs of a multi-GPU context: "rccl" (one
 * broadcast), "peer" (hipMemcpyPeerAsync from devices[0], peer access enabled where the
 * hardware offers it), "local" (all workers share one GPU: device-to-device copies), "none"
 * (single-device context, or no keys loaded).  No counterpart in the reference. */
const char* tfhe_hip_key_transport(tfhe_hip_ctx* ctx);
/* Releases the context and everything it owns on its GPU(s), including the device memory of
 * sample handles that are still alive: such a handle stays valid only as an argument of
 * tfhe_hip_samples_free. */
void tfhe_hip_destroy(tfhe_hip_ctx* ctx);
const char* tfhe_hip_last_error(tfhe_hip_ctx* ctx);

/* Work is enqueued on `hip_stream` (a hipStream_t; NULL = the default stream).
 * Waits for the work already enqueued on the previous stream first: the
 * context's scratch buffers are shared by all calls. */
int tfhe_hip_set_stream(tfhe_hip_ctx* ctx, void* hip_stream);
/* Waits for the context's stream. */
int tfhe_hip_synchronize(tfhe_hip_ctx* ctx);

/* How tfhe_blind_rotate (src/bootstrapping.rs:109-134) is mapped to the GPU.
 * BATCH:   one ciphertext per wavefront, one wavefront per SIMD; a transform takes a
 *          half-wave, the two digit polynomials of a level run side by side.
 * BATCH64: one ciphertext per wavefront, two wavefronts per SIMD; a transform takes
 *          the whole wavefront, the two digit polynomials run one after the other.
 *          Both are throughput forms with persistent workgroups.
 * SPLIT:   one ciphertext per workgroup, one wavefront per (digit polynomial, key
 *          polynomial) product of tgsw_extern_mul_to_tlwe (src/tgsw.rs:229-264): 4 l
 *          wavefronts, the two that share a digit polynomial's spectrum pass it through
 *          LDS; under a third of the latency.
 * AUTO (default): whole rounds of BATCH64 (one ciphertext per wavefront slot, the
 *          fastest throughput form on MI355X for both parameter sets) and, for what is
 *          left over, the cheapest of one more BATCH64 round, one BATCH round, SPLIT
 *          rounds, or one BATCH round followed by SPLIT rounds, by measured cost per
 *          round; the parts run one after another on the context's stream.  On 256
 *          compute units: up to 768 gates SPLIT, up to 1024 BATCH, up to 1536 BATCH
 *          then SPLIT, up to 2048 BATCH64.
 * Results are identical in every mode. */
enum { TFHE_HIP_BR_AUTO = 0, TFHE_HIP_BR_BATCH = 1, TFHE_HIP_BR_SPLIT = 2, TFHE_HIP_BR_BATCH64 = 3 };
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

/* generate_keys on the GPU: LweKey::generate (src/lwe.rs:294-309),
 * TGswKey::generate (src/tgsw.rs:81-89, src/tlwe.rs:43-64) and
 * LweBootstrappingKey::create (src/lwe.rs:474-493: LweKey::extract,
 * LweKeySwitchKey::create :530-558 with its recentred Gaussian noises, and one
 * TGswKey::encrypt per key bit, src/tgsw.rs:91-176, src/tlwe.rs:112-134), with the
 * key structure the reference's code produces (only decomposition row 0 of a
 * TGSW sample is encrypted; its body keeps one Gaussian coefficient among its
 * first N; the gadget adds key bit * l to coefficient 0 of every polynomial of row
 * 0; key-switch samples of digit 0 are zero).  ks_stdev / bk_stdev are
 * alpha_min of the in/out and of the TLWE parameters (Parameters::with,
 * src/lwe.rs:179-232).
 * The reference draws from rand's thread_rng (an OS-seeded ChaCha generator that
 * callers cannot seed); here every random value is a documented function of
 * `seed` (32 bytes = the ChaCha20 key, RFC 8439 block function; the addressing
 * is described at the kernels), so a seed reproduces its keys.
 * NOT SECURE KEY MATERIAL: as in the reference, the gadget term lands on an unencrypted
 * coefficient (the body polynomial of a row-0 sample is [key bit * l, 0 .. 0, noise]), so
 * bk_out and the device-side cloud key expose every LWE secret key bit in the clear.  The
 * output reproduces the reference's key structure for parity tests and benchmarks; it must
 * not protect real data.  The device copies of the secret keys and noises are zeroed
 * before the call returns.
 * On return the context holds the keys exactly as after tfhe_hip_load_keys (on a
 * multi-GPU context: generated on devices[0], then replicated) and the SECRET
 * keys are in lwe_key [n] and tlwe_key [(k+1)][N] (may be NULL).  bk_out, ks_a_out,
 * ks_b_out, ks_var_out (each may be NULL) receive the cloud key in the layouts
 * of tfhe_hip_load_keys, e.g. for tfhe_hip_key_file_pack. */
int tfhe_hip_generate_keys(tfhe_hip_ctx* ctx, const tfhe_hip_params* params, double ks_stdev,
                           double bk_stdev, const uint8_t seed[32], int32_t* lwe_key,
                           int32_t* tlwe_key, int32_t* bk_out, int32_t* ks_a_out, int32_t* ks_b_out,
                           double* ks_var_out);
/* The keystream behind tfhe_hip_generate_keys (test hook): out [nblocks][16] =
 * ChaCha20 blocks first_block .. of (seed, purpose, stream), where the block's
 * state words 12..15 are block counter, purpose, stream low, stream high. */
int tfhe_hip_keygen_stream(tfhe_hip_ctx* ctx, const uint8_t seed[32], uint32_t purpose,
                           uint64_t stream, uint32_t first_block, uint32_t* out, size_t nblocks);

/* The key file: one flat little-endian image of the key material above, so that
 * keys are generated once and loaded from disk (the reference derives
 * Serialize/Deserialize for CloudKey, LweBootstrappingKey and LweKeySwitchKey,
 * src/lwe.rs:247-251,432-446,495-514, but depends on no serializer, so it
 * defines no byte format of its own).  Layout:
 *    0  char[8]  "TFHEHIPK"
 *    8  u32      version (1)          12  u32  header size (64)
 *   16  i32[8]   n, N, k, l, bg_bit, ks_t, ks_base_bit, n_ks
 *   48  u32      1 if the variances are present      52  u32  0
 *   56  u64      bytes that follow the header
 *   64  i32      bk [n][l][k+1][k+1][N], then ks_a [n_ks][t][base][n], then
 *                ks_b [n_ks][t][base]; zero padding to a multiple of 8; then
 *                f64 ks_var [n_ks][t][base] if present
 * i.e. the Vec<i32> payloads of the serde field order without length prefixes.
 * tfhe_hip_key_file_size: size of the image; tfhe_hip_key_file_pack: writes it
 * into `out` (capacity bytes; ks_var may be NULL); tfhe_hip_load_keys_blob:
 * validates an image and loads it as tfhe_hip_load_keys does. */
int tfhe_hip_key_file_size(const tfhe_hip_params* params, int has_ks_var, size_t* bytes);
int tfhe_hip_key_file_pack(const tfhe_hip_params* params, const int32_t* bk, const int32_t* ks_a,
                           const int32_t* ks_b, const double* ks_var, void* out, size_t capacity);
int tfhe_hip_load_keys_blob(tfhe_hip_ctx* ctx, const void* blob, size_t bytes);

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

/* tfhe_hip_gate_dev for callers that cannot allocate (code under HIP graph
 * capture, tight serving loops, harnesses that forbid hipMalloc).  No
 * counterpart in the reference.  Same arguments and results as
 * tfhe_hip_gate_dev, but the scratch of the call (mod-switched operands,
 * extracted samples: about 4 n + 4 224 bytes per ciphertext, twice that per mux
 * gate) lives in a fixed 4 MiB array inside the library instead of the context's
 * grown-on-demand pool, and nothing is allocated or freed: no hipMalloc, no
 * hipFree, no events.  Limits: the batch must fit the array (668 gates or 334
 * mux gates at n = 500; 618 / 309 at n = 630), larger batches fail with
 * TFHE_HIP_ERR_ARG and a message that states the limit; keys must be loaded
 * before.  Every GPU the library runs on holds one copy of the array until the
 * process ends, and every call on that GPU (from any context) uses it: calls that
 * overlap on one GPU race.  Nothing in it needs to start at zero (each call
 * writes every entry before reading it). */
int tfhe_hip_gate_dev_noalloc(tfhe_hip_ctx* ctx, int gate, const int32_t* ca_a, const int32_t* ca_b,
                              const int32_t* cb_a, const int32_t* cb_b, const int32_t* cc_a,
                              const int32_t* cc_b, int32_t* out_a, int32_t* out_b, double* var_out,
                              size_t B);

/* gates::not (src/gates.rs:122-124: mask and body negated, variance kept; `var`
 * may be NULL = zeros) and gates::constant (src/gates.rs:12-20: the noiseless
 * trivial sample of +-1/8) on device buffers. */
int tfhe_hip_not_dev(tfhe_hip_ctx* ctx, const int32_t* a, const int32_t* b, const double* var,
                     int32_t* out_a, int32_t* out_b, double* var_out, size_t B);
int tfhe_hip_constant_dev(tfhe_hip_ctx* ctx, int value, int32_t* out_a, int32_t* out_b,
                          double* var_out, size_t B);

/* Device-resident LweSample batches (src/lwe.rs:20-27: coefficients, b,
 * current_variance), so that chained gates never return to the host: a handle
 * owns a [B][n] mask, [B] body and [B] variance array on the context's GPU (on
 * a multi-GPU context: one contiguous share per worker).  Handles are made for
 * the loaded keys' n; upload and download copy whole batches and return when the
 * caller's buffers may be reused / are filled; the gate calls only enqueue.
 * var may be NULL (upload: zeros; download: skipped). */
typedef struct tfhe_hip_samples tfhe_hip_samples;
int tfhe_hip_samples_alloc(tfhe_hip_ctx* ctx, size_t B, tfhe_hip_samples** out);
void tfhe_hip_samples_free(tfhe_hip_samples* s);
size_t tfhe_hip_samples_count(const tfhe_hip_samples* s);
int tfhe_hip_samples_upload(tfhe_hip_ctx* ctx, tfhe_hip_samples* s, const int32_t* a,
                            const int32_t* b, const double* var);
int tfhe_hip_samples_download(tfhe_hip_ctx* ctx, const tfhe_hip_samples* s, int32_t* a, int32_t* b,
                              double* var);
/* out = gate(ca, cb[, cc]) as tfhe_hip_gate; out may not alias an operand. */
int tfhe_hip_samples_gate(tfhe_hip_ctx* ctx, int gate, const tfhe_hip_samples* ca,
                          const tfhe_hip_samples* cb, const tfhe_hip_samples* cc,
                          tfhe_hip_samples* out);
int tfhe_hip_samples_bootstrap(tfhe_hip_ctx* ctx, int32_t mu, const tfhe_hip_samples* x,
                               tfhe_hip_samples* out);
int tfhe_hip_samples_not(tfhe_hip_ctx* ctx, const tfhe_hip_samples* x, tfhe_hip_samples* out);
int tfhe_hip_samples_constant(tfhe_hip_ctx* ctx, int value, tfhe_hip_samples* out);

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

/* circuits::compare (src/circuits/circuits.rs:33-47, a <= b, the first element
 * of the slices compared first) and circuits::eq (src/circuits/circuits.rs:50-70)
 * on `count` independent pairs of `bits`-bit operands, level by level like
 * tfhe_hip_circuit_add: the xor of every bit of every instance is one blind
 * rotation, then the carry chain (one mux per bit) or the and-fold runs one
 * level of `count` gates per bit; intermediate ciphertexts stay on the GPU.
 * Operands bit-sliced as for tfhe_hip_circuit_add; out_a [count][n], out_b
 * [count], out_var [count] or NULL.  compare with bits = 0 returns
 * constant(true), as the reference's loop does; eq needs bits >= 1 (the
 * reference asserts that its operands are not empty). */
int tfhe_hip_circuit_compare(tfhe_hip_ctx* ctx, const int32_t* a_a, const int32_t* a_b,
                             const int32_t* b_a, const int32_t* b_b, int32_t* out_a, int32_t* out_b,
                             double* out_var, size_t count, int bits);
int tfhe_hip_circuit_compare_dev(tfhe_hip_ctx* ctx, const int32_t* a_a, const int32_t* a_b,
                                 const int32_t* b_a, const int32_t* b_b, int32_t* out_a,
                                 int32_t* out_b, double* out_var, size_t count, int bits);
int tfhe_hip_circuit_eq(tfhe_hip_ctx* ctx, const int32_t* a_a, const int32_t* a_b, const int32_t* b_a,
                        const int32_t* b_b, int32_t* out_a, int32_t* out_b, double* out_var,
                        size_t count, int bits);
int tfhe_hip_circuit_eq_dev(tfhe_hip_ctx* ctx, const int32_t* a_a, const int32_t* a_b,
                            const int32_t* b_a, const int32_t* b_b, int32_t* out_a, int32_t* out_b,
                            double* out_var, size_t count, int bits);

/* poly_multiplier / torus_polynomial_mul_r (src/numerics.rs:138-155,185-261)
 * on B independent pairs of N-coefficient polynomials d [B][N], s [B][N].
 * accumulate == 0: out [B][2N-1] = the full saturated product;
 * accumulate != 0: out [B][N] += first N coefficients (wrapping).
 * The result equals the exact integer convolution (rounded, saturated) as long as
 * an f64 FFT can resolve it: for every operand the path produces (|d| <= Bg/2 <=
 * 2^10 against any Torus32 polynomial) the distance to the nearest rounding
 * boundary stays above 0.4 (tfhe_hip_polymul_margin).  With |d| near 2^15 and
 * beyond, exact coefficients approach 2^53 and no f64 FFT, the reference's
 * included, determines the integer result any more. */
int tfhe_hip_polymul(tfhe_hip_ctx* ctx, const int32_t* d, const int32_t* s, int32_t* out,
                     int accumulate, size_t B);
int tfhe_hip_polymul_dev(tfhe_hip_ctx* ctx, const int32_t* d, const int32_t* s, int32_t* out,
                         int accumulate, size_t B);

/* Rounding margin of the products above, measured inside the kernel: the
 * reference rounds each coefficient of an f64 FFT product to the nearest integer
 * (src/numerics.rs:253-258), so two FFTs agree on the result as long as each
 * one's error stays below 0.5.  margin [B][2] receives, per product, the largest
 * |x - rint(x)| before rounding over all 2N-1 coefficients and the largest over
 * the coefficients whose rounding decides the result (|rint(x)| <= 2^31; larger
 * values saturate whatever their fraction is).  Test hook; nothing else is
 * written. */
int tfhe_hip_polymul_margin(tfhe_hip_ctx* ctx, const int32_t* d, const int32_t* s, double* margin,
                            size_t B);
int tfhe_hip_polymul_margin_dev(tfhe_hip_ctx* ctx, const int32_t* d, const int32_t* s,
                                double* margin, size_t B);

/* tgsw_extern_mul_to_tlwe (src/tgsw.rs:229-264) of B accumulators
 * t [B][k+2][N] with the loaded bk[bk_index]; out [B][k+2][N].  The _margin
 * form also returns the rounding margins (as above, over the 4 l products of
 * each accumulator) in margin [B][2], which may be NULL. */
int tfhe_hip_extprod(tfhe_hip_ctx* ctx, const int32_t* t, int bk_index, int32_t* out, size_t B);
int tfhe_hip_extprod_margin(tfhe_hip_ctx* ctx, const int32_t* t, int bk_index, int32_t* out,
                            double* margin, size_t B);
int tfhe_hip_extprod_dev(tfhe_hip_ctx* ctx, const int32_t* t, int bk_index, int32_t* out,
                         double* margin, size_t B);

/* tfhe_blind_rotate (src/bootstrapping.rs:109-134) of B accumulators
 * acc [B][k+2][N] (updated in place) by bara [B][n] with the loaded key.  Every
 * i32 is accepted as a rotation amount and reduced as mul_by_monomial does
 * (src/polynomial.rs:289-291: (bara / N) mod 2 picks the sign pattern, bara % N
 * the shift); the bootstrap itself only produces [-N, N-1]. */
int tfhe_hip_blind_rotate(tfhe_hip_ctx* ctx, int32_t* acc, const int32_t* bara, size_t B);
int tfhe_hip_blind_rotate_dev(tfhe_hip_ctx* ctx, int32_t* acc, const int32_t* bara, size_t B);

#ifdef __cplusplus
}
#endif
#endif
