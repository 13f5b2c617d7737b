// tfhe_hip.hip — HIP kernels (gfx950) and the C ABI of libtfhe_hip.so.
//
// The bootstrapped-gate path of the reference crate, per ciphertext:
//   gate prologue (src/gates.rs)            -> prologue_kernel
//   mod switch, test vector, blind rotate,
//   sample extract (src/bootstrapping.rs,
//   src/tgsw.rs, src/tlwe.rs,
//   src/numerics.rs, src/polynomial.rs)     -> blind_rotate_kernel
//   key switch (src/lwe.rs:601-648)         -> keyswitch_kernel
//
// One wavefront owns one ciphertext for the whole blind rotation: its
// accumulator lives in registers, every FFT exchange is wave-local (see
// wave_fft.hpp), and the FFT-domain bootstrapping key is streamed from L2.
// A workgroup is four independent wavefronts sharing two twiddle tables in LDS.

#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <string>
#include <vector>

#include "ctx.hpp"
#include "wave_fft.hpp"
#include "wave_fft64.hpp"

using wfft::cplx;
using wfft::static_for;

// Diagnostic build only (-DTFHE_STAMPS): per-phase cycle counts of one
// wavefront, read back with tfhe_hip_debug_stamps().  In the normal build
// TFHE_STAMP expands to nothing and no stamp executes.
#ifdef TFHE_STAMPS
__device__ unsigned long long g_stamps[16];
struct StampCtx {
  unsigned long long acc[16] = {0};
  unsigned long long last = 0;
  __device__ __forceinline__ void start() { last = clock64(); }
  __device__ __forceinline__ void mark(int k) {
    __builtin_amdgcn_sched_barrier(0);       // instructions do not move across a stamp
    const unsigned long long t = clock64();
    acc[k] += t - last;
    last = t;
    __builtin_amdgcn_sched_barrier(0);
  }
};
#define TFHE_STAMP(ctx, k) (ctx).mark(k)
#else
struct StampCtx {
  __device__ __forceinline__ void start() {}
};
#define TFHE_STAMP(ctx, k) ((void)0)
#endif

// Depth (in groups of four 16-byte loads) of the software pipelines that fetch
// the two key blocks of a level; tuned on MI355X.
#ifndef TFHE_K0_DEPTH
#define TFHE_K0_DEPTH 3
#endif
// The first loads of a level's first key block are issued before (1) or after (0) the
// forward transform.
#ifndef TFHE_K0_EARLY
#define TFHE_K0_EARLY 0
#endif
#ifndef TFHE_K1_DEPTH
#define TFHE_K1_DEPTH 2
#endif


// Kernels on the wavefront-per-transform transforms: the pointwise product of two spectra inside
// the first radix-2 stage of the inverse transform that follows it (wfft64::butterfly_ww: ten
// instructions per pair of spectrum values instead of twelve).
#ifndef TFHE_PRODUCT_STAGE1
#define TFHE_PRODUCT_STAGE1 1
#endif

// Low-latency kernel: the first loads of a wavefront's key block are issued before the
// forward transform; depth of their software pipeline in groups of four.
#ifndef TFHE_SPLIT_KEY_DEPTH
#define TFHE_SPLIT_KEY_DEPTH 2
#endif

// Where the accumulator of the blind rotation lives between uses: parked in
// accumulation registers by hand (0) or left to the register allocator (1, which
// keeps it in the registers the twisted transforms leave free: +2 % on MI355X).
#ifndef TFHE_ACC_VGPR
#define TFHE_ACC_VGPR 1
#endif

namespace {

#if TFHE_ACC_VGPR
__device__ __forceinline__ unsigned acc_put(unsigned v) { return v; }
__device__ __forceinline__ unsigned acc_get(unsigned a) { return a; }
#else
__device__ __forceinline__ unsigned acc_put(unsigned v) { return wfft::agpr_put(v); }
__device__ __forceinline__ unsigned acc_get(unsigned a) { return wfft::agpr_get(a); }
#endif

constexpr int kN = 1024;
constexpr int kWavesPerBlock = 4;
constexpr int kBlock = 64 * kWavesPerBlock;
constexpr int kTableBytes = 2 * 1024 * (int)sizeof(cplx);                 // tE + twist
constexpr int kLdsBytes = kWavesPerBlock * wfft::kWaveBytes + kTableBytes;  // 160 KiB

__device__ __forceinline__ int32_t wadd(int32_t a, int32_t b) { return (int32_t)((uint32_t)a + (uint32_t)b); }
__device__ __forceinline__ int32_t wsub(int32_t a, int32_t b) { return (int32_t)((uint32_t)a - (uint32_t)b); }
__device__ __forceinline__ int32_t wneg(int32_t a) { return (int32_t)(0u - (uint32_t)a); }

#ifndef TFHE_TEXTBOOK
// decode_message(x, 2N), src/numerics.rs:126-130 with N = 1024.
__device__ __forceinline__ int32_t mod_switch_2n(int32_t x) { return wadd(x, 1 << 20) >> 21; }
#else
// Textbook build: the nearest multiple of 1/2N as an UNSIGNED residue in [0, 2N)
// (src/numerics.rs:126-130 shifts arithmetically and yields [-N, N)).
__device__ __forceinline__ int32_t mod_switch_2n(int32_t x) { return (int32_t)(((uint32_t)x + (1u << 20)) >> 21); }
#endif

// The power of mul_by_monomial reduced as src/polynomial.rs:289-291 does it:
// cycle = (power / N).modulo(2) == 1, pw = power % N (truncating division and
// remainder; the low bit of a two's-complement quotient is its positive
// remainder modulo 2).  Any i32 power is accepted.
struct MonomialPower {
  int pw;
  bool cycle;
};
__device__ __forceinline__ MonomialPower monomial_power(int power) {
  return {power % kN, ((power / kN) & 1) != 0};
}

// Which form of the twist (wave_fft.hpp, SCALAR parameter of the twisted transforms) the
// half-wave kernels use: scalar loads (measured on MI355X: +3..7 % throughput over the LDS table
// where one wavefront per SIMD is busy on its own ciphertext).
#ifndef TFHE_BATCH_SCALAR_TWIST
#define TFHE_BATCH_SCALAR_TWIST 1
#endif
#ifndef TFHE_POLYMUL_SCALAR_TWIST
#define TFHE_POLYMUL_SCALAR_TWIST 1
#endif
constexpr bool kBatchScalar = TFHE_BATCH_SCALAR_TWIST != 0;
constexpr bool kPolymulScalar = TFHE_POLYMUL_SCALAR_TWIST != 0;

// Copies the pair of twiddle tables a kernel's form of the twist needs (make_tables: 1024
// entries each, the scalar form's pair first) into LDS.
template <bool SCALAR>
__device__ __forceinline__ void load_tables(char* lds_tables, const cplx* __restrict__ tables) {
  cplx* dst = reinterpret_cast<cplx*>(lds_tables);
  const cplx* src = tables + (SCALAR ? 0 : 2048);
  for (int i = threadIdx.x; i < 2048; i += blockDim.x) dst[i] = src[i];
  __syncthreads();
}

// Per-wavefront LDS share (wfft::kWaveBytes): the FFT exchange area, then the
// previous accumulator [2][N] int32 and the sample being decomposed as [N][2]
// int32 (coefficient-major, so one 8-byte load returns both polynomials).
constexpr int kPrevOff = wfft::kExchangeBytes;
constexpr int kTtOff = wfft::kExchangeBytes + 2 * kN * 4;
static_assert(kTtOff + 2 * kN * 4 <= wfft::kWaveBytes, "per-wavefront LDS share");

// The sample to decompose is stored XORed with tt_mask: digit p of
// src/tgsw.rs:299-328, ((c + 0x80000000) >> decal_p) & (Bg - 1)) - Bg/2, is then
// the sign-extended bit field [decal_p, decal_p + bg_bit) of the stored word
// (adding 2^31 flips bit 31; subtracting Bg/2 from a field equals flipping its
// top bit and sign-extending; for p = 0 the two flips of bit 31 cancel).
template <int L, int BGBIT>
constexpr uint32_t tt_mask() {
  uint32_t m = 0;
  for (int p = 1; p < L; p++) m |= 1u << (31 - p * BGBIT);
  return m;
}

// ---------------------------------------------------------------------------
// External product core (src/tgsw.rs:229-264, src/tlwe.rs:149-157,
// src/numerics.rs:138-155).
//
// tt (LDS) : the TLWE sample being decomposed, [N][2] int32, XORed with tt_mask.
// acc_a[r] : lane (h, q) accumulates coefficient t = q + 32 r of result
//            polynomial u = h; handles of values parked in accumulation
//            registers (wfft::agpr_*).  Every product is rounded and saturated
//            to i32 on its own before the wrapping add, as the reference does.
// bk       : FFT-domain key of this iteration, [l][2][32][64] cplx.
// ---------------------------------------------------------------------------
// One decomposition level of the external product.  Half-wave h takes digit
// polynomial h (the level-p digits of accumulator polynomial h): its twisted
// transform D_h (wave_fft.hpp), then the two products with the key polynomials
// of TGSW row (h, p): D_h * K_{h,h} contributes to result polynomial h, which is
// the one this half accumulates, and D_h * K_{h,1-h} to the other half's.  Lane
// (h, q) hands coefficient t = q + 32 r of the first product to
// acc.own(integral_constant<r>, value); the second product's coefficients, which
// belong to result polynomial 1 - h, are rounded by acc.prepare(value) as they
// appear and handed over together, acc.cross(int32 (&)[32]), so that an exchange
// between the half-waves is issued for all of them at once.
// tt: this lane's base (+ 8 q) into the [N][2] int32 sample (XORed with tt_mask).
// bk_level: [2][32][64] cplx: block 0 holds K_{h,h}, block 1 K_{h,1-h} in lane (h, q).
template <int BGBIT, bool SCALAR, class Acc>
__device__ __forceinline__ void level_product(int p, const char* tt, const cplx* __restrict__ bk_level,
                                              char* wave_tiles, const cplx* tE, const cplx* tq,
                                              int lane, Acc&& acc, StampCtx& st) {
  const int h = lane >> 5, q = lane & 31;
  char* tile = wave_tiles + h * wfft::kHalfBytes;
  double re[32], im[32];
  unsigned hold[128];      // parked spectrum D_h
  // gadget decomposition, src/tgsw.rs:299-328 (offset 0x80000000, tgsw.rs:50-52)
  const int decal = 32 - (p + 1) * BGBIT;
  const char* th = tt + 4 * h;
  wfft::wave_lds_fence();
  static_for<0, 32>([&](auto ic) WFFT_INLINE {
    constexpr int r = decltype(ic)::value;
    const int32_t t = *reinterpret_cast<const int32_t*>(th + r * 256);
    re[r] = (double)(int32_t)__builtin_amdgcn_sbfe(t, decal, BGBIT);
  });
  TFHE_STAMP(st, 1);
  const cplx* b0 = bk_level + lane;
  const cplx* b1 = bk_level + 2048 + lane;
  auto load_k0 = [&](auto ic) WFFT_INLINE { return b0[decltype(ic)::value * 64]; };
  auto load_k1 = [&](auto ic) WFFT_INLINE { return b1[decltype(ic)::value * 64]; };
  wfft::Pipe32<4, TFHE_K0_DEPTH, cplx> k0p;
#if TFHE_K0_EARLY
  k0p.prologue(load_k0);
  __builtin_amdgcn_sched_barrier(0);
#endif
  wfft::fwd1024_twisted_real<SCALAR>(re, im, tile, tE, tq, q);
  TFHE_STAMP(st, 2);
  {
    // Z = D_h * K_{h,h} in place; a copy of D_h is parked for the second product
#if !TFHE_K0_EARLY
    k0p.prologue(load_k0);
#endif
    k0p.run2(load_k0, [&](auto i0, const cplx& ka, auto i1, const cplx& kb) WFFT_INLINE {
      constexpr int s0 = decltype(i0)::value, s1 = decltype(i1)::value;
      wfft::agpr_put_f64(re[s0], hold[4 * s0 + 0], hold[4 * s0 + 1]);
      wfft::agpr_put_f64(im[s0], hold[4 * s0 + 2], hold[4 * s0 + 3]);
      wfft::agpr_put_f64(re[s1], hold[4 * s1 + 0], hold[4 * s1 + 1]);
      wfft::agpr_put_f64(im[s1], hold[4 * s1 + 2], hold[4 * s1 + 3]);
      wfft::cmul2<false>(re[s0], im[s0], ka, re[s1], im[s1], kb);
    });
  }
  TFHE_STAMP(st, 4);
  wfft::inv1024_twisted_real<false, SCALAR>(
      re, im, tile, tE, tq, q,
      [&](auto rc, double v) WFFT_INLINE { acc.own(rc, v); }, [](auto, double) {});
  TFHE_STAMP(st, 5);
  {
    // Z = D_h * K_{h,1-h} as D_h comes back from the accumulation registers
    wfft::Pipe32<4, TFHE_K1_DEPTH, cplx> k1p;
    k1p.prologue(load_k1);
    k1p.run2(load_k1, [&](auto i0, const cplx& ka, auto i1, const cplx& kb) WFFT_INLINE {
      constexpr int s0 = decltype(i0)::value, s1 = decltype(i1)::value;
      re[s0] = wfft::agpr_get_f64(hold[4 * s0 + 0], hold[4 * s0 + 1]);
      im[s0] = wfft::agpr_get_f64(hold[4 * s0 + 2], hold[4 * s0 + 3]);
      re[s1] = wfft::agpr_get_f64(hold[4 * s1 + 0], hold[4 * s1 + 1]);
      im[s1] = wfft::agpr_get_f64(hold[4 * s1 + 2], hold[4 * s1 + 3]);
      wfft::cmul2<false>(re[s0], im[s0], ka, re[s1], im[s1], kb);
    });
  }
  TFHE_STAMP(st, 6);
  int32_t other[32];
  wfft::inv1024_twisted_real<false, SCALAR>(
      re, im, tile, tE, tq, q,
      [&](auto rc, double v) WFFT_INLINE { other[decltype(rc)::value] = acc.prepare(v); },
      [](auto, double) {});
  acc.cross(other);
  TFHE_STAMP(st, 7);
}

struct NoMargin {
  __device__ __forceinline__ void track(double) {}
};
// Largest |x - rint(x)| seen before rounding (see polymul64_kernel).
struct Margin {
  double all = 0.0, deciding = 0.0;
  __device__ __forceinline__ void track(double x) {
    const double rx = rint(x), f = fabs(x - rx);
    all = fmax(all, f);
    if (fabs(rx) <= 2147483648.0) deciding = fmax(deciding, f);
  }
};

// Sinks of level_product.  RegisterSink: wrapping adds of rounded, saturated products
// (numerics.rs:138-155) into the accumulator a wavefront keeps in accumulation registers;
// the products for the other half-wave's polynomial cross over through the wavefront's
// exchange area in LDS (TFHE_CROSS_LDS, +0.8 % over 32 ds_bpermute on MI355X).
#ifndef TFHE_CROSS_LDS
#define TFHE_CROSS_LDS 1
#endif
template <class M>
struct RegisterSink {
  unsigned (&acc_a)[32];
  M& margin;
  int partner;        // the other half-wave's lane, as a ds_bpermute address
  char* mine;         // this lane's and the partner lane's 16 bytes in the wavefront's exchange area
  char* theirs;
  __device__ __forceinline__ int32_t prepare(double v) {
    margin.track(v);
    return wfft::round_sat_i32(v);
  }
  template <class RC>
  __device__ __forceinline__ void own(RC, double v) {
    constexpr int r = RC::value;
    acc_a[r] = acc_put((unsigned)wadd((int32_t)acc_get(acc_a[r]), prepare(v)));
  }
  __device__ __forceinline__ void cross(int32_t (&x)[32]) {
#if TFHE_CROSS_LDS
    // through the wavefront's exchange area, free between two transforms: 8 + 8 16-byte
    // accesses per lane instead of 32 ds_bpermute
    typedef int int4v __attribute__((ext_vector_type(4)));
    wfft::wave_lds_fence();
    static_for<0, 8>([&](auto gc) WFFT_INLINE {
      constexpr int g = decltype(gc)::value;
      *reinterpret_cast<int4v*>(mine + g * 1024) =
          int4v{x[4 * g], x[4 * g + 1], x[4 * g + 2], x[4 * g + 3]};
    });
    wfft::wave_lds_fence();
    static_for<0, 8>([&](auto gc) WFFT_INLINE {
      constexpr int g = decltype(gc)::value;
      const int4v v = *reinterpret_cast<const int4v*>(theirs + g * 1024);
      x[4 * g] = v.x; x[4 * g + 1] = v.y; x[4 * g + 2] = v.z; x[4 * g + 3] = v.w;
    });
    wfft::wave_lds_fence();
#else
    static_for<0, 32>([&](auto ic) WFFT_INLINE {
      constexpr int r = decltype(ic)::value;
      x[r] = __builtin_amdgcn_ds_bpermute(partner, x[r]);
    });
#endif
    static_for<0, 32>([&](auto ic) WFFT_INLINE {
      constexpr int r = decltype(ic)::value;
      acc_a[r] = acc_put((unsigned)wadd((int32_t)acc_get(acc_a[r]), x[r]));
    });
  }
};

template <int L, int BGBIT, class M = NoMargin>
__device__ __forceinline__ void extprod_accumulate(unsigned (&acc_a)[32], const cplx* __restrict__ bk,
                                                   char* wave_lds, const cplx* tE, const cplx* tq,
                                                   int lane, StampCtx& st, M&& margin = M{}) {
  const char* tt = wave_lds + kTtOff + (lane & 31) * 8;
  RegisterSink<std::remove_reference_t<M>> accumulate{acc_a, margin, (lane ^ 32) << 2, wave_lds + lane * 16,
                                                      wave_lds + (lane ^ 32) * 16};
#pragma unroll 1
  for (int p = 0; p < L; p++)
    level_product<BGBIT, kBatchScalar>(p, tt, bk + (size_t)p * 2 * 2048, wave_lds, tE, tq, lane, accumulate,
                                       st);
}

// t = X^abar (prev + acc) - acc for polynomials 0 and 1
// (src/bootstrapping.rs:136-146, src/polynomial.rs:281-313), then prev <- acc.
// acc is parked in accumulation registers (lane (h, q) register r = polynomial
// h, coefficient q + 32 r); prev and t (XORed with mask, see tt_mask) live in
// the wavefront's LDS share; the
// sum prev + acc is staged through the exchange area for the rotation.
// GENERAL: abar is any i32 (caller-supplied powers of the blind-rotate hook);
// otherwise abar is in [-N, N-1], all that decode_message(x, 2N) produces
// (src/numerics.rs:126-130), and cycle is set exactly when abar == -N.
template <bool GENERAL>
__device__ __forceinline__ void rotate_diff(const unsigned (&acc_a)[32], int abar, char* wave_lds,
                                            int lane, uint32_t mask) {
  const int h = lane >> 5, q = lane & 31;
  int32_t* S = reinterpret_cast<int32_t*>(wave_lds) + h * kN;              // [2][N] prev + acc
  int32_t* pv = reinterpret_cast<int32_t*>(wave_lds + kPrevOff) + h * kN + q;
  int32_t* tt = reinterpret_cast<int32_t*>(wave_lds + kTtOff) + 2 * q + h;
  // Loads and stores are issued in separate batches: the compiler cannot tell
  // the LDS areas apart and would otherwise serialise load, wait, store.
  int32_t v[32];
  wfft::wave_lds_fence();
  static_for<0, 32>([&](auto ic) WFFT_INLINE { v[decltype(ic)::value] = pv[32 * decltype(ic)::value]; });
  wfft::wave_lds_fence();
  static_for<0, 32>([&](auto ic) WFFT_INLINE {
    constexpr int r = decltype(ic)::value;
    const int32_t a = (int32_t)acc_get(acc_a[r]);
    S[q + 32 * r] = wadd(v[r], a);
    pv[32 * r] = a;
  });
  wfft::wave_lds_fence();
  // X^abar, polynomial.rs:281-313: with cycle = (abar / N) mod 2 and
  // pw = abar % N (both truncating, as in Rust), coefficient n comes from
  // (n - pw) mod N; it is negated when it wraps (pw > 0 and n < pw) unless cycle
  // is set, and negated everywhere else exactly when cycle is set.
  if constexpr (GENERAL) {
    const MonomialPower mp = monomial_power(abar);
    const int base = q - mp.pw;
    static_for<0, 32>([&](auto ic) WFFT_INLINE {
      constexpr int r = decltype(ic)::value;
      v[r] = S[(base + 32 * r) & (kN - 1)];
    });
    wfft::wave_lds_fence();
    static_for<0, 32>([&](auto ic) WFFT_INLINE {
      constexpr int r = decltype(ic)::value;
      const bool neg = (base < -32 * r) != mp.cycle;
      const int32_t x = neg ? wneg(v[r]) : v[r];
      tt[64 * r] = (int32_t)((uint32_t)wsub(x, (int32_t)acc_get(acc_a[r])) ^ mask);
    });
  } else {
    // pw = abar and cycle = 0 except for abar = -N (pw = 0, cycle = 1: all negated);
    // `wrap` folds both into one comparison per coefficient.
    const int base = q - abar;
    const int wrap = abar > 0 ? base : (abar == -kN ? -(1 << 20) : (1 << 20));
    static_for<0, 32>([&](auto ic) WFFT_INLINE {
      constexpr int r = decltype(ic)::value;
      v[r] = S[(base + 32 * r) & (kN - 1)];
    });
    wfft::wave_lds_fence();
    static_for<0, 32>([&](auto ic) WFFT_INLINE {
      constexpr int r = decltype(ic)::value;
      const bool neg = wrap < -32 * r;
      const int32_t x = neg ? wneg(v[r]) : v[r];
      tt[64 * r] = (int32_t)((uint32_t)wsub(x, (int32_t)acc_get(acc_a[r])) ^ mask);
    });
  }
  wfft::wave_lds_fence();
}

struct BlindRotateArgs {
  const int32_t* bara;      // [B][n]   mod-switched mask (src/bootstrapping.rs:51-55)
  const int32_t* barb;      // [B]      mod-switched body (bootstrap mode)
  const cplx* bkfft;        // [n][l][2][32][64]
  const cplx* tables;       // make_tables: two pairs of 1024-entry tables
  const cplx* tables64;     // make_tables64
  int32_t* acc_io;          // hook mode: [B][3][N] accumulators, updated in place
  int32_t* out_a;           // bootstrap mode: [B][N] extracted mask
  int32_t* out_b;           // bootstrap mode: [B]
  int32_t mu;
  int32_t n;
  int32_t B;
};

// tfhe_blind_rotate_and_extract (src/bootstrapping.rs:77-134) when HOOK is
// false; tfhe_blind_rotate on caller-supplied accumulators when HOOK is true.
//
// The accumulator's last polynomial (index k+1, where TLweSample::trivial puts
// the test vector, src/tlwe.rs:99-109) is never decomposed and never written
// by the external product (src/tgsw.rs:280-297, src/tlwe.rs:149-157), so it is
// constant over the loop; only polynomials 0..k are carried.
template <int L, int BGBIT, bool HOOK>
__global__ __launch_bounds__(kBlock, 1) void blind_rotate_kernel(BlindRotateArgs args) {
  __shared__ __attribute__((aligned(16))) char lds[kLdsBytes];
  char* lds_tables = lds + kWavesPerBlock * wfft::kWaveBytes;
  load_tables<kBatchScalar>(lds_tables, args.tables);
  const cplx* tE = reinterpret_cast<const cplx*>(lds_tables);
  const cplx* tq = tE + 1024;

  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);   // wave-uniform, kept scalar
  const int h = lane >> 5, q = lane & 31;
  char* wave_tiles = lds + wave * wfft::kWaveBytes;
  // Workgroups are persistent (one per compute unit): wavefront slot
  // s = blockIdx.x * 4 + wave takes ciphertexts s, s + slots, s + 2 slots, ...
  const int slot = blockIdx.x * kWavesPerBlock + wave;
  const int slots = gridDim.x * kWavesPerBlock;
  StampCtx st;
  st.start();

#pragma unroll 1
  for (int b = slot; b < args.B; b += slots) {
  unsigned acc_a[32];              // parked in accumulation registers
  wfft::wave_lds_fence();
  static_for<0, 32>([&](auto ic) WFFT_INLINE {
    constexpr int r = decltype(ic)::value;
    acc_a[r] = acc_put(
        HOOK ? (unsigned)args.acc_io[((size_t)b * 3 + h) * kN + q + 32 * r] : 0u);
    // the previous accumulator starts at zero, bootstrapping.rs:116-117
    reinterpret_cast<int32_t*>(wave_tiles + kPrevOff)[h * kN + q + 32 * r] = 0;
  });

  const int32_t* bara = args.bara + (size_t)b * args.n;
  int abar64 = 0;          // the next 64 rotation amounts, one per lane
#pragma unroll 1
  for (int i = 0; i < args.n; i++) {
    if ((i & 63) == 0) abar64 = (i + lane < args.n) ? bara[i + lane] : 0;
    const int abar = __builtin_amdgcn_readlane(abar64, i & 63);
    if (abar == 0) continue;                                   // bootstrapping.rs:122-126
    // Lane-derived addresses are recomputed every iteration (a handful of
    // instructions) instead of being kept live, or spilled, across the loop.
    int ln = lane;
    asm volatile("" : "+v"(ln));
    TFHE_STAMP(st, 8);
    rotate_diff<HOOK>(acc_a, abar, wave_tiles, ln, tt_mask<L, BGBIT>());   // also prev <- acc
    TFHE_STAMP(st, 0);
    extprod_accumulate<L, BGBIT>(acc_a, args.bkfft + (size_t)i * L * 2 * 2048, wave_tiles, tE, tq, ln,
                                 st);
#ifdef TFHE_STAMPS
    if (i == args.n - 1 && b == args.B / 2 && lane == 0)
      for (int k = 0; k < 16; k++) g_stamps[k] = st.acc[k];
#endif
  }

  if (HOOK) {
    static_for<0, 32>([&](auto ic) WFFT_INLINE {
      constexpr int r = decltype(ic)::value;
      args.acc_io[((size_t)b * 3 + h) * kN + q + 32 * r] = (int32_t)acc_get(acc_a[r]);
    });
  } else {
    // extract_lwe, src/tlwe.rs:159-180 (k = 1): a[0] = acc0[0], a[j] = -acc0[N-j].
    if (h == 0) {
      int32_t* oa = args.out_a + (size_t)b * kN;
      static_for<0, 32>([&](auto ic) WFFT_INLINE {
        constexpr int r = decltype(ic)::value;
        const int t = q + 32 * r;
        const int32_t v = (int32_t)acc_get(acc_a[r]);
        if (t == 0) oa[0] = v;
        else oa[kN - t] = wneg(v);
      });
    }
    if (lane == 0) {
      // b = coefficient 0 of X^{(-barb) mod 2N} * [mu]*N  (bootstrapping.rs:58-96,
      // polynomial.rs:281-313): +mu iff barb >= 0.
      const int32_t barb = args.barb[b];
      args.out_b[b] = barb >= 0 ? args.mu : wneg(args.mu);
    }
  }
  }
}

// tgsw_extern_mul_to_tlwe on caller-supplied samples (parity hook).
// margin (optional, [B][2]): the rounding margins of the 4 l products of each sample.
template <int L, int BGBIT, bool MARGIN>
__global__ __launch_bounds__(kBlock, 1) void extprod_kernel(const int32_t* __restrict__ t,
                                                            const cplx* __restrict__ bk_i,
                                                            const cplx* __restrict__ tables,
                                                            int32_t* __restrict__ out,
                                                            double* __restrict__ margin, int B) {
  __shared__ __attribute__((aligned(16))) char lds[kLdsBytes];
  char* lds_tables = lds + kWavesPerBlock * wfft::kWaveBytes;
  load_tables<kBatchScalar>(lds_tables, tables);
  const cplx* tE = reinterpret_cast<const cplx*>(lds_tables);
  const cplx* tq = tE + 1024;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int h = lane >> 5, q = lane & 31;
  const int b = blockIdx.x * kWavesPerBlock + wave;
  if (b >= B) return;
  char* wave_tiles = lds + wave * wfft::kWaveBytes;
  unsigned acc_a[32];
  static_for<0, 32>([&](auto ic) WFFT_INLINE {
    constexpr int r = decltype(ic)::value;
    reinterpret_cast<int32_t*>(wave_tiles + kTtOff)[2 * (q + 32 * r) + h] =
        (int32_t)((uint32_t)t[((size_t)b * 3 + h) * kN + q + 32 * r] ^ tt_mask<L, BGBIT>());
    acc_a[r] = acc_put(0u);
  });
  StampCtx st;
  if constexpr (MARGIN) {
    Margin m;
    extprod_accumulate<L, BGBIT>(acc_a, bk_i, wave_tiles, tE, tq, lane, st, m);
    for (int off = 32; off >= 1; off >>= 1) {
      m.all = fmax(m.all, __shfl_xor(m.all, off, 64));
      m.deciding = fmax(m.deciding, __shfl_xor(m.deciding, off, 64));
    }
    if (lane == 0) {
      margin[2 * (size_t)b + 0] = m.all;
      margin[2 * (size_t)b + 1] = m.deciding;
    }
  } else {
    extprod_accumulate<L, BGBIT>(acc_a, bk_i, wave_tiles, tE, tq, lane, st);
  }
  static_for<0, 32>([&](auto ic) WFFT_INLINE {
    constexpr int r = decltype(ic)::value;
    out[((size_t)b * 3 + h) * kN + q + 32 * r] = (int32_t)acc_get(acc_a[r]);
    if (h == 0) out[((size_t)b * 3 + 2) * kN + q + 32 * r] = 0;     // tgsw.rs:236-244
  });
}

// One-time transform of the bootstrapping key (replaces the per-call forward
// FFT of the key polynomial at src/numerics.rs:216-238).  With K_{j,u} the key
// polynomial that digit polynomial j multiplies for result polynomial u (flat
// input [i][p][j][u][N]), wavefront (i, p, block) writes the twisted transforms
// of K_{h,h} (block 0) or K_{h,1-h} (block 1) from half-wave h, scaled by 1/1024
// for the unnormalised inverse transform.
__global__ __launch_bounds__(kBlock, 1) void bk_fft_kernel(const int32_t* __restrict__ bk,
                                                           const cplx* __restrict__ tables,
                                                           cplx* __restrict__ bkfft, int count) {
  __shared__ __attribute__((aligned(16))) char lds[kLdsBytes];
  char* lds_tables = lds + kWavesPerBlock * wfft::kWaveBytes;
  load_tables<kBatchScalar>(lds_tables, tables);
  const cplx* tE = reinterpret_cast<const cplx*>(lds_tables);
  const cplx* tq = tE + 1024;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int h = lane >> 5, q = lane & 31;
  const int s = blockIdx.x * kWavesPerBlock + wave;        // (i * l + p) * 2 + block
  if (s >= count) return;
  char* wave_tiles = lds + wave * wfft::kWaveBytes;
  const int block = s & 1, u = block ? 1 - h : h;
  const int32_t* p0 = bk + (((size_t)(s >> 1) * 2 + h) * 2 + u) * kN;
  double re[32], im[32];
  static_for<0, 32>([&](auto ic) WFFT_INLINE {
    constexpr int r = decltype(ic)::value;
    re[r] = (double)p0[q + 32 * r];
  });
  wfft::fwd1024_twisted_real<kBatchScalar>(re, im, wave_tiles + h * wfft::kHalfBytes, tE, tq, q);
  cplx* dst = bkfft + (size_t)s * 2048 + lane;
  static_for<0, 32>([&](auto ic) WFFT_INLINE {
    constexpr int p = decltype(ic)::value;
    cplx v;
    v.x = re[p] * (1.0 / 1024.0);
    v.y = im[p] * (1.0 / 1024.0);
    dst[p * 64] = v;
  });
}

// poly_multiplier / torus_polynomial_mul_r (src/numerics.rs:138-155,185-261) on independent
// pairs, one product per WAVEFRONT (wave_fft64.hpp): the twisted transforms of d and s, their
// pointwise product, and one inverse transform whose untwisted real part is the low half of the
// linear convolution and whose negated imaginary part is the high half.  A wavefront needs 16
// complex registers for a transform and as many for the spectrum of d that waits while s is
// transformed, so two wavefronts share a SIMD and one's LDS round trips hide behind the other's
// arithmetic.  Eight wavefronts per workgroup, persistent; the operands of the NEXT product are
// copied global -> LDS by the DMA path (global_load_lds, no registers) as soon as this
// product's operands have been read.
//
// MARGIN (test hook): nothing is written to out; margin [B][2] receives, per product, the
// largest |x - rint(x)| over all 2N-1 coefficients before rounding and the largest over the
// coefficients whose rounding decides the result (|rint(x)| <= 2^31; beyond that the cast
// saturates whatever the fraction is).
typedef __attribute__((address_space(3))) void* lds_dma_ptr;

constexpr int kWaves64 = 8;
constexpr int kStage64Bytes = 2 * kN * 4;                                  // d | s of one product
constexpr int kWave64Bytes = wfft64::kTileBytes + kStage64Bytes;
constexpr int kLds64Bytes = kWaves64 * kWave64Bytes + wfft64::kTableBytes;

__device__ __forceinline__ wfft64::Tables load_tables64(char* lds_tables, const cplx* __restrict__ tables64) {
  cplx* dst = reinterpret_cast<cplx*>(lds_tables);
  for (int i = threadIdx.x; i < wfft64::kT1Entries + wfft64::kT2Entries; i += blockDim.x) dst[i] = tables64[i];
  __syncthreads();
  return {dst, dst + wfft64::kT1Entries};
}

template <bool ACCUMULATE, bool MARGIN>
__global__ __launch_bounds__(64 * kWaves64) void polymul64_kernel(const int32_t* __restrict__ d,
                                                                  const int32_t* __restrict__ s,
                                                                  const cplx* __restrict__ tables64,
                                                                  int32_t* __restrict__ out,
                                                                  double* __restrict__ margin, int B) {
  __shared__ __attribute__((aligned(16))) char lds[kLds64Bytes];
  const wfft64::Tables tb = load_tables64(lds + kWaves64 * kWave64Bytes, tables64);
  const int lane0 = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  char* tile = lds + wave * kWave64Bytes;
  char* stage = tile + wfft64::kTileBytes;
  const int stride = gridDim.x * kWaves64;

  auto fetch = [&](int prod, int lane) WFFT_INLINE {
    const int32_t* src[2] = {d + (size_t)prod * kN, s + (size_t)prod * kN};
    static_for<0, 8>([&](auto ic) WFFT_INLINE {
      constexpr int i = decltype(ic)::value;
      __builtin_amdgcn_global_load_lds((const void*)(src[i >> 2] + (i & 3) * 256 + lane * 4),
                                       (lds_dma_ptr)(stage + i * 1024), 16, 0, 0);
    });
  };

  int prod = blockIdx.x * kWaves64 + wave;
  if (prod < B) fetch(prod, lane0);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");             // the first operands have landed
#pragma unroll 1
  for (; prod < B; prod += stride) {
    auto fresh_lane = [&]() WFFT_INLINE {
      int l = lane0;
      asm volatile("" : "+v"(l));
      return l;
    };
    double re[16], im[16];
    double pre[16], pim[16]; // spectrum of d, waiting while s is transformed
    {
      const int lane = fresh_lane();
      const int32_t* in = reinterpret_cast<const int32_t*>(stage) + lane;
      static_for<0, 16>([&](auto ic) WFFT_INLINE {
        constexpr int r = decltype(ic)::value;
        re[r] = (double)in[64 * r];
      });
      wfft64::fwd1024_twisted_real<4>(re, im, tile, tb, lane);
      static_for<0, 16>([&](auto ic) WFFT_INLINE {
        constexpr int p = decltype(ic)::value;
        pre[p] = re[p];
        pim[p] = im[p];
      });
      constexpr double kScale = 1.0 / 1024.0;                    // a power of two: exact
      static_for<0, 16>([&](auto ic) WFFT_INLINE {
        constexpr int r = decltype(ic)::value;
        re[r] = (double)in[kN + 64 * r] * kScale;
      });
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      if (prod + stride < B) fetch(prod + stride, lane);
      wfft64::fwd1024_twisted_real<4>(re, im, tile, tb, lane);
    }
    __builtin_amdgcn_sched_barrier(0);
#if TFHE_PRODUCT_STAGE1
    static_for<0, 8>([&](auto mc) WFFT_INLINE {
      constexpr int p0 = 2 * decltype(mc)::value, p1 = p0 + 1;
      wfft64::butterfly_ww<false>(re[p0], im[p0], re[p1], im[p1], cplx{pre[p0], pim[p0]}, cplx{pre[p1], pim[p1]});
    });
#else
    wfft::chunked_for32<4>([&](auto ic) WFFT_INLINE {
      constexpr int p = decltype(ic)::value;
      if constexpr (p < 16) {
        const double pr = pre[p], pi = pim[p];
        const double qr = re[p], qi = im[p];
        re[p] = pr * qr - pi * qi;
        im[p] = pr * qi + pi * qr;
      }
    });
#endif
    __builtin_amdgcn_sched_barrier(0);
    const int lane = fresh_lane();
    // afterwards re = coefficients t < N, im = coefficients N + t of this product, t = lane + 64 r
    wfft64::inv1024_twisted_real<!ACCUMULATE, 4, TFHE_PRODUCT_STAGE1 != 0>(
        re, im, tile, tb, lane,
        [&](auto rc, double v) WFFT_INLINE { re[decltype(rc)::value] = v; },
        [&](auto rc, double v) WFFT_INLINE { im[decltype(rc)::value] = v; });
    if (MARGIN) {
      Margin mg;
      static_for<0, 16>([&](auto ic) WFFT_INLINE {
        constexpr int r = decltype(ic)::value;
        mg.track(re[r]);
        if (r < 15 || lane < 63) mg.track(im[r]);
      });
      for (int off = 32; off >= 1; off >>= 1) {
        mg.all = fmax(mg.all, __shfl_xor(mg.all, off, 64));
        mg.deciding = fmax(mg.deciding, __shfl_xor(mg.deciding, off, 64));
      }
      if (lane == 0) {
        margin[2 * (size_t)prod + 0] = mg.all;
        margin[2 * (size_t)prod + 1] = mg.deciding;
      }
    } else {
      // The rounded coefficients go through the (now free) tile in coefficient order, so that a
      // lane stores 16 contiguous bytes: 4 + 4 or 8 wide accesses instead of 16 + 16 or 32 of four
      // bytes (the narrow stores cost the full product a tenth of its time).  Rows of the full
      // form are 2N - 1 words long, i.e. 4-byte aligned only, which the wide accesses allow.
      typedef int int4u __attribute__((ext_vector_type(4), aligned(4)));
      typedef int int3u __attribute__((ext_vector_type(3), aligned(4)));
      typedef int int4v __attribute__((ext_vector_type(4)));
      int32_t* T = reinterpret_cast<int32_t*>(tile);
      wfft::wave_lds_fence();
      static_for<0, 16>([&](auto ic) WFFT_INLINE {
        constexpr int r = decltype(ic)::value;
        T[lane + 64 * r] = wfft::round_sat_i32(re[r]);
        if (!ACCUMULATE) T[kN + lane + 64 * r] = wfft::round_sat_i32(im[r]);
      });
      wfft::wave_lds_fence();
      const int4v* Q = reinterpret_cast<const int4v*>(tile) + lane;      // coefficients 4 lane + 256 j ..
      if (ACCUMULATE) {
        int4u* dst = reinterpret_cast<int4u*>(out + (size_t)prod * kN + 4 * lane);
        int4v old[4];
        static_for<0, 4>([&](auto jc) WFFT_INLINE { old[decltype(jc)::value] = dst[64 * decltype(jc)::value]; });
        static_for<0, 4>([&](auto jc) WFFT_INLINE {
          constexpr int j = decltype(jc)::value;
          const int4v q = Q[64 * j];
          dst[64 * j] = int4v{wadd(old[j].x, q.x), wadd(old[j].y, q.y), wadd(old[j].z, q.z), wadd(old[j].w, q.w)};
        });
      } else {
        int32_t* dst = out + (size_t)prod * (2 * kN - 1) + 4 * lane;
        static_for<0, 8>([&](auto jc) WFFT_INLINE {
          constexpr int j = decltype(jc)::value;
          const int4v q = Q[64 * j];
          if (j < 7 || lane < 63) *reinterpret_cast<int4u*>(dst + 256 * j) = q;
          else *reinterpret_cast<int3u*>(dst + 256 * j) = int3u{q.x, q.y, q.z};     // 2N - 1 coefficients
        });
      }
      wfft::wave_lds_fence();
    }
    // The next product's operands (fetched above) must have landed; this product's stores need
    // not have.  Memory operations complete in the order issued and at least kStoresAfterFetch
    // of them were issued after the fetch, so that many may stay in flight (waiting for all of
    // them exposed the write latency once per product: a tenth of the kernel's time).
    constexpr int kStoresAfterFetch = MARGIN ? 0 : 8;
    asm volatile("s_waitcnt vmcnt(%0)" : : "n"(kStoresAfterFetch) : "memory");
  }
}

// ---------------------------------------------------------------------------
// tfhe_blind_rotate_and_extract / tfhe_blind_rotate with the transforms of wave_fft64.hpp:
// one wavefront still owns one ciphertext for all n iterations, but a transform now takes the
// whole wavefront (64 lanes x 16 complex registers) and the two digit polynomials of a level
// are processed one after the other.  A wavefront then needs half the registers and half the
// LDS of blind_rotate_kernel, so TWO wavefronts share a SIMD and the LDS round trips,
// key loads and waits of one hide behind the arithmetic of the other; and both result
// polynomials live in the same lanes, so the second product of a level needs no exchange.
//
// Lane `lane`, register r holds coefficient lane + 64 r of a polynomial.  Everything a
// ciphertext needs between iterations stays on the chip: the accumulator acc[2][16] and the
// sample being decomposed (PackedSample: only the bits the levels read, 20 - 22 registers) in
// registers; the previous accumulator, touched once per iteration, in 8 KiB of the wavefront's
// LDS share (which also stages prev + acc for the rotation) next to the 9 KiB exchange tile.
// Global memory is read for the key and the rotation amounts and written for the result only.
// The key is the one blind_rotate_kernel reads ([n][l][block][32][64], half-wave (h, q),
// register p_old: value q + 32 br5(p_old) of K_{h,h} / K_{h,1-h}): wave_fft64.hpp's spectrum
// layout wants value kr + 16 c + 64 br4(p) in lane (kr3 kr2 c1 c0 kr1 kr0), register p, which is
// q = kr + 16 (c & 1), p_old = p + 16 (c >> 1): a per-lane base address, then the same
// register stride.
// ---------------------------------------------------------------------------
constexpr int kPrev64Bytes = 2 * kN * 4;
static_assert(wfft64::kTileBytes + kPrev64Bytes == kWave64Bytes, "one LDS share for both wavefront-per-transform kernels");
#ifndef TFHE_K64_DEPTH
#define TFHE_K64_DEPTH 3
#endif
// The wavefronts of a workgroup meet at a barrier every TFHE_ITER_BARRIER iterations (a power of
// two, 0 = never).  Without it the two wavefronts of a SIMD drift apart (the older one is served
// first) until every wavefront streams the key on its own: 18 GB instead of 4 GB fetched per
// 16 384-gate launch, and 5 % more time.
#ifndef TFHE_ITER_BARRIER
#define TFHE_ITER_BARRIER 8
#endif

// The sample being decomposed, XORed with tt_mask, packed into registers: a level only reads
// the top KEEP = l bg_bit bits of a coefficient.  For coefficient t = lane + 64 r, word[r] holds
// those bits of polynomial 0 where they are (bits 31 .. 32 - KEEP) and the top 32 - KEEP bits of
// polynomial 1 below them; the remaining REM = 2 KEEP - 32 bits of polynomial 1 are packed PER to a
// word in rem[].  20 registers for l = 2, Bg = 2^10 and 22 for l = 3, Bg = 2^7 instead of 32, which
// is what lets the sample stay in registers and the previous accumulator take its place in LDS.
template <int L, int BGBIT>
struct PackedSample {
  static constexpr int KEEP = L * BGBIT;
  static constexpr int REM = 2 * KEEP - 32;
  static_assert(KEEP < 32 && REM > 0 && REM <= 16, "packing of the decomposed sample");
  static constexpr int PER = 32 / REM;
  static constexpr int NREM = (16 + PER - 1) / PER;
  uint32_t word[16];
  uint32_t rem[NREM];

  // x0, x1: coefficient lane + 64 r of polynomials 0 and 1 (already XORed with tt_mask)
  template <int R>
  __device__ __forceinline__ void put(uint32_t x0, uint32_t x1) {
    word[R] = (x0 & (~0u << (32 - KEEP))) | (x1 >> KEEP);
    const uint32_t rest = __builtin_amdgcn_ubfe(x1, 32 - KEEP, REM);
    if constexpr (R % PER == 0) rem[R / PER] = rest;
    else rem[R / PER] |= rest << ((R % PER) * REM);
  }
  // the word whose bit field [decal, decal + bg_bit) is digit p of polynomial H, coefficient lane + 64 R
  template <int H, int R>
  __device__ __forceinline__ uint32_t get() const {
    if constexpr (H == 0) return word[R];
    else
      return (word[R] << KEEP) | (__builtin_amdgcn_ubfe(rem[R / PER], (R % PER) * REM, REM) << (32 - KEEP));
  }
};

struct RegisterSink64 {
  int32_t (&acc)[2][16];
  template <int U, class RC, class M>
  __device__ __forceinline__ void add(RC, double v, M& margin) {
    margin.track(v);
    acc[U][RC::value] = wadd(acc[U][RC::value], wfft::round_sat_i32(v));
  }
};

// One decomposition level of the external product (src/tgsw.rs:229-264, src/tlwe.rs:149-157,
// src/numerics.rs:138-155): for h = 0, 1 the level-p digits of tt[h] (src/tgsw.rs:299-328),
// their twisted transform D_h, and the products D_h K_{h,h} -> result polynomial h and
// D_h K_{h,1-h} -> result polynomial 1 - h, each rounded and saturated on its own before the
// wrapping add.  bk_level: [2][32][64] complex (block, register, lane of blind_rotate_kernel).
template <int BGBIT, class TT, class M>
__device__ __forceinline__ void level_product64(int p, const TT& tt, const cplx* __restrict__ bk_level,
                                                int32_t (&acc)[2][16], char* tile, const wfft64::Tables& tb,
                                                int lane, M& margin) {
  const int decal = 32 - (p + 1) * BGBIT;
  static_for<0, 2>([&](auto hc) WFFT_INLINE {
    constexpr int h = decltype(hc)::value;
    double re[16], im[16], zr[16], zi[16];
    static_for<0, 16>([&](auto rc) WFFT_INLINE {
      constexpr int r = decltype(rc)::value;
      re[r] = (double)(int32_t)__builtin_amdgcn_sbfe(tt.template get<h, r>(), decal, BGBIT);
    });
    // this lane's spectrum values in the half-wave layout of the key (see the kernel's comment)
    const int key_lane = (((lane >> 3) & 1) * 16) * 64 + h * 32 + (((lane >> 4) << 2) | (lane & 3)) +
                         16 * ((lane >> 2) & 1);
    const cplx* ka = bk_level + key_lane;
    const cplx* kb = bk_level + 2048 + key_lane;
#ifdef TFHE_ABLATE_KEY
    auto load_a = [&](auto ic) WFFT_INLINE { (void)ka; return cplx{0.75 + decltype(ic)::value, 0.5}; };
    auto load_b = [&](auto ic) WFFT_INLINE { (void)kb; return cplx{0.25 + decltype(ic)::value, 1.5}; };
#else
    auto load_a = [&](auto ic) WFFT_INLINE { return ka[decltype(ic)::value * 64]; };
    auto load_b = [&](auto ic) WFFT_INLINE { return kb[decltype(ic)::value * 64]; };
#endif
    wfft64::fwd1024_twisted_real(re, im, tile, tb, lane);
    {
      // Z = D_h K_{h,h} into registers of its own: D_h stays where it is for the second product
      wfft64::Pipe16<4, TFHE_K64_DEPTH, cplx> kp;
      kp.prologue(load_a);
      kp.run2(load_a, [&](auto i0, const cplx& k0, auto i1, const cplx& k1) WFFT_INLINE {
        constexpr int s0 = decltype(i0)::value, s1 = decltype(i1)::value;
#if TFHE_PRODUCT_STAGE1
        wfft64::butterfly_ww_to(re[s0], im[s0], re[s1], im[s1], k0, k1, zr[s0], zi[s0], zr[s1], zi[s1]);
#else
        zr[s0] = re[s0] * k0.x - im[s0] * k0.y;
        zi[s0] = re[s0] * k0.y + im[s0] * k0.x;
        zr[s1] = re[s1] * k1.x - im[s1] * k1.y;
        zi[s1] = re[s1] * k1.y + im[s1] * k1.x;
#endif
      });
    }
    RegisterSink64 sink{acc};
    wfft64::inv1024_twisted_real<false, WFFT64_TW_GROUP, TFHE_PRODUCT_STAGE1 != 0>(
        zr, zi, tile, tb, lane, [&](auto rc, double v) WFFT_INLINE { sink.template add<h>(rc, v, margin); },
        [](auto, double) {});
    {
      wfft64::Pipe16<4, TFHE_K64_DEPTH, cplx> kp;
      kp.prologue(load_b);
      kp.run2(load_b, [&](auto i0, const cplx& k0, auto i1, const cplx& k1) WFFT_INLINE {
        constexpr int s0 = decltype(i0)::value, s1 = decltype(i1)::value;
#if TFHE_PRODUCT_STAGE1
        wfft64::butterfly_ww<false>(re[s0], im[s0], re[s1], im[s1], k0, k1);
#else
        wfft::cmul2<false>(re[s0], im[s0], k0, re[s1], im[s1], k1);
#endif
      });
    }
    wfft64::inv1024_twisted_real<false, WFFT64_TW_GROUP, TFHE_PRODUCT_STAGE1 != 0>(
        re, im, tile, tb, lane, [&](auto rc, double v) WFFT_INLINE { sink.template add<1 - h>(rc, v, margin); },
        [](auto, double) {});
  });
}

// t = X^abar (prev + acc) - acc for polynomials 0 and 1 (src/bootstrapping.rs:136-146,
// src/polynomial.rs:281-313), XORed with mask (see tt_mask) and packed into tt; then prev <- acc.
// prev_lds: 8 KiB of this wavefront's LDS share.  Between iterations it holds the previous
// accumulator as lane-private 16-byte granules ((h, g) at [(4 h + g) * 64 + lane]); during the
// rotation it stages prev + acc in coefficient order for the gather.  A wavefront's LDS
// operations execute in program order, so the phases need no wait between them.
// have_prev = false stands for the all-zero previous accumulator a blind rotation starts with.
// GENERAL as in rotate_diff.
template <bool GENERAL, class TT>
__device__ __forceinline__ void rotate_diff64(const int32_t (&acc)[2][16], int abar, char* prev_lds, bool have_prev,
                                              int lane, uint32_t mask, TT& tt) {
  typedef int int4v __attribute__((ext_vector_type(4)));
  int32_t* S = reinterpret_cast<int32_t*>(prev_lds);               // [2][N]: prev + acc
  int4v* pv = reinterpret_cast<int4v*>(prev_lds) + lane;
  int32_t v[2][16];
  wfft::wave_lds_fence();
  if (have_prev) {
    static_for<0, 8>([&](auto ic) WFFT_INLINE {
      constexpr int h = decltype(ic)::value >> 2, g = decltype(ic)::value & 3;
      const int4v x = pv[(4 * h + g) * 64];
      v[h][4 * g] = x.x; v[h][4 * g + 1] = x.y; v[h][4 * g + 2] = x.z; v[h][4 * g + 3] = x.w;
    });
  } else {
    static_for<0, 32>([&](auto ic) WFFT_INLINE { v[decltype(ic)::value >> 4][decltype(ic)::value & 15] = 0; });
  }
  wfft::wave_lds_fence();
  static_for<0, 32>([&](auto ic) WFFT_INLINE {
    constexpr int h = decltype(ic)::value >> 4, r = decltype(ic)::value & 15;
    S[h * kN + lane + 64 * r] = wadd(v[h][r], acc[h][r]);
  });
  wfft::wave_lds_fence();
  // coefficient n comes from (n - pw) mod N, negated as rotate_diff explains
  int base, wrap = 0;
  bool cycle = false;
  if constexpr (GENERAL) {
    const MonomialPower mp = monomial_power(abar);
    base = lane - mp.pw;
    cycle = mp.cycle;
  } else {
    base = lane - abar;
    wrap = abar > 0 ? base : (abar == -kN ? -(1 << 20) : (1 << 20));
  }
  static_for<0, 32>([&](auto ic) WFFT_INLINE {
    constexpr int h = decltype(ic)::value >> 4, r = decltype(ic)::value & 15;
    v[h][r] = S[h * kN + ((base + 64 * r) & (kN - 1))];
  });
  wfft::wave_lds_fence();
  static_for<0, 8>([&](auto ic) WFFT_INLINE {
    constexpr int h = decltype(ic)::value >> 2, g = decltype(ic)::value & 3;
    pv[(4 * h + g) * 64] = int4v{acc[h][4 * g], acc[h][4 * g + 1], acc[h][4 * g + 2], acc[h][4 * g + 3]};
  });
  wfft::wave_lds_fence();
  static_for<0, 16>([&](auto rc) WFFT_INLINE {
    constexpr int r = decltype(rc)::value;
    uint32_t x[2];
    static_for<0, 2>([&](auto hc) WFFT_INLINE {
      constexpr int h = decltype(hc)::value;
      const bool neg = GENERAL ? ((base < -64 * r) != cycle) : (wrap < -64 * r);
      const int32_t y = neg ? wneg(v[h][r]) : v[h][r];
      x[h] = (uint32_t)wsub(y, acc[h][r]) ^ mask;
    });
    tt.template put<r>(x[0], x[1]);
  });
}

template <int L, int BGBIT, bool HOOK>
__global__ __launch_bounds__(64 * kWaves64) void blind_rotate64_kernel(BlindRotateArgs args) {
  __shared__ __attribute__((aligned(16))) char lds[kLds64Bytes];
  const wfft64::Tables tb = load_tables64(lds + kWaves64 * kWave64Bytes, args.tables64);
  const int lane0 = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  char* tile = lds + wave * kWave64Bytes;
  char* prev_lds = tile + wfft64::kTileBytes;
  // persistent workgroups: wavefront slot s takes ciphertexts s, s + slots, ...  Slots are numbered
  // wavefront-major (all workgroups' wavefront 0 first), so that a batch smaller than a full round
  // spreads over all compute units, one wavefront per SIMD first, instead of filling a few of them.
  const int slot = wave * gridDim.x + blockIdx.x;
  const int slots = gridDim.x * kWaves64;
  NoMargin no_margin;

#pragma unroll 1
  for (int b = slot; b < args.B; b += slots) {
    int32_t acc[2][16];
    static_for<0, 32>([&](auto ic) WFFT_INLINE {
      constexpr int h = decltype(ic)::value >> 4, r = decltype(ic)::value & 15;
      acc[h][r] = HOOK ? args.acc_io[((size_t)b * 3 + h) * kN + lane0 + 64 * r] : 0;
    });
    bool have_prev = false;                                       // prev = 0, bootstrapping.rs:116-117

    const int32_t* bara = args.bara + (size_t)b * args.n;
    int abar64 = 0;          // the next 64 rotation amounts, one per lane
#pragma unroll 1
    for (int i = 0; i < args.n; i++) {
      if ((i & 63) == 0) abar64 = (i + lane0 < args.n) ? bara[i + lane0] : 0;
#if TFHE_ITER_BARRIER
      // the wavefronts of a workgroup walk the key together (wavefronts that have finished do
      // not count for the barrier)
      if ((i & (TFHE_ITER_BARRIER - 1)) == 0) __builtin_amdgcn_s_barrier();
#endif
      const int abar = __builtin_amdgcn_readlane(abar64, i & 63);
      if (abar == 0) continue;                                   // bootstrapping.rs:122-126
      int lane = lane0;
      asm volatile("" : "+v"(lane));
      PackedSample<L, BGBIT> tt;
      rotate_diff64<HOOK>(acc, abar, prev_lds, have_prev, lane, tt_mask<L, BGBIT>(), tt);   // also prev <- acc
      have_prev = true;
      const cplx* bk_i = args.bkfft + (size_t)i * L * 4 * 1024;
#pragma unroll 1
      for (int p = 0; p < L; p++)
        level_product64<BGBIT>(p, tt, bk_i + (size_t)p * 4 * 1024, acc, tile, tb, lane, no_margin);
    }

    if (HOOK) {
      static_for<0, 32>([&](auto ic) WFFT_INLINE {
        constexpr int h = decltype(ic)::value >> 4, r = decltype(ic)::value & 15;
        args.acc_io[((size_t)b * 3 + h) * kN + lane0 + 64 * r] = acc[h][r];
      });
    } else {
      // extract_lwe, src/tlwe.rs:159-180 (k = 1): a[0] = acc0[0], a[j] = -acc0[N-j].
      int32_t* oa = args.out_a + (size_t)b * kN;
      static_for<0, 16>([&](auto ic) WFFT_INLINE {
        constexpr int r = decltype(ic)::value;
        const int t = lane0 + 64 * r;
        if (t == 0) oa[0] = acc[0][r];
        else oa[kN - t] = wneg(acc[0][r]);
      });
      if (lane0 == 0) {
        const int32_t barb = args.barb[b];                       // bootstrapping.rs:58-96
        args.out_b[b] = barb >= 0 ? args.mu : wneg(args.mu);
      }
    }
  }
}

// ---------------------------------------------------------------------------
// Low-latency mapping on the transforms of wave_fft64.hpp: one workgroup per ciphertext, one
// WAVEFRONT per (digit polynomial, key polynomial) product of tgsw_extern_mul_to_tlwe
// (src/tgsw.rs:229-264): wavefront u 2 l + (2 p + h) multiplies the spectrum D of the level-p digits
// of accumulator polynomial h by K_{h,u} and, after the inverse transform, adds its rounded product
// into the shared accumulator in LDS (ds_add_u32; each product rounded on its own).  The two
// wavefronts of a pair (p, h) need the same D: the one with u = 0 decomposes and transforms (numbered
// so that these 2 l wavefronts spread over all SIMDs and run the forward transform with a SIMD to
// themselves) and hands D to its partner through LDS, in the two wavefronts' exchange tiles, which
// are idle between a forward transform's last and an inverse transform's first transposition.
// An iteration costs the latency of one forward and one inverse transform.  The accumulator, the
// previous accumulator and the sample being decomposed are shared in LDS ([2][N] int32 each, the
// sample XORed with tt_mask).  Two workgroup barriers per iteration (accumulator complete, sample
// complete); the handover is ordered inside a pair by two counters in LDS, "spectrum written" and
// "spectrum read" (the wavefront that computed it waits for the latter before its inverse transform
// writes to its tile again; by then its partner has long read, so neither wait costs a workgroup
// barrier's alignment of all wavefronts: 6 - 8 % of a gate's latency).
// ---------------------------------------------------------------------------
template <int L>
constexpr int split64_lds_bytes() {
  return 4 * L * wfft64::kTileBytes + 3 * 2 * kN * 4 + wfft64::kTableBytes + 2 * 2 * L * 4;
}
// counters in LDS between the wavefronts of a pair; a wavefront's LDS operations are served in
// program order, so a counter written after the data (or after the reads) is seen after them
__device__ __forceinline__ void lds_counter_set(int* c, int v) {
  __hip_atomic_store(c, v, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
}
__device__ __forceinline__ void lds_counter_wait(int* c, int v) {
  while (__hip_atomic_load(c, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP) != v) __builtin_amdgcn_s_sleep(1);
}

template <int L, int BGBIT, bool HOOK>
__global__ __launch_bounds__(64 * 4 * L) void blind_rotate_split64_kernel(BlindRotateArgs args) {
  constexpr int W = 4 * L;
  __shared__ __attribute__((aligned(16))) char lds[split64_lds_bytes<L>()];
  char* shared = lds + W * wfft64::kTileBytes;
  int32_t* prevL = reinterpret_cast<int32_t*>(shared);
  int32_t* accL = prevL + 2 * kN;
  int32_t* ttL = accL + 2 * kN;
  int* counters = reinterpret_cast<int*>(shared + 3 * 2 * kN * 4 + wfft64::kTableBytes);
  if (threadIdx.x < 2 * 2 * L) counters[threadIdx.x] = 0;       // before the barrier in load_tables64
  const wfft64::Tables tb = load_tables64(shared + 3 * 2 * kN * 4, args.tables64);
  const int lane0 = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int pair = wave % (2 * L), u = wave / (2 * L), p = pair >> 1, h = pair & 1;
  int* written = counters + pair;                              // iterations whose spectrum is in LDS
  int* consumed = counters + 2 * L + pair;                     // iterations whose spectrum has been read
  int done = 0;                                                // iterations run (those with abar = 0 are skipped)
  char* tile = lds + wave * wfft64::kTileBytes;
  // where a pair's spectrum changes hands: values 0..8 of a lane in the tile of the wavefront that
  // computes it, 9..15 in its partner's; [value][re | im][lane] f64
  constexpr int kInFirstTile = wfft64::kTileBytes / 1024;
  static_assert(kInFirstTile * 1024 == wfft64::kTileBytes && 16 - kInFirstTile <= kInFirstTile, "spectrum in two tiles");
  char* xchg0 = lds + pair * wfft64::kTileBytes;
  char* xchg1 = lds + (pair + 2 * L) * wfft64::kTileBytes;
  const int b = blockIdx.x;
  constexpr uint32_t kMaskTt = tt_mask<L, BGBIT>();
  const int decal = 32 - (p + 1) * BGBIT;

  for (int x = threadIdx.x; x < 2 * kN; x += 64 * W) {
    prevL[x] = 0;                                              // bootstrapping.rs:116-117
    accL[x] = HOOK ? args.acc_io[(size_t)b * 3 * kN + x] : 0;
  }

  const int32_t* bara = args.bara + (size_t)b * args.n;
  int abar64 = 0;
#pragma unroll 1
  for (int i = 0; i < args.n; i++) {
    if ((i & 63) == 0) abar64 = (i + lane0 < args.n) ? bara[i + lane0] : 0;
    const int abar = __builtin_amdgcn_readlane(abar64, i & 63);
    if (abar == 0) continue;                                   // bootstrapping.rs:122-126
    done++;
    __syncthreads();                                           // acc of the last iteration is complete
    int lane = lane0;
    asm volatile("" : "+v"(lane));                             // keeps per-lane addresses out of the loop's live-in set
    // t = X^abar (prev + acc) - acc (src/bootstrapping.rs:136-146, src/polynomial.rs:281-313), the
    // 2 N coefficients spread over the workgroup's threads; then prev <- acc.
    {
      const MonomialPower mp = monomial_power(abar);         // polynomial.rs:289-291
      constexpr int kSteps = (2 * kN + 64 * W - 1) / (64 * W);
      int32_t keep[kSteps];
      static_for<0, kSteps>([&](auto ec) WFFT_INLINE {
        constexpr int e = decltype(ec)::value;
        const int x = (int)threadIdx.x + 64 * W * e;           // index into [2][N]
        if (x < 2 * kN) {
          const int hh = x >> 10, n = x & (kN - 1);
          const int src = (n - mp.pw) & (kN - 1);
          const int32_t v = wadd(prevL[hh * kN + src], accL[hh * kN + src]);
          const int32_t a = accL[x];
          const int32_t y = ((n < mp.pw) != mp.cycle) ? wneg(v) : v;
          ttL[x] = (int32_t)((uint32_t)wsub(y, a) ^ kMaskTt);
          keep[e] = a;
        }
      });
      __syncthreads();                                         // tt complete, prev no longer read
      static_for<0, kSteps>([&](auto ec) WFFT_INLINE {
        constexpr int e = decltype(ec)::value;
        const int x = (int)threadIdx.x + 64 * W * e;
        if (x < 2 * kN) prevL[x] = keep[e];
      });
    }
    double re[16], im[16];
    // K_{h,u}: block 0 of the level holds K_{h,h}, block 1 K_{h,1-h}, in the lane layout that
    // level_product64 explains; the first loads go out before the forward transform (a single
    // ciphertext's key comes from HBM, not from L2)
    const int key_lane = (((lane >> 3) & 1) * 16) * 64 + h * 32 + (((lane >> 4) << 2) | (lane & 3)) +
                         16 * ((lane >> 2) & 1);
    const cplx* kcol = args.bkfft + ((size_t)i * L + p) * 4 * 1024 + (h == u ? 0 : 2048) + key_lane;
    auto load_k = [&](auto ic) WFFT_INLINE { return kcol[decltype(ic)::value * 64]; };
    wfft64::Pipe16<4, TFHE_SPLIT_KEY_DEPTH, cplx> kp;
    kp.prologue(load_k);
    __builtin_amdgcn_sched_barrier(0);
    if (u == 0) {
      static_for<0, 16>([&](auto rc) WFFT_INLINE {
        constexpr int r = decltype(rc)::value;
        re[r] = (double)(int32_t)__builtin_amdgcn_sbfe(ttL[h * kN + lane + 64 * r], decal, BGBIT);
      });
      wfft64::fwd1024_twisted_real(re, im, tile, tb, lane);
      static_for<0, 16>([&](auto sc) WFFT_INLINE {
        constexpr int v = decltype(sc)::value;
        char* at = (v < kInFirstTile ? xchg0 + v * 1024 : xchg1 + (v - kInFirstTile) * 1024) + lane * 8;
        *reinterpret_cast<double*>(at) = re[v];
        *reinterpret_cast<double*>(at + 512) = im[v];
      });
      lds_counter_set(written, done);
    } else {
      lds_counter_wait(written, done);
      static_for<0, 16>([&](auto sc) WFFT_INLINE {
        constexpr int v = decltype(sc)::value;
        const char* at = (v < kInFirstTile ? xchg0 + v * 1024 : xchg1 + (v - kInFirstTile) * 1024) + lane * 8;
        re[v] = *reinterpret_cast<const double*>(at);
        im[v] = *reinterpret_cast<const double*>(at + 512);
      });
      lds_counter_set(consumed, done);
    }
    kp.run2(load_k, [&](auto i0, const cplx& k0, auto i1, const cplx& k1) WFFT_INLINE {
      constexpr int s0 = decltype(i0)::value, s1 = decltype(i1)::value;
#if TFHE_PRODUCT_STAGE1
      wfft64::butterfly_ww<false>(re[s0], im[s0], re[s1], im[s1], k0, k1);
#else
      wfft::cmul2<false>(re[s0], im[s0], k0, re[s1], im[s1], k1);
#endif
    });
    wfft64::inv1024_twisted_real<false, WFFT64_TW_GROUP, TFHE_PRODUCT_STAGE1 != 0>(
        re, im, tile, tb, lane,
        [&](auto rc, double v) WFFT_INLINE {
          constexpr int r = decltype(rc)::value;
          atomicAdd(reinterpret_cast<unsigned*>(accL) + u * kN + lane + 64 * r, (unsigned)wfft::round_sat_i32(v));
        },
        [](auto, double) {},
        [&]() WFFT_INLINE {
          if (u == 0) lds_counter_wait(consumed, done);        // the partner has read what this tile held
        });
  }
  __syncthreads();

  if (HOOK) {
    for (int x = threadIdx.x; x < 2 * kN; x += 64 * W) args.acc_io[(size_t)b * 3 * kN + x] = accL[x];
  } else {
    // extract_lwe, src/tlwe.rs:159-180 (k = 1): a[0] = acc0[0], a[t] = -acc0[N-t].
    int32_t* oa = args.out_a + (size_t)b * kN;
    for (int t = threadIdx.x; t < kN; t += 64 * W) oa[t] = t == 0 ? accL[0] : wneg(accL[kN - t]);
    if (threadIdx.x == 0) {
      const int32_t barb = args.barb[b];                       // bootstrapping.rs:58-96
      args.out_b[b] = barb >= 0 ? args.mu : wneg(args.mu);
    }
  }
}

#ifdef TFHE_TEXTBOOK
// ===========================================================================
// The textbook-correct variant (libtfhe_hip_textbook.so, built with -DTFHE_TEXTBOOK; the
// library without that flag contains none of this).  The same path with the crate's
// deviations from TFHE corrected where the crate has them:
//   * gadget h[p] = 2^(32 - (p+1) bg_bit) on every level, decomposition offset over all levels
//     (src/tgsw.rs:43-52), every TGSW row a TLWE encryption of zero (src/tgsw.rs:118-127),
//     message * h[p] on polynomial j of row (p, j) (src/tgsw.rs:129-167);
//   * TLWE samples of k mask polynomials and a body, products negacyclic mod X^N + 1 and not
//     saturated (src/tlwe.rs:86,112-134, src/numerics.rs:138-155,185-261);
//   * the external product decomposes and updates the body too (src/tgsw.rs:229-297);
//   * acc <- acc + bk_i [.] (X^abar acc - acc) with abar, bbar in [0, 2N)
//     (src/bootstrapping.rs:50-55,109-146);
//   * key switch over N k coefficients (src/lwe.rs:458-461): n_ks = N k in the parameters;
//   * NOR bootstraps with MU = +1/8 (src/gates.rs:136 uses -1/8, which inverts the gate).
// The products of one external product are not rounded one by one any more, so they are summed
// in the frequency domain: (k+1) l forward and k+1 inverse transforms per iteration instead of
// (k+1) l and (k+1)^2 l.  The negacyclic product is the low half of the linear convolution
// minus the high half, both of which the twisted inverse transform delivers.  Exactness: the
// summed coefficients stay below 2^50, where the transforms' error is of the order 0.05, far
// from the 0.5 that would move a rounding; the result is then reduced modulo 2^32.
//
// One wavefront per ciphertext (wave_fft64.hpp transforms).
// Key: [n][l][j][u][16][64] complex, the spectrum (scaled by 1/1024) of polynomial u of row
// (level, component j) of bk[i].
// ===========================================================================
template <int L, int BGBIT>
constexpr uint32_t tb_offset() {           // sum over the levels of Bg/2 * 2^(32 - (p+1) bg_bit)
  uint32_t o = 0;
  for (int p = 0; p < L; p++) o += (1u << (BGBIT - 1)) << (32 - (p + 1) * BGBIT);
  return o;
}
template <int L, int BGBIT>
constexpr uint32_t tb_field_tops() {       // the top bit of every digit field: field - Bg/2 = sign-extended (field ^ top)
  uint32_t m = 0;
  for (int p = 0; p < L; p++) m |= 1u << (31 - p * BGBIT);
  return m;
}

// x mod 2^32 of the integer nearest to x (ties to even), exact for |x| < 2^51
__device__ __forceinline__ int32_t round_wrap_i32(double x) {
  return (int32_t)(uint32_t)(unsigned long long)__double_as_longlong(x + 6755399441055744.0);
}

// Two wavefronts per SIMD, as blind_rotate64_kernel, and the same split of a wavefront's state:
// the two result spectra take 128 of its 256 registers, the sample being decomposed (with the
// offset added and the top bit of every digit field flipped, so that digit p is a sign-extended bit
// field) is packed into 20 - 22 more (PackedSample), and the accumulator, touched at both ends of an
// iteration only, lives in 8 KiB of the wavefront's LDS share as lane-private 16-byte granules
// ((h, g) at [(4 h + g) * 64 + lane]); the rotation stages it there in coefficient order for its
// gather.  Global memory is read for the key and written for the result.
typedef int tb_int4 __attribute__((ext_vector_type(4)));

template <int L, int BGBIT>
__global__ __launch_bounds__(64 * kWaves64) void blind_rotate_textbook_kernel(BlindRotateArgs args) {
  __shared__ __attribute__((aligned(16))) char lds[kLds64Bytes];
  const wfft64::Tables tb = load_tables64(lds + kWaves64 * kWave64Bytes, args.tables64);
  const int lane0 = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  char* tile = lds + wave * kWave64Bytes;
  char* acc_lds = tile + wfft64::kTileBytes;
  const int slot = wave * gridDim.x + blockIdx.x, slots = gridDim.x * kWaves64;      // as blind_rotate64_kernel

#pragma unroll 1
  for (int b = slot; b < args.B; b += slots) {
    {
      // acc = trivial(X^(2N - barb) [mu]*N), src/bootstrapping.rs:89-96: polynomial 0 (mask) = 0,
      // polynomial 1 (body) = the rotated test vector
      tb_int4* acc_g = reinterpret_cast<tb_int4*>(acc_lds) + lane0;
      const int w0 = (2 * kN - args.barb[b]) & (2 * kN - 1);
      const bool flip = w0 >= kN;
      const int w = flip ? w0 - kN : w0;
      wfft::wave_lds_fence();
      static_for<0, 4>([&](auto gc) WFFT_INLINE {
        constexpr int g = decltype(gc)::value;
        acc_g[g * 64] = tb_int4{0, 0, 0, 0};
        tb_int4 v;
        v.x = ((lane0 + 64 * (4 * g + 0) < w) != flip) ? wneg(args.mu) : args.mu;
        v.y = ((lane0 + 64 * (4 * g + 1) < w) != flip) ? wneg(args.mu) : args.mu;
        v.z = ((lane0 + 64 * (4 * g + 2) < w) != flip) ? wneg(args.mu) : args.mu;
        v.w = ((lane0 + 64 * (4 * g + 3) < w) != flip) ? wneg(args.mu) : args.mu;
        acc_g[(4 + g) * 64] = v;
      });
    }
    const int32_t* bara = args.bara + (size_t)b * args.n;
    int abar64 = 0;
#pragma unroll 1
    for (int i = 0; i < args.n; i++) {
      if ((i & 63) == 0) abar64 = (i + lane0 < args.n) ? bara[i + lane0] : 0;
#if TFHE_ITER_BARRIER
      if ((i & (TFHE_ITER_BARRIER - 1)) == 0) __builtin_amdgcn_s_barrier();      // as blind_rotate64_kernel
#endif
      const int abar = __builtin_amdgcn_readlane(abar64, i & 63);
      if (abar == 0) continue;
      int lane = lane0;
      asm volatile("" : "+v"(lane));
      tb_int4* acc_g = reinterpret_cast<tb_int4*>(acc_lds) + lane;
      // t = X^abar acc - acc with the decomposition offset added and the top bit of every digit
      // field flipped, packed into registers
      PackedSample<L, BGBIT> tt;
      {
        int32_t acc[2][16], v[2][16];
        wfft::wave_lds_fence();
        static_for<0, 8>([&](auto ic) WFFT_INLINE {
          constexpr int h = decltype(ic)::value >> 2, g = decltype(ic)::value & 3;
          const tb_int4 x = acc_g[(4 * h + g) * 64];
          acc[h][4 * g] = x.x; acc[h][4 * g + 1] = x.y; acc[h][4 * g + 2] = x.z; acc[h][4 * g + 3] = x.w;
        });
        int32_t* S = reinterpret_cast<int32_t*>(acc_lds);
        wfft::wave_lds_fence();
        static_for<0, 32>([&](auto ic) WFFT_INLINE {
          constexpr int h = decltype(ic)::value >> 4, r = decltype(ic)::value & 15;
          S[h * kN + lane + 64 * r] = acc[h][r];
        });
        wfft::wave_lds_fence();
        const bool flip = abar >= kN;
        const int w = flip ? abar - kN : abar;
        static_for<0, 32>([&](auto ic) WFFT_INLINE {
          constexpr int h = decltype(ic)::value >> 4, r = decltype(ic)::value & 15;
          v[h][r] = S[h * kN + ((lane + 64 * r - w) & (kN - 1))];
        });
        wfft::wave_lds_fence();
        static_for<0, 8>([&](auto ic) WFFT_INLINE {                  // the accumulator back as granules
          constexpr int h = decltype(ic)::value >> 2, g = decltype(ic)::value & 3;
          acc_g[(4 * h + g) * 64] = tb_int4{acc[h][4 * g], acc[h][4 * g + 1], acc[h][4 * g + 2], acc[h][4 * g + 3]};
        });
        wfft::wave_lds_fence();
        static_for<0, 16>([&](auto rc) WFFT_INLINE {
          constexpr int r = decltype(rc)::value;
          uint32_t x[2];
          static_for<0, 2>([&](auto hc) WFFT_INLINE {
            constexpr int h = decltype(hc)::value;
            const bool neg = (lane + 64 * r < w) != flip;
            const int32_t y = neg ? wneg(v[h][r]) : v[h][r];
            x[h] = ((uint32_t)wsub(y, acc[h][r]) + tb_offset<L, BGBIT>()) ^ tb_field_tops<L, BGBIT>();
          });
          tt.template put<r>(x[0], x[1]);
        });
      }
      // spectra of the two result polynomials, summed over the (k+1) l rows
      double ur[2][16], ui[2][16];
      static_for<0, 32>([&](auto ic) WFFT_INLINE {
        ur[decltype(ic)::value >> 4][decltype(ic)::value & 15] = 0.0;
        ui[decltype(ic)::value >> 4][decltype(ic)::value & 15] = 0.0;
      });
      const cplx* bk_i = args.bkfft + (size_t)i * L * 4 * 1024 + lane;
#pragma unroll 1
      for (int p = 0; p < L; p++) {
        const int decal = 32 - (p + 1) * BGBIT;
        static_for<0, 2>([&](auto jc) WFFT_INLINE {                   // row = 2 p + j: level p, component j
          constexpr int j = decltype(jc)::value;
          double re[16], im[16];
          static_for<0, 16>([&](auto rc) WFFT_INLINE {
            constexpr int r = decltype(rc)::value;
            re[r] = (double)(int32_t)__builtin_amdgcn_sbfe(tt.template get<j, r>(), decal, BGBIT);
          });
          wfft64::fwd1024_twisted_real<WFFT64_TW_GROUP, false>(re, im, tile, tb, lane);
          const cplx* krow = bk_i + (size_t)(2 * p + j) * 2 * 1024;
          static_for<0, 2>([&](auto uc) WFFT_INLINE {
            constexpr int u = decltype(uc)::value;
            auto load_k = [&](auto ic) WFFT_INLINE { return krow[u * 1024 + decltype(ic)::value * 64]; };
            wfft64::Pipe16<4, TFHE_K64_DEPTH, cplx> kp;
            kp.prologue(load_k);
            kp.run2(load_k, [&](auto i0, const cplx& k0, auto i1, const cplx& k1) WFFT_INLINE {
              constexpr int s0 = decltype(i0)::value, s1 = decltype(i1)::value;
              ur[u][s0] += re[s0] * k0.x - im[s0] * k0.y;
              ui[u][s0] += re[s0] * k0.y + im[s0] * k0.x;
              ur[u][s1] += re[s1] * k1.x - im[s1] * k1.y;
              ui[u][s1] += re[s1] * k1.y + im[s1] * k1.x;
            });
          });
        });
      }
      static_for<0, 2>([&](auto uc) WFFT_INLINE {
        constexpr int u = decltype(uc)::value;
        double lo_v[16];
        wfft64::inv1024_twisted_real<true, WFFT64_TW_GROUP, false, false>(
            ur[u], ui[u], tile, tb, lane, [&](auto rc, double v) WFFT_INLINE { lo_v[decltype(rc)::value] = v; },
            [&](auto rc, double v) WFFT_INLINE { ui[u][decltype(rc)::value] = v; });
        // negacyclic product = low half - high half of the convolution; acc[u] += it, modulo 2^32
        wfft::wave_lds_fence();
        static_for<0, 4>([&](auto gc) WFFT_INLINE {
          constexpr int g = decltype(gc)::value;
          tb_int4 x = acc_g[(4 * u + g) * 64];
          x.x = wadd(x.x, round_wrap_i32(lo_v[4 * g + 0] - ui[u][4 * g + 0]));
          x.y = wadd(x.y, round_wrap_i32(lo_v[4 * g + 1] - ui[u][4 * g + 1]));
          x.z = wadd(x.z, round_wrap_i32(lo_v[4 * g + 2] - ui[u][4 * g + 2]));
          x.w = wadd(x.w, round_wrap_i32(lo_v[4 * g + 3] - ui[u][4 * g + 3]));
          acc_g[(4 * u + g) * 64] = x;
        });
        wfft::wave_lds_fence();
      });
    }
    // extract_lwe, src/tlwe.rs:159-180 (k = 1): a[0] = acc0[0], a[j] = -acc0[N-j], b = body[0]
    const tb_int4* acc_g = reinterpret_cast<const tb_int4*>(acc_lds) + lane0;
    int32_t* oa = args.out_a + (size_t)b * kN;
    wfft::wave_lds_fence();
    static_for<0, 4>([&](auto gc) WFFT_INLINE {
      constexpr int g = decltype(gc)::value;
      const tb_int4 x = acc_g[g * 64];
      const int32_t c[4] = {x.x, x.y, x.z, x.w};
      static_for<0, 4>([&](auto ec) WFFT_INLINE {
        constexpr int r = 4 * g + decltype(ec)::value;
        const int t = lane0 + 64 * r;
        if (t == 0) oa[0] = c[decltype(ec)::value];
        else oa[kN - t] = wneg(c[decltype(ec)::value]);
      });
    });
    const tb_int4 body0 = acc_g[4 * 64];
    if (lane0 == 0) args.out_b[b] = body0.x;
    wfft::wave_lds_fence();
  }
}

constexpr int kTbWaves = kWaves64;
constexpr int kTbLdsBytes = kLds64Bytes;

// One-time transform of the bootstrapping key: wavefront s = ((i l + p) 2 + j) 2 + u writes the
// spectrum of polynomial u of row (p, j) of bk[i] (flat input [i][p][j][u][N]), scaled by 1/1024.
__global__ __launch_bounds__(64 * kTbWaves) void bk_fft_textbook_kernel(const int32_t* __restrict__ bk,
                                                                            const cplx* __restrict__ tables64,
                                                                            cplx* __restrict__ bkfft, int count) {
  __shared__ __attribute__((aligned(16))) char lds[kTbLdsBytes];
  const wfft64::Tables tb = load_tables64(lds + kWaves64 * kWave64Bytes, tables64);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int s = blockIdx.x * kTbWaves + wave;
  if (s >= count) return;
  const int32_t* p0 = bk + (size_t)s * kN + lane;
  double re[16], im[16];
  static_for<0, 16>([&](auto ic) WFFT_INLINE { re[decltype(ic)::value] = (double)p0[64 * decltype(ic)::value]; });
  wfft64::fwd1024_twisted_real(re, im, lds + wave * kWave64Bytes, tb, lane);
  cplx* dst = bkfft + (size_t)s * 1024 + lane;
  static_for<0, 16>([&](auto ic) WFFT_INLINE {
    constexpr int p = decltype(ic)::value;
    cplx v;
    v.x = re[p] * (1.0 / 1024.0);
    v.y = im[p] * (1.0 / 1024.0);
    dst[p * 64] = v;
  });
}
#endif  // TFHE_TEXTBOOK

// Gate prologue + mod switch: x = C + wa*ca + wb*cb (wrapping, src/gates.rs,
// src/lwe.rs:47-136), then bara/barb = decode_message(x, 2N)
// (src/bootstrapping.rs:50-55).
__global__ void prologue_kernel(const int32_t* __restrict__ ca_a, const int32_t* __restrict__ ca_b,
                                const int32_t* __restrict__ cb_a, const int32_t* __restrict__ cb_b,
                                int wa, int wb, int32_t c, int n, size_t total,
                                int32_t* __restrict__ bara, int32_t* __restrict__ barb) {
  const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= total) return;
  const size_t b = idx / (size_t)(n + 1);
  const int i = (int)(idx - b * (size_t)(n + 1));
  if (i < n) {
    uint32_t x = (uint32_t)wa * (uint32_t)ca_a[b * n + i];
    if (cb_a) x += (uint32_t)wb * (uint32_t)cb_a[b * n + i];
    bara[b * n + i] = mod_switch_2n((int32_t)x);
  } else {
    uint32_t x = (uint32_t)c + (uint32_t)wa * (uint32_t)ca_b[b];
    if (cb_b) x += (uint32_t)wb * (uint32_t)cb_b[b];
    barb[b] = mod_switch_2n((int32_t)x);
  }
}

// lwe_key_switch (src/lwe.rs:601-648).  One workgroup per ciphertext.  Thread
// (g, v) owns four consecutive output coefficients (the body b is stored right
// after the mask in each key row) and walks the input coefficients i = g, g + G,
// ...; the G partial sums meet in LDS (wrapping adds commute).  The T loads of
// one coefficient are in flight together; digit 0, which the reference skips
// (lwe.rs:638), is masked off (and its rows and variances are zero in the
// packed key, so that the variance sum needs no branch).  in2 (optional) and c_b implement the sum
// trivial(c) + u1 + u2 of the mux gate (src/gates.rs:264-266).
constexpr int kKsBlock = 512;

template <int T>   // T = ks_t when known at compile time, 0 = run-time t
__global__ __launch_bounds__(kKsBlock) void keyswitch_kernel(
    const int32_t* __restrict__ in_a, const int32_t* __restrict__ in_b,
    const int32_t* __restrict__ in2_a, const int32_t* __restrict__ in2_b, int32_t c_b,
    int in_stride, const int32_t* __restrict__ ks_rows, const double* __restrict__ ks_var,
    int row_stride, int n, int n_ks, int t_rt, int32_t* __restrict__ out_a,
    int32_t* __restrict__ out_b, double* __restrict__ var_out) {
  __shared__ int32_t aibar_s[kN];
  __shared__ int4 part[kKsBlock];
  const int t = T ? T : t_rt;
  const int b = blockIdx.x;
  const int nvec = row_stride / 4;
  const int tpg = (nvec + 31) / 32 * 32;            // threads per group
  const int G = kKsBlock / tpg;
  const int g = threadIdx.x / tpg, v = threadIdx.x - g * tpg;
  const int32_t* xa = in_a + (size_t)b * in_stride;
  const int32_t* xa2 = in2_a ? in2_a + (size_t)b * in_stride : nullptr;
  const int32_t prec_offset = 1 << (32 - (1 + 2 * t));
  for (int i = threadIdx.x; i < n_ks; i += kKsBlock) {
    int32_t ai = xa[i];
    if (xa2) ai = wadd(ai, xa2[i]);
    aibar_s[i] = wadd(ai, prec_offset);
  }
  __syncthreads();
  uint32_t a0 = 0, a1 = 0, a2 = 0, a3 = 0;
  if (g < G && v < nvec) {
    const int32_t* col = ks_rows + 4 * v;
    for (int i = g; i < n_ks; i += G) {
      const int32_t aibar = aibar_s[i];
      if (T) {
        int4 row[T ? T : 1];
#pragma unroll
        for (int j = 0; j < T; j++) {
          const int aij = (aibar >> (32 - (j + 1) * 2)) & 3;
          row[j] = make_int4(0, 0, 0, 0);
          if (aij != 0)      // digit 0 costs no memory traffic
            row[j] = *reinterpret_cast<const int4*>(col + ((size_t)(i * T + j) * 4 + aij) * row_stride);
        }
#pragma unroll
        for (int j = 0; j < T; j++) {
          a0 -= (uint32_t)row[j].x; a1 -= (uint32_t)row[j].y; a2 -= (uint32_t)row[j].z; a3 -= (uint32_t)row[j].w;
        }
      } else {
        for (int j = 0; j < t; j++) {
          const int aij = (aibar >> (32 - (j + 1) * 2)) & 3;
          const int4 row = *reinterpret_cast<const int4*>(col + ((size_t)(i * t + j) * 4 + aij) * row_stride);
          a0 -= (uint32_t)row.x; a1 -= (uint32_t)row.y; a2 -= (uint32_t)row.z; a3 -= (uint32_t)row.w;
        }
      }
    }
  }
  part[threadIdx.x] = make_int4((int)a0, (int)a1, (int)a2, (int)a3);
  __syncthreads();
  if (g == 0 && v < nvec) {
    for (int k = 1; k < G; k++) {
      const int4 o = part[k * tpg + v];
      a0 += (uint32_t)o.x; a1 += (uint32_t)o.y; a2 += (uint32_t)o.z; a3 += (uint32_t)o.w;
    }
    const uint32_t acc4[4] = {a0, a1, a2, a3};
    for (int e = 0; e < 4; e++) {
      const int x = 4 * v + e;
      if (x < n) out_a[(size_t)b * n + x] = (int32_t)acc4[e];
      else if (x == n) {
        int32_t bb = in_b[b];
        if (in2_b) bb = wadd(wadd(c_b, bb), in2_b[b]);
        out_b[b] = wadd(bb, (int32_t)acc4[e]);
      }
    }
  }
  if (var_out && threadIdx.x == kKsBlock - 1) {
    // summed in the reference's order (lwe.rs:109); the entries of digit 0 are +0.0
    double var = 0.0;
    for (int i = 0; i < n_ks; i++) {
      const int32_t aibar = aibar_s[i];
      const double* vi = ks_var + (size_t)i * t * 4;
#pragma unroll 8
      for (int j = 0; j < t; j++) var += vi[j * 4 + ((aibar >> (32 - (j + 1) * 2)) & 3)];
    }
    var_out[b] = var;
  }
}

// lwe_key_switch for large batches: one workgroup takes kKsTile ciphertexts and
// walks the key once for all of them, so a key row is fetched once per tile
// instead of once per ciphertext (the gather is bound by L2 / Infinity Cache
// bandwidth, not by arithmetic).  For every (coefficient, digit position) the
// three rows of the non-zero digits are loaded and each ciphertext subtracts the
// one its digit selects; the digit is wave-uniform, so the selection is a scalar
// branch.  Thread layout and the order of the variance sum as in keyswitch_kernel.
// ciphertexts per workgroup; 16 and 32 halve and quarter the L2 reads but the compiler then keeps
// the partial sums in scratch memory (256 registers + 428 / 1 616 bytes per lane)
#ifndef TFHE_KS_TILE
#define TFHE_KS_TILE 8
#endif
constexpr int kKsTile = TFHE_KS_TILE;
#ifndef TFHE_KS_LDS_TILE
#define TFHE_KS_LDS_TILE 32
#endif
constexpr int kKsLdsTile = TFHE_KS_LDS_TILE;       // keyswitch_lds_kernel

template <int TC>   // TC = ks_t when known at compile time, 0 = run-time t_rt
__global__ __launch_bounds__(kKsBlock) void keyswitch_tile_kernel(
    const int32_t* __restrict__ in_a, const int32_t* __restrict__ in_b,
    const int32_t* __restrict__ in2_a, const int32_t* __restrict__ in2_b, int32_t c_b,
    int in_stride, const int32_t* __restrict__ ks_rows, const double* __restrict__ ks_var,
    int row_stride, int n, int n_ks, int t_rt, int B, int32_t* __restrict__ out_a,
    int32_t* __restrict__ out_b, double* __restrict__ var_out) {
  const int T = TC ? TC : t_rt;
  __shared__ int32_t aibar_s[kKsTile][kN];
  __shared__ int4 part[kKsBlock];
  const int b0 = blockIdx.x * kKsTile;
  const int nvec = row_stride / 4;
  const int tpg = (nvec + 63) / 64 * 64;      // whole wavefronts per group: the digits stay wave-uniform
  const int G = kKsBlock / tpg;
  const int g = threadIdx.x / tpg, v = threadIdx.x - g * tpg;
  const int32_t prec_offset = 1 << (32 - (1 + 2 * T));
  for (int c = 0; c < kKsTile; c++) {
    const int b = b0 + c;
    for (int i = threadIdx.x; i < n_ks; i += kKsBlock) {
      int32_t x = 0;                                  // all digits 0: nothing is subtracted
      if (b < B) {
        int32_t ai = in_a[(size_t)b * in_stride + i];
        if (in2_a) ai = wadd(ai, in2_a[(size_t)b * in_stride + i]);
        x = wadd(ai, prec_offset);
      }
      aibar_s[c][i] = x;
    }
  }
  __syncthreads();
  uint32_t acc[kKsTile][4];
#pragma unroll
  for (int c = 0; c < kKsTile; c++) acc[c][0] = acc[c][1] = acc[c][2] = acc[c][3] = 0;
  if (g < G && v < nvec) {
    const int32_t* col = ks_rows + 4 * v;
    for (int i = g; i < n_ks; i += G) {
      int32_t ab[kKsTile];
#pragma unroll
      for (int c = 0; c < kKsTile; c++) ab[c] = __builtin_amdgcn_readfirstlane(aibar_s[c][i]);
#pragma unroll 2
      for (int j = 0; j < T; j++) {
        const int32_t* r = col + (size_t)(i * T + j) * 4 * row_stride;
        const int4 r1 = *reinterpret_cast<const int4*>(r + row_stride);
        const int4 r2 = *reinterpret_cast<const int4*>(r + 2 * row_stride);
        const int4 r3 = *reinterpret_cast<const int4*>(r + 3 * row_stride);
        const int sh = 32 - (j + 1) * 2;
#pragma unroll
        for (int c = 0; c < kKsTile; c++) {
          const int d = (ab[c] >> sh) & 3;            // uniform
          if (d != 0) {
            const int4 row = d == 1 ? r1 : (d == 2 ? r2 : r3);
            acc[c][0] -= (uint32_t)row.x; acc[c][1] -= (uint32_t)row.y;
            acc[c][2] -= (uint32_t)row.z; acc[c][3] -= (uint32_t)row.w;
          }
        }
      }
    }
  }
#pragma unroll
  for (int c = 0; c < kKsTile; c++) {
    const int b = b0 + c;
    __syncthreads();
    part[threadIdx.x] = make_int4((int)acc[c][0], (int)acc[c][1], (int)acc[c][2], (int)acc[c][3]);
    __syncthreads();
    if (b < B && g == 0 && v < nvec) {
      uint32_t a4[4] = {acc[c][0], acc[c][1], acc[c][2], acc[c][3]};
      for (int k = 1; k < G; k++) {
        const int4 o = part[k * tpg + v];
        a4[0] += (uint32_t)o.x; a4[1] += (uint32_t)o.y; a4[2] += (uint32_t)o.z; a4[3] += (uint32_t)o.w;
      }
      for (int e = 0; e < 4; e++) {
        const int x = 4 * v + e;
        if (x < n) out_a[(size_t)b * n + x] = (int32_t)a4[e];
        else if (x == n) {
          int32_t bb = in_b[b];
          if (in2_b) bb = wadd(wadd(c_b, bb), in2_b[b]);
          out_b[b] = wadd(bb, (int32_t)a4[e]);
        }
      }
    }
  }
  // variances: thread kKsBlock-1-c sums ciphertext c in the reference's order (lwe.rs:109)
  const int cv = kKsBlock - 1 - (int)threadIdx.x;
  if (var_out && cv >= 0 && cv < kKsTile && b0 + cv < B) {
    double var = 0.0;
    for (int i = 0; i < n_ks; i++) {
      const int32_t aibar = aibar_s[cv][i];
      const double* vi = ks_var + (size_t)i * T * 4;
#pragma unroll
      for (int j = 0; j < T; j++) var += vi[j * 4 + ((aibar >> (32 - (j + 1) * 2)) & 3)];
    }
    var_out[b0 + cv] = var;
  }
}

// lwe_key_switch for the largest batches (t = 8, the crate's value): a workgroup takes TILE
// ciphertexts, so a key row is fetched from L2 once per TILE ciphertexts (keyswitch_tile_kernel is
// bound by exactly those reads).  Selecting one of a position's rows per ciphertext by scalar
// branches, as keyswitch_tile_kernel does, does not scale to such tiles (the compiler runs out of
// scalar registers and keeps the sums in scratch memory); here every wavefront keeps the
// position's four 16-byte slices per lane (digit 0: zeros) in a 4 KiB LDS area of its own and
// every ciphertext reads back the slice its digit selects: one scalar bit-field extract, one
// address add, one ds_read_b128 and four subtractions per (ciphertext, position), no branch.
// A wavefront's LDS operations execute in order and the area is private to it, so the next
// position's slices can be stored without any wait.  The digits of the tile (16 bits per
// coefficient: t = 8 digits of 2 bits) are staged in LDS.  Thread layout, partial sums and the
// order of the variance sum as in keyswitch_kernel.
template <int TILE>
__global__ __launch_bounds__(kKsBlock) void keyswitch_lds_kernel(
    const int32_t* __restrict__ in_a, const int32_t* __restrict__ in_b,
    const int32_t* __restrict__ in2_a, const int32_t* __restrict__ in2_b, int32_t c_b,
    int in_stride, const int32_t* __restrict__ ks_rows, const double* __restrict__ ks_var,
    int row_stride, int n, int n_ks, int B, int32_t* __restrict__ out_a,
    int32_t* __restrict__ out_b, double* __restrict__ var_out) {
  constexpr int T = 8;
  static_assert(TILE >= 8 && TILE <= 64 && (TILE & (TILE - 1)) == 0, "one lane per ciphertext of the tile reads its digits");
  __shared__ uint16_t digits[TILE][kN];                          // top 16 bits of a_i + offset
  __shared__ int4 rows_s[kKsBlock / 64][4][64];                  // per wavefront: digit, lane; reused for the partial sums
  const int b0 = blockIdx.x * TILE;
  const int nvec = row_stride / 4;
  const int tpg = (nvec + 63) / 64 * 64;      // whole wavefronts per group: the digits stay wave-uniform
  const int G = kKsBlock / tpg;
  const int g = threadIdx.x / tpg, v = threadIdx.x - g * tpg;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int32_t prec_offset = 1 << (32 - (1 + 2 * T));
  for (int c = 0; c < TILE; c++) {
    const int b = b0 + c;
    for (int i = threadIdx.x; i < n_ks; i += kKsBlock) {
      uint32_t x = 0;                                 // all digits 0: nothing is subtracted
      if (b < B) {
        int32_t ai = in_a[(size_t)b * in_stride + i];
        if (in2_a) ai = wadd(ai, in2_a[(size_t)b * in_stride + i]);
        x = (uint32_t)wadd(ai, prec_offset);
      }
      digits[c][i] = (uint16_t)(x >> 16);
    }
  }
  rows_s[wave][0][lane] = make_int4(0, 0, 0, 0);
  __syncthreads();
  uint32_t acc[TILE][4];
#pragma unroll
  for (int c = 0; c < TILE; c++) acc[c][0] = acc[c][1] = acc[c][2] = acc[c][3] = 0;
  if (g < G) {
    // whole wavefronts enter (g is wave-uniform); lanes beyond the row work on a copy of its last
    // slice, so that every lane is active when the digits are read across lanes below
    const int32_t* col = ks_rows + 4 * min(v, nvec - 1);
    int4* mine = &rows_s[wave][0][lane];                          // + 64 d: the slice of digit d
    for (int i = g; i < n_ks; i += G) {
      // lane c holds the 16 digit bits of ciphertext c; they become scalars one by one below
      const uint32_t dv = digits[lane & (TILE - 1)][i];
      const int32_t* r = col + (size_t)i * T * 4 * row_stride;
      int4 r1 = *reinterpret_cast<const int4*>(r + row_stride);
      int4 r2 = *reinterpret_cast<const int4*>(r + 2 * row_stride);
      int4 r3 = *reinterpret_cast<const int4*>(r + 3 * row_stride);
#pragma unroll 1
      for (int j = 0; j < T; j++) {
        mine[64] = r1;
        mine[128] = r2;
        mine[192] = r3;
        if (j + 1 < T) {
          const int32_t* rn = r + (size_t)(j + 1) * 4 * row_stride;
          r1 = *reinterpret_cast<const int4*>(rn + row_stride);
          r2 = *reinterpret_cast<const int4*>(rn + 2 * row_stride);
          r3 = *reinterpret_cast<const int4*>(rn + 3 * row_stride);
        }
        const int sh = 16 - (j + 1) * 2;
        // eight slices in flight at a time: all of them at once would take the registers of the sums
#pragma unroll
        for (int c0 = 0; c0 < TILE; c0 += 8) {
          int4 row[8];
#pragma unroll
          for (int e = 0; e < 8; e++) {
            const uint32_t d = ((uint32_t)__builtin_amdgcn_readlane((int)dv, c0 + e) >> sh) & 3u;      // uniform
            row[e] = mine[64 * d];
          }
#pragma unroll
          for (int e = 0; e < 8; e++) {
            acc[c0 + e][0] -= (uint32_t)row[e].x; acc[c0 + e][1] -= (uint32_t)row[e].y;
            acc[c0 + e][2] -= (uint32_t)row[e].z; acc[c0 + e][3] -= (uint32_t)row[e].w;
          }
          __builtin_amdgcn_sched_barrier(0);
        }
      }
    }
  }
  int4* part = &rows_s[0][0][0];                                   // kKsBlock entries are there
#pragma unroll
  for (int c = 0; c < TILE; c++) {
    const int b = b0 + c;
    __syncthreads();
    part[threadIdx.x] = make_int4((int)acc[c][0], (int)acc[c][1], (int)acc[c][2], (int)acc[c][3]);
    __syncthreads();
    if (b < B && g == 0 && v < nvec) {
      uint32_t a4[4] = {acc[c][0], acc[c][1], acc[c][2], acc[c][3]};
      for (int k = 1; k < G; k++) {
        const int4 o = part[k * tpg + v];
        a4[0] += (uint32_t)o.x; a4[1] += (uint32_t)o.y; a4[2] += (uint32_t)o.z; a4[3] += (uint32_t)o.w;
      }
      for (int e = 0; e < 4; e++) {
        const int x = 4 * v + e;
        if (x < n) out_a[(size_t)b * n + x] = (int32_t)a4[e];
        else if (x == n) {
          int32_t bb = in_b[b];
          if (in2_b) bb = wadd(wadd(c_b, bb), in2_b[b]);
          out_b[b] = wadd(bb, (int32_t)a4[e]);
        }
      }
    }
  }
  // variances: thread kKsBlock-1-c sums ciphertext c in the reference's order (lwe.rs:109)
  const int cv = kKsBlock - 1 - (int)threadIdx.x;
  if (var_out && cv >= 0 && cv < TILE && b0 + cv < B) {
    double var = 0.0;
    for (int i = 0; i < n_ks; i++) {
      const uint32_t dg = digits[cv][i];
      const double* vi = ks_var + (size_t)i * T * 4;
#pragma unroll
      for (int j = 0; j < T; j++) var += vi[j * 4 + ((dg >> (16 - (j + 1) * 2)) & 3)];
    }
    var_out[b0 + cv] = var;
  }
}

// gates::not (src/gates.rs:122-124, src/lwe.rs:138-152): mask and body negated
// (wrapping), variance kept.  idx runs over B * (n + 1) values.
__global__ void not_kernel(const int32_t* __restrict__ a, const int32_t* __restrict__ b,
                           const double* __restrict__ var, int32_t* __restrict__ out_a,
                           int32_t* __restrict__ out_b, double* __restrict__ out_var, size_t na,
                           size_t B) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < na) out_a[i] = wneg(a[i]);
  else if (i < na + B) {
    out_b[i - na] = wneg(b[i - na]);
    if (out_var) out_var[i - na] = var ? var[i - na] : 0.0;
  }
}

__global__ void fill_kernel(int32_t* __restrict__ p, int32_t v, size_t count) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < count) p[i] = v;
}

// Packs the key-switch key into rows [a_0 .. a_{n-1}, b, 0 pad] of row_stride
// int32; the rows of digit 0 (row index = 0 mod 4), which lwe.rs:638 never
// reads, become zero.  ks_var_zero_kernel does the same to their variances.
__global__ void ks_pack_kernel(const int32_t* __restrict__ ks_a, const int32_t* __restrict__ ks_b,
                               int n, int row_stride, size_t rows, int32_t* __restrict__ dst) {
  const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= rows * (size_t)row_stride) return;
  const size_t row = idx / row_stride;
  const int x = (int)(idx - row * row_stride);
  const int32_t val = x < n ? ks_a[row * n + x] : (x == n ? ks_b[row] : 0);
  dst[idx] = (row & 3) == 0 ? 0 : val;
}

__global__ void ks_var_zero_kernel(double* __restrict__ var, size_t rows) {
  const size_t i = ((size_t)blockIdx.x * blockDim.x + threadIdx.x) * 4;
  if (i < rows) var[i] = 0.0;
}

// ---------------------------------------------------------------------------
// Key generation (src/lwe.rs:294-309, 474-493, 530-558; src/tgsw.rs:81-176;
// src/tlwe.rs:43-64, 112-134).  Randomness is the ChaCha20 keystream (RFC 8439
// block function; the reference draws from rand 0.7's thread_rng, a ChaCha
// generator seeded by the OS, which cannot be seeded by the caller), addressed
// so that every value is a pure function of (seed, purpose, stream, position):
//   state[4..11] = seed, state[12] = block counter,
//   state[13] = purpose, state[14..15] = stream (low, high word).
// ---------------------------------------------------------------------------
enum : uint32_t {
  kRngSecretLwe = 1,   // stream 0: word x, top bit = key bit x
  kRngSecretTlwe = 2,  // stream 0: word u*N + x, top bit = bit x of key polynomial u
  kRngKsNoise = 3,     // stream 0: words 4 idx .. 4 idx + 3 -> normal number idx
  kRngKsMask = 4,      // stream = row ((i*t + j)*base + h): word x = mask coefficient x
  kRngBkMask = 5,      // stream = i*(k+1) + col: word x = coefficient x of the mask polynomial
  kRngBkNoise = 6      // stream = i*(k+1) + col: words 0..3 -> the one Gaussian that is read
};

struct ChaChaKey {
  uint32_t w[8];
};

__host__ __device__ inline uint32_t rotl32(uint32_t v, int c) { return (v << c) | (v >> (32 - c)); }

__host__ __device__ inline void chacha20_block(const ChaChaKey& key, uint32_t counter, uint32_t purpose,
                                               uint64_t stream, uint32_t (&out)[16]) {
  uint32_t in[16] = {0x61707865u, 0x3320646eu, 0x79622d32u, 0x6b206574u,
                     key.w[0], key.w[1], key.w[2], key.w[3], key.w[4], key.w[5], key.w[6], key.w[7],
                     counter, purpose, (uint32_t)stream, (uint32_t)(stream >> 32)};
  uint32_t x[16];
  for (int i = 0; i < 16; i++) x[i] = in[i];
#define TFHE_QR(a, b, c, d)                                        \
  x[a] += x[b]; x[d] = rotl32(x[d] ^ x[a], 16);                    \
  x[c] += x[d]; x[b] = rotl32(x[b] ^ x[c], 12);                    \
  x[a] += x[b]; x[d] = rotl32(x[d] ^ x[a], 8);                     \
  x[c] += x[d]; x[b] = rotl32(x[b] ^ x[c], 7);
  for (int r = 0; r < 10; r++) {
    TFHE_QR(0, 4, 8, 12) TFHE_QR(1, 5, 9, 13) TFHE_QR(2, 6, 10, 14) TFHE_QR(3, 7, 11, 15)
    TFHE_QR(0, 5, 10, 15) TFHE_QR(1, 6, 11, 12) TFHE_QR(2, 7, 8, 13) TFHE_QR(3, 4, 9, 14)
  }
#undef TFHE_QR
  for (int i = 0; i < 16; i++) out[i] = x[i] + in[i];
}

// One word of a stream (computes its whole block; for sparse reads).
__device__ inline uint32_t chacha20_word(const ChaChaKey& key, uint32_t purpose, uint64_t stream, uint32_t word) {
  uint32_t blk[16];
  chacha20_block(key, word >> 4, purpose, stream, blk);
  return blk[word & 15];
}

// Standard normal from four keystream words (Box-Muller on two 53-bit uniforms,
// the first in (0, 1]).  rand_distr::Normal (src/numerics.rs:45-47) samples the same
// distribution with another algorithm.
__device__ inline double normal_from_words(const uint32_t* w) {
  const uint64_t a = ((uint64_t)w[1] << 32 | w[0]) >> 11, b = ((uint64_t)w[3] << 32 | w[2]) >> 11;
  const double u1 = ((double)a + 1.0) * (1.0 / 9007199254740992.0);
  const double u2 = (double)b * (1.0 / 9007199254740992.0);
  return sqrt(-2.0 * log(u1)) * cospi(2.0 * u2);
}

// f64_to_torus_32, src/numerics.rs:56-58: trunc(d * 2^32) as i32, a saturating cast
// (v_cvt_i32_f64 truncates, saturates and maps NaN to 0, as Rust's `as`).
__device__ inline int32_t f64_to_torus32(double d) {
  const double x = d * 4294967296.0;
  int32_t i;
  asm("v_cvt_i32_f64 %0, %1" : "=v"(i) : "v"(x));
  return i;
}

// Raw keystream words (test hook): out[b*16 + e] = word e of block first_block + b.
__global__ void keygen_stream_kernel(ChaChaKey key, uint32_t purpose, uint64_t stream, uint32_t first_block,
                                     uint32_t* __restrict__ out, size_t nblocks) {
  const size_t b = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= nblocks) return;
  uint32_t blk[16];
  chacha20_block(key, first_block + (uint32_t)b, purpose, stream, blk);
  for (int e = 0; e < 16; e++) out[b * 16 + e] = blk[e];
}

// LweKey::generate (src/lwe.rs:294-309) and TLweKey::generate (src/tlwe.rs:43-64): binary keys.
__global__ void keygen_secret_kernel(ChaChaKey key, int n, int tlwe_count, int32_t* __restrict__ lwe_key,
                                     int32_t* __restrict__ tlwe_key) {
  const int b = blockIdx.x * blockDim.x + threadIdx.x;        // one keystream block = 16 key bits
  const int nb_lwe = (n + 15) / 16, nb_tlwe = (tlwe_count + 15) / 16;
  if (b >= nb_lwe + nb_tlwe) return;
  const bool is_lwe = b < nb_lwe;
  const int blk_index = is_lwe ? b : b - nb_lwe;
  uint32_t blk[16];
  chacha20_block(key, (uint32_t)blk_index, is_lwe ? kRngSecretLwe : kRngSecretTlwe, 0, blk);
  int32_t* dst = is_lwe ? lwe_key : tlwe_key;
  const int count = is_lwe ? n : tlwe_count;
  for (int e = 0; e < 16; e++) {
    const int x = blk_index * 16 + e;
    if (x < count) dst[x] = (int32_t)(blk[e] >> 31);
  }
}

// The Gaussian noises of LweKeySwitchKey::create (src/lwe.rs:536-541), before recentring,
// and their sum in fixed order per workgroup (partial[blockIdx.x]).
__global__ __launch_bounds__(256) void keygen_ks_noise_kernel(ChaChaKey key, double alpha, int size_ks,
                                                              double* __restrict__ noise,
                                                              double* __restrict__ partial) {
  __shared__ double red[256];
  const int idx = blockIdx.x * 256 + threadIdx.x;
  double v = 0.0;
  if (idx < size_ks) {
    uint32_t blk[16];
    chacha20_block(key, (uint32_t)(idx >> 2), kRngKsNoise, 0, blk);
    v = alpha * normal_from_words(blk + 4 * (idx & 3));
    noise[idx] = v;
  }
  red[threadIdx.x] = v;
  __syncthreads();
  for (int s = 128; s > 0; s >>= 1) {
    if ((int)threadIdx.x < s) red[threadIdx.x] += red[threadIdx.x + s];
    __syncthreads();
  }
  if (threadIdx.x == 0) partial[blockIdx.x] = red[0];
}

// error = sum / size_ks (src/lwe.rs:539), partial sums added in index order.
__global__ void keygen_ks_mean_kernel(const double* __restrict__ partial, int count, int size_ks,
                                      double* __restrict__ mean) {
  if (blockIdx.x == 0 && threadIdx.x == 0) {
    double sum = 0.0;
    for (int i = 0; i < count; i++) sum += partial[i];
    *mean = sum / (double)size_ks;
  }
}

// LweKeySwitchKey::create (src/lwe.rs:543-557): one workgroup per sample ks[i][j][h].
// h = 0 stays the all-zero trivial sample; h >= 1 is
// encrypt_with_external_noise(message = key'[i] * h * 2^(32 - (j+1) base_bit), noise - error)
// (src/lwe.rs:334-358): uniform mask, b = message + f64_to_torus32(noise) + <a, s>.
__global__ __launch_bounds__(256) void keygen_ks_kernel(ChaChaKey key, int n, int t, int base_bit,
                                                        double alpha, const int32_t* __restrict__ lwe_key,
                                                        const int32_t* __restrict__ extracted_key,
                                                        const double* __restrict__ noise,
                                                        const double* __restrict__ mean,
                                                        int32_t* __restrict__ ks_a, int32_t* __restrict__ ks_b,
                                                        double* __restrict__ ks_var) {
  __shared__ uint32_t red[256];
  const int base = 1 << base_bit;
  const size_t row = blockIdx.x;
  const int h = (int)(row % base), j = (int)((row / base) % t), i = (int)(row / ((size_t)base * t));
  int32_t* a_out = ks_a + row * n;
  uint32_t dot = 0;
  for (int b0 = threadIdx.x; b0 * 16 < n; b0 += blockDim.x) {
    uint32_t blk[16] = {0};
    if (h != 0) chacha20_block(key, (uint32_t)b0, kRngKsMask, row, blk);
    for (int e = 0; e < 16; e++) {
      const int x = b0 * 16 + e;
      if (x < n) {
        a_out[x] = (int32_t)blk[e];
        dot += blk[e] * (uint32_t)lwe_key[x];
      }
    }
  }
  red[threadIdx.x] = dot;
  __syncthreads();
  for (int s = 128; s > 0; s >>= 1) {
    if ((int)threadIdx.x < s) red[threadIdx.x] += red[threadIdx.x + s];
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    if (h == 0) {
      ks_b[row] = 0;
      ks_var[row] = 0.0;
    } else {
      const int index = (i * t + j) * (base - 1) + (h - 1);              // src/lwe.rs:545-556
      const uint32_t message = (uint32_t)(extracted_key[i] * h) * (1u << (32 - (j + 1) * base_bit));
      const int32_t b = wadd((int32_t)message, f64_to_torus32(noise[index] - *mean));
      ks_b[row] = wadd(b, (int32_t)red[0]);
      ks_var[row] = alpha * alpha;
    }
  }
}

// bk[i] = TGswKey::encrypt(key_in[i]) (src/lwe.rs:487-490) for k = 1, in the layout of
// tfhe_hip_load_keys ([n][l][2][2][N], first N coefficients of every polynomial).
// encrypt_zero fills row 0 only (src/tgsw.rs:118-127); TLweSample::encrypt_zero
// (src/tlwe.rs:112-134) multiplies the key with the sample's old, all-zero polynomials, so
// the body is the Gaussian polynomial left-padded to 2N-1 coefficients: of its first N
// coefficients only the last one, the first Gaussian draw, is non-zero.  add_mu_int_h
// (src/tgsw.rs:129-167) with h = [l, 0, ...] (src/tgsw.rs:43-48) then adds key_in[i] * l to
// coefficient 0 of both polynomials of both row-0 samples.  One workgroup per (i, column).
__global__ __launch_bounds__(256) void keygen_bk_kernel(ChaChaKey key, int l, double alpha,
                                                        const int32_t* __restrict__ lwe_key,
                                                        int32_t* __restrict__ bk) {
  const int i = blockIdx.x >> 1, col = blockIdx.x & 1;
  const uint64_t stream = (uint64_t)blockIdx.x;
  int32_t* sample0 = bk + (((size_t)i * l + 0) * 2 + col) * 2 * kN;       // row 0, column col
  const int32_t mu_h = (int32_t)((uint32_t)lwe_key[i] * (uint32_t)l);
  // mask polynomial: uniform (64 keystream blocks), coefficient 0 += key bit * l
  for (int b0 = threadIdx.x; b0 < kN / 16; b0 += blockDim.x) {
    uint32_t blk[16];
    chacha20_block(key, (uint32_t)b0, kRngBkMask, stream, blk);
    for (int e = 0; e < 16; e++) {
      const int x = b0 * 16 + e;
      sample0[x] = x == 0 ? wadd((int32_t)blk[e], mu_h) : (int32_t)blk[e];
    }
  }
  // body polynomial: [key bit * l, 0 ... 0, g0]
  for (int x = threadIdx.x; x < kN; x += blockDim.x) sample0[kN + x] = 0;
  __syncthreads();
  if (threadIdx.x == 0) {
    uint32_t blk[16];
    chacha20_block(key, 0, kRngBkNoise, stream, blk);
    sample0[kN + kN - 1] = f64_to_torus32(alpha * normal_from_words(blk));   // gaussian32(0, alpha)
    sample0[kN] = wadd(sample0[kN], mu_h);
  }
  // rows 1 .. l-1 of this column stay zero (src/tgsw.rs:108-116)
  for (int p = 1; p < l; p++) {
    int32_t* z = bk + (((size_t)i * l + p) * 2 + col) * 2 * kN;
    for (int x = threadIdx.x; x < 2 * kN; x += blockDim.x) z[x] = 0;
  }
}


#ifdef TFHE_TEXTBOOK
// Textbook build: bk[i] = TGSW encryption of lwe_key[i] with EVERY row a TLWE encryption of zero
// (src/tgsw.rs:118-127 fills row 0 only) under the TLWE key: uniform mask a, body
// b = e + a (x) s mod X^N + 1 (src/tlwe.rs:112-134 multiplies the key with zeros), and the
// gadget lwe_key[i] * 2^(32 - (p+1) bg_bit) on coefficient 0 of polynomial j of row (p, j)
// (src/tgsw.rs:43-48,129-167 use h = [l, 0, ..] and add it to every polynomial).  One
// workgroup per row; layout of tfhe_hip_load_keys, [n][l][2][2][N].
__global__ __launch_bounds__(256) void keygen_bk_textbook_kernel(ChaChaKey key, int l, int bg_bit, double alpha,
                                                                 const int32_t* __restrict__ lwe_key,
                                                                 const int32_t* __restrict__ tlwe_key,
                                                                 int32_t* __restrict__ bk) {
  __shared__ int32_t a_s[kN];
  __shared__ int32_t s_s[kN];
  const int row = blockIdx.x;                              // (i l + p) 2 + j
  const int j = row & 1, p = (row >> 1) % l, i = (row >> 1) / l;
  int32_t* out = bk + (size_t)row * 2 * kN;
  for (int b0 = threadIdx.x; b0 < kN / 16; b0 += blockDim.x) {
    uint32_t blk[16];
    chacha20_block(key, (uint32_t)b0, kRngBkMask, (uint64_t)row, blk);
    for (int e = 0; e < 16; e++) a_s[b0 * 16 + e] = (int32_t)blk[e];
  }
  for (int x = threadIdx.x; x < kN; x += blockDim.x) s_s[x] = tlwe_key[x];
  __syncthreads();
  const int32_t gadget = (int32_t)((uint32_t)lwe_key[i] << (32 - (p + 1) * bg_bit));
  for (int x = threadIdx.x; x < kN; x += blockDim.x) {
    uint32_t blk[16];
    chacha20_block(key, (uint32_t)x, kRngBkNoise, (uint64_t)row, blk);
    uint32_t body = (uint32_t)f64_to_torus32(alpha * normal_from_words(blk));
    for (int v = 0; v < kN; v++)
      if (s_s[v]) body += (x >= v) ? (uint32_t)a_s[x - v] : 0u - (uint32_t)a_s[x - v + kN];
    const int32_t g0 = x == 0 ? gadget : 0;
    out[x] = wadd(a_s[x], j == 0 ? g0 : 0);
    out[kN + x] = wadd((int32_t)body, j == 1 ? g0 : 0);
  }
}
#endif

}  // namespace

// Scratch of tfhe_hip_gate_dev_noalloc, the entry point that allocates nothing:
// mod-switched operands and extracted samples of one call (the layout of
// get_scratch).  Every GPU the library runs on holds its own copy of this array
// until the process ends; host code reaches it only through the per-device
// address that tfhe_hip_create obtains with hipGetSymbolAddress.
constexpr size_t kFixedScratchBytes = (size_t)4 << 20;
static __device__ __align__(256) unsigned char scratch[kFixedScratchBytes];

// ===========================================================================
// Host side
// ===========================================================================

using tfhe_detail::fail;

int tfhe_detail::check_params(tfhe_hip_ctx* ctx, const tfhe_hip_params* p) {
  if (!p) return fail(ctx, TFHE_HIP_ERR_ARG, "params is NULL");
  if (p->N != kN || p->k != 1) return fail(ctx, TFHE_HIP_ERR_ARG, "only N=1024, k=1 is supported");
  if (!((p->l == 2 && p->bg_bit == 10) || (p->l == 3 && p->bg_bit == 7)))
    return fail(ctx, TFHE_HIP_ERR_ARG, "supported (l, bg_bit): (2, 10) and (3, 7)");
  if (p->ks_base_bit != 2 || p->ks_t < 1 || p->ks_t * 2 > 31)
    return fail(ctx, TFHE_HIP_ERR_ARG, "key switch: base_bit must be 2 and 1 <= t <= 15");
  if (p->n < 1 || p->n > 1020 || p->n_ks < 1 || p->n_ks > p->N * p->k)
    return fail(ctx, TFHE_HIP_ERR_ARG, "need 1 <= n <= 1020 and 1 <= n_ks <= N*k");
  return 0;
}

namespace {
using tfhe_detail::check_params;

int row_stride_of(const tfhe_hip_params* p) { return ((p->n + 1 + 3) / 4) * 4; }

void blob_sizes(const tfhe_hip_params* p, size_t* bk, size_t* ks, size_t* ks_rows) {
  *bk = (size_t)p->n * p->l * 2 * 2048 * sizeof(cplx);
  size_t rows = (size_t)p->n_ks * p->ks_t * 4;
  *ks_rows = rows * row_stride_of(p) * sizeof(int32_t);
  *ks = *ks_rows + rows * sizeof(double);
}

int ensure_scratch(tfhe_hip_ctx* ctx, size_t bytes) {
  if (bytes <= ctx->scratch_bytes) return 0;
  if (ctx->scratch) HIP_TRY(ctx, hipFree(ctx->scratch));
  ctx->scratch = nullptr;
  ctx->scratch_bytes = 0;
  HIP_TRY(ctx, hipMalloc(&ctx->scratch, bytes));
  ctx->scratch_bytes = bytes;
  return 0;
}

int alloc_keys(tfhe_hip_ctx* ctx, const tfhe_hip_params* p) {
  if (ctx->bkfft) { HIP_TRY(ctx, hipFree(ctx->bkfft)); ctx->bkfft = nullptr; }
  if (ctx->ksblob) { HIP_TRY(ctx, hipFree(ctx->ksblob)); ctx->ksblob = nullptr; }
  ctx->have_keys = false;
  ctx->P = *p;
  ctx->row_stride = row_stride_of(p);
  blob_sizes(p, &ctx->bk_bytes, &ctx->ks_bytes, &ctx->ks_rows_bytes);
  HIP_TRY(ctx, hipMalloc((void**)&ctx->bkfft, ctx->bk_bytes));
  HIP_TRY(ctx, hipMalloc((void**)&ctx->ksblob, ctx->ks_bytes));
  return 0;
}

// Brackets one kernel launch with events when profiling is on.
struct ProfileScope {
  tfhe_hip_ctx* ctx;
  hipEvent_t e0 = nullptr, e1 = nullptr;
  int kind;
  ProfileScope(tfhe_hip_ctx* c, int k) : ctx(c), kind(k) {
    // at most kMaxProfiledLaunches pairs are kept between two reads; later launches go untimed
    if (!ctx->profile || ctx->ev[kind].size() >= kMaxProfiledLaunches) return;
    if (hipEventCreate(&e0) != hipSuccess) { e0 = nullptr; return; }
    if (hipEventCreate(&e1) != hipSuccess) {
      (void)hipEventDestroy(e0);
      e0 = e1 = nullptr;
      return;
    }
    (void)hipEventRecord(e0, ctx->stream);
  }
  ~ProfileScope() {
    if (!e0) return;
    (void)hipEventRecord(e1, ctx->stream);
    try {
      ctx->ev[kind].emplace_back(e0, e1);
    } catch (...) {                       // nothing may be thrown across the C ABI
      (void)hipEventDestroy(e0);
      (void)hipEventDestroy(e1);
    }
  }
  static constexpr size_t kMaxProfiledLaunches = 4096;
};

struct Weights { int wa, wb; int32_t c, mu; };

bool gate_weights(int gate, Weights* w) {
  const int32_t e8 = 1 << 29, m8 = -(1 << 29), e4 = 1 << 30, m4 = (int32_t)0xC0000000u;
  switch (gate) {                                            // src/gates.rs
    case TFHE_HIP_NAND:  *w = {-1, -1, e8, e8}; return true;   // :23-41
    case TFHE_HIP_OR:    *w = {+1, +1, e8, e8}; return true;   // :44-61
    case TFHE_HIP_AND:   *w = {+1, +1, m8, e8}; return true;   // :64-80
    case TFHE_HIP_XOR:   *w = {+2, +2, e4, e8}; return true;   // :83-99
    case TFHE_HIP_XNOR:  *w = {-2, -2, m4, e8}; return true;   // :102-119
#ifndef TFHE_TEXTBOOK
    case TFHE_HIP_NOR:   *w = {-1, -1, m8, m8}; return true;   // :127-143
#else
    // :136 bootstraps NOR with MU = -1/8, which inverts its output; the textbook build uses
    // +1/8 like every other gate
    case TFHE_HIP_NOR:   *w = {-1, -1, m8, e8}; return true;
#endif
    case TFHE_HIP_ANDNY: *w = {-1, +1, m8, e8}; return true;   // :147-162
    case TFHE_HIP_ANDYN: *w = {+1, -1, m8, e8}; return true;   // :166-182
    case TFHE_HIP_ORNY:  *w = {-1, +1, e8, e8}; return true;   // :186-202
    case TFHE_HIP_ORYN:  *w = {+1, -1, e8, e8}; return true;   // :206-222
    default: return false;
  }
}

// scratch layout for a batch of B: bara [B][n], barb [B], extracted samples ea [B][N], eb [B]
// (mux asks for 2 B: its two bootstraps run as one batch)
struct Scratch {
  int32_t *bara, *barb, *ea, *eb;
};

// Lays the four arrays out behind `base`; returns the bytes they take.
size_t scratch_layout(size_t n, size_t B, char* base, Scratch* s) {
  auto up = [](size_t x) { return (x + 63) / 64 * 64; };
  const size_t o_bara = 0, o_barb = o_bara + up(B * n * 4);
  const size_t o_ea = o_barb + up(B * 4), o_eb = o_ea + up(B * kN * 4), total = o_eb + up(B * 4);
  s->bara = (int32_t*)(base + o_bara); s->barb = (int32_t*)(base + o_barb);
  s->ea = (int32_t*)(base + o_ea); s->eb = (int32_t*)(base + o_eb);
  return total;
}

int get_scratch(tfhe_hip_ctx* ctx, size_t B, Scratch* s) {
  const size_t total = scratch_layout(ctx->P.n, B, nullptr, s);
  if (int rc = ensure_scratch(ctx, total)) return rc;
  scratch_layout(ctx->P.n, B, (char*)ctx->scratch, s);
  return 0;
}

int launch_prologue(tfhe_hip_ctx* ctx, const int32_t* ca_a, const int32_t* ca_b, const int32_t* cb_a,
                    const int32_t* cb_b, int wa, int wb, int32_t c, size_t B, int32_t* bara,
                    int32_t* barb) {
  const size_t total = B * (size_t)(ctx->P.n + 1);
  const unsigned grid = (unsigned)((total + 255) / 256);
  ProfileScope prof(ctx, 0);
  hipLaunchKernelGGL(prologue_kernel, dim3(grid), dim3(256), 0, ctx->stream, ca_a, ca_b, cb_a, cb_b,
                     wa, wb, c, ctx->P.n, total, bara, barb);
  HIP_TRY(ctx, hipGetLastError());
  return 0;
}

// One launch of one mapping on `stream`.
int launch_blind_rotate_mode(tfhe_hip_ctx* ctx, int mode, hipStream_t stream, const BlindRotateArgs& a, size_t B) {
  const bool hook = a.acc_io != nullptr;
  if (mode == TFHE_HIP_BR_SPLIT) {
    const dim3 grid((unsigned)B);
    if (ctx->P.l == 2) {
      if (hook) hipLaunchKernelGGL((blind_rotate_split64_kernel<2, 10, true>), grid, dim3(512), 0, stream, a);
      else hipLaunchKernelGGL((blind_rotate_split64_kernel<2, 10, false>), grid, dim3(512), 0, stream, a);
    } else {
      if (hook) hipLaunchKernelGGL((blind_rotate_split64_kernel<3, 7, true>), grid, dim3(768), 0, stream, a);
      else hipLaunchKernelGGL((blind_rotate_split64_kernel<3, 7, false>), grid, dim3(768), 0, stream, a);
    }
  } else if (mode == TFHE_HIP_BR_BATCH64) {
    const unsigned grid = (unsigned)std::min<size_t>(B, (size_t)ctx->num_cus);
    const dim3 block(64 * kWaves64);
    if (ctx->P.l == 2) {
      if (hook) hipLaunchKernelGGL((blind_rotate64_kernel<2, 10, true>), dim3(grid), block, 0, stream, a);
      else hipLaunchKernelGGL((blind_rotate64_kernel<2, 10, false>), dim3(grid), block, 0, stream, a);
    } else {
      if (hook) hipLaunchKernelGGL((blind_rotate64_kernel<3, 7, true>), dim3(grid), block, 0, stream, a);
      else hipLaunchKernelGGL((blind_rotate64_kernel<3, 7, false>), dim3(grid), block, 0, stream, a);
    }
  } else {
    const unsigned grid = (unsigned)std::min<size_t>((B + kWavesPerBlock - 1) / kWavesPerBlock,
                                                     (size_t)ctx->num_cus);
    if (ctx->P.l == 2) {
      if (hook) hipLaunchKernelGGL((blind_rotate_kernel<2, 10, true>), dim3(grid), dim3(kBlock), 0, stream, a);
      else hipLaunchKernelGGL((blind_rotate_kernel<2, 10, false>), dim3(grid), dim3(kBlock), 0, stream, a);
    } else {
      if (hook) hipLaunchKernelGGL((blind_rotate_kernel<3, 7, true>), dim3(grid), dim3(kBlock), 0, stream, a);
      else hipLaunchKernelGGL((blind_rotate_kernel<3, 7, false>), dim3(grid), dim3(kBlock), 0, stream, a);
    }
  }
  HIP_TRY(ctx, hipGetLastError());
  return 0;
}

int launch_blind_rotate(tfhe_hip_ctx* ctx, const int32_t* bara, const int32_t* barb, int32_t mu,
                        int32_t* acc_io, int32_t* out_a, int32_t* out_b, size_t B) {
  BlindRotateArgs a;
  a.bara = bara; a.barb = barb; a.bkfft = ctx->bkfft; a.tables = ctx->tables;
  a.tables64 = ctx->tables64;
  a.acc_io = acc_io; a.out_a = out_a; a.out_b = out_b; a.mu = mu; a.n = ctx->P.n; a.B = (int32_t)B;
#ifdef TFHE_TEXTBOOK
  {
    // the textbook build has one mapping: one ciphertext per wavefront, two wavefronts per SIMD
    if (acc_io) return fail(ctx, TFHE_HIP_ERR_ARG, "the blind-rotate hook is not part of the textbook build");
    ProfileScope prof(ctx, 1);
    const unsigned grid = (unsigned)std::min<size_t>(B, (size_t)ctx->num_cus);      // wavefront-major slots
    if (ctx->P.l == 2)
      hipLaunchKernelGGL((blind_rotate_textbook_kernel<2, 10>), dim3(grid), dim3(64 * kTbWaves), 0, ctx->stream, a);
    else
      hipLaunchKernelGGL((blind_rotate_textbook_kernel<3, 7>), dim3(grid), dim3(64 * kTbWaves), 0, ctx->stream, a);
    HIP_TRY(ctx, hipGetLastError());
    return 0;
  }
#endif
  // Which mappings a batch gets in automatic mode.  Each mapping takes its ciphertexts in rounds:
  // one workgroup per ciphertext (SPLIT) a compute unit's worth, the persistent form with one
  // wavefront per SIMD (BATCH) four per compute unit, the one with two wavefronts per SIMD
  // (BATCH64) eight, and a round costs the same whether full or not.  Measured on MI355X
  // (tools/time_batch_sizes.py) the rounds cost 0.31 : 1 : 1.75 at both parameter sets, so whole
  // BATCH64 rounds are the cheapest per ciphertext and what is left over after them goes to the
  // cheapest of: one more BATCH64 round; one BATCH round; SPLIT rounds; one BATCH round and SPLIT
  // rounds.  The parts run one after another on the context's stream.
  ProfileScope prof(ctx, 1);
  if (ctx->br_mode != TFHE_HIP_BR_AUTO) return launch_blind_rotate_mode(ctx, ctx->br_mode, ctx->stream, a, B);
  constexpr size_t kCostSplit = 31, kCostBatch = 100, kCostBatch64 = 175;
  const size_t cus = (size_t)ctx->num_cus;
  const size_t slots32 = cus * kWavesPerBlock, slots64 = cus * kWaves64;
  const size_t rem = B % slots64;
  size_t n64 = B - rem, n32 = 0, nsplit = 0;               // ciphertexts per mapping, in this order
  if (rem) {
    const size_t over = rem > slots32 ? rem - slots32 : 0;  // what one BATCH round leaves
    const size_t cost_split = (rem + cus - 1) / cus * kCostSplit;
    const size_t cost_batch = kCostBatch + (over + cus - 1) / cus * kCostSplit;
    if (kCostBatch64 <= cost_split && kCostBatch64 <= cost_batch) n64 = B;
    else if (cost_split <= cost_batch) nsplit = rem;
    else { n32 = rem - over; nsplit = over; }
  }
  size_t done = 0;
  const size_t parts[3][2] = {{n64, TFHE_HIP_BR_BATCH64}, {n32, TFHE_HIP_BR_BATCH}, {nsplit, TFHE_HIP_BR_SPLIT}};
  for (const auto& part : parts) {
    if (part[0] == 0) continue;
    BlindRotateArgs sub = a;
    sub.bara = bara + done * (size_t)ctx->P.n;
    if (barb) sub.barb = barb + done;
    if (acc_io) sub.acc_io = acc_io + done * 3 * kN;
    if (out_a) sub.out_a = out_a + done * kN;
    if (out_b) sub.out_b = out_b + done;
    sub.B = (int32_t)part[0];
    if (int rc = launch_blind_rotate_mode(ctx, (int)part[1], ctx->stream, sub, part[0])) return rc;
    done += part[0];
  }
  return 0;
}

int launch_keyswitch(tfhe_hip_ctx* ctx, const int32_t* in_a, const int32_t* in_b, const int32_t* in2_a,
                     const int32_t* in2_b, int32_t c_b, int32_t* out_a, int32_t* out_b,
                     double* var_out, size_t B) {
  const double* ks_var = (const double*)(ctx->ksblob + ctx->ks_rows_bytes);
  ProfileScope prof(ctx, 2);
  // The 32-ciphertext form runs one workgroup per compute unit, i.e. in rounds of 32 ciphertexts per
  // compute unit, at about three quarters of the tile kernel's time per ciphertext when a round is
  // full (measured: 8 192 gates 0.96 against 1.20 ms, 16 384 gates 1.96 against 2.81 ms on Bit80); it
  // is used when its whole rounds cost less than the tile kernel's exact count.
  const size_t lds_round = (size_t)ctx->num_cus * kKsLdsTile;
  const size_t lds_rounds = (B + lds_round - 1) / lds_round;
  if (ctx->P.ks_t == 8 && B >= lds_round && 3 * lds_rounds * lds_round < 4 * B) {
    const unsigned grid = (unsigned)((B + kKsLdsTile - 1) / kKsLdsTile);
    hipLaunchKernelGGL(keyswitch_lds_kernel<kKsLdsTile>, dim3(grid), dim3(kKsBlock), 0, ctx->stream, in_a, in_b,
                       in2_a, in2_b, c_b, kN, (const int32_t*)ctx->ksblob, ks_var, ctx->row_stride, ctx->P.n,
                       ctx->P.n_ks, (int)B, out_a, out_b, var_out);
  } else if (B >= (size_t)ctx->num_cus * kKsTile) {
    // enough ciphertexts to give every compute unit a tile: share the key reads
    const unsigned grid = (unsigned)((B + kKsTile - 1) / kKsTile);
    if (ctx->P.ks_t == 8)
      hipLaunchKernelGGL(keyswitch_tile_kernel<8>, dim3(grid), dim3(kKsBlock), 0, ctx->stream, in_a, in_b, in2_a,
                         in2_b, c_b, kN, (const int32_t*)ctx->ksblob, ks_var, ctx->row_stride, ctx->P.n,
                         ctx->P.n_ks, 8, (int)B, out_a, out_b, var_out);
    else
      hipLaunchKernelGGL(keyswitch_tile_kernel<0>, dim3(grid), dim3(kKsBlock), 0, ctx->stream, in_a, in_b, in2_a,
                         in2_b, c_b, kN, (const int32_t*)ctx->ksblob, ks_var, ctx->row_stride, ctx->P.n,
                         ctx->P.n_ks, ctx->P.ks_t, (int)B, out_a, out_b, var_out);
  } else if (ctx->P.ks_t == 8)
    hipLaunchKernelGGL(keyswitch_kernel<8>, dim3((unsigned)B), dim3(kKsBlock), 0, ctx->stream, in_a, in_b,
                       in2_a, in2_b, c_b, kN, (const int32_t*)ctx->ksblob, ks_var, ctx->row_stride,
                       ctx->P.n, ctx->P.n_ks, ctx->P.ks_t, out_a, out_b, var_out);
  else
    hipLaunchKernelGGL(keyswitch_kernel<0>, dim3((unsigned)B), dim3(kKsBlock), 0, ctx->stream, in_a, in_b,
                       in2_a, in2_b, c_b, kN, (const int32_t*)ctx->ksblob, ks_var, ctx->row_stride,
                       ctx->P.n, ctx->P.n_ks, ctx->P.ks_t, out_a, out_b, var_out);
  HIP_TRY(ctx, hipGetLastError());
  return 0;
}

}  // namespace

int tfhe_detail::need_keys(tfhe_hip_ctx* ctx) {
  if (!ctx) return TFHE_HIP_ERR_ARG;
  if (!ctx->shards.empty())
    return fail(ctx, TFHE_HIP_ERR_ARG, "device-pointer entry points need a single-device context");
  if (!ctx->have_keys) return fail(ctx, TFHE_HIP_ERR_STATE, "no keys loaded");
  HIP_TRY(ctx, hipSetDevice(ctx->device));
  return 0;
}

int tfhe_detail::alloc_key_blobs(tfhe_hip_ctx* ctx, const tfhe_hip_params* p) {
  if (int rc = check_params(ctx, p)) return rc;
  HIP_TRY(ctx, hipSetDevice(ctx->device));
  return alloc_keys(ctx, p);
}

namespace {
using tfhe_detail::need_keys;

void make_tables(std::vector<cplx>& t) {
  t.resize(4096);
  const long double two_pi = 6.283185307179586476925286766559L;
  auto root = [&](cplx& w, long double turns) {      // exp(-2 pi i turns)
    w.x = (double)cosl(-two_pi * turns);
    w.y = (double)sinl(-two_pi * turns);
  };
  for (int a = 0; a < 32; a++)
    for (int b = 0; b < 32; b++) {
      // scalar form of the twist: forward [kb][q] = w1024^(q kb) w4096^q,
      // inverse [n1][kb] = w1024^(n1 kb) w4096^n1 (used conjugated)
      root(t[a * 32 + b], (long double)(a * b) / 1024.0L + (long double)b / 4096.0L);
      root(t[1024 + a * 32 + b], (long double)(a * b) / 1024.0L + (long double)a / 4096.0L);
      // table form: [kb][q] = w1024^(q kb) for both directions, then the twist [r][q] = w4096^(q + 32 r)
      root(t[2048 + a * 32 + b], (long double)(a * b) / 1024.0L);
      root(t[3072 + a * 32 + b], (long double)(b + 32 * a) / 4096.0L);
    }
}

// wave_fft64.hpp: T1[kr][lane] = w1024^(lane kr) w4096^lane, then T2[c - 1][ql] = w64^(ql c)
void make_tables64(std::vector<cplx>& t) {
  t.resize(wfft64::kT1Entries + wfft64::kT2Entries);
  const long double two_pi = 6.283185307179586476925286766559L;
  auto root = [&](cplx& w, long double turns) {
    w.x = (double)cosl(-two_pi * turns);
    w.y = (double)sinl(-two_pi * turns);
  };
  for (int kr = 0; kr < 16; kr++)
    for (int lane = 0; lane < 64; lane++)
      root(t[kr * 64 + lane], (long double)(lane * kr) / 1024.0L + (long double)lane / 4096.0L);
  for (int c = 1; c < 4; c++)
    for (int ql = 0; ql < 16; ql++) root(t[wfft64::kT1Entries + (c - 1) * 16 + ql], (long double)(ql * c) / 64.0L);
}

// (cos, -sin)(2 pi r / 64) for wfft64::g_w64
void make_w64(double (&w)[32]) {
  const long double two_pi = 6.283185307179586476925286766559L;
  for (int r = 0; r < 16; r++) {
    w[2 * r] = (double)cosl(-two_pi * (long double)r / 64.0L);
    w[2 * r + 1] = (double)sinl(-two_pi * (long double)r / 64.0L);
  }
}

// wfft64::g_inv64: for b = 0..7 the factors alpha = w64^-b, beta = w16^-b alpha, alpha' = w64^-(b+8),
// beta' = w16^-b alpha' (w_n = exp(-2 pi i / n)), as (re, im) pairs
void make_inv64(double (&w)[64]) {
  const long double two_pi = 6.283185307179586476925286766559L;
  for (int b = 0; b < 8; b++) {
    const int sixtyfourths[4] = {b, 5 * b, b + 8, 5 * b + 8};
    for (int k = 0; k < 4; k++) {
      w[8 * b + 2 * k] = (double)cosl(two_pi * (long double)sixtyfourths[k] / 64.0L);
      w[8 * b + 2 * k + 1] = (double)sinl(two_pi * (long double)sixtyfourths[k] / 64.0L);
    }
  }
}

// (cos, -sin)(2 pi r / 128) for wfft::g_w128
void make_w128(double (&w)[64]) {
  const long double two_pi = 6.283185307179586476925286766559L;
  for (int r = 0; r < 32; r++) {
    w[2 * r] = (double)cosl(-two_pi * (long double)r / 128.0L);
    w[2 * r + 1] = (double)sinl(-two_pi * (long double)r / 128.0L);
  }
}


// Several two-input gates with the same test-vector value mu, run as ONE
// blind rotation over the concatenated batch (prologue and key switch per
// gate, since their operands and results live in different arrays).
struct GateItem {
  int gate;
  const int32_t *ca_a, *ca_b, *cb_a, *cb_b;
  int32_t *out_a, *out_b;
  double* var_out;
  size_t B;
};

int run_gate_group(tfhe_hip_ctx* ctx, const GateItem* items, int count) {
  size_t total = 0;
  Weights w0{};
  for (int k = 0; k < count; k++) {
    Weights w;
    if (!gate_weights(items[k].gate, &w)) return fail(ctx, TFHE_HIP_ERR_ARG, "unknown gate id");
    if (k == 0) w0 = w;
    else if (w.mu != w0.mu) return fail(ctx, TFHE_HIP_ERR_ARG, "gates of a group must share mu");
    total += items[k].B;
  }
  if (total == 0) return 0;
  Scratch s;
  if (int rc = get_scratch(ctx, total, &s)) return rc;
  const size_t n = ctx->P.n;
  size_t off = 0;
  for (int k = 0; k < count; k++) {
    const GateItem& g = items[k];
    Weights w;
    gate_weights(g.gate, &w);
    if (g.B == 0) continue;
    if (int rc = launch_prologue(ctx, g.ca_a, g.ca_b, g.cb_a, g.cb_b, w.wa, w.wb, w.c, g.B, s.bara + off * n,
                                 s.barb + off)) return rc;
    off += g.B;
  }
  if (int rc = launch_blind_rotate(ctx, s.bara, s.barb, w0.mu, nullptr, s.ea, s.eb, total)) return rc;
  off = 0;
  for (int k = 0; k < count; k++) {
    const GateItem& g = items[k];
    if (g.B == 0) continue;
    if (int rc = launch_keyswitch(ctx, s.ea + off * kN, s.eb + off, nullptr, nullptr, 0, g.out_a, g.out_b,
                                  g.var_out, g.B)) return rc;
    off += g.B;
  }
  return 0;
}

int ensure_work(tfhe_hip_ctx* ctx, size_t bytes) {
  if (bytes <= ctx->work_bytes) return 0;
  if (ctx->work) HIP_TRY(ctx, hipFree(ctx->work));
  ctx->work = nullptr;
  ctx->work_bytes = 0;
  HIP_TRY(ctx, hipMalloc(&ctx->work, bytes));
  ctx->work_bytes = bytes;
  return 0;
}

// Staging helpers for the host-pointer entry points.
struct DevBuf {
  void* p = nullptr;
  ~DevBuf() { if (p) (void)hipFree(p); }
};

}  // namespace

extern "C" {

#ifdef TFHE_STAMPS
int tfhe_hip_debug_stamps(unsigned long long* out) {
  return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_stamps), 16 * sizeof(unsigned long long));
}
#endif

int tfhe_hip_create(tfhe_hip_ctx** out, int device) {
  if (!out) return TFHE_HIP_ERR_ARG;
  *out = nullptr;
  int count = 0;
  if (hipGetDeviceCount(&count) != hipSuccess || device < 0 || device >= count) return TFHE_HIP_ERR_RUNTIME;
  tfhe_hip_ctx* ctx = new tfhe_hip_ctx();
  ctx->device = device;
  if (hipSetDevice(device) != hipSuccess) { delete ctx; return TFHE_HIP_ERR_RUNTIME; }
  int cus = 0;
  if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, device) == hipSuccess && cus > 0)
    ctx->num_cus = cus;
  // this device's copy of the fixed scratch array (the device is current)
  if (hipGetSymbolAddress(&ctx->fixed_scratch, HIP_SYMBOL(scratch)) != hipSuccess || !ctx->fixed_scratch) {
    delete ctx;
    return TFHE_HIP_ERR_RUNTIME;
  }
  std::vector<cplx> t;
  make_tables(t);
  // this device's copy of the twist factors; every context of a device writes the same bytes
  double w128[64];
  make_w128(w128);
  if (hipMemcpyToSymbol(HIP_SYMBOL(wfft::g_w128), w128, sizeof(w128)) != hipSuccess) {
    delete ctx;
    return TFHE_HIP_ERR_RUNTIME;
  }
  if (hipMalloc((void**)&ctx->tables, t.size() * sizeof(cplx)) != hipSuccess ||
      hipMemcpy(ctx->tables, t.data(), t.size() * sizeof(cplx), hipMemcpyHostToDevice) != hipSuccess) {
    tfhe_hip_destroy(ctx);
    return TFHE_HIP_ERR_RUNTIME;
  }
  std::vector<cplx> t64;
  make_tables64(t64);
  double w64[32], inv64[64];
  make_w64(w64);
  make_inv64(inv64);
  if (hipMemcpyToSymbol(HIP_SYMBOL(wfft64::g_w64), w64, sizeof(w64)) != hipSuccess ||
      hipMemcpyToSymbol(HIP_SYMBOL(wfft64::g_inv64), inv64, sizeof(inv64)) != hipSuccess ||
      hipMalloc((void**)&ctx->tables64, t64.size() * sizeof(cplx)) != hipSuccess ||
      hipMemcpy(ctx->tables64, t64.data(), t64.size() * sizeof(cplx), hipMemcpyHostToDevice) != hipSuccess) {
    tfhe_hip_destroy(ctx);
    return TFHE_HIP_ERR_RUNTIME;
  }

  *out = ctx;
  return 0;
}

int tfhe_hip_set_blind_rotate_mode(tfhe_hip_ctx* ctx, int mode) {
  if (!ctx) return TFHE_HIP_ERR_ARG;
  if (mode != TFHE_HIP_BR_AUTO && mode != TFHE_HIP_BR_BATCH && mode != TFHE_HIP_BR_SPLIT &&
      mode != TFHE_HIP_BR_BATCH64)
    return fail(ctx, TFHE_HIP_ERR_ARG, "unknown blind-rotate mode");
  ctx->br_mode = mode;
  for (tfhe_hip_ctx* s : ctx->shards) s->br_mode = mode;
  return 0;
}

void tfhe_hip_destroy(tfhe_hip_ctx* ctx) {
  if (!ctx) return;
  if (!ctx->shards.empty()) {                 // multi-GPU context: owns nothing but its shards
    for (tfhe_hip_ctx* s : ctx->shards) tfhe_hip_destroy(s);
    delete ctx;
    return;
  }
  (void)hipSetDevice(ctx->device);
  (void)hipStreamSynchronize(ctx->stream);
  tfhe_detail::release_handles(ctx);
  if (ctx->tables) (void)hipFree(ctx->tables);
  if (ctx->tables64) (void)hipFree(ctx->tables64);
  if (ctx->bkfft) (void)hipFree(ctx->bkfft);
  if (ctx->ksblob) (void)hipFree(ctx->ksblob);
  if (ctx->scratch) (void)hipFree(ctx->scratch);
  if (ctx->work) (void)hipFree(ctx->work);
  if (ctx->stage_dev) (void)hipFree(ctx->stage_dev);
  if (ctx->stage_pin) (void)hipHostFree(ctx->stage_pin);
  if (ctx->owns_stream && ctx->stream) (void)hipStreamDestroy(ctx->stream);
  for (auto& v : ctx->ev)
    for (auto& pr : v) {
      (void)hipEventDestroy(pr.first);
      (void)hipEventDestroy(pr.second);
    }
  delete ctx;
}

const char* tfhe_hip_last_error(tfhe_hip_ctx* ctx) { return ctx ? ctx->err.c_str() : "null context"; }

const char* tfhe_hip_variant(void) {
#ifdef TFHE_TEXTBOOK
  return "textbook";
#else
  return "reference";
#endif
}

int tfhe_hip_set_stream(tfhe_hip_ctx* ctx, void* hip_stream) {
  if (!ctx) return TFHE_HIP_ERR_ARG;
  if (!ctx->shards.empty())
    return fail(ctx, TFHE_HIP_ERR_ARG, "a multi-GPU context runs on streams of its own");
  if ((hipStream_t)hip_stream == ctx->stream) return 0;
  // The context's scratch buffers are shared by every call: work still running on
  // the old stream must finish before calls on the new one may reuse them.
  HIP_TRY(ctx, hipSetDevice(ctx->device));
  HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
  ctx->stream = (hipStream_t)hip_stream;
  return 0;
}

int tfhe_hip_synchronize(tfhe_hip_ctx* ctx) {
  if (!ctx) return TFHE_HIP_ERR_ARG;
  if (!ctx->shards.empty()) {
    for (tfhe_hip_ctx* s : ctx->shards)
      if (int rc = tfhe_hip_synchronize(s)) return fail(ctx, rc, s->err);
    return 0;
  }
  HIP_TRY(ctx, hipSetDevice(ctx->device));
  HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
  return 0;
}

int tfhe_hip_profile_enable(tfhe_hip_ctx* ctx, int on) {
  if (!ctx) return TFHE_HIP_ERR_ARG;
  ctx->profile = on != 0;
  for (tfhe_hip_ctx* s : ctx->shards) s->profile = on != 0;
  return 0;
}

int tfhe_hip_profile_read(tfhe_hip_ctx* ctx, double ms_sum[3], int launches[3]) {
  if (!ctx || !ms_sum || !launches) return TFHE_HIP_ERR_ARG;
  if (!ctx->shards.empty()) {                 // sums over the devices
    for (int k = 0; k < 3; k++) { ms_sum[k] = 0.0; launches[k] = 0; }
    for (tfhe_hip_ctx* s : ctx->shards) {
      double ms[3];
      int n[3];
      if (int rc = tfhe_hip_profile_read(s, ms, n)) return fail(ctx, rc, s->err);
      for (int k = 0; k < 3; k++) { ms_sum[k] += ms[k]; launches[k] += n[k]; }
    }
    return 0;
  }
  HIP_TRY(ctx, hipSetDevice(ctx->device));
  HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
  for (int k = 0; k < 3; k++) {
    ms_sum[k] = 0.0;
    launches[k] = (int)ctx->ev[k].size();
    for (auto& pr : ctx->ev[k]) {
      float ms = 0.f;
      (void)hipEventSynchronize(pr.second);
      if (hipEventElapsedTime(&ms, pr.first, pr.second) == hipSuccess) ms_sum[k] += ms;
      (void)hipEventDestroy(pr.first);
      (void)hipEventDestroy(pr.second);
    }
    ctx->ev[k].clear();
  }
  return 0;
}

int tfhe_hip_key_blob_sizes(const tfhe_hip_params* params, size_t* bk_bytes, size_t* ks_bytes) {
  if (check_params(nullptr, params) || !bk_bytes || !ks_bytes) return TFHE_HIP_ERR_ARG;
  size_t rows;
  blob_sizes(params, bk_bytes, ks_bytes, &rows);
  return 0;
}

}  // extern "C"

namespace {

// Key material in DEVICE memory, in the layouts of tfhe_hip_load_keys (ks_var may be
// NULL = zeros), to the two device blobs the kernels read: the bootstrapping key in
// the FFT domain, the key-switch key as padded rows followed by its variances.
int keys_from_device(tfhe_hip_ctx* ctx, const tfhe_hip_params* params, const int32_t* bk_dev,
                     const int32_t* ks_a_dev, const int32_t* ks_b_dev, const double* ks_var_dev) {
  if (int rc = alloc_keys(ctx, params)) return rc;
  const tfhe_hip_params& P = ctx->P;
  {
#ifdef TFHE_TEXTBOOK
    const size_t count = (size_t)P.n * P.l * 4;            // (i, level, component, polynomial)
    hipLaunchKernelGGL(bk_fft_textbook_kernel, dim3((unsigned)((count + kTbWaves - 1) / kTbWaves)),
                       dim3(64 * kTbWaves), 0, ctx->stream, bk_dev, ctx->tables64, ctx->bkfft, (int)count);
#else
    const size_t count = (size_t)P.n * P.l * 2;            // (i, row, col) triples
    const unsigned grid = (unsigned)((count + kWavesPerBlock - 1) / kWavesPerBlock);
    hipLaunchKernelGGL(bk_fft_kernel, dim3(grid), dim3(kBlock), 0, ctx->stream, bk_dev, ctx->tables,
                       ctx->bkfft, (int)count);
#endif
    HIP_TRY(ctx, hipGetLastError());

  }
  {
    const size_t rows = (size_t)P.n_ks * P.ks_t * 4;
    const size_t total = rows * (size_t)ctx->row_stride;
    hipLaunchKernelGGL(ks_pack_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, ctx->stream,
                       ks_a_dev, ks_b_dev, P.n, ctx->row_stride, rows, (int32_t*)ctx->ksblob);
    HIP_TRY(ctx, hipGetLastError());
    if (ks_var_dev)
      HIP_TRY(ctx, hipMemcpyAsync(ctx->ksblob + ctx->ks_rows_bytes, ks_var_dev, rows * sizeof(double),
                                  hipMemcpyDeviceToDevice, ctx->stream));
    else
      HIP_TRY(ctx, hipMemsetAsync(ctx->ksblob + ctx->ks_rows_bytes, 0, rows * sizeof(double), ctx->stream));
    hipLaunchKernelGGL(ks_var_zero_kernel, dim3((unsigned)((rows / 4 + 255) / 256)), dim3(256), 0, ctx->stream,
                       (double*)(ctx->ksblob + ctx->ks_rows_bytes), rows);
    HIP_TRY(ctx, hipGetLastError());
  }
  HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
  ctx->have_keys = true;
  return 0;
}

struct KeySizes {
  size_t bk, ks_rows, ks_a, ks_b, ks_var;     // bytes of the raw (time-domain) arrays
};
KeySizes raw_key_sizes(const tfhe_hip_params* p) {
  KeySizes z;
  z.bk = (size_t)p->n * p->l * 4 * kN * sizeof(int32_t);
  z.ks_rows = (size_t)p->n_ks * p->ks_t * 4;
  z.ks_a = z.ks_rows * p->n * sizeof(int32_t);
  z.ks_b = z.ks_rows * sizeof(int32_t);
  z.ks_var = z.ks_rows * sizeof(double);
  return z;
}

}  // namespace

extern "C" {

int tfhe_hip_load_keys(tfhe_hip_ctx* ctx, const tfhe_hip_params* params, const int32_t* bk,
                       const int32_t* ks_a, const int32_t* ks_b, const double* ks_var) {
  if (!ctx) return TFHE_HIP_ERR_ARG;
  if (int rc = check_params(ctx, params)) return rc;
  if (!bk || !ks_a || !ks_b) return fail(ctx, TFHE_HIP_ERR_ARG, "key pointer is NULL");
  if (!ctx->shards.empty()) return tfhe_detail::load_keys_multi(ctx, params, bk, ks_a, ks_b, ks_var);
  HIP_TRY(ctx, hipSetDevice(ctx->device));
  ctx->have_keys = false;
  const KeySizes z = raw_key_sizes(params);
  DevBuf dbk, da, db, dv;
  HIP_TRY(ctx, hipMalloc(&dbk.p, z.bk));
  HIP_TRY(ctx, hipMalloc(&da.p, z.ks_a));
  HIP_TRY(ctx, hipMalloc(&db.p, z.ks_b));
  HIP_TRY(ctx, hipMemcpyAsync(dbk.p, bk, z.bk, hipMemcpyHostToDevice, ctx->stream));
  HIP_TRY(ctx, hipMemcpyAsync(da.p, ks_a, z.ks_a, hipMemcpyHostToDevice, ctx->stream));
  HIP_TRY(ctx, hipMemcpyAsync(db.p, ks_b, z.ks_b, hipMemcpyHostToDevice, ctx->stream));
  if (ks_var) {
    HIP_TRY(ctx, hipMalloc(&dv.p, z.ks_var));
    HIP_TRY(ctx, hipMemcpyAsync(dv.p, ks_var, z.ks_var, hipMemcpyHostToDevice, ctx->stream));
  }
  return keys_from_device(ctx, params, (const int32_t*)dbk.p, (const int32_t*)da.p, (const int32_t*)db.p,
                          (const double*)dv.p);
}

int tfhe_hip_keygen_stream(tfhe_hip_ctx* ctx, const uint8_t seed[32], uint32_t purpose, uint64_t stream,
                           uint32_t first_block, uint32_t* out, size_t nblocks) {
  if (!ctx) return TFHE_HIP_ERR_ARG;
  if (!seed || !out) return fail(ctx, TFHE_HIP_ERR_ARG, "pointer is NULL");
  if (nblocks == 0) return 0;
  tfhe_hip_ctx* c = ctx->shards.empty() ? ctx : ctx->shards[0];
  HIP_TRY(ctx, hipSetDevice(c->device));
  ChaChaKey key;
  memcpy(key.w, seed, 32);
  DevBuf d;
  HIP_TRY(ctx, hipMalloc(&d.p, nblocks * 64));
  hipLaunchKernelGGL(keygen_stream_kernel, dim3((unsigned)((nblocks + 255) / 256)), dim3(256), 0, c->stream,
                     key, purpose, stream, first_block, (uint32_t*)d.p, nblocks);
  HIP_TRY(ctx, hipGetLastError());
  HIP_TRY(ctx, hipMemcpyAsync(out, d.p, nblocks * 64, hipMemcpyDeviceToHost, c->stream));
  HIP_TRY(ctx, hipStreamSynchronize(c->stream));
  return 0;
}

int tfhe_hip_generate_keys(tfhe_hip_ctx* ctx, const tfhe_hip_params* params, double ks_stdev,
                           double bk_stdev, const uint8_t seed[32], int32_t* lwe_key, int32_t* tlwe_key,
                           int32_t* bk_out, int32_t* ks_a_out, int32_t* ks_b_out, double* ks_var_out) {
  if (!ctx) return TFHE_HIP_ERR_ARG;
  if (int rc = check_params(ctx, params)) return rc;
  if (!seed || !lwe_key) return fail(ctx, TFHE_HIP_ERR_ARG, "seed or secret-key pointer is NULL");
  if (!(ks_stdev >= 0.0) || !(bk_stdev >= 0.0)) return fail(ctx, TFHE_HIP_ERR_ARG, "standard deviations must be >= 0");
  tfhe_hip_ctx* c = ctx->shards.empty() ? ctx : ctx->shards[0];     // generate on the first GPU
  HIP_TRY(ctx, hipSetDevice(c->device));
  c->have_keys = false;
  ctx->have_keys = false;
  const tfhe_hip_params& P = *params;
  ChaChaKey key;
  memcpy(key.w, seed, 32);
  const KeySizes z = raw_key_sizes(params);
  const int tlwe_count = (P.k + 1) * P.N;
  const int size_ks = P.n_ks * P.ks_t * ((1 << P.ks_base_bit) - 1);
  const int noise_blocks = (size_ks + 255) / 256;
  DevBuf dbk, da, db, dv, dkey, dtkey, dnoise, dpartial;
  auto run = [&]() -> int {
    HIP_TRY(c, hipMalloc(&dbk.p, z.bk));
    HIP_TRY(c, hipMalloc(&da.p, z.ks_a));
    HIP_TRY(c, hipMalloc(&db.p, z.ks_b));
    HIP_TRY(c, hipMalloc(&dv.p, z.ks_var));
    HIP_TRY(c, hipMalloc(&dkey.p, (size_t)P.n * 4));
    HIP_TRY(c, hipMalloc(&dtkey.p, (size_t)tlwe_count * 4));
    HIP_TRY(c, hipMalloc(&dnoise.p, (size_t)size_ks * 8));
    HIP_TRY(c, hipMalloc(&dpartial.p, (size_t)(noise_blocks + 1) * 8));
    int32_t* k_lwe = (int32_t*)dkey.p;
    int32_t* k_tlwe = (int32_t*)dtkey.p;
    double* partial = (double*)dpartial.p;
    const int secret_blocks = (P.n + 15) / 16 + (tlwe_count + 15) / 16;
    hipLaunchKernelGGL(keygen_secret_kernel, dim3((secret_blocks + 255) / 256), dim3(256), 0, c->stream, key,
                       P.n, tlwe_count, k_lwe, k_tlwe);
    HIP_TRY(c, hipGetLastError());
    hipLaunchKernelGGL(keygen_ks_noise_kernel, dim3(noise_blocks), dim3(256), 0, c->stream, key, ks_stdev,
                       size_ks, (double*)dnoise.p, partial);
    HIP_TRY(c, hipGetLastError());
    hipLaunchKernelGGL(keygen_ks_mean_kernel, dim3(1), dim3(64), 0, c->stream, (const double*)partial,
                       noise_blocks, size_ks, partial + noise_blocks);
    HIP_TRY(c, hipGetLastError());
    // LweKey::extract (src/lwe.rs:369-386): the first k polynomials of the TLWE key; the
    // key-switch key has n_ks rows (src/lwe.rs:461) and reads bits 0 .. n_ks-1 of it
    hipLaunchKernelGGL(keygen_ks_kernel, dim3((unsigned)z.ks_rows), dim3(256), 0, c->stream, key, P.n, P.ks_t,
                       P.ks_base_bit, ks_stdev, (const int32_t*)k_lwe, (const int32_t*)k_tlwe,
                       (const double*)dnoise.p, (const double*)(partial + noise_blocks), (int32_t*)da.p,
                       (int32_t*)db.p, (double*)dv.p);
    HIP_TRY(c, hipGetLastError());
#ifdef TFHE_TEXTBOOK
    hipLaunchKernelGGL(keygen_bk_textbook_kernel, dim3((unsigned)(P.n * P.l * 2)), dim3(256), 0, c->stream, key, P.l,
                       P.bg_bit, bk_stdev, (const int32_t*)k_lwe, (const int32_t*)k_tlwe, (int32_t*)dbk.p);
#else
    hipLaunchKernelGGL(keygen_bk_kernel, dim3((unsigned)P.n * 2), dim3(256), 0, c->stream, key, P.l, bk_stdev,
                       (const int32_t*)k_lwe, (int32_t*)dbk.p);
#endif
    HIP_TRY(c, hipGetLastError());
    HIP_TRY(c, hipMemcpyAsync(lwe_key, k_lwe, (size_t)P.n * 4, hipMemcpyDeviceToHost, c->stream));
    if (tlwe_key) HIP_TRY(c, hipMemcpyAsync(tlwe_key, k_tlwe, (size_t)tlwe_count * 4, hipMemcpyDeviceToHost, c->stream));
    if (bk_out) HIP_TRY(c, hipMemcpyAsync(bk_out, dbk.p, z.bk, hipMemcpyDeviceToHost, c->stream));
    if (ks_a_out) HIP_TRY(c, hipMemcpyAsync(ks_a_out, da.p, z.ks_a, hipMemcpyDeviceToHost, c->stream));
    if (ks_b_out) HIP_TRY(c, hipMemcpyAsync(ks_b_out, db.p, z.ks_b, hipMemcpyDeviceToHost, c->stream));
    if (ks_var_out) HIP_TRY(c, hipMemcpyAsync(ks_var_out, dv.p, z.ks_var, hipMemcpyDeviceToHost, c->stream));
    if (int rc = keys_from_device(c, params, (const int32_t*)dbk.p, (const int32_t*)da.p, (const int32_t*)db.p,
                                  (const double*)dv.p)) return rc;
    // the secret keys and the noises do not outlive the call in device memory
    HIP_TRY(c, hipMemsetAsync(dkey.p, 0, (size_t)P.n * 4, c->stream));
    HIP_TRY(c, hipMemsetAsync(dtkey.p, 0, (size_t)tlwe_count * 4, c->stream));
    HIP_TRY(c, hipMemsetAsync(dnoise.p, 0, (size_t)size_ks * 8, c->stream));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    return 0;
  };
  if (int rc = run()) return c == ctx ? rc : fail(ctx, rc, c->err);
  if (c != ctx) return tfhe_detail::replicate_keys_multi(ctx, params);
  return 0;
}


int tfhe_hip_export_keys_dev(tfhe_hip_ctx* ctx, void* bk_blob_dev, void* ks_blob_dev) {
  if (int rc = need_keys(ctx)) return rc;
  if (!bk_blob_dev || !ks_blob_dev) return fail(ctx, TFHE_HIP_ERR_ARG, "blob pointer is NULL");
  HIP_TRY(ctx, hipMemcpyAsync(bk_blob_dev, ctx->bkfft, ctx->bk_bytes, hipMemcpyDeviceToDevice, ctx->stream));
  HIP_TRY(ctx, hipMemcpyAsync(ks_blob_dev, ctx->ksblob, ctx->ks_bytes, hipMemcpyDeviceToDevice, ctx->stream));
  HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
  return 0;
}

int tfhe_hip_import_keys_dev(tfhe_hip_ctx* ctx, const tfhe_hip_params* params, const void* bk_blob_dev,
                             const void* ks_blob_dev) {
  if (!ctx) return TFHE_HIP_ERR_ARG;
  if (int rc = check_params(ctx, params)) return rc;
  if (!bk_blob_dev || !ks_blob_dev) return fail(ctx, TFHE_HIP_ERR_ARG, "blob pointer is NULL");
  if (!ctx->shards.empty())
    return fail(ctx, TFHE_HIP_ERR_ARG, "device-pointer entry points need a single-device context");
  HIP_TRY(ctx, hipSetDevice(ctx->device));
  if (int rc = alloc_keys(ctx, params)) return rc;
  HIP_TRY(ctx, hipMemcpyAsync(ctx->bkfft, bk_blob_dev, ctx->bk_bytes, hipMemcpyDeviceToDevice, ctx->stream));
  HIP_TRY(ctx, hipMemcpyAsync(ctx->ksblob, ks_blob_dev, ctx->ks_bytes, hipMemcpyDeviceToDevice, ctx->stream));
  HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
  ctx->have_keys = true;
  return 0;
}

int tfhe_hip_bootstrap_noks_dev(tfhe_hip_ctx* ctx, int32_t mu, const int32_t* a, const int32_t* b,
                                int32_t* out_a, int32_t* out_b, size_t B) {
  if (int rc = need_keys(ctx)) return rc;
  if (B == 0) return 0;
  if (!a || !b || !out_a || !out_b) return fail(ctx, TFHE_HIP_ERR_ARG, "sample pointer is NULL");
  Scratch s;
  if (int rc = get_scratch(ctx, B, &s)) return rc;
  if (int rc = launch_prologue(ctx, a, b, nullptr, nullptr, 1, 0, 0, B, s.bara, s.barb)) return rc;
  return launch_blind_rotate(ctx, s.bara, s.barb, mu, nullptr, out_a, out_b, B);
}

int tfhe_hip_keyswitch_dev(tfhe_hip_ctx* ctx, const int32_t* a, const int32_t* b, int32_t* out_a,
                           int32_t* out_b, double* var_out, size_t B) {
  if (int rc = need_keys(ctx)) return rc;
  if (B == 0) return 0;
  if (!a || !b || !out_a || !out_b) return fail(ctx, TFHE_HIP_ERR_ARG, "sample pointer is NULL");
  return launch_keyswitch(ctx, a, b, nullptr, nullptr, 0, out_a, out_b, var_out, B);
}

int tfhe_hip_bootstrap_dev(tfhe_hip_ctx* ctx, int32_t mu, const int32_t* a, const int32_t* b,
                           int32_t* out_a, int32_t* out_b, double* var_out, size_t B) {
  if (int rc = need_keys(ctx)) return rc;
  if (B == 0) return 0;
  if (!a || !b || !out_a || !out_b) return fail(ctx, TFHE_HIP_ERR_ARG, "sample pointer is NULL");
  Scratch s;
  if (int rc = get_scratch(ctx, B, &s)) return rc;
  if (int rc = launch_prologue(ctx, a, b, nullptr, nullptr, 1, 0, 0, B, s.bara, s.barb)) return rc;
  if (int rc = launch_blind_rotate(ctx, s.bara, s.barb, mu, nullptr, s.ea, s.eb, B)) return rc;
  return launch_keyswitch(ctx, s.ea, s.eb, nullptr, nullptr, 0, out_a, out_b, var_out, B);
}

}  // extern "C"

namespace {

// The gate on device buffers.  fixed = false: scratch from the context's pool
// (grown with hipMalloc on demand); fixed = true: the library's fixed scratch
// array, no allocation of any kind.
int gate_on_device(tfhe_hip_ctx* ctx, bool fixed, int gate, const int32_t* ca_a, const int32_t* ca_b,
                   const int32_t* cb_a, const int32_t* cb_b, const int32_t* cc_a, const int32_t* cc_b,
                   int32_t* out_a, int32_t* out_b, double* var_out, size_t B) {
  if (int rc = need_keys(ctx)) return rc;
  if (B == 0) return 0;
  if (!ca_a || !ca_b || !cb_a || !cb_b || !out_a || !out_b)
    return fail(ctx, TFHE_HIP_ERR_ARG, "sample pointer is NULL");
  Scratch s;
  auto take_scratch = [&](size_t count) -> int {
    if (!fixed) return get_scratch(ctx, count, &s);
    if (!ctx->fixed_scratch) return fail(ctx, TFHE_HIP_ERR_STATE, "the fixed scratch array is unavailable");
    const size_t per = scratch_layout(ctx->P.n, 1, nullptr, &s);
    if (count > kFixedScratchBytes / per || scratch_layout(ctx->P.n, count, nullptr, &s) > kFixedScratchBytes)
      return fail(ctx, TFHE_HIP_ERR_ARG,
                  "batch too large for the fixed scratch (" + std::to_string(kFixedScratchBytes >> 20) +
                      " MiB): at most about " + std::to_string(kFixedScratchBytes / per) +
                      " ciphertexts per call with these parameters, half as many mux gates");
    scratch_layout(ctx->P.n, count, (char*)ctx->fixed_scratch, &s);
    return 0;
  };
  if (gate == TFHE_HIP_MUX) {                                 // src/gates.rs:239-268
    if (!cc_a || !cc_b) return fail(ctx, TFHE_HIP_ERR_ARG, "mux needs a third operand");
    // u1 = bootstrap_noks(and(a, b)), u2 = bootstrap_noks(andny(a, c)) share mu and run as ONE
    // blind rotation over 2 B ciphertexts; the key switch then takes trivial(1/8) + u1 + u2.
    if (int rc = take_scratch(2 * B)) return rc;
    Weights w1, w2;
    gate_weights(TFHE_HIP_AND, &w1);
    gate_weights(TFHE_HIP_ANDNY, &w2);
    const size_t n = ctx->P.n;
    if (int rc = launch_prologue(ctx, ca_a, ca_b, cb_a, cb_b, w1.wa, w1.wb, w1.c, B, s.bara, s.barb)) return rc;
    if (int rc = launch_prologue(ctx, ca_a, ca_b, cc_a, cc_b, w2.wa, w2.wb, w2.c, B, s.bara + B * n, s.barb + B)) return rc;
    if (int rc = launch_blind_rotate(ctx, s.bara, s.barb, w1.mu, nullptr, s.ea, s.eb, 2 * B)) return rc;
    return launch_keyswitch(ctx, s.ea, s.eb, s.ea + B * kN, s.eb + B, 1 << 29, out_a, out_b, var_out, B);
  }
  Weights w;
  if (!gate_weights(gate, &w)) return fail(ctx, TFHE_HIP_ERR_ARG, "unknown gate id");
  if (int rc = take_scratch(B)) return rc;
  if (int rc = launch_prologue(ctx, ca_a, ca_b, cb_a, cb_b, w.wa, w.wb, w.c, B, s.bara, s.barb)) return rc;
  if (int rc = launch_blind_rotate(ctx, s.bara, s.barb, w.mu, nullptr, s.ea, s.eb, B)) return rc;
  return launch_keyswitch(ctx, s.ea, s.eb, nullptr, nullptr, 0, out_a, out_b, var_out, B);
}

}  // namespace

extern "C" {

int tfhe_hip_gate_dev(tfhe_hip_ctx* ctx, int gate, const int32_t* ca_a, const int32_t* ca_b,
                      const int32_t* cb_a, const int32_t* cb_b, const int32_t* cc_a,
                      const int32_t* cc_b, int32_t* out_a, int32_t* out_b, double* var_out, size_t B) {
  return gate_on_device(ctx, false, gate, ca_a, ca_b, cb_a, cb_b, cc_a, cc_b, out_a, out_b, var_out, B);
}

int tfhe_hip_gate_dev_noalloc(tfhe_hip_ctx* ctx, int gate, const int32_t* ca_a, const int32_t* ca_b,
                              const int32_t* cb_a, const int32_t* cb_b, const int32_t* cc_a,
                              const int32_t* cc_b, int32_t* out_a, int32_t* out_b, double* var_out,
                              size_t B) {
  if (!ctx) return TFHE_HIP_ERR_ARG;
  // The per-launch timing of tfhe_hip_profile_* creates events; not on this path.
  const bool profile = ctx->profile;
  ctx->profile = false;
  const int rc = gate_on_device(ctx, true, gate, ca_a, ca_b, cb_a, cb_b, cc_a, cc_b, out_a, out_b, var_out, B);
  ctx->profile = profile;
  return rc;
}

int tfhe_hip_polymul_dev(tfhe_hip_ctx* ctx, const int32_t* d, const int32_t* s, int32_t* out,
                         int accumulate, size_t B) {
  if (!ctx) return TFHE_HIP_ERR_ARG;
  if (!ctx->shards.empty())
    return fail(ctx, TFHE_HIP_ERR_ARG, "device-pointer entry points need a single-device context");
  if (B == 0) return 0;
  if (!d || !s || !out) return fail(ctx, TFHE_HIP_ERR_ARG, "polynomial pointer is NULL");
  if (B > (size_t)INT32_MAX) return fail(ctx, TFHE_HIP_ERR_ARG, "batch too large");
  HIP_TRY(ctx, hipSetDevice(ctx->device));
  const unsigned grid64 = (unsigned)std::min<size_t>((B + kWaves64 - 1) / kWaves64, (size_t)ctx->num_cus);
  if (accumulate)
    hipLaunchKernelGGL((polymul64_kernel<true, false>), dim3(grid64), dim3(64 * kWaves64), 0, ctx->stream, d, s, ctx->tables64, out, (double*)nullptr, (int)B);
  else
    hipLaunchKernelGGL((polymul64_kernel<false, false>), dim3(grid64), dim3(64 * kWaves64), 0, ctx->stream, d, s, ctx->tables64, out, (double*)nullptr, (int)B);
  HIP_TRY(ctx, hipGetLastError());
  return 0;
}

// circuits::add (src/circuits/circuits.rs:105-120) on `count` independent
// operand pairs, level by level: x = xor(a, b) and g = and(a, b) of every bit
// of every instance first (they do not depend on the carry), then per bit
// sum = xor(x, carry) and t = and(carry, x) in one blind rotation, and
// carry' = or(t, g) (full_adder, circuits.rs:83-97).
int tfhe_hip_circuit_add_dev(tfhe_hip_ctx* ctx, const int32_t* a_a, const int32_t* a_b,
                             const int32_t* b_a, const int32_t* b_b, int32_t* sum_a, int32_t* sum_b,
                             int32_t* carry_a, int32_t* carry_b, double* sum_var, double* carry_var,
                             size_t count, int bits) {
  if (int rc = need_keys(ctx)) return rc;
  if (bits < 0) return fail(ctx, TFHE_HIP_ERR_ARG, "bits must be >= 0");
  if (count == 0) return 0;
  if (!carry_a || !carry_b || (bits > 0 && (!a_a || !a_b || !b_a || !b_b || !sum_a || !sum_b)))
    return fail(ctx, TFHE_HIP_ERR_ARG, "sample pointer is NULL");
  const size_t n = ctx->P.n, sa = count * n, sb = count;      // one bit slice: masks, bodies
  // workspace: x, g [bits] slices; t, carry ping-pong: 3 slices
  const size_t slice = sa + sb;
  if (int rc = ensure_work(ctx, (2 * (size_t)bits + 3) * slice * sizeof(int32_t))) return rc;
  int32_t* w = (int32_t*)ctx->work;
  int32_t *x_a = w, *x_b = x_a + (size_t)bits * sa;
  int32_t *g_a = x_b + (size_t)bits * sb, *g_b = g_a + (size_t)bits * sa;
  int32_t *t_a = g_b + (size_t)bits * sb, *t_b = t_a + sa;
  int32_t *c0_a = t_b + sb, *c0_b = c0_a + sa, *c1_a = c0_b + sb, *c1_b = c1_a + sa;
  // carry-independent gates of all bits, in groups of at most kMaxGroup ciphertexts
  constexpr size_t kMaxGroup = 65536;
  const int bits_per_group = (int)std::max<size_t>(1, kMaxGroup / (2 * count));
  for (int i0 = 0; i0 < bits; i0 += bits_per_group) {
    const size_t nb = (size_t)std::min(bits_per_group, bits - i0);
    const GateItem items[2] = {
        {TFHE_HIP_XOR, a_a + i0 * sa, a_b + i0 * sb, b_a + i0 * sa, b_b + i0 * sb, x_a + i0 * sa, x_b + i0 * sb, nullptr, nb * count},
        {TFHE_HIP_AND, a_a + i0 * sa, a_b + i0 * sb, b_a + i0 * sa, b_b + i0 * sb, g_a + i0 * sa, g_b + i0 * sb, nullptr, nb * count}};
    if (int rc = run_gate_group(ctx, items, 2)) return rc;
  }
  // carry_in = constant(false) = trivial(encode_message(-1, 8)), src/gates.rs:12-19
  int32_t *cin_a = c0_a, *cin_b = c0_b, *cout_a = c1_a, *cout_b = c1_b;
  if (bits == 0) { cin_a = carry_a; cin_b = carry_b; }
  HIP_TRY(ctx, hipMemsetAsync(cin_a, 0, sa * sizeof(int32_t), ctx->stream));
  hipLaunchKernelGGL(fill_kernel, dim3((unsigned)((sb + 255) / 256)), dim3(256), 0, ctx->stream, cin_b,
                     -(1 << 29), sb);
  HIP_TRY(ctx, hipGetLastError());
  for (int i = 0; i < bits; i++) {
    const bool last = i == bits - 1;
    const GateItem lvl1[2] = {
        {TFHE_HIP_XOR, x_a + i * sa, x_b + i * sb, cin_a, cin_b, sum_a + i * sa, sum_b + i * sb,
         sum_var ? sum_var + i * sb : nullptr, count},
        {TFHE_HIP_AND, cin_a, cin_b, x_a + i * sa, x_b + i * sb, t_a, t_b, nullptr, count}};
    if (int rc = run_gate_group(ctx, lvl1, 2)) return rc;
    const GateItem lvl2 = {TFHE_HIP_OR, t_a, t_b, g_a + i * sa, g_b + i * sb, last ? carry_a : cout_a,
                           last ? carry_b : cout_b, last ? carry_var : nullptr, count};
    if (int rc = run_gate_group(ctx, &lvl2, 1)) return rc;
    std::swap(cin_a, cout_a);
    std::swap(cin_b, cout_b);
  }
  if (bits == 0 && carry_var) HIP_TRY(ctx, hipMemsetAsync(carry_var, 0, sb * sizeof(double), ctx->stream));
  return 0;
}


// circuits::compare (src/circuits/circuits.rs:33-47: a <= b) on `count` independent
// operand pairs: condition_i = xor(a_i, b_i) of every bit of every instance in one
// blind rotation, then per bit carry = mux(condition_i, b_i, carry) starting from
// constant(true) (le, circuits.rs:44-47), one blind rotation of 2 count ciphertexts each.
int tfhe_hip_circuit_compare_dev(tfhe_hip_ctx* ctx, const int32_t* a_a, const int32_t* a_b,
                                 const int32_t* b_a, const int32_t* b_b, int32_t* out_a,
                                 int32_t* out_b, double* out_var, size_t count, int bits) {
  if (int rc = need_keys(ctx)) return rc;
  if (bits < 0) return fail(ctx, TFHE_HIP_ERR_ARG, "bits must be >= 0");
  if (count == 0) return 0;
  if (!out_a || !out_b || (bits > 0 && (!a_a || !a_b || !b_a || !b_b)))
    return fail(ctx, TFHE_HIP_ERR_ARG, "sample pointer is NULL");
  const size_t n = ctx->P.n, sa = count * n, sb = count, slice = sa + sb;
  // workspace: condition [bits] slices, carry ping-pong 2 slices
  if (int rc = ensure_work(ctx, ((size_t)bits + 2) * slice * sizeof(int32_t))) return rc;
  int32_t* w = (int32_t*)ctx->work;
  int32_t *x_a = w, *x_b = x_a + (size_t)bits * sa;
  int32_t *c0_a = x_b + (size_t)bits * sb, *c0_b = c0_a + sa, *c1_a = c0_b + sb, *c1_b = c1_a + sa;
  constexpr size_t kMaxGroup = 65536;
  const int bits_per_group = (int)std::max<size_t>(1, kMaxGroup / count);
  for (int i0 = 0; i0 < bits; i0 += bits_per_group) {
    const size_t nb = (size_t)std::min(bits_per_group, bits - i0);
    const GateItem item = {TFHE_HIP_XOR, a_a + i0 * sa, a_b + i0 * sb, b_a + i0 * sa, b_b + i0 * sb,
                           x_a + i0 * sa, x_b + i0 * sb, nullptr, nb * count};
    if (int rc = run_gate_group(ctx, &item, 1)) return rc;
  }
  int32_t *cin_a = bits == 0 ? out_a : c0_a, *cin_b = bits == 0 ? out_b : c0_b, *cout_a = c1_a, *cout_b = c1_b;
  if (int rc = tfhe_hip_constant_dev(ctx, 1, cin_a, cin_b, bits == 0 ? out_var : nullptr, count)) return rc;
  for (int i = 0; i < bits; i++) {
    const bool last = i == bits - 1;
    if (int rc = gate_on_device(ctx, false, TFHE_HIP_MUX, x_a + i * sa, x_b + i * sb, b_a + i * sa,
                                b_b + i * sb, cin_a, cin_b, last ? out_a : cout_a, last ? out_b : cout_b,
                                last ? out_var : nullptr, count)) return rc;
    std::swap(cin_a, cout_a);
    std::swap(cin_b, cout_b);
  }
  return 0;
}

// circuits::eq (src/circuits/circuits.rs:50-63) on `count` independent operand pairs:
// eq_bit(a_i, b_i) = xor(xor(a_i, b_i), constant(true)) (circuits.rs:66-70) of every bit of
// every instance in two blind rotations, then the fold res = and(res, bit_i) from
// constant(true), one blind rotation of count ciphertexts per bit.  The reference asserts
// that the operands are not empty.
int tfhe_hip_circuit_eq_dev(tfhe_hip_ctx* ctx, const int32_t* a_a, const int32_t* a_b,
                            const int32_t* b_a, const int32_t* b_b, int32_t* out_a, int32_t* out_b,
                            double* out_var, size_t count, int bits) {
  if (int rc = need_keys(ctx)) return rc;
  if (bits < 1) return fail(ctx, TFHE_HIP_ERR_ARG, "eq needs at least one bit");
  if (count == 0) return 0;
  if (!out_a || !out_b || !a_a || !a_b || !b_a || !b_b)
    return fail(ctx, TFHE_HIP_ERR_ARG, "sample pointer is NULL");
  const size_t n = ctx->P.n, sa = count * n, sb = count, slice = sa + sb;
  constexpr size_t kMaxGroup = 65536;
  const int bits_per_group = (int)std::max<size_t>(1, kMaxGroup / count);
  const size_t gb = (size_t)std::min(bits_per_group, bits);       // bits of the largest group
  // workspace: x, e [bits] slices; constant(true) for one group; res ping-pong 2 slices
  if (int rc = ensure_work(ctx, (2 * (size_t)bits + gb + 2) * slice * sizeof(int32_t))) return rc;
  int32_t* w = (int32_t*)ctx->work;
  int32_t *x_a = w, *x_b = x_a + (size_t)bits * sa;
  int32_t *e_a = x_b + (size_t)bits * sb, *e_b = e_a + (size_t)bits * sa;
  int32_t *one_a = e_b + (size_t)bits * sb, *one_b = one_a + gb * sa;
  int32_t *r0_a = one_b + gb * sb, *r0_b = r0_a + sa, *r1_a = r0_b + sb, *r1_b = r1_a + sa;
  if (int rc = tfhe_hip_constant_dev(ctx, 1, one_a, one_b, nullptr, gb * count)) return rc;
  for (int i0 = 0; i0 < bits; i0 += bits_per_group) {
    const size_t nb = (size_t)std::min(bits_per_group, bits - i0);
    const GateItem x = {TFHE_HIP_XOR, a_a + i0 * sa, a_b + i0 * sb, b_a + i0 * sa, b_b + i0 * sb,
                        x_a + i0 * sa, x_b + i0 * sb, nullptr, nb * count};
    if (int rc = run_gate_group(ctx, &x, 1)) return rc;
    const GateItem e = {TFHE_HIP_XOR, x_a + i0 * sa, x_b + i0 * sb, one_a, one_b, e_a + i0 * sa,
                        e_b + i0 * sb, nullptr, nb * count};
    if (int rc = run_gate_group(ctx, &e, 1)) return rc;
  }
  // the fold starts from constant(true): the first `count` entries of `one`
  const int32_t *res_a = one_a, *res_b = one_b;
  int32_t *nxt_a = r0_a, *nxt_b = r0_b, *alt_a = r1_a, *alt_b = r1_b;
  for (int i = 0; i < bits; i++) {
    const bool last = i == bits - 1;
    const GateItem g = {TFHE_HIP_AND, res_a, res_b, e_a + i * sa, e_b + i * sb, last ? out_a : nxt_a,
                        last ? out_b : nxt_b, last ? out_var : nullptr, count};
    if (int rc = run_gate_group(ctx, &g, 1)) return rc;
    res_a = nxt_a; res_b = nxt_b;
    std::swap(nxt_a, alt_a);
    std::swap(nxt_b, alt_b);
  }
  return 0;
}

int tfhe_hip_not_dev(tfhe_hip_ctx* ctx, const int32_t* a, const int32_t* b, const double* var,
                     int32_t* out_a, int32_t* out_b, double* var_out, size_t B) {
  if (int rc = need_keys(ctx)) return rc;
  if (B == 0) return 0;
  if (!a || !b || !out_a || !out_b) return fail(ctx, TFHE_HIP_ERR_ARG, "sample pointer is NULL");
  const size_t na = B * (size_t)ctx->P.n, total = na + B;
  hipLaunchKernelGGL(not_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, ctx->stream, a, b, var,
                     out_a, out_b, var_out, na, B);
  HIP_TRY(ctx, hipGetLastError());
  return 0;
}

int tfhe_hip_constant_dev(tfhe_hip_ctx* ctx, int value, int32_t* out_a, int32_t* out_b,
                          double* var_out, size_t B) {
  if (int rc = need_keys(ctx)) return rc;
  if (B == 0) return 0;
  if (!out_a || !out_b) return fail(ctx, TFHE_HIP_ERR_ARG, "sample pointer is NULL");
  // LweSample::trivial(encode_message(+-1, 8)), src/gates.rs:12-20: zero mask, variance 0
  HIP_TRY(ctx, hipMemsetAsync(out_a, 0, B * (size_t)ctx->P.n * sizeof(int32_t), ctx->stream));
  hipLaunchKernelGGL(fill_kernel, dim3((unsigned)((B + 255) / 256)), dim3(256), 0, ctx->stream, out_b,
                     value ? (1 << 29) : -(1 << 29), B);
  HIP_TRY(ctx, hipGetLastError());
  if (var_out) HIP_TRY(ctx, hipMemsetAsync(var_out, 0, B * sizeof(double), ctx->stream));
  return 0;
}

int tfhe_hip_polymul_margin_dev(tfhe_hip_ctx* ctx, const int32_t* d, const int32_t* s, double* margin,
                                size_t B) {
  if (!ctx) return TFHE_HIP_ERR_ARG;
  if (!ctx->shards.empty())
    return fail(ctx, TFHE_HIP_ERR_ARG, "device-pointer entry points need a single-device context");
  if (B == 0) return 0;
  if (!d || !s || !margin) return fail(ctx, TFHE_HIP_ERR_ARG, "polynomial pointer is NULL");
  if (B > (size_t)INT32_MAX) return fail(ctx, TFHE_HIP_ERR_ARG, "batch too large");
  HIP_TRY(ctx, hipSetDevice(ctx->device));
  const unsigned grid64 = (unsigned)std::min<size_t>((B + kWaves64 - 1) / kWaves64, (size_t)ctx->num_cus);
  hipLaunchKernelGGL((polymul64_kernel<false, true>), dim3(grid64), dim3(64 * kWaves64), 0, ctx->stream, d, s,
                       ctx->tables64, (int32_t*)nullptr, margin, (int)B);
  HIP_TRY(ctx, hipGetLastError());
  return 0;
}

int tfhe_hip_extprod_dev(tfhe_hip_ctx* ctx, const int32_t* t, int bk_index, int32_t* out,
                         double* margin, size_t B) {
  if (int rc = need_keys(ctx)) return rc;
#ifdef TFHE_TEXTBOOK
  return fail(ctx, TFHE_HIP_ERR_ARG, "the external-product hook is not part of the textbook build");
#endif
  if (B == 0) return 0;
  if (!t || !out) return fail(ctx, TFHE_HIP_ERR_ARG, "sample pointer is NULL");
  if (bk_index < 0 || bk_index >= ctx->P.n) return fail(ctx, TFHE_HIP_ERR_ARG, "bk_index out of range");
  if (B > (size_t)INT32_MAX) return fail(ctx, TFHE_HIP_ERR_ARG, "batch too large");
  const cplx* bk_i = ctx->bkfft + (size_t)bk_index * ctx->P.l * 2 * 2048;
  const dim3 grid((unsigned)((B + kWavesPerBlock - 1) / kWavesPerBlock)), block(kBlock);
  const bool l2 = ctx->P.l == 2;
  if (margin) {
    if (l2) hipLaunchKernelGGL((extprod_kernel<2, 10, true>), grid, block, 0, ctx->stream, t, bk_i, ctx->tables, out, margin, (int)B);
    else hipLaunchKernelGGL((extprod_kernel<3, 7, true>), grid, block, 0, ctx->stream, t, bk_i, ctx->tables, out, margin, (int)B);
  } else {
    if (l2) hipLaunchKernelGGL((extprod_kernel<2, 10, false>), grid, block, 0, ctx->stream, t, bk_i, ctx->tables, out, margin, (int)B);
    else hipLaunchKernelGGL((extprod_kernel<3, 7, false>), grid, block, 0, ctx->stream, t, bk_i, ctx->tables, out, margin, (int)B);
  }
  HIP_TRY(ctx, hipGetLastError());
  return 0;
}

int tfhe_hip_blind_rotate_dev(tfhe_hip_ctx* ctx, int32_t* acc, const int32_t* bara, size_t B) {
  if (int rc = need_keys(ctx)) return rc;
  if (B == 0) return 0;
  if (!acc || !bara) return fail(ctx, TFHE_HIP_ERR_ARG, "pointer is NULL");
  if (B > (size_t)INT32_MAX) return fail(ctx, TFHE_HIP_ERR_ARG, "batch too large");
  return launch_blind_rotate(ctx, bara, nullptr, 0, acc, nullptr, nullptr, B);
}

}  // extern "C"
