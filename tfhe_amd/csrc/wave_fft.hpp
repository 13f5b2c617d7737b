// wave_fft.hpp — f64 FFT machinery for one 64-lane wavefront on gfx950.
//
// A half-wave (32 lanes) owns one 1024-point complex transform; the two halves of
// a wavefront run two independent transforms in the same instructions.  Element
// n = q + 32*r of a 1024-point sequence lives in lane q, register r, and the
// 1024-point DFT is the four-step 32 x 32 decomposition:
//
//   32-point DFT over the 32 registers of each lane      (straight-line code)
//   twiddle by w1024^(q*kb)                              (table in LDS)
//   32 x 32 transpose between lanes and registers        (through LDS, swizzled)
//   32-point DFT over the registers again
//
// Real and imaginary parts are transposed one after the other through the same
// 8 KiB tile, which halves the exchange area (the other half of a wavefront's
// LDS share holds per-ciphertext state of the caller).
//
// Products of real polynomials use the TWISTED transform on top of it
// (fwd1024_twisted_real / inv1024_twisted_real): the values of a polynomial at
// the 1024 points w4096^(4k+1), which determine a real polynomial of degree
// < 2048, so a linear convolution of two polynomials of 1024 coefficients needs
// no zero padding and one 1024-point transform per polynomial.
//
// No workgroup barrier is used anywhere: every exchange is between lanes of
// the same wavefront, whose LDS operations execute in program order.
//
// Register-index conventions (br = 5-bit bit reversal):
//   "natural"  : register r holds element q + 32*r
//   "S-layout" : register p holds element q + 32*br(p)   (spectra are kept so)
// Forward transforms go natural -> S-layout, inverse transforms S-layout ->
// natural, so no explicit reordering is ever executed.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include <type_traits>

// Group size of the software pipelines that fetch twiddles from LDS (tuned).
#ifndef WFFT_TW_GROUP
#define WFFT_TW_GROUP 8
#endif
#ifndef WFFT_TWIST_GROUP
#define WFFT_TWIST_GROUP 8
#endif
// Scheduling fences: between the groups of a software pipeline (1/0), and
// every WFFT_TAIL_CHUNK elements of the inverse transform's element-wise tail
// (0 = none).
#ifndef WFFT_PIPE_BARRIER
#define WFFT_PIPE_BARRIER 1
#endif
#ifndef WFFT_TAIL_CHUNK
#define WFFT_TAIL_CHUNK 0
#endif
// Butterflies that multiply as interleaved inline-asm pairs (1) or plain C++ (0).
#ifndef WFFT_ASM_DIF
#define WFFT_ASM_DIF 0
#endif
#ifndef WFFT_ASM_DIT
#define WFFT_ASM_DIT 1
#endif

// Default of the twisted transforms' SCALAR parameter.  1: the register part of the twist
// w4096^n (n = q + 32 r -> w128^r) is read with scalar loads from the constant segment and the
// lane part w4096^q is folded into the inter-pass twiddle tables; 0: the whole twist comes from
// a table in LDS.  WFFT_SCALAR_GROUP registers share one scalar load (4 -> s_load_dwordx16; 2 and
// 8 measured slower, 8 through scalar register spills).
#ifndef WFFT_SCALAR_TWIST
#define WFFT_SCALAR_TWIST 1
#endif

namespace wfft {

// (cos, -sin)(2 pi r / 128), r = 0..31; written by the host when a context is created
__constant__ double g_w128[64];

#ifndef WFFT_SCALAR_GROUP
#define WFFT_SCALAR_GROUP 4
#endif

// The address of g_w128 (constant address space, so that loads from it are scalar loads) in a
// scalar register the optimiser cannot see through: the loads then stay where they are written
// instead of being hoisted out of the enclosing loop (all 64 values live at once do not fit the
// scalar register file).
typedef const double __attribute__((address_space(4))) * const_seg_ptr;
__device__ __forceinline__ const_seg_ptr w128_ptr() {
  const_seg_ptr p = (const_seg_ptr)g_w128;
  asm volatile("" : "+s"(p));
  return p;
}

constexpr int kN = 1024;           // polynomial size
constexpr int kHalfBytes = 8192;       // one 32x32 f64 tile (real OR imaginary parts of a half-wave)
constexpr int kExchangeBytes = 16384;  // exchange area of one wavefront (two halves)
constexpr int kStateBytes = 16384;     // caller's per-wavefront state behind the exchange area
constexpr int kWaveBytes = kExchangeBytes + kStateBytes;

struct alignas(16) cplx {
  double x, y;
};

// Every lambda handed to static_for must be inlined at every call site: a
// register array captured by reference by an out-of-line lambda is forced
// into scratch memory.
#define WFFT_INLINE __attribute__((always_inline))

template <int I, int END, class F>
__device__ __forceinline__ void static_for(F&& f) {
  if constexpr (I < END) {
    f(std::integral_constant<int, I>{});
    static_for<I + 1, END>(f);
  }
}

// static_for over [0, 32) in chunks of CH iterations; the compiler may not move
// instructions across chunk boundaries.  Bounds how many loads it hoists at
// once, i.e. the register pressure of load-heavy loops.
template <int CH, class F>
__device__ __forceinline__ void chunked_for32(F&& f) {
  static_for<0, 32 / CH>([&](auto cc) WFFT_INLINE {
    constexpr int c = decltype(cc)::value;
    static_for<0, CH>([&](auto ic) WFFT_INLINE {
      f(std::integral_constant<int, c * CH + decltype(ic)::value>{});
    });
    __builtin_amdgcn_sched_barrier(0);
  });
}

// Software-pipelined pass over [0, 32) in groups of G: load(i) for group g + D
// is issued before use(i, value) runs for group g, so that the latency of the
// loads (LDS tables, key slices) hides behind the arithmetic of D groups.  The
// compiler may not move instructions across group boundaries, which keeps at
// most (D + 1) G loaded values live.  prologue() issues the first D groups and
// may be called well before run().
template <int G, int D, class T>
struct Pipe32 {
  T buf[D + 1][G];
  template <class LoadF>
  __device__ __forceinline__ void prologue(LoadF&& load) {
    static_for<0, D * G>([&](auto ic) WFFT_INLINE {
      constexpr int i = decltype(ic)::value;
      buf[i / G][i % G] = load(ic);
    });
  }
  // As run, but use2(i, value_i, i + 1, value_{i+1}) takes two consecutive elements.
  template <class LoadF, class Use2F>
  __device__ __forceinline__ void run2(LoadF&& load, Use2F&& use2) {
    static_assert(G % 2 == 0, "pairs");
    static_for<0, 32 / G>([&](auto gc) WFFT_INLINE {
      constexpr int g = decltype(gc)::value;
      if constexpr (g + D < 32 / G) {
        static_for<0, G>([&](auto ic) WFFT_INLINE {
          constexpr int e = decltype(ic)::value;
          buf[(g + D) % (D + 1)][e] = load(std::integral_constant<int, (g + D) * G + e>{});
        });
      }
      static_for<0, G / 2>([&](auto ic) WFFT_INLINE {
        constexpr int e = 2 * decltype(ic)::value;
        use2(std::integral_constant<int, g * G + e>{}, buf[g % (D + 1)][e],
             std::integral_constant<int, g * G + e + 1>{}, buf[g % (D + 1)][e + 1]);
      });
#if WFFT_PIPE_BARRIER
      __builtin_amdgcn_sched_barrier(0);
#endif
    });
  }
  template <class LoadF, class UseF>
  __device__ __forceinline__ void run(LoadF&& load, UseF&& use) {
    static_for<0, 32 / G>([&](auto gc) WFFT_INLINE {
      constexpr int g = decltype(gc)::value;
      if constexpr (g + D < 32 / G) {
        static_for<0, G>([&](auto ic) WFFT_INLINE {
          constexpr int e = decltype(ic)::value;
          buf[(g + D) % (D + 1)][e] = load(std::integral_constant<int, (g + D) * G + e>{});
        });
      }
      static_for<0, G>([&](auto ic) WFFT_INLINE {
        constexpr int e = decltype(ic)::value;
        use(std::integral_constant<int, g * G + e>{}, buf[g % (D + 1)][e]);
      });
#if WFFT_PIPE_BARRIER
      __builtin_amdgcn_sched_barrier(0);
#endif
    });
  }
};

// As pipelined_for32, but use2(i, value_i, i + 1, value_{i+1}) takes two
// consecutive elements at a time.
template <int G, class T, class LoadF, class Use2F>
__device__ __forceinline__ void pipelined_pairs32(LoadF&& load, Use2F&& use2) {
  Pipe32<G, 1, T> pipe;
  pipe.prologue(load);
  pipe.run2(load, use2);
}

template <int G, class T, class LoadF, class UseF>
__device__ __forceinline__ void pipelined_for32(LoadF&& load, UseF&& use) {
  Pipe32<G, 1, T> pipe;
  pipe.prologue(load);
  pipe.run(load, use);
}

__host__ __device__ constexpr int br5(int v) {
  return ((v & 1) << 4) | ((v & 2) << 2) | (v & 4) | ((v & 8) >> 2) | ((v & 16) >> 4);
}

// cos(2*pi*j/32), sin(2*pi*j/32) for j = 0..15, symmetric by construction.
#define WFFT_C1 0.98078528040323043
#define WFFT_C2 0.92387953251128674
#define WFFT_C3 0.83146961230254524
#define WFFT_C4 0.70710678118654752
#define WFFT_C5 0.55557023301960222
#define WFFT_C6 0.38268343236508977
#define WFFT_C7 0.19509032201612827
__host__ __device__ constexpr double cos32(int j) {
  constexpr double t[17] = {1.0, WFFT_C1, WFFT_C2, WFFT_C3, WFFT_C4, WFFT_C5, WFFT_C6, WFFT_C7, 0.0,
                            -WFFT_C7, -WFFT_C6, -WFFT_C5, -WFFT_C4, -WFFT_C3, -WFFT_C2, -WFFT_C1, -1.0};
  return t[j];
}
__host__ __device__ constexpr double sin32(int j) {
  constexpr double t[17] = {0.0, WFFT_C7, WFFT_C6, WFFT_C5, WFFT_C4, WFFT_C3, WFFT_C2, WFFT_C1, 1.0,
                            WFFT_C1, WFFT_C2, WFFT_C3, WFFT_C4, WFFT_C5, WFFT_C6, WFFT_C7, 0.0};
  return t[j];
}

// (dr + i di) * w32^(+-TJ); forward uses exp(-2 pi i TJ/32), inverse its conjugate.
template <int TJ, bool INV>
__device__ __forceinline__ void mul_w32(double dr, double di, double& outr, double& outi) {
  if constexpr (TJ == 0) {
    outr = dr; outi = di;
  } else if constexpr (TJ == 8) {
    if constexpr (!INV) { outr = di; outi = -dr; } else { outr = -di; outi = dr; }
  } else if constexpr (TJ == 4) {
    if constexpr (!INV) { outr = (dr + di) * WFFT_C4; outi = (di - dr) * WFFT_C4; }
    else { outr = (dr - di) * WFFT_C4; outi = (di + dr) * WFFT_C4; }
  } else if constexpr (TJ == 12) {
    if constexpr (!INV) { outr = (di - dr) * WFFT_C4; outi = -(di + dr) * WFFT_C4; }
    else { outr = -(dr + di) * WFFT_C4; outi = (dr - di) * WFFT_C4; }
  } else {
    constexpr double c = cos32(TJ);
    constexpr double s = INV ? sin32(TJ) : -sin32(TJ);   // w = c + i s
    outr = dr * c - di * s;
    outi = dr * s + di * c;
  }
}

// Two complex products x <- x * w (CONJ: x * conj(w)) with the four
// multiplications issued before the four fused multiply-adds that consume
// them: a vector instruction whose operand was produced by the instruction
// right before it issues 3 clocks late on a SIMD that runs a single wavefront,
// and the compiler pairs each multiplication with its consumer.
template <bool CONJ>
__device__ __forceinline__ void cmul2(double& ar, double& ai, const cplx& wa, double& br, double& bi,
                                      const cplx& wb) {
  double t0, t1, t2, t3;
  if constexpr (!CONJ)
    asm("v_mul_f64 %4, %1, %9\n\tv_mul_f64 %5, %0, %9\n\tv_mul_f64 %6, %3, %11\n\tv_mul_f64 %7, %2, %11\n\t"
        "v_fma_f64 %0, %0, %8, -%4\n\tv_fma_f64 %1, %1, %8, %5\n\t"
        "v_fma_f64 %2, %2, %10, -%6\n\tv_fma_f64 %3, %3, %10, %7"
        : "+v"(ar), "+v"(ai), "+v"(br), "+v"(bi), "=&v"(t0), "=&v"(t1), "=&v"(t2), "=&v"(t3)
        : "v"(wa.x), "v"(wa.y), "v"(wb.x), "v"(wb.y));
  else
    asm("v_mul_f64 %4, %1, %9\n\tv_mul_f64 %5, %0, %9\n\tv_mul_f64 %6, %3, %11\n\tv_mul_f64 %7, %2, %11\n\t"
        "v_fma_f64 %0, %0, %8, %4\n\tv_fma_f64 %1, %1, %8, -%5\n\t"
        "v_fma_f64 %2, %2, %10, %6\n\tv_fma_f64 %3, %3, %10, -%7"
        : "+v"(ar), "+v"(ai), "+v"(br), "+v"(bi), "=&v"(t0), "=&v"(t1), "=&v"(t2), "=&v"(t3)
        : "v"(wa.x), "v"(wa.y), "v"(wb.x), "v"(wb.y));
}

// Butterflies of one radix-2 stage: b = 0..15, j = b % HALF, operands
// i0 = (b / HALF) * 2 HALF + j and i1 = i0 + HALF, twiddle w32^(+-TJ) with
// TJ = j * (16 / HALF).  TJ = 0 and TJ = 8 (w = 1, -+i) need no multiplication.
template <int HALF>
__host__ __device__ constexpr int stage_tj(int b) { return (b % HALF) * (16 / HALF); }
template <int HALF>
__host__ __device__ constexpr int stage_i0(int b) { return (b / HALF) * 2 * HALF + b % HALF; }
template <int HALF>
__host__ __device__ constexpr int nontrivial_count() {
  int c = 0;
  for (int b = 0; b < 16; b++) c += (stage_tj<HALF>(b) != 0 && stage_tj<HALF>(b) != 8);
  return c;
}
template <int HALF>
__host__ __device__ constexpr int nontrivial_butterfly(int k) {      // k-th butterfly that multiplies
  int c = 0;
  for (int b = 0; b < 16; b++)
    if (stage_tj<HALF>(b) != 0 && stage_tj<HALF>(b) != 8) {
      if (c == k) return b;
      c++;
    }
  return -1;
}

// Two decimation-in-frequency twiddle products d <- d * w (w = c + i s, constants
// in scalar registers), multiplications ahead of the FMAs that consume them.
__device__ __forceinline__ void cmul2_const(double& ar, double& ai, double c0, double s0, double& br,
                                            double& bi, double c1, double s1) {
  double t0, t1, t2, t3;
  asm("v_mul_f64 %4, %1, %9\n\tv_mul_f64 %5, %0, %9\n\tv_mul_f64 %6, %3, %11\n\tv_mul_f64 %7, %2, %11\n\t"
      "v_fma_f64 %0, %0, %8, -%4\n\tv_fma_f64 %1, %1, %8, %5\n\t"
      "v_fma_f64 %2, %2, %10, -%6\n\tv_fma_f64 %3, %3, %10, %7"
      : "+v"(ar), "+v"(ai), "+v"(br), "+v"(bi), "=&v"(t0), "=&v"(t1), "=&v"(t2), "=&v"(t3)
      : "s"(c0), "s"(s0), "s"(c1), "s"(s1));
}

// Two decimation-in-time butterflies a' = a + w b, b' = a - w b = 2 a - a' in six
// FMAs each (instead of a complex product and four additions), interleaved so
// that every instruction's producer is four instructions back.  The new a comes
// back in fresh registers (na), b is updated in place.
__device__ __forceinline__ void dit_butterfly2(double a0r, double a0i, double& b0r, double& b0i, double c0,
                                               double s0, double& n0r, double& n0i, double a1r, double a1i,
                                               double& b1r, double& b1i, double c1, double s1, double& n1r,
                                               double& n1i) {
  asm("v_fma_f64 %0, -%5, %13, %8\n\tv_fma_f64 %1, %5, %12, %9\n\t"
      "v_fma_f64 %2, -%7, %15, %10\n\tv_fma_f64 %3, %7, %14, %11\n\t"
      "v_fma_f64 %0, %4, %12, %0\n\tv_fma_f64 %1, %4, %13, %1\n\t"
      "v_fma_f64 %2, %6, %14, %2\n\tv_fma_f64 %3, %6, %15, %3\n\t"
      "v_fma_f64 %4, 2.0, %8, -%0\n\tv_fma_f64 %5, 2.0, %9, -%1\n\t"
      "v_fma_f64 %6, 2.0, %10, -%2\n\tv_fma_f64 %7, 2.0, %11, -%3"
      : "=&v"(n0r), "=&v"(n0i), "=&v"(n1r), "=&v"(n1i), "+v"(b0r), "+v"(b0i), "+v"(b1r), "+v"(b1i)
      : "v"(a0r), "v"(a0i), "v"(a1r), "v"(a1i), "s"(c0), "s"(s0), "s"(c1), "s"(s1));
}

// Decimation in frequency, natural in -> bit-reversed out.
template <int HALF, bool INV>
__device__ __forceinline__ void dif_stage(double (&re)[32], double (&im)[32]) {
  static_for<0, 16>([&](auto ic) WFFT_INLINE {
    constexpr int b = decltype(ic)::value;
    constexpr int TJ = stage_tj<HALF>(b);
    constexpr int i0 = stage_i0<HALF>(b), i1 = i0 + HALF;
    const double ar = re[i0], ai = im[i0], br_ = re[i1], bi = im[i1];
    re[i0] = ar + br_;
    im[i0] = ai + bi;
    if constexpr (TJ == 0 || TJ == 8 || !WFFT_ASM_DIF) mul_w32<TJ, INV>(ar - br_, ai - bi, re[i1], im[i1]);
    else { re[i1] = ar - br_; im[i1] = ai - bi; }
  });
  static_for<0, WFFT_ASM_DIF ? nontrivial_count<HALF>() / 2 : 0>([&](auto kc) WFFT_INLINE {
    constexpr int b0 = nontrivial_butterfly<HALF>(2 * decltype(kc)::value);
    constexpr int b1 = nontrivial_butterfly<HALF>(2 * decltype(kc)::value + 1);
    constexpr int x0 = stage_i0<HALF>(b0) + HALF, x1 = stage_i0<HALF>(b1) + HALF;
    constexpr int T0 = stage_tj<HALF>(b0), T1 = stage_tj<HALF>(b1);
    cmul2_const(re[x0], im[x0], cos32(T0), INV ? sin32(T0) : -sin32(T0), re[x1], im[x1], cos32(T1),
                INV ? sin32(T1) : -sin32(T1));
  });
}

// Decimation in time, bit-reversed in -> natural out.  With VIEW_BR the stage
// works on position i = register br5(i): registers in natural order are then
// the bit-reversed input a decimation in time wants, and output k lands in
// register br5(k), i.e. the transform maps the natural layout to the S-layout
// exactly as the decimation-in-frequency passes do, with the cheaper butterflies.
template <int HALF, bool INV, bool VIEW_BR = false>
__device__ __forceinline__ void dit_stage(double (&re)[32], double (&im)[32]) {
  static_for<0, 16>([&](auto ic) WFFT_INLINE {
    constexpr int b = decltype(ic)::value;
    constexpr int TJ = stage_tj<HALF>(b);
    if constexpr (TJ == 0 || TJ == 8 || !WFFT_ASM_DIT) {
      constexpr int p0 = stage_i0<HALF>(b), p1 = p0 + HALF;
      constexpr int i0 = VIEW_BR ? br5(p0) : p0, i1 = VIEW_BR ? br5(p1) : p1;
      double tr, ti;
      mul_w32<TJ, INV>(re[i1], im[i1], tr, ti);
      const double ar = re[i0], ai = im[i0];
      re[i0] = ar + tr;
      im[i0] = ai + ti;
      re[i1] = ar - tr;
      im[i1] = ai - ti;
    }
  });
  static_for<0, WFFT_ASM_DIT ? nontrivial_count<HALF>() / 2 : 0>([&](auto kc) WFFT_INLINE {
    constexpr int b0 = nontrivial_butterfly<HALF>(2 * decltype(kc)::value);
    constexpr int b1 = nontrivial_butterfly<HALF>(2 * decltype(kc)::value + 1);
    constexpr int q0 = stage_i0<HALF>(b0), q1 = stage_i0<HALF>(b1);
    constexpr int x0 = VIEW_BR ? br5(q0) : q0, y0 = VIEW_BR ? br5(q0 + HALF) : q0 + HALF;
    constexpr int x1 = VIEW_BR ? br5(q1) : q1, y1 = VIEW_BR ? br5(q1 + HALF) : q1 + HALF;
    constexpr int T0 = stage_tj<HALF>(b0), T1 = stage_tj<HALF>(b1);
    double n0r, n0i, n1r, n1i;
    dit_butterfly2(re[x0], im[x0], re[y0], im[y0], cos32(T0), INV ? sin32(T0) : -sin32(T0), n0r, n0i,
                   re[x1], im[x1], re[y1], im[y1], cos32(T1), INV ? sin32(T1) : -sin32(T1), n1r, n1i);
    re[x0] = n0r; im[x0] = n0i; re[x1] = n1r; im[x1] = n1i;
  });
}

template <bool INV>
__device__ __forceinline__ void dif32(double (&re)[32], double (&im)[32]) {
  dif_stage<16, INV>(re, im);
  dif_stage<8, INV>(re, im);
  dif_stage<4, INV>(re, im);
  dif_stage<2, INV>(re, im);
  dif_stage<1, INV>(re, im);
}

// Forward 32-point transform, natural in -> S-layout out: decimation in
// frequency (0), or decimation in time on the bit-reversed register view (1; same
// layouts, 68 fewer instructions per pass thanks to the six-FMA butterflies).  With
// the twisted transforms (1) is 3 % faster in the blind rotation on MI355X; with the
// zero-padded form this file had before, whose register pressure was higher, it was
// 2 % slower.
#ifndef WFFT_FWD_DIT
#define WFFT_FWD_DIT 1
#endif
__device__ __forceinline__ void fwd32(double (&re)[32], double (&im)[32]) {
#if WFFT_FWD_DIT
  dit_stage<1, false, true>(re, im);
  dit_stage<2, false, true>(re, im);
  dit_stage<4, false, true>(re, im);
  dit_stage<8, false, true>(re, im);
  dit_stage<16, false, true>(re, im);
#else
  dif32<false>(re, im);
#endif
}

template <bool INV>
__device__ __forceinline__ void dit32(double (&re)[32], double (&im)[32]) {
  dit_stage<1, INV>(re, im);
  dit_stage<2, INV>(re, im);
  dit_stage<4, INV>(re, im);
  dit_stage<8, INV>(re, im);
  dit_stage<16, INV>(re, im);
}

// Orders this wavefront's LDS accesses for the compiler.  The hardware runs a
// wavefront's LDS instructions in program order, so no wait is needed.
__device__ __forceinline__ void wave_lds_fence() {
  __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
  __builtin_amdgcn_wave_barrier();
}

__device__ __forceinline__ double lds_load_f64(const char* base, int byte_off) {
  return *reinterpret_cast<const double*>(base + byte_off);
}
__device__ __forceinline__ void lds_store_f64(char* base, int byte_off, double x) {
  *reinterpret_cast<double*>(base + byte_off) = x;
}
struct alignas(16) f64x2 {
  double a, b;
};

// 32 x 32 transpose of one f64 per (lane, register) of a half-wave through its
// 8 KiB tile.  Element (row, col) sits at
//   row*256 + (((col >> 1) ^ (row & 15)) * 2 + (col & 1)) * 8 :
// 16-byte granules hold two adjacent columns of a row and are XOR-swizzled, so
// the row-wise 8-byte stores and the column-wise 16-byte loads (two columns per
// instruction) are both bank-conflict free.  The stores are issued element by
// element by the caller (fused with the twiddle multiplication, so that the
// LDS store path works while the vector ALU does).
template <int ROW>
__device__ __forceinline__ void transpose_store(char* tile, int q, double x) {
  lds_store_f64(tile, ROW * 256 + (((q >> 1) ^ (ROW & 15)) << 4) + ((q & 1) << 3), x);
}
// OUT_BR: register p receives column index br5(p) (else p).
// PAIR16: fetch in the order (0,1), (16,17), (2,3), (18,19), ... (operands of
//         the first decimation-in-frequency stage) instead of ascending; LDS
//         returns data in issue order, so arithmetic can start after two loads.
template <bool OUT_BR, bool PAIR16>
__device__ __forceinline__ void transpose_load(double (&v)[32], const char* tile, int q) {
  wave_lds_fence();
  const char* row = tile + q * 256;
  const int qs = (q & 15) << 4;
  static_for<0, 16>([&](auto ic) WFFT_INLINE {
    constexpr int i = decltype(ic)::value;
    constexpr int cp = PAIR16 ? ((i >> 1) + 8 * (i & 1)) : i;     // column pair (2cp, 2cp+1)
    const f64x2 w = *reinterpret_cast<const f64x2*>(row + ((cp << 4) ^ qs));
    constexpr int p0 = OUT_BR ? br5(2 * cp) : 2 * cp;
    constexpr int p1 = OUT_BR ? br5(2 * cp + 1) : 2 * cp + 1;
    v[p0] = w.a;
    v[p1] = w.b;
  });
  wave_lds_fence();
}
// Sends the 32 registers of v through the tile: v[p] is stored to row ROWOF(p).
template <bool ROW_BR, bool OUT_BR, bool PAIR16>
__device__ __forceinline__ void transpose_f64(double (&v)[32], char* tile, int q) {
  static_for<0, 32>([&](auto ic) WFFT_INLINE {
    constexpr int p = decltype(ic)::value;
    transpose_store<ROW_BR ? br5(p) : p>(tile, q, v[p]);
  });
  transpose_load<OUT_BR, PAIR16>(v, tile, q);
}

// Forward 1024-point DFT of one half-wave.  natural in, S-layout out.
// tE: LDS table, entry [kb*32 + q] = exp(-2 pi i q kb / 1024).
__device__ __forceinline__ void fft1024_fwd(double (&re)[32], double (&im)[32], char* tile,
                                            const cplx* tE, int q) {
  fwd32(re, im);
  // twiddle; the real parts leave for the tile at once, the imaginary parts
  // follow when the real parts have come back (same tile, program order).
  wave_lds_fence();
  pipelined_pairs32<WFFT_TW_GROUP, cplx>(
      [&](auto ic) WFFT_INLINE { return tE[br5(decltype(ic)::value) * 32 + q]; },
      [&](auto i0, const cplx& w0, auto i1, const cplx& w1) WFFT_INLINE {
        constexpr int p0 = decltype(i0)::value, p1 = decltype(i1)::value;
        cmul2<false>(re[p0], im[p0], w0, re[p1], im[p1], w1);      // register 0 takes w = 1
        transpose_store<br5(p0)>(tile, q, re[p0]);
        transpose_store<br5(p1)>(tile, q, re[p1]);
      });
  transpose_load<false, true>(re, tile, q);
  transpose_f64<true, false, true>(im, tile, q);
  fwd32(re, im);
}

// Inverse (unnormalised) 1024-point DFT of one half-wave.  S-layout in, natural out.
__device__ __forceinline__ void fft1024_inv(double (&re)[32], double (&im)[32], char* tile,
                                            const cplx* tE, int q) {
  dit32<true>(re, im);
  wave_lds_fence();
  pipelined_pairs32<WFFT_TW_GROUP, cplx>(
      [&](auto ic) WFFT_INLINE { return tE[decltype(ic)::value * 32 + q]; },
      [&](auto i0, const cplx& w0, auto i1, const cplx& w1) WFFT_INLINE {
        constexpr int k0 = decltype(i0)::value, k1 = decltype(i1)::value;
        cmul2<true>(re[k0], im[k0], w0, re[k1], im[k1], w1);       // conjugate twiddle
        transpose_store<k0>(tile, q, re[k0]);
        transpose_store<k1>(tile, q, re[k1]);
      });
  transpose_load<true, false>(re, tile, q);
  transpose_f64<false, true, false>(im, tile, q);
  dit32<true>(re, im);
}

// ---------------------------------------------------------------------------
// Twisted 1024-point transforms: one REAL polynomial of N = 1024 coefficients per
// half-wave.  A(k) = sum_n a[n] w4096^n w1024^(n k) is the value of a(x) at
// z_k = w4096^(4k+1), k = 0..1023.  These 1024 points and their complex
// conjugates are the 2048 roots of z^2048 = -i, and a real polynomial takes
// conjugate values at conjugate points, so the 1024 values determine a real
// polynomial of degree < 2048: the product c = a b of two real polynomials of
// degree < 1024 has C(k) = A(k) B(k), and the unnormalised inverse transform
// gives y[n] = 1024 (c[n] - i c[n + 1024]) w4096^n (z_k^1024 = -i).  The low half
// of the linear convolution is Re(y[n] conj(w4096^n)) / 1024, the high half
// -Im(...) / 1024.  No zero padding, no separation of packed spectra, and both
// half-waves do the same useful work.
// The twist w4096^n, n = q + 32 r, splits into a lane part w4096^q and a register
// part w128^r.  With SCALAR the register part is 32 wave-uniform
// factors read by scalar loads from g_w128 and the lane part is folded into the
// inter-pass twiddles, which need one table per direction:
//   tE[kb*32 + q]  = w1024^(q kb) w4096^q    (forward)
//   tq[n1*32 + kb] = w1024^(n1 kb) w4096^n1  (inverse, used conjugated)
// Without it tE[kb*32 + q] = w1024^(q kb) serves both directions and
// tq[r*32 + q] = w4096^(q + 32 r) is the whole twist, read from LDS: 96 more
// ds_read_b128 per decomposition level, which cost the one-wavefront-per-ciphertext
// blind rotation 3 % (Bit80) to 7 % (Bit128) on MI355X.  The one-workgroup-per-ciphertext
// kernel, whose wavefronts wait on every dependent step, is 4 % faster with the LDS table
// (a scalar load's latency is exposed there), so the choice is a template parameter.
// ---------------------------------------------------------------------------
// x[n] in re (n = q + 32 r, natural layout) -> spectrum in the S-layout: lane q,
// register p holds A(q + 32 br5(p)).
template <bool SCALAR = (WFFT_SCALAR_TWIST != 0)>
__device__ __forceinline__ void fwd1024_twisted_real(double (&re)[32], double (&im)[32], char* tile,
                                                     const cplx* tE, const cplx* tq, int q) {
  if constexpr (SCALAR) {
    // w4096^(q + 32 r) = w4096^q w128^r: the register part comes from the constant segment
    // (scalar loads, no LDS traffic), the lane part is folded into the inter-pass twiddle tE
    (void)tq;
    static_for<0, 32 / WFFT_SCALAR_GROUP>([&](auto gc) WFFT_INLINE {
      const_seg_ptr g = w128_ptr();
      static_for<0, WFFT_SCALAR_GROUP>([&](auto ic) WFFT_INLINE {
        constexpr int r = decltype(gc)::value * WFFT_SCALAR_GROUP + decltype(ic)::value;
        const double c = g[2 * r], s = g[2 * r + 1];
        im[r] = re[r] * s;
        re[r] = re[r] * c;
      });
    });
  } else {
    wave_lds_fence();
    pipelined_pairs32<WFFT_TWIST_GROUP, cplx>(
        [&](auto ic) WFFT_INLINE { return tq[decltype(ic)::value * 32 + q]; },
        [&](auto i0, const cplx& w0, auto i1, const cplx& w1) WFFT_INLINE {
          constexpr int r0 = decltype(i0)::value, r1 = decltype(i1)::value;
          im[r0] = re[r0] * w0.y;
          im[r1] = re[r1] * w1.y;
          re[r0] = re[r0] * w0.x;
          re[r1] = re[r1] * w1.x;
        });
  }
  fft1024_fwd(re, im, tile, tE, q);
}

// Spectrum C in the S-layout -> lo(integral_constant<r>, 1024 c[t]) and, when HI,
// hi(integral_constant<r>, 1024 c[t + 1024]) for t = q + 32 r.
template <bool HI, bool SCALAR = (WFFT_SCALAR_TWIST != 0), class Lo, class Hi>
__device__ __forceinline__ void inv1024_twisted_real(double (&re)[32], double (&im)[32], char* tile,
                                                     const cplx* tE, const cplx* tq, int q, Lo&& lo,
                                                     Hi&& hi) {
  if constexpr (SCALAR) {
    // tq is the inverse's inter-pass table here (conjugated lane part of the twist folded in)
    (void)tE;
    fft1024_inv(re, im, tile, tq, q);
    static_for<0, 32 / WFFT_SCALAR_GROUP>([&](auto gc) WFFT_INLINE {
      const_seg_ptr g = w128_ptr();
      static_for<0, WFFT_SCALAR_GROUP>([&](auto ic) WFFT_INLINE {
        constexpr int r = decltype(gc)::value * WFFT_SCALAR_GROUP + decltype(ic)::value;
        const std::integral_constant<int, r> rc;
        const double c = g[2 * r], s = g[2 * r + 1];
        const double l = re[r] * c + im[r] * s;          // y conj(w128^r)
        if constexpr (HI) hi(rc, re[r] * s - im[r] * c);
        lo(rc, l);
      });
    });
  } else {
    fft1024_inv(re, im, tile, tE, q);
    wave_lds_fence();
    pipelined_pairs32<WFFT_TWIST_GROUP, cplx>(
        [&](auto ic) WFFT_INLINE { return tq[decltype(ic)::value * 32 + q]; },
        [&](auto i0, const cplx& w0, auto i1, const cplx& w1) WFFT_INLINE {
          constexpr int r0 = decltype(i0)::value, r1 = decltype(i1)::value;
          // y conj(w) = (yr c + yi s) + i (yi c - yr s)
          const double l0 = re[r0] * w0.x + im[r0] * w0.y;
          const double l1 = re[r1] * w1.x + im[r1] * w1.y;
          if constexpr (HI) {
            const double h0 = re[r0] * w0.y - im[r0] * w0.x;
            const double h1 = re[r1] * w1.y - im[r1] * w1.x;
            hi(i0, h0);
            hi(i1, h1);
          }
          lo(i0, l0);
          lo(i1, l1);
        });
  }
}

// Explicit parking of long-lived values in the accumulation registers (AGPRs).
// gfx950 has 512 registers per lane at one wave per SIMD, but vector ALU
// instructions address only the first 256; the other 256 are reached with
// v_accvgpr_write/read.  Values that are idle across a whole FFT (the second
// spectrum of a pair, the accumulator) are parked there by hand so that the
// compiler's allocator only has to fit the live FFT working set.
// (Plain asm, not volatile: these are pure moves, and volatile asm in the
// function keeps the FFT register arrays from being promoted out of scratch.)
__device__ __forceinline__ unsigned agpr_put(unsigned v) {
  unsigned a;
  asm("v_accvgpr_write_b32 %0, %1" : "=a"(a) : "v"(v));
  return a;
}
__device__ __forceinline__ unsigned agpr_get(unsigned a) {
  unsigned v;
  asm("v_accvgpr_read_b32 %0, %1" : "=v"(v) : "a"(a));
  return v;
}
__device__ __forceinline__ void agpr_put_f64(double v, unsigned& lo, unsigned& hi) {
  const unsigned long long u = (unsigned long long)__double_as_longlong(v);
  lo = agpr_put((unsigned)u);
  hi = agpr_put((unsigned)(u >> 32));
}
__device__ __forceinline__ double agpr_get_f64(unsigned lo, unsigned hi) {
  const unsigned long long u = ((unsigned long long)agpr_get(hi) << 32) | agpr_get(lo);
  return __longlong_as_double((long long)u);
}

// Rust `f64::round() as i32`: round half away from zero, saturate, NaN -> 0.
// v_cvt_i32_f64 truncates, saturates out-of-range values and maps NaN to 0,
// which is exactly the cast (a C++ cast would be UB out of range).
// The rounding is v_rndne_f64 (ties to even) where Rust rounds ties away from
// zero.  The two differ only when the fractional part is exactly 0.5; v is an
// exactly-integer convolution value plus FFT rounding error, so a tie means an
// error of 0.5, the point where no f64 FFT (the reference's included) pins the
// result any more.
__device__ __forceinline__ int32_t round_sat_i32(double v) {
  const double r = rint(v);
  int32_t i;
  asm("v_cvt_i32_f64 %0, %1" : "=v"(i) : "v"(r));
  return i;
}

}  // namespace wfft
