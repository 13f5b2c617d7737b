// wave_fft64.hpp — f64 FFT machinery in which one WAVEFRONT (64 lanes x 16 complex registers)
// owns one 1024-point transform, so that a wavefront needs half the registers of the
// half-wave form in wave_fft.hpp and two wavefronts fit on a SIMD.
//
// Element n = lane + 64 r of a sequence lives in lane `lane`, register r.  With
// n = 64 r + 16 a + ql (a = lane >> 4, ql = lane & 15) and k = kr + 16 c + 64 kq,
//   w1024^(nk) = w16^(r kr) . w1024^(lane kr) . w4^(a c) . w64^(ql c) . w16^(ql kq),
// so the 1024-point DFT is
//   1. a 16-point DFT over the registers                        r  -> kr
//   2. the twiddle T1[kr][lane] = w1024^(lane kr)               (table in LDS)
//   3. lane bits (5, 4) <-> register bits (kr3, kr2)            (v_permlane32_swap, v_permlane16_swap:
//                                                                no LDS traffic, no arithmetic)
//   4. a 4-point DFT over the registers that now hold a         a  -> c
//   5. the twiddle T2 = w64^(ql c)                              (three factors per lane)
//   6. a 16 x 16 transpose inside every row of 16 lanes         (through LDS): registers <-> ql
//   7. a 16-point DFT over the registers                        ql -> kq
// and the inverse is the mirror image with conjugated factors.  Afterwards lane
// (kr3 kr2 | c1 c0 kr1 kr0), register p holds X[kr + 16 c + 64 br4(p)] ("spectrum layout"); the
// forward transform writes it and the inverse reads it, nothing else depends on it.
// tests/wave64_model.py is the numpy model of exactly this data movement.
//
// Register-index conventions: a LOGICAL index x of a 16-point pass (kr, or (c, kr1 kr0)) lives
// in physical register br4(x); the registers of a pass's natural side (r, ql) are in order.
#pragma once

#include "wave_fft.hpp"

// Timing experiments (never the shipped build): WFFT64_ABLATE_SWAP, WFFT64_ABLATE_TRANSPOSE and
// WFFT64_ABLATE_T1 (with -DWFFT64_FUSED_T1=0) here, TFHE_ABLATE_KEY and TFHE_ABLATE_PM_STORE in the kernel unit, each compile
// one kind of data movement out; the results are wrong by construction, only the launch time is
// of interest (tools/sweep_variants.py builds such variants side by side).
namespace wfft64 {

using wfft::cplx;
using wfft::f64x2;
using wfft::static_for;
using wfft::wave_lds_fence;

constexpr int kRowStride = 144;                   // 16 f64 of a tile row + 16 bytes of padding
constexpr int kTileBytes = 4 * 16 * kRowStride;   // one f64 per (lane, register) of a wavefront: 9216
constexpr int kT1Entries = 16 * 64;               // T1[kr][lane]
constexpr int kT2Entries = 3 * 16;                // T2[c - 1][ql]
constexpr int kTableBytes = (kT1Entries + kT2Entries) * (int)sizeof(cplx);

// (cos, -sin)(2 pi r / 64), r = 0..15: the register part w4096^(64 r) of the twist
__constant__ double g_w64[32];

__device__ __forceinline__ wfft::const_seg_ptr w64_ptr() {
  wfft::const_seg_ptr p = (wfft::const_seg_ptr)g_w64;
  asm volatile("" : "+s"(p));
  return p;
}

// For butterfly b = 0..7 of the inverse transform's last radix-2 stage (a = y[b], c = y[b + 8],
// W = w16^-b) with the register part of the untwist folded in: alpha = w64^-b, beta = W alpha,
// alpha' = w64^-(b + 8), beta' = W alpha', as (re, im) pairs in this order.
__constant__ double g_inv64[64];

__device__ __forceinline__ wfft::const_seg_ptr inv64_ptr() {
  wfft::const_seg_ptr p = (wfft::const_seg_ptr)g_inv64;
  asm volatile("" : "+s"(p));
  return p;
}

__host__ __device__ constexpr int br4(int v) {
  return ((v & 1) << 3) | ((v & 2) << 1) | ((v & 4) >> 1) | ((v & 8) >> 3);
}

// Butterflies of one radix-2 stage of a 16-point pass: b = 0..7 (see wfft::stage_tj / stage_i0).
template <int HALF>
__host__ __device__ constexpr int nontrivial_count() {
  int c = 0;
  for (int b = 0; b < 8; b++) c += (wfft::stage_tj<HALF>(b) != 0 && wfft::stage_tj<HALF>(b) != 8);
  return c;
}
template <int HALF>
__host__ __device__ constexpr int nontrivial_butterfly(int k) {
  int c = 0;
  for (int b = 0; b < 8; b++)
    if (wfft::stage_tj<HALF>(b) != 0 && wfft::stage_tj<HALF>(b) != 8) {
      if (c == k) return b;
      c++;
    }
  return -1;
}

// Decimation in time, bit-reversed in -> natural out; with VIEW_BR on the bit-reversed register
// view (natural in -> output k in register br4(k)), as wfft::dit_stage.
template <int HALF, bool INV, bool VIEW_BR>
__device__ __forceinline__ void dit_stage(double (&re)[16], double (&im)[16]) {
  static_for<0, 8>([&](auto ic) WFFT_INLINE {
    constexpr int b = decltype(ic)::value;
    constexpr int TJ = wfft::stage_tj<HALF>(b);
    if constexpr (TJ == 0 || TJ == 8) {
      constexpr int p0 = wfft::stage_i0<HALF>(b), p1 = p0 + HALF;
      constexpr int i0 = VIEW_BR ? br4(p0) : p0, i1 = VIEW_BR ? br4(p1) : p1;
      double tr, ti;
      wfft::mul_w32<TJ, INV>(re[i1], im[i1], tr, ti);
      const double ar = re[i0], ai = im[i0];
      re[i0] = ar + tr;
      im[i0] = ai + ti;
      re[i1] = ar - tr;
      im[i1] = ai - ti;
    }
  });
  static_for<0, nontrivial_count<HALF>() / 2>([&](auto kc) WFFT_INLINE {
    constexpr int b0 = nontrivial_butterfly<HALF>(2 * decltype(kc)::value);
    constexpr int b1 = nontrivial_butterfly<HALF>(2 * decltype(kc)::value + 1);
    constexpr int q0 = wfft::stage_i0<HALF>(b0), q1 = wfft::stage_i0<HALF>(b1);
    constexpr int x0 = VIEW_BR ? br4(q0) : q0, y0 = VIEW_BR ? br4(q0 + HALF) : q0 + HALF;
    constexpr int x1 = VIEW_BR ? br4(q1) : q1, y1 = VIEW_BR ? br4(q1 + HALF) : q1 + HALF;
    constexpr int T0 = wfft::stage_tj<HALF>(b0), T1 = wfft::stage_tj<HALF>(b1);
    double n0r, n0i, n1r, n1i;
    wfft::dit_butterfly2(re[x0], im[x0], re[y0], im[y0], wfft::cos32(T0),
                         INV ? wfft::sin32(T0) : -wfft::sin32(T0), n0r, n0i, re[x1], im[x1], re[y1], im[y1],
                         wfft::cos32(T1), INV ? wfft::sin32(T1) : -wfft::sin32(T1), n1r, n1i);
    re[x0] = n0r; im[x0] = n0i; re[x1] = n1r; im[x1] = n1i;
  });
}

// Forward 16-point pass: natural in -> output k in register br4(k).
__device__ __forceinline__ void fwd16(double (&re)[16], double (&im)[16]) {
  dit_stage<1, false, true>(re, im);
  dit_stage<2, false, true>(re, im);
  dit_stage<4, false, true>(re, im);
  dit_stage<8, false, true>(re, im);
}
// Inverse (unnormalised) 16-point pass: input k in register br4(k) -> natural out.
__device__ __forceinline__ void inv16(double (&re)[16], double (&im)[16]) {
  dit_stage<1, true, false>(re, im);
  dit_stage<2, true, false>(re, im);
  dit_stage<4, true, false>(re, im);
  dit_stage<8, true, false>(re, im);
}

// The register part w64^r of the twist inside the first stage of the forward 16-point pass
// (whose butterflies pair registers r and r + 8, without twiddles): for REAL inputs x = re[r],
// y = re[r + 8] the outputs x w64^r +- y w64^(r+8) cost six instructions per butterfly instead of
// four products and four additions.
__device__ __forceinline__ void fwd16_first_stage_twisted_real(double (&re)[16], double (&im)[16]) {
  static_for<0, 4>([&](auto gc) WFFT_INLINE {
    wfft::const_seg_ptr g = w64_ptr();
    static_for<0, 2>([&](auto ic) WFFT_INLINE {
      constexpr int r = decltype(gc)::value * 2 + decltype(ic)::value;
      const double c0 = g[2 * r], s0 = g[2 * r + 1], c1 = g[2 * r + 16], s1 = g[2 * r + 17];
      const double x = re[r], y = re[r + 8];
      const double pr = x * c0, pi = x * s0;
      re[r] = __builtin_fma(y, c1, pr);
      im[r] = __builtin_fma(y, s1, pi);
      re[r + 8] = __builtin_fma(-y, c1, pr);
      im[r + 8] = __builtin_fma(-y, s1, pi);
    });
  });
}
// The last stage of the inverse 16-point pass with the register part of the untwist, when only
// the real part of y conj(w64^r) is wanted: Re((a +- W c) alpha) in four instructions per output
// instead of a butterfly (six, or four without a twiddle) and two per output.
template <class Lo>
__device__ __forceinline__ void inv16_last_stage_untwisted_real(double (&re)[16], double (&im)[16], Lo&& lo) {
  static_for<0, 4>([&](auto gc) WFFT_INLINE {
    wfft::const_seg_ptr g = inv64_ptr();
    static_for<0, 2>([&](auto ic) WFFT_INLINE {
      constexpr int b = decltype(gc)::value * 2 + decltype(ic)::value;
      const double ar = re[b], ai = im[b], cr = re[b + 8], ci = im[b + 8];
      const double x0 = __builtin_fma(-ci, g[8 * b + 3], __builtin_fma(cr, g[8 * b + 2],
                            __builtin_fma(-ai, g[8 * b + 1], ar * g[8 * b + 0])));
      const double x1 = __builtin_fma(ci, g[8 * b + 7], __builtin_fma(-cr, g[8 * b + 6],
                            __builtin_fma(-ai, g[8 * b + 5], ar * g[8 * b + 4])));
      lo(std::integral_constant<int, b>{}, x0);
      lo(std::integral_constant<int, b + 8>{}, x1);
    });
  });
}

// Software-pipelined pass over [0, 16) in groups of G (see wfft::Pipe32).
template <int G, int D, class T>
struct Pipe16 {
  T buf[D + 1][G];
  template <class LoadF>
  __device__ __forceinline__ void prologue(LoadF&& load) {
    static_for<0, D * G>([&](auto ic) WFFT_INLINE {
      constexpr int i = decltype(ic)::value;
      buf[i / G][i % G] = load(ic);
    });
  }
  template <class LoadF, class Use2F>
  __device__ __forceinline__ void run2(LoadF&& load, Use2F&& use2) {
    static_assert(G % 2 == 0 && 16 % G == 0, "pairs");
    static_for<0, 16 / G>([&](auto gc) WFFT_INLINE {
      constexpr int g = decltype(gc)::value;
      if constexpr (g + D < 16 / G) {
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
      __builtin_amdgcn_sched_barrier(0);
    });
  }
};

// Default size of the groups in which the T1 twiddles are fetched (template parameter TWG of the
// transforms): two at a time, because a group of four costs the blind rotation the registers that
// keep its decomposed sample out of scratch memory; the isolated product has room for four.
#ifndef WFFT64_TW_GROUP
#define WFFT64_TW_GROUP 2
#endif

// ---- step 3: lane bits <-> register bits ---------------------------------------------------
// v_permlane32_swap_b32 a, b swaps lanes 32..63 of a with lanes 0..31 of b; v_permlane16_swap_b32
// swaps the odd rows (of 16 lanes) of a with the even rows of b.  Applied to the register pair
// that differs in one index bit, either exchanges that register-index bit with a lane bit.
template <bool ROW16>
__device__ __forceinline__ void swap_f64(double& a, double& b) {
  unsigned alo = (unsigned)__double2loint(a), ahi = (unsigned)__double2hiint(a);
  unsigned blo = (unsigned)__double2loint(b), bhi = (unsigned)__double2hiint(b);
  if constexpr (ROW16) {
    auto l = __builtin_amdgcn_permlane16_swap(alo, blo, false, false);
    auto h = __builtin_amdgcn_permlane16_swap(ahi, bhi, false, false);
    a = __hiloint2double((int)h[0], (int)l[0]);
    b = __hiloint2double((int)h[1], (int)l[1]);
  } else {
    auto l = __builtin_amdgcn_permlane32_swap(alo, blo, false, false);
    auto h = __builtin_amdgcn_permlane32_swap(ahi, bhi, false, false);
    a = __hiloint2double((int)h[0], (int)l[0]);
    b = __hiloint2double((int)h[1], (int)l[1]);
  }
}
// logical register bit 3 <-> lane bit 5, logical bit 2 <-> lane bit 4 (its own inverse)
__device__ __forceinline__ void exchange_lane_bits(double (&re)[16], double (&im)[16]) {
  static_for<0, 8>([&](auto ic) WFFT_INLINE {
    constexpr int x = decltype(ic)::value;                 // logical index with bit 3 clear
    swap_f64<false>(re[br4(x)], re[br4(x | 8)]);
    swap_f64<false>(im[br4(x)], im[br4(x | 8)]);
  });
  static_for<0, 8>([&](auto ic) WFFT_INLINE {
    constexpr int e = decltype(ic)::value;
    constexpr int x = (e & 3) | ((e & 4) << 1);            // logical index with bit 2 clear
    swap_f64<true>(re[br4(x)], re[br4(x | 4)]);
    swap_f64<true>(im[br4(x)], im[br4(x | 4)]);
  });
}

// The same exchange through the wavefront's tile: every lane stores its 16 registers and loads
// the 16 that belong to it afterwards (one 8-byte access per register each way, real and
// imaginary parts one after the other).  A v_permlane*_swap costs a SIMD about 3 ns beside a
// stream of v_fma_f64 from a second wavefront, a ds_write_b64 2 ns and an 8-byte read under
// 1 ns, and a swap moves two registers where the LDS pair moves one each way: 64 swaps against
// 32 + 32 LDS accesses per transform.
// Tile layout (f64 units): [xhi][a][xlo][ql] with kExchStride per xhi, where the writer is lane
// (a | ql) and (xhi, xlo) its logical register index; lane (A | ql) then reads, for its new
// logical register (yhi, ylo), the entry [A][yhi][ylo][ql].  The 16 f64 of padding per xhi put
// the 8-byte reads of lanes (A, ql) and (A + 1, ql) on different halves of the 64 banks.
#ifndef WFFT64_LDS_EXCHANGE
#define WFFT64_LDS_EXCHANGE 0
#endif
constexpr int kExchStride = 4 * 4 * 16 + 16;
static_assert(4 * kExchStride * 8 <= kTileBytes, "the exchange fits the transpose tile");
struct ExchAddr {
  char* w;          // + ((x >> 2) * kExchStride + (x & 3) * 16) * 8: where this lane stores logical register x
  const char* r;    // + ((y >> 2) * 64 + (y & 3) * 16) * 8: where its new logical register y waits
};
__device__ __forceinline__ ExchAddr exch_addr(char* tile, int lane) {
  const int a = lane >> 4, ql = lane & 15;
  return {tile + (a * 64 + ql) * 8, tile + (a * kExchStride + ql) * 8};
}
template <int BYTE_OFFSET>
__device__ __forceinline__ void exch_store(const ExchAddr& t, double x) {
  typedef __attribute__((address_space(3))) char* lds_ptr;
  const unsigned a = (unsigned)(uintptr_t)(lds_ptr)t.w;
  asm volatile("ds_write_b64 %0, %1 offset:%2" : : "v"(a), "v"(x), "n"(BYTE_OFFSET) : "memory");
}
__device__ __forceinline__ void exchange_component_lds(double (&v)[16], const ExchAddr& t) {
  static_for<0, 16>([&](auto ic) WFFT_INLINE {
    constexpr int x = decltype(ic)::value;
    exch_store<((x >> 2) * kExchStride + (x & 3) * 16) * 8>(t, v[br4(x)]);
  });
  wave_lds_fence();
  static_for<0, 16>([&](auto ic) WFFT_INLINE {
    constexpr int y = decltype(ic)::value;
    v[br4(y)] = *reinterpret_cast<const double*>(t.r + ((y >> 2) * 64 + (y & 3) * 16) * 8);
  });
  wave_lds_fence();
}
__device__ __forceinline__ void exchange_lane_bits(double (&re)[16], double (&im)[16], char* tile, int lane) {
#ifdef WFFT64_ABLATE_SWAP
  (void)tile; (void)lane; (void)re; (void)im;
#elif WFFT64_LDS_EXCHANGE
  const ExchAddr t = exch_addr(tile, lane);
  wave_lds_fence();
  exchange_component_lds(re, t);
  exchange_component_lds(im, t);
#else
  (void)tile;
  (void)lane;
  exchange_lane_bits(re, im);
#endif
}

// ---- step 4: 4-point DFT over logical bits (3, 2), in place ---------------------------------
template <bool INV>
__device__ __forceinline__ void dft4_high_bits(double (&re)[16], double (&im)[16]) {
  static_for<0, 4>([&](auto jc) WFFT_INLINE {
    constexpr int j = decltype(jc)::value;
    constexpr int p0 = br4(j), p1 = br4(4 + j), p2 = br4(8 + j), p3 = br4(12 + j);
    const double sr = re[p0] + re[p2], si = im[p0] + im[p2];       // x0 + x2
    const double dr = re[p0] - re[p2], di = im[p0] - im[p2];       // x0 - x2
    const double tr = re[p1] + re[p3], ti = im[p1] + im[p3];       // x1 + x3
    const double ur = re[p1] - re[p3], ui = im[p1] - im[p3];       // x1 - x3
    re[p0] = sr + tr; im[p0] = si + ti;
    re[p2] = sr - tr; im[p2] = si - ti;
    if constexpr (!INV) {                                          // X1 = d - i u, X3 = d + i u
      re[p1] = dr + ui; im[p1] = di - ur;
      re[p3] = dr - ui; im[p3] = di + ur;
    } else {
      re[p1] = dr - ui; im[p1] = di + ur;
      re[p3] = dr + ui; im[p3] = di - ur;
    }
  });
}

// The inverse 4-point pass with the conjugated T2 twiddles on its inputs, x[c] conj(w)^c with
// w = w1 (w2 = w1^2), as four decimation-in-time butterflies a +- v b of six FMAs each:
//   s, d = x0 +- conj(w2) x2      t, u = x1 +- conj(w2) x3
//   X0, X2 = s +- conj(w1) t      X1, X3 = d +- i conj(w1) u
// 24 instructions for four points instead of three complex products and sixteen additions (28).
// v = (vr, vi) comes in vector registers; the new a returns in fresh registers, b in place.
__device__ __forceinline__ void dit_butterfly2_v(double a0r, double a0i, double& b0r, double& b0i, double vr0,
                                                 double vi0, double& n0r, double& n0i, double a1r, double a1i,
                                                 double& b1r, double& b1i, double vr1, double vi1, double& n1r,
                                                 double& n1i) {
  asm("v_fma_f64 %0, -%5, %13, %8\n\tv_fma_f64 %1, %5, %12, %9\n\t"
      "v_fma_f64 %2, -%7, %15, %10\n\tv_fma_f64 %3, %7, %14, %11\n\t"
      "v_fma_f64 %0, %4, %12, %0\n\tv_fma_f64 %1, %4, %13, %1\n\t"
      "v_fma_f64 %2, %6, %14, %2\n\tv_fma_f64 %3, %6, %15, %3\n\t"
      "v_fma_f64 %4, 2.0, %8, -%0\n\tv_fma_f64 %5, 2.0, %9, -%1\n\t"
      "v_fma_f64 %6, 2.0, %10, -%2\n\tv_fma_f64 %7, 2.0, %11, -%3"
      : "=&v"(n0r), "=&v"(n0i), "=&v"(n1r), "=&v"(n1i), "+v"(b0r), "+v"(b0i), "+v"(b1r), "+v"(b1i)
      : "v"(a0r), "v"(a0i), "v"(a1r), "v"(a1i), "v"(vr0), "v"(vi0), "v"(vr1), "v"(vi1));
}
// w1 = w64^ql, w2 = w64^(2 ql) as stored in T2 (not conjugated)
__device__ __forceinline__ void inv_dft4_with_t2(double (&re)[16], double (&im)[16], const cplx& w1, const cplx& w2) {
  const double m1i = -w1.y, m2i = -w2.y;                        // conj(w) = (w.x, -w.y)
  static_for<0, 2>([&](auto jc) WFFT_INLINE {
    constexpr int j0 = 2 * decltype(jc)::value, j1 = j0 + 1;    // two groups of four points at a time
    constexpr int a0 = br4(j0), a1 = br4(4 + j0), a2 = br4(8 + j0), a3 = br4(12 + j0);
    constexpr int b0 = br4(j1), b1 = br4(4 + j1), b2 = br4(8 + j1), b3 = br4(12 + j1);
    double sr0, si0, sr1, si1, tr0, ti0, tr1, ti1;
    // s = x0 + conj(w2) x2 (fresh), x2 <- d;   t = x1 + conj(w2) x3 (fresh), x3 <- u
    dit_butterfly2_v(re[a0], im[a0], re[a2], im[a2], w2.x, m2i, sr0, si0, re[b0], im[b0], re[b2], im[b2], w2.x, m2i,
                     sr1, si1);
    dit_butterfly2_v(re[a1], im[a1], re[a3], im[a3], w2.x, m2i, tr0, ti0, re[b1], im[b1], re[b3], im[b3], w2.x, m2i,
                     tr1, ti1);
    double x0r, x0i, x1r, x1i, y0r, y0i, y1r, y1i;
    // X0 = s + conj(w1) t (fresh), t <- X2
    dit_butterfly2_v(sr0, si0, tr0, ti0, w1.x, m1i, x0r, x0i, sr1, si1, tr1, ti1, w1.x, m1i, x1r, x1i);
    // X1 = d + i conj(w1) u (fresh), u <- X3;  i conj(w1) = (w1.y, w1.x)
    dit_butterfly2_v(re[a2], im[a2], re[a3], im[a3], w1.y, w1.x, y0r, y0i, re[b2], im[b2], re[b3], im[b3], w1.y,
                     w1.x, y1r, y1i);
    re[a0] = x0r; im[a0] = x0i; re[a2] = tr0; im[a2] = ti0; re[a1] = y0r; im[a1] = y0i;     // a3 holds X3
    re[b0] = x1r; im[b0] = x1i; re[b2] = tr1; im[b2] = ti1; re[b1] = y1r; im[b1] = y1i;
  });
}

// ---- steps 2 + 4 (forward) and 2 + 7 (inverse): T1 inside the radix-2 stage that follows it ----
// A butterfly whose two inputs both carry a twiddle, a' = wa a + wb b, b' = wa a - wb b, in ten
// instructions (p = wa a: 4, a' = p + wb b: 4, b' = 2 p - a': 2) instead of two complex products and
// four additions (12), in place.  CONJ multiplies by the conjugates.
#ifndef WFFT64_FUSED_T1
#define WFFT64_FUSED_T1 1
#endif
template <bool CONJ>
__device__ __forceinline__ void butterfly_ww(double& ar, double& ai, double& br, double& bi, const cplx& wa,
                                             const cplx& wb) {
  double t0, t1;
  if constexpr (!CONJ)
    asm("v_mul_f64 %4, %1, %7\n\tv_mul_f64 %5, %1, %6\n\t"
        "v_fma_f64 %4, %0, %6, -%4\n\tv_fma_f64 %5, %0, %7, %5\n\t"
        "v_fma_f64 %0, -%3, %9, %4\n\tv_fma_f64 %1, %3, %8, %5\n\t"
        "v_fma_f64 %0, %2, %8, %0\n\tv_fma_f64 %1, %2, %9, %1\n\t"
        "v_fma_f64 %2, 2.0, %4, -%0\n\tv_fma_f64 %3, 2.0, %5, -%1"
        : "+v"(ar), "+v"(ai), "+v"(br), "+v"(bi), "=&v"(t0), "=&v"(t1)
        : "v"(wa.x), "v"(wa.y), "v"(wb.x), "v"(wb.y));
  else
    asm("v_mul_f64 %4, %1, %7\n\tv_mul_f64 %5, %1, %6\n\t"
        "v_fma_f64 %4, %0, %6, %4\n\tv_fma_f64 %5, -%0, %7, %5\n\t"
        "v_fma_f64 %0, %3, %9, %4\n\tv_fma_f64 %1, %3, %8, %5\n\t"
        "v_fma_f64 %0, %2, %8, %0\n\tv_fma_f64 %1, -%2, %9, %1\n\t"
        "v_fma_f64 %2, 2.0, %4, -%0\n\tv_fma_f64 %3, 2.0, %5, -%1"
        : "+v"(ar), "+v"(ai), "+v"(br), "+v"(bi), "=&v"(t0), "=&v"(t1)
        : "v"(wa.x), "v"(wa.y), "v"(wb.x), "v"(wb.y));
}
// The same butterfly (not conjugated) with a and b left alone and the results in registers of
// their own: na = wa a + wb b, nb = wa a - wb b.
__device__ __forceinline__ void butterfly_ww_to(double ar, double ai, double br, double bi, const cplx& wa,
                                                const cplx& wb, double& nar, double& nai, double& nbr, double& nbi) {
  asm("v_mul_f64 %2, %5, %9\n\tv_mul_f64 %3, %5, %8\n\t"
      "v_fma_f64 %2, %4, %8, -%2\n\tv_fma_f64 %3, %4, %9, %3\n\t"
      "v_fma_f64 %0, -%7, %11, %2\n\tv_fma_f64 %1, %7, %10, %3\n\t"
      "v_fma_f64 %0, %6, %10, %0\n\tv_fma_f64 %1, %6, %11, %1\n\t"
      "v_fma_f64 %2, 2.0, %2, -%0\n\tv_fma_f64 %3, 2.0, %3, -%1"
      : "=&v"(nar), "=&v"(nai), "=&v"(nbr), "=&v"(nbi)
      : "v"(ar), "v"(ai), "v"(br), "v"(bi), "v"(wa.x), "v"(wa.y), "v"(wb.x), "v"(wb.y));
}
// Second half of the 4-point pass over logical bits (3, 2), its first half (s, d = x0 +- x2 in the
// registers of x0, x2; t, u = x1 +- x3 in those of x1, x3) done by butterfly_ww.
template <bool INV>
__device__ __forceinline__ void dft4_high_bits_second_stage(double (&re)[16], double (&im)[16]) {
  static_for<0, 4>([&](auto jc) WFFT_INLINE {
    constexpr int j = decltype(jc)::value;
    constexpr int p0 = br4(j), p1 = br4(4 + j), p2 = br4(8 + j), p3 = br4(12 + j);
    const double sr = re[p0], si = im[p0], dr = re[p2], di = im[p2];
    const double tr = re[p1], ti = im[p1], ur = re[p3], ui = im[p3];
    re[p0] = sr + tr; im[p0] = si + ti;
    re[p2] = sr - tr; im[p2] = si - ti;
    if constexpr (!INV) {                                          // X1 = d - i u, X3 = d + i u
      re[p1] = dr + ui; im[p1] = di - ur;
      re[p3] = dr - ui; im[p3] = di + ur;
    } else {
      re[p1] = dr - ui; im[p1] = di + ur;
      re[p3] = dr + ui; im[p3] = di - ur;
    }
  });
}

// ---- step 6: 16 x 16 transpose inside every row of 16 lanes ----------------------------------
// Row R = lane >> 4 owns 16 tile rows of kRowStride bytes; lane (R, x) stores the f64 of its
// register with index i at [R][i][x] and then loads row [R][x][0..15].  The padding of 16 bytes
// per row keeps the 8-byte stores of a row and the 16-byte loads of eight lanes on distinct
// banks, with plain immediate offsets (no address arithmetic per access).
struct TileAddr {
  char* w;          // + i * kRowStride: where this lane stores the register with index i
  const char* r;    // + 16 g: granule g (elements 2 g, 2 g + 1) of this lane's row
};
// One 8-byte store per register.  Written as an instruction of its own (1): the compiler would
// otherwise fuse two rows into ds_write2_b64, which costs MI355X's LDS more than two ds_write_b64.
#ifndef WFFT64_ASM_STORE
#define WFFT64_ASM_STORE 1
#endif
template <int ROW>
__device__ __forceinline__ void tile_store(const TileAddr& t, double x) {
#ifdef WFFT64_ABLATE_TRANSPOSE
  (void)t; (void)x;
  return;
#endif
#if WFFT64_ASM_STORE
  typedef __attribute__((address_space(3))) char* lds_ptr;
  const unsigned a = (unsigned)(uintptr_t)(lds_ptr)t.w;
  asm volatile("ds_write_b64 %0, %1 offset:%2" : : "v"(a), "v"(x), "n"(ROW * kRowStride) : "memory");
#else
  *reinterpret_cast<double*>(t.w + ROW * kRowStride) = x;
#endif
}
__device__ __forceinline__ TileAddr tile_addr(char* tile, int lane) {
  char* row0 = tile + (lane >> 4) * (16 * kRowStride);
  return {row0 + (lane & 15) * 8, row0 + (lane & 15) * kRowStride};
}

// The real, then the imaginary parts go through the tile (a wavefront's LDS operations execute
// in program order, so no wait separates the trips).  STORE_BR: the register with index i is
// physical register br4(i) (else i); LOAD_BR likewise for the registers that receive the row.
template <bool STORE_BR, bool LOAD_BR>
__device__ __forceinline__ void row_transpose_load(double (&v)[16], const TileAddr& t) {
#ifdef WFFT64_ABLATE_TRANSPOSE
  (void)t; (void)v;
  return;
#endif
  wave_lds_fence();
  static_for<0, 8>([&](auto gc) WFFT_INLINE {
    constexpr int g = decltype(gc)::value;
    const f64x2 w = *reinterpret_cast<const f64x2*>(t.r + 16 * g);
    v[LOAD_BR ? br4(2 * g) : 2 * g] = w.a;
    v[LOAD_BR ? br4(2 * g + 1) : 2 * g + 1] = w.b;
  });
  wave_lds_fence();
}
template <bool STORE_BR, bool LOAD_BR>
__device__ __forceinline__ void row_transpose(double (&v)[16], const TileAddr& t) {
  static_for<0, 16>([&](auto ic) WFFT_INLINE {
    constexpr int i = decltype(ic)::value;
    tile_store<i>(t, v[STORE_BR ? br4(i) : i]);
  });
  row_transpose_load<STORE_BR, LOAD_BR>(v, t);
}

// Tables in LDS: T1 [16][64] then T2 [3][16] (make_tables64 on the host).
struct Tables {
  const cplx* t1;
  const cplx* t2;
};

// ---- the twisted transforms of a REAL polynomial (see wave_fft.hpp for the mathematics) ------
// x[lane + 64 r] in re[r] -> spectrum layout.  The register part w64^r of the twist w4096^n comes
// from the constant segment (scalar loads), the lane part w4096^lane is folded into T1.
// FUSE: T1 inside the radix-2 stage next to it (butterfly_ww) and, in the inverse transform, the
// untwist inside the last stage; fewer instructions, longer dependency chains and a few more live
// registers: it pays in the throughput kernels (+2.6 % gates/s on MI355X) and in the kernel that
// runs one ciphertext per workgroup (-2.5 % latency), not in the textbook kernel (registers).
template <int TWG = WFFT64_TW_GROUP, bool FUSE = (WFFT64_FUSED_T1 != 0)>
__device__ __forceinline__ void fwd1024_twisted_real(double (&re)[16], double (&im)[16], char* tile,
                                                     const Tables& tb, int lane) {
  fwd16_first_stage_twisted_real(re, im);
  dit_stage<2, false, true>(re, im);
  dit_stage<4, false, true>(re, im);
  dit_stage<8, false, true>(re, im);
  if constexpr (FUSE) {
    // T1 after the exchange, inside the first half of the 4-point pass: the element now in lane
    // (a | ql), logical register y came from lane (y3 y2 | ql), logical register (a1 a0 y1 y0) = kr.
    // Registers 2 m and 2 m + 1 hold the logical pair (y, y | 8), i.e. (x0, x2) or (x1, x3).
    exchange_lane_bits(re, im, tile, lane);
    wave_lds_fence();
    Pipe16<TWG, 1, cplx> pipe;
    const cplx* t1x = tb.t1 + ((lane >> 4) << 8) + (lane & 15);
    auto load = [&](auto ic) WFFT_INLINE {
      constexpr int y = br4(decltype(ic)::value);
      return t1x[(y & 3) * 64 + (y >> 2) * 16];
    };
    pipe.prologue(load);
    pipe.run2(load, [&](auto i0, const cplx& w0, auto i1, const cplx& w1) WFFT_INLINE {
      constexpr int p0 = decltype(i0)::value, p1 = decltype(i1)::value;
      butterfly_ww<false>(re[p0], im[p0], re[p1], im[p1], w0, w1);
    });
    dft4_high_bits_second_stage<false>(re, im);
  } else {
    wave_lds_fence();
    Pipe16<TWG, 1, cplx> pipe;
#ifdef WFFT64_ABLATE_T1
    auto load = [&](auto ic) WFFT_INLINE { return cplx{0.75 + decltype(ic)::value, 0.5}; };
#else
    auto load = [&](auto ic) WFFT_INLINE { return tb.t1[br4(decltype(ic)::value) * 64 + lane]; };
#endif
    pipe.prologue(load);
    pipe.run2(load, [&](auto i0, const cplx& w0, auto i1, const cplx& w1) WFFT_INLINE {
      constexpr int p0 = decltype(i0)::value, p1 = decltype(i1)::value;      // physical: kr = br4(p)
      wfft::cmul2<false>(re[p0], im[p0], w0, re[p1], im[p1], w1);
    });
    exchange_lane_bits(re, im, tile, lane);
    dft4_high_bits<false>(re, im);
  }
  const TileAddr ta = tile_addr(tile, lane);
  {
    // T2 = w64^(ql c) on the registers (c, j), c = 1..3; the real parts leave for the tile at once
    const cplx* t2 = tb.t2 + (lane & 15);
    const cplx w1 = t2[0], w2 = t2[16], w3 = t2[32];
    const cplx* wc[3] = {&w1, &w2, &w3};
    static_for<0, 4>([&](auto jc) WFFT_INLINE {
      constexpr int j = decltype(jc)::value;
      tile_store<j>(ta, re[br4(j)]);
    });
    static_for<1, 4>([&](auto cc) WFFT_INLINE {
      constexpr int c = decltype(cc)::value;
      static_for<0, 2>([&](auto hc) WFFT_INLINE {
        constexpr int x0 = 4 * c + 2 * decltype(hc)::value, x1 = x0 + 1;
        wfft::cmul2<false>(re[br4(x0)], im[br4(x0)], *wc[c - 1], re[br4(x1)], im[br4(x1)], *wc[c - 1]);
        tile_store<x0>(ta, re[br4(x0)]);
        tile_store<x1>(ta, re[br4(x1)]);
      });
    });
  }
  row_transpose_load<true, false>(re, ta);
  row_transpose<true, false>(im, ta);
  fwd16(re, im);
}

// Spectrum C (spectrum layout) -> lo(integral_constant<r>, 1024 c[t]) and, when HI,
// hi(integral_constant<r>, 1024 c[t + 1024]) for t = lane + 64 r.
// STAGE1_DONE: the caller has run the first radix-2 stage of the first 16-point pass (registers
// 2 m, 2 m + 1 hold C[2 m] +- C[2 m + 1]), as butterfly_ww does together with a pointwise product
// C = D K.  FUSE as in the forward transform.
// before_tile() runs after the first 16-point pass, before the transform first writes to its tile.
struct NoHook {
  __device__ __forceinline__ void operator()() const {}
};
template <bool HI, int TWG = WFFT64_TW_GROUP, bool STAGE1_DONE = false, bool FUSE = (WFFT64_FUSED_T1 != 0), class Lo,
          class Hi, class BeforeTile = NoHook>
__device__ __forceinline__ void inv1024_twisted_real(double (&re)[16], double (&im)[16], char* tile,
                                                     const Tables& tb, int lane, Lo&& lo, Hi&& hi,
                                                     BeforeTile&& before_tile = BeforeTile{}) {
  if constexpr (!STAGE1_DONE) dit_stage<1, true, false>(re, im);
  dit_stage<2, true, false>(re, im);
  dit_stage<4, true, false>(re, im);
  dit_stage<8, true, false>(re, im);
  before_tile();
  const TileAddr ta = tile_addr(tile, lane);
  row_transpose<false, true>(re, ta);
  row_transpose<false, true>(im, ta);
  {
    const cplx* t2 = tb.t2 + (lane & 15);
    const cplx w1 = t2[0], w2 = t2[16];
    inv_dft4_with_t2(re, im, w1, w2);
  }
  exchange_lane_bits(re, im, tile, lane);
  wave_lds_fence();
  constexpr bool kLastStageUntwists = FUSE && !HI;
  {
    Pipe16<TWG, 1, cplx> pipe;
#ifdef WFFT64_ABLATE_T1
    auto load = [&](auto ic) WFFT_INLINE { return cplx{0.75 + decltype(ic)::value, 0.5}; };
#else
    auto load = [&](auto ic) WFFT_INLINE { return tb.t1[br4(decltype(ic)::value) * 64 + lane]; };
#endif
    pipe.prologue(load);
    // conj(T1), then the 16-point pass whose first stage pairs registers 2 m, 2 m + 1
    pipe.run2(load, [&](auto i0, const cplx& w0, auto i1, const cplx& w1) WFFT_INLINE {
      constexpr int p0 = decltype(i0)::value, p1 = decltype(i1)::value;
      if constexpr (FUSE) butterfly_ww<true>(re[p0], im[p0], re[p1], im[p1], w0, w1);
      else wfft::cmul2<true>(re[p0], im[p0], w0, re[p1], im[p1], w1);
    });
  }
  if constexpr (!FUSE) dit_stage<1, true, false>(re, im);
  dit_stage<2, true, false>(re, im);
  dit_stage<4, true, false>(re, im);
  if constexpr (kLastStageUntwists) {
    inv16_last_stage_untwisted_real(re, im, lo);
  } else {
    dit_stage<8, true, false>(re, im);
    static_for<0, 4>([&](auto gc) WFFT_INLINE {
      wfft::const_seg_ptr g = w64_ptr();
      static_for<0, 4>([&](auto ic) WFFT_INLINE {
        constexpr int r = decltype(gc)::value * 4 + decltype(ic)::value;
        const std::integral_constant<int, r> rc;
        const double c = g[2 * r], s = g[2 * r + 1];
        const double l = re[r] * c + im[r] * s;          // y conj(w64^r)
        if constexpr (HI) hi(rc, re[r] * s - im[r] * c);
        lo(rc, l);
      });
    });
  }
}

}  // namespace wfft64
