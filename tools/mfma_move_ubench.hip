// Can the matrix pipe move data while the vector pipe computes?  A 4 x 4 transposition between
// four f64 registers and the four 16-lane groups of a wavefront (what the transforms do with
// v_permlane32_swap / v_permlane16_swap) written as four v_mfma_f64_16x16x4_f64 with 0/1 selection
// matrices, (1) checked for the lane layout and exactness, (2) timed between a stream of
// independent v_fma_f64 against the swaps it would replace, at one and two wavefronts per SIMD.
//   hipcc --offload-arch=gfx950 -O3 -o mfma_move_ubench tools/mfma_move_ubench.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef double d4 __attribute__((ext_vector_type(4)));

// selection matrix m as the A operand (lane l holds A[i = l % 16][k = l / 16]; register v of
// lane group g of the result holds row i = 4 v + g): A_m[4 v + g][k] = (m == g) && (k == v)
__device__ __forceinline__ double sel(int m, int lane) {
  const int i = lane & 15, k = lane >> 4;
  return ((i & 3) == m && (i >> 2) == k) ? 1.0 : 0.0;
}

__global__ void check(double* out) {
  const int lane = threadIdx.x;
  double x[4];
  for (int r = 0; r < 4; r++) x[r] = 1000.0 * r + lane + 0.125;    // X[r][g][j]
  d4 d = {0, 0, 0, 0};
  for (int m = 0; m < 4; m++) d = __builtin_amdgcn_mfma_f64_16x16x4f64(sel(m, lane), x[m], d, 0, 0, 0);
  for (int v = 0; v < 4; v++) out[v * 64 + lane] = d[v];
}

enum { NONE, SWAPS, MFMAS };
template <int KIND, int FMAS>
__global__ __launch_bounds__(512) void k(unsigned long long* out, double seed, int iters) {
  double x[16];
  unsigned a[16];
  d4 acc[4];
  const int lane = threadIdx.x & 63;
  double s0 = sel(0, lane), s1 = sel(1, lane), s2 = sel(2, lane), s3 = sel(3, lane);
  for (int i = 0; i < 16; i++) { x[i] = seed + i + threadIdx.x; a[i] = i * 7 + threadIdx.x; }
  for (int i = 0; i < 4; i++) acc[i] = d4{seed, 0, 0, 0};
  double w = 1.0000001, z = 0.5;
  const unsigned long long t0 = clock64(), r0 = wall_clock64();
  for (int it = 0; it < iters; it++) {
#pragma unroll
    for (int c = 0; c < 64; c++) {
      if (FMAS) asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(x[c & 15]) : "v"(w), "v"(z));
      if (KIND == SWAPS && (c & 1)) {          // 32 swaps = four groups of 4 f64 registers
        const int e = c >> 1;
        asm volatile("v_permlane32_swap_b32 %0, %1" : "+v"(a[e & 15]), "+v"(a[(e + 8) & 15]));
      }
      if (KIND == MFMAS && (c & 3) == 3) {     // 16 mfma = the same four groups
        const int e = c >> 2, g = e & 3, m = e >> 2;
        const double sm = m == 0 ? s0 : m == 1 ? s1 : m == 2 ? s2 : s3;
        asm volatile("v_mfma_f64_16x16x4_f64 %0, %1, %2, %0" : "+v"(acc[g]) : "v"(sm), "v"(x[(4 * g + m + 8) & 15]));
      }
    }
  }
  const unsigned long long t1 = clock64(), r1 = wall_clock64();
  if (threadIdx.x == 32 && blockIdx.x == 17) { out[0] = t1 - t0; out[2] = r1 - r0; }
  double s = 0;
  for (int i = 0; i < 16; i++) s += x[i] + a[i];
  for (int i = 0; i < 4; i++) s += acc[i].x + acc[i].w;
  if (s == 12345.678) out[1] = 1;
}

template <int KIND, int FMAS>
void run(const char* name, unsigned long long* d, int threads) {
  const int iters = 40000, blocks = 256;
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  k<KIND, FMAS><<<blocks, threads>>>(d, 1.0, iters);
  hipEventRecord(e0);
  k<KIND, FMAS><<<blocks, threads>>>(d, 1.0, iters);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms;
  hipEventElapsedTime(&ms, e0, e1);
  unsigned long long h[3];
  hipMemcpy(h, d, 24, hipMemcpyDeviceToHost);
  const int wps = threads / 256;
  printf("%-34s %d wave(s)/SIMD: %8.2f ns per group and SIMD, shader clock %.3f GHz\n", name, wps,
         ms * 1e6 / iters / wps, (double)h[0] / (double)h[2] * 0.1);
}

int main() {
  double* o;
  hipMalloc(&o, 256 * 8);
  check<<<1, 64>>>(o);
  double h[256];
  hipMemcpy(h, o, sizeof h, hipMemcpyDeviceToHost);
  int bad = 0;
  for (int v = 0; v < 4; v++)
    for (int l = 0; l < 64; l++) {
      const int g = l >> 4, j = l & 15;
      const double want = 1000.0 * g + (16 * v + j) + 0.125;       // X[g][v][j]
      if (h[v * 64 + l] != want) bad++;
    }
  printf("transposition by four mfma: %s (%d of 256 values differ); lane 17 holds %.3f %.3f %.3f %.3f\n",
         bad ? "LAYOUT DIFFERS" : "exact", bad, h[17], h[64 + 17], h[128 + 17], h[192 + 17]);
  unsigned long long* d;
  hipMalloc(&d, 32);
  for (int threads : {256, 512}) {
    run<NONE, 1>("64 v_fma_f64", d, threads);
    run<SWAPS, 1>("64 fma + 32 v_permlane32_swap", d, threads);
    run<MFMAS, 1>("64 fma + 16 v_mfma_f64_16x16x4", d, threads);
    run<MFMAS, 0>("16 v_mfma_f64_16x16x4 alone", d, threads);
  }
  return 0;
}
