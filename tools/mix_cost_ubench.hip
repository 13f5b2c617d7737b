// Marginal cost of the non-arithmetic instructions of a transform when they are interleaved
// with a stream of independent v_fma_f64, at one, two and four wavefronts per SIMD, all compute
// units busy; and the shader clock the chip actually runs at under each mix (s_memtime ticks
// against the 100 MHz s_memrealtime counter).
//   hipcc --offload-arch=gfx950 -O3 -o mix_ubench tools/mix_cost_ubench.hip
// One group = 64 v_fma_f64 (16 chains, three register sources) plus EXTRA instructions of one
// kind spread evenly between them; the LDS / memory results are waited for at the end of the group.
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f4 __attribute__((ext_vector_type(4)));
typedef unsigned u2 __attribute__((ext_vector_type(2)));

enum { NONE, DS_WRITE_B64, DS_READ_B128, MOV_DPP, CNDMASK_DPP, PERMLANE32_SWAP, V_MOV, GLOBAL_LOAD_X4, ADD_F64, CVT, DS_WRITE_B128,
       FMA_F32, ACCVGPR };

template <int KIND, int EXTRA, int FMAS>
__global__ __launch_bounds__(1024) void k(unsigned long long* out, const f4* __restrict__ g, double seed, int iters) {
  __shared__ __attribute__((aligned(16))) char lds[65536];
  double x[16], y[8];
  unsigned a[16];
  float f[8];
  f4 r[4];
  for (int i = 0; i < 16; i++) { x[i] = seed + i + threadIdx.x; a[i] = i * 7 + threadIdx.x; }
  for (int i = 0; i < 8; i++) { y[i] = seed * i; f[i] = (float)i; }
  for (int i = 0; i < 4; i++) r[i] = f4{0, 0, 0, 0};
  double w = 1.0000001, z = 0.5, st = seed;
  const unsigned addr = threadIdx.x * 16, addr8 = threadIdx.x * 8;
  const f4* gp = g + threadIdx.x;
  for (int i = threadIdx.x; i < 16384; i += blockDim.x) reinterpret_cast<float*>(lds)[i] = 0;
  __syncthreads();
  const unsigned long long t0 = clock64(), r0 = wall_clock64();
  for (int it = 0; it < iters; it++) {
    constexpr int every = EXTRA ? (FMAS ? FMAS / EXTRA : 1) : 1 << 30;
    constexpr int total = FMAS ? FMAS : EXTRA;
#pragma unroll
    for (int c = 0; c < total; c++) {
      if (FMAS) asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(x[c & 15]) : "v"(w), "v"(z));
      if (EXTRA && (c % every) == every - 1 && c / every < EXTRA) {
        const int e = c / every;
        if (KIND == DS_WRITE_B64) asm volatile("ds_write_b64 %0, %1" : : "v"(addr8 + (e & 1) * 8192), "v"(st));
        if (KIND == DS_WRITE_B128) asm volatile("ds_write_b128 %0, %1" : : "v"(addr + (e & 1) * 16384), "v"(r[3]));
        if (KIND == DS_READ_B128) asm volatile("ds_read_b128 %0, %1" : "=v"(r[e & 3]) : "v"(addr + (e & 1) * 16384));
        if (KIND == MOV_DPP) asm volatile("v_mov_b32_dpp %0, %1 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf" : "+v"(a[e & 15]) : "v"(a[(e + 5) & 15]));
        if (KIND == CNDMASK_DPP) asm volatile("v_cndmask_b32_dpp %0, %1, %2, vcc quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf" : "=v"(a[e & 15]) : "v"(a[(e + 5) & 15]), "v"(a[(e + 9) & 15]) : "vcc");
        if (KIND == PERMLANE32_SWAP) asm volatile("v_permlane32_swap_b32 %0, %1" : "+v"(a[e & 15]), "+v"(a[(e + 8) & 15]));
        if (KIND == V_MOV) asm volatile("v_mov_b32 %0, %1" : "=v"(a[e & 15]) : "v"(a[(e + 5) & 15]));
        if (KIND == GLOBAL_LOAD_X4) asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(r[e & 3]) : "v"(gp + (e & 7) * 1024));
        if (KIND == ADD_F64) asm volatile("v_add_f64 %0, %0, %1" : "+v"(y[e & 7]) : "v"(z));
        if (KIND == CVT) asm volatile("v_cvt_f64_i32 %0, %1" : "=v"(y[e & 7]) : "v"(a[e & 15]));
        if (KIND == FMA_F32) asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(f[e & 7]) : "v"(f[(e + 3) & 7]));
        if (KIND == ACCVGPR) asm volatile("v_accvgpr_write_b32 a[%1], %0" : : "v"(a[e & 15]), "n"(e & 15));
      }
    }
    asm volatile("s_waitcnt lgkmcnt(0) vmcnt(0)" ::: "memory");
  }
  const unsigned long long t1 = clock64(), r1 = wall_clock64();
  if (threadIdx.x == 32 && blockIdx.x == 17) { out[0] = t1 - t0; out[2] = r1 - r0; }
  double s = 0;
  for (int i = 0; i < 16; i++) s += x[i] + a[i];
  for (int i = 0; i < 8; i++) s += y[i] + f[i];
  for (int i = 0; i < 4; i++) s += r[i].x + r[i].w;
  if (s == 12345.678) out[1] = 1;
}

template <int KIND, int EXTRA, int FMAS = 64>
void run(const char* name, unsigned long long* d, const f4* g, int threads) {
  const int iters = 40000, blocks = 256;
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  k<KIND, EXTRA, FMAS><<<blocks, threads>>>(d, g, 1.0, iters);
  hipEventRecord(e0);
  k<KIND, EXTRA, FMAS><<<blocks, threads>>>(d, g, 1.0, iters);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms;
  hipEventElapsedTime(&ms, e0, e1);
  unsigned long long h[3];
  hipMemcpy(h, d, 24, hipMemcpyDeviceToHost);
  const int wps = threads / 256;
  const double ns_group_simd = ms * 1e6 / iters / wps;
  printf("%-34s %d wave(s)/SIMD: %8.2f ns per group and SIMD, s_memtime %.3f GHz-equivalent (ticks / 100 MHz wall ticks * 0.1)\n",
         name, wps, ns_group_simd, (double)h[0] / (double)h[2] * 0.1);
}

int main() {
  unsigned long long* d;
  hipMalloc(&d, 32);
  f4* g;
  hipMalloc(&g, 1 << 20);
  hipMemset(g, 0, 1 << 20);
  for (int threads : {256, 512, 1024}) {
    run<NONE, 0>("64 v_fma_f64", d, g, threads);
    run<ADD_F64, 64, 0>("64 v_add_f64 alone", d, g, threads);
    run<V_MOV, 64, 0>("64 v_mov_b32 alone", d, g, threads);
    run<MOV_DPP, 64, 0>("64 v_mov_b32_dpp alone", d, g, threads);
    run<CNDMASK_DPP, 64, 0>("64 v_cndmask_b32_dpp alone", d, g, threads);
    run<PERMLANE32_SWAP, 64, 0>("64 v_permlane32_swap alone", d, g, threads);
    run<DS_WRITE_B64, 16, 0>("16 ds_write_b64 alone", d, g, threads);
    run<DS_READ_B128, 16, 0>("16 ds_read_b128 alone", d, g, threads);
    run<ADD_F64, 32>("64 fma + 32 v_add_f64", d, g, threads);
    run<V_MOV, 32>("64 fma + 32 v_mov_b32", d, g, threads);
    run<FMA_F32, 32>("64 fma + 32 v_fma_f32", d, g, threads);
    run<ACCVGPR, 32>("64 fma + 32 v_accvgpr_write", d, g, threads);
    run<MOV_DPP, 32>("64 fma + 32 v_mov_b32_dpp", d, g, threads);
    run<CNDMASK_DPP, 32>("64 fma + 32 v_cndmask_b32_dpp", d, g, threads);
    run<PERMLANE32_SWAP, 32>("64 fma + 32 v_permlane32_swap", d, g, threads);
    run<CVT, 32>("64 fma + 32 v_cvt_f64_i32", d, g, threads);
    run<DS_WRITE_B64, 8>("64 fma + 8 ds_write_b64", d, g, threads);
    run<DS_WRITE_B64, 16>("64 fma + 16 ds_write_b64", d, g, threads);
    run<DS_WRITE_B128, 8>("64 fma + 8 ds_write_b128", d, g, threads);
    run<DS_READ_B128, 8>("64 fma + 8 ds_read_b128", d, g, threads);
    run<DS_READ_B128, 16>("64 fma + 16 ds_read_b128", d, g, threads);
    run<GLOBAL_LOAD_X4, 8>("64 fma + 8 global_load_dwordx4", d, g, threads);
  }
  return 0;
}
