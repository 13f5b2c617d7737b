// Cost of LDS instructions by width and form, lane-contiguous (conflict-free) addresses, one
// workgroup per compute unit, one and two wavefronts per SIMD.
//   hipcc --offload-arch=gfx950 -O3 -o lds_ubench tools/lds_cost_ubench.hip
// Measured on MI355X (ns per instruction and SIMD, all compute units busy, one wavefront per SIMD):
// ds_read_b128 8.2, ds_read_b64 4.7, ds_read_b32 4.8, ds_read2_b64 26.9, ds_write_b128 23.0,
// ds_write_b64 11.1, ds_write_b32 8.1, ds_write2st64_b64 23.8, ds_write2_b64 27.6: a stored byte costs
// 2.7 times a loaded one, and the two-address forms are no cheaper than two instructions.
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f4 __attribute__((ext_vector_type(4)));
typedef float f2 __attribute__((ext_vector_type(2)));
template <int OP>
__global__ __launch_bounds__(1024) void kl(unsigned long long* out, double seed, int iters) {
  __shared__ __attribute__((aligned(16))) char lds[65536];
  f4 r4[2]; f2 r2[2]; float r1[2]; double st = seed; f4 s4 = f4{1, 2, 3, 4}; float s1 = 1.f;
  r4[0] = r4[1] = f4{0, 0, 0, 0}; r2[0] = r2[1] = f2{0, 0}; r1[0] = r1[1] = 0;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  // per-wave 4 KB region; lane-contiguous granules
  const unsigned a16 = wave * 4096 + lane * 16, a8 = wave * 4096 + lane * 8, a4 = wave * 4096 + lane * 4;
  for (int i = threadIdx.x; i < 16384; i += blockDim.x) reinterpret_cast<float*>(lds)[i] = 0;
  __syncthreads();
  for (int it = 0; it < iters; it++) {
#pragma unroll
    for (int rep = 0; rep < 4; rep++) {
#pragma unroll
      for (int c = 0; c < 16; c++) {
        const unsigned o = (c & 1) * 2048;
        if (OP == 0) asm volatile("ds_read_b128 %0, %1" : "=v"(r4[c & 1]) : "v"(a16 + (c & 1) * 1024));
        if (OP == 1) asm volatile("ds_read_b64 %0, %1" : "=v"(r2[c & 1]) : "v"(a8 + o));
        if (OP == 2) asm volatile("ds_read_b32 %0, %1" : "=v"(r1[c & 1]) : "v"(a4 + o));
        if (OP == 3) asm volatile("ds_write_b128 %0, %1" : : "v"(a16 + (c & 1) * 1024), "v"(s4));
        if (OP == 4) asm volatile("ds_write_b64 %0, %1" : : "v"(a8 + o), "v"(st));
        if (OP == 5) asm volatile("ds_write_b32 %0, %1" : : "v"(a4 + o), "v"(s1));
        if (OP == 6) asm volatile("ds_write2st64_b64 %0, %1, %2 offset1:2" : : "v"(a8), "v"(st), "v"(st));
        if (OP == 7) asm volatile("ds_write2_b64 %0, %1, %2 offset1:1" : : "v"(a16), "v"(st), "v"(st));
        if (OP == 8) asm volatile("ds_read2_b64 %0, %1 offset1:1" : "=v"(r4[c & 1]) : "v"(a16));
      }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    }
  }
  double s = r4[0].x + r4[1].y + r2[0].x + r2[1].y + r1[0] + r1[1];
  if (s == 12345.678) out[1] = 1;
}
template <int OP> void run(const char* name, unsigned long long* d, int threads) {
  const int iters = 20000, blocks = 256;
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  kl<OP><<<blocks, threads>>>(d, 1.0, iters);
  hipEventRecord(e0);
  kl<OP><<<blocks, threads>>>(d, 1.0, iters);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  const double n = (double)iters * 64;
  printf("%-22s %d wave(s)/SIMD: %6.2f ns per instruction of a wavefront, %6.2f ns per instruction and SIMD\n", name,
         threads / 256, ms * 1e6 / n, ms * 1e6 / n / (threads / 256));
}
int main() {
  unsigned long long* d; hipMalloc(&d, 32);
  for (int threads : {256, 512}) {
    run<0>("ds_read_b128", d, threads); run<1>("ds_read_b64", d, threads); run<2>("ds_read_b32", d, threads);
    run<8>("ds_read2_b64", d, threads);
    run<3>("ds_write_b128", d, threads); run<4>("ds_write_b64", d, threads); run<5>("ds_write_b32", d, threads);
    run<6>("ds_write2st64_b64", d, threads); run<7>("ds_write2_b64", d, threads);
  }
  return 0;
}
