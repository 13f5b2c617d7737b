// Cost of an FP64 vector instruction for ONE wavefront per SIMD as a function of the
// distance (in instructions) to its producer.  hipcc --offload-arch=gfx950 -O3 -o ubench tools/valu_issue_ubench.hip
// Measured on MI355X (clock64 ticks, 2.4 GHz): distance 1: 8.5, 2: 6.4, 3: 5.9, 4: 5.6, >= 6: 5.2-5.4.
#include <hip/hip_runtime.h>
#include <stdio.h>
template <int D, int OP, bool HALF_WAVE = false>
__global__ __launch_bounds__(256, 1) void k(unsigned long long* out, double seed, int iters) {
  double x[16];
  for (int i = 0; i < 16; i++) x[i] = seed + i + threadIdx.x;
  double w = 1.0000001, z = 0.5;
  unsigned long long t0 = clock64();
  if (!HALF_WAVE || (threadIdx.x & 32))       // HALF_WAVE: only lanes 32-63 of each wavefront run the loop
  for (int it = 0; it < iters; it++) {
#pragma unroll
    for (int rep = 0; rep < 48 / D; rep++) {
#pragma unroll
      for (int c = 0; c < D; c++) {
        if (OP == 0) asm volatile("v_add_f64 %0, %0, %1" : "+v"(x[c]) : "v"(z));
        if (OP == 1) asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(x[c]) : "v"(w), "v"(z));
        if (OP == 2) asm volatile("v_fma_f64 %0, %1, %2, %0" : "+v"(x[c]) : "v"(w), "v"(z));
        if (OP == 3) asm volatile("v_mul_f64 %0, %0, %1" : "+v"(x[c]) : "v"(w));
      }
    }
  }
  unsigned long long t1 = clock64();
  if (threadIdx.x == 32 && blockIdx.x == 0) out[0] = t1 - t0;
  double s = 0; for (int i = 0; i < 16; i++) s += x[i];
  if (s == 12345.678) out[1] = 1;
}
template <int D, int OP, bool HALF_WAVE = false> void run(unsigned long long* d) {
  unsigned long long h[2];
  const int iters = 256;
  k<D, OP, HALF_WAVE><<<1, 256>>>(d, 1.0, iters);
  k<D, OP, HALF_WAVE><<<1, 256>>>(d, 1.0, iters);
  hipMemcpy(h, d, 16, hipMemcpyDeviceToHost);
  printf(" %5.2f", (double)h[0] / (iters * (48 / D) * D));
}
template <int OP> void row(const char* name, unsigned long long* d) {
  printf("%-28s", name);
  run<1, OP>(d); run<2, OP>(d); run<3, OP>(d); run<4, OP>(d); run<6, OP>(d); run<8, OP>(d); run<12, OP>(d); run<16, OP>(d);
  printf("\n");
}
int main() {
  unsigned long long* d; hipMalloc(&d, 16);
  printf("ticks per instruction, dependency distance  1     2     3     4     6     8    12    16\n");
  row<0>("v_add_f64 (src0 dep)", d);
  row<1>("v_fma_f64 (src0 dep)", d);
  row<2>("v_fma_f64 (src2 dep)", d);
  row<3>("v_mul_f64 (src0 dep)", d);
  printf("%-28s", "v_fma_f64, lanes 32-63 only");
  run<1, 1, true>(d); run<2, 1, true>(d); run<3, 1, true>(d); run<4, 1, true>(d); run<6, 1, true>(d); run<8, 1, true>(d);
  run<12, 1, true>(d); run<16, 1, true>(d);
  printf("\n");
  return 0;
}
