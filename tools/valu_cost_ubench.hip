// Cost of one vector instruction for ONE wavefront per SIMD, by instruction type, in s_memtime
// ticks, in nanoseconds (HIP events) and in ticks of the 100 MHz wall clock; one workgroup and
// one workgroup per compute unit.  hipcc --offload-arch=gfx950 -O3 -o ubench tools/valu_cost_ubench.hip
// Measured on MI355X: 1.88-2.12 ns (one compute unit busy) and 1.90-2.27 ns (all busy) per
// instruction whatever its type (v_mov_b32 1.96, v_add_f32 1.88, v_add_f64 2.07, v_fma_f64 2.12,
// v_accvgpr_write_b32 2.25); four clocks of 2.4 GHz would be 1.67 ns.
// Second part: a transform-like mix (12 v_fma_f64, 2 ds_read_b128, 2 ds_write_b64) at one, two and
// four wavefronts per SIMD.  Measured: 94.6 ns per group with one wavefront per SIMD (the 12 FMAs
// alone: 34.2), 47.9 ns per group and SIMD with two, 46.7 with four: a lone wavefront sits out the
// LDS round trip at the wait that ends each group, a second wavefront fills it.
#include <hip/hip_runtime.h>
#include <stdio.h>
template <int OP>
__global__ __launch_bounds__(1024) void k(unsigned long long* out, double seed, int iters) {
  double x[16]; float f[16]; unsigned a[16];
  for (int i = 0; i < 16; i++) { x[i] = seed + i + threadIdx.x; f[i] = (float)x[i]; a[i] = i; }
  double w = 1.0000001, z = 0.5; float wf = 1.0001f, zf = 0.5f;
  unsigned long long t0 = clock64();
  unsigned long long r0 = wall_clock64();
  for (int it = 0; it < iters; it++) {
#pragma unroll
    for (int rep = 0; rep < 4; rep++) {
#pragma unroll
      for (int c = 0; c < 16; c++) {
        if (OP == 0) asm volatile("v_add_f64 %0, %0, %1" : "+v"(x[c]) : "v"(z));
        if (OP == 1) asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(x[c]) : "v"(w), "v"(z));
        if (OP == 2) asm volatile("v_add_f32 %0, %0, %1" : "+v"(f[c]) : "v"(zf));
        if (OP == 3) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(f[c]) : "v"(wf), "v"(zf));
        if (OP == 4) asm volatile("v_mov_b32 %0, %1" : "+v"(a[c]) : "v"(a[(c + 1) & 15]));
        if (OP == 5) asm volatile("v_accvgpr_write_b32 a[%1], %0" : : "v"(a[c]), "n"(c));
        if (OP == 6) asm volatile("v_mul_f64 %0, %0, %1" : "+v"(x[c]) : "v"(w));
        if (OP == 7) asm volatile("v_fma_f64 %0, %0, %1, %0" : "+v"(x[c]) : "v"(w));
        if (OP == 8) asm volatile("v_fma_f64 %0, %0, 0.5, 1.0" : "+v"(x[c]));
        if (OP == 9) asm volatile("v_add_f64 %0, %0, 1.0" : "+v"(x[c]));
      }
    }
  }
  unsigned long long t1 = clock64();
  unsigned long long r1 = wall_clock64();
  if (threadIdx.x == 32 && blockIdx.x == 0) { out[0] = t1 - t0; out[2] = r1 - r0; }
  double s = 0; for (int i = 0; i < 16; i++) s += x[i] + f[i] + a[i];
  if (s == 12345.678) out[1] = 1;
}
// LDS instructions and a transform-like mix (12 v_fma_f64, 2 ds_read_b128, 2 ds_write_b64 per 16),
// to see whether a second wavefront on the SIMD overlaps one's LDS transfers with the other's
// arithmetic.  OP 0: ds_read_b128, 1: ds_write_b64, 2: the mix, 3: the mix's 12 FMAs alone.
typedef float f4 __attribute__((ext_vector_type(4)));
template <int OP>
__global__ __launch_bounds__(1024) void kl(unsigned long long* out, double seed, int iters) {
  __shared__ __attribute__((aligned(16))) char lds[1024 * 16 * 2];
  double x[12]; f4 r[2]; double st = seed;
  for (int i = 0; i < 12; i++) x[i] = seed + i + threadIdx.x;
  r[0] = r[1] = f4{0, 0, 0, 0};
  const unsigned addr = threadIdx.x * 16;
  *reinterpret_cast<f4*>(lds + addr) = r[0];
  *reinterpret_cast<f4*>(lds + 16384 + addr) = r[0];
  __syncthreads();
  double w = 1.0000001, z = 0.5;
  for (int it = 0; it < iters; it++) {
#pragma unroll
    for (int rep = 0; rep < 4; rep++) {
      if (OP == 0) {
#pragma unroll
        for (int c = 0; c < 16; c++) asm volatile("ds_read_b128 %0, %1" : "=v"(r[c & 1]) : "v"(addr + (c & 1) * 16384));
      } else if (OP == 1) {
#pragma unroll
        for (int c = 0; c < 16; c++) asm volatile("ds_write_b64 %0, %1" : : "v"(addr + (c & 1) * 16384), "v"(st));
      } else {
#pragma unroll
        for (int c = 0; c < 12; c++) {
          asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(x[c]) : "v"(w), "v"(z));
          if (OP == 2 && (c == 2 || c == 8)) asm volatile("ds_read_b128 %0, %1" : "=v"(r[c == 2]) : "v"(addr + (c == 2) * 16384));
          if (OP == 2 && (c == 5 || c == 11)) asm volatile("ds_write_b64 %0, %1" : : "v"(addr + (c == 5) * 16384), "v"(st));
        }
      }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    }
  }
  double s = r[0].x + r[1].y; for (int i = 0; i < 12; i++) s += x[i];
  if (s == 12345.678) out[1] = 1;
}
template <int OP> void run_lds(const char* name, unsigned long long* d, int threads, int per16) {
  const int iters = 20000, blocks = 256;
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  kl<OP><<<blocks, threads>>>(d, 1.0, iters);
  hipEventRecord(e0);
  kl<OP><<<blocks, threads>>>(d, 1.0, iters);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  const double groups = (double)iters * 4;      // groups of `per16` instructions per wavefront
  printf("%-44s %d wavefront(s) per SIMD: %7.2f ns per group of a wavefront, %6.2f ns per group and SIMD\n", name,
         threads / 256, ms * 1e6 / groups, ms * 1e6 / groups / (threads / 256));
  (void)per16;
}

template <int OP> void run(const char* name, unsigned long long* d, int blocks, int threads = 256) {
  unsigned long long h[3];
  const int iters = 20000;
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  k<OP><<<blocks, threads>>>(d, 1.0, iters);
  hipEventRecord(e0);
  k<OP><<<blocks, threads>>>(d, 1.0, iters);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  hipMemcpy(h, d, 24, hipMemcpyDeviceToHost);
  const double n = (double)iters * 64;
  // FMA-equivalent rate of the whole launch: 64 lanes x 2 flop per wave-instruction
  const double tflops = (double)blocks * (threads / 64) * n * 128.0 / (ms * 1e-3) / 1e12;
  printf("%-28s blocks %3d x %4d threads: %5.2f clock64 ticks, %6.3f ns per instruction of a wavefront, "
         "%5.1f T lane-FMA-flop/s\n", name, blocks, threads, h[0] / n, ms * 1e6 / n, tflops);
}
int main() {
  unsigned long long* d; hipMalloc(&d, 32);
  for (int blocks : {1, 256}) {
    run<0>("v_add_f64 v,v,v", d, blocks);
    run<9>("v_add_f64 v,v,const", d, blocks);
    run<6>("v_mul_f64 v,v,v", d, blocks);
    run<1>("v_fma_f64 v,v,v,v", d, blocks);
    run<7>("v_fma_f64 v,v,v,v(same)", d, blocks);
    run<8>("v_fma_f64 v,v,const,const", d, blocks);
    run<2>("v_add_f32", d, blocks);
    run<3>("v_fma_f32", d, blocks);
    run<4>("v_mov_b32", d, blocks);
    run<5>("v_accvgpr_write_b32", d, blocks);
  }
  // more wavefronts per SIMD: the per-SIMD rate, not only the per-wavefront cadence
  for (int threads : {512, 1024}) {
    run<1>("v_fma_f64 v,v,v,v", d, 256, threads);
    run<0>("v_add_f64 v,v,v", d, 256, threads);
    run<3>("v_fma_f32", d, 256, threads);
  }
  for (int threads : {256, 512, 1024}) {
    run_lds<0>("16 ds_read_b128", d, threads, 16);
    run_lds<1>("16 ds_write_b64", d, threads, 16);
    run_lds<3>("12 v_fma_f64", d, threads, 12);
    run_lds<2>("12 v_fma_f64 + 2 ds_read_b128 + 2 ds_write_b64", d, threads, 16);
  }
  return 0;
}
