// VALU issue-rate probe for gfx950: cycles per wave-instruction of v_fma_f32, v_pk_fma_f32,
// v_pk_add_f32, v_pk_mul_f32 at 1, 2, 4 waves per SIMD (one workgroup of 256/512/1024 threads
// on one CU).  hipcc --offload-arch=gfx950 -O3 valu_rate.hip -o valu_rate && ./valu_rate
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float v2f __attribute__((ext_vector_type(2)));
#define REP 64
template <int KIND>
__global__ void probe(float* out, long long* cyc, int iters) {
  float a0 = threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
  v2f p0 = {a0, a1}, p1 = {a2, a3}, p2 = {a4, a5}, p3 = {a6, a7}, p4 = {a1, a0}, p5 = {a3, a2}, p6 = {a5, a4}, p7 = {a7, a6};
  const float c = 1.0000001f, d = 1e-9f;
  const v2f pc = {c, c}, pd = {d, d};
  long long t0 = __builtin_amdgcn_s_memtime();
  for (int i = 0; i < iters; ++i) {
#pragma unroll
    for (int r = 0; r < REP / 8; ++r) {
      if (KIND == 0) {
        asm volatile("v_fma_f32 %0, %0, %8, %9\n v_fma_f32 %1, %1, %8, %9\n v_fma_f32 %2, %2, %8, %9\n v_fma_f32 %3, %3, %8, %9\n"
                     "v_fma_f32 %4, %4, %8, %9\n v_fma_f32 %5, %5, %8, %9\n v_fma_f32 %6, %6, %8, %9\n v_fma_f32 %7, %7, %8, %9\n"
                     : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(c), "v"(d));
      } else if (KIND == 1) {
        asm volatile("v_pk_fma_f32 %0, %0, %8, %9\n v_pk_fma_f32 %1, %1, %8, %9\n v_pk_fma_f32 %2, %2, %8, %9\n v_pk_fma_f32 %3, %3, %8, %9\n"
                     "v_pk_fma_f32 %4, %4, %8, %9\n v_pk_fma_f32 %5, %5, %8, %9\n v_pk_fma_f32 %6, %6, %8, %9\n v_pk_fma_f32 %7, %7, %8, %9\n"
                     : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3), "+v"(p4), "+v"(p5), "+v"(p6), "+v"(p7) : "v"(pc), "v"(pd));
      } else if (KIND == 2) {
        asm volatile("v_pk_add_f32 %0, %0, %8\n v_pk_add_f32 %1, %1, %8\n v_pk_add_f32 %2, %2, %8\n v_pk_add_f32 %3, %3, %8\n"
                     "v_pk_add_f32 %4, %4, %8\n v_pk_add_f32 %5, %5, %8\n v_pk_add_f32 %6, %6, %8\n v_pk_add_f32 %7, %7, %8\n"
                     : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3), "+v"(p4), "+v"(p5), "+v"(p6), "+v"(p7) : "v"(pd));
      } else if (KIND == 3) {
        asm volatile("v_pk_mul_f32 %0, %0, %8\n v_pk_mul_f32 %1, %1, %8\n v_pk_mul_f32 %2, %2, %8\n v_pk_mul_f32 %3, %3, %8\n"
                     "v_pk_mul_f32 %4, %4, %8\n v_pk_mul_f32 %5, %5, %8\n v_pk_mul_f32 %6, %6, %8\n v_pk_mul_f32 %7, %7, %8\n"
                     : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3), "+v"(p4), "+v"(p5), "+v"(p6), "+v"(p7) : "v"(pc));
      } else {
        asm volatile("v_add_f32 %0, %0, %8\n v_add_f32 %1, %1, %8\n v_add_f32 %2, %2, %8\n v_add_f32 %3, %3, %8\n"
                     "v_add_f32 %4, %4, %8\n v_add_f32 %5, %5, %8\n v_add_f32 %6, %6, %8\n v_add_f32 %7, %7, %8\n"
                     : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(d));
      }
    }
  }
  long long t1 = __builtin_amdgcn_s_memtime();
  out[blockIdx.x * blockDim.x + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 + p0.x + p1.y + p2.x + p3.y + p4.x + p5.y + p6.x + p7.y;
  if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}
int main() {
  float* out; long long* cyc;
  hipMalloc(&out, 1024 * 4); hipMalloc(&cyc, 8);
  const char* names[5] = {"v_fma_f32", "v_pk_fma_f32", "v_pk_add_f32", "v_pk_mul_f32", "v_add_f32"};
  const int iters = 20000;
  for (int kind = 0; kind < 5; ++kind)
    for (int threads = 256; threads <= 1024; threads *= 2) {
      long long h = 0;
      hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
      float ms = 0.f;
      for (int rep = 0; rep < 2; ++rep) {
        hipEventRecord(e0, 0);
        switch (kind) {
          case 0: hipLaunchKernelGGL(probe<0>, dim3(1), dim3(threads), 0, 0, out, cyc, iters); break;
          case 1: hipLaunchKernelGGL(probe<1>, dim3(1), dim3(threads), 0, 0, out, cyc, iters); break;
          case 2: hipLaunchKernelGGL(probe<2>, dim3(1), dim3(threads), 0, 0, out, cyc, iters); break;
          case 3: hipLaunchKernelGGL(probe<3>, dim3(1), dim3(threads), 0, 0, out, cyc, iters); break;
          default: hipLaunchKernelGGL(probe<4>, dim3(1), dim3(threads), 0, 0, out, cyc, iters); break;
        }
        hipEventRecord(e1, 0);
        hipDeviceSynchronize();
        hipEventElapsedTime(&ms, e0, e1);
        hipMemcpy(&h, cyc, 8, hipMemcpyDeviceToHost);
      }
      // s_memtime ticks at 100 MHz?  report raw ticks per instruction and per wave-instruction slot
      printf("%-14s waves/SIMD=%d  ticks=%lld  ticks per instr per wave = %.3f   wall %.3f ms = %.2f ns per instr per wave\n",
             names[kind], threads / 256, h, (double)h / ((double)iters * REP), ms, 1e6 * ms / ((double)iters * REP));
    }
  return 0;
}
