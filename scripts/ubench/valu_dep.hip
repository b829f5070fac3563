// Dependent-issue latency probe (gfx950): one wave per SIMD, chains of length 1, 2, 4 of
// v_add_f32 / v_pk_add_f32 / v_pk_fma_f32 / v_fma_f32: ns per instruction.
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float v2f __attribute__((ext_vector_type(2)));
template <int KIND, int CH>
__global__ void probe(float* out, int iters) {
  float a[4] = {(float)threadIdx.x, 1.f, 2.f, 3.f};
  v2f p[4] = {{a[0], 1.f}, {2.f, 3.f}, {4.f, 5.f}, {6.f, 7.f}};
  const float d = 1e-9f, c = 1.0000001f;
  const v2f pd = {d, d}, pc = {c, c};
  for (int i = 0; i < iters; ++i) {
#pragma unroll
    for (int r = 0; r < 64 / CH; ++r) {
#pragma unroll
      for (int j = 0; j < CH; ++j) {
        if (KIND == 0) asm volatile("v_add_f32 %0, %0, %1" : "+v"(a[j]) : "v"(d));
        if (KIND == 1) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(p[j]) : "v"(pd));
        if (KIND == 2) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(p[j]) : "v"(pc), "v"(pd));
        if (KIND == 3) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[j]) : "v"(c), "v"(d));
      }
    }
  }
  out[threadIdx.x] = a[0] + a[1] + a[2] + a[3] + p[0].x + p[1].y + p[2].x + p[3].y;
}
template <int KIND, int CH>
static void run(const char* name, float* out) {
  const int iters = 20000;
  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  float ms = 0;
  for (int rep = 0; rep < 2; ++rep) {
    (void)hipEventRecord(e0, 0);
    hipLaunchKernelGGL((probe<KIND, CH>), dim3(1), dim3(256), 0, 0, out, iters);
    (void)hipEventRecord(e1, 0);
    (void)hipDeviceSynchronize();
    (void)hipEventElapsedTime(&ms, e0, e1);
  }
  printf("%-14s chains=%d  %.2f ns per instr\n", name, CH, 1e6 * ms / (iters * 64.0));
}
int main() {
  float* out; (void)hipMalloc(&out, 4096);
  run<0, 1>("v_add_f32", out); run<0, 2>("v_add_f32", out); run<0, 4>("v_add_f32", out);
  run<3, 1>("v_fma_f32", out); run<3, 2>("v_fma_f32", out); run<3, 4>("v_fma_f32", out);
  run<1, 1>("v_pk_add_f32", out); run<1, 2>("v_pk_add_f32", out); run<1, 4>("v_pk_add_f32", out);
  run<2, 1>("v_pk_fma_f32", out); run<2, 2>("v_pk_fma_f32", out); run<2, 4>("v_pk_fma_f32", out);
  return 0;
}
