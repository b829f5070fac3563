// Does the VGPR bank of the source operands decide whether v_fma_f32 issues in 2 or in 4 cycles?
// Fixed physical registers: A = all three sources in the same bank (index mod 4), B = three banks,
// C = dst==src0 accumulate with two shared constants (the shape of valu_ops.hip).
#include <hip/hip_runtime.h>
#include <stdio.h>
template <int KIND>
__global__ void probe(float* out, int iters) {
  float seed = threadIdx.x * 1e-9f;
  asm volatile("v_mov_b32 v20, %0\n v_mov_b32 v21, %0\n v_mov_b32 v22, %0\n v_mov_b32 v23, %0\n v_mov_b32 v24, %0\n v_mov_b32 v25, %0\n"
               "v_mov_b32 v26, %0\n v_mov_b32 v27, %0\n v_mov_b32 v28, %0\n v_mov_b32 v29, %0\n v_mov_b32 v30, %0\n v_mov_b32 v31, %0\n"
               "v_mov_b32 v32, %0\n v_mov_b32 v33, %0\n v_mov_b32 v34, %0\n v_mov_b32 v35, %0\n v_mov_b32 v36, %0\n v_mov_b32 v40, %0\n v_mov_b32 v44, %0\n"
               :: "v"(seed) : "v20","v21","v22","v23","v24","v25","v26","v27","v28","v29","v30","v31","v32","v33","v34","v35","v36","v40","v44");
  for (int i = 0; i < iters; ++i) {
#pragma unroll
    for (int r = 0; r < 8; ++r) {
      if (KIND == 0)  // sources v24,v28,v32 / v36.. : all bank 0; 8 different destinations
        asm volatile("v_fma_f32 v20, v24, v28, v32\n v_fma_f32 v21, v28, v32, v36\n v_fma_f32 v22, v32, v36, v40\n v_fma_f32 v23, v36, v40, v44\n"
                     "v_fma_f32 v25, v24, v28, v32\n v_fma_f32 v26, v28, v32, v36\n v_fma_f32 v27, v32, v36, v40\n v_fma_f32 v29, v36, v40, v44\n"
                     ::: "v20","v21","v22","v23","v25","v26","v27","v29");
      else if (KIND == 1)  // sources in three different banks
        asm volatile("v_fma_f32 v20, v24, v29, v34\n v_fma_f32 v21, v28, v33, v30\n v_fma_f32 v22, v32, v25, v26\n v_fma_f32 v23, v36, v29, v34\n"
                     "v_fma_f32 v35, v24, v33, v30\n v_fma_f32 v31, v28, v25, v26\n v_fma_f32 v27, v32, v29, v34\n v_fma_f32 v40, v36, v33, v30\n"
                     ::: "v20","v21","v22","v23","v35","v31","v27","v40");
      else if (KIND == 2)  // two sources in one bank, third elsewhere
        asm volatile("v_fma_f32 v20, v24, v28, v33\n v_fma_f32 v21, v28, v32, v29\n v_fma_f32 v22, v32, v36, v25\n v_fma_f32 v23, v36, v40, v34\n"
                     "v_fma_f32 v35, v24, v28, v33\n v_fma_f32 v31, v28, v32, v29\n v_fma_f32 v27, v32, v36, v25\n v_fma_f32 v30, v36, v40, v34\n"
                     ::: "v20","v21","v22","v23","v35","v31","v27","v30");
      else if (KIND == 3)  // v_fmac (dst is also a source), two other sources same bank as dst
        asm volatile("v_fmac_f32 v20, v24, v28\n v_fmac_f32 v32, v36, v40\n v_fmac_f32 v44, v24, v28\n v_fmac_f32 v20, v36, v40\n"
                     "v_fmac_f32 v32, v24, v28\n v_fmac_f32 v44, v36, v40\n v_fmac_f32 v20, v24, v28\n v_fmac_f32 v32, v36, v40\n"
                     ::: "v20","v32","v44");
      else  // v_fmac, operands spread over banks
        asm volatile("v_fmac_f32 v20, v25, v30\n v_fmac_f32 v21, v26, v31\n v_fmac_f32 v22, v27, v28\n v_fmac_f32 v23, v24, v29\n"
                     "v_fmac_f32 v32, v25, v30\n v_fmac_f32 v33, v26, v31\n v_fmac_f32 v34, v27, v28\n v_fmac_f32 v35, v24, v29\n"
                     ::: "v20","v21","v22","v23","v32","v33","v34","v35");
    }
  }
  float r;
  asm volatile("v_add_f32 %0, v20, v21\n v_add_f32 %0, %0, v22\n v_add_f32 %0, %0, v32" : "=v"(r));
  out[blockIdx.x * blockDim.x + threadIdx.x] = r;
}
template <int K> void run(const char* name, float* out) {
  const int iters = 4000;
  printf("%-44s", name);
  for (int threads = 256; threads <= 1024; threads *= 2) {
    float ms = 0.f;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int rep = 0; rep < 2; ++rep) {
      hipEventRecord(e0, 0);
      hipLaunchKernelGGL(probe<K>, dim3(256), dim3(threads), 0, 0, out, iters);
      hipEventRecord(e1, 0);
      hipDeviceSynchronize();
      hipEventElapsedTime(&ms, e0, e1);
    }
    printf("  w/SIMD=%d: %.2f ns/instr/SIMD", threads / 256, 1e6 * ms / ((double)iters * 64 * (threads / 256)));
  }
  printf("\n");
}
int main() {
  float* out; hipMalloc(&out, 256 * 1024 * 4);
  run<0>("v_fma: 3 sources in ONE bank", out);
  run<1>("v_fma: 3 sources in three banks", out);
  run<2>("v_fma: 2 sources in one bank", out);
  run<3>("v_fmac: dst and both sources in one bank", out);
  run<4>("v_fmac: operands spread over banks", out);
  return 0;
}
