// Issue cost of the packed fp32 VALU instructions on gfx950 against v_fma_f32: independent
// destinations, one dependent chain, and op_sel broadcast operands.  ns per instruction per SIMD at
// 1 / 2 / 4 waves per SIMD (a wave64 v_fma_f32 is 4 cycles = 1.67 ns at 2.4 GHz).
#include <hip/hip_runtime.h>
#include <stdio.h>
template <int KIND>
__global__ void probe(float* out, int iters) {
  float seed = threadIdx.x * 1e-9f;
  asm volatile("v_mov_b32 v20, %0\n v_mov_b32 v21, %0\n v_mov_b32 v22, %0\n v_mov_b32 v23, %0\n v_mov_b32 v24, %0\n v_mov_b32 v25, %0\n"
               "v_mov_b32 v26, %0\n v_mov_b32 v27, %0\n v_mov_b32 v28, %0\n v_mov_b32 v29, %0\n v_mov_b32 v30, %0\n v_mov_b32 v31, %0\n"
               "v_mov_b32 v32, %0\n v_mov_b32 v33, %0\n v_mov_b32 v34, %0\n v_mov_b32 v35, %0\n v_mov_b32 v36, %0\n v_mov_b32 v37, %0\n"
               "v_mov_b32 v38, %0\n v_mov_b32 v39, %0\n v_mov_b32 v40, %0\n v_mov_b32 v41, %0\n v_mov_b32 v42, %0\n v_mov_b32 v43, %0\n"
               :: "v"(seed) : "v20","v21","v22","v23","v24","v25","v26","v27","v28","v29","v30","v31","v32","v33","v34","v35","v36","v37","v38","v39","v40","v41","v42","v43");
  for (int i = 0; i < iters; ++i) {
#pragma unroll
    for (int r = 0; r < 8; ++r) {
      if (KIND == 0)  // scalar fma, 8 independent destinations
        asm volatile("v_fma_f32 v20, v36, v38, v20\n v_fma_f32 v21, v37, v39, v21\n v_fma_f32 v22, v36, v40, v22\n v_fma_f32 v23, v37, v41, v23\n"
                     "v_fma_f32 v24, v36, v38, v24\n v_fma_f32 v25, v37, v39, v25\n v_fma_f32 v26, v36, v40, v26\n v_fma_f32 v27, v37, v41, v27\n"
                     ::: "v20","v21","v22","v23","v24","v25","v26","v27");
      else if (KIND == 1)  // packed fma, 8 independent destination pairs
        asm volatile("v_pk_fma_f32 v[20:21], v[36:37], v[38:39], v[20:21]\n v_pk_fma_f32 v[22:23], v[36:37], v[40:41], v[22:23]\n"
                     "v_pk_fma_f32 v[24:25], v[36:37], v[38:39], v[24:25]\n v_pk_fma_f32 v[26:27], v[36:37], v[40:41], v[26:27]\n"
                     "v_pk_fma_f32 v[28:29], v[36:37], v[38:39], v[28:29]\n v_pk_fma_f32 v[30:31], v[36:37], v[40:41], v[30:31]\n"
                     "v_pk_fma_f32 v[32:33], v[36:37], v[38:39], v[32:33]\n v_pk_fma_f32 v[34:35], v[36:37], v[40:41], v[34:35]\n"
                     ::: "v20","v21","v22","v23","v24","v25","v26","v27","v28","v29","v30","v31","v32","v33","v34","v35");
      else if (KIND == 2)  // packed fma, one dependent chain
        asm volatile("v_pk_fma_f32 v[20:21], v[36:37], v[20:21], v[38:39]\n v_pk_fma_f32 v[20:21], v[36:37], v[20:21], v[38:39]\n"
                     "v_pk_fma_f32 v[20:21], v[36:37], v[20:21], v[38:39]\n v_pk_fma_f32 v[20:21], v[36:37], v[20:21], v[38:39]\n"
                     "v_pk_fma_f32 v[20:21], v[36:37], v[20:21], v[38:39]\n v_pk_fma_f32 v[20:21], v[36:37], v[20:21], v[38:39]\n"
                     "v_pk_fma_f32 v[20:21], v[36:37], v[20:21], v[38:39]\n v_pk_fma_f32 v[20:21], v[36:37], v[20:21], v[38:39]\n"
                     ::: "v20","v21");
      else if (KIND == 3)  // scalar fma, one dependent chain
        asm volatile("v_fma_f32 v20, v36, v20, v38\n v_fma_f32 v20, v36, v20, v38\n v_fma_f32 v20, v36, v20, v38\n v_fma_f32 v20, v36, v20, v38\n"
                     "v_fma_f32 v20, v36, v20, v38\n v_fma_f32 v20, v36, v20, v38\n v_fma_f32 v20, v36, v20, v38\n v_fma_f32 v20, v36, v20, v38\n"
                     ::: "v20");
      else if (KIND == 4)  // packed mul / add with op_sel broadcast, independent
        asm volatile("v_pk_mul_f32 v[20:21], v[36:37], v[38:39] op_sel_hi:[0,1]\n v_pk_add_f32 v[22:23], v[36:37], v[40:41] op_sel:[1,0]\n"
                     "v_pk_mul_f32 v[24:25], v[36:37], v[38:39] op_sel_hi:[0,1]\n v_pk_add_f32 v[26:27], v[36:37], v[40:41] op_sel:[1,0]\n"
                     "v_pk_mul_f32 v[28:29], v[36:37], v[38:39] op_sel_hi:[0,1]\n v_pk_add_f32 v[30:31], v[36:37], v[40:41] op_sel:[1,0]\n"
                     "v_pk_mul_f32 v[32:33], v[36:37], v[38:39] op_sel_hi:[0,1]\n v_pk_add_f32 v[34:35], v[36:37], v[40:41] op_sel:[1,0]\n"
                     ::: "v20","v21","v22","v23","v24","v25","v26","v27","v28","v29","v30","v31","v32","v33","v34","v35");
      else if (KIND == 5)  // two interleaved dependent packed chains
        asm volatile("v_pk_fma_f32 v[20:21], v[36:37], v[20:21], v[38:39]\n v_pk_fma_f32 v[22:23], v[36:37], v[22:23], v[38:39]\n"
                     "v_pk_fma_f32 v[20:21], v[36:37], v[20:21], v[38:39]\n v_pk_fma_f32 v[22:23], v[36:37], v[22:23], v[38:39]\n"
                     "v_pk_fma_f32 v[20:21], v[36:37], v[20:21], v[38:39]\n v_pk_fma_f32 v[22:23], v[36:37], v[22:23], v[38:39]\n"
                     "v_pk_fma_f32 v[20:21], v[36:37], v[20:21], v[38:39]\n v_pk_fma_f32 v[22:23], v[36:37], v[22:23], v[38:39]\n"
                     ::: "v20","v21","v22","v23");
      else  // v_pk_mov_b32 permutes, independent
        asm volatile("v_pk_mov_b32 v[20:21], v[36:37], v[38:39] op_sel:[0,1]\n v_pk_mov_b32 v[22:23], v[36:37], v[40:41] op_sel:[1,0]\n"
                     "v_pk_mov_b32 v[24:25], v[36:37], v[38:39] op_sel:[0,1]\n v_pk_mov_b32 v[26:27], v[36:37], v[40:41] op_sel:[1,0]\n"
                     "v_pk_mov_b32 v[28:29], v[36:37], v[38:39] op_sel:[0,1]\n v_pk_mov_b32 v[30:31], v[36:37], v[40:41] op_sel:[1,0]\n"
                     "v_pk_mov_b32 v[32:33], v[36:37], v[38:39] op_sel:[0,1]\n v_pk_mov_b32 v[34:35], v[36:37], v[40:41] op_sel:[1,0]\n"
                     ::: "v20","v21","v22","v23","v24","v25","v26","v27","v28","v29","v30","v31","v32","v33","v34","v35");
    }
  }
  float r;
  asm volatile("v_add_f32 %0, v20, v21\n v_add_f32 %0, %0, v22\n v_add_f32 %0, %0, v32" : "=v"(r));
  out[blockIdx.x * blockDim.x + threadIdx.x] = r;
}
template <int K> void run(const char* name, float* out) {
  const int iters = 4000;
  printf("%-52s", name);
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
    printf("  w/SIMD=%d: %.2f ns", threads / 256, 1e6 * ms / ((double)iters * 64 * (threads / 256)));
  }
  printf("\n");
}
int main() {
  float* out; hipMalloc(&out, 256 * 1024 * 4);
  run<0>("v_fma_f32, 8 independent", out);
  run<3>("v_fma_f32, one dependent chain", out);
  run<1>("v_pk_fma_f32, 8 independent pairs", out);
  run<2>("v_pk_fma_f32, one dependent chain", out);
  run<5>("v_pk_fma_f32, two interleaved chains", out);
  run<4>("v_pk_mul/add_f32 with op_sel, independent", out);
  run<6>("v_pk_mov_b32, independent", out);
  return 0;
}
