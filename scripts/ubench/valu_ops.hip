// Issue rate of common VALU opcodes on gfx950: cycles per wave-instruction per SIMD at 1 / 2 / 4 waves
// per SIMD (one workgroup on one CU), eight independent destination registers per opcode.
// hipcc --offload-arch=gfx950 -O3 valu_ops.hip -o valu_ops && ./valu_ops
#include <hip/hip_runtime.h>
#include <stdio.h>
#define OP8(S) S(0) S(1) S(2) S(3) S(4) S(5) S(6) S(7)
#define REGS : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(c), "v"(d)
#define K_FMA(i) "v_fma_f32 %" #i ", %" #i ", %8, %9\n"
#define K_ADD(i) "v_add_f32 %" #i ", %" #i ", %9\n"
#define K_MUL(i) "v_mul_f32 %" #i ", %" #i ", %8\n"
#define K_FLOOR(i) "v_floor_f32 %" #i ", %" #i "\n"
#define K_CVTI(i) "v_cvt_i32_f32 %" #i ", %" #i "\n"
#define K_CVTF(i) "v_cvt_f32_i32 %" #i ", %" #i "\n"
#define K_CND(i) "v_cndmask_b32 %" #i ", %" #i ", %8, vcc\n"
#define K_CMP(i) "v_cmp_ge_f32 vcc, %" #i ", %8\n"
#define K_ADDU(i) "v_add_u32 %" #i ", %" #i ", %9\n"
#define K_MADU(i) "v_mad_u32_u24 %" #i ", %" #i ", %8, %9\n"
#define K_LSHL(i) "v_lshlrev_b32 %" #i ", 2, %" #i "\n"
#define K_MOV(i) "v_mov_b32 %" #i ", %8\n"
#define K_MAX(i) "v_max_f32 %" #i ", %" #i ", %9\n"
#define K_MED3(i) "v_med3_i32 %" #i ", %" #i ", %8, %9\n"
#define K_FRACT(i) "v_fract_f32 %" #i ", %" #i "\n"
#define K_RCP(i) "v_rcp_f32 %" #i ", %" #i "\n"
#define K_LSHLADD(i) "v_lshl_add_u32 %" #i ", %" #i ", 2, %9\n"
#define K_FMAC(i) "v_fmac_f32 %" #i ", %8, %9\n"
#define K_MULLO(i) "v_mul_lo_u32 %" #i ", %" #i ", %8\n"
#define K_MADI64(i) "v_add3_u32 %" #i ", %" #i ", %8, %9\n"
template <int KIND>
__global__ void probe(float* out, long long* cyc, int iters) {
  float a0 = threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
  const float c = 1.0000001f, d = 1e-9f;
  long long t0 = __builtin_amdgcn_s_memtime();
  for (int i = 0; i < iters; ++i) {
#pragma unroll
    for (int r = 0; r < 8; ++r) {
#define CASE(n, M) if (KIND == n) asm volatile(OP8(M) REGS);
      CASE(0, K_FMA) CASE(1, K_ADD) CASE(2, K_MUL) CASE(3, K_FLOOR) CASE(4, K_CVTI) CASE(5, K_CVTF)
      CASE(6, K_CND) CASE(7, K_CMP) CASE(8, K_ADDU) CASE(9, K_MADU) CASE(10, K_LSHL) CASE(11, K_MOV)
      CASE(12, K_MAX) CASE(13, K_MED3) CASE(14, K_FRACT) CASE(15, K_RCP) CASE(16, K_LSHLADD) CASE(17, K_FMAC)
      CASE(18, K_MULLO) CASE(19, K_MADI64)
    }
  }
  long long t1 = __builtin_amdgcn_s_memtime();
  out[blockIdx.x * blockDim.x + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7;
  if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}
template <int K> void run(const char* name, float* out, long long* cyc) {
  const int iters = 5000;
  printf("%-16s", name);
  for (int threads = 256; threads <= 1024; threads *= 2) {
    float ms = 0.f;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int rep = 0; rep < 2; ++rep) {
      hipEventRecord(e0, 0);
      hipLaunchKernelGGL(probe<K>, dim3(1), dim3(threads), 0, 0, out, cyc, iters);
      hipEventRecord(e1, 0);
      hipDeviceSynchronize();
      hipEventElapsedTime(&ms, e0, e1);
    }
    long long h = 0; hipMemcpy(&h, cyc, 8, hipMemcpyDeviceToHost);
    const double n = (double)iters * 64 * (threads / 256);  // wave-instructions per SIMD
    printf("  w/SIMD=%d: %.2f cyc/instr/SIMD (memtime) %.2f ns", threads / 256, (double)h / n, 1e6 * ms / n);
  }
  printf("\n");
}
int main() {
  float* out; long long* cyc;
  hipMalloc(&out, 1024 * 4); hipMalloc(&cyc, 8);
  run<0>("v_fma_f32", out, cyc); run<1>("v_add_f32", out, cyc); run<2>("v_mul_f32", out, cyc); run<17>("v_fmac_f32", out, cyc);
  run<3>("v_floor_f32", out, cyc); run<14>("v_fract_f32", out, cyc); run<4>("v_cvt_i32_f32", out, cyc); run<5>("v_cvt_f32_i32", out, cyc);
  run<6>("v_cndmask_b32", out, cyc); run<7>("v_cmp_ge_f32", out, cyc); run<12>("v_max_f32", out, cyc); run<13>("v_med3_i32", out, cyc);
  run<8>("v_add_u32", out, cyc); run<9>("v_mad_u32_u24", out, cyc); run<10>("v_lshlrev_b32", out, cyc); run<16>("v_lshl_add_u32", out, cyc);
  run<19>("v_add3_u32", out, cyc); run<18>("v_mul_lo_u32", out, cyc); run<11>("v_mov_b32", out, cyc); run<15>("v_rcp_f32", out, cyc);
  return 0;
}
