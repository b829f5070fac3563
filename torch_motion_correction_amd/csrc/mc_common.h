// Common device/host helpers for libmcorr (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#define MC_OK 0
#define MC_ERR_ARG -1      // bad argument (null pointer, unsupported size ...)
#define MC_ERR_UNSUPPORTED -2  // size / mode not built into this library

#define MC_WG 256  // workgroup size of every FFT kernel: 4 wavefronts of 64

struct __attribute__((aligned(8))) cfloat {
  float x, y;
};

__device__ __forceinline__ cfloat cmake(float a, float b) { return cfloat{a, b}; }
__device__ __forceinline__ cfloat cadd(cfloat a, cfloat b) { return cfloat{a.x + b.x, a.y + b.y}; }
__device__ __forceinline__ cfloat csub(cfloat a, cfloat b) { return cfloat{a.x - b.x, a.y - b.y}; }
__device__ __forceinline__ cfloat cmul(cfloat a, cfloat b) {
  return cfloat{a.x * b.x - a.y * b.y, a.x * b.y + a.y * b.x};
}
// conj(a) * b
__device__ __forceinline__ cfloat cmulc(cfloat a, cfloat b) {
  return cfloat{a.x * b.x + a.y * b.y, a.x * b.y - a.y * b.x};
}
__device__ __forceinline__ cfloat cconj(cfloat a) { return cfloat{a.x, -a.y}; }
__device__ __forceinline__ cfloat cscale(cfloat a, float s) { return cfloat{a.x * s, a.y * s}; }
// multiply by -i (DIR<0) or +i (DIR>0)
template <int DIR>
__device__ __forceinline__ cfloat cmul_i(cfloat a) {
  return DIR < 0 ? cfloat{a.y, -a.x} : cfloat{-a.y, a.x};
}

// sin / cos of an fp32 angle in radians on the transcendental unit (v_sin_f32 / v_cos_f32 take
// revolutions): two-term Cody-Waite reduction of the angle AS GIVEN (the reference's own fp32 angle,
// correct_motion.py:488-494 via fourier_shift_dft_2d), so the result is sin(ang) / cos(ang) to ~1e-6
// absolute for |ang| < 1e5 -- a dozen instructions instead of the ~100 of sincosf.
__device__ __forceinline__ void mc_sincos(float ang, float* sn, float* cs) {
  const float k = rintf(ang * 0.15915494309189535f);
  float r = __builtin_fmaf(-k, 6.28318548202514648f, ang);  // 2 pi rounded to fp32 ...
  r = __builtin_fmaf(-k, -1.74845553e-07f, r);              // ... and the rest of it
  const float rev = r * 0.15915494309189535f;
  *sn = __builtin_amdgcn_sinf(rev);
  *cs = __builtin_amdgcn_cosf(rev);
}

static inline int mc_check_launch() {
  hipError_t e = hipGetLastError();
  return e == hipSuccess ? MC_OK : (int)e;
}

static inline bool mc_is_pow2(int n) { return n > 0 && (n & (n - 1)) == 0; }
static inline int mc_ilog2(int n) {
  int l = 0;
  while ((1 << l) < n) ++l;
  return l;
}
