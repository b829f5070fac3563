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
