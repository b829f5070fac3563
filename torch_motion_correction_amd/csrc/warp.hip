// Deformation-field warp: cubic-spline lattice evaluation, bicubic shift upsample and
// bicubic frame resample (+ optional fused frame sum).
//
// Reference path (correct_motion.py:18-185, deformation_field_utils.py:9-93): per frame
//   lattice(2,10gh,10gw) = spline(field)(t_i, linspace, linspace)
//   shifts(h,w,2)        = grid_sample(lattice, bicubic, reflection, align_corners) / pixel_spacing
//   out(h,w)             = grid_sample(frame, pixel+shift, bicubic, border, align_corners),
//                          zero where the coordinate leaves [0,h-1]x[0,w-1]
// The reference materialises the coordinate grid, the normalised grid and the shift
// grid (3 x 128 MiB per 4096^2 frame) and gathers 16+16 taps per pixel.  Here the
// x-direction of the shift upsample is hoisted into a small per-frame table
// E[c][lattice row][x] (the reference's own summation order: x taps first, then y),
// so each pixel needs 4 table rows per channel; coordinates live in registers only.
//
// The fp32 coordinate chain is reproduced operation by operation (no FMA
// contraction in this file): at coordinates ~4096 one ulp is 2.4e-4 px, which is
// visible at the 1e-4 parity bar.
#pragma clang fp contract(off)
#include <type_traits>
#include <stdlib.h>
#include <stdio.h>
#include "mc_common.h"
#include "mcorr.h"

// ATen cubic convolution coefficients, A = -0.75 (UpSample.h / GridSamplerKernel.cpp)
__device__ __forceinline__ void cubic_coeffs(float t, float c[4]) {
  const float A = -0.75f;
  float x = t + 1.f;
  c[0] = ((A * x - 5.f * A) * x + 8.f * A) * x - 4.f * A;
  x = t;
  c[1] = ((A + 2.f) * x - (A + 3.f)) * x * x + 1.f;
  x = 1.f - t;
  c[2] = ((A + 2.f) * x - (A + 3.f)) * x * x + 1.f;
  x = 2.f - t;
  c[3] = ((A * x - 5.f * A) * x + 8.f * A) * x - 4.f * A;
}

// grid_sample align_corners=True un-normalisation applied to an array coordinate that
// went through array_to_grid_sample:  ((c / (0.5 n - 0.5) - 1) + 1) * ((n - 1) / 2)
// The division by the loop-invariant d = 0.5 n - 0.5 is done as q = c r, e = fma(-q, d, c),
// q' = fma(e, r, q) with r = RN(1/d): that IS the correctly rounded quotient (Markstein; the one
// exception, a divisor whose significand is all ones, cannot occur for d with <= 15 significant
// bits; checked against exact rational arithmetic for the frame sizes in use, tests/test_host.py)
// at 3 instructions instead of the ~12 of a general IEEE division -- twice per pixel.
__device__ __forceinline__ float grid_chain(float c, float n) {
  const float d = 0.5f * n - 0.5f;
  const float r = 1.0f / d;
  float q = c * r;
  const float e = __builtin_fmaf(-q, d, c);
  q = __builtin_fmaf(e, r, q);
  const float g = q - 1.f;
  return (g + 1.f) * ((n - 1.f) / 2.f);
}

// s / d for a loop-invariant divisor, same three-instruction correctly rounded form
__device__ __forceinline__ float div_invariant(float s, float d) {
  const float r = 1.0f / d;
  const float q = s * r;
  return __builtin_fmaf(__builtin_fmaf(-q, d, s), r, q);
}

__device__ __forceinline__ int reflect_index(int i, int size) {
  const int span = size - 1;
  if (span <= 0) return 0;
  int a = i < 0 ? -i : i;
  const int flips = a / span;
  const int extra = a - flips * span;
  int r = (flips & 1) ? span - extra : extra;
  if (r < 0) r = 0;
  if (r > size - 1) r = size - 1;
  return r;
}

// Per-axis tables of the lattice upsample (get_pixel_shifts, correct_motion.py:161-179):
// for pixel index p of an axis of length n sampled from a lattice axis of length G.
__global__ void warp_axis_tables(int n, int G, int* __restrict__ tap, float* __restrict__ coef) {
  const int p = blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= n) return;
  const float normalized = (float)p / (float)(n - 1);
  const float interp = normalized * (float)(G - 1);
  const float u = grid_chain(interp, (float)G);
  const float fl = floorf(u);
  float c[4];
  cubic_coeffs(u - fl, c);
  const int i0 = (int)fl;
  for (int k = 0; k < 4; ++k) {
    tap[4 * p + k] = reflect_index(i0 - 1 + k, G);
    coef[4 * p + k] = c[k];
  }
}

// E[f][c][R][x] = sum_j cx_j(x) * lattice[f][c][R][tap_j(x)]   (x-direction first)
__global__ void warp_etab(const float* __restrict__ lattice, int GH, int GW, int w,
                          const int* __restrict__ xtap, const float* __restrict__ xcoef,
                          float* __restrict__ etab) {
  const int x = blockIdx.x * blockDim.x + threadIdx.x;
  const int row = blockIdx.y;  // (f*2 + c)*GH + R
  if (x >= w) return;
  const float* L = lattice + (int64_t)row * GW;
  const int4 t = *reinterpret_cast<const int4*>(xtap + 4 * x);
  const float4 c = *reinterpret_cast<const float4*>(xcoef + 4 * x);
  etab[(int64_t)row * w + x] = ((c.x * L[t.x] + c.y * L[t.y]) + c.z * L[t.z]) + c.w * L[t.w];
}

#define WARP_TX 32   // threads across, 4 px each -> 128 px
#define WARP_TY 8    // thread rows, 2 adjacent pixel rows each -> 16 rows
#define WARP_PX 4
#define WARP_ROWS 2

struct WarpArgs {
  const float* frames;
  int nframes, h, w, GH;
  const float* etab;   // [f][2][GH][w]
  const int* ytap;     // [h][4]
  const float* ycoef;  // [h][4]
  float pixel_spacing;
  float* out_frames;
  float* out_sum;
  int tiles_x, tiles_y;
};

typedef float f4u __attribute__((ext_vector_type(4), aligned(4)));  // dword-aligned 16-B load

// Everything below that is not the coordinate chain may contract to FMA: ATen's own
// vectorised kernel is built with FMA contraction and differs from any fixed op order
// at the ulp level anyway (probed).
#pragma clang fp contract(fast)
__device__ __forceinline__ void cubic_coeffs_fast(float t, float c[4]) {
  const float A = -0.75f;
  float x = t + 1.f;
  c[0] = ((A * x - 5.f * A) * x + 8.f * A) * x - 4.f * A;
  x = t;
  c[1] = ((A + 2.f) * x - (A + 3.f)) * x * x + 1.f;
  x = 1.f - t;
  c[2] = ((A + 2.f) * x - (A + 3.f)) * x * x + 1.f;
  x = 2.f - t;
  c[3] = ((A * x - 5.f * A) * x + 8.f * A) * x - 4.f * A;
}
__device__ __forceinline__ float dot4(const float4 c, float e0, float e1, float e2, float e3) {
  return ((c.x * e0 + c.y * e1) + c.z * e2) + c.w * e3;
}
// 5-tap accumulate: taps 0..4 of a window, weights shifted by one when `up`
__device__ __forceinline__ float dot5(const float wt[4], bool up, float v0, float v1, float v2,
                                      float v3, float v4) {
  const float a0 = up ? 0.f : wt[0];
  const float a1 = up ? wt[0] : wt[1];
  const float a2 = up ? wt[1] : wt[2];
  const float a3 = up ? wt[2] : wt[3];
  const float a4 = up ? wt[3] : 0.f;
  return (((a0 * v0 + a1 * v1) + a2 * v2) + a3 * v3) + a4 * v4;
}
#pragma clang fp contract(off)

struct TapWindow {   // rows by..by+4, cols bx..bx+7 of one frame
  float v[5][8];
  int by, bx;
  bool valid;
};

__device__ __forceinline__ void window_load_row(TapWindow& win, int i, const float* fr, int w) {
  const float* r = fr + (int64_t)(win.by + i) * w + win.bx;
  const f4u lo = *reinterpret_cast<const f4u*>(r);
  const f4u hi = *reinterpret_cast<const f4u*>(r + 4);
  win.v[i][0] = lo.x; win.v[i][1] = lo.y; win.v[i][2] = lo.z; win.v[i][3] = lo.w;
  win.v[i][4] = hi.x; win.v[i][5] = hi.y; win.v[i][6] = hi.z; win.v[i][7] = hi.w;
}

// Position the window at (by, bx); reuse rows when it only moved down by one.
__device__ __forceinline__ void window_seek(TapWindow& win, int by, int bx, const float* fr, int w) {
  if (win.valid && win.bx == bx && win.by == by) return;
  if (win.valid && win.bx == bx && win.by + 1 == by) {
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 8; ++j) win.v[i][j] = win.v[i + 1][j];
    win.by = by;
    window_load_row(win, 4, fr, w);
    return;
  }
  win.by = by;
  win.bx = bx;
  win.valid = true;
#pragma unroll
  for (int i = 0; i < 5; ++i) window_load_row(win, i, fr, w);
}

template <bool UNIT_PS>
__device__ __forceinline__ void warp_row(const WarpArgs& a, const float* fr, int y, int x0,
                                         const float4 yc, const float4 Ey[4], const float4 Ex[4],
                                         TapWindow& win, float res[WARP_PX]) {
  const int h = a.h, w = a.w;
  const float fh = (float)h, fw = (float)w;
  float uy[WARP_PX], ux[WARP_PX], fy[WARP_PX], fx[WARP_PX];
  bool inside[WARP_PX];
  const float ey[4][4] = {{Ey[0].x, Ey[0].y, Ey[0].z, Ey[0].w}, {Ey[1].x, Ey[1].y, Ey[1].z, Ey[1].w},
                          {Ey[2].x, Ey[2].y, Ey[2].z, Ey[2].w}, {Ey[3].x, Ey[3].y, Ey[3].z, Ey[3].w}};
  const float ex[4][4] = {{Ex[0].x, Ex[0].y, Ex[0].z, Ex[0].w}, {Ex[1].x, Ex[1].y, Ex[1].z, Ex[1].w},
                          {Ex[2].x, Ex[2].y, Ex[2].z, Ex[2].w}, {Ex[3].x, Ex[3].y, Ex[3].z, Ex[3].w}};
  float fby = 3.0e38f, fbx = 3.0e38f;
#pragma unroll
  for (int k = 0; k < WARP_PX; ++k) {
    float sy = dot4(yc, ey[0][k], ey[1][k], ey[2][k], ey[3][k]);
    float sx = dot4(yc, ex[0][k], ex[1][k], ex[2][k], ex[3][k]);
    if (!UNIT_PS) {
      sy = div_invariant(sy, a.pixel_spacing);
      sx = div_invariant(sx, a.pixel_spacing);
    }
    const float cy = (float)y + sy, cx = (float)(x0 + k) + sx;
    inside[k] = (cy >= 0.f) && (cy <= fh - 1.f) && (cx >= 0.f) && (cx <= fw - 1.f);
    uy[k] = grid_chain(cy, fh);
    ux[k] = grid_chain(cx, fw);
    fy[k] = floorf(uy[k]);
    fx[k] = floorf(ux[k]);
    fby = fminf(fby, fy[k]);
    fbx = fminf(fbx, fx[k] - (float)k);
  }
  bool ok = (fby >= 1.f) && (fby + 3.f <= fh - 1.f) && (fbx >= 1.f) && (fbx + 6.f <= fw - 1.f);
#pragma unroll
  for (int k = 0; k < WARP_PX; ++k) {
    const float dy = fy[k] - fby, dx = fx[k] - (float)k - fbx;
    ok = ok && (dy == 0.f || dy == 1.f) && (dx == 0.f || dx == 1.f);
  }
  if (ok) {
    window_seek(win, (int)fby - 1, (int)fbx - 1, fr, w);
#pragma unroll
    for (int k = 0; k < WARP_PX; ++k) {
      float wy[4], wx[4];
      cubic_coeffs_fast(uy[k] - fy[k], wy);
      cubic_coeffs_fast(ux[k] - fx[k], wx);
      const bool upy = fy[k] != fby, upx = (fx[k] - (float)k) != fbx;
      float rowv[5];
#pragma unroll
      for (int i = 0; i < 5; ++i)
        rowv[i] = dot5(wx, upx, win.v[i][k], win.v[i][k + 1], win.v[i][k + 2], win.v[i][k + 3],
                       win.v[i][k + 4]);
      const float o = dot5(wy, upy, rowv[0], rowv[1], rowv[2], rowv[3], rowv[4]);
      res[k] = inside[k] ? o : 0.f;
    }
  } else {
#pragma unroll
    for (int k = 0; k < WARP_PX; ++k) {
      float wy[4], wx[4];
      cubic_coeffs_fast(uy[k] - fy[k], wy);
      cubic_coeffs_fast(ux[k] - fx[k], wx);
      // border padding: clip each tap coordinate (ATen clip_coordinates), in float first
      // so that huge coordinates cannot overflow the int conversion
      float rowv[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const float ty = fminf(fmaxf(fy[k] + (float)(i - 1), 0.f), fh - 1.f);
        const float* r = fr + (int64_t)(int)ty * w;
        float t4[4];
#pragma unroll
        for (int j = 0; j < 4; ++j)
          t4[j] = r[(int)fminf(fmaxf(fx[k] + (float)(j - 1), 0.f), fw - 1.f)];
        rowv[i] = dot4(make_float4(wx[0], wx[1], wx[2], wx[3]), t4[0], t4[1], t4[2], t4[3]);
      }
      const float o = dot4(make_float4(wy[0], wy[1], wy[2], wy[3]), rowv[0], rowv[1], rowv[2], rowv[3]);
      res[k] = inside[k] ? o : 0.f;
    }
  }
}

__device__ __forceinline__ void load_etab4(const float* E, int64_t rowstride, const int4 yt, int x0,
                                           int w, float4 out[4]) {
  const int rows[4] = {yt.x, yt.y, yt.z, yt.w};
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const float* p = E + (int64_t)rows[i] * rowstride + x0;
    if (x0 + 3 < w && ((rowstride & 3) == 0)) {
      out[i] = *reinterpret_cast<const float4*>(p);
    } else {
      out[i].x = p[0];
      out[i].y = x0 + 1 < w ? p[1] : 0.f;
      out[i].z = x0 + 2 < w ? p[2] : 0.f;
      out[i].w = x0 + 3 < w ? p[3] : 0.f;
    }
  }
}

template <bool WRITE_FRAMES, bool WRITE_SUM, bool UNIT_PS>
__global__ __launch_bounds__(WARP_TX* WARP_TY) void warp_main(WarpArgs a) {
  // XCD-aware tile order: blocks b, b+8, b+16.. share an XCD (round-robin dispatch);
  // give each XCD a contiguous band of tile rows so vertical halos hit its own L2.
  const int nt = a.tiles_x * a.tiles_y;
  const int b = blockIdx.x;
  int tile = b;
  if ((nt & 7) == 0) tile = (b & 7) * (nt >> 3) + (b >> 3);
  const int tyi = tile / a.tiles_x, txi = tile - tyi * a.tiles_x;
  const int x0 = txi * (WARP_TX * WARP_PX) + threadIdx.x * WARP_PX;
  const int ya = tyi * (WARP_TY * WARP_ROWS) + threadIdx.y * WARP_ROWS;
  const int h = a.h, w = a.w;
  if (ya >= h || x0 >= w) return;
  const int64_t hw = (int64_t)h * w;
  const bool two = (ya + 1 < h);
  const int yb = two ? ya + 1 : ya;
  const int4 yta = *reinterpret_cast<const int4*>(a.ytap + 4 * ya);
  const float4 yca = *reinterpret_cast<const float4*>(a.ycoef + 4 * ya);
  const int4 ytb = *reinterpret_cast<const int4*>(a.ytap + 4 * yb);
  const float4 ycb = *reinterpret_cast<const float4*>(a.ycoef + 4 * yb);
  const bool same = (yta.x == ytb.x) && (yta.y == ytb.y) && (yta.z == ytb.z) && (yta.w == ytb.w);
  const bool full = (x0 + WARP_PX <= w);
  float acc[WARP_ROWS][WARP_PX] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};

  for (int f = 0; f < a.nframes; ++f) {
    const float* fr = a.frames + (int64_t)f * hw;
    const float* E = a.etab + (int64_t)f * 2 * a.GH * w;
    float4 Ey[4], Ex[4];
    load_etab4(E, w, yta, x0, w, Ey);
    load_etab4(E + (int64_t)a.GH * w, w, yta, x0, w, Ex);
    TapWindow win;
    win.valid = false;
    win.by = win.bx = 0;
    float res[WARP_ROWS][WARP_PX];
    warp_row<UNIT_PS>(a, fr, ya, x0, yca, Ey, Ex, win, res[0]);
    if (two) {
      if (!same) {
        load_etab4(E, w, ytb, x0, w, Ey);
        load_etab4(E + (int64_t)a.GH * w, w, ytb, x0, w, Ex);
      }
      warp_row<UNIT_PS>(a, fr, yb, x0, ycb, Ey, Ex, win, res[1]);
    }
#pragma unroll
    for (int r = 0; r < WARP_ROWS; ++r) {
      if (r == 1 && !two) break;
      if (WRITE_FRAMES) {
        float* o = a.out_frames + (int64_t)f * hw + (int64_t)(ya + r) * w + x0;
        if (full && ((((uintptr_t)o) & 15) == 0)) {
          *reinterpret_cast<float4*>(o) = make_float4(res[r][0], res[r][1], res[r][2], res[r][3]);
        } else {
          for (int k = 0; k < WARP_PX && x0 + k < w; ++k) o[k] = res[r][k];
        }
      }
      if (WRITE_SUM) {
#pragma unroll
        for (int k = 0; k < WARP_PX; ++k) acc[r][k] += res[r][k];
      }
    }
  }
  if (WRITE_SUM) {
    for (int r = 0; r < WARP_ROWS; ++r) {
      if (r == 1 && !two) break;
      float* o = a.out_sum + (int64_t)(ya + r) * w + x0;
      for (int k = 0; k < WARP_PX && x0 + k < w; ++k) o[k] = acc[r][k];  // this thread owns the pixel for all frames
    }
  }
}

// ------------------------------------------------------------------ rigid warp
// A (2,nt,1,1) field gives every frame one shift (sy, sx) [px].  The coordinate chain
// of sample_image_2d then depends on y alone (rows) and x alone (columns), so the
// bicubic resample is a separable correlation whose 4 taps per axis sit at
// floor(u)-1..floor(u)+2.  floor(u(p)) - p takes at most two adjacent values along an
// axis (u = p + s up to fp32 rounding), so with S = min(floor(u(p)) - p) every output
// uses the 5 input samples p+S-1 .. p+S+3 with the 4 weights placed at offset
// d = floor(u(p)) - p - S in {0,1} (the fifth weight is an exact zero): same products,
// same summation order as the 4-tap form, but a perfectly regular access pattern.
// Rows/columns whose coordinate leaves [0,n-1] get all-zero weights (the reference
// zeroes those samples).  The reference's own per-pixel shift is the bicubic upsample
// of a constant lattice, i.e. s*(1 +- ~2e-6); here s is used as is.
#define RIGID_LANES 64
#define RIGID_WAVES 4
#define RIGID_ROWS 8                                   // output rows per wave
#ifndef RIGID_MINW
#define RIGID_MINW 2
#endif
#define RIGID_TROWS (RIGID_WAVES * RIGID_ROWS + 4)     // input rows per tile (36)
#define RIGID_QUADS (RIGID_LANES + 4)                  // float4 columns per tile row (68)
#define RIGID_PLANE (RIGID_QUADS + 1)                  // plane stride in floats (odd: no conflicts)
#define RIGID_RSTRIDE (4 * RIGID_PLANE)                // LDS floats per tile row
#define RIGID_NQ (RIGID_TROWS * RIGID_QUADS)           // quads per tile (2448)
#define RIGID_QPT ((RIGID_NQ + 255) / 256)             // quads per thread (10)

// pass 0: S[f][axis] = min_p floor(u(p)) - p.  One workgroup per (frame, axis): no atomics,
// no pre-set of S (80 words fought over by 1280 workgroups cost 20 us in atomics alone).
__global__ __launch_bounds__(256) void rigid_base(const float* __restrict__ shifts, int nframes, int h,
                                                  int w, int* __restrict__ S) {
  const int f = blockIdx.x, axis = blockIdx.y;
  const int n = axis == 0 ? h : w;
  const float s = shifts[2 * f + axis];
  // clamp the (finite) offset so absurd shifts cannot overflow
  const float lim = 3.0f * (float)n + 16.f;
  int di = 0x7fffffff;
  for (int p = threadIdx.x; p < n; p += 256) {
    const float u = grid_chain((float)p + s, (float)n);
    const float d = floorf(u) - (float)p;
    di = min(di, (int)fminf(fmaxf(d, -lim), lim));
  }
  for (int off = 32; off > 0; off >>= 1) di = min(di, __shfl_xor(di, off));
  __shared__ int part[4];
  if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = di;
  __syncthreads();
  if (threadIdx.x == 0) S[2 * f + axis] = min(min(part[0], part[1]), min(part[2], part[3]));
}

// pass 1: W[f][axis][k][p], k = 0..4
__global__ void rigid_weights(const float* __restrict__ shifts, int nframes, int h, int w,
                              const int* __restrict__ S, float* __restrict__ Wy,
                              float* __restrict__ Wx) {
  const int f = blockIdx.y, axis = blockIdx.z;
  const int n = axis == 0 ? h : w;
  const int p = blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= n) return;
  const float s = shifts[2 * f + axis];
  const float c = (float)p + s;
  const bool inside = (c >= 0.f) && (c <= (float)n - 1.f);
  const float u = grid_chain(c, (float)n);
  const float fl = floorf(u);
  float wt[4];
  cubic_coeffs_fast(u - fl, wt);
  const float lim = 3.0f * (float)n + 16.f;
  const int d = (int)fminf(fmaxf(fl - (float)p, -lim), lim) - S[2 * f + axis];
  float out[5] = {0.f, 0.f, 0.f, 0.f, 0.f};
  if (inside && d >= 0 && d <= 1) {
    for (int k = 0; k < 4; ++k) out[k + d] = wt[k];
  }
  if (axis == 0) {
    float* W = Wy + ((int64_t)f * n + p) * 5;  // [f][y][5]: a strip's weights are contiguous
    for (int k = 0; k < 5; ++k) W[k] = out[k];
  } else {
    float* W = Wx + (int64_t)f * 5 * n;  // [f][5][x]: float4 per tap for 4 adjacent columns
    for (int k = 0; k < 5; ++k) W[(int64_t)k * n + p] = out[k];
  }
}

// pass 0 for the movie pipeline, everything between the peak search and the weight tables in ONE launch:
// the pipeline's tail used to be six launches of a few microseconds each (shifts * pixel_spacing, a
// contiguous copy, spline_lattice_kernel, another copy, / pixel_spacing, rigid_base), all on the
// estimator's critical chain and each waiting for a wave slot under the previous movie's warp.  One
// workgroup per (frame, axis): the frame's lattice value = the (2,t,1,1) field's spline in time at
// t_f = f / (t - 1) -- spline_lattice_kernel's arithmetic in spline_lattice_kernel's order, lattice point
// (0, 0), so the result is bit for bit what the generic route gives --, then rigid_base's reduction.
//   field[axis][f] = shifts[f][axis] * ps;  shifts_px[f][axis] = lattice / ps;  S[f][axis] as rigid_base
__global__ __launch_bounds__(256) void rigid_tail(const float* __restrict__ shifts, float ps, const int* __restrict__ idx_t,
                                                  const float* __restrict__ w_t, const float* __restrict__ w_y,
                                                  const float* __restrict__ w_x, int nframes, int h, int w,
                                                  float* __restrict__ field, float* __restrict__ shifts_px,
                                                  int* __restrict__ S) {
  const int f = blockIdx.x, axis = blockIdx.y;
  float vt = 0.f;
  for (int kt = 0; kt < 4; ++kt) {
    const float d = shifts[2 * idx_t[4 * f + kt] + axis] * ps;  // the field's node value (deformation_field_utils.py:129-162)
    float vy = 0.f;
    for (int ky = 0; ky < 4; ++ky) {
      float vx = 0.f;
      for (int kx = 0; kx < 4; ++kx) vx += d * w_x[kx];
      vy += vx * w_y[ky];
    }
    vt += vy * w_t[4 * f + kt];
  }
  const float s = vt / ps;
  if (threadIdx.x == 0) {
    field[axis * nframes + f] = shifts[2 * f + axis] * ps;
    shifts_px[2 * f + axis] = s;
  }
  const int n = axis == 0 ? h : w;
  const float lim = 3.0f * (float)n + 16.f;
  int di = 0x7fffffff;
  for (int p = threadIdx.x; p < n; p += 256) {
    const float u = grid_chain((float)p + s, (float)n);
    const float d = floorf(u) - (float)p;
    di = min(di, (int)fminf(fmaxf(d, -lim), lim));
  }
  for (int off = 32; off > 0; off >>= 1) di = min(di, __shfl_xor(di, off));
  __shared__ int part[4];
  if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = di;
  __syncthreads();
  if (threadIdx.x == 0) S[2 * f + axis] = min(min(part[0], part[1]), min(part[2], part[3]));
}

struct RigidArgs {
  const float* frames;
  int nframes, h, w;
  const int* S;     // [f][2]
  const float* Wy;  // [f][h][5]
  const float* Wx;  // [f][5][w]
  float* out_frames;
  float* out_sum;
  int tiles_x, tiles_y;
  int frames_in_grid;  // c > 0: blockIdx.y selects a chunk of c frames (no fused sum); 0: all frames in-block
};

#pragma clang fp contract(fast)
// Workgroup = 4 waves = tile of 256 x 32 output pixels.  Per frame the tile's 36 x 272
// input window (row/column indices clipped to the image = border padding) is fetched
// ONCE with 16-byte loads and parked in LDS de-interleaved by (column mod 4): the
// window is misaligned by m = (x_tile + Sx - 1) mod 4 floats, and with four planes lane l
// reads tap j at plane (m+j)&3, index l + ((m+j)>>2): consecutive lanes, consecutive
// banks.  Loads for frame f+1 are in flight (registers) while frame f is computed.
template <bool WRITE_FRAMES, bool WRITE_SUM>
__global__ __launch_bounds__(RIGID_LANES* RIGID_WAVES, RIGID_MINW) void warp_rigid(RigidArgs a) {
  __shared__ float tileS[RIGID_TROWS * RIGID_RSTRIDE];
  const int nt = a.tiles_x * a.tiles_y;
  const int b = blockIdx.x;
  int tile = b;
  if ((nt & 7) == 0) tile = (b & 7) * (nt >> 3) + (b >> 3);  // one band of tile rows per XCD
  const int tyi = tile / a.tiles_x, txi = tile - tyi * a.tiles_x;
  const int h = a.h, w = a.w;
  const int lane = threadIdx.x;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.y);
  const int tid = wave * RIGID_LANES + lane;
  const int xt = txi * (RIGID_LANES * 4);
  const int yt = tyi * (RIGID_WAVES * RIGID_ROWS);
  const int x0 = xt + lane * 4;
  const int y0 = yt + wave * RIGID_ROWS;
  const int64_t hw = (int64_t)h * w;
  const bool full = (x0 + 4 <= w) && ((w & 3) == 0);
  const bool wq = ((w & 3) == 0);
  float acc[RIGID_ROWS][4];
#pragma unroll
  for (int r = 0; r < RIGID_ROWS; ++r)
#pragma unroll
    for (int k = 0; k < 4; ++k) acc[r][k] = 0.f;

  const int f_lo = a.frames_in_grid ? (int)blockIdx.y * a.frames_in_grid : 0;
  const int f_hi = a.frames_in_grid ? min(f_lo + a.frames_in_grid, a.nframes) : a.nframes;

  float4 pre[RIGID_QPT];
  auto fetch = [&](int f) {  // issue the tile loads of frame f into `pre`
    const float* fr = a.frames + (int64_t)f * hw;
    const int Sy = a.S[2 * f], Sx = a.S[2 * f + 1];
    const int cxt = xt + Sx - 1;
    const int ax = cxt & ~3;  // aligned-down first column (two's complement: works for cxt < 0)
    const bool fast = wq && ax >= 0 && ax + 4 * RIGID_QUADS <= w;
#pragma unroll
    for (int i = 0; i < RIGID_QPT; ++i) {
      const int q = tid + i * 256;
      if (q < RIGID_NQ) {
        const int tr = q / RIGID_QUADS, qc = q - tr * RIGID_QUADS;
        int r = yt + Sy - 1 + tr;
        r = r < 0 ? 0 : (r > h - 1 ? h - 1 : r);
        const float* row = fr + (int64_t)r * w;
        const int c = ax + 4 * qc;
        if (fast) {
          pre[i] = *reinterpret_cast<const float4*>(row + c);
        } else {
          float e[4];
#pragma unroll
          for (int k = 0; k < 4; ++k) {
            int cc = c + k;
            cc = cc < 0 ? 0 : (cc > w - 1 ? w - 1 : cc);
            e[k] = row[cc];
          }
          pre[i] = make_float4(e[0], e[1], e[2], e[3]);
        }
      }
    }
  };
  auto park = [&]() {  // registers -> LDS planes
#pragma unroll
    for (int i = 0; i < RIGID_QPT; ++i) {
      const int q = tid + i * 256;
      if (q < RIGID_NQ) {
        const int tr = q / RIGID_QUADS, qc = q - tr * RIGID_QUADS;
        float* d = tileS + tr * RIGID_RSTRIDE + qc;
        d[0] = pre[i].x;
        d[RIGID_PLANE] = pre[i].y;
        d[2 * RIGID_PLANE] = pre[i].z;
        d[3 * RIGID_PLANE] = pre[i].w;
      }
    }
  };

  fetch(f_lo);
  park();
  __syncthreads();
  for (int f = f_lo; f < f_hi; ++f) {
    if (f + 1 < f_hi) fetch(f + 1);
    const int Sx = a.S[2 * f + 1];
    const int m = (xt + Sx - 1) & 3;
    // row weights of this wave's strip: Wy[f][y][5] -> 40 consecutive floats, one per lane
    float wyv = 0.f;
    {
      const int64_t idx = (int64_t)y0 * 5 + lane;
      if (lane < 5 * RIGID_ROWS && idx < (int64_t)h * 5) wyv = a.Wy[(int64_t)f * 5 * h + idx];
    }
    float wx[5][4];
    const float* Wx = a.Wx + (int64_t)f * 5 * w + x0;
#pragma unroll
    for (int j = 0; j < 5; ++j) {
      if (full) {
        const float4 t = *reinterpret_cast<const float4*>(Wx + (int64_t)j * w);
        wx[j][0] = t.x; wx[j][1] = t.y; wx[j][2] = t.z; wx[j][3] = t.w;
      } else {
#pragma unroll
        for (int k = 0; k < 4; ++k) wx[j][k] = (x0 + k < w) ? Wx[(int64_t)j * w + k] : 0.f;
      }
    }
    // per-tap LDS offsets (wave-uniform): plane (m+j)&3, index lane + ((m+j)>>2)
    int toff[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) toff[j] = ((m + j) & 3) * RIGID_PLANE + ((m + j) >> 2);
    const float* wrow = tileS + (wave * RIGID_ROWS) * RIGID_RSTRIDE + lane;
    float H[5][4];
#pragma unroll
    for (int rr = 0; rr < RIGID_ROWS + 4; ++rr) {
      // keep at most two rows of LDS reads in flight: without this the scheduler hoists
      // all 96 reads and the kernel needs > 240 VGPRs
  #ifndef RIGID_SB
#define RIGID_SB 2
#endif
    if (RIGID_SB > 0 && (rr % RIGID_SB) == 0) __builtin_amdgcn_sched_barrier(0);
      const float* src = wrow + rr * RIGID_RSTRIDE;
      float v[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) v[j] = src[toff[j]];
      float* Hn = H[rr % 5];
#pragma unroll
      for (int k = 0; k < 4; ++k)
        Hn[k] = (((wx[0][k] * v[k] + wx[1][k] * v[k + 1]) + wx[2][k] * v[k + 2]) +
                 wx[3][k] * v[k + 3]) + wx[4][k] * v[k + 4];
      if (rr >= 4) {
        const int ro = rr - 4;
        const int yo = y0 + ro;
        float wy[5];
#pragma unroll
        for (int i = 0; i < 5; ++i)
          wy[i] = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(wyv), ro * 5 + i));
        float o[4];
#pragma unroll
        for (int k = 0; k < 4; ++k)
          o[k] = (((wy[0] * H[(ro + 0) % 5][k] + wy[1] * H[(ro + 1) % 5][k]) +
                   wy[2] * H[(ro + 2) % 5][k]) + wy[3] * H[(ro + 3) % 5][k]) +
                 wy[4] * H[(ro + 4) % 5][k];
        if (yo < h && x0 < w) {
          if (WRITE_FRAMES) {
            float* dst = a.out_frames + (int64_t)f * hw + (int64_t)yo * w + x0;
            if (full) {
              *reinterpret_cast<float4*>(dst) = make_float4(o[0], o[1], o[2], o[3]);
            } else {
              for (int k = 0; k < 4 && x0 + k < w; ++k) dst[k] = o[k];
            }
          }
          if (WRITE_SUM) {
#pragma unroll
            for (int k = 0; k < 4; ++k) acc[ro][k] += o[k];
          }
        }
      }
    }
    __syncthreads();  // everyone is done reading the tile of frame f
    if (f + 1 < f_hi) {
      park();
      __syncthreads();
    }
  }
  if (WRITE_SUM && x0 < w) {
#pragma unroll
    for (int ro = 0; ro < RIGID_ROWS; ++ro) {
      const int yo = y0 + ro;
      if (yo < h) {
        float* dst = a.out_sum + (int64_t)yo * w + x0;
        for (int k = 0; k < 4 && x0 + k < w; ++k) dst[k] = acc[ro][k];  // one block owns the tile's sum
      }
    }
  }
}
#pragma clang fp contract(off)

#pragma clang fp contract(fast)
// ------------------------------------------------------------------ rigid warp, LDS-DMA
// Same mathematics as warp_rigid; the 36 x 272 input window of a tile goes HBM -> LDS by
// `global_load_lds_dwordx4` (no VGPR staging, row-major image).  The global side of that
// DMA only needs 4-byte alignment, so the window starts exactly at column x_tile + Sx - 1:
// a lane's 8-float window is two aligned 16-byte LDS reads whatever the shift, and there
// is ONE strip body (four misalignment specialisations of it were 24 KB of straight-line
// code, which thrashed the instruction cache once blocks of different frames shared a CU,
// and cost 16 more VGPRs: the fused-sum kernel now fits 4 workgroups per CU).
// Default: single buffer, 4 workgroups per CU cover each other's DMA latency.  NBUF = 2
// double-buffers inside the workgroup (2 per CU) and leaves the row stores of frame f in
// flight under frame f+1.  Columns outside the image (border padding = clipped tap
// coordinate) are re-fetched element-wise for edge tiles only.
// Requires w % 4 == 0 and 16-byte aligned frames (host checks; else warp_rigid).
#define RD_QUADS_PAD (((RIGID_NQ + 63) / 64) * 64)  // DMA granule: 64 lanes x 16 B
typedef __attribute__((address_space(3))) void* lds_vptr;

// a0 b0 + a1 b1 + ... + a4 b4 as one product and four fused multiply-adds in this order: the fp32 and
// the fp16 strip bodies then round identically (left to the contraction pass the two bodies fused
// different products and differed in the last bit for 5 % of the pixels)
__device__ __forceinline__ float rigid_dot5(float a0, float b0, float a1, float b1, float a2, float b2, float a3,
                                            float b3, float a4, float b4) {
  float r = a0 * b0;
  r = __builtin_fmaf(a1, b1, r);
  r = __builtin_fmaf(a2, b2, r);
  r = __builtin_fmaf(a3, b3, r);
  return __builtin_fmaf(a4, b4, r);
}

#ifdef MC_RIGID_STAMP
__device__ unsigned long long g_rigid_stamps[8];
#define RSTAMP(v) do { __builtin_amdgcn_sched_barrier(0); v = __builtin_amdgcn_s_memtime(); __builtin_amdgcn_s_waitcnt(0xC07F); __builtin_amdgcn_sched_barrier(0); } while (0)
#else
#define RSTAMP(v) do { } while (0)
#endif
typedef float rigid_f4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void rigid_store4(float* p, float a, float b, float c, float d) {
  // corrected frames are written once and never read again by this kernel: non-temporal stores
  // (fused launch 1.25 -> 1.15 ms at 40 x 4096^2, frames only 0.98 -> 0.95; scripts/ubench/stream_copy.hip
  // shows the same 5 % on a plain tiled copy)
  const rigid_f4 v = {a, b, c, d};
#ifdef MC_RIGID_PLAIN_STORES
  *reinterpret_cast<rigid_f4*>(p) = v;
#else
  __builtin_nontemporal_store(v, reinterpret_cast<rigid_f4*>(p));
#endif
}

template <bool WRITE_FRAMES, bool WRITE_SUM, bool FULL, int QUADS>
__device__ __forceinline__ void rigid_strip_dma(const RigidArgs& a, const float4* wrow, int f,
                                                int y0, int x0, float wyv,
                                                const float (&wx)[5][4],
                                                float (&acc)[RIGID_ROWS][4]) {
  const int h = a.h, w = a.w;
  // FULL: the whole tile lies inside the image -> no per-row predicates, the strip is
  // one basic block and the scheduler can run the LDS reads ahead of the arithmetic
  float* orow = WRITE_FRAMES ? a.out_frames + (int64_t)f * h * w + (int64_t)y0 * w + x0 : nullptr;
  float H[5][4];
#pragma unroll
  for (int rr = 0; rr < RIGID_ROWS + 4; ++rr) {
#ifndef RIGID_SB
#define RIGID_SB 2
#endif
    if (RIGID_SB > 0 && (rr % (RIGID_SB > 0 ? RIGID_SB : 1)) == 0) __builtin_amdgcn_sched_barrier(0);
    const float4 q0 = wrow[rr * QUADS], q1 = wrow[rr * QUADS + 1];
    const float e[8] = {q0.x, q0.y, q0.z, q0.w, q1.x, q1.y, q1.z, q1.w};
    float* Hn = H[rr % 5];
#pragma unroll
    for (int k = 0; k < 4; ++k)
      Hn[k] = rigid_dot5(wx[0][k], e[k], wx[1][k], e[k + 1], wx[2][k], e[k + 2], wx[3][k], e[k + 3], wx[4][k], e[k + 4]);
    if (rr >= 4) {
      const int ro = rr - 4;
      float wy[5];
#pragma unroll
      for (int i = 0; i < 5; ++i)
        wy[i] = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(wyv), ro * 5 + i));
      float o[4];
#pragma unroll
      for (int k = 0; k < 4; ++k)
        o[k] = rigid_dot5(wy[0], H[(ro + 0) % 5][k], wy[1], H[(ro + 1) % 5][k], wy[2], H[(ro + 2) % 5][k], wy[3],
                          H[(ro + 3) % 5][k], wy[4], H[(ro + 4) % 5][k]);
      if (FULL || (y0 + ro < h && x0 < w)) {
        if (WRITE_FRAMES) rigid_store4(orow + (int64_t)ro * w, o[0], o[1], o[2], o[3]);
        if (WRITE_SUM) {
#pragma unroll
          for (int k = 0; k < 4; ++k) acc[ro][k] += o[k];
        }
      }
    }
  }
}

// Tile geometry: WX waves side by side (256 columns each) x WY waves down (8 rows each), i.e. a
// tile of 256 WX x 8 WY output pixels per workgroup of 64 WX WY threads.  (1, 4) is the 256 x 32
// tile of round 1; wider tiles make every row piece a workgroup touches longer (1 KiB per 256
// columns: fewer partial 128-byte lines at the two ends, longer runs inside a DRAM page).
#ifndef RIGID_DMA_MINW
#define RIGID_DMA_MINW 4
#endif
#if !defined(RIGID_DMA_VGPRS) || defined(MC_EXPERIMENTS)
#define RIGID_DMA_VGPR_ATTR
#else
#define RIGID_DMA_VGPR_ATTR __attribute__((amdgpu_num_vgpr(RIGID_DMA_VGPRS)))
#endif
template <bool WRITE_FRAMES, bool WRITE_SUM, int NBUF, int WX, int WY>
__global__ __launch_bounds__(RIGID_LANES* WX* WY, NBUF == 1 ? RIGID_DMA_MINW : 2)  // 16 (8) waves per CU whatever the tile
RIGID_DMA_VGPR_ATTR void warp_rigid_dma(RigidArgs a) {
  constexpr int NWAVES = WX * WY;
  constexpr int TROWS = WY * RIGID_ROWS + 4;       // input rows per tile
  constexpr int QUADS = WX * RIGID_LANES + 4;      // float4 columns per tile row
  constexpr int NQ = TROWS * QUADS;
  constexpr int QUADS_PAD = ((NQ + 63) / 64) * 64;  // DMA granule: 64 lanes x 16 B
  extern __shared__ __attribute__((aligned(16))) char smem_rd[];
  float4* const b0 = reinterpret_cast<float4*>(smem_rd);
  float4* const b1 = NBUF == 2 ? b0 + QUADS_PAD : b0;
  const int nt = a.tiles_x * a.tiles_y;
  const int b = blockIdx.x;
  int tile = b;
  if ((nt & 7) == 0) tile = (b & 7) * (nt >> 3) + (b >> 3);  // one band of tile rows per XCD
  const int tyi = tile / a.tiles_x, txi = tile - tyi * a.tiles_x;
  const int h = a.h, w = a.w;
  const int lane = threadIdx.x;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.y);
  const int wvx = wave % WX, wvy = wave / WX;
  const int tid = wave * RIGID_LANES + lane;
  const int xt = txi * (RIGID_LANES * 4 * WX);
  const int yt = tyi * (WY * RIGID_ROWS);
  const int x0 = xt + wvx * (RIGID_LANES * 4) + lane * 4;
  const int y0 = yt + wvy * RIGID_ROWS;
  const int64_t hw = (int64_t)h * w;
  float acc[RIGID_ROWS][4];
#pragma unroll
  for (int r = 0; r < RIGID_ROWS; ++r)
#pragma unroll
    for (int k = 0; k < 4; ++k) acc[r][k] = 0.f;
  const bool full_tile = yt + WY * RIGID_ROWS <= h && xt + RIGID_LANES * 4 * WX <= w;
  const int f_lo = a.frames_in_grid ? (int)blockIdx.y * a.frames_in_grid : 0;
  const int f_hi = a.frames_in_grid ? min(f_lo + a.frames_in_grid, a.nframes) : a.nframes;

  // DMA of one frame's window into `dst`: granule i = quads [64 i, 64 i + 64)
  auto dma = [&](int f, float4* dst) {
    const float* fr = a.frames + (int64_t)f * hw;
    const int Sy = a.S[2 * f], Sx = a.S[2 * f + 1];
    const int ax = xt + Sx - 1;  // any multiple of 4 BYTES: the global side of the DMA needs no more
    for (int i = wave; i < QUADS_PAD / 64; i += NWAVES) {
      int q = i * 64 + lane;
      q = q < NQ ? q : NQ - 1;  // tail lanes re-load the last quad into the pad
      const int tr = q / QUADS, qc = q - tr * QUADS;
      int r = yt + Sy - 1 + tr;
      r = r < 0 ? 0 : (r > h - 1 ? h - 1 : r);
      int c = ax + 4 * qc;
      c = c < 0 ? 0 : (c > w - 4 ? w - 4 : c);  // whole quads inside the row; clamped ones are patched
      __builtin_amdgcn_global_load_lds(fr + (int64_t)r * w + c, (lds_vptr)(dst + i * 64), 16, 0, 0);
    }
  };
  // border padding for edge tiles: a quad whose 4 columns are not all inside the row was
  // DMA'd from a clamped address and holds the wrong columns; its elements are re-fetched
  // one by one at their clipped column (a handful of quads per row, edge tiles only)
  auto patch = [&](int f, float4* t4) {
    const int Sy = a.S[2 * f], Sx = a.S[2 * f + 1];
    const int ax = xt + Sx - 1;
    if (ax >= 0 && ax + 4 * QUADS <= w) return false;
    const float* fr = a.frames + (int64_t)f * hw;
    float* t = reinterpret_cast<float*>(t4);
    for (int q = tid; q < NQ; q += RIGID_LANES * NWAVES) {
      const int tr = q / QUADS, qc = q - tr * QUADS;
      const int s0 = ax + 4 * qc;
      if (s0 >= 0 && s0 <= w - 4) continue;
      int r = yt + Sy - 1 + tr;
      r = r < 0 ? 0 : (r > h - 1 ? h - 1 : r);
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        int c = s0 + e;
        c = c < 0 ? 0 : (c > w - 1 ? w - 1 : c);
        t[4 * q + e] = fr[(int64_t)r * w + c];
      }
    }
    return true;
  };

  float wx[5][4], wxn[5][4];
  float wyv = 0.f, wyvn = 0.f;
  auto load_weights = [&](int f, float (&W5)[5][4], float& Wv) {
    Wv = 0.f;
    const int64_t idx = (int64_t)y0 * 5 + lane;
    if (lane < 5 * RIGID_ROWS && idx < (int64_t)h * 5) Wv = a.Wy[(int64_t)f * 5 * h + idx];
    const float* Wx = a.Wx + (int64_t)f * 5 * w + x0;
#pragma unroll
    for (int j = 0; j < 5; ++j) {
      float4 t = make_float4(0.f, 0.f, 0.f, 0.f);
      if (x0 < w) t = *reinterpret_cast<const float4*>(Wx + (int64_t)j * w);
      W5[j][0] = t.x; W5[j][1] = t.y; W5[j][2] = t.z; W5[j][3] = t.w;
    }
  };
  const int strip = (wvy * RIGID_ROWS) * QUADS + wvx * RIGID_LANES + lane;  // this lane's first quad

  load_weights(f_lo, wx, wyv);
  dma(f_lo, b0);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  if (patch(f_lo, b0)) __syncthreads();
  int cur = 0;
#ifdef MC_RIGID_STAMP
  unsigned long long st_c = 0, st_b1 = 0, st_i = 0, st_w = 0, st_b2 = 0;
#endif
  for (int f = f_lo; f < f_hi; ++f) {
#ifdef MC_RIGID_STAMP
    unsigned long long T0, T1, T2, T3, T4, T5;
    RSTAMP(T0);
#endif
    if (NBUF == 2 && f + 1 < f_hi) {
      // both kinds of loads for frame f+1 go out BEFORE this frame's stores: the vector-memory
      // counter retires in order, so waiting for "all but the 8 newest" operations below waits for
      // the loads and leaves this frame's 8 row stores draining under the next frame
      dma(f + 1, cur ? b0 : b1);
      load_weights(f + 1, wxn, wyvn);
    }
    const float4* t = (cur ? b1 : b0) + strip;
    if (full_tile) rigid_strip_dma<WRITE_FRAMES, WRITE_SUM, true, QUADS>(a, t, f, y0, x0, wyv, wx, acc);
    else rigid_strip_dma<WRITE_FRAMES, WRITE_SUM, false, QUADS>(a, t, f, y0, x0, wyv, wx, acc);
    RSTAMP(T1);
    if (f + 1 < f_hi) {
      if constexpr (NBUF == 1) {
        // single buffer, several workgroups per CU: other workgroups cover this one's latency,
        // so nothing is double-buffered here (registers are the scarce resource)
        __syncthreads();  // everyone must be done reading before the tile is refilled
        RSTAMP(T2);
        dma(f + 1, b0);
        load_weights(f + 1, wx, wyv);
        RSTAMP(T3);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        RSTAMP(T4);
        __syncthreads();
        RSTAMP(T5);
#ifdef MC_RIGID_STAMP
        st_c += T1 - T0; st_b1 += T2 - T1; st_i += T3 - T2; st_w += T4 - T3; st_b2 += T5 - T4;
#endif
        if (patch(f + 1, b0)) __syncthreads();
      } else {
        RSTAMP(T2);
        if (WRITE_FRAMES && full_tile) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");  // RIGID_ROWS stores
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        RSTAMP(T4);
        __syncthreads();  // DMA of f+1 landed for every wave; everyone is done with buf[cur]
        RSTAMP(T5);
#ifdef MC_RIGID_STAMP
        st_c += T1 - T0; st_w += T4 - T2; st_b2 += T5 - T4;
#endif
        cur ^= 1;
        if (patch(f + 1, cur ? b1 : b0)) __syncthreads();
#pragma unroll
        for (int j = 0; j < 5; ++j)
#pragma unroll
          for (int k = 0; k < 4; ++k) wx[j][k] = wxn[j][k];
        wyv = wyvn;
      }
    }
  }
#ifdef MC_RIGID_STAMP
  if (lane == 0) {
    atomicAdd(&g_rigid_stamps[0], st_c); atomicAdd(&g_rigid_stamps[1], st_b1); atomicAdd(&g_rigid_stamps[2], st_i);
    atomicAdd(&g_rigid_stamps[3], st_w); atomicAdd(&g_rigid_stamps[4], st_b2); atomicAdd(&g_rigid_stamps[5], 1ull);
  }
#endif
  if (WRITE_SUM && x0 < w) {
#pragma unroll
    for (int ro = 0; ro < RIGID_ROWS; ++ro) {
      const int yo = y0 + ro;
      if (yo < h) {
        // one block owns its tile's sum over all frames: a plain store (no zero fill, no read-back);
        // w % 4 == 0 on this path, so the quad is 16-byte aligned whenever out_sum is
        float* dst = a.out_sum + (int64_t)yo * w + x0;
        if ((((uintptr_t)a.out_sum) & 15) == 0) {
          *reinterpret_cast<float4*>(dst) = make_float4(acc[ro][0], acc[ro][1], acc[ro][2], acc[ro][3]);
        } else {
#pragma unroll
          for (int k = 0; k < 4; ++k) dst[k] = acc[ro][k];
        }
      }
    }
  }
}

#ifdef MC_EXPERIMENTS
// ------------------------------------------------------------------ rigid warp, loader wave + compute waves
// EXPERIMENT (built only with -DMC_EXPERIMENTS, selected with MC_RIGID_LS=1; scripts/build_variant.sh).
// Round 3.  In-kernel stamps of warp_rigid_dma (MC_RIGID_STAMP) showed where a frame step goes: 46 % of
// a wave's cycles sit in the ISSUE of its LDS-DMA instructions (the vector-memory queue is full, the
// wave cannot do anything else while it waits for a slot), 14 % in pure arithmetic, 13 % in store
// back-pressure, the rest in waits and barriers -- and double-buffering inside those waves does not
// help, because the "asynchronous" DMA blocks the issuing wave all the same.  Here the two jobs are
// different waves of one workgroup per CU:
//   * ONE loader wave issues every LDS-DMA of the workgroup, one unit ahead, from per-lane byte offsets
//     it computed once (73 VGPRs: a loader has nothing else to keep) and a scalar base per unit, so an
//     interior unit costs it no vector arithmetic at all; it sits in the back-pressured issue so that
//     nobody else has to, and the CU's read path never runs dry;
//   * 8 compute waves (2 x 4, 256 columns x 8 rows each) only read LDS, compute and store: reads and
//     writes flow at the same time, which is what a copy needs to reach the chip's mixed ceiling
//     (scripts/ubench/stream_copy.hip: 6.2 TB/s for a plain copy against 5.5-5.7 read-only).
// A tile is 512 x 64 output pixels, processed per frame as two 32-row halves ("units") that alternate
// between two LDS windows; the four rows the halves share are fetched twice by the same CU within
// microseconds (an L2 hit), so the y-halo reaching the fabric is 68 / 64 instead of 36 / 32, and the
// 512 tiles of a 4096^2 frame are exactly two per CU.  One barrier per unit: when it releases, unit
// u + 1 has landed (the loader waited for its own DMAs) and unit u's window is free for unit u + 2.
// The weight rows of a frame reach LDS the same way (loaded during the previous frame's second unit,
// moved to registers at the start of the frame), so the compute waves issue no loads at all.
// Requires w % 4 == 0 and 16-byte aligned frames (host checks; else warp_rigid_dma / warp_rigid).
// MEASURED (40 x 4096^2, same box, non-temporal stores in both): bit-identical output; alone 1.039-1.051 ms
// against 1.054-1.058 for warp_rigid_dma; one loader wave is NOT enough (1.28 ms: vmcnt allows 63 DMAs in
// flight per wave), two give 1.11, four 1.13.  Stamps: the loader spends 80 % of a unit back-pressured in
// issue, the compute waves idle half of the time -- the launch is bound by what the memory system gives
// this access pattern (36 x 2 KB row pieces per window), not by anything a CU does.  Under the two-stream
// pipeline it LOSES (step 1.78-1.81 against 1.74-1.76 ms): a 10-wave workgroup holding all 160 KB of LDS
// leaves the estimator's column kernels (32 KB of LDS per workgroup) nowhere to run.  Not the default.
#define RLS_WX 2
#define RLS_WY 4
#define RLS_NC (RLS_WX * RLS_WY)                 // compute waves
#define RLS_TROWS (RLS_WY * RIGID_ROWS + 4)      // window rows per unit (36)
#define RLS_Q (RLS_WX * RIGID_LANES + 1)         // float4 columns per window row (129)
#define RLS_NQ (RLS_TROWS * RLS_Q)               // 4644
#define RLS_CHUNKS ((RLS_NQ + 63) / 64)          // 73 DMA instructions per unit
#define RLS_PADQ (RLS_CHUNKS * 64)
#define RLS_TW (RLS_WX * RIGID_LANES * 4)        // 512
#define RLS_TH (2 * RLS_WY * RIGID_ROWS)         // 64
#define RLS_LDS_BYTES ((2 * RLS_PADQ + 5 * RLS_TW / 4) * 16 + RLS_TH * 5 * 4)

template <bool WRITE_FRAMES, bool WRITE_SUM, int NLOAD>
__global__ __launch_bounds__(RIGID_LANES*(RLS_NC + NLOAD), 3) void warp_rigid_ls(RigidArgs a) {
  extern __shared__ __attribute__((aligned(16))) char smem_rd[];
  float4* const buf0 = reinterpret_cast<float4*>(smem_rd);
  float4* const buf1 = buf0 + RLS_PADQ;
  float4* const wxs = buf1 + RLS_PADQ;                        // [5][128] quads: Wx[f][j][xt .. xt + 512)
  float* const wys = reinterpret_cast<float*>(wxs + 5 * RLS_TW / 4);  // [64][5]: Wy[f][yt .. yt + 64)[5]
  const int nt = a.tiles_x * a.tiles_y;
  const int b = blockIdx.x;
  int tile = b;
  if ((nt & 7) == 0) tile = (b & 7) * (nt >> 3) + (b >> 3);  // one band of tile rows per XCD
  const int tyi = tile / a.tiles_x, txi = tile - tyi * a.tiles_x;
  const int h = a.h, w = a.w;
  const int lane = threadIdx.x;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.y);
  const int xt = txi * RLS_TW, yt = tyi * RLS_TH;
  const int64_t hw = (int64_t)h * w;
  const int nframes = a.nframes;

  if (wave >= RLS_NC) {
    // ---------------------------------------------------------------- loader wave(s): chunk i belongs to loader i % NLOAD
    const int lw = wave - RLS_NC;
    constexpr int MYCH = (RLS_CHUNKS + NLOAD - 1) / NLOAD;
    unsigned off[MYCH];  // byte offset of this lane's quad of chunk i from the window's first sample
#pragma unroll
    for (int m = 0; m < MYCH; ++m) {
      int q = (m * NLOAD + lw) * 64 + lane;
      q = q < RLS_NQ ? q : RLS_NQ - 1;  // tail lanes re-load the last quad into the pad
      const int tr = q / RLS_Q, qc = q - tr * RLS_Q;
      off[m] = (unsigned)(tr * w + 4 * qc) * 4u;
    }
    auto load_unit = [&](int f, int hf, float4* dst) {
      const float* fr = a.frames + (int64_t)f * hw;
      const int Sy = a.S[2 * f], Sx = a.S[2 * f + 1];
      const int ry = yt + hf * (RLS_TH / 2) + Sy - 1;  // image row of window row 0
      const int ax = xt + Sx - 1;                       // image column of window column 0
      const bool interior = ry >= 0 && ry + RLS_TROWS <= h && ax >= 0 && ax + 4 * RLS_Q <= w;
      if (interior) {
        const char* ub = reinterpret_cast<const char*>(fr + (int64_t)ry * w + ax);
#pragma unroll
        for (int m = 0; m < MYCH; ++m) {
          const int i = m * NLOAD + lw;
          if (i >= RLS_CHUNKS) break;
          // the 32-bit offset is made opaque at the point of use: otherwise its zero-extension is
          // hoisted out of the frame loop, the table becomes 146 registers and spills (and a reload
          // in the middle of the DMA stream waits for vmcnt(0))
          unsigned o = off[m];
          asm volatile("" : "+v"(o));
          __builtin_amdgcn_global_load_lds(reinterpret_cast<const float*>(ub + o), (lds_vptr)(dst + i * 64), 16, 0, 0);
        }
        return;
      }
      // edge unit: rows clipped to the image (border padding), quads kept whole inside the row; the
      // quads that had to move are patched element by element below
#pragma unroll 1
      for (int i = lw; i < RLS_CHUNKS; i += NLOAD) {
        int q = i * 64 + lane;
        q = q < RLS_NQ ? q : RLS_NQ - 1;
        const int tr = q / RLS_Q, qc = q - tr * RLS_Q;
        int r = ry + tr;
        r = r < 0 ? 0 : (r > h - 1 ? h - 1 : r);
        int c = ax + 4 * qc;
        c = c < 0 ? 0 : (c > w - 4 ? w - 4 : c);
        __builtin_amdgcn_global_load_lds(fr + (int64_t)r * w + c, (lds_vptr)(dst + i * 64), 16, 0, 0);
      }
      // columns outside the row: [0, nl) and [nr, RLS_Q) quads of every window row hold the wrong
      // samples; re-fetch their elements at the clipped column (own DMAs: one wave, one vmcnt)
      int nl = ax < 0 ? (-ax + 3) >> 2 : 0;
      nl = nl > RLS_Q ? RLS_Q : nl;
      int nr = w - 4 - ax >= 0 ? ((w - 4 - ax) >> 2) + 1 : 0;
      nr = nr > RLS_Q ? RLS_Q : (nr < nl ? nl : nr);
      const int nbad = nl + (RLS_Q - nr);
      if (nbad == 0) return;
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      float* t = reinterpret_cast<float*>(dst);
      const int per_row = 4 * nbad, items = RLS_TROWS * per_row;
      for (int it = lane; it < items; it += 64) {
        const int tr = it / per_row, k = it - tr * per_row;
        const int bq = k >> 2, e = k & 3;
        const int qc = bq < nl ? bq : nr + (bq - nl);
        if (NLOAD > 1 && ((tr * RLS_Q + qc) >> 6) % NLOAD != lw) continue;  // only quads this wave's DMAs wrote
        int r = ry + tr;
        r = r < 0 ? 0 : (r > h - 1 ? h - 1 : r);
        int c = ax + 4 * qc + e;
        c = c < 0 ? 0 : (c > w - 1 ? w - 1 : c);
        t[(tr * RLS_Q + qc) * 4 + e] = fr[(int64_t)r * w + c];
      }
    };
    // weight rows: uniform base per (frame, tap) + a 32-bit lane offset (columns past the row end are
    // clipped: their outputs are never stored)
    unsigned wxo[RLS_WX], wyo[RLS_TH * 5 / 64];
#pragma unroll
    for (int i = 0; i < RLS_WX; ++i) {
      int c = xt + 4 * (64 * i + lane);
      c = c > w - 4 ? w - 4 : c;
      wxo[i] = (unsigned)c * 4u;
    }
#pragma unroll
    for (int i = 0; i < RLS_TH * 5 / 64; ++i) {
      int idx = yt * 5 + i * 64 + lane;
      idx = idx > h * 5 - 1 ? h * 5 - 1 : idx;
      wyo[i] = (unsigned)idx * 4u;
    }
    auto load_weights = [&](int f) {
#pragma unroll
      for (int j = 0; j < 5; ++j) {
        const char* base = reinterpret_cast<const char*>(a.Wx + ((int64_t)f * 5 + j) * w);
#pragma unroll
        for (int i = 0; i < RLS_WX; ++i) {
          unsigned o = wxo[i];
          asm volatile("" : "+v"(o));
          __builtin_amdgcn_global_load_lds(reinterpret_cast<const float*>(base + o),
                                           (lds_vptr)(wxs + j * (RLS_TW / 4) + i * 64), 16, 0, 0);
        }
      }
      const char* base = reinterpret_cast<const char*>(a.Wy + (int64_t)f * 5 * h);
#pragma unroll
      for (int i = 0; i < RLS_TH * 5 / 64; ++i) {
        unsigned o = wyo[i];
        asm volatile("" : "+v"(o));
        __builtin_amdgcn_global_load_lds(reinterpret_cast<const float*>(base + o), (lds_vptr)(wys + i * 64), 4, 0, 0);
      }
    };
    load_unit(0, 0, buf0);
    if (lw == NLOAD - 1) load_weights(0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
#ifdef MC_RIGID_STAMP
    unsigned long long T0, T1, T2, T3, sli = 0, slw = 0, slb = 0;
#endif
    for (int f = 0; f < nframes; ++f) {
      RSTAMP(T0);
      load_unit(f, 1, buf1);  // under the first half of frame f
      RSTAMP(T1);
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      RSTAMP(T2);
      __syncthreads();
      RSTAMP(T3);
#ifdef MC_RIGID_STAMP
      sli += T1 - T0; slw += T2 - T1; slb += T3 - T2;
#endif
      if (f + 1 < nframes) {  // under the second half: the next frame's first window and its weights
        load_unit(f + 1, 0, buf0);
        if (lw == NLOAD - 1) load_weights(f + 1);
        RSTAMP(T1);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        RSTAMP(T2);
#ifdef MC_RIGID_STAMP
        sli += T1 - T3; slw += T2 - T1;
#endif
      }
      __syncthreads();
#ifdef MC_RIGID_STAMP
      RSTAMP(T3);
      slb += T3 - T2;
#endif
    }
#ifdef MC_RIGID_STAMP
    if (lane == 0) {
      atomicAdd(&g_rigid_stamps[2], sli); atomicAdd(&g_rigid_stamps[3], slw); atomicAdd(&g_rigid_stamps[4], slb);
      atomicAdd(&g_rigid_stamps[6], 1ull);
    }
#endif
    return;
  }

  // ------------------------------------------------------------------ compute waves
  const int wvx = wave % RLS_WX, wvy = wave / RLS_WX;
  const int x0 = xt + wvx * (RIGID_LANES * 4) + lane * 4;
  float acc0[RIGID_ROWS][4], acc1[RIGID_ROWS][4];
#pragma unroll
  for (int r = 0; r < RIGID_ROWS; ++r)
#pragma unroll
    for (int k = 0; k < 4; ++k) acc0[r][k] = acc1[r][k] = 0.f;
  const int ya = yt + wvy * RIGID_ROWS, yb = ya + RLS_TH / 2;
  const bool full_x = xt + RLS_TW <= w;
  const bool full_a = full_x && yt + RLS_TH / 2 <= h, full_b = full_x && yt + RLS_TH <= h;
  const int strip = (wvy * RIGID_ROWS) * RLS_Q + wvx * RIGID_LANES + lane;  // this lane's first quad
  float wx[5][4];
#ifdef MC_RIGID_STAMP
  unsigned long long st_c = 0, st_b = 0;
#endif
  __syncthreads();  // first window and the first frame's weights have landed
  for (int f = 0; f < nframes; ++f) {
#pragma unroll
    for (int j = 0; j < 5; ++j) {
      const float4 t = wxs[j * (RLS_TW / 4) + wvx * RIGID_LANES + lane];
      wx[j][0] = t.x; wx[j][1] = t.y; wx[j][2] = t.z; wx[j][3] = t.w;
    }
    // both halves' row weights now: the loader overwrites them during the second half
    float wya = 0.f, wyb = 0.f;
    if (lane < 5 * RIGID_ROWS) {
      wya = wys[(wvy * RIGID_ROWS) * 5 + lane];
      wyb = wys[(RLS_TH / 2 + wvy * RIGID_ROWS) * 5 + lane];
    }
#ifdef MC_RIGID_STAMP
    unsigned long long C0, C1, C2, C3, C4;
#endif
    RSTAMP(C0);
    if (full_a) rigid_strip_dma<WRITE_FRAMES, WRITE_SUM, true, RLS_Q>(a, buf0 + strip, f, ya, x0, wya, wx, acc0);
    else rigid_strip_dma<WRITE_FRAMES, WRITE_SUM, false, RLS_Q>(a, buf0 + strip, f, ya, x0, wya, wx, acc0);
    RSTAMP(C1);
    __syncthreads();
    RSTAMP(C2);
    if (full_b) rigid_strip_dma<WRITE_FRAMES, WRITE_SUM, true, RLS_Q>(a, buf1 + strip, f, yb, x0, wyb, wx, acc1);
    else rigid_strip_dma<WRITE_FRAMES, WRITE_SUM, false, RLS_Q>(a, buf1 + strip, f, yb, x0, wyb, wx, acc1);
    RSTAMP(C3);
    __syncthreads();
    RSTAMP(C4);
#ifdef MC_RIGID_STAMP
    st_c += (C1 - C0) + (C3 - C2); st_b += (C2 - C1) + (C4 - C3);
#endif
  }
#ifdef MC_RIGID_STAMP
  if (lane == 0) {
    atomicAdd(&g_rigid_stamps[0], st_c); atomicAdd(&g_rigid_stamps[1], st_b); atomicAdd(&g_rigid_stamps[5], 1ull);
  }
#endif
  if (WRITE_SUM && x0 < w) {
    const bool al = (((uintptr_t)a.out_sum) & 15) == 0;  // w % 4 == 0 on this path
#pragma unroll
    for (int ro = 0; ro < RIGID_ROWS; ++ro) {
      if (ya + ro < h) {
        float* dst = a.out_sum + (int64_t)(ya + ro) * w + x0;
        if (al) *reinterpret_cast<float4*>(dst) = make_float4(acc0[ro][0], acc0[ro][1], acc0[ro][2], acc0[ro][3]);
        else
#pragma unroll
          for (int k = 0; k < 4; ++k) dst[k] = acc0[ro][k];
      }
      if (yb + ro < h) {
        float* dst = a.out_sum + (int64_t)(yb + ro) * w + x0;
        if (al) *reinterpret_cast<float4*>(dst) = make_float4(acc1[ro][0], acc1[ro][1], acc1[ro][2], acc1[ro][3]);
        else
#pragma unroll
          for (int k = 0; k < 4; ++k) dst[k] = acc1[ro][k];
      }
    }
  }
}

#endif  // MC_EXPERIMENTS

// ------------------------------------------------------------------ rigid warp, LDS-DMA, fp16 frames
// The same kernel for frames stored as fp16 (N2: fp16 storage read natively): the window goes
// HBM -> LDS as the raw 16-bit samples -- half the bytes of the fp32 kernel on the read side -- and is
// widened on the way from LDS to the registers (8 v_cvt_f32_f16 per window row of a lane).  The
// global side of the DMA needs 4-byte alignment, so the window starts at the EVEN column at or left
// of x_tile + Sx - 1; the parity p of that column is wave-uniform per frame and selects one of two
// strip bodies with compile-time sample positions: a lane's 8-sample window is the halfs
// [4 L + p, 4 L + p + 8) of the tile row = two (p = 0) or three (p = 1) aligned ds_read_b64.
// Tile rows hold QH = 32 WX + 1 units of 8 samples.  Requires w % 8 == 0 and 16-byte aligned frames.
__device__ __forceinline__ float rh_lo(unsigned v) {
  return (float)__builtin_bit_cast(_Float16, (unsigned short)(v & 0xffffu));
}
__device__ __forceinline__ float rh_hi(unsigned v) { return (float)__builtin_bit_cast(_Float16, (unsigned short)(v >> 16)); }

template <bool WRITE_FRAMES, bool WRITE_SUM, bool FULL, int P, int UNITS8>
__device__ __forceinline__ void rigid_strip_half(const RigidArgs& a, const uint2* wrow, int f, int y0, int x0,
                                                 float wyv, const float (&wx)[5][4],
                                                 float (&acc)[RIGID_ROWS][4]) {
  const int h = a.h, w = a.w;
  float* orow = WRITE_FRAMES ? a.out_frames + (int64_t)f * h * w + (int64_t)y0 * w + x0 : nullptr;
  float H[5][4];
#pragma unroll
  for (int rr = 0; rr < RIGID_ROWS + 4; ++rr) {
    if (RIGID_SB > 0 && (rr % (RIGID_SB > 0 ? RIGID_SB : 1)) == 0) __builtin_amdgcn_sched_barrier(0);
    const uint2 u0 = wrow[rr * UNITS8], u1 = wrow[rr * UNITS8 + 1];
    float e[8];
    if constexpr (P == 0) {
      e[0] = rh_lo(u0.x); e[1] = rh_hi(u0.x); e[2] = rh_lo(u0.y); e[3] = rh_hi(u0.y);
      e[4] = rh_lo(u1.x); e[5] = rh_hi(u1.x); e[6] = rh_lo(u1.y); e[7] = rh_hi(u1.y);
    } else {
      const unsigned u2x = reinterpret_cast<const unsigned*>(wrow + rr * UNITS8 + 2)[0];
      e[0] = rh_hi(u0.x); e[1] = rh_lo(u0.y); e[2] = rh_hi(u0.y); e[3] = rh_lo(u1.x);
      e[4] = rh_hi(u1.x); e[5] = rh_lo(u1.y); e[6] = rh_hi(u1.y); e[7] = rh_lo(u2x);
    }
    float* Hn = H[rr % 5];
#pragma unroll
    for (int k = 0; k < 4; ++k)
      Hn[k] = rigid_dot5(wx[0][k], e[k], wx[1][k], e[k + 1], wx[2][k], e[k + 2], wx[3][k], e[k + 3], wx[4][k], e[k + 4]);
    if (rr >= 4) {
      const int ro = rr - 4;
      float wy[5];
#pragma unroll
      for (int i = 0; i < 5; ++i)
        wy[i] = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(wyv), ro * 5 + i));
      float o[4];
#pragma unroll
      for (int k = 0; k < 4; ++k)
        o[k] = rigid_dot5(wy[0], H[(ro + 0) % 5][k], wy[1], H[(ro + 1) % 5][k], wy[2], H[(ro + 2) % 5][k], wy[3],
                          H[(ro + 3) % 5][k], wy[4], H[(ro + 4) % 5][k]);
      if (FULL || (y0 + ro < h && x0 < w)) {
        if (WRITE_FRAMES) *reinterpret_cast<float4*>(orow + (int64_t)ro * w) = make_float4(o[0], o[1], o[2], o[3]);
        if (WRITE_SUM) {
#pragma unroll
          for (int k = 0; k < 4; ++k) acc[ro][k] += o[k];
        }
      }
    }
  }
}

template <bool WRITE_FRAMES, bool WRITE_SUM, int WX, int WY>
__global__ __launch_bounds__(RIGID_LANES* WX* WY, 4)
void warp_rigid_dma_h(RigidArgs a) {
  constexpr int NWAVES = WX * WY;
  constexpr int TROWS = WY * RIGID_ROWS + 4;   // input rows per tile
  constexpr int QH = WX * 32 + 1;              // 16-byte units (8 samples) per tile row
  constexpr int NQ = TROWS * QH;
  constexpr int UNITS_PAD = ((NQ + 63) / 64) * 64;  // DMA granule: 64 lanes x 16 B
  extern __shared__ __attribute__((aligned(16))) char smem_rd[];
  float4* const b0 = reinterpret_cast<float4*>(smem_rd);
  const _Float16* const frames = reinterpret_cast<const _Float16*>(a.frames);
  const int nt = a.tiles_x * a.tiles_y;
  const int b = blockIdx.x;
  int tile = b;
  if ((nt & 7) == 0) tile = (b & 7) * (nt >> 3) + (b >> 3);  // one band of tile rows per XCD
  const int tyi = tile / a.tiles_x, txi = tile - tyi * a.tiles_x;
  const int h = a.h, w = a.w;
  const int lane = threadIdx.x;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.y);
  const int wvx = wave % WX, wvy = wave / WX;
  const int tid = wave * RIGID_LANES + lane;
  const int xt = txi * (RIGID_LANES * 4 * WX);
  const int yt = tyi * (WY * RIGID_ROWS);
  const int x0 = xt + wvx * (RIGID_LANES * 4) + lane * 4;
  const int y0 = yt + wvy * RIGID_ROWS;
  const int64_t hw = (int64_t)h * w;
  float acc[RIGID_ROWS][4];
#pragma unroll
  for (int r = 0; r < RIGID_ROWS; ++r)
#pragma unroll
    for (int k = 0; k < 4; ++k) acc[r][k] = 0.f;
  const bool full_tile = yt + WY * RIGID_ROWS <= h && xt + RIGID_LANES * 4 * WX <= w;
  const int f_lo = a.frames_in_grid ? (int)blockIdx.y * a.frames_in_grid : 0;
  const int f_hi = a.frames_in_grid ? min(f_lo + a.frames_in_grid, a.nframes) : a.nframes;

  auto dma = [&](int f) {
    const _Float16* fr = frames + (int64_t)f * hw;
    const int Sy = a.S[2 * f], Sx = a.S[2 * f + 1];
    const int axe = (xt + Sx - 1) & ~1;  // even column: 4-byte aligned on the global side
    for (int i = wave; i < UNITS_PAD / 64; i += NWAVES) {
      int q = i * 64 + lane;
      q = q < NQ ? q : NQ - 1;  // tail lanes re-load the last unit into the pad
      const int tr = q / QH, qc = q - tr * QH;
      int r = yt + Sy - 1 + tr;
      r = r < 0 ? 0 : (r > h - 1 ? h - 1 : r);
      int c = axe + 8 * qc;
      c = c < 0 ? 0 : (c > w - 8 ? w - 8 : c);  // whole units inside the row; clamped ones are patched
      __builtin_amdgcn_global_load_lds(fr + (int64_t)r * w + c, (lds_vptr)(b0 + i * 64), 16, 0, 0);
    }
  };
  // border padding for edge tiles: a unit whose 8 columns are not all inside the row was DMA'd from a
  // clamped address; its samples are re-fetched one by one at their clipped column
  auto patch = [&](int f) {
    const int Sy = a.S[2 * f], Sx = a.S[2 * f + 1];
    const int axe = (xt + Sx - 1) & ~1;
    if (axe >= 0 && axe + 8 * QH <= w) return false;
    const _Float16* fr = frames + (int64_t)f * hw;
    _Float16* t = reinterpret_cast<_Float16*>(b0);
    for (int q = tid; q < NQ; q += RIGID_LANES * NWAVES) {
      const int tr = q / QH, qc = q - tr * QH;
      const int s0 = axe + 8 * qc;
      if (s0 >= 0 && s0 <= w - 8) continue;
      int r = yt + Sy - 1 + tr;
      r = r < 0 ? 0 : (r > h - 1 ? h - 1 : r);
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        int c = s0 + e;
        c = c < 0 ? 0 : (c > w - 1 ? w - 1 : c);
        t[8 * q + e] = fr[(int64_t)r * w + c];
      }
    }
    return true;
  };

  float wx[5][4];
  float wyv = 0.f;
  auto load_weights = [&](int f) {
    wyv = 0.f;
    const int64_t idx = (int64_t)y0 * 5 + lane;
    if (lane < 5 * RIGID_ROWS && idx < (int64_t)h * 5) wyv = a.Wy[(int64_t)f * 5 * h + idx];
    const float* Wx = a.Wx + (int64_t)f * 5 * w + x0;
#pragma unroll
    for (int j = 0; j < 5; ++j) {
      float4 t = make_float4(0.f, 0.f, 0.f, 0.f);
      if (x0 < w) t = *reinterpret_cast<const float4*>(Wx + (int64_t)j * w);
      wx[j][0] = t.x; wx[j][1] = t.y; wx[j][2] = t.z; wx[j][3] = t.w;
    }
  };
  // this lane's first 8-byte unit (4 samples): row (wvy RIGID_ROWS), sample 4 (64 wvx + lane)
  const uint2* const strip = reinterpret_cast<const uint2*>(b0) + (wvy * RIGID_ROWS) * (2 * QH) + wvx * RIGID_LANES + lane;

  load_weights(f_lo);
  dma(f_lo);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  if (patch(f_lo)) __syncthreads();
  for (int f = f_lo; f < f_hi; ++f) {
    const int par = __builtin_amdgcn_readfirstlane((xt + a.S[2 * f + 1] - 1) & 1);
    if (par) {
      if (full_tile) rigid_strip_half<WRITE_FRAMES, WRITE_SUM, true, 1, 2 * QH>(a, strip, f, y0, x0, wyv, wx, acc);
      else rigid_strip_half<WRITE_FRAMES, WRITE_SUM, false, 1, 2 * QH>(a, strip, f, y0, x0, wyv, wx, acc);
    } else {
      if (full_tile) rigid_strip_half<WRITE_FRAMES, WRITE_SUM, true, 0, 2 * QH>(a, strip, f, y0, x0, wyv, wx, acc);
      else rigid_strip_half<WRITE_FRAMES, WRITE_SUM, false, 0, 2 * QH>(a, strip, f, y0, x0, wyv, wx, acc);
    }
    if (f + 1 < f_hi) {
      __syncthreads();  // everyone must be done reading before the tile is refilled
      dma(f + 1);
      load_weights(f + 1);
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();
      if (patch(f + 1)) __syncthreads();
    }
  }
  if (WRITE_SUM && x0 < w) {
#pragma unroll
    for (int ro = 0; ro < RIGID_ROWS; ++ro) {
      const int yo = y0 + ro;
      if (yo < h) {
        float* dst = a.out_sum + (int64_t)yo * w + x0;
        if ((((uintptr_t)a.out_sum) & 15) == 0) {
          *reinterpret_cast<float4*>(dst) = make_float4(acc[ro][0], acc[ro][1], acc[ro][2], acc[ro][3]);
        } else {
#pragma unroll
          for (int k = 0; k < 4; ++k) dst[k] = acc[ro][k];
        }
      }
    }
  }
}
// ------------------------------------------------------------------ rigid warp from RAW frames (N2)
// The same resampler fed from the raw detector bytes: c = raw * gain - mu_f (gain_correct and
// set_frames_mean_zero of the reference's pipeline, examples/ttMotion.py:90-121, 180-199) is formed on the
// way from LDS to the registers, so no conditioned fp32 movie exists and the HBM side of a 40 x 4096^2 u8
// movie reads 0.8 GB instead of 3.3.
//  * A tile's input windows of all frames overlap almost entirely (they differ by the integer part of the
//    drift), so the GAIN values they need stay in LDS for the whole frame loop: a cache of (36 + RR_MY) x
//    (516 + RR_MX + 4) floats shared by the workgroup, de-interleaved into four planes by column mod 4 so
//    that a lane's eight values come from eight conflict-free ds_read_b32 whatever the window's offset in
//    the cache.  It is re-centred when a frame's window leaves it -- every wave meets the same frames in
//    the same order, so the re-load is a workgroup barrier they all arrive at (never a correctness matter;
//    once or twice per tile for a drift of +-8 px).
//  * The raw bytes are so few (3.4 KB per wave and frame for u8) that every WAVE keeps its own 12-row
//    window, double-buffered for u8: no workgroup barrier in the frame loop at all, the eight waves of a
//    tile drift apart and cover each other's store and DMA latencies (the barrier-coupled first version
//    ran at the speed of the fp32 kernel although it reads a quarter of its bytes).  Windows are 16-byte
//    units DMA'd from the 4-byte aligned column at or left of the window (u8: a multiple of 4 samples,
//    i16: of 2); the sub-unit offset m is wave-uniform and resolved by v_alignbyte on the three (five)
//    dwords a lane reads per window row.
// One workgroup of 8 waves (512 x 32 output pixels) per CU.  Requires w % 4 == 0, 16-byte aligned buffers.
#define RR_MY 8
#define RR_MX 8
#define RR_TROWS (4 * RIGID_ROWS + 4)                   // 36 rows of a tile's window
#define RR_WROWS (RIGID_ROWS + 4)                       // 12 rows of a wave's window
#define RR_GR (RR_TROWS + RR_MY)                        // 44 cached gain rows
#define RR_GQ (2 * RIGID_LANES + 1 + RR_MX / 4 + 1)     // 132 quads per cached row (528 columns)
#define RR_GAIN_BYTES (RR_GR * 4 * RR_GQ * 4)
#define RR_PAR_MAX 256                                  // frames whose {Sy, Sx, mu} are kept in LDS

template <int KIND>
struct RawWin {
  static constexpr int SB = KIND == 0 ? 1 : 2;                 // bytes per sample
  static constexpr int UPS = 16 / SB;                          // samples per 16-byte unit
  static constexpr int AL = 4 / SB;                            // samples per 4 bytes: window start granule
  static constexpr int NU = (RIGID_LANES * 4 + 4 + AL - 1 + UPS - 1) / UPS;  // units per wave-window row (18 / 34)
  static constexpr int RSTRIDE = NU * 16;                      // bytes per window row in LDS
  static constexpr int NUNITS = RR_WROWS * NU;
  static constexpr int UNITS_PAD = ((NUNITS + 63) / 64) * 64;
  static constexpr int NBUF = KIND == 0 ? 2 : 1;
  static constexpr int WAVE_BYTES = NBUF * UNITS_PAD * 16;
  static constexpr int LDS_BYTES = RR_GAIN_BYTES + 8 * WAVE_BYTES + RR_PAR_MAX * 16;
};

struct RigidRawArgs {
  RigidArgs r;        // r.frames = the raw movie
  const float* gain;  // (h, w)
  const float* mu;    // [f] frame means (mc_raw_movie_stats), subtracted after the gain multiply
};

template <int KIND, int RR0>
__device__ __forceinline__ void rigid_raw_read4(unsigned rawrow, unsigned (&d)[4][5]) {
  // four window rows of this lane's dwords (3 per row for u8, 5 for i16) by INLINE ASSEMBLY: a ds_read the
  // compiler can see, from the array an LDS-DMA is in flight to, gets an s_waitcnt vmcnt(0) in front of it
  // (it cannot tell the two halves of the window array apart) -- which waits for the DMA issued a moment ago
  // and for every store.  The frame loop waits for exactly the DMA that filled THIS window before the strip.
  using RW = RawWin<KIND>;
  constexpr int B = RR0 * RW::RSTRIDE, S = RW::RSTRIDE;
  if constexpr (KIND == 0) {
    asm volatile(
        "ds_read_b32 %0, %12 offset:%13\n\tds_read_b32 %1, %12 offset:%14\n\tds_read_b32 %2, %12 offset:%15\n\t"
        "ds_read_b32 %3, %12 offset:%16\n\tds_read_b32 %4, %12 offset:%17\n\tds_read_b32 %5, %12 offset:%18\n\t"
        "ds_read_b32 %6, %12 offset:%19\n\tds_read_b32 %7, %12 offset:%20\n\tds_read_b32 %8, %12 offset:%21\n\t"
        "ds_read_b32 %9, %12 offset:%22\n\tds_read_b32 %10, %12 offset:%23\n\tds_read_b32 %11, %12 offset:%24\n\t"
        "s_waitcnt lgkmcnt(0)"
        : "=&v"(d[0][0]), "=&v"(d[0][1]), "=&v"(d[0][2]), "=&v"(d[1][0]), "=&v"(d[1][1]), "=&v"(d[1][2]),
          "=&v"(d[2][0]), "=&v"(d[2][1]), "=&v"(d[2][2]), "=&v"(d[3][0]), "=&v"(d[3][1]), "=&v"(d[3][2])
        : "v"(rawrow), "i"(B), "i"(B + 4), "i"(B + 8), "i"(B + S), "i"(B + S + 4), "i"(B + S + 8), "i"(B + 2 * S),
          "i"(B + 2 * S + 4), "i"(B + 2 * S + 8), "i"(B + 3 * S), "i"(B + 3 * S + 4), "i"(B + 3 * S + 8)
        : "memory");
  } else {
#pragma unroll
    for (int r = 0; r < 4; ++r)
      asm volatile("ds_read_b32 %0, %5 offset:%6\n\tds_read_b32 %1, %5 offset:%7\n\tds_read_b32 %2, %5 offset:%8\n\t"
                   "ds_read_b32 %3, %5 offset:%9\n\tds_read_b32 %4, %5 offset:%10"
                   : "=&v"(d[r][0]), "=&v"(d[r][1]), "=&v"(d[r][2]), "=&v"(d[r][3]), "=&v"(d[r][4])
                   : "v"(rawrow + (unsigned)(B + r * S)), "i"(0), "i"(4), "i"(8), "i"(12), "i"(16)
                   : "memory");
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  }
}

// The strip on 2-float vectors: at two waves per SIMD (all the 93 KB gain cache allows) the kernel is bound by
// vector-instruction ISSUE, one instruction per ~5 cycles and wave, and v_pk_fma_f32 does two of the separable
// passes' multiply-adds per issue slot (mc_wave_fft.h's K1 is written the same way for the same reason).  Lane
// arithmetic and its order are rigid_dot5's, so the results are those of the scalar strip bit for bit.
typedef float rr_f2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ rr_f2 rr_dot5(const rr_f2 (&wv)[5], rr_f2 e0, rr_f2 e1, rr_f2 e2, rr_f2 e3, rr_f2 e4) {
  rr_f2 r = wv[0] * e0;
  r = __builtin_elementwise_fma(wv[1], e1, r);
  r = __builtin_elementwise_fma(wv[2], e2, r);
  r = __builtin_elementwise_fma(wv[3], e3, r);
  return __builtin_elementwise_fma(wv[4], e4, r);
}

template <bool WRITE_FRAMES, bool WRITE_SUM, bool FULL, int KIND>
__device__ __forceinline__ void rigid_strip_raw(const RigidArgs& a, unsigned rawrow, int m, const float* gplane,
                                                const int (&gofs)[8], float negmu, int f, int y0, int x0, float wyv,
                                                const float (&wx)[5][4], float (&acc)[RIGID_ROWS][4]) {
  const int h = a.h, w = a.w;
  float* orow = WRITE_FRAMES ? a.out_frames + (int64_t)f * h * w + (int64_t)y0 * w + x0 : nullptr;
  rr_f2 WA[5], WB[5];  // x weights of output columns (0,1) and (2,3), per tap
#pragma unroll
  for (int j = 0; j < 5; ++j) {
    WA[j] = rr_f2{wx[j][0], wx[j][1]};
    WB[j] = rr_f2{wx[j][2], wx[j][3]};
  }
  const rr_f2 nm = {negmu, negmu};
  rr_f2 HA[5], HB[5];  // horizontal-pass results of the last five window rows
  unsigned d[4][5];
#pragma unroll
  for (int rr = 0; rr < RIGID_ROWS + 4; ++rr) {
    if ((rr & 3) == 0) {
      __builtin_amdgcn_sched_barrier(0);
      if (rr == 0) rigid_raw_read4<KIND, 0>(rawrow, d);
      else if (rr == 4) rigid_raw_read4<KIND, 4>(rawrow, d);
      else rigid_raw_read4<KIND, 8>(rawrow, d);
    }
    float rv[8];
    if constexpr (KIND == 0) {
      const unsigned w0 = __builtin_amdgcn_alignbyte(d[rr & 3][1], d[rr & 3][0], m);
      const unsigned w1 = __builtin_amdgcn_alignbyte(d[rr & 3][2], d[rr & 3][1], m);
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        rv[k] = (float)((w0 >> (8 * k)) & 0xffu);
        rv[4 + k] = (float)((w1 >> (8 * k)) & 0xffu);
      }
    } else {
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const unsigned wj = __builtin_amdgcn_alignbyte(d[rr & 3][j + 1], d[rr & 3][j], 2 * m);
        rv[2 * j] = (float)(short)(wj & 0xffffu);
        rv[2 * j + 1] = (float)((int)wj >> 16);
      }
    }
    float g[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) g[k] = gplane[gofs[k] + rr * (4 * RR_GQ)];
    // conditioned samples e[k] = raw * gain - mu as even pairs (e0,e1) .. (e6,e7) and the odd pairs between them
    rr_f2 E[7];
#pragma unroll
    for (int i = 0; i < 4; ++i)
      E[2 * i] = __builtin_elementwise_fma(rr_f2{rv[2 * i], rv[2 * i + 1]}, rr_f2{g[2 * i], g[2 * i + 1]}, nm);
#pragma unroll
    for (int i = 0; i < 3; ++i) E[2 * i + 1] = __builtin_shufflevector(E[2 * i], E[2 * i + 2], 1, 2);
    HA[rr % 5] = rr_dot5(WA, E[0], E[1], E[2], E[3], E[4]);
    HB[rr % 5] = rr_dot5(WB, E[2], E[3], E[4], E[5], E[6]);
    if (rr >= 4) {
      const int ro = rr - 4;
      rr_f2 wy[5];
#pragma unroll
      for (int i = 0; i < 5; ++i) {
        const float s = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(wyv), ro * 5 + i));
        wy[i] = rr_f2{s, s};
      }
      const rr_f2 oa = rr_dot5(wy, HA[(ro + 0) % 5], HA[(ro + 1) % 5], HA[(ro + 2) % 5], HA[(ro + 3) % 5], HA[(ro + 4) % 5]);
      const rr_f2 ob = rr_dot5(wy, HB[(ro + 0) % 5], HB[(ro + 1) % 5], HB[(ro + 2) % 5], HB[(ro + 3) % 5], HB[(ro + 4) % 5]);
      if (FULL || (y0 + ro < h && x0 < w)) {
        if (WRITE_FRAMES) rigid_store4(orow + (int64_t)ro * w, oa.x, oa.y, ob.x, ob.y);
        if (WRITE_SUM) {
          acc[ro][0] += oa.x; acc[ro][1] += oa.y; acc[ro][2] += ob.x; acc[ro][3] += ob.y;
        }
      }
    }
  }
}

// a wave's LDS hand-off to itself (DS operations of one wave execute in order): compiler fence only
__device__ __forceinline__ void rr_wave_sync() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

template <bool WRITE_FRAMES, bool WRITE_SUM, int KIND>
__global__ __launch_bounds__(RIGID_LANES * 8, 2) void warp_rigid_raw(RigidRawArgs ra) {
  using RW = RawWin<KIND>;
  constexpr int WX = 2, WY = 4, NWAVES = 8, NBUF = RW::NBUF;
  const RigidArgs& a = ra.r;
  // THREE separate LDS objects, so that the compiler's alias scopes tell the gain cache and the parameter
  // table (read in the frame loop) apart from the raw windows (the LDS-DMA destination): reads of the former
  // then need no vmcnt wait while a window DMA is in flight
  __shared__ float gplane[RR_GR * 4 * RR_GQ];                               // [RR_GR][4 planes][RR_GQ]
  __shared__ __attribute__((aligned(16))) char rawwin[8 * RW::WAVE_BYTES];  // per wave: NBUF windows
  __shared__ int4 s_par[RR_PAR_MAX];                                        // {Sy, Sx, mu} per frame
  const unsigned char* const raw = reinterpret_cast<const unsigned char*>(a.frames);
  const int nt = a.tiles_x * a.tiles_y;
  const int b = blockIdx.x;
  int tile = b;
  if ((nt & 7) == 0) tile = (b & 7) * (nt >> 3) + (b >> 3);  // one band of tile rows per XCD
  const int tyi = tile / a.tiles_x, txi = tile - tyi * a.tiles_x;
  const int h = a.h, w = a.w;
  const int lane = threadIdx.x;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.y);
  const int wvx = wave % WX, wvy = wave / WX;
  const int tid = wave * RIGID_LANES + lane;
  const int xt = txi * (RIGID_LANES * 4 * WX), yt = tyi * (WY * RIGID_ROWS);
  const int xw = xt + wvx * (RIGID_LANES * 4);  // first output column of this wave
  const int x0 = xw + lane * 4, y0 = yt + wvy * RIGID_ROWS;
  const int64_t hw = (int64_t)h * w;
  char* const wb0 = rawwin + wave * RW::WAVE_BYTES;  // this wave's raw window(s)
  char* const wb1 = NBUF == 2 ? wb0 + RW::UNITS_PAD * 16 : wb0;
  // Per-frame parameters {Sy, Sx, mu} live in LDS: read from global memory inside the frame loop they
  // become VECTOR loads (the compiler cannot prove that the frame stores do not alias them), and the
  // s_waitcnt vmcnt(0) in front of their first use drains every DMA and every store in flight -- the
  // first version of this kernel ran at the fp32 kernel's speed because of exactly that.
  for (int i = tid; i < a.nframes && i < RR_PAR_MAX; i += RIGID_LANES * NWAVES)
    s_par[i] = make_int4(a.S[2 * i], a.S[2 * i + 1], __float_as_int(ra.mu[i]), 0);
  __syncthreads();
  auto par = [&](int f) {  // nframes <= RR_PAR_MAX (host): an LDS read, never a (flat) load that waits for vmcnt(0)
    const int4 p = s_par[f];
    return make_int4(__builtin_amdgcn_readfirstlane(p.x), __builtin_amdgcn_readfirstlane(p.y),
                     __builtin_amdgcn_readfirstlane(p.z), 0);
  };
  float acc[RIGID_ROWS][4];
#pragma unroll
  for (int r = 0; r < RIGID_ROWS; ++r)
#pragma unroll
    for (int k = 0; k < 4; ++k) acc[r][k] = 0.f;
  const bool full_tile = yt + WY * RIGID_ROWS <= h && xt + RIGID_LANES * 4 * WX <= w;

  // the wave's raw window of frame f -> LDS: unit u = (window row, 16 bytes); rows clipped to the image,
  // units kept whole inside the row (the ones that had to move are patched)
  auto win_x = [&](int f) { return xw + par(f).y - 1; };   // image column of the WAVE's window column 0
  auto win_y = [&](int f) { return y0 + par(f).x - 1; };   // image row of the wave's window row 0
  auto dma = [&](int f, char* dst) {
    const unsigned char* fr = raw + (int64_t)f * hw * RW::SB;
    const int wy = win_y(f), axa = win_x(f) & ~(RW::AL - 1);
#pragma unroll
    for (int i = 0; i < RW::UNITS_PAD / 64; ++i) {
      int u = i * 64 + lane;
      u = u < RW::NUNITS ? u : RW::NUNITS - 1;
      const int tr = u / RW::NU, uc = u - tr * RW::NU;
      int r = wy + tr;
      r = r < 0 ? 0 : (r > h - 1 ? h - 1 : r);
      int c = axa + RW::UPS * uc;
      c = c < 0 ? 0 : (c > w - RW::UPS ? w - RW::UPS : c);
      __builtin_amdgcn_global_load_lds(reinterpret_cast<const float*>(fr + ((int64_t)r * w + c) * RW::SB),
                                       (lds_vptr)(dst + (i * 64) * 16), 16, 0, 0);
    }
  };
  auto patch = [&](int f, char* dst) {  // after the wave's vmcnt wait; its own window only
    const int axa = win_x(f) & ~(RW::AL - 1);
    if (axa >= 0 && axa + RW::UPS * RW::NU <= w) return;
    const unsigned char* fr = raw + (int64_t)f * hw * RW::SB;
    const int wy = win_y(f);
    int nl = axa < 0 ? (-axa + RW::UPS - 1) / RW::UPS : 0;
    nl = nl > RW::NU ? RW::NU : nl;
    int nr = w - RW::UPS - axa >= 0 ? (w - RW::UPS - axa) / RW::UPS + 1 : 0;
    nr = nr > RW::NU ? RW::NU : (nr < nl ? nl : nr);
    const int nbad = nl + (RW::NU - nr), per_row = nbad * RW::UPS, items = RR_WROWS * per_row;
    rr_wave_sync();
    for (int it = lane; it < items; it += RIGID_LANES) {
      const int tr = it / per_row, k = it - tr * per_row;
      const int bu = k / RW::UPS, e = k - bu * RW::UPS;
      const int uc = bu < nl ? bu : nr + (bu - nl);
      int r = wy + tr;
      r = r < 0 ? 0 : (r > h - 1 ? h - 1 : r);
      int c = axa + RW::UPS * uc + e;
      c = c < 0 ? 0 : (c > w - 1 ? w - 1 : c);
      if constexpr (KIND == 0) reinterpret_cast<unsigned char*>(dst)[tr * RW::RSTRIDE + uc * 16 + e] = fr[(int64_t)r * w + c];
      else reinterpret_cast<unsigned short*>(dst)[tr * (RW::RSTRIDE / 2) + uc * 8 + e] =
               reinterpret_cast<const unsigned short*>(fr)[(int64_t)r * w + c];
    }
    rr_wave_sync();
  };
  // gain cache (workgroup): rows [gy0, gy0 + RR_GR) x columns [gx0, gx0 + 4 RR_GQ), gx0 % 4 == 0, border =
  // clipped index.  Coverage is tested for the TILE's window, so every wave takes the same decision.
  int gy0 = 0, gx0 = 0;
  auto tile_x = [&](int f) { return xt + par(f).y - 1; };
  auto tile_y = [&](int f) { return yt + par(f).x - 1; };
  auto cache_covers = [&](int f) {
    const int wy = tile_y(f), ax = tile_x(f);
    return wy >= gy0 && wy + RR_TROWS <= gy0 + RR_GR && ax >= gx0 && ax + 4 * (2 * RIGID_LANES + 1) <= gx0 + 4 * RR_GQ;
  };
  auto cache_load = [&](int f) {
    gy0 = tile_y(f) - RR_MY / 2;
    gx0 = (tile_x(f) - RR_MX / 2) & ~3;
    for (int i = tid; i < RR_GR * RR_GQ; i += RIGID_LANES * NWAVES) {
      const int gr = i / RR_GQ, gq = i - gr * RR_GQ;
      int r = gy0 + gr;
      r = r < 0 ? 0 : (r > h - 1 ? h - 1 : r);
      const int c = gx0 + 4 * gq;
      float v[4];
      if (c >= 0 && c + 3 <= w - 1) {
        const float4 q = *reinterpret_cast<const float4*>(ra.gain + (int64_t)r * w + c);
        v[0] = q.x; v[1] = q.y; v[2] = q.z; v[3] = q.w;
      } else {
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          int cc = c + e;
          cc = cc < 0 ? 0 : (cc > w - 1 ? w - 1 : cc);
          v[e] = ra.gain[(int64_t)r * w + cc];
        }
      }
#pragma unroll
      for (int e = 0; e < 4; ++e) gplane[(gr * 4 + e) * RR_GQ + gq] = v[e];
    }
  };

  float wx[5][4], wxn[5][4];
  float wyv = 0.f, wyvn = 0.f;
  auto load_weights = [&](int f, float (&W5)[5][4], float& Wv) {
    Wv = 0.f;
    const int64_t idx = (int64_t)y0 * 5 + lane;
    if (lane < 5 * RIGID_ROWS && idx < (int64_t)h * 5) Wv = a.Wy[(int64_t)f * 5 * h + idx];
    const float* Wx = a.Wx + (int64_t)f * 5 * w + x0;
#pragma unroll
    for (int j = 0; j < 5; ++j) {
      float4 t = make_float4(0.f, 0.f, 0.f, 0.f);
      if (x0 < w) t = *reinterpret_cast<const float4*>(Wx + (int64_t)j * w);
      W5[j][0] = t.x; W5[j][1] = t.y; W5[j][2] = t.z; W5[j][3] = t.w;
    }
  };

  load_weights(0, wx, wyv);
  dma(0, wb0);
  cache_load(0);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  patch(0, wb0);
  __syncthreads();  // the gain cache is complete
  int cur = 0;
#ifdef MC_RIGID_STAMP
  unsigned long long Q0, Q1, Q2, Q3, Q4, sq_i = 0, sq_p = 0, sq_c = 0, sq_w = 0;
#endif
  for (int f = 0; f < a.nframes; ++f) {
    RSTAMP(Q0);
    if (NBUF == 2 && f + 1 < a.nframes) {
      dma(f + 1, cur ? wb0 : wb1);  // lands under this frame's arithmetic
      load_weights(f + 1, wxn, wyvn);
    }
    RSTAMP(Q1);
    if (!cache_covers(f)) {  // the same frames for every wave of the tile: a rendezvous, then the re-load
      __syncthreads();       // everyone has finished the frames that used the old position
      cache_load(f);
      __syncthreads();
    }
    const int ax = win_x(f);
    const int m = ax & (RW::AL - 1);
    const int dx = ax - gx0, a4 = dx & 3, q4 = dx >> 2;
    const int grow = win_y(f) - gy0;
    int gofs[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) gofs[k] = (grow * 4 + ((a4 + k) & 3)) * RR_GQ + q4 + ((a4 + k) >> 2) + lane;
    const unsigned rawrow = (unsigned)reinterpret_cast<uintptr_t>((cur ? wb1 : wb0) + lane * 4 * RW::SB);  // LDS byte address
    const float negmu = -__int_as_float(par(f).z);
    RSTAMP(Q2);
    if (full_tile) rigid_strip_raw<WRITE_FRAMES, WRITE_SUM, true, KIND>(a, rawrow, m, gplane, gofs, negmu, f, y0, x0, wyv, wx, acc);
    else rigid_strip_raw<WRITE_FRAMES, WRITE_SUM, false, KIND>(a, rawrow, m, gplane, gofs, negmu, f, y0, x0, wyv, wx, acc);
    RSTAMP(Q3);
    if (f + 1 < a.nframes) {
      if constexpr (NBUF == 1) {
        rr_wave_sync();  // this wave's reads of the window are done (in-order DS queue)
        dma(f + 1, wb0);
        load_weights(f + 1, wx, wyv);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        patch(f + 1, wb0);
      } else {
        if (WRITE_FRAMES && full_tile) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");  // the 8 row stores stay in flight
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        cur ^= 1;
        patch(f + 1, cur ? wb1 : wb0);
#pragma unroll
        for (int j = 0; j < 5; ++j)
#pragma unroll
          for (int k = 0; k < 4; ++k) wx[j][k] = wxn[j][k];
        wyv = wyvn;
      }
    }
#ifdef MC_RIGID_STAMP
    RSTAMP(Q4);
    sq_i += Q1 - Q0; sq_p += Q2 - Q1; sq_c += Q3 - Q2; sq_w += Q4 - Q3;
#endif
  }
#ifdef MC_RIGID_STAMP
  if (lane == 0) {
    atomicAdd(&g_rigid_stamps[0], sq_i); atomicAdd(&g_rigid_stamps[1], sq_p); atomicAdd(&g_rigid_stamps[2], sq_c);
    atomicAdd(&g_rigid_stamps[3], sq_w); atomicAdd(&g_rigid_stamps[5], 1ull);
  }
#endif
  if (WRITE_SUM && x0 < w) {
#pragma unroll
    for (int ro = 0; ro < RIGID_ROWS; ++ro) {
      const int yo = y0 + ro;
      if (yo < h) {
        float* dst = a.out_sum + (int64_t)yo * w + x0;
        if ((((uintptr_t)a.out_sum) & 15) == 0) {
          *reinterpret_cast<float4*>(dst) = make_float4(acc[ro][0], acc[ro][1], acc[ro][2], acc[ro][3]);
        } else {
#pragma unroll
          for (int k = 0; k < 4; ++k) dst[k] = acc[ro][k];
        }
      }
    }
  }
}
#pragma clang fp contract(off)

// ------------------------------------------------------------------ general warp, LDS tile
// Per (tile, frame): the shift at the tile centre positions a (32+3+2*MG) x (256+3+2*MG)
// input window that is DMA'd into LDS; every pixel then runs the reference's per-pixel
// coordinate chain (strict fp32, see file header) and gathers its 4x4 taps from LDS.
// Lane l owns pixels x = x_tile + l + 64k (k = 0..3): adjacent lanes read adjacent LDS
// words, so the data-dependent gathers are bank-conflict free.
//
// Whether ALL taps of a tile fit the window is decided up front, rigorously: a pixel's
// shift is a bicubic (A = -0.75) interpolation of lattice nodes, sum(w) = 1 and
// sum|w| <= 1.375^2 < 1.9 in 2-D, so with rho = half the range of the nodes that can
// influence the tile every shift lies within 1.9*rho of the mid-range value and within
// 3.8*rho of the centre pixel's.  Tile-frames that fail the test are only flagged here and
// are processed afterwards by warp_field_slow (generic global gathers).
// The x-direction of the shift-lattice upsample comes from the E table (warp_etab); a
// thread caches its 4 px x 4 lattice rows x 2 channels of E in registers while
// consecutive pixel rows use the same lattice rows (they almost always do).
#define GW_MG 6
#define GW_ROWS (RIGID_WAVES * RIGID_ROWS + 3 + 2 * GW_MG)              // 47
#define GW_QUADS ((RIGID_LANES * 4 + 3 + 2 * GW_MG + 3 + 3) / 4)         // 70 (alignment slack)
#define GW_STRIDE (4 * GW_QUADS)                                          // 280 floats
#define GW_NQ (GW_ROWS * GW_QUADS)
#define GW_QUADS_PAD (((GW_NQ + 63) / 64) * 64)

#pragma clang fp contract(fast)
__device__ __forceinline__ float gw_dot4(const float w[4], float a, float b, float c, float d) {
  return ((w[0] * a + w[1] * b) + w[2] * c) + w[3] * d;
}
#pragma clang fp contract(off)

struct FieldArgs {
  WarpArgs w;
  const float* lattice;  // [f][2][GH][GW]
  const int* xtap;       // [w][4]
  int GW;
  unsigned char* flags;  // [f][tile]: 1 = irregular, left to warp_field_slow
};

__device__ __forceinline__ int wave_min_i(int v) {
  for (int off = 32; off > 0; off >>= 1) {
    const int o = __shfl_xor(v, off);
    v = o < v ? o : v;
  }
  return v;
}
__device__ __forceinline__ int wave_max_i(int v) {
  for (int off = 32; off > 0; off >>= 1) {
    const int o = __shfl_xor(v, off);
    v = o > v ? o : v;
  }
  return v;
}
__device__ __forceinline__ float wave_min_f(float v) {
  for (int off = 32; off > 0; off >>= 1) v = fminf(v, __shfl_xor(v, off));
  return v;
}
__device__ __forceinline__ float wave_max_f(float v) {
  for (int off = 32; off > 0; off >>= 1) v = fmaxf(v, __shfl_xor(v, off));
  return v;
}

#ifdef MC_EXPERIMENTS  // the first LDS-tile kernel (MC_WARP_FIELD=1): superseded by warp_field2 / warp_field3
template <bool WRITE_FRAMES, bool WRITE_SUM, bool UNIT_PS>
__global__ __launch_bounds__(RIGID_LANES* RIGID_WAVES, 2) void warp_field(FieldArgs fa) {
  const WarpArgs& a = fa.w;
  extern __shared__ __attribute__((aligned(16))) char smem_gw[];
  float4* const tile4 = reinterpret_cast<float4*>(smem_gw);
  float* const tile = reinterpret_cast<float*>(smem_gw);
  __shared__ int s_ytap[RIGID_WAVES * RIGID_ROWS][4];
  __shared__ float s_ycoef[RIGID_WAVES * RIGID_ROWS][4];
  const int nt = a.tiles_x * a.tiles_y;
  const int b = blockIdx.x;
  int tl = b;
  if ((nt & 7) == 0) tl = (b & 7) * (nt >> 3) + (b >> 3);
  const int tyi = tl / a.tiles_x, txi = tl - tyi * a.tiles_x;
  const int h = a.h, w = a.w;
  const float fh = (float)h, fw = (float)w;
  const int lane = threadIdx.x;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.y);
  const int tid = wave * RIGID_LANES + lane;
  const int xt = txi * (RIGID_LANES * 4);
  const int yt = tyi * (RIGID_WAVES * RIGID_ROWS);
  const int y0 = yt + wave * RIGID_ROWS;
  const int64_t hw = (int64_t)h * w;
  // frame-invariant per-row lattice taps of this tile
  if (tid < RIGID_WAVES * RIGID_ROWS) {
    const int y = yt + tid < h ? yt + tid : h - 1;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      s_ytap[tid][k] = a.ytap[4 * y + k];
      s_ycoef[tid][k] = a.ycoef[4 * y + k];
    }
  }
  // lattice footprint of the tile (frame-invariant): node rows [R0,R1], node columns [C0,C1]
  int R0, R1, C0, C1;
  {
    int lo = 0x7fffffff, hi = -1;
    if (lane < RIGID_WAVES * RIGID_ROWS) {
      const int y = yt + lane < h ? yt + lane : h - 1;
      for (int k = 0; k < 4; ++k) {
        const int v = a.ytap[4 * y + k];
        lo = v < lo ? v : lo;
        hi = v > hi ? v : hi;
      }
    }
    R0 = wave_min_i(lo);
    R1 = wave_max_i(hi);
    lo = 0x7fffffff;
    hi = -1;
    for (int k = 0; k < 4; ++k) {
      const int x = xt + lane + 64 * k;
      const int xs = x < w ? x : w - 1;
      for (int j = 0; j < 4; ++j) {
        const int v = fa.xtap[4 * xs + j];
        lo = v < lo ? v : lo;
        hi = v > hi ? v : hi;
      }
    }
    C0 = wave_min_i(lo);
    C1 = wave_max_i(hi);
  }
  // centre pixel of the tile (clipped to the image)
  const int yc = (yt + 16 < h) ? yt + 16 : h - 1;
  const int xc = (xt + 128 < w) ? xt + 128 : w - 1;
  const int4 ytc = *reinterpret_cast<const int4*>(a.ytap + 4 * yc);
  const float4 ycc = *reinterpret_cast<const float4*>(a.ycoef + 4 * yc);
  float acc[RIGID_ROWS][4];
#pragma unroll
  for (int r = 0; r < RIGID_ROWS; ++r)
#pragma unroll
    for (int k = 0; k < 4; ++k) acc[r][k] = 0.f;
  __syncthreads();

  for (int f = 0; f < a.nframes; ++f) {
    const float* fr = a.frames + (int64_t)f * hw;
    const float* E = a.etab + (int64_t)f * 2 * a.GH * w;
    const int64_t chs = (int64_t)a.GH * w;  // channel stride of E
    // 0. regularity: range of the lattice nodes that can influence this tile
    {
      const float* L = fa.lattice + (int64_t)f * 2 * a.GH * fa.GW;
      const int ncol = C1 - C0 + 1, nnode = (R1 - R0 + 1) * ncol;
      float lo_y = 3.0e38f, hi_y = -3.0e38f, lo_x = 3.0e38f, hi_x = -3.0e38f;
      for (int i = lane; i < nnode; i += RIGID_LANES) {
        const int R = R0 + i / ncol, Cc = C0 + i % ncol;
        const float vy = L[(int64_t)R * fa.GW + Cc], vx = L[(int64_t)(a.GH + R) * fa.GW + Cc];
        lo_y = fminf(lo_y, vy); hi_y = fmaxf(hi_y, vy);
        lo_x = fminf(lo_x, vx); hi_x = fmaxf(hi_x, vx);
      }
      const float ry = 0.5f * (wave_max_f(hi_y) - wave_min_f(lo_y)) / a.pixel_spacing;
      const float rx = 0.5f * (wave_max_f(hi_x) - wave_min_f(lo_x)) / a.pixel_spacing;
      // |shift - shift_centre| <= 3.8*rho; taps span [-1,+2] around floor(); coordinate
      // rounding adds < 0.01 px.  NaNs fail the comparison and go to the slow kernel.
      const bool regular = (3.8f * ry + 1.05f <= (float)GW_MG) && (3.8f * rx + 1.05f <= (float)GW_MG);
      if (!regular) {  // workgroup-uniform: every wave computed the same numbers
        if (tid == 0) fa.flags[(int64_t)f * nt + tl] = 1;
        continue;
      }
    }
    // 1. window origin from the shift at the tile centre (identical in every lane)
    int wy0, ax;
    {
      const float* Ec = E + xc;
      float sy = dot4(ycc, Ec[(int64_t)ytc.x * w], Ec[(int64_t)ytc.y * w], Ec[(int64_t)ytc.z * w],
                      Ec[(int64_t)ytc.w * w]);
      float sx = dot4(ycc, Ec[chs + (int64_t)ytc.x * w], Ec[chs + (int64_t)ytc.y * w],
                      Ec[chs + (int64_t)ytc.z * w], Ec[chs + (int64_t)ytc.w * w]);
      if (!UNIT_PS) {
        sy = div_invariant(sy, a.pixel_spacing);
        sx = div_invariant(sx, a.pixel_spacing);
      }
      const float lim = 4.f * (fh + fw);
      const float dy = fminf(fmaxf(floorf(grid_chain((float)yc + sy, fh)) - (float)yc, -lim), lim);
      const float dx = fminf(fmaxf(floorf(grid_chain((float)xc + sx, fw)) - (float)xc, -lim), lim);
      wy0 = __builtin_amdgcn_readfirstlane(yt + (int)dy - 1 - GW_MG);
      ax = __builtin_amdgcn_readfirstlane((xt + (int)dx - 1 - GW_MG) & ~3);
    }
    // 2. window -> LDS (the previous frame's reads are behind the barrier at the loop's end)
    for (int i = wave; i < GW_QUADS_PAD / 64; i += RIGID_WAVES) {
      int q = i * 64 + lane;
      q = q < GW_NQ ? q : GW_NQ - 1;
      const int tr = q / GW_QUADS, qc = q - tr * GW_QUADS;
      int r = wy0 + tr;
      r = r < 0 ? 0 : (r > h - 1 ? h - 1 : r);
      int c = ax + 4 * qc;
      c = c < 0 ? 0 : (c > w - 4 ? w - 4 : c);
      __builtin_amdgcn_global_load_lds(fr + (int64_t)r * w + c, (lds_vptr)(tile4 + i * 64), 16, 0, 0);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (ax < 0 || ax + GW_STRIDE > w) {  // border padding: clipped columns (edge tiles only)
      for (int i = tid; i < GW_ROWS * GW_STRIDE; i += RIGID_LANES * RIGID_WAVES) {
        const int tr = i / GW_STRIDE, e = i - tr * GW_STRIDE;
        const int c = ax + e;
        if (c < 0 || c > w - 1) {
          const int cc = c < 0 ? 0 : w - 1;
          int qsrc = (cc & ~3) - ax;
          qsrc = qsrc < 0 ? 0 : (qsrc > GW_STRIDE - 4 ? GW_STRIDE - 4 : qsrc);
          tile[tr * GW_STRIDE + e] = tile[tr * GW_STRIDE + qsrc + (cc & 3)];
        }
      }
      __syncthreads();
    }
    // 3. pixels
    int4 ycache = make_int4(-1, -1, -1, -1);
    float ey[4][4], ex[4][4];  // [lattice tap][pixel k]
#pragma unroll
    for (int r = 0; r < RIGID_ROWS; ++r) {
      const int y = y0 + r;
      if (y >= h) break;
      const int row = wave * RIGID_ROWS + r;
      const int4 yt4 = make_int4(s_ytap[row][0], s_ytap[row][1], s_ytap[row][2], s_ytap[row][3]);
      const float4 yc4 = make_float4(s_ycoef[row][0], s_ycoef[row][1], s_ycoef[row][2], s_ycoef[row][3]);
      if (yt4.x != ycache.x || yt4.y != ycache.y || yt4.z != ycache.z || yt4.w != ycache.w) {
        ycache = yt4;  // wave-uniform: depends on y only
        const int rows4[4] = {yt4.x, yt4.y, yt4.z, yt4.w};
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
          for (int k = 0; k < 4; ++k) {
            const int x = xt + lane + 64 * k;
            const int xs = x < w ? x : w - 1;
            ey[i][k] = E[(int64_t)rows4[i] * w + xs];
            ex[i][k] = E[chs + (int64_t)rows4[i] * w + xs];
          }
      }
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        // one pixel at a time: without this the scheduler interleaves all 32 unrolled
        // pixel bodies and the kernel spills
        __builtin_amdgcn_sched_barrier(0);
        const int x = xt + lane + 64 * k;
        if (x >= w) continue;
        float sy = dot4(yc4, ey[0][k], ey[1][k], ey[2][k], ey[3][k]);
        float sx = dot4(yc4, ex[0][k], ex[1][k], ex[2][k], ex[3][k]);
        if (!UNIT_PS) {
          sy = div_invariant(sy, a.pixel_spacing);
          sx = div_invariant(sx, a.pixel_spacing);
        }
        const float cy = (float)y + sy, cx = (float)x + sx;
        const bool inside = (cy >= 0.f) && (cy <= fh - 1.f) && (cx >= 0.f) && (cx <= fw - 1.f);
        const float uy = grid_chain(cy, fh), ux = grid_chain(cx, fw);
        const float fy = floorf(uy), fx = floorf(ux);
        float wy[4], wx[4];
        cubic_coeffs_fast(uy - fy, wy);
        cubic_coeffs_fast(ux - fx, wx);
        // in range by the regularity test; the clamp only keeps a NaN/garbage coordinate
        // from reading outside the LDS tile
        int ly = (int)fy - 1 - wy0, lx = (int)fx - 1 - ax;
        ly = ly < 0 ? 0 : (ly > GW_ROWS - 4 ? GW_ROWS - 4 : ly);
        lx = lx < 0 ? 0 : (lx > GW_STRIDE - 4 ? GW_STRIDE - 4 : lx);
        const float* t0 = tile + ly * GW_STRIDE + lx;
        float rowv[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const float* t = t0 + i * GW_STRIDE;
          rowv[i] = gw_dot4(wx, t[0], t[1], t[2], t[3]);
        }
        float o = gw_dot4(wy, rowv[0], rowv[1], rowv[2], rowv[3]);
        o = inside ? o : 0.f;
        if (WRITE_FRAMES) a.out_frames[(int64_t)f * hw + (int64_t)y * w + x] = o;
        if (WRITE_SUM) acc[r][k] += o;
      }
    }
    __syncthreads();  // everyone is done with the tile before the next frame overwrites it
  }
  if (WRITE_SUM) {
#pragma unroll
    for (int r = 0; r < RIGID_ROWS; ++r) {
      const int y = y0 + r;
      if (y >= h) break;
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const int x = xt + lane + 64 * k;
        if (x < w) a.out_sum[(int64_t)y * w + x] = acc[r][k];  // warp_field_slow adds its tile-frames afterwards
      }
    }
  }
}
#endif  // MC_EXPERIMENTS

// ------------------------------------------------------------------ general warp, second version
// Same tiling, window DMA and per-pixel chain as warp_field, rebuilt around what limits it -- VALU
// issue and the number of window bytes:
//  * the window margin follows the field: mg = ceil(3.8 rho + 1.05) per axis and tile-frame (2 for
//    the smooth fields of real movies) instead of the fixed 6, lanes outside the needed window
//    issue no DMA (window bytes 1.6x -> 1.3x of the tile);
//  * 3 workgroups per CU (one 52 KB window each), so a workgroup's DMA wait hides under two others;
//  * tile-frames whose window lies inside the image (all but the frame's rim) take a body without
//    the zero-outside test and without index clamps;
//  * cubic-convolution weights in factored form: c0 = A t u^2, c3 = A u t^2, c1 = 1 - t^2 ((A+3) -
//    (A+2) t), c2 likewise in u = 1 - t (11 instead of 17 operations per axis; the same polynomials
//    as ATen's Horner forms, values equal to ~1e-7).
#ifndef GW2_MINW
#define GW2_MINW 3
#endif
#ifndef GW2_ILP
#define GW2_ILP 1
#endif
#pragma clang fp contract(fast)
__device__ __forceinline__ void cubic_coeffs_factored(float t, float c[4]) {
  const float A = -0.75f;
  const float u = 1.f - t;
  const float atu = (A * t) * u;
  c[0] = atu * u;
  c[3] = atu * t;
  c[1] = 1.f - (t * t) * ((A + 3.f) - (A + 2.f) * t);
  c[2] = 1.f - (u * u) * ((A + 3.f) - (A + 2.f) * u);
}
#pragma clang fp contract(off)

template <bool WRITE_FRAMES, bool WRITE_SUM, bool UNIT_PS>
__global__ __launch_bounds__(RIGID_LANES* RIGID_WAVES, GW2_MINW) void warp_field2(FieldArgs fa) {
  const WarpArgs& a = fa.w;
  extern __shared__ __attribute__((aligned(16))) char smem_gw[];
  float4* const tile4 = reinterpret_cast<float4*>(smem_gw);
  float* const tile = reinterpret_cast<float*>(smem_gw);
  __shared__ int s_ytap[RIGID_WAVES * RIGID_ROWS][4];
  __shared__ float s_ycoef[RIGID_WAVES * RIGID_ROWS][4];
  const int nt = a.tiles_x * a.tiles_y;
  const int b = blockIdx.x;
  int tl = b;
  if ((nt & 7) == 0) tl = (b & 7) * (nt >> 3) + (b >> 3);
  const int tyi = tl / a.tiles_x, txi = tl - tyi * a.tiles_x;
  const int h = a.h, w = a.w;
  const float fh = (float)h, fw = (float)w;
  const int lane = threadIdx.x;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.y);
  const int tid = wave * RIGID_LANES + lane;
  const int xt = txi * (RIGID_LANES * 4);
  const int yt = tyi * (RIGID_WAVES * RIGID_ROWS);
  const int y0 = yt + wave * RIGID_ROWS;
  const int64_t hw = (int64_t)h * w;
  if (tid < RIGID_WAVES * RIGID_ROWS) {  // frame-invariant per-row lattice taps of this tile
    const int y = yt + tid < h ? yt + tid : h - 1;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      s_ytap[tid][k] = a.ytap[4 * y + k];
      s_ycoef[tid][k] = a.ycoef[4 * y + k];
    }
  }
  // lattice footprint of the tile (frame-invariant): node rows [R0,R1], node columns [C0,C1]
  int R0, R1, C0, C1;
  {
    int lo = 0x7fffffff, hi = -1;
    if (lane < RIGID_WAVES * RIGID_ROWS) {
      const int y = yt + lane < h ? yt + lane : h - 1;
      for (int k = 0; k < 4; ++k) {
        const int v = a.ytap[4 * y + k];
        lo = v < lo ? v : lo;
        hi = v > hi ? v : hi;
      }
    }
    R0 = wave_min_i(lo);
    R1 = wave_max_i(hi);
    lo = 0x7fffffff;
    hi = -1;
    for (int k = 0; k < 4; ++k) {
      const int x = xt + lane + 64 * k;
      const int xs = x < w ? x : w - 1;
      for (int j = 0; j < 4; ++j) {
        const int v = fa.xtap[4 * xs + j];
        lo = v < lo ? v : lo;
        hi = v > hi ? v : hi;
      }
    }
    C0 = wave_min_i(lo);
    C1 = wave_max_i(hi);
  }
  const int yc = (yt + 16 < h) ? yt + 16 : h - 1;  // centre pixel of the tile (clipped to the image)
  const int xc = (xt + 128 < w) ? xt + 128 : w - 1;
  const int4 ytc = *reinterpret_cast<const int4*>(a.ytap + 4 * yc);
  const float4 ycc = *reinterpret_cast<const float4*>(a.ycoef + 4 * yc);
  const bool whole_tile = yt + RIGID_WAVES * RIGID_ROWS <= h && xt + RIGID_LANES * 4 <= w;
  float acc[RIGID_ROWS][4];
#pragma unroll
  for (int r = 0; r < RIGID_ROWS; ++r)
#pragma unroll
    for (int k = 0; k < 4; ++k) acc[r][k] = 0.f;
  __syncthreads();

  for (int f = 0; f < a.nframes; ++f) {
    const float* fr = a.frames + (int64_t)f * hw;
    const float* E = a.etab + (int64_t)f * 2 * a.GH * w;
    const int64_t chs = (int64_t)a.GH * w;  // channel stride of E
    // 0. regularity -> window margins: range of the lattice nodes that can influence this tile.
    // |shift - shift_centre| <= 3.8 rho (bicubic: sum|w| < 1.9 in 2-D, twice for the centre's own
    // deviation); taps span [-1,+2] around floor(); coordinate rounding adds < 0.01 px.
    int mgy, mgx;
    {
      const float* L = fa.lattice + (int64_t)f * 2 * a.GH * fa.GW;
      const int ncol = C1 - C0 + 1, nnode = (R1 - R0 + 1) * ncol;
      float lo_y = 3.0e38f, hi_y = -3.0e38f, lo_x = 3.0e38f, hi_x = -3.0e38f;
      for (int i = lane; i < nnode; i += RIGID_LANES) {
        const int R = R0 + i / ncol, Cc = C0 + i % ncol;
        const float vy = L[(int64_t)R * fa.GW + Cc], vx = L[(int64_t)(a.GH + R) * fa.GW + Cc];
        lo_y = fminf(lo_y, vy); hi_y = fmaxf(hi_y, vy);
        lo_x = fminf(lo_x, vx); hi_x = fmaxf(hi_x, vx);
      }
      const float ry = 0.5f * (wave_max_f(hi_y) - wave_min_f(lo_y)) / a.pixel_spacing;
      const float rx = 0.5f * (wave_max_f(hi_x) - wave_min_f(lo_x)) / a.pixel_spacing;
      const float ny = 3.8f * ry + 1.05f, nx = 3.8f * rx + 1.05f;
      // NaNs fail the comparison and go to the slow kernel (workgroup-uniform: every wave
      // computed the same numbers)
      if (!((ny <= (float)GW_MG) && (nx <= (float)GW_MG))) {
        if (tid == 0) fa.flags[(int64_t)f * nt + tl] = 1;
        continue;
      }
      mgy = __builtin_amdgcn_readfirstlane((int)ceilf(ny));
      mgx = __builtin_amdgcn_readfirstlane((int)ceilf(nx));
    }
    const int nrows = RIGID_WAVES * RIGID_ROWS + 3 + 2 * mgy;          // <= GW_ROWS
    int nq = (RIGID_LANES * 4 + 6 + 2 * mgx + 3) / 4;                    // <= GW_QUADS
    nq = nq < GW_QUADS ? nq : GW_QUADS;
    // 1. window origin from the shift at the tile centre (identical in every lane)
    int wy0, ax;
    {
      const float* Ec = E + xc;
      float sy = dot4(ycc, Ec[(int64_t)ytc.x * w], Ec[(int64_t)ytc.y * w], Ec[(int64_t)ytc.z * w],
                      Ec[(int64_t)ytc.w * w]);
      float sx = dot4(ycc, Ec[chs + (int64_t)ytc.x * w], Ec[chs + (int64_t)ytc.y * w],
                      Ec[chs + (int64_t)ytc.z * w], Ec[chs + (int64_t)ytc.w * w]);
      if (!UNIT_PS) {
        sy = div_invariant(sy, a.pixel_spacing);
        sx = div_invariant(sx, a.pixel_spacing);
      }
      const float lim = 4.f * (fh + fw);
      const float dy = fminf(fmaxf(floorf(grid_chain((float)yc + sy, fh)) - (float)yc, -lim), lim);
      const float dx = fminf(fmaxf(floorf(grid_chain((float)xc + sx, fw)) - (float)xc, -lim), lim);
      wy0 = __builtin_amdgcn_readfirstlane(yt + (int)dy - 1 - mgy);
      ax = __builtin_amdgcn_readfirstlane((xt + (int)dx - 1 - mgx) & ~3);
    }
    // 2. window -> LDS (the previous frame's reads are behind the barrier at the loop's end); the
    // LDS image keeps the fixed row stride, lanes outside the needed rows / quads issue nothing
    for (int i = wave; i < GW_QUADS_PAD / 64; i += RIGID_WAVES) {
      const int q = i * 64 + lane;
      const int tr = q / GW_QUADS, qc = q - tr * GW_QUADS;
      if (tr < nrows && qc < nq) {
        int r = wy0 + tr;
        r = r < 0 ? 0 : (r > h - 1 ? h - 1 : r);
        int c = ax + 4 * qc;
        c = c < 0 ? 0 : (c > w - 4 ? w - 4 : c);
        __builtin_amdgcn_global_load_lds(fr + (int64_t)r * w + c, (lds_vptr)(tile4 + i * 64), 16, 0, 0);
      }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    const bool interior = whole_tile && wy0 >= 0 && wy0 + nrows <= h && ax >= 0 && ax + 4 * nq <= w;
    if (ax < 0 || ax + 4 * nq > w) {  // border padding: clipped columns (edge tiles only)
      for (int i = tid; i < nrows * GW_STRIDE; i += RIGID_LANES * RIGID_WAVES) {
        const int tr = i / GW_STRIDE, e = i - tr * GW_STRIDE;
        const int c = ax + e;
        if (e < 4 * nq && (c < 0 || c > w - 1)) {
          const int cc = c < 0 ? 0 : w - 1;
          int qsrc = (cc & ~3) - ax;
          qsrc = qsrc < 0 ? 0 : (qsrc > 4 * nq - 4 ? 4 * nq - 4 : qsrc);
          tile[tr * GW_STRIDE + e] = tile[tr * GW_STRIDE + qsrc + (cc & 3)];
        }
      }
      __syncthreads();
    }
    // 3. pixels
    int4 ycache = make_int4(-1, -1, -1, -1);
    float ey[4][4], ex[4][4];  // [lattice tap][pixel k]
    const int oy = 1 + wy0, ox = 1 + ax;
#pragma unroll
    for (int r = 0; r < RIGID_ROWS; ++r) {
      const int y = y0 + r;
      if (y >= h) break;
      // the row tables are frame-invariant: without an opaque index LICM lifts all 8 rows' taps
      // and weights out of the frame loop (64 VGPRs for the whole kernel)
      int row = wave * RIGID_ROWS + r;
      asm volatile("" : "+s"(row));
      const int4 yt4 = make_int4(s_ytap[row][0], s_ytap[row][1], s_ytap[row][2], s_ytap[row][3]);
      const float4 yc4 = make_float4(s_ycoef[row][0], s_ycoef[row][1], s_ycoef[row][2], s_ycoef[row][3]);
      if (yt4.x != ycache.x || yt4.y != ycache.y || yt4.z != ycache.z || yt4.w != ycache.w) {
        ycache = yt4;  // wave-uniform: depends on y only
        const int rows4[4] = {yt4.x, yt4.y, yt4.z, yt4.w};
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
          for (int k = 0; k < 4; ++k) {
            const int x = xt + lane + 64 * k;
            const int xs = x < w ? x : w - 1;
            ey[i][k] = E[(int64_t)rows4[i] * w + xs];
            ex[i][k] = E[chs + (int64_t)rows4[i] * w + xs];
          }
      }
      float* orow = WRITE_FRAMES ? a.out_frames + (int64_t)f * hw + (int64_t)y * w + xt + lane : nullptr;
      if (interior) {
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          if (k % GW2_ILP == 0) __builtin_amdgcn_sched_barrier(0);  // GW2_ILP pixels in flight (register pressure)
          float sy = dot4(yc4, ey[0][k], ey[1][k], ey[2][k], ey[3][k]);
          float sx = dot4(yc4, ex[0][k], ex[1][k], ex[2][k], ex[3][k]);
          if (!UNIT_PS) {
            sy = div_invariant(sy, a.pixel_spacing);
            sx = div_invariant(sx, a.pixel_spacing);
          }
          const float uy = grid_chain((float)y + sy, fh), ux = grid_chain((float)(xt + lane + 64 * k) + sx, fw);
          const float fy = floorf(uy), fx = floorf(ux);
          float wy[4], wx[4];
          cubic_coeffs_factored(uy - fy, wy);
          cubic_coeffs_factored(ux - fx, wx);
          const float* t0 = tile + ((int)fy - oy) * GW_STRIDE + ((int)fx - ox);
          float rowv[4];
#pragma unroll
          for (int i = 0; i < 4; ++i) {
            const float* t = t0 + i * GW_STRIDE;
            rowv[i] = gw_dot4(wx, t[0], t[1], t[2], t[3]);
          }
          const float o = gw_dot4(wy, rowv[0], rowv[1], rowv[2], rowv[3]);
          if (WRITE_FRAMES) orow[64 * k] = o;
          if (WRITE_SUM) acc[r][k] += o;
        }
      } else {
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          __builtin_amdgcn_sched_barrier(0);
          const int x = xt + lane + 64 * k;
          if (x >= w) continue;
          float sy = dot4(yc4, ey[0][k], ey[1][k], ey[2][k], ey[3][k]);
          float sx = dot4(yc4, ex[0][k], ex[1][k], ex[2][k], ex[3][k]);
          if (!UNIT_PS) {
            sy = div_invariant(sy, a.pixel_spacing);
            sx = div_invariant(sx, a.pixel_spacing);
          }
          const float cy = (float)y + sy, cx = (float)x + sx;
          const bool inside = (cy >= 0.f) && (cy <= fh - 1.f) && (cx >= 0.f) && (cx <= fw - 1.f);
          const float uy = grid_chain(cy, fh), ux = grid_chain(cx, fw);
          const float fy = floorf(uy), fx = floorf(ux);
          float wy[4], wx[4];
          cubic_coeffs_factored(uy - fy, wy);
          cubic_coeffs_factored(ux - fx, wx);
          // in range by the regularity test; the clamp only keeps a garbage coordinate from
          // reading outside the LDS tile
          int ly = (int)fy - oy, lx = (int)fx - ox;
          ly = ly < 0 ? 0 : (ly > GW_ROWS - 4 ? GW_ROWS - 4 : ly);
          lx = lx < 0 ? 0 : (lx > GW_STRIDE - 4 ? GW_STRIDE - 4 : lx);
          const float* t0 = tile + ly * GW_STRIDE + lx;
          float rowv[4];
#pragma unroll
          for (int i = 0; i < 4; ++i) {
            const float* t = t0 + i * GW_STRIDE;
            rowv[i] = gw_dot4(wx, t[0], t[1], t[2], t[3]);
          }
          float o = gw_dot4(wy, rowv[0], rowv[1], rowv[2], rowv[3]);
          o = inside ? o : 0.f;
          if (WRITE_FRAMES) orow[64 * k] = o;
          if (WRITE_SUM) acc[r][k] += o;
        }
      }
    }
    __syncthreads();  // everyone is done with the tile before the next frame overwrites it
  }
  if (WRITE_SUM) {
#pragma unroll
    for (int r = 0; r < RIGID_ROWS; ++r) {
      const int y = y0 + r;
      if (y >= h) break;
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const int x = xt + lane + 64 * k;
        if (x < w) a.out_sum[(int64_t)y * w + x] = acc[r][k];  // warp_field_slow adds its tile-frames afterwards
      }
    }
  }
}

// ------------------------------------------------------------------ general warp, third version
// What the counters said about warp_field / warp_field2 (40 x 4092 x 5760, sum only, 3.0 ms): 88
// VALU instructions per pixel, almost all on the 2-cycle pipe, i.e. 1.1 ms of issue -- but a wave
// issued only every ~10 cycles (the per-pixel chain is one long dependency, 8 cycles from a result
// to its use, plus an LDS round trip per pixel) and 3 waves per SIMD were all the registers (32
// partial sums + 32 cached lattice values per lane) and the LDS (one window per workgroup) allowed;
// 37 % of the wave cycles were waits.  So: thread-level parallelism instead of registers.
//  * One workgroup of 16 waves per CU and tile; a wave owns TWO pixel rows (8 partial sums per
//    lane, ~70 VGPRs): 4 waves per SIMD.
//  * The window of frame f+1 is DMA'd into a second LDS buffer while frame f is computed (one
//    barrier per frame, no exposed DMA wait).
//  * The x-upsampled lattice rows the tile needs (E, <= 6 rows x 256 columns x 2 channels) are
//    DMA'd into LDS with the window instead of being cached in registers.
//  * Window origin, margins and the regularity verdict of every (tile, frame) come from a small
//    plan kernel, so no wave ever waits for a dependent global load inside the frame loop.
#define GW3_WAVES 16
typedef const __attribute__((address_space(3))) float* gw3_lds_cfptr;
#ifndef GW3_ILP
#define GW3_ILP 1  // pixels of a lane in flight together (interior tile-frames); 1, 2 and 4 measure the same
#endif
#define GW3_RW (RIGID_WAVES * RIGID_ROWS / GW3_WAVES)  // pixel rows per wave (2)
#define GW3_EROWS 6                                     // lattice rows staged per tile
#define GW3_PLAN_MAX 128                                // frames whose plan entries are kept in LDS
#define GW3_QH 36                                       // 16-byte units (8 fp16 samples) per stage row
#define GW3_STAGE_UNITS ((((GW_ROWS * GW3_QH) + 63) / 64) * 64)

// plan[f * nt + tile] = {wy0, ax, mgy | mgx << 8 | irregular << 16, first staged lattice row R0}
__global__ __launch_bounds__(64) void warp_field_plan(FieldArgs fa, int unit_ps, int half, int4* __restrict__ plan) {
  const WarpArgs& a = fa.w;
  const int nt = a.tiles_x * a.tiles_y;
  const int tl = blockIdx.x, f = blockIdx.y;
  const int tyi = tl / a.tiles_x, txi = tl - tyi * a.tiles_x;
  const int h = a.h, w = a.w;
  const float fh = (float)h, fw = (float)w;
  const int lane = threadIdx.x;
  const int xt = txi * (RIGID_LANES * 4), yt = tyi * (RIGID_WAVES * RIGID_ROWS);
  int R0, R1, C0, C1;
  {
    int lo = 0x7fffffff, hi = -1;
    if (lane < RIGID_WAVES * RIGID_ROWS) {
      const int y = yt + lane < h ? yt + lane : h - 1;
      for (int k = 0; k < 4; ++k) {
        const int v = a.ytap[4 * y + k];
        lo = v < lo ? v : lo;
        hi = v > hi ? v : hi;
      }
    }
    R0 = wave_min_i(lo);
    R1 = wave_max_i(hi);
    lo = 0x7fffffff;
    hi = -1;
    for (int k = 0; k < 4; ++k) {
      const int x = xt + lane + 64 * k;
      const int xs = x < w ? x : w - 1;
      for (int j = 0; j < 4; ++j) {
        const int v = fa.xtap[4 * xs + j];
        lo = v < lo ? v : lo;
        hi = v > hi ? v : hi;
      }
    }
    C0 = wave_min_i(lo);
    C1 = wave_max_i(hi);
  }
  const int yc = (yt + 16 < h) ? yt + 16 : h - 1;
  const int xc = (xt + 128 < w) ? xt + 128 : w - 1;
  const int4 ytc = *reinterpret_cast<const int4*>(a.ytap + 4 * yc);
  const float4 ycc = *reinterpret_cast<const float4*>(a.ycoef + 4 * yc);
  const float* E = a.etab + (int64_t)f * 2 * a.GH * w;
  const int64_t chs = (int64_t)a.GH * w;
  const float* L = fa.lattice + (int64_t)f * 2 * a.GH * fa.GW;
  const int ncol = C1 - C0 + 1, nnode = (R1 - R0 + 1) * ncol;
  float lo_y = 3.0e38f, hi_y = -3.0e38f, lo_x = 3.0e38f, hi_x = -3.0e38f;
  for (int i = lane; i < nnode; i += RIGID_LANES) {
    const int R = R0 + i / ncol, Cc = C0 + i % ncol;
    const float vy = L[(int64_t)R * fa.GW + Cc], vx = L[(int64_t)(a.GH + R) * fa.GW + Cc];
    lo_y = fminf(lo_y, vy); hi_y = fmaxf(hi_y, vy);
    lo_x = fminf(lo_x, vx); hi_x = fmaxf(hi_x, vx);
  }
  const float ry = 0.5f * (wave_max_f(hi_y) - wave_min_f(lo_y)) / a.pixel_spacing;
  const float rx = 0.5f * (wave_max_f(hi_x) - wave_min_f(lo_x)) / a.pixel_spacing;
  const float ny = 3.8f * ry + 1.05f, nx = 3.8f * rx + 1.05f;
  const bool regular = (ny <= (float)GW_MG) && (nx <= (float)GW_MG) && (R1 - R0 + 1 <= GW3_EROWS);
  int4 out = make_int4(0, 0, 1 << 16, R0);
  if (regular) {
    const int mgy = (int)ceilf(ny), mgx = (int)ceilf(nx);
    const float* Ec = E + xc;
    float sy = dot4(ycc, Ec[(int64_t)ytc.x * w], Ec[(int64_t)ytc.y * w], Ec[(int64_t)ytc.z * w],
                    Ec[(int64_t)ytc.w * w]);
    float sx = dot4(ycc, Ec[chs + (int64_t)ytc.x * w], Ec[chs + (int64_t)ytc.y * w],
                    Ec[chs + (int64_t)ytc.z * w], Ec[chs + (int64_t)ytc.w * w]);
    if (!unit_ps) {
      sy = div_invariant(sy, a.pixel_spacing);
      sx = div_invariant(sx, a.pixel_spacing);
    }
    const float lim = 4.f * (fh + fw);
    const float dy = fminf(fmaxf(floorf(grid_chain((float)yc + sy, fh)) - (float)yc, -lim), lim);
    const float dx = fminf(fmaxf(floorf(grid_chain((float)xc + sx, fw)) - (float)xc, -lim), lim);
    // fp32 frames: the window starts at a 16-byte aligned column; fp16 frames: at the exact column
    // (the fp16 -> fp32 staging pass places it, warp_field3)
    const int axx = xt + (int)dx - 1 - mgx;
    out = make_int4(yt + (int)dy - 1 - mgy, half ? axx : (axx & ~3), mgy | (mgx << 8), R0);
  }
  if (lane == 0) {
    plan[(int64_t)f * nt + tl] = out;
    if (!regular) fa.flags[(int64_t)f * nt + tl] = 1;
  }
}

template <bool WRITE_FRAMES, bool WRITE_SUM, bool UNIT_PS, bool HALF>
__global__ __launch_bounds__(RIGID_LANES* GW3_WAVES, 4) void warp_field3(FieldArgs fa, const int4* __restrict__ plan) {
  const WarpArgs& a = fa.w;
  extern __shared__ __attribute__((aligned(16))) char smem_gw[];
  // fp32 frames: [window 0][window 1][E 0][E 1].  fp16 frames: [fp32 window][fp16 stage 0][fp16
  // stage 1][E 0][E 1] -- the DMA lands the raw fp16 window in a stage, one pass per frame widens it
  // into the single fp32 window (10 elements per thread; doing it per tap would be 24 instructions
  // per pixel), so the HBM side moves half the bytes and the arithmetic is unchanged.
  auto win_of = [&](int bi) { return reinterpret_cast<float4*>(smem_gw) + (HALF ? 0 : bi) * GW_QUADS_PAD; };
  auto stage_of = [&](int bi) {
    return reinterpret_cast<float4*>(smem_gw + GW_QUADS_PAD * 16) + bi * GW3_STAGE_UNITS;
  };
  auto est_of = [&](int bi) {
    return reinterpret_cast<float*>(smem_gw + (HALF ? GW_QUADS_PAD * 16 + 2 * GW3_STAGE_UNITS * 16
                                                    : 2 * GW_QUADS_PAD * 16)) + bi * (2 * GW3_EROWS * 256);
  };
  __shared__ int s_ytap[RIGID_WAVES * RIGID_ROWS][4];
  __shared__ float s_ycoef[RIGID_WAVES * RIGID_ROWS][4];
  const int nt = a.tiles_x * a.tiles_y;
  const int b = blockIdx.x;
  int tl = b;
  if ((nt & 7) == 0) tl = (b & 7) * (nt >> 3) + (b >> 3);
  const int tyi = tl / a.tiles_x, txi = tl - tyi * a.tiles_x;
  const int h = a.h, w = a.w;
  const float fh = (float)h, fw = (float)w;
  const int lane = threadIdx.x;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.y);
  const int tid = wave * RIGID_LANES + lane;
  const int xt = txi * (RIGID_LANES * 4);
  const int yt = tyi * (RIGID_WAVES * RIGID_ROWS);
  const int y0 = yt + wave * GW3_RW;
  const int64_t hw = (int64_t)h * w;
  const int64_t chs = (int64_t)a.GH * w;
  if (tid < RIGID_WAVES * RIGID_ROWS) {
    const int y = yt + tid < h ? yt + tid : h - 1;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      s_ytap[tid][k] = a.ytap[4 * y + k];
      s_ycoef[tid][k] = a.ycoef[4 * y + k];
    }
  }
  const bool whole_tile = yt + RIGID_WAVES * RIGID_ROWS <= h && xt + RIGID_LANES * 4 <= w;
  float acc[GW3_RW][4];
#pragma unroll
  for (int r = 0; r < GW3_RW; ++r)
#pragma unroll
    for (int k = 0; k < 4; ++k) acc[r][k] = 0.f;

  // the tile's plan entries of all frames go to LDS once: a per-frame global load would put its
  // latency in front of every frame's DMA
  // (kept in LDS GW3_PLAN_MAX frames at a time: choosing between an LDS and a global load per frame
  // compiles to a flat load that waits for every outstanding DMA)
  __shared__ int4 s_plan[GW3_PLAN_MAX];
  auto stage_plan = [&](int f0) {
    for (int i = tid; f0 + i < a.nframes && i < GW3_PLAN_MAX; i += RIGID_LANES * GW3_WAVES)
      s_plan[i] = plan[(int64_t)(f0 + i) * nt + tl];
  };
  stage_plan(0);
  __syncthreads();
  auto fetch_plan = [&](int f) {
    const int4 p = s_plan[f & (GW3_PLAN_MAX - 1)];
    return make_int4(__builtin_amdgcn_readfirstlane(p.x), __builtin_amdgcn_readfirstlane(p.y),
                     __builtin_amdgcn_readfirstlane(p.z), __builtin_amdgcn_readfirstlane(p.w));
  };
  // The (window row, 16-byte unit) of each DMA unit this thread issues does not depend on the frame:
  // the divisions are done once (per frame they were 13 of the kernel's 98 VALU instructions per pixel)
  constexpr int DMA_UNITS = HALF ? GW3_STAGE_UNITS / 64 : GW_QUADS_PAD / 64;  // wave-units of 64 lanes
  constexpr int DMA_NIT = (DMA_UNITS + GW3_WAVES - 1) / GW3_WAVES;
  constexpr int DMA_QROW = HALF ? GW3_QH : GW_QUADS;
  int dma_tr[DMA_NIT], dma_qc[DMA_NIT];
#pragma unroll
  for (int it = 0; it < DMA_NIT; ++it) {
    const int i = wave + it * GW3_WAVES;
    const int q = i * 64 + lane;
    dma_tr[it] = i < DMA_UNITS ? q / DMA_QROW : 0x7fff;  // beyond the window: never issued
    dma_qc[it] = q - (q / DMA_QROW) * DMA_QROW;
  }
  // window + lattice rows of frame f -> LDS buffer `bi` (nothing for an irregular tile-frame)
  auto dma = [&](int f, const int4 p, int bi) {
    if (p.z >> 16) return;
    const float* fr = a.frames + (int64_t)f * hw;
    const int mgy = p.z & 255, mgx = (p.z >> 8) & 255;
    const int nrows = RIGID_WAVES * RIGID_ROWS + 3 + 2 * mgy;
    int nq = (RIGID_LANES * 4 + 6 + 2 * mgx + 3) / 4;
    nq = nq < GW_QUADS ? nq : GW_QUADS;
    if (HALF) {  // 16-byte units of 8 samples from the 8-aligned column at or left of the window
      const _Float16* frh = reinterpret_cast<const _Float16*>(a.frames) + (int64_t)f * hw;
      const int axa = p.y & ~7;
      const int nqh = (p.y - axa + RIGID_LANES * 4 + 3 + 2 * mgx + 7) / 8;  // <= GW3_QH
#pragma unroll
      for (int it = 0; it < DMA_NIT; ++it) {
        if (dma_tr[it] < nrows && dma_qc[it] < nqh) {
          int r = p.x + dma_tr[it];
          r = r < 0 ? 0 : (r > h - 1 ? h - 1 : r);
          int c = axa + 8 * dma_qc[it];
          c = c < 0 ? 0 : (c > w - 8 ? w - 8 : c);  // clamped units are never read (see widen)
          const unsigned off = __umul24((unsigned)r, (unsigned)w) + (unsigned)c;  // h w < 2^32 (checked by the host)
          __builtin_amdgcn_global_load_lds(frh + off, (lds_vptr)(stage_of(bi) + (wave + it * GW3_WAVES) * 64), 16, 0, 0);
        }
      }
    } else {
#pragma unroll
      for (int it = 0; it < DMA_NIT; ++it) {
        if (dma_tr[it] < nrows && dma_qc[it] < nq) {
          int r = p.x + dma_tr[it];
          r = r < 0 ? 0 : (r > h - 1 ? h - 1 : r);
          int c = p.y + 4 * dma_qc[it];
          c = c < 0 ? 0 : (c > w - 4 ? w - 4 : c);
          const unsigned off = __umul24((unsigned)r, (unsigned)w) + (unsigned)c;
          __builtin_amdgcn_global_load_lds(fr + off, (lds_vptr)(win_of(bi) + (wave + it * GW3_WAVES) * 64), 16, 0, 0);
        }
      }
    }
    if (wave < 2 * GW3_EROWS) {  // one 1 KiB piece per (channel, lattice row)
      const int ch = wave / GW3_EROWS, er = wave - ch * GW3_EROWS;
      int R = p.w + er;
      R = R > a.GH - 1 ? a.GH - 1 : R;
      int c = xt + 4 * lane;
      c = c > w - 4 ? w - 4 : c;  // columns beyond the image are never used
      const float* src = a.etab + (int64_t)f * 2 * chs + ch * chs + (int64_t)R * w + c;
      __builtin_amdgcn_global_load_lds(src, (lds_vptr)(est_of(bi) + (ch * GW3_EROWS + er) * 256), 16, 0, 0);
    }
  };

  // fp16: stage `bi` -> the fp32 window.  Window column j is absolute column p.y + j; border padding
  // = the clipped column, which always lies in an unclamped unit of the stage (w % 8 == 0).
  auto widen = [&](const int4 p, int bi) {
    if (p.z >> 16) return;
    const int mgy = p.z & 255, mgx = (p.z >> 8) & 255;
    const int nrows = RIGID_WAVES * RIGID_ROWS + 3 + 2 * mgy;
    const int ncols = RIGID_LANES * 4 + 3 + 2 * mgx;  // <= GW_STRIDE
    const int axa = p.y & ~7;
    const _Float16* st = reinterpret_cast<const _Float16*>(stage_of(bi));
    float* wn = reinterpret_cast<float*>(win_of(0));
    for (int i = tid; i < nrows * ncols; i += RIGID_LANES * GW3_WAVES) {
      const int tr = i / ncols, j = i - tr * ncols;
      int xa = p.y + j;
      xa = xa < 0 ? 0 : (xa > w - 1 ? w - 1 : xa);
      wn[tr * GW_STRIDE + j] = (float)st[tr * (8 * GW3_QH) + (xa - axa)];
    }
  };

  int4 pc = fetch_plan(0);
  dma(0, pc, 0);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  if (HALF) {
    widen(pc, 0);
    __syncthreads();
  }
  for (int f = 0; f < a.nframes; ++f) {
    const int bi = f & 1;
    int4 pn = pc;
    if (f + 1 < a.nframes) {
      if (((f + 1) & (GW3_PLAN_MAX - 1)) == 0) {  // next block of plan entries (every thread already holds pc)
        __syncthreads();
        stage_plan(f + 1);
        __syncthreads();
      }
      pn = fetch_plan(f + 1);
      dma(f + 1, pn, bi ^ 1);  // lands under this frame's arithmetic
    }
    if (!(pc.z >> 16)) {
      const int wy0 = pc.x, ax = pc.y, mgy = pc.z & 255, mgx = (pc.z >> 8) & 255, R0 = pc.w;
      const int nrows = RIGID_WAVES * RIGID_ROWS + 3 + 2 * mgy;
      int nq = (RIGID_LANES * 4 + 6 + 2 * mgx + 3) / 4;
      nq = nq < GW_QUADS ? nq : GW_QUADS;
      float* const tile = reinterpret_cast<float*>(win_of(bi));
      const float* const es = est_of(bi);
      if (!HALF && (ax < 0 || ax + 4 * nq > w)) {  // border padding: clipped columns (edge tiles only)
        for (int i = tid; i < nrows * GW_STRIDE; i += RIGID_LANES * GW3_WAVES) {
          const int tr = i / GW_STRIDE, e = i - tr * GW_STRIDE;
          const int c = ax + e;
          if (e < 4 * nq && (c < 0 || c > w - 1)) {
            const int cc = c < 0 ? 0 : w - 1;
            int qsrc = (cc & ~3) - ax;
            qsrc = qsrc < 0 ? 0 : (qsrc > 4 * nq - 4 ? 4 * nq - 4 : qsrc);
            tile[tr * GW_STRIDE + e] = tile[tr * GW_STRIDE + qsrc + (cc & 3)];
          }
        }
        __syncthreads();
      }
      const bool interior_rt = whole_tile && wy0 >= 0 && wy0 + nrows <= h && ax >= 0 &&
                               ax + (HALF ? RIGID_LANES * 4 + 3 + 2 * mgx : 4 * nq) <= w;
      const int oy = 1 + wy0, ox = 1 + ax;
      const unsigned tap_base = (unsigned)(uintptr_t)(lds_vptr)tile - 4u * (unsigned)(oy * GW_STRIDE + ox);
      // The tile-frame's two bodies are separate instantiations: the interior one (no zero-outside
      // test, no column predicate) is straight-line code for all of a wave's pixels, so the LDS reads
      // of one pixel are scheduled under the arithmetic of another instead of every pixel ending in an
      // exec-mask branch.
      auto pixels = [&](auto interior_tag) {
        constexpr bool interior = decltype(interior_tag)::value;
      // What the counters say (profiles/r03_warp_field3_pmc.txt, 40 x 4092 x 5760 sum only): 87 VALU
      // instructions per pixel at ~4.4 cycles each with 4 waves per SIMD, LDS address pipe 52 % busy.
      // Measured alternatives, same results, none faster: 2 or 4 pixels of a lane in flight together
      // (GW3_ILP), the wave's two rows statement by statement, the y/x sides of the chain or row pairs
      // of the taps on 2-float vectors (a v_pk_*_f32 instruction costs 1.35-1.6 x a scalar one here,
      // scripts/ubench/pk_rate.hip, and the pairs have to be built with moves: 3.24 ms against 2.98).
      // Removing 11 % of the instructions (frame-invariant DMA indices, v_fract / v_cvt_flr, one tap
      // address) bought 4 % of the time.
      if constexpr (interior) {
        // Interior tile-frames (all but the frame's rim): GW3_ILP pixels of a lane side by side -- their
        // chains (shift, coordinate, weights, address) are independent, so a dependent instruction of one
        // issues behind an instruction of the other -- then all their taps in one batch of LDS reads.
        // Coordinates are positive here: v_fract_f32 IS u - floor(u) (exact either way) and
        // v_cvt_flr_i32_f32 is the floor as an integer; the tap address is one 24-bit multiply-add and one
        // shift-add from a per-frame base that holds the window origin; the 16 taps are single reads with
        // 16-bit immediate offsets from that ONE address (paired into ds_read2_b32, whose 8-bit offsets do
        // not reach the next window row, they cost 6 address adds per pixel).
#pragma unroll
        for (int r = 0; r < GW3_RW; ++r) {
          const int y = y0 + r;
          const int row = wave * GW3_RW + r;
          const float4 yc4 = make_float4(s_ycoef[row][0], s_ycoef[row][1], s_ycoef[row][2], s_ycoef[row][3]);
          const float* e0 = es + (s_ytap[row][0] - R0) * 256 + lane;
          const float* e1 = es + (s_ytap[row][1] - R0) * 256 + lane;
          const float* e2 = es + (s_ytap[row][2] - R0) * 256 + lane;
          const float* e3 = es + (s_ytap[row][3] - R0) * 256 + lane;
          float* orow = WRITE_FRAMES ? a.out_frames + (int64_t)f * hw + (int64_t)y * w + xt + lane : nullptr;
#pragma unroll
          for (int k0 = 0; k0 < 4; k0 += GW3_ILP) {
            float wy[GW3_ILP][4], wx[GW3_ILP][4], tp[GW3_ILP][16];
            gw3_lds_cfptr t0[GW3_ILP];
#pragma unroll
            for (int q = 0; q < GW3_ILP; ++q) {
              const int k = k0 + q;
              float sy = dot4(yc4, e0[64 * k], e1[64 * k], e2[64 * k], e3[64 * k]);
              float sx = dot4(yc4, e0[64 * k + GW3_EROWS * 256], e1[64 * k + GW3_EROWS * 256],
                              e2[64 * k + GW3_EROWS * 256], e3[64 * k + GW3_EROWS * 256]);
              if (!UNIT_PS) {
                sy = div_invariant(sy, a.pixel_spacing);
                sx = div_invariant(sx, a.pixel_spacing);
              }
              const float uy = grid_chain((float)y + sy, fh), ux = grid_chain((float)(xt + lane + 64 * k) + sx, fw);
              cubic_coeffs_factored(__builtin_amdgcn_fractf(uy), wy[q]);
              cubic_coeffs_factored(__builtin_amdgcn_fractf(ux), wx[q]);
              int iy, ix;
              asm("v_cvt_flr_i32_f32 %0, %1" : "=v"(iy) : "v"(uy));
              asm("v_cvt_flr_i32_f32 %0, %1" : "=v"(ix) : "v"(ux));
              t0[q] = (gw3_lds_cfptr)(uintptr_t)(__umul24((unsigned)iy, 4u * GW_STRIDE) + tap_base + ((unsigned)ix << 2));
            }
#define GW3_RD(i, j) "ds_read_b32 %" #i ", %16 offset:%c" #j "\n"
#pragma unroll
            for (int q = 0; q < GW3_ILP; ++q)
              asm volatile(GW3_RD(0, 17) GW3_RD(1, 18) GW3_RD(2, 19) GW3_RD(3, 20) GW3_RD(4, 21) GW3_RD(5, 22) GW3_RD(6, 23)
                           GW3_RD(7, 24) GW3_RD(8, 25) GW3_RD(9, 26) GW3_RD(10, 27) GW3_RD(11, 28) GW3_RD(12, 29)
                           GW3_RD(13, 30) GW3_RD(14, 31) GW3_RD(15, 32)
                           : "=&v"(tp[q][0]), "=&v"(tp[q][1]), "=&v"(tp[q][2]), "=&v"(tp[q][3]), "=&v"(tp[q][4]),
                             "=&v"(tp[q][5]), "=&v"(tp[q][6]), "=&v"(tp[q][7]), "=&v"(tp[q][8]), "=&v"(tp[q][9]),
                             "=&v"(tp[q][10]), "=&v"(tp[q][11]), "=&v"(tp[q][12]), "=&v"(tp[q][13]), "=&v"(tp[q][14]),
                             "=&v"(tp[q][15])
                           : "v"(t0[q]), "n"(0), "n"(4), "n"(8), "n"(12), "n"(4 * GW_STRIDE), "n"(4 * GW_STRIDE + 4),
                             "n"(4 * GW_STRIDE + 8), "n"(4 * GW_STRIDE + 12), "n"(8 * GW_STRIDE), "n"(8 * GW_STRIDE + 4),
                             "n"(8 * GW_STRIDE + 8), "n"(8 * GW_STRIDE + 12), "n"(12 * GW_STRIDE), "n"(12 * GW_STRIDE + 4),
                             "n"(12 * GW_STRIDE + 8), "n"(12 * GW_STRIDE + 12)
                           : "memory");
#undef GW3_RD
            // one wait for the batch; the operand lists tie every tap to it
#pragma unroll
            for (int q = 0; q < GW3_ILP; ++q)
              asm volatile("s_waitcnt lgkmcnt(0)"
                           : "+v"(tp[q][0]), "+v"(tp[q][1]), "+v"(tp[q][2]), "+v"(tp[q][3]), "+v"(tp[q][4]), "+v"(tp[q][5]),
                             "+v"(tp[q][6]), "+v"(tp[q][7]), "+v"(tp[q][8]), "+v"(tp[q][9]), "+v"(tp[q][10]),
                             "+v"(tp[q][11]), "+v"(tp[q][12]), "+v"(tp[q][13]), "+v"(tp[q][14]), "+v"(tp[q][15])
                           :: "memory");
#pragma unroll
            for (int q = 0; q < GW3_ILP; ++q) {
              float rowv[4];
#pragma unroll
              for (int i = 0; i < 4; ++i)
                rowv[i] = gw_dot4(wx[q], tp[q][4 * i], tp[q][4 * i + 1], tp[q][4 * i + 2], tp[q][4 * i + 3]);
              const float o = gw_dot4(wy[q], rowv[0], rowv[1], rowv[2], rowv[3]);
              if (WRITE_FRAMES) orow[64 * (k0 + q)] = o;
              if (WRITE_SUM) acc[r][k0 + q] += o;
            }
          }
        }
      } else
#pragma unroll
      for (int r = 0; r < GW3_RW; ++r) {
        const int y = y0 + r;
        if (!interior && y >= h) break;
        const int row = wave * GW3_RW + r;
        const float4 yc4 = make_float4(s_ycoef[row][0], s_ycoef[row][1], s_ycoef[row][2], s_ycoef[row][3]);
        const float* e0 = es + (s_ytap[row][0] - R0) * 256 + lane;
        const float* e1 = es + (s_ytap[row][1] - R0) * 256 + lane;
        const float* e2 = es + (s_ytap[row][2] - R0) * 256 + lane;
        const float* e3 = es + (s_ytap[row][3] - R0) * 256 + lane;
        float* orow = WRITE_FRAMES ? a.out_frames + (int64_t)f * hw + (int64_t)y * w + xt + lane : nullptr;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          const int x = xt + lane + 64 * k;
          if (!interior && x >= w) continue;
          float sy = dot4(yc4, e0[64 * k], e1[64 * k], e2[64 * k], e3[64 * k]);
          float sx = dot4(yc4, e0[64 * k + GW3_EROWS * 256], e1[64 * k + GW3_EROWS * 256],
                          e2[64 * k + GW3_EROWS * 256], e3[64 * k + GW3_EROWS * 256]);
          if (!UNIT_PS) {
            sy = div_invariant(sy, a.pixel_spacing);
            sx = div_invariant(sx, a.pixel_spacing);
          }
          const float cy = (float)y + sy, cx = (float)x + sx;
          const float uy = grid_chain(cy, fh), ux = grid_chain(cx, fw);
          float wy[4], wx[4];
          float rowv[4];
          const float fy = floorf(uy), fx = floorf(ux);
          cubic_coeffs_factored(uy - fy, wy);
          cubic_coeffs_factored(ux - fx, wx);
          int ly = (int)fy - oy, lx = (int)fx - ox;
          const bool inside = (cy >= 0.f) && (cy <= fh - 1.f) && (cx >= 0.f) && (cx <= fw - 1.f);
          // in range by the regularity test; the clamp only keeps a garbage coordinate from
          // reading outside the LDS tile
          ly = ly < 0 ? 0 : (ly > GW_ROWS - 4 ? GW_ROWS - 4 : ly);
          lx = lx < 0 ? 0 : (lx > GW_STRIDE - 4 ? GW_STRIDE - 4 : lx);
          const float* t0 = tile + ly * GW_STRIDE + lx;
#pragma unroll
          for (int i = 0; i < 4; ++i) {
            const float* t = t0 + i * GW_STRIDE;
            rowv[i] = gw_dot4(wx, t[0], t[1], t[2], t[3]);
          }
          float o = gw_dot4(wy, rowv[0], rowv[1], rowv[2], rowv[3]);
          o = inside ? o : 0.f;
          if (WRITE_FRAMES) orow[64 * k] = o;
          if (WRITE_SUM) acc[r][k] += o;
        }
      }
      };
      if (interior_rt) pixels(std::true_type{});
      else pixels(std::false_type{});
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // DMA of f+1 (and this frame's stores)
    __syncthreads();  // buffer bi is free again, buffer bi^1 is complete
    if (HALF && f + 1 < a.nframes) {
      widen(pn, bi ^ 1);
      __syncthreads();
    }
    pc = pn;
  }
  if (WRITE_SUM) {
#pragma unroll
    for (int r = 0; r < GW3_RW; ++r) {
      const int y = y0 + r;
      if (y >= h) break;
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const int x = xt + lane + 64 * k;
        if (x < w) a.out_sum[(int64_t)y * w + x] = acc[r][k];  // warp_field_slow adds its tile-frames afterwards
      }
    }
  }
}

// Tile-frames warp_field flagged as irregular: generic per-pixel gathers from global
// memory (border padding by clipping every tap coordinate).  One workgroup per tile, so
// the += on out_sum cannot race.
template <bool UNIT_PS, bool HALF = false>
__global__ __launch_bounds__(RIGID_LANES* RIGID_WAVES) void warp_field_slow(FieldArgs fa, int write_frames,
                                                                           int write_sum) {
  const WarpArgs& a = fa.w;
  const int nt = a.tiles_x * a.tiles_y;
  const int tl = blockIdx.x;
  bool any = false;
  for (int f = 0; f < a.nframes; ++f) any = any || fa.flags[(int64_t)f * nt + tl];
  if (!any) return;
  const int tyi = tl / a.tiles_x, txi = tl - tyi * a.tiles_x;
  const int h = a.h, w = a.w;
  const float fh = (float)h, fw = (float)w;
  const int64_t hw = (int64_t)h * w;
  const int64_t chs = (int64_t)a.GH * w;
  const int xt = txi * (RIGID_LANES * 4), yt = tyi * (RIGID_WAVES * RIGID_ROWS);
  for (int i = threadIdx.y * RIGID_LANES + threadIdx.x; i < RIGID_LANES * 4 * RIGID_WAVES * RIGID_ROWS;
       i += RIGID_LANES * RIGID_WAVES) {
    const int y = yt + i / (RIGID_LANES * 4), x = xt + i % (RIGID_LANES * 4);
    if (y >= h || x >= w) continue;
    const int4 yt4 = *reinterpret_cast<const int4*>(a.ytap + 4 * y);
    const float4 yc4 = *reinterpret_cast<const float4*>(a.ycoef + 4 * y);
    float accp = 0.f;
    for (int f = 0; f < a.nframes; ++f) {
      if (!fa.flags[(int64_t)f * nt + tl]) continue;
      const float* fr = a.frames + (int64_t)f * hw;
      const _Float16* frh = reinterpret_cast<const _Float16*>(a.frames) + (int64_t)f * hw;
      const float* E = a.etab + (int64_t)f * 2 * a.GH * w + x;
      float sy = dot4(yc4, E[(int64_t)yt4.x * w], E[(int64_t)yt4.y * w], E[(int64_t)yt4.z * w],
                      E[(int64_t)yt4.w * w]);
      float sx = dot4(yc4, E[chs + (int64_t)yt4.x * w], E[chs + (int64_t)yt4.y * w],
                      E[chs + (int64_t)yt4.z * w], E[chs + (int64_t)yt4.w * w]);
      if (!UNIT_PS) {
        sy = div_invariant(sy, a.pixel_spacing);
        sx = div_invariant(sx, a.pixel_spacing);
      }
      const float cy = (float)y + sy, cx = (float)x + sx;
      const bool inside = (cy >= 0.f) && (cy <= fh - 1.f) && (cx >= 0.f) && (cx <= fw - 1.f);
      const float uy = grid_chain(cy, fh), ux = grid_chain(cx, fw);
      const float fy = floorf(uy), fx = floorf(ux);
      float wy[4], wx[4];
      cubic_coeffs_fast(uy - fy, wy);
      cubic_coeffs_fast(ux - fx, wx);
      float rowv[4];
      for (int ii = 0; ii < 4; ++ii) {
        const float ty = fminf(fmaxf(fy + (float)(ii - 1), 0.f), fh - 1.f);
        const int64_t ro = (int64_t)(int)ty * w;
        float t4[4];
        for (int j = 0; j < 4; ++j) {
          const int64_t o = ro + (int)fminf(fmaxf(fx + (float)(j - 1), 0.f), fw - 1.f);
          t4[j] = HALF ? (float)frh[o] : fr[o];
        }
        rowv[ii] = gw_dot4(wx, t4[0], t4[1], t4[2], t4[3]);
      }
      float o = gw_dot4(wy, rowv[0], rowv[1], rowv[2], rowv[3]);
      o = inside ? o : 0.f;
      if (write_frames) a.out_frames[(int64_t)f * hw + (int64_t)y * w + x] = o;
      accp += o;
    }
    if (write_sum) a.out_sum[(int64_t)y * w + x] += accp;
  }
}

// get_pixel_shifts (correct_motion.py:132-185) for one lattice: out (h, w, 2) px.
__global__ void warp_pixel_shifts(const float* __restrict__ etab, const int* __restrict__ ytap,
                                  const float* __restrict__ ycoef, int h, int w, int GH,
                                  float pixel_spacing, float* __restrict__ out) {
  const int x = blockIdx.x * blockDim.x + threadIdx.x;
  const int y = blockIdx.y;
  if (x >= w) return;
  const int4 yt = *reinterpret_cast<const int4*>(ytap + 4 * y);
  const float4 yc = *reinterpret_cast<const float4*>(ycoef + 4 * y);
  for (int c = 0; c < 2; ++c) {
    const float* E = etab + (int64_t)c * GH * w + x;
    const float s = ((yc.x * E[(int64_t)yt.x * w] + yc.y * E[(int64_t)yt.y * w]) +
                     yc.z * E[(int64_t)yt.z * w]) + yc.w * E[(int64_t)yt.w * w];
    out[((int64_t)y * w + x) * 2 + c] = s / pixel_spacing;
  }
}

// get_pixel_shifts at caller-supplied pixel coordinates (the `pixel_grid` argument,
// correct_motion.py:167-168): coords (n, 2) yx in pixels of an (h, w) frame -> out (n, 2) px.
// Same fp32 chain as warp_axis_tables with (float)p replaced by the given coordinate; x taps
// first, then y (ATen's bicubic grid_sample order), reflection padding per tap.
__global__ void warp_pixel_shifts_at(const float* __restrict__ lattice, int GH, int GW, int h, int w,
                                     float pixel_spacing, const float* __restrict__ coords, int64_t n,
                                     float* __restrict__ out) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  int tap[2][4];
  float coef[2][4];
  for (int axis = 0; axis < 2; ++axis) {
    const int len = axis == 0 ? h : w, G = axis == 0 ? GH : GW;
    const float normalized = coords[2 * i + axis] / (float)(len - 1);
    const float interp = normalized * (float)(G - 1);
    const float u = grid_chain(interp, (float)G);
    const float fl = floorf(u);
    cubic_coeffs(u - fl, coef[axis]);
    // clamp in float first: a far-away coordinate must not overflow the int conversion
    const int i0 = (int)fminf(fmaxf(fl, -1.0e9f), 1.0e9f);
    for (int k = 0; k < 4; ++k) tap[axis][k] = reflect_index(i0 - 1 + k, G);
  }
  for (int c = 0; c < 2; ++c) {
    const float* L = lattice + (int64_t)c * GH * GW;
    float rowv[4];
    for (int ky = 0; ky < 4; ++ky) {
      const float* r = L + (int64_t)tap[0][ky] * GW;
      rowv[ky] = ((coef[1][0] * r[tap[1][0]] + coef[1][1] * r[tap[1][1]]) + coef[1][2] * r[tap[1][2]]) +
                 coef[1][3] * r[tap[1][3]];
    }
    const float sft = ((coef[0][0] * rowv[0] + coef[0][1] * rowv[1]) + coef[0][2] * rowv[2]) + coef[0][3] * rowv[3];
    out[2 * i + c] = sft / pixel_spacing;
  }
}

// ------------------------------------------------------------------ spline lattice
// out[c][it][iy][ix] = sum_kt wt sum_ky wy sum_kx wx * data[c][idx_t][idx_y][idx_x]
// (x innermost, then y, then t -- the separable order of the spline library).
__global__ void spline_lattice_kernel(const float* __restrict__ data, int c, int nt, int nh, int nw,
                                      const int* __restrict__ idx_t, const float* __restrict__ w_t,
                                      int NT, const int* __restrict__ idx_y,
                                      const float* __restrict__ w_y, int NY,
                                      const int* __restrict__ idx_x, const float* __restrict__ w_x,
                                      int NX, float* __restrict__ out) {
  const int64_t total = (int64_t)c * NT * NY * NX;
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= total) return;
  const int ix = (int)(i % NX);
  const int iy = (int)((i / NX) % NY);
  const int it = (int)((i / ((int64_t)NX * NY)) % NT);
  const int ch = (int)(i / ((int64_t)NX * NY * NT));
  const float* d = data + (int64_t)ch * nt * nh * nw;
  float vt = 0.f;
  for (int kt = 0; kt < 4; ++kt) {
    const float* dt = d + (int64_t)idx_t[4 * it + kt] * nh * nw;
    float vy = 0.f;
    for (int ky = 0; ky < 4; ++ky) {
      const float* dy = dt + (int64_t)idx_y[4 * iy + ky] * nw;
      float vx = 0.f;
      for (int kx = 0; kx < 4; ++kx) vx += dy[idx_x[4 * ix + kx]] * w_x[4 * ix + kx];
      vy += vx * w_y[4 * iy + ky];
    }
    vt += vy * w_t[4 * it + kt];
  }
  out[i] = vt;
}

// Spline grid at scattered points: per point 3 x 4 taps (host tables, as for the lattice); same
// summation order as spline_lattice_kernel.  out[i][ch].
__global__ void spline_points_kernel(const float* __restrict__ data, int c, int nt, int nh, int nw,
                                     const int* __restrict__ idx_t, const float* __restrict__ w_t,
                                     const int* __restrict__ idx_y, const float* __restrict__ w_y,
                                     const int* __restrict__ idx_x, const float* __restrict__ w_x,
                                     int64_t npoints, float* __restrict__ out) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= npoints * c) return;
  const int64_t pt = i / c;
  const int ch = (int)(i - pt * c);
  const float* d = data + (int64_t)ch * nt * nh * nw;
  float vt = 0.f;
  for (int kt = 0; kt < 4; ++kt) {
    const float* dt = d + (int64_t)idx_t[4 * pt + kt] * nh * nw;
    float vy = 0.f;
    for (int ky = 0; ky < 4; ++ky) {
      const float* dy = dt + (int64_t)idx_y[4 * pt + ky] * nw;
      float vx = 0.f;
      for (int kx = 0; kx < 4; ++kx) vx += dy[idx_x[4 * pt + kx]] * w_x[4 * pt + kx];
      vy += vx * w_y[4 * pt + ky];
    }
    vt += vy * w_t[4 * pt + kt];
  }
  out[i] = vt;
}

static int64_t field_flag_bytes(int nframes, int h, int w) {
  const int64_t tx = (w + RIGID_LANES * 4 - 1) / (RIGID_LANES * 4);
  const int64_t ty = (h + RIGID_WAVES * RIGID_ROWS - 1) / (RIGID_WAVES * RIGID_ROWS);
  return ((int64_t)nframes * tx * ty + 15) & ~(int64_t)15;
}

static int64_t etab_floats(int nframes, int GH, int w) {
  return (((int64_t)nframes * 2 * GH * w) + 3) & ~(int64_t)3;  // keep the int4 tables aligned
}

extern "C" {

int mc_spline_lattice(const float* data, int c, int nt, int nh, int nw, const int* idx_t,
                      const float* w_t, int NT, const int* idx_y, const float* w_y, int NY,
                      const int* idx_x, const float* w_x, int NX, float* out, void* stream) {
  if (!data || !idx_t || !w_t || !idx_y || !w_y || !idx_x || !w_x || !out) return MC_ERR_ARG;
  if (c < 1 || nt < 1 || nh < 1 || nw < 1 || NT < 1 || NY < 1 || NX < 1) return MC_ERR_ARG;
  const int64_t total = (int64_t)c * NT * NY * NX;
  hipLaunchKernelGGL(spline_lattice_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0,
                     (hipStream_t)stream, data, c, nt, nh, nw, idx_t, w_t, NT, idx_y, w_y, NY,
                     idx_x, w_x, NX, out);
  return mc_check_launch();
}

int mc_spline_points(const float* data, int c, int nt, int nh, int nw, const int* idx_t, const float* w_t,
                     const int* idx_y, const float* w_y, const int* idx_x, const float* w_x, int64_t npoints,
                     float* out, void* stream) {
  if (!data || !idx_t || !w_t || !idx_y || !w_y || !idx_x || !w_x || !out) return MC_ERR_ARG;
  if (c < 1 || nt < 1 || nh < 1 || nw < 1 || npoints < 1) return MC_ERR_ARG;
  const int64_t total = npoints * c;
  hipLaunchKernelGGL(spline_points_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0,
                     (hipStream_t)stream, data, c, nt, nh, nw, idx_t, w_t, idx_y, w_y, idx_x, w_x, npoints, out);
  return mc_check_launch();
}

int mc_warp_scratch_bytes(int nframes, int h, int w, int GH, int GW, int64_t* bytes) {
  if (!bytes || nframes < 1 || h < 2 || w < 2 || GH < 1 || GW < 1) return MC_ERR_ARG;
  // etab floats + (ytap,ycoef,xtap,xcoef) + one flag byte and one 16-byte plan entry per
  // (frame, 256x32 tile)
  *bytes = (etab_floats(nframes, GH, w) + 8 * (int64_t)(h + w)) * 4 + 17 * field_flag_bytes(nframes, h, w);
  return MC_OK;
}

int mc_warp_frames(const float* frames, int nframes, int h, int w, const float* lattice, int GH,
                   int GW, float pixel_spacing, float* scratch, float* out_frames, float* out_sum,
                   void* stream) {
  return mc_warp_frames_t(frames, MC_STORE_F32, nframes, h, w, lattice, GH, GW, pixel_spacing, scratch,
                          out_frames, out_sum, stream);
}

int mc_warp_frames_t(const void* frames_any, int storage, int nframes, int h, int w, const float* lattice,
                     int GH, int GW, float pixel_spacing, float* scratch, float* out_frames, float* out_sum,
                     void* stream) {
  if (storage != MC_STORE_F32 && storage != MC_STORE_F16) return MC_ERR_UNSUPPORTED;
  const bool half = storage == MC_STORE_F16;
  // fp16 frames take the LDS-staged kernel only: 16-byte rows of 8 samples and the reference's sparse
  // lattice (10 nodes per patch); anything else is MC_ERR_UNSUPPORTED and the caller widens the stack
  if (half && ((w % 8) || (((uintptr_t)frames_any) & 15) || (int64_t)32 * (GH - 1) * 2 > (int64_t)3 * (h - 1)))
    return MC_ERR_UNSUPPORTED;
  const float* frames = static_cast<const float*>(frames_any);
  if (!frames || !lattice || !scratch || (!out_frames && !out_sum)) return MC_ERR_ARG;
  if (nframes < 1 || h < 2 || w < 2 || GH < 1 || GW < 1 || !(pixel_spacing > 0.f)) return MC_ERR_ARG;
  hipStream_t s = (hipStream_t)stream;
  float* etab = scratch;
  if (((uintptr_t)scratch) & 15) return MC_ERR_ARG;
  int* ytap = reinterpret_cast<int*>(scratch + etab_floats(nframes, GH, w));
  float* ycoef = reinterpret_cast<float*>(ytap + 4 * (int64_t)h);
  int* xtap = reinterpret_cast<int*>(ycoef + 4 * (int64_t)h);
  float* xcoef = reinterpret_cast<float*>(xtap + 4 * (int64_t)w);
  hipLaunchKernelGGL(warp_axis_tables, dim3((h + 255) / 256), dim3(256), 0, s, h, GH, ytap, ycoef);
  hipLaunchKernelGGL(warp_axis_tables, dim3((w + 255) / 256), dim3(256), 0, s, w, GW, xtap, xcoef);
  hipLaunchKernelGGL(warp_etab, dim3((w + 255) / 256, nframes * 2 * GH), dim3(256), 0, s, lattice,
                     GH, GW, w, xtap, xcoef, etab);
  WarpArgs a;
  a.frames = frames; a.nframes = nframes; a.h = h; a.w = w; a.GH = GH; a.etab = etab;
  a.ytap = ytap; a.ycoef = ycoef; a.pixel_spacing = pixel_spacing;
  a.out_frames = out_frames; a.out_sum = out_sum;
  const bool unit = (pixel_spacing == 1.0f);
  if ((w % 4 == 0) && ((((uintptr_t)frames) & 15) == 0)) {
    a.tiles_x = (w + RIGID_LANES * 4 - 1) / (RIGID_LANES * 4);
    a.tiles_y = (h + RIGID_WAVES * RIGID_ROWS - 1) / (RIGID_WAVES * RIGID_ROWS);
    FieldArgs fa;
    fa.w = a;
    fa.lattice = lattice;
    fa.xtap = xtap;
    fa.GW = GW;
    fa.flags = reinterpret_cast<unsigned char*>(xcoef + 4 * (int64_t)w);
    hipError_t e = hipMemsetAsync(fa.flags, 0, (size_t)field_flag_bytes(nframes, h, w), s);
    if (e != hipSuccess) return (int)e;
    dim3 grid(a.tiles_x * a.tiles_y), block(RIGID_LANES, RIGID_WAVES);
    const size_t lds = (size_t)GW_QUADS_PAD * 16;
    int field_version = 3;  // 3: warp_field3; dense lattices fall back to warp_field2
#ifdef MC_EXPERIMENTS
    if (const char* v = getenv("MC_WARP_FIELD")) field_version = atoi(v);  // 1: the first LDS-tile kernel
#endif
    // version 3 stages <= GW3_EROWS lattice rows per tile: 32 pixel rows must span <= 1.5 lattice
    // cells (always for the reference's 10 nodes per patch; not for a per-pixel lattice)
    // (warp_field3 addresses a frame with 32-bit element offsets built by 24-bit multiplies)
    const bool small32 = h < (1 << 24) && w < (1 << 24) && (int64_t)h * w < ((int64_t)1 << 31);
    if ((field_version == 3 || half) && small32 && (int64_t)32 * (GH - 1) * 2 <= (int64_t)3 * (h - 1)) {
      int4* plan = reinterpret_cast<int4*>(fa.flags + field_flag_bytes(nframes, h, w));
      hipLaunchKernelGGL(warp_field_plan, dim3(a.tiles_x * a.tiles_y, nframes), dim3(64), 0, s, fa, unit ? 1 : 0,
                         half ? 1 : 0, plan);
      const size_t lds3 = (half ? (size_t)GW_QUADS_PAD * 16 + (size_t)2 * GW3_STAGE_UNITS * 16
                                : (size_t)2 * GW_QUADS_PAD * 16) + (size_t)2 * 2 * GW3_EROWS * 256 * 4;
      dim3 block3(RIGID_LANES, GW3_WAVES);
#define MC_GW3_GO(F, S, U, H)                                                                     \
  do {                                                                                            \
    auto k = warp_field3<F, S, U, H>;                                                             \
    (void)hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds3); \
    hipLaunchKernelGGL(k, grid, block3, lds3, s, fa, (const int4*)plan);                          \
  } while (0)
#define MC_GW3_LAUNCH(F, S)                             \
  do {                                                  \
    if (unit && half) MC_GW3_GO(F, S, true, true);      \
    else if (unit) MC_GW3_GO(F, S, true, false);        \
    else if (half) MC_GW3_GO(F, S, false, true);        \
    else MC_GW3_GO(F, S, false, false);                 \
  } while (0)
      if (out_frames && out_sum) MC_GW3_LAUNCH(true, true);
      else if (out_frames) MC_GW3_LAUNCH(true, false);
      else MC_GW3_LAUNCH(false, true);
#undef MC_GW3_LAUNCH
#undef MC_GW3_GO
#define MC_SLOW_GO(U, H) hipLaunchKernelGGL((warp_field_slow<U, H>), grid, block, 0, s, fa, out_frames ? 1 : 0, out_sum ? 1 : 0)
      if (unit && half) MC_SLOW_GO(true, true);
      else if (unit) MC_SLOW_GO(true, false);
      else if (half) MC_SLOW_GO(false, true);
      else MC_SLOW_GO(false, false);
#undef MC_SLOW_GO
      return mc_check_launch();
    }
    if (half) return MC_ERR_UNSUPPORTED;
#ifdef MC_EXPERIMENTS
#define MC_GW_LAUNCH(F, S)                                                                  \
  do {                                                                                      \
    if (field_version == 1) {  /* (version 3 falls back to 2 for dense lattices) */          \
      if (unit) hipLaunchKernelGGL((warp_field<F, S, true>), grid, block, lds, s, fa);      \
      else hipLaunchKernelGGL((warp_field<F, S, false>), grid, block, lds, s, fa);          \
    } else {                                                                                \
      if (unit) hipLaunchKernelGGL((warp_field2<F, S, true>), grid, block, lds, s, fa);     \
      else hipLaunchKernelGGL((warp_field2<F, S, false>), grid, block, lds, s, fa);         \
    }                                                                                       \
  } while (0)
#else
#define MC_GW_LAUNCH(F, S)                                                                  \
  do {                                                                                      \
    if (unit) hipLaunchKernelGGL((warp_field2<F, S, true>), grid, block, lds, s, fa);       \
    else hipLaunchKernelGGL((warp_field2<F, S, false>), grid, block, lds, s, fa);           \
  } while (0)
#endif
    if (out_frames && out_sum) MC_GW_LAUNCH(true, true);
    else if (out_frames) MC_GW_LAUNCH(true, false);
    else MC_GW_LAUNCH(false, true);
#undef MC_GW_LAUNCH
    if (unit) hipLaunchKernelGGL((warp_field_slow<true>), grid, block, 0, s, fa, out_frames ? 1 : 0, out_sum ? 1 : 0);
    else hipLaunchKernelGGL((warp_field_slow<false>), grid, block, 0, s, fa, out_frames ? 1 : 0, out_sum ? 1 : 0);
    return mc_check_launch();
  }
  // rows that are not whole float4 quads (or an unaligned stack): the first, untiled kernel.  It reads
  // fp32 only -- an fp16 stack never gets here (its rows are whole 8-sample units, checked above), and
  // must not: the kernel would read twice the buffer's bytes
  if (half) return MC_ERR_UNSUPPORTED;
  a.tiles_x = (w + WARP_TX * WARP_PX - 1) / (WARP_TX * WARP_PX);
  a.tiles_y = (h + WARP_TY * WARP_ROWS - 1) / (WARP_TY * WARP_ROWS);
  dim3 grid(a.tiles_x * a.tiles_y), block(WARP_TX, WARP_TY);
#define MC_WARP_LAUNCH(F, S)                                                     \
  do {                                                                           \
    if (unit) hipLaunchKernelGGL((warp_main<F, S, true>), grid, block, 0, s, a); \
    else hipLaunchKernelGGL((warp_main<F, S, false>), grid, block, 0, s, a);     \
  } while (0)
  if (out_frames && out_sum) MC_WARP_LAUNCH(true, true);
  else if (out_frames) MC_WARP_LAUNCH(true, false);
  else MC_WARP_LAUNCH(false, true);
#undef MC_WARP_LAUNCH
  return mc_check_launch();
}

int mc_pixel_shifts(const float* lattice, int GH, int GW, int h, int w, float pixel_spacing,
                    float* scratch, float* out, void* stream) {
  if (!lattice || !scratch || !out || h < 2 || w < 2 || GH < 1 || GW < 1 || !(pixel_spacing > 0.f))
    return MC_ERR_ARG;
  if (((uintptr_t)scratch) & 15) return MC_ERR_ARG;
  hipStream_t s = (hipStream_t)stream;
  float* etab = scratch;
  int* ytap = reinterpret_cast<int*>(scratch + etab_floats(1, GH, w));
  float* ycoef = reinterpret_cast<float*>(ytap + 4 * (int64_t)h);
  int* xtap = reinterpret_cast<int*>(ycoef + 4 * (int64_t)h);
  float* xcoef = reinterpret_cast<float*>(xtap + 4 * (int64_t)w);
  hipLaunchKernelGGL(warp_axis_tables, dim3((h + 255) / 256), dim3(256), 0, s, h, GH, ytap, ycoef);
  hipLaunchKernelGGL(warp_axis_tables, dim3((w + 255) / 256), dim3(256), 0, s, w, GW, xtap, xcoef);
  hipLaunchKernelGGL(warp_etab, dim3((w + 255) / 256, 2 * GH), dim3(256), 0, s, lattice, GH, GW, w,
                     xtap, xcoef, etab);
  hipLaunchKernelGGL(warp_pixel_shifts, dim3((w + 255) / 256, h), dim3(256), 0, s, etab, ytap, ycoef,
                     h, w, GH, pixel_spacing, out);
  return mc_check_launch();
}

int mc_pixel_shifts_at(const float* lattice, int GH, int GW, int h, int w, float pixel_spacing,
                       const float* coords_yx, int64_t n, float* out, void* stream) {
  if (!lattice || !coords_yx || !out || h < 2 || w < 2 || GH < 1 || GW < 1 || n < 1 || !(pixel_spacing > 0.f))
    return MC_ERR_ARG;
  hipLaunchKernelGGL(warp_pixel_shifts_at, dim3((unsigned)((n + 255) / 256)), dim3(256), 0,
                     (hipStream_t)stream, lattice, GH, GW, h, w, pixel_spacing, coords_yx, n, out);
  return mc_check_launch();
}

int mc_warp_rigid(const float* frames, int nframes, int h, int w, const float* shifts_px,
                  float* scratch, float* out_frames, float* out_sum, void* stream);

int mc_warp_rigid_scratch_bytes(int nframes, int h, int w, int64_t* bytes) {
  if (!bytes || nframes < 1 || h < 2 || w < 2) return MC_ERR_ARG;
  *bytes = ((int64_t)nframes * 5 * (h + w) + 2 * (int64_t)nframes + 8) * 4;
  return MC_OK;
}

// phase 0: weight tables + resampling (mc_warp_rigid); 1: tables only; 2: resampling only, the
// tables of an earlier phase-1 call with the same arguments are in `scratch`
static int warp_rigid_impl(const void* frames_any, int storage, int nframes, int h, int w, const float* shifts_px,
                           float* scratch, float* out_frames, float* out_sum, int phase, void* stream);

int mc_warp_rigid_phase(const float* frames, int nframes, int h, int w, const float* shifts_px,
                        float* scratch, float* out_frames, float* out_sum, int phase, void* stream) {
  return warp_rigid_impl(frames, MC_STORE_F32, nframes, h, w, shifts_px, scratch, out_frames, out_sum, phase, stream);
}

int mc_warp_rigid_phase_t(const void* frames, int storage, int nframes, int h, int w, const float* shifts_px,
                          float* scratch, float* out_frames, float* out_sum, int phase, void* stream) {
  return warp_rigid_impl(frames, storage, nframes, h, w, shifts_px, scratch, out_frames, out_sum, phase, stream);
}

static int warp_rigid_impl(const void* frames_any, int storage, int nframes, int h, int w, const float* shifts_px,
                           float* scratch, float* out_frames, float* out_sum, int phase, void* stream) {
  const float* frames = static_cast<const float*>(frames_any);
  if (storage != MC_STORE_F32 && storage != MC_STORE_F16) return MC_ERR_UNSUPPORTED;
  // fp16 frames: only the LDS-DMA kernel's 512 x 32 geometry, rows of whole 8-sample units
  if (storage == MC_STORE_F16 && ((w % 8) != 0 || (((uintptr_t)frames_any) & 15) ||
                                  (out_frames && (((uintptr_t)out_frames) & 15))))
    return MC_ERR_UNSUPPORTED;
  if (!frames || !shifts_px || !scratch || (phase != 1 && !out_frames && !out_sum)) return MC_ERR_ARG;  // phase 1 writes no image
  if (nframes < 1 || h < 2 || w < 2 || (((uintptr_t)scratch) & 15) || phase < 0 || phase > 2) return MC_ERR_ARG;
  hipStream_t s = (hipStream_t)stream;
  float* Wy = scratch;
  float* Wx = Wy + (int64_t)nframes * 5 * h;
  int* S = reinterpret_cast<int*>(Wx + (int64_t)nframes * 5 * w);
  const int n = h > w ? h : w;
  dim3 tg((n + 255) / 256, nframes, 2);
  if (phase != 2) {
    hipLaunchKernelGGL(rigid_base, dim3(nframes, 2), dim3(256), 0, s, shifts_px, nframes, h, w, S);
    hipLaunchKernelGGL(rigid_weights, tg, dim3(256), 0, s, shifts_px, nframes, h, w, S, Wy, Wx);
    if (phase == 1) return mc_check_launch();
  }
  RigidArgs a;
  a.frames = frames; a.nframes = nframes; a.h = h; a.w = w; a.S = S; a.Wy = Wy; a.Wx = Wx;
  a.out_frames = out_frames; a.out_sum = out_sum;
  // 512 x 32 tiles, single-buffered, 2 workgroups of 8 waves per CU.  Other tile shapes and the
  // double-buffered form exist in -DMC_EXPERIMENTS builds only (MC_RIGID_GEOM = "WXWY" as two digits,
  // MC_RIGID_NBUF, MC_RIGID_DMA=0; measured in DESIGN.md section 4: none is faster)
  int use_dma = 1, geom = 24, nbuf = 1;
#ifdef MC_EXPERIMENTS
  if (const char* v = getenv("MC_RIGID_DMA")) use_dma = atoi(v);
  if (const char* v = getenv("MC_RIGID_GEOM")) geom = atoi(v);
  if (const char* v = getenv("MC_RIGID_NBUF")) nbuf = atoi(v);
#endif
  (void)nbuf;
  const bool dma_ok = use_dma && (w % 4 == 0) && ((((uintptr_t)frames) & 15) == 0) &&
                      (!out_frames || ((((uintptr_t)out_frames) & 15) == 0));
  const int WX = dma_ok ? geom / 10 : 1, WY = dma_ok ? geom % 10 : RIGID_WAVES;
  a.tiles_x = (w + RIGID_LANES * 4 * WX - 1) / (RIGID_LANES * 4 * WX);
  a.tiles_y = (h + WY * RIGID_ROWS - 1) / (WY * RIGID_ROWS);
  // Without the fused sum every frame is its own block: blocks are dispatched frame-major, so the
  // resident ones always work on neighbouring tiles of ONE frame and halo rows / shared 128-byte
  // lines hit in L2 (FETCH_SIZE 2.76 GB for 2.68 GB of frames).  With the sum a block keeps its
  // tile's partial sums in registers over all frames; blocks drift apart in time and the same
  // halos miss (3.7 GB) -- measured 0.97 ms vs 1.28 ms at 40 x 4096^2.
  a.frames_in_grid = out_sum ? 0 : 1;
  dim3 grid(a.tiles_x * a.tiles_y, a.frames_in_grid ? (nframes + a.frames_in_grid - 1) / a.frames_in_grid : 1),
      block(RIGID_LANES, WX * WY);
  if (storage == MC_STORE_F16) {
    constexpr int HX = 2, HY = 4;  // 512 x 32 tiles
    a.tiles_x = (w + RIGID_LANES * 4 * HX - 1) / (RIGID_LANES * 4 * HX);
    a.tiles_y = (h + HY * RIGID_ROWS - 1) / (HY * RIGID_ROWS);
    dim3 gridh(a.tiles_x * a.tiles_y, a.frames_in_grid ? (nframes + a.frames_in_grid - 1) / a.frames_in_grid : 1),
        blockh(RIGID_LANES, HX * HY);
    const size_t ldsh = (size_t)((((HY * RIGID_ROWS + 4) * (HX * 32 + 1)) + 63) / 64) * 64 * 16;
#define MC_RDH_GO(F, S)                                                                              \
  do {                                                                                               \
    auto k = warp_rigid_dma_h<F, S, HX, HY>;                                                          \
    (void)hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)ldsh); \
    hipLaunchKernelGGL(k, gridh, blockh, ldsh, s, a);                                                \
  } while (0)
    if (out_frames && out_sum) MC_RDH_GO(true, true);
    else if (out_frames) MC_RDH_GO(true, false);
    else MC_RDH_GO(false, true);
#undef MC_RDH_GO
    return mc_check_launch();
  }
#ifdef MC_EXPERIMENTS
  static int use_ls = -1;
  if (use_ls < 0) {
    const char* v = getenv("MC_RIGID_LS");
    use_ls = v ? atoi(v) : 0;
  }
  if (dma_ok && use_ls && out_sum && (out_frames || use_ls > 1)) {
    // fused sum: one loader wave + 8 compute waves per 512 x 64 tile (warp_rigid_ls)
    a.tiles_x = (w + RLS_TW - 1) / RLS_TW;
    a.tiles_y = (h + RLS_TH - 1) / RLS_TH;
    a.frames_in_grid = 0;
    static int nload = -1;
    if (nload < 0) {
      const char* v = getenv("MC_RIGID_NLOAD");
      nload = v ? atoi(v) : 2;  // one wave's 63 outstanding DMAs (vmcnt) are not enough: 1.28 / 1.11 / 1.13 ms at 1 / 2 / 4
    }
    const dim3 gl(a.tiles_x * a.tiles_y), bl(RIGID_LANES, RLS_NC + nload);
#ifdef MC_RIGID_STAMP
    unsigned long long z[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    (void)hipMemcpyToSymbol(HIP_SYMBOL(g_rigid_stamps), z, sizeof z);
#endif
#define MC_RLS_GO(F, NL)                                                                                      \
  do {                                                                                                        \
    auto k = warp_rigid_ls<F, true, NL>;                                                                       \
    (void)hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, RLS_LDS_BYTES);      \
    hipLaunchKernelGGL(k, gl, bl, RLS_LDS_BYTES, s, a);                                                       \
  } while (0)
    if (out_frames) {
      if (nload == 1) MC_RLS_GO(true, 1);
      else if (nload == 2) MC_RLS_GO(true, 2);
      else MC_RLS_GO(true, 4);
    } else {
      if (nload == 1) MC_RLS_GO(false, 1);
      else if (nload == 2) MC_RLS_GO(false, 2);
      else MC_RLS_GO(false, 4);
    }
#undef MC_RLS_GO
#ifdef MC_RIGID_STAMP
    (void)hipStreamSynchronize(s);
    (void)hipMemcpyFromSymbol(z, HIP_SYMBOL(g_rigid_stamps), sizeof z);
    if (z[5] && z[6]) {
      const double dc = (double)z[5] * nframes * 2, dl = (double)z[6] * nframes * 2;
      fprintf(stderr, "rigid_ls stamps (cycles per wave and UNIT): compute+stores %.0f  compute-barrier %.0f | loader: issue %.0f  vmcnt-wait %.0f  barrier %.0f\n",
              z[0] / dc, z[1] / dc, z[2] / dl, z[3] / dl, z[4] / dl);
    }
#endif
    return mc_check_launch();
  }
#endif  // MC_EXPERIMENTS
  if (dma_ok) {
#define MC_RD_GO(F, S, NB, GX, GY)                                                                  \
  do {                                                                                              \
    auto k = warp_rigid_dma<F, S, NB, GX, GY>;                                                       \
    const size_t lds = (size_t)NB * ((((GY * RIGID_ROWS + 4) * (GX * RIGID_LANES + 4)) + 63) / 64) * 64 * 16; \
    (void)hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); \
    hipLaunchKernelGGL(k, grid, block, lds, s, a);                                                  \
  } while (0)
#ifdef MC_EXPERIMENTS
#define MC_RD_GEOM(F, S, NB)                                   \
  do {                                                         \
    if (geom == 14) MC_RD_GO(F, S, NB, 1, 4);                  \
    else if (geom == 22) MC_RD_GO(F, S, NB, 2, 2);             \
    else if (geom == 24) MC_RD_GO(F, S, NB, 2, 4);             \
    else if (geom == 23) MC_RD_GO(F, S, NB, 2, 3);             \
    else if (geom == 43) MC_RD_GO(F, S, NB, 4, 3);             \
    else if (geom == 41) MC_RD_GO(F, S, NB, 4, 1);             \
    else if (geom == 42) MC_RD_GO(F, S, NB, 4, 2);             \
    else if (geom == 18) MC_RD_GO(F, S, NB, 1, 8);             \
    else if (geom == 16) MC_RD_GO(F, S, NB, 1, 6);             \
    else if (geom == 12) MC_RD_GO(F, S, NB, 1, 2);             \
    else if (geom == 28) MC_RD_GO(F, S, NB, 2, 8);             \
    else if (geom == 26) MC_RD_GO(F, S, NB, 2, 6);             \
    else return MC_ERR_UNSUPPORTED;                                   \
  } while (0)
#else
#define MC_RD_GEOM(F, S, NB) MC_RD_GO(F, S, NB, 2, 4)  /* 512 x 32 tiles, 2 workgroups of 8 waves per CU */
#endif
#ifdef MC_EXPERIMENTS
#define MC_RD_LAUNCH(F, S)                                     \
  do {                                                         \
    if (nbuf == 1) MC_RD_GEOM(F, S, 1);                        \
    else MC_RD_GEOM(F, S, 2);                                  \
  } while (0)
#else
#define MC_RD_LAUNCH(F, S) MC_RD_GEOM(F, S, 1)
#endif
#ifdef MC_RIGID_STAMP
    unsigned long long z[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    (void)hipMemcpyToSymbol(HIP_SYMBOL(g_rigid_stamps), z, sizeof z);
#endif
    if (out_frames && out_sum) MC_RD_LAUNCH(true, true);
    else if (out_frames) MC_RD_LAUNCH(true, false);
    else MC_RD_LAUNCH(false, true);
#ifdef MC_RIGID_STAMP
    (void)hipStreamSynchronize(s);
    (void)hipMemcpyFromSymbol(z, HIP_SYMBOL(g_rigid_stamps), sizeof z);
    if (z[5]) {
      const double d = (double)z[5] * (nframes - 1);
      fprintf(stderr, "rigid stamps (cycles per wave and frame): compute %.0f  barrier1 %.0f  dma-issue %.0f  vmcnt-wait %.0f  barrier2 %.0f  waves %llu frames_out=%d sum=%d\n",
              z[0] / d, z[1] / d, z[2] / d, z[3] / d, z[4] / d, z[5], out_frames != nullptr, out_sum != nullptr);
    }
#endif
#undef MC_RD_LAUNCH
#undef MC_RD_GEOM
#undef MC_RD_GO
    return mc_check_launch();
  }
  if (out_frames && out_sum) hipLaunchKernelGGL((warp_rigid<true, true>), grid, block, 0, s, a);
  else if (out_frames) hipLaunchKernelGGL((warp_rigid<true, false>), grid, block, 0, s, a);
  else hipLaunchKernelGGL((warp_rigid<false, true>), grid, block, 0, s, a);
  return mc_check_launch();
}

// The movie pipeline's tail in two launches (rigid_tail + rigid_weights): integer-peak shifts (t,2) px ->
// field (2,t) Angstrom, the warp's shifts_px (t,2) and its weight tables in `scratch` (the layout of
// mc_warp_rigid_phase, phase 1).  idx_t / w_t: the 4 time taps per frame of the field's spline
// (spline.axis_taps(t, linspace(0,1,t))), w_y / w_x: the taps of lattice point 0 on a 1-sample axis.
int mc_rigid_tables_from_shifts(const float* shifts, float pixel_spacing, const int* idx_t, const float* w_t,
                                const float* w_y, const float* w_x, int nframes, int h, int w, float* field,
                                float* shifts_px, float* scratch, void* stream) {
  if (!shifts || !idx_t || !w_t || !w_y || !w_x || !field || !shifts_px || !scratch) return MC_ERR_ARG;
  if (nframes < 1 || h < 2 || w < 2 || !(pixel_spacing > 0.f) || (((uintptr_t)scratch) & 15)) return MC_ERR_ARG;
  hipStream_t s = (hipStream_t)stream;
  float* Wy = scratch;
  float* Wx = Wy + (int64_t)nframes * 5 * h;
  int* S = reinterpret_cast<int*>(Wx + (int64_t)nframes * 5 * w);
  const int n = h > w ? h : w;
  hipLaunchKernelGGL(rigid_tail, dim3(nframes, 2), dim3(256), 0, s, shifts, pixel_spacing, idx_t, w_t, w_y, w_x, nframes,
                     h, w, field, shifts_px, S);
  hipLaunchKernelGGL(rigid_weights, dim3((n + 255) / 256, nframes, 2), dim3(256), 0, s, (const float*)shifts_px, nframes,
                     h, w, (const int*)S, Wy, Wx);
  return mc_check_launch();
}

// N2: the rigid warp straight from a raw u8 / i16 movie + gain reference (warp_rigid_raw); phase as in
// mc_warp_rigid_phase.  Shapes it has no kernel for (w % 4, unaligned buffers): MC_ERR_UNSUPPORTED -- the
// caller conditions the movie into an fp32 copy first.
int mc_warp_rigid_raw(const void* raw, int storage, const float* gain, const float* mu, int nframes, int h, int w,
                      const float* shifts_px, float* scratch, float* out_frames, float* out_sum, int phase,
                      void* stream) {
  if (storage != MC_STORE_U8 && storage != MC_STORE_I16) return MC_ERR_UNSUPPORTED;
  if (!raw || !gain || !mu || !shifts_px || !scratch || (phase != 1 && !out_frames && !out_sum)) return MC_ERR_ARG;
  if (nframes < 1 || h < 2 || w < 2 || (((uintptr_t)scratch) & 15) || phase < 0 || phase > 2) return MC_ERR_ARG;
  if ((w % 4) || (((uintptr_t)raw) & 15) || (((uintptr_t)gain) & 15) || (out_frames && (((uintptr_t)out_frames) & 15)) ||
      w < 16 || nframes > RR_PAR_MAX)
    return MC_ERR_UNSUPPORTED;
  hipStream_t s = (hipStream_t)stream;
  float* Wy = scratch;
  float* Wx = Wy + (int64_t)nframes * 5 * h;
  int* S = reinterpret_cast<int*>(Wx + (int64_t)nframes * 5 * w);
  const int n = h > w ? h : w;
  if (phase != 2) {
    hipLaunchKernelGGL(rigid_base, dim3(nframes, 2), dim3(256), 0, s, shifts_px, nframes, h, w, S);
    hipLaunchKernelGGL(rigid_weights, dim3((n + 255) / 256, nframes, 2), dim3(256), 0, s, shifts_px, nframes, h, w, S,
                       Wy, Wx);
    if (phase == 1) return mc_check_launch();
  }
  RigidRawArgs ra;
  ra.r.frames = static_cast<const float*>(raw); ra.r.nframes = nframes; ra.r.h = h; ra.r.w = w;
  ra.r.S = S; ra.r.Wy = Wy; ra.r.Wx = Wx; ra.r.out_frames = out_frames; ra.r.out_sum = out_sum;
  ra.r.tiles_x = (w + 511) / 512; ra.r.tiles_y = (h + 31) / 32; ra.r.frames_in_grid = 0;
  ra.gain = gain; ra.mu = mu;
  const dim3 grid(ra.r.tiles_x * ra.r.tiles_y), block(RIGID_LANES, 8);
#define MC_RAW_GO(F, SM, K) hipLaunchKernelGGL((warp_rigid_raw<F, SM, K>), grid, block, 0, s, ra) /* static LDS */
#define MC_RAW_MODE(K)                                     \
  do {                                                     \
    if (out_frames && out_sum) MC_RAW_GO(true, true, K);   \
    else if (out_frames) MC_RAW_GO(true, false, K);        \
    else MC_RAW_GO(false, true, K);                        \
  } while (0)
#ifdef MC_RIGID_STAMP
  unsigned long long z[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  (void)hipMemcpyToSymbol(HIP_SYMBOL(g_rigid_stamps), z, sizeof z);
#endif
  if (storage == MC_STORE_U8) MC_RAW_MODE(0);
  else MC_RAW_MODE(1);
#ifdef MC_RIGID_STAMP
  (void)hipStreamSynchronize(s);
  (void)hipMemcpyFromSymbol(z, HIP_SYMBOL(g_rigid_stamps), sizeof z);
  if (z[5]) {
    const double d = (double)z[5] * nframes;
    fprintf(stderr, "rigid_raw stamps (cycles per wave and frame): dma+weights issue %.0f  cache/addr %.0f  strip %.0f  wait+copy %.0f\n",
            z[0] / d, z[1] / d, z[2] / d, z[3] / d);
  }
#endif
#undef MC_RAW_MODE
#undef MC_RAW_GO
  return mc_check_launch();
}

int mc_warp_rigid(const float* frames, int nframes, int h, int w, const float* shifts_px,
                  float* scratch, float* out_frames, float* out_sum, void* stream) {
  return mc_warp_rigid_phase(frames, nframes, h, w, shifts_px, scratch, out_frames, out_sum, 0, stream);
}

}  // extern "C"
