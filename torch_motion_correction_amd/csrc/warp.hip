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
__device__ __forceinline__ float grid_chain(float c, float n) {
  const float g = c / (0.5f * n - 0.5f) - 1.f;
  return (g + 1.f) * ((n - 1.f) / 2.f);
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

#define WARP_TX 32   // threads across (4 px each) -> 128 px
#define WARP_TY 8    // thread rows; each thread does rows ty and ty+8 -> 16 rows
#define WARP_PX 4

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

struct __attribute__((packed, aligned(4))) f4u {  // 16-byte load, dword aligned
  float x, y, z, w;
};

__device__ __forceinline__ int clampi(int v, int lo, int hi) { return v < lo ? lo : (v > hi ? hi : v); }

template <bool WRITE_FRAMES, bool WRITE_SUM>
__global__ __launch_bounds__(WARP_TX* WARP_TY) void warp_main(WarpArgs a) {
  // XCD-aware tile order: blocks b, b+8, b+16.. share an XCD (round-robin dispatch);
  // give each XCD a contiguous band of tile rows so vertical halos hit its own L2.
  const int nt = a.tiles_x * a.tiles_y;
  int b = blockIdx.x;
  int tile = b;
  if ((nt & 7) == 0) tile = (b & 7) * (nt >> 3) + (b >> 3);
  const int tyi = tile / a.tiles_x, txi = tile - tyi * a.tiles_x;
  const int x0 = txi * (WARP_TX * WARP_PX) + threadIdx.x * WARP_PX;
  const int h = a.h, w = a.w;
  const float fh = (float)h, fw = (float)w;
  const int64_t hw = (int64_t)h * w;

#pragma unroll
  for (int half = 0; half < 2; ++half) {
    const int y = tyi * (WARP_TY * 2) + half * WARP_TY + threadIdx.y;
    if (y >= h || x0 >= w) continue;
    const int4 yt = *reinterpret_cast<const int4*>(a.ytap + 4 * y);
    const float4 yc = *reinterpret_cast<const float4*>(a.ycoef + 4 * y);
    const bool full = (x0 + WARP_PX <= w);
    float acc[WARP_PX] = {0.f, 0.f, 0.f, 0.f};

    for (int f = 0; f < a.nframes; ++f) {
      const float* fr = a.frames + (int64_t)f * hw;
      const float* E = a.etab + (int64_t)f * 2 * a.GH * w;
      float res[WARP_PX];
      float uy[WARP_PX], ux[WARP_PX];
      bool inside[WARP_PX];
#pragma unroll
      for (int k = 0; k < WARP_PX; ++k) {
        const int x = x0 + k < w ? x0 + k : w - 1;
        const float* Ey = E + x;
        const float* Ex = E + (int64_t)a.GH * w + x;
        float sy = ((yc.x * Ey[(int64_t)yt.x * w] + yc.y * Ey[(int64_t)yt.y * w]) +
                    yc.z * Ey[(int64_t)yt.z * w]) + yc.w * Ey[(int64_t)yt.w * w];
        float sx = ((yc.x * Ex[(int64_t)yt.x * w] + yc.y * Ex[(int64_t)yt.y * w]) +
                    yc.z * Ex[(int64_t)yt.z * w]) + yc.w * Ex[(int64_t)yt.w * w];
        sy = sy / a.pixel_spacing;
        sx = sx / a.pixel_spacing;
        const float cy = (float)y + sy, cx = (float)x + sx;
        inside[k] = (cy >= 0.f) && (cy <= fh - 1.f) && (cx >= 0.f) && (cx <= fw - 1.f);
        uy[k] = grid_chain(cy, fh);
        ux[k] = grid_chain(cx, fw);
      }
      const float fy0 = floorf(uy[0]), fx0 = floorf(ux[0]);
      const int iy0 = (int)fy0, ix0 = (int)fx0;
      bool regular = full && iy0 >= 1 && iy0 + 2 <= h - 1 && ix0 >= 1 && ix0 + 6 <= w - 1;
#pragma unroll
      for (int k = 1; k < WARP_PX; ++k)
        regular = regular && (floorf(uy[k]) == fy0) && (floorf(ux[k]) == fx0 + (float)k);
      if (regular) {
        float v[4][8];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const float* r = fr + (int64_t)(iy0 - 1 + i) * w + (ix0 - 1);
          const f4u lo = *reinterpret_cast<const f4u*>(r);
          const f4u hi = *reinterpret_cast<const f4u*>(r + 4);
          v[i][0] = lo.x; v[i][1] = lo.y; v[i][2] = lo.z; v[i][3] = lo.w;
          v[i][4] = hi.x; v[i][5] = hi.y; v[i][6] = hi.z; v[i][7] = hi.w;
        }
#pragma unroll
        for (int k = 0; k < WARP_PX; ++k) {
          float wy[4], wx[4];
          cubic_coeffs(uy[k] - fy0, wy);
          cubic_coeffs(ux[k] - (fx0 + (float)k), wx);
          float rowv[4];
#pragma unroll
          for (int i = 0; i < 4; ++i)
            rowv[i] = ((wx[0] * v[i][k] + wx[1] * v[i][k + 1]) + wx[2] * v[i][k + 2]) +
                      wx[3] * v[i][k + 3];
          const float o = ((wy[0] * rowv[0] + wy[1] * rowv[1]) + wy[2] * rowv[2]) + wy[3] * rowv[3];
          res[k] = inside[k] ? o : 0.f;
        }
      } else {
#pragma unroll
        for (int k = 0; k < WARP_PX; ++k) {
          const float fy = floorf(uy[k]), fx = floorf(ux[k]);
          float wy[4], wx[4];
          cubic_coeffs(uy[k] - fy, wy);
          cubic_coeffs(ux[k] - fx, wx);
          // border padding: clip each tap coordinate (ATen clip_coordinates), in float
          // first so that huge coordinates cannot overflow the int conversion
          float rowv[4];
#pragma unroll
          for (int i = 0; i < 4; ++i) {
            const float ty = fminf(fmaxf(fy + (float)(i - 1), 0.f), fh - 1.f);
            const float* r = fr + (int64_t)(int)ty * w;
            float s = 0.f;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
              const float tx = fminf(fmaxf(fx + (float)(j - 1), 0.f), fw - 1.f);
              const float term = wx[j] * r[(int)tx];
              s = (j == 0) ? term : s + term;
            }
            rowv[i] = s;
          }
          const float o = ((wy[0] * rowv[0] + wy[1] * rowv[1]) + wy[2] * rowv[2]) + wy[3] * rowv[3];
          res[k] = inside[k] ? o : 0.f;
        }
      }
      if (WRITE_FRAMES) {
        float* o = a.out_frames + (int64_t)f * hw + (int64_t)y * w + x0;
        if (full && ((((uintptr_t)o) & 15) == 0)) {
          *reinterpret_cast<float4*>(o) = make_float4(res[0], res[1], res[2], res[3]);
        } else {
          for (int k = 0; k < WARP_PX && x0 + k < w; ++k) o[k] = res[k];
        }
      }
      if (WRITE_SUM) {
#pragma unroll
        for (int k = 0; k < WARP_PX; ++k) acc[k] += res[k];
      }
    }
    if (WRITE_SUM) {
      float* o = a.out_sum + (int64_t)y * w + x0;
      for (int k = 0; k < WARP_PX && x0 + k < w; ++k) o[k] += acc[k];
    }
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

int mc_warp_scratch_bytes(int nframes, int h, int w, int GH, int GW, int64_t* bytes) {
  if (!bytes || nframes < 1 || h < 2 || w < 2 || GH < 1 || GW < 1) return MC_ERR_ARG;
  // etab floats + (ytap,ycoef,xtap,xcoef)
  *bytes = (etab_floats(nframes, GH, w) + 8 * (int64_t)(h + w)) * 4;
  return MC_OK;
}

int mc_warp_frames(const float* frames, int nframes, int h, int w, const float* lattice, int GH,
                   int GW, float pixel_spacing, float* scratch, float* out_frames, float* out_sum,
                   void* stream) {
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
  a.tiles_x = (w + WARP_TX * WARP_PX - 1) / (WARP_TX * WARP_PX);
  a.tiles_y = (h + WARP_TY * 2 - 1) / (WARP_TY * 2);
  dim3 grid(a.tiles_x * a.tiles_y), block(WARP_TX, WARP_TY);
  if (out_frames && out_sum) hipLaunchKernelGGL((warp_main<true, true>), grid, block, 0, s, a);
  else if (out_frames) hipLaunchKernelGGL((warp_main<true, false>), grid, block, 0, s, a);
  else hipLaunchKernelGGL((warp_main<false, true>), grid, block, 0, s, a);
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

}  // extern "C"
