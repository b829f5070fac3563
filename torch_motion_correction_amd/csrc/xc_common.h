// Types shared by the pruned transform engines: xc_fft.hip (power-of-two lengths) and xcg_fft.hip
// (any other length: mixed radix / chirp-z).
#pragma once
#include <stdlib.h>
#include <string.h>
#include <type_traits>
#include "mc_fft.h"
#include "mcorr.h"

struct XcGeom {
  int W, H;      // transform size
  int nkx;       // kept rfft columns [0, nkx)
  int kyp, kyn;  // kept ky rows [0, kyp) and [H-kyn, H); nky = kyp + kyn
  int y0, ny;    // rows [y0, y0+ny) of the window can be non-zero (ny % RG == 0)
  int x0, x1;    // columns [x0, x1) can be non-zero (both even)
  int RG;        // rows per workgroup in K1/K4
};

struct XcBox {  // central box of normalize_image (utils.py:76-81) in window coordinates
  int hl, hu, wl, wu;
};

// index of fft row ky among the kept rows [0, kyp) + [H - kyn, H), or -1
__device__ __forceinline__ int kept_index(int ky, int H, int kyp, int kyn) {
  if (ky < kyp) return ky;
  if (ky >= H - kyn) return ky - (H - kyn) + kyp;
  return -1;
}

struct PeakCand {
  float v;
  int idx;
};
__device__ __forceinline__ void cand_merge(float& bv, int& bi, float v, int i) {
  if (v > bv || (v == bv && i < bi)) {
    bv = v;
    bi = i;
  }
}

__device__ __forceinline__ int float_order(float f) {  // order-preserving float -> int
  const int i = __float_as_int(f);
  return i >= 0 ? i : i ^ 0x7fffffff;
}

#define MC_DISPATCH_CASE(V, ...) \
  case V: {                      \
    constexpr int L = V;         \
    __VA_ARGS__;                 \
  } break;

// rows_pow2 / cols_pow2: which dimension the calling kernel transforms with the
// power-of-two FFT (the other one may be any length handled by the chirp-z kernels)
static int geom_from(const mc_xc_geom* q, XcGeom* g, bool rows_pow2 = true, bool cols_pow2 = true) {
  if (!q) return MC_ERR_ARG;
  if (q->W < 4 || q->W > 16384 || ((q->W & 1) && q->W > 8191) || q->H < 2 || q->H > 8192) return MC_ERR_UNSUPPORTED;
  if (rows_pow2 && (!mc_is_pow2(q->W) || q->W < 32 || q->W > 8192)) return MC_ERR_UNSUPPORTED;
  if (cols_pow2 && (!mc_is_pow2(q->H) || q->H < 16 || q->H > 4096)) return MC_ERR_UNSUPPORTED;
  if (q->nkx < 1 || q->nkx > q->W / 2 + 1) return MC_ERR_ARG;
  if (q->kyp < 0 || q->kyn < 0 || q->kyp + q->kyn < 1 || q->kyp + q->kyn > q->H) return MC_ERR_ARG;
  if (q->RG < 1 || q->ny < 1 || q->ny % q->RG || q->H % q->RG) return MC_ERR_ARG;
  if (rows_pow2 && (q->RG % (MC_WG / fft_threads(q->W / 2)))) return MC_ERR_ARG;  // rows vs sub-groups
  if (q->y0 < 0 || q->y0 + q->ny > q->H) return MC_ERR_ARG;
  if (q->x0 < 0 || q->x1 > q->W || q->x0 >= q->x1) return MC_ERR_ARG;
  if (!(q->W & 1) && ((q->x0 & 1) || (q->x1 & 1))) return MC_ERR_ARG;
  g->W = q->W; g->H = q->H; g->nkx = q->nkx; g->kyp = q->kyp; g->kyn = q->kyn;
  g->y0 = q->y0; g->ny = q->ny; g->x0 = q->x0; g->x1 = q->x1; g->RG = q->RG;
  return MC_OK;
}

// the two small kernels of the arg-max that both engines launch live in xc_fft.hip
void mc_launch_row_bounds(const cfloat* T2, float* bounds, int nkx, int H, int npairs, hipStream_t stream);
void mc_launch_peak_final(const float* part_val, const int* part_idx, int ngrp, int H, int W, int* peaks,
                          float* shifts, const int* shift_rows, int npairs, hipStream_t stream);
