// libmcorr -- the loss and gradient of estimate_local_motion
// (reference: estimate_motion_optimizer.py:361-417 forward, :466-514 shift + filters,
//  :611-671 losses; the reference differentiates that graph with autograd every iteration).
//
// What does not change between iterations is hoisted out of the loop: the masked, filtered
// patch spectra  P[b][f][k] = rfft2(patch_b,f * mask)[k] * bandpass[k] * b_envelope[k]  are
// computed ONCE (the pruned K1/K2 transforms of xc_fft.hip: only bins inside the band are
// kept, everything else is multiplied by zero in the reference).  An iteration is then
//     G_f[k] = P[b][f][k] exp(-2 pi i (fy[k] sy_f + fx[k] sx_f))         (Fourier shift)
//     S[k]   = sum_f G_f[k]
// and every loss of the reference, with the leave-one-out reference  R_f = (S - G_f)/(t-1),
// reduces to a few sums per (patch, frame), h[k] = 1 (plain mean over the half spectrum,
// "mse") or the Hermitian multiplicity of column kx ("cc", "ncc": Parseval of the irfftn):
//     qy_f = sum_k h fy Im(conj(S) G_f)      qx_f likewise with fx
//     d2_f = sum_k h |t G_f - S|^2           (mse:  |G_f - R_f|^2 = d2_f / (t-1)^2)
//     gs_f = sum_k h Re(G_f conj(S - G_f))   (cc:   sum_x x_f y_f = gs_f / ((t-1) N))
//     ey_f = sum_k h |S - G_f|^2             (ncc:  sum_x y_f^2  = ey_f / ((t-1)^2 N))
//     ex_f = sum_k h |G_f|^2                 (ncc:  sum_x x_f^2  = ex_f / N)
// Because sum_f (G_f - R_f) = 0 the mse gradient collapses to
//     dL/dsy_f = -4 pi w t/(t-1)^2 qy_f,     and for cc  dL/dsy_f = -4 pi w/(t-1) qy_f
// (w = the reference's mean normalisation); ncc needs one more pass (local_ncc_grad).
//
// HBM-bound: one iteration streams P twice (0.4 GB for 60 patches x 40 frames x 1024^2 at
// 1 A/px).  Layout: P is (npatch, t, nbins) complex, bins = kx-major (kx * nky + ky) as K2
// writes them, so a tile of consecutive bins is one coalesced run per frame.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "mc_common.h"
#include "mcorr.h"

#define LM_TILE 1024   // bins per workgroup (S tile in LDS: 8 KB)
#define LM_WG 256
#define LM_MAXT 512

struct LmArgs {
  const float2* P;
  const float* shifts;  // (npatch, t, 2) px, (y, x)
  const float* fy;      // (nky) cycles/px of the kept rows
  const float* fx;      // (nkx)
  const float* hx;      // (nkx) weight of column kx, or nullptr = 1
  const float* ab;      // ncc gradient only: (npatch, t, 2) = dL/dn_f, dL/dey_f
  float* out;           // (npatch, ntiles, t, NOUT)
  int t, nkx, nky, ntiles;
};

__device__ __forceinline__ float2 lm_shifted(float2 p, float fyk, float fxk, float sy, float sx) {
  float s, c;
  sincospif(-2.0f * (fyk * sy + fxk * sx), &s, &c);  // exact range reduction for large shifts
  return make_float2(p.x * c - p.y * s, p.x * s + p.y * c);
}

__device__ __forceinline__ float lm_wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
  return v;
}

// MODE 0: the six sums above.  MODE 1: ncc gradient sums (2 per frame).
template <int MODE>
__global__ __launch_bounds__(LM_WG) void local_loss_kernel(LmArgs a) {
  __shared__ float2 Sx[LM_TILE];
  __shared__ float2 Cx[MODE == 1 ? LM_TILE : 1];
  __shared__ float sh[LM_MAXT * 2];
  __shared__ float abx[MODE == 1 ? LM_MAXT * 2 : 1];
  const int tile = blockIdx.x, b = blockIdx.y, tid = threadIdx.x;
  const int t = a.t, nbins = a.nkx * a.nky;
  const float2* P = a.P + (int64_t)b * t * nbins;
  for (int i = tid; i < 2 * t; i += LM_WG) {
    sh[i] = a.shifts[(int64_t)b * t * 2 + i];
    if (MODE == 1) abx[i] = a.ab[(int64_t)b * t * 2 + i];
  }
  __syncthreads();
  const float inv = t > 1 ? 1.f / (float)(t - 1) : 1.f;
  // pass 1: one thread per bin, loop over the frames
  for (int j = tid; j < LM_TILE; j += LM_WG) {
    const int k = tile * LM_TILE + j;
    float2 S = make_float2(0.f, 0.f), C = make_float2(0.f, 0.f);
    if (k < nbins) {
      const int kx = k / a.nky, ky = k - kx * a.nky;
      const float fyk = a.fy[ky], fxk = a.fx[kx];
      for (int f = 0; f < t; ++f) {
        const float2 g = lm_shifted(P[(int64_t)f * nbins + k], fyk, fxk, sh[2 * f], sh[2 * f + 1]);
        S.x += g.x; S.y += g.y;
      }
      if (MODE == 1) {  // C = sum_f a_f conj(G_f) + 2 b_f conj(R_f)
        for (int f = 0; f < t; ++f) {
          const float2 g = lm_shifted(P[(int64_t)f * nbins + k], fyk, fxk, sh[2 * f], sh[2 * f + 1]);
          const float af = abx[2 * f], bf = 2.f * abx[2 * f + 1] * inv;
          C.x += af * g.x + bf * (S.x - g.x);
          C.y -= af * g.y + bf * (S.y - g.y);
        }
      }
    }
    Sx[j] = S;
    if (MODE == 1) Cx[j] = C;
  }
  __syncthreads();
  // pass 2: one wavefront per frame, lanes over the tile's bins
  const int wave = tid >> 6, lane = tid & 63;
  for (int f = wave; f < t; f += LM_WG / 64) {
    const float sy = sh[2 * f], sx = sh[2 * f + 1];
    float acc[6] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    for (int j = lane; j < LM_TILE; j += 64) {
      const int k = tile * LM_TILE + j;
      if (k >= nbins) break;
      const int kx = k / a.nky, ky = k - kx * a.nky;
      const float fyk = a.fy[ky], fxk = a.fx[kx];
      const float hk = a.hx ? a.hx[kx] : 1.f;
      const float2 g = lm_shifted(P[(int64_t)f * nbins + k], fyk, fxk, sy, sx);
      const float2 S = Sx[j];
      if (MODE == 0) {
        const float im = S.x * g.y - S.y * g.x;  // Im(conj(S) G)
        const float dx = (float)t * g.x - S.x, dy = (float)t * g.y - S.y;
        const float rx = S.x - g.x, ry = S.y - g.y;
        acc[0] += hk * fyk * im;
        acc[1] += hk * fxk * im;
        acc[2] += hk * (dx * dx + dy * dy);
        acc[3] += hk * (g.x * rx + g.y * ry);
        acc[4] += hk * (rx * rx + ry * ry);
        acc[5] += hk * (g.x * g.x + g.y * g.y);
      } else {
        // V = a_f conj(R_f) + (C - C_f)/(t-1),  C_f = a_f conj(G_f) + 2 b_f conj(R_f)
        const float af = abx[2 * f], bf = 2.f * abx[2 * f + 1] * inv;
        const float rx = (S.x - g.x) * inv, ry = (S.y - g.y) * inv;
        const float2 C = Cx[j];
        const float cfx = af * g.x + bf * (S.x - g.x), cfy = -(af * g.y + bf * (S.y - g.y));
        const float vx = af * rx + (C.x - cfx) * inv, vy = -af * ry + (C.y - cfy) * inv;
        const float im = vx * g.y + vy * g.x;  // Im(V G)
        acc[0] += hk * fyk * im;
        acc[1] += hk * fxk * im;
      }
    }
    constexpr int NOUT = MODE == 0 ? 6 : 2;
#pragma unroll
    for (int c = 0; c < NOUT; ++c) acc[c] = lm_wave_sum(acc[c]);
    if (lane == 0) {
      float* o = a.out + (((int64_t)b * a.ntiles + tile) * t + f) * NOUT;
#pragma unroll
      for (int c = 0; c < NOUT; ++c) o[c] = acc[c];
    }
  }
}

extern "C" {

int mc_local_loss_tiles(int nkx, int nky, int* ntiles) {
  if (!ntiles || nkx < 1 || nky < 1) return MC_ERR_ARG;
  *ntiles = (nkx * nky + LM_TILE - 1) / LM_TILE;
  return MC_OK;
}

static int lm_launch(int mode, const void* spectra, const float* shifts_px, const float* fy, const float* fx,
                     const float* hx, const float* ab, int npatch, int t, int nkx, int nky, float* partial,
                     void* stream) {
  if (!spectra || !shifts_px || !fy || !fx || !partial || (mode == 1 && !ab)) return MC_ERR_ARG;
  if (npatch < 1 || t < 1 || t > LM_MAXT || nkx < 1 || nky < 1 || npatch > 65535) return MC_ERR_ARG;
  LmArgs a;
  a.P = (const float2*)spectra; a.shifts = shifts_px; a.fy = fy; a.fx = fx; a.hx = hx; a.ab = ab;
  a.out = partial; a.t = t; a.nkx = nkx; a.nky = nky;
  a.ntiles = (nkx * nky + LM_TILE - 1) / LM_TILE;
  dim3 grid(a.ntiles, npatch), block(LM_WG);
  if (mode == 0) hipLaunchKernelGGL(local_loss_kernel<0>, grid, block, 0, (hipStream_t)stream, a);
  else hipLaunchKernelGGL(local_loss_kernel<1>, grid, block, 0, (hipStream_t)stream, a);
  return mc_check_launch();
}

int mc_local_loss_sums(const void* spectra, const float* shifts_px, const float* fy, const float* fx,
                       const float* hx, int npatch, int t, int nkx, int nky, float* partial, void* stream) {
  return lm_launch(0, spectra, shifts_px, fy, fx, hx, nullptr, npatch, t, nkx, nky, partial, stream);
}

int mc_local_ncc_grad(const void* spectra, const float* shifts_px, const float* fy, const float* fx,
                      const float* hx, const float* ab, int npatch, int t, int nkx, int nky, float* partial,
                      void* stream) {
  return lm_launch(1, spectra, shifts_px, fy, fx, hx, ab, npatch, t, nkx, nky, partial, stream);
}

}  // extern "C"
