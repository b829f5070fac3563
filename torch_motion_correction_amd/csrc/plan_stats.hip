// Plan constants (mask, filter) and normalisation statistics.
// All arithmetic that decides *which* pixels/bins are inside follows the fp32 op
// order of the torch expressions the reference evaluates, without FMA contraction.
#pragma clang fp contract(off)
#include <hip/hip_fp16.h>
#include "mc_common.h"
#include "mcorr.h"

// ------------------------------------------------------------------ circle mask
// torch_grid_utils.circle (xc.py:69-74): inside <=> sqrt(dy^2+dx^2) < radius in fp32.
__device__ __forceinline__ bool disk_inside(int dy, int dx, float radius) {
  const float fy = (float)dy, fx = (float)dx;
  return sqrtf(fy * fy + fx * fx) < radius;
}

// halfw[y] = largest a >= 0 with (y, cx +- a) inside, or -1 when the row is empty.
__global__ void mask_halfwidth(int* __restrict__ halfw, int h, int w, float radius) {
  const int y = blockIdx.x * blockDim.x + threadIdx.x;
  if (y >= h) return;
  const int cy = h / 2;
  int a = -1;
  for (int dx = 0; dx <= w; ++dx) {
    if (disk_inside(y - cy, dx, radius)) a = dx;
    else break;
  }
  halfw[y] = a;
}

__global__ void mask_fill(float* __restrict__ mask, const int* __restrict__ halfw, int h, int w,
                          float radius, float smoothing) {
  const int x = blockIdx.x * blockDim.x + threadIdx.x;
  const int y = blockIdx.y;
  if (x >= w) return;
  const int cy = h / 2, cx = w / 2;
  const int dy = y - cy, dx = x - cx;
  float out = 0.f;
  if (disk_inside(dy, dx, radius)) {
    out = 1.f;
  } else if (smoothing > 0.f) {
    const double D = sqrt((double)dy * dy + (double)dx * dx);
    const double excess = D - (double)radius;
    if (excess <= (double)smoothing + 2.0) {
      // exact EDT to the digital disk: the nearest disk pixel is within excess+2 rows
      const int reach = (int)excess + 3;
      int lo = y - reach, hi = y + reach;
      if (lo < 0) lo = 0;
      if (hi > h - 1) hi = h - 1;
      long long best = -1;
      const int adx = dx < 0 ? -dx : dx;
      for (int yy = lo; yy <= hi; ++yy) {
        const int a = halfw[yy];
        if (a < 0) continue;
        // columns of the row that are inside and inside the image
        int gap;
        if (dx >= 0) {
          int right = cx + a;
          if (right > w - 1) right = w - 1;
          gap = x - right;
        } else {
          int left = cx - a;
          if (left < 0) left = 0;
          gap = left - x;
        }
        (void)adx;
        if (gap < 0) gap = 0;
        const long long ddy = (long long)(y - yy);
        const long long d2 = ddy * ddy + (long long)gap * gap;
        if (best < 0 || d2 < best) best = d2;
      }
      if (best > 0) {
        const float d = (float)sqrt((double)best);
        if (d <= smoothing) {
          const float halfpi = 1.5707963267948966f;
          out = cosf(halfpi * (d / smoothing));
        }
      }
    }
  }
  mask[(int64_t)y * w + x] = out;
}

// ------------------------------------------------------------------ xc filter
// torch.fft.fftfreq / rfftfreq: k * (1/n); norm = sqrt(fy^2 + fx^2) in fp32.
__global__ void xc_filter_fill(float* __restrict__ filt, int W, int H, int nkx, int kyp, int kyn,
                               float low, float high, float B, float pixel_size) {
  const int nky = kyp + kyn;
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= nkx * nky) return;
  const int kx = i / nky, kyi = i - kx * nky;
  const int ky = kyi < kyp ? kyi : kyi - kyp + (H - kyn);
  const int kk = (ky < (H + 1) / 2) ? ky : ky - H;
  const float fy = (float)kk * (float)(1.0 / (double)H);
  const float fx = (float)kx * (float)(1.0 / (double)W);
  const float f = sqrtf(fy * fy + fx * fx);
  float v = 0.f;
  if (f > low && f <= high) {
    const float fp = f / pixel_size;
    v = expf(-(B * (fp * fp)) / 4.f);
  }
  filt[i] = v;
}

// ------------------------------------------------------------------ dose weighting
// Exposure filter of the reference's example pipeline (examples/ttMotion.py:331-351:
// rfft2 -> dose_weight_movie(crit_exposure_bfactor=-1) -> irfft2 -> sum), accumulated in
// Fourier space so that only ONE inverse transform per movie is needed:
//   A[kx][ky] (+)= sum_f q_f(k) S[f][kx][ky],   q_f = exp(-0.5 N_f / N_c(k)),
//   N_f = pre_exposure + dose_per_frame (f + 1)   (dose at the END of frame f),
//   N_c(k) = (0.24499 k^-1.6649 + 2.8141) * voltage_scale,  k = |f| / pixel_size  [1/A]
// (Grant & Grigorieff 2015), and on the last chunk A *= 1 / sqrt(sum_f q_f^2) over ALL frames
// ("restore power").  |f| as torch_fourier_filter builds it: sqrt(fy^2 + fx^2) from
// fftfreq / rfftfreq in fp32, clamped at 1e-6 (the DC term keeps weight ~1).  The third-party
// package is absent from the reference tree and the reference has no test for this step:
// semantics as restated in oracle/thirdparty_semantics.py, parity unpinned.
__global__ void dose_accumulate_kernel(const float2* __restrict__ S, int nframes, int frame0,
                                       int total_frames, float2* __restrict__ A, int W, int H,
                                       int nkx, float pixel_size, float pre_exposure,
                                       float dose_per_frame, float vscale, int first, int last) {
  const int64_t n = (int64_t)nkx * H;
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const int kx = (int)(i / H), ky = (int)(i - (int64_t)kx * H);
  const int kk = (ky < (H + 1) / 2) ? ky : ky - H;
  const float fy = (float)kk * (float)(1.0 / (double)H);
  const float fx = (float)kx * (float)(1.0 / (double)W);
  const float f = fmaxf(sqrtf(fy * fy + fx * fx) / pixel_size, 1e-6f);
  const float ncrit = (0.24499f * powf(f, -1.6649f) + 2.8141f) * vscale;
  const float mh = -0.5f / ncrit;
  float2 a = first ? make_float2(0.f, 0.f) : A[i];
  for (int j = 0; j < nframes; ++j) {
    const float q = expf(mh * (pre_exposure + dose_per_frame * (float)(frame0 + j + 1)));
    const float2 v = S[(int64_t)j * n + i];
    a.x += q * v.x;
    a.y += q * v.y;
  }
  if (last) {
    float qq = 0.f;
    for (int j = 0; j < total_frames; ++j) {
      const float q = expf(mh * (pre_exposure + dose_per_frame * (float)(j + 1)));
      qq += q * q;
    }
    const float r = 1.0f / sqrtf(qq);
    a.x *= r;
    a.y *= r;
  }
  A[i] = a;
}

// ------------------------------------------------------------------ input conditioning
// The caller-side preparation of the reference's pipeline (examples/ttMotion.py:90-121 gain
// multiply, :174-199 per-frame mean-zero) for raw detector frames of any storage type:
//   out[f] = raw[f] * gain - mean(raw[f] * gain)      (fp32 out; gain / mean-zero optional)
// pass 1 accumulates the per-frame sums in double, pass 2 applies.  KIND: 0 u8, 1 i16, 2 f16, 3 f32.
template <int KIND>
__device__ __forceinline__ float cond_load(const void* p, int64_t i) {
  if (KIND == 0) return (float)reinterpret_cast<const unsigned char*>(p)[i];
  if (KIND == 1) return (float)reinterpret_cast<const short*>(p)[i];
  if (KIND == 2) return __half2float(reinterpret_cast<const __half*>(p)[i]);
  return reinterpret_cast<const float*>(p)[i];
}

template <int KIND>
__global__ __launch_bounds__(256) void cond_sum_kernel(const void* __restrict__ raw,
                                                       const float* __restrict__ gain, int64_t hw,
                                                       double* __restrict__ sums) {
  const int f = blockIdx.y;
  const int64_t base = (int64_t)f * hw;
  double s = 0.0;
  float ps = 0.f;
  int n = 0;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < hw; i += (int64_t)gridDim.x * 256) {
    const float v = cond_load<KIND>(raw, base + i) * (gain ? gain[i] : 1.f);
    ps += v;
    if (++n == 16) {  // flush the fp32 partial regularly
      s += ps; ps = 0.f; n = 0;
    }
  }
  s += ps;
  for (int off = 32; off > 0; off >>= 1) s += __shfl_down(s, off);
  __shared__ double part[4];
  if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) atomicAdd(&sums[f], (part[0] + part[1]) + (part[2] + part[3]));
}

template <int KIND>
__global__ __launch_bounds__(256) void cond_apply_kernel(const void* __restrict__ raw,
                                                         const float* __restrict__ gain, int64_t hw,
                                                         const double* __restrict__ sums,
                                                         float* __restrict__ out) {
  const int f = blockIdx.y;
  const int64_t base = (int64_t)f * hw;
  const float mean = sums ? (float)(sums[f] / (double)hw) : 0.f;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < hw; i += (int64_t)gridDim.x * 256)
    out[base + i] = cond_load<KIND>(raw, base + i) * (gain ? gain[i] : 1.f) - mean;
}

// Vector form (hw % 8 == 0, 16-byte aligned buffers): 8 pixels per thread -- 8-byte (u8) to 32-byte
// (f32) loads, two float4 stores.  The gain reference is as large as a frame and is read again for
// every frame (5.4 GB per 40 x 4096^2 stack and pass, against 0.67 GB of 8-bit samples): a workgroup
// therefore takes COND_FR consecutive frames per pixel tile with the gain values in registers.  All
// 40 frames per tile were tried too: that scatters every workgroup's accesses over the whole stack
// and is slower than no reuse at all.
template <int KIND>
__device__ __forceinline__ void cond_load8(const void* p, int64_t i, float (&v)[8]) {
  if (KIND == 0) {
    const uint2 q = *reinterpret_cast<const uint2*>(reinterpret_cast<const unsigned char*>(p) + i);
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      v[k] = (float)((q.x >> (8 * k)) & 0xffu);
      v[4 + k] = (float)((q.y >> (8 * k)) & 0xffu);
    }
  } else if (KIND == 1) {
    const uint4 q = *reinterpret_cast<const uint4*>(reinterpret_cast<const short*>(p) + i);
    const unsigned int u[4] = {q.x, q.y, q.z, q.w};
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      v[2 * k] = (float)(short)(u[k] & 0xffffu);
      v[2 * k + 1] = (float)(short)(u[k] >> 16);
    }
  } else if (KIND == 2) {
    const uint4 q = *reinterpret_cast<const uint4*>(reinterpret_cast<const __half*>(p) + i);
    const unsigned int u[4] = {q.x, q.y, q.z, q.w};
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      v[2 * k] = __half2float(__ushort_as_half((unsigned short)(u[k] & 0xffffu)));
      v[2 * k + 1] = __half2float(__ushort_as_half((unsigned short)(u[k] >> 16)));
    }
  } else {
    const float4 a = *reinterpret_cast<const float4*>(reinterpret_cast<const float*>(p) + i);
    const float4 b = *reinterpret_cast<const float4*>(reinterpret_cast<const float*>(p) + i + 4);
    v[0] = a.x; v[1] = a.y; v[2] = a.z; v[3] = a.w; v[4] = b.x; v[5] = b.y; v[6] = b.z; v[7] = b.w;
  }
}

#define COND_FR 8  // frames per workgroup pass: the gain values of a pixel tile are used COND_FR times
template <int KIND, bool APPLY>
__global__ __launch_bounds__(256) void cond_vec_kernel(const void* __restrict__ raw,
                                                       const float* __restrict__ gain, int64_t hw,
                                                       int nframes, double* __restrict__ sums,
                                                       float* __restrict__ out) {
  const int f0 = blockIdx.y * COND_FR;
  float mean[COND_FR];
  double s[COND_FR];
#pragma unroll
  for (int ff = 0; ff < COND_FR; ++ff) {
    s[ff] = 0.0;
    mean[ff] = (APPLY && sums && f0 + ff < nframes) ? (float)(sums[f0 + ff] / (double)hw) : 0.f;
  }
  for (int64_t i = ((int64_t)blockIdx.x * 256 + threadIdx.x) * 8; i < hw; i += (int64_t)gridDim.x * 256 * 8) {
    float g[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) g[k] = 1.f;
    if (gain) {
      const float4 a = *reinterpret_cast<const float4*>(gain + i), b = *reinterpret_cast<const float4*>(gain + i + 4);
      g[0] = a.x; g[1] = a.y; g[2] = a.z; g[3] = a.w; g[4] = b.x; g[5] = b.y; g[6] = b.z; g[7] = b.w;
    }
#pragma unroll
    for (int ff = 0; ff < COND_FR; ++ff) {
      if (f0 + ff >= nframes) break;
      const int64_t base = (int64_t)(f0 + ff) * hw;
      float v[8];
      cond_load8<KIND>(raw, base + i, v);
#pragma unroll
      for (int k = 0; k < 8; ++k) v[k] *= g[k];
      if (APPLY) {
        const float m = mean[ff];
        float* o = out + base + i;
        *reinterpret_cast<float4*>(o) = make_float4(v[0] - m, v[1] - m, v[2] - m, v[3] - m);
        *reinterpret_cast<float4*>(o + 4) = make_float4(v[4] - m, v[5] - m, v[6] - m, v[7] - m);
      } else {
        s[ff] += (double)(((v[0] + v[1]) + (v[2] + v[3])) + ((v[4] + v[5]) + (v[6] + v[7])));
      }
    }
  }
  if (!APPLY) {
    __shared__ double part[COND_FR][4];
#pragma unroll
    for (int ff = 0; ff < COND_FR; ++ff) {
      double r = s[ff];
      for (int off = 32; off > 0; off >>= 1) r += __shfl_down(r, off);
      if ((threadIdx.x & 63) == 0) part[ff][threadIdx.x >> 6] = r;
    }
    __syncthreads();
    if (threadIdx.x < COND_FR && f0 + (int)threadIdx.x < nframes) {
      const int ff = threadIdx.x;
      atomicAdd(&sums[f0 + ff], (part[ff][0] + part[ff][1]) + (part[ff][2] + part[ff][3]));
    }
  }
}

// ------------------------------------------------------------------ N2: statistics of a RAW movie
// The fused raw path (mc_xc_rows_forward_raw, mc_warp_rigid_raw) never materialises the conditioned
// movie c_f = raw_f * gain - mu_f (examples/ttMotion.py:90-121, 180-199).  One pass over the raw bytes
// gives everything the estimator and the warp need to condition on the fly:
//   stats[f] = { sum_all v, sum_box v, sum_box v^2 },  v = raw * gain,  box = normalize_image's central box
// and raw_stats_finalize turns them into mu_f (the frame means, as mc_condition_movie rounds them), the
// joint central-box mean and unbiased standard deviation of the CONDITIONED frames (utils.py:76-84:
// sum_box (v - mu_f) = S_box - n mu_f, sum_box (v - mu_f)^2 = Q_box - 2 mu_f S_box + n mu_f^2, in double) and
// the per-frame offset sub_f = mu_f + mean that K1 subtracts.  The gain tile of a pixel group is held in
// registers over COND_FR frames, as in cond_vec_kernel.
template <int KIND>
__global__ __launch_bounds__(256) void raw_stats_kernel(const void* __restrict__ raw, const float* __restrict__ gain,
                                                        int h, int w, int nframes, int hl, int hu, int wl, int wu,
                                                        double* __restrict__ stats) {
  const int f0 = blockIdx.y * COND_FR;
  const int64_t hw = (int64_t)h * w;
  double sa[COND_FR], sb[COND_FR], qb[COND_FR];
#pragma unroll
  for (int ff = 0; ff < COND_FR; ++ff) sa[ff] = sb[ff] = qb[ff] = 0.0;
  for (int64_t i = ((int64_t)blockIdx.x * 256 + threadIdx.x) * 8; i < hw; i += (int64_t)gridDim.x * 256 * 8) {
    float g[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) g[k] = 1.f;
    if (gain) {
      const float4 a = *reinterpret_cast<const float4*>(gain + i), b = *reinterpret_cast<const float4*>(gain + i + 4);
      g[0] = a.x; g[1] = a.y; g[2] = a.z; g[3] = a.w; g[4] = b.x; g[5] = b.y; g[6] = b.z; g[7] = b.w;
    }
    // w % 8 == 0 (host): the 8 pixels lie in one row; box weights per pixel, the same for every frame
    const int y = (int)(i / w), x = (int)(i - (int64_t)y * w);
    float bw[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) bw[k] = (y >= hl && y < hu && x + k >= wl && x + k < wu) ? 1.f : 0.f;
    const bool any_box = y >= hl && y < hu && x + 7 >= wl && x < wu;
#pragma unroll
    for (int ff = 0; ff < COND_FR; ++ff) {
      if (f0 + ff >= nframes) break;
      float v[8];
      cond_load8<KIND>(raw, (int64_t)(f0 + ff) * hw + i, v);
#pragma unroll
      for (int k = 0; k < 8; ++k) v[k] *= g[k];
      sa[ff] += (double)(((v[0] + v[1]) + (v[2] + v[3])) + ((v[4] + v[5]) + (v[6] + v[7])));
      if (any_box) {
        float ps = 0.f, pq = 0.f;
#pragma unroll
        for (int k = 0; k < 8; ++k) {
          ps = __builtin_fmaf(bw[k], v[k], ps);
          pq = __builtin_fmaf(bw[k] * v[k], v[k], pq);
        }
        sb[ff] += (double)ps;
        qb[ff] += (double)pq;
      }
    }
  }
  __shared__ double part[COND_FR][3][4];
#pragma unroll
  for (int ff = 0; ff < COND_FR; ++ff) {
    double r0 = sa[ff], r1 = sb[ff], r2 = qb[ff];
    for (int off = 32; off > 0; off >>= 1) {
      r0 += __shfl_down(r0, off);
      r1 += __shfl_down(r1, off);
      r2 += __shfl_down(r2, off);
    }
    if ((threadIdx.x & 63) == 0) {
      part[ff][0][threadIdx.x >> 6] = r0;
      part[ff][1][threadIdx.x >> 6] = r1;
      part[ff][2][threadIdx.x >> 6] = r2;
    }
  }
  __syncthreads();
  if (threadIdx.x < COND_FR * 3) {
    const int ff = threadIdx.x / 3, c = threadIdx.x - 3 * ff;
    if (f0 + ff < nframes)
      atomicAdd(&stats[3 * (f0 + ff) + c], (part[ff][c][0] + part[ff][c][1]) + (part[ff][c][2] + part[ff][c][3]));
  }
}

// scalar form for shapes the vector kernel does not take (w % 8 != 0 or unaligned buffers)
template <int KIND>
__global__ __launch_bounds__(256) void raw_stats_scalar_kernel(const void* __restrict__ raw,
                                                               const float* __restrict__ gain, int h, int w, int hl,
                                                               int hu, int wl, int wu, double* __restrict__ stats) {
  const int f = blockIdx.y;
  const int64_t hw = (int64_t)h * w, base = (int64_t)f * hw;
  double sa = 0.0, sb = 0.0, qb = 0.0;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < hw; i += (int64_t)gridDim.x * 256) {
    const float v = cond_load<KIND>(raw, base + i) * (gain ? gain[i] : 1.f);
    const int y = (int)(i / w), x = (int)(i - (int64_t)y * w);
    sa += (double)v;
    if (y >= hl && y < hu && x >= wl && x < wu) {
      sb += (double)v;
      qb += (double)v * (double)v;
    }
  }
  for (int off = 32; off > 0; off >>= 1) {
    sa += __shfl_down(sa, off);
    sb += __shfl_down(sb, off);
    qb += __shfl_down(qb, off);
  }
  __shared__ double part[3][4];
  if ((threadIdx.x & 63) == 0) {
    part[0][threadIdx.x >> 6] = sa;
    part[1][threadIdx.x >> 6] = sb;
    part[2][threadIdx.x >> 6] = qb;
  }
  __syncthreads();
  if (threadIdx.x < 3)
    atomicAdd(&stats[3 * f + threadIdx.x], (part[threadIdx.x][0] + part[threadIdx.x][1]) +
                                               (part[threadIdx.x][2] + part[threadIdx.x][3]));
}

// out: mu[t], sub[t] = mu + mean, mean_rstd[0..1] = {mean, 1 / std} of the conditioned central box
// (all frames jointly, unbiased: torch.std_mean, utils.py:82-83); mean_zero = 0: mu = 0
__global__ void raw_stats_finalize(const double* __restrict__ stats, int nframes, int64_t hw, int64_t nbox,
                                   int mean_zero, float* __restrict__ mu, float* __restrict__ sub,
                                   float* __restrict__ mean_rstd) {
  if (threadIdx.x != 0 || blockIdx.x != 0) return;
  double S = 0.0, Q = 0.0;
  for (int f = 0; f < nframes; ++f) {
    const float m = mean_zero ? (float)(stats[3 * f] / (double)hw) : 0.f;  // as mc_condition_movie rounds it
    mu[f] = m;
    const double md = (double)m, sb = stats[3 * f + 1], qb = stats[3 * f + 2];
    S += sb - (double)nbox * md;
    Q += qb - 2.0 * md * sb + (double)nbox * md * md;
  }
  const double N = (double)nbox * nframes;
  const double mean = S / N;
  double var = (Q - N * mean * mean) / (N - 1.0);
  var = var > 0.0 ? var : 0.0;
  const float meanf = (float)mean;
  mean_rstd[0] = meanf;
  mean_rstd[1] = (float)(1.0 / sqrt(var));
  for (int f = 0; f < nframes; ++f) sub[f] = mu[f] + meanf;
}

// ------------------------------------------------------------------ hot pixels
// The example pipeline's remove_hot_pixels (examples/ttMotion.py:127-172) sits between the gain
// multiply and the mean-zero step: per frame, a pixel of v = raw * gain is hot when
// v > mean + thr * std or v < mean - thr * std (numpy mean / population std of the whole frame).
// That DETECTION is deterministic and is reproduced; the example then overwrites each hot pixel
// with a RANDOM one of its neighbours (np.random.choice, in place, so the result also depends on
// the visiting order): no deterministic counterpart exists.  Our rule: the mean of the (up to 8)
// neighbours that are not hot themselves, taken from the frame BEFORE any replacement; the frame
// mean if every neighbour is hot.  The per-frame mean subtracted afterwards is the mean AFTER the
// replacement, as in the example's order of steps.
template <int KIND>
__global__ __launch_bounds__(256) void cond_stats2_kernel(const void* __restrict__ raw,
                                                          const float* __restrict__ gain, int64_t hw,
                                                          double* __restrict__ stats /* [f][3] */) {
  const int f = blockIdx.y;
  const int64_t base = (int64_t)f * hw;
  double s = 0.0, q = 0.0;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < hw; i += (int64_t)gridDim.x * 256) {
    const double v = (double)(cond_load<KIND>(raw, base + i) * (gain ? gain[i] : 1.f));
    s += v;
    q += v * v;
  }
  for (int off = 32; off > 0; off >>= 1) {
    s += __shfl_down(s, off);
    q += __shfl_down(q, off);
  }
  __shared__ double part[2][4];
  if ((threadIdx.x & 63) == 0) { part[0][threadIdx.x >> 6] = s; part[1][threadIdx.x >> 6] = q; }
  __syncthreads();
  if (threadIdx.x == 0) {
    atomicAdd(&stats[3 * f], (part[0][0] + part[0][1]) + (part[0][2] + part[0][3]));
    atomicAdd(&stats[3 * f + 1], (part[1][0] + part[1][1]) + (part[1][2] + part[1][3]));
  }
}

struct HotLimits {
  float lo, hi, mean;
};
__device__ __forceinline__ HotLimits hot_limits(const double* stats, int f, int64_t hw, float thr) {
  const double m = stats[3 * f] / (double)hw;
  double var = stats[3 * f + 1] / (double)hw - m * m;
  var = var > 0 ? var : 0;
  const double sd = sqrt(var);
  return HotLimits{(float)(m - (double)thr * sd), (float)(m + (double)thr * sd), (float)m};
}

// value of a hot pixel's replacement (see above); (y, x) inside the frame
template <int KIND>
__device__ __forceinline__ float hot_replacement(const void* raw, const float* gain, int64_t base, int h,
                                                 int w, int y, int x, const HotLimits L) {
  float acc = 0.f;
  int n = 0;
  for (int dy = -1; dy <= 1; ++dy)
    for (int dx = -1; dx <= 1; ++dx) {
      const int yy = y + dy, xx = x + dx;
      if ((dy | dx) == 0 || yy < 0 || yy >= h || xx < 0 || xx >= w) continue;
      const int64_t j = (int64_t)yy * w + xx;
      const float v = cond_load<KIND>(raw, base + j) * (gain ? gain[j] : 1.f);
      if (v > L.hi || v < L.lo) continue;
      acc += v;
      ++n;
    }
  return n ? acc / (float)n : L.mean;
}

// MODE 0: find the hot pixels, accumulate sum(replacement - value) and their number per frame;
// MODE 1: write out = (hot ? replacement : value) - mean_after.
template <int KIND, int MODE>
__global__ __launch_bounds__(256) void cond_hot_kernel(const void* __restrict__ raw,
                                                       const float* __restrict__ gain, int h, int w,
                                                       float thr, int mean_zero, double* __restrict__ stats,
                                                       int* __restrict__ hot_count, float* __restrict__ out) {
  const int f = blockIdx.y;
  const int64_t hw = (int64_t)h * w, base = (int64_t)f * hw;
  const HotLimits L = hot_limits(stats, f, hw, thr);
  const float mean_after = (MODE == 1 && mean_zero) ? (float)((stats[3 * f] + stats[3 * f + 2]) / (double)hw) : 0.f;
  double delta = 0.0;
  int cnt = 0;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < hw; i += (int64_t)gridDim.x * 256) {
    float v = cond_load<KIND>(raw, base + i) * (gain ? gain[i] : 1.f);
    if (v > L.hi || v < L.lo) {
      const int y = (int)(i / w), x = (int)(i - (int64_t)y * w);
      const float r = hot_replacement<KIND>(raw, gain, base, h, w, y, x, L);
      if (MODE == 0) { delta += (double)r - (double)v; ++cnt; }
      v = r;
    }
    if (MODE == 1) out[base + i] = v - mean_after;
  }
  if (MODE == 0 && cnt) {  // rare: a handful of pixels per frame
    atomicAdd(&stats[3 * f + 2], delta);
    if (hot_count) atomicAdd(&hot_count[f], cnt);
  }
}

// ------------------------------------------------------------------ statistics
template <typename T>
__global__ __launch_bounds__(256) void box_stats_partial(const T* __restrict__ stack, int h,
                                                         int w, int hl, int hu, int wl, int wu,
                                                         double* __restrict__ acc) {
  // grid: (row chunks, t); each block reduces rows [r0, r1) of one frame's box
  const int f = blockIdx.y;
  const int rows_per = (hu - hl + gridDim.x - 1) / gridDim.x;
  const int r0 = hl + blockIdx.x * rows_per;
  int r1 = r0 + rows_per;
  if (r1 > hu) r1 = hu;
  const T* frame = stack + (int64_t)f * h * w;
  double s = 0.0, q = 0.0;
  // 4 samples per load when every row segment of the box is aligned to it and a multiple of 4 long
  const bool vec = ((w | wl | (wu - wl)) & 3) == 0 && (reinterpret_cast<uintptr_t>(stack) & 15) == 0 &&
                   ((((int64_t)h * w) & 3) == 0);
  for (int y = r0; y < r1; ++y) {
    const T* row = frame + (int64_t)y * w;
    float ps = 0.f, pq = 0.f;
    int n = 0;
    if (vec) {
      for (int x = wl + 4 * threadIdx.x; x < wu; x += 1024) {
        float4 v;
        if (sizeof(T) == 4) {
          v = *reinterpret_cast<const float4*>(row + x);
        } else {
          typedef _Float16 h4 __attribute__((ext_vector_type(4)));
          const h4 hv = *reinterpret_cast<const h4*>(row + x);
          v = make_float4((float)hv.x, (float)hv.y, (float)hv.z, (float)hv.w);
        }
        ps += (v.x + v.y) + (v.z + v.w);
        pq += (v.x * v.x + v.y * v.y) + (v.z * v.z + v.w * v.w);
        if (++n == 4) {  // flush the fp32 partials every 16 samples, as the scalar loop does
          s += ps; q += pq; ps = 0.f; pq = 0.f; n = 0;
        }
      }
      s += ps;
      q += pq;
      continue;
    }
    for (int x = wl + threadIdx.x; x < wu; x += 256) {
      const float v = (float)row[x];
      ps += v;
      pq += v * v;
      if (++n == 16) {  // flush the fp32 partials regularly
        s += ps; q += pq; ps = 0.f; pq = 0.f; n = 0;
      }
    }
    s += ps;
    q += pq;
  }
  for (int off = 32; off > 0; off >>= 1) {
    s += __shfl_down(s, off);
    q += __shfl_down(q, off);
  }
  __shared__ double ss[4], sq[4];
  if ((threadIdx.x & 63) == 0) {
    ss[threadIdx.x >> 6] = s;
    sq[threadIdx.x >> 6] = q;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    s = ss[0] + ss[1] + ss[2] + ss[3];
    q = sq[0] + sq[1] + sq[2] + sq[3];
    atomicAdd(&acc[0], s);
    atomicAdd(&acc[1], q);
  }
}

__global__ void box_stats_final(const double* __restrict__ acc, double count,
                                float* __restrict__ out3) {
  const double mean = acc[0] / count;
  double var = (acc[1] - acc[0] * acc[0] / count) / (count - 1.0);
  if (var < 0) var = 0;
  const float stdf = (float)sqrt(var);
  out3[0] = (float)mean;
  out3[1] = 1.0f / stdf;
  out3[2] = stdf;
}

__global__ void normalize_kernel(const float* __restrict__ src, float* __restrict__ dst, int64_t n,
                                 const float* __restrict__ mean_rstd) {
  const float mean = mean_rstd[0], stdv = mean_rstd[2];
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride)
    dst[i] = (src[i] - mean) / stdv;
}

__global__ void sum_frames_kernel(const float* __restrict__ frames, int nframes, int64_t hw,
                                  float* __restrict__ sum) {
  const int64_t stride = (int64_t)gridDim.x * blockDim.x * 4;
  for (int64_t i = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) * 4; i < hw; i += stride) {
    if (i + 3 < hw) {
      float4 a = make_float4(0, 0, 0, 0);
      for (int f = 0; f < nframes; ++f) {
        const float4 v = *reinterpret_cast<const float4*>(frames + (int64_t)f * hw + i);
        a.x += v.x; a.y += v.y; a.z += v.z; a.w += v.w;
      }
      *reinterpret_cast<float4*>(sum + i) = a;
    } else {
      for (int64_t j = i; j < hw; ++j) {
        float a = 0;
        for (int f = 0; f < nframes; ++f) a += frames[(int64_t)f * hw + j];
        sum[j] = a;
      }
    }
  }
}

extern "C" {

int mc_abi_version(void) { return MCORR_ABI_VERSION; }

int mc_circle_mask(float* mask, int* halfw, int h, int w, float radius, float smoothing_radius,
                   void* stream) {
  if (!mask || !halfw || h < 1 || w < 1 || !(radius >= 0.f) || !(smoothing_radius >= 0.f))
    return MC_ERR_ARG;
  hipLaunchKernelGGL(mask_halfwidth, dim3((h + 63) / 64), dim3(64), 0, (hipStream_t)stream, halfw,
                     h, w, radius);
  hipLaunchKernelGGL(mask_fill, dim3((w + 255) / 256, h), dim3(256), 0, (hipStream_t)stream, mask,
                     halfw, h, w, radius, smoothing_radius);
  return mc_check_launch();
}

int mc_xc_filter(float* filt, const mc_xc_geom* q, float low, float high, float b_factor,
                 float pixel_size, void* stream) {
  if (!filt || !q || q->nkx < 1 || q->kyp + q->kyn < 1 || !(pixel_size > 0.f)) return MC_ERR_ARG;
  const int n = q->nkx * (q->kyp + q->kyn);
  hipLaunchKernelGGL(xc_filter_fill, dim3((n + 255) / 256), dim3(256), 0, (hipStream_t)stream,
                     filt, q->W, q->H, q->nkx, q->kyp, q->kyn, low, high, b_factor, pixel_size);
  return mc_check_launch();
}

int mc_dose_accumulate(const void* S, int nframes, int frame0, int total_frames, void* A, int W,
                       int H, float pixel_size, float pre_exposure, float dose_per_frame,
                       float voltage, int first, int last, void* stream) {
  if (!S || !A || nframes < 1 || frame0 < 0 || total_frames < frame0 + nframes || W < 2 ||
      H < 1 || !(pixel_size > 0.f) || !(dose_per_frame >= 0.f))
    return MC_ERR_ARG;
  const float vscale = voltage >= 300.f ? 1.0f : (voltage >= 200.f ? 0.8f : 0.75f);
  const int nkx = W / 2 + 1;
  const int64_t n = (int64_t)nkx * H;
  hipLaunchKernelGGL(dose_accumulate_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0,
                     (hipStream_t)stream, (const float2*)S, nframes, frame0, total_frames, (float2*)A, W,
                     H, nkx, pixel_size, pre_exposure, dose_per_frame, vscale, first, last);
  return mc_check_launch();
}

int mc_condition_movie_hot(const void* raw, int kind, const float* gain, int nframes, int h, int w,
                           int mean_zero, float threshold, double* stats, int* hot_count, float* out,
                           void* stream) {
  if (!raw || !out || !stats || nframes < 1 || h < 1 || w < 1 || kind < 0 || kind > 3 || !(threshold > 0.f))
    return MC_ERR_ARG;
  hipStream_t st = (hipStream_t)stream;
  const int64_t hw = (int64_t)h * w;
  hipError_t e = hipMemsetAsync(stats, 0, sizeof(double) * 3 * nframes, st);
  if (e != hipSuccess) return (int)e;
  if (hot_count) {
    e = hipMemsetAsync(hot_count, 0, sizeof(int) * nframes, st);
    if (e != hipSuccess) return (int)e;
  }
  int64_t blocks = (hw + 256 * 8 - 1) / (256 * 8);
  if (blocks > 2048) blocks = 2048;
  dim3 grid((unsigned)blocks, nframes);
#define MC_HOT(K)                                                                                        \
  do {                                                                                                   \
    hipLaunchKernelGGL(cond_stats2_kernel<K>, grid, dim3(256), 0, st, raw, gain, hw, stats);             \
    hipLaunchKernelGGL((cond_hot_kernel<K, 0>), grid, dim3(256), 0, st, raw, gain, h, w, threshold,      \
                       mean_zero, stats, hot_count, (float*)nullptr);                                    \
    hipLaunchKernelGGL((cond_hot_kernel<K, 1>), grid, dim3(256), 0, st, raw, gain, h, w, threshold,      \
                       mean_zero, stats, (int*)nullptr, out);                                            \
  } while (0)
  switch (kind) {
    case 0: MC_HOT(0); break;
    case 1: MC_HOT(1); break;
    case 2: MC_HOT(2); break;
    default: MC_HOT(3); break;
  }
#undef MC_HOT
  return mc_check_launch();
}

int mc_raw_movie_stats(const void* raw, int kind, const float* gain, int nframes, int h, int w, int hl, int hu,
                       int wl, int wu, int mean_zero, double* stats, float* mu, float* sub, float* mean_rstd,
                       void* stream) {
  if (!raw || !stats || !mu || !sub || !mean_rstd || nframes < 1 || h < 1 || w < 1) return MC_ERR_ARG;
  if (kind < 0 || kind > 3) return MC_ERR_UNSUPPORTED;
  if (hl < 0 || hu > h || wl < 0 || wu > w || hl >= hu || wl >= wu) return MC_ERR_ARG;
  hipStream_t st = (hipStream_t)stream;
  const int64_t hw = (int64_t)h * w;
  hipError_t e = hipMemsetAsync(stats, 0, sizeof(double) * 3 * nframes, st);
  if (e != hipSuccess) return (int)e;
  const bool tiled = (w % 8 == 0) && ((reinterpret_cast<uintptr_t>(raw) & (kind == 0 ? 7 : 15)) == 0) &&
                     (!gain || (reinterpret_cast<uintptr_t>(gain) & 15) == 0);
  if (tiled) {
    int64_t tb = (hw / 8 + 255) / 256;
    if (tb > 2048) tb = 2048;
    const dim3 grid((unsigned)tb, (nframes + COND_FR - 1) / COND_FR);
    switch (kind) {
      case 0: hipLaunchKernelGGL(raw_stats_kernel<0>, grid, dim3(256), 0, st, raw, gain, h, w, nframes, hl, hu, wl, wu, stats); break;
      case 1: hipLaunchKernelGGL(raw_stats_kernel<1>, grid, dim3(256), 0, st, raw, gain, h, w, nframes, hl, hu, wl, wu, stats); break;
      case 2: hipLaunchKernelGGL(raw_stats_kernel<2>, grid, dim3(256), 0, st, raw, gain, h, w, nframes, hl, hu, wl, wu, stats); break;
      default: hipLaunchKernelGGL(raw_stats_kernel<3>, grid, dim3(256), 0, st, raw, gain, h, w, nframes, hl, hu, wl, wu, stats); break;
    }
  } else {
    int64_t blocks = (hw + 256 * 8 - 1) / (256 * 8);
    if (blocks > 2048) blocks = 2048;
    const dim3 grid((unsigned)blocks, nframes);
    switch (kind) {
      case 0: hipLaunchKernelGGL(raw_stats_scalar_kernel<0>, grid, dim3(256), 0, st, raw, gain, h, w, hl, hu, wl, wu, stats); break;
      case 1: hipLaunchKernelGGL(raw_stats_scalar_kernel<1>, grid, dim3(256), 0, st, raw, gain, h, w, hl, hu, wl, wu, stats); break;
      case 2: hipLaunchKernelGGL(raw_stats_scalar_kernel<2>, grid, dim3(256), 0, st, raw, gain, h, w, hl, hu, wl, wu, stats); break;
      default: hipLaunchKernelGGL(raw_stats_scalar_kernel<3>, grid, dim3(256), 0, st, raw, gain, h, w, hl, hu, wl, wu, stats); break;
    }
  }
  hipLaunchKernelGGL(raw_stats_finalize, dim3(1), dim3(64), 0, st, (const double*)stats, nframes, hw,
                     (int64_t)(hu - hl) * (wu - wl), mean_zero, mu, sub, mean_rstd);
  return mc_check_launch();
}

int mc_condition_movie(const void* raw, int kind, const float* gain, int nframes, int64_t hw,
                       int mean_zero, double* sums, float* out, void* stream) {
  if (!raw || !out || nframes < 1 || hw < 1 || kind < 0 || kind > 3 || (mean_zero && !sums))
    return MC_ERR_ARG;
  hipStream_t st = (hipStream_t)stream;
  const bool tiled = (hw % 8 == 0) && ((reinterpret_cast<uintptr_t>(raw) & (kind == 0 ? 7 : 15)) == 0) &&
                     ((reinterpret_cast<uintptr_t>(out) & 15) == 0) &&
                     (!gain || (reinterpret_cast<uintptr_t>(gain) & 15) == 0) && hw / 8 / 256 < 0x7fffffff;
  if (tiled) {
    if (mean_zero) {
      hipError_t e = hipMemsetAsync(sums, 0, sizeof(double) * nframes, st);
      if (e != hipSuccess) return (int)e;
    }
    int64_t tb = (hw / 8 + 255) / 256;
    if (tb > 2048) tb = 2048;
    const dim3 tgrid((unsigned)tb, (nframes + COND_FR - 1) / COND_FR);
#define MC_COND_T(K)                                                                                    \
  do {                                                                                                  \
    if (mean_zero)                                                                                      \
      hipLaunchKernelGGL((cond_vec_kernel<K, false>), tgrid, dim3(256), 0, st, raw, gain, hw, nframes,    \
                         sums, (float*)nullptr);                                                        \
    hipLaunchKernelGGL((cond_vec_kernel<K, true>), tgrid, dim3(256), 0, st, raw, gain, hw, nframes,       \
                       mean_zero ? sums : (double*)nullptr, out);                                       \
  } while (0)
    switch (kind) {
      case 0: MC_COND_T(0); break;
      case 1: MC_COND_T(1); break;
      case 2: MC_COND_T(2); break;
      default: MC_COND_T(3); break;
    }
#undef MC_COND_T
    return mc_check_launch();
  }
  int64_t blocks = (hw + 256 * 8 - 1) / (256 * 8);
  if (blocks > 2048) blocks = 2048;
  dim3 grid((unsigned)blocks, nframes);
#define MC_COND(K)                                                                                   \
  do {                                                                                               \
    if (mean_zero) hipLaunchKernelGGL(cond_sum_kernel<K>, grid, dim3(256), 0, st, raw, gain, hw, sums); \
    hipLaunchKernelGGL(cond_apply_kernel<K>, grid, dim3(256), 0, st, raw, gain, hw,                  \
                       mean_zero ? (const double*)sums : (const double*)nullptr, out);               \
  } while (0)
  if (mean_zero) {
    hipError_t e = hipMemsetAsync(sums, 0, sizeof(double) * nframes, st);
    if (e != hipSuccess) return (int)e;
  }
  switch (kind) {
    case 0: MC_COND(0); break;
    case 1: MC_COND(1); break;
    case 2: MC_COND(2); break;
    default: MC_COND(3); break;
  }
#undef MC_COND
  return mc_check_launch();
}

int mc_central_box_stats(const float* stack, int t, int h, int w, int hl, int hu, int wl, int wu,
                         double* acc, float* out3, void* stream) {
  return mc_central_box_stats_t(stack, MC_STORE_F32, t, h, w, hl, hu, wl, wu, acc, out3, stream);
}

int mc_central_box_stats_t(const void* stack, int storage, int t, int h, int w, int hl, int hu, int wl,
                           int wu, double* acc, float* out3, void* stream) {
  if (!stack || !acc || !out3 || t < 1 || hl < 0 || hu > h || wl < 0 || wu > w || hl >= hu ||
      wl >= wu)
    return MC_ERR_ARG;
  if (storage != MC_STORE_F32 && storage != MC_STORE_F16) return MC_ERR_UNSUPPORTED;
  hipError_t e = hipMemsetAsync(acc, 0, 2 * sizeof(double), (hipStream_t)stream);
  if (e != hipSuccess) return (int)e;
  int chunks = (hu - hl + 15) / 16;
  if (chunks > 256) chunks = 256;
  if (storage == MC_STORE_F32)
    hipLaunchKernelGGL(box_stats_partial<float>, dim3(chunks, t), dim3(256), 0, (hipStream_t)stream,
                       (const float*)stack, h, w, hl, hu, wl, wu, acc);
  else
    hipLaunchKernelGGL(box_stats_partial<_Float16>, dim3(chunks, t), dim3(256), 0, (hipStream_t)stream,
                       (const _Float16*)stack, h, w, hl, hu, wl, wu, acc);
  const double count = (double)t * (hu - hl) * (wu - wl);
  hipLaunchKernelGGL(box_stats_final, dim3(1), dim3(1), 0, (hipStream_t)stream, acc, count, out3);
  return mc_check_launch();
}

int mc_normalize(const float* src, float* dst, int64_t n, const float* mean_rstd, void* stream) {
  if (!src || !dst || !mean_rstd || n < 1) return MC_ERR_ARG;
  int64_t blocks = (n + 255) / 256;
  if (blocks > 8192) blocks = 8192;
  hipLaunchKernelGGL(normalize_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream,
                     src, dst, n, mean_rstd);
  return mc_check_launch();
}

int mc_sum_frames(const float* frames, int nframes, int64_t hw, float* sum, void* stream) {
  if (!frames || !sum || nframes < 1 || hw < 1 || (hw & 3)) return MC_ERR_ARG;
  int64_t blocks = (hw / 4 + 255) / 256;
  if (blocks > 4096) blocks = 4096;
  hipLaunchKernelGGL(sum_frames_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream,
                     frames, nframes, hw, sum);
  return mc_check_launch();
}

}  // extern "C"
