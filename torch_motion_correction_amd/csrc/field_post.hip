// Small kernels around the patch shift search: leave-one-out reference spectra,
// sub-pixel parabola + outlier rejection + field accumulation, Savitzky-Golay
// smoothing + mean subtraction.  (estimate_motion_xc.py:310-328, :357-388, :391-410,
// :414-627.)  The data are tiny (<= a few hundred patches); one workgroup per frame.
#pragma clang fp contract(off)
#include "mc_common.h"
#include "mcorr.h"

// ------------------------------------------------------------------ reference spectra
// REF[f][g][i] = inv * sum_{o != f} (table[f][o] ? V : U)[o][g][i]
//             = inv * ( T - U_f + sum_{o in S_f} (V_o - U_o) ),   T = sum_o U_o,
// S_f = {o != f : table[f][o]}.  The host turns the t x t table into a schedule: for
// frame f either "add these o to the running sum" (S_f grew: the t <= 50 case adds one
// frame per step) or "rebuild from this list" (an entry reset: memo eviction of
// patch_grid/_patch_grid.py:336-347, t > 50).  sched_ptr[f]..sched_ptr[f+1] indexes
// sched_idx; sched_rebuild[f] says which.
__global__ void ref_mean_except_current(const float2* __restrict__ U, const float2* __restrict__ V,
                                        const int* __restrict__ sched_ptr,
                                        const int* __restrict__ sched_idx,
                                        const uint8_t* __restrict__ sched_rebuild,
                                        float2* __restrict__ REF, int t, int64_t n /* complex */,
                                        float inv) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  float2 T = make_float2(0.f, 0.f);
  for (int o = 0; o < t; ++o) {
    const float2 u = U[(int64_t)o * n + i];
    T.x += u.x;
    T.y += u.y;
  }
  float2 d = make_float2(0.f, 0.f);
  for (int f = 0; f < t; ++f) {
    if (sched_rebuild[f]) d = make_float2(0.f, 0.f);
    for (int q = sched_ptr[f]; q < sched_ptr[f + 1]; ++q) {
      const int o = sched_idx[q];
      const float2 v = V[(int64_t)o * n + i], u = U[(int64_t)o * n + i];
      d.x += v.x - u.x;
      d.y += v.y - u.y;
    }
    const float2 uf = U[(int64_t)f * n + i];
    REF[(int64_t)f * n + i] = make_float2(((T.x - uf.x) + d.x) * inv, ((T.y - uf.y) + d.y) * inv);
  }
}

// ------------------------------------------------------------------ per-frame shifts
#define MC_MAX_PATCHES 4096

__device__ float lower_median(const float* v, int n, int tid, float* slot) {
  // rank counting; ties broken by index; writes the element of rank (n-1)/2 to *slot
  const int want = (n - 1) / 2;
  for (int i = tid; i < n; i += blockDim.x) {
    const float vi = v[i];
    int rank = 0;
    for (int j = 0; j < n; ++j) rank += (v[j] < vi) || (v[j] == vi && j < i);
    if (rank == want) *slot = vi;
  }
  __syncthreads();
  return *slot;
}

__device__ double block_sum(double x, double* red) {
  for (int off = 32; off > 0; off >>= 1) x += __shfl_down(x, off);
  __syncthreads();
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = x;
  __syncthreads();
  double s = 0;
  for (int w = 0; w < (int)(blockDim.x >> 6); ++w) s += red[w];
  return s;
}

__global__ __launch_bounds__(256) void field_accumulate(
    const int* __restrict__ peaks, const float* __restrict__ nb, const int* __restrict__ frames,
    int npatch, int P, int t, float pixel_spacing, float thr, int flags,
    float* __restrict__ field) {
  __shared__ float sy[MC_MAX_PATCHES], sx[MC_MAX_PATCHES];
  __shared__ unsigned char bad[MC_MAX_PATCHES];
  __shared__ double red[4];
  __shared__ float med[2];
  const int fi = blockIdx.x, tid = threadIdx.x;
  const int frame = frames[fi];
  const bool subpix = flags & 1, reject = flags & 2;
  for (int g = tid; g < npatch; g += blockDim.x) {
    const int p = fi * npatch + g;
    const int pk = peaks[p];
    const int iy = pk / P, ix = pk - iy * P;
    float fy = (float)iy, fx = (float)ix;
    if (subpix && iy >= 1 && iy < P - 1 && ix >= 1 && ix < P - 1) {
      const float* q = nb + (int64_t)p * 9;
      float v0 = q[1], v1 = q[4], v2 = q[7];  // column through the peak
      if (v2 != v0) fy += (0.5f * (v0 - v2)) / ((v0 - 2.f * v1) + v2);
      v0 = q[3]; v1 = q[4]; v2 = q[5];        // row through the peak
      if (v2 != v0) fx += (0.5f * (v0 - v2)) / ((v0 - 2.f * v1) + v2);
    }
    const float half = (float)(P / 2);
    sy[g] = fy <= half ? fy : fy - (float)P;
    sx[g] = fx <= half ? fx : fx - (float)P;
  }
  __syncthreads();
  if (reject && npatch > 1) {
    const float my = lower_median(sy, npatch, tid, &med[0]);
    const float mx = lower_median(sx, npatch, tid, &med[1]);
    double ay = 0, ax = 0;
    for (int g = tid; g < npatch; g += blockDim.x) { ay += sy[g]; ax += sx[g]; }
    const double meany = block_sum(ay, red) / npatch;
    const double meanx = block_sum(ax, red) / npatch;
    double qy = 0, qx = 0;
    for (int g = tid; g < npatch; g += blockDim.x) {
      qy += (sy[g] - meany) * (sy[g] - meany);
      qx += (sx[g] - meanx) * (sx[g] - meanx);
    }
    float sdy = (float)sqrt(block_sum(qy, red) / (npatch - 1));
    float sdx = (float)sqrt(block_sum(qx, red) / (npatch - 1));
    sdy = fmaxf(sdy, 1e-6f);
    sdx = fmaxf(sdx, 1e-6f);
    double vy = 0, vx = 0, cnt = 0;
    for (int g = tid; g < npatch; g += blockDim.x) {
      const bool b = (fabsf(sy[g] - my) / sdy > thr) || (fabsf(sx[g] - mx) / sdx > thr);
      bad[g] = b;
      if (!b) { vy += sy[g]; vx += sx[g]; cnt += 1; }
    }
    const double n_ok = block_sum(cnt, red);
    const double sum_y = block_sum(vy, red), sum_x = block_sum(vx, red);
    const float ry = n_ok > 0 ? (float)(sum_y / n_ok) : my;
    const float rx = n_ok > 0 ? (float)(sum_x / n_ok) : mx;
    __syncthreads();
    for (int g = tid; g < npatch; g += blockDim.x)
      if (bad[g]) { sy[g] = ry; sx[g] = rx; }
    __syncthreads();
  }
  for (int g = tid; g < npatch; g += blockDim.x) {
    field[((int64_t)0 * t + frame) * npatch + g] += sy[g] * pixel_spacing;
    field[((int64_t)1 * t + frame) * npatch + g] += sx[g] * pixel_spacing;
  }
}

// ------------------------------------------------------------------ smoothing + centring
__global__ __launch_bounds__(256) void field_smooth_center(const float* __restrict__ in,
                                                           float* __restrict__ out, int t,
                                                           int npatch, int window,
                                                           int subtract_mean) {
  __shared__ double red[4];
  const int tid = threadIdx.x;
  const int nseries = 2 * npatch;
  const int half = window / 2;
  for (int s = tid; s < nseries; s += blockDim.x) {
    const int c = s / npatch, g = s - c * npatch;
    const float* x = in + (int64_t)c * t * npatch + g;  // stride npatch along t
    float* y = out + (int64_t)c * t * npatch + g;
    if (window < 3) {
      for (int i = 0; i < t; ++i) y[(int64_t)i * npatch] = x[(int64_t)i * npatch];
      continue;
    }
    const double cw = 1.0 / (double)window;
    // scipy's kernel for an even window sits half a sample late (savgol_coeffs pos = w/2 - 0.5,
    // convolve1d origin w/2): y[i] = mean(x[i-w/2+1 .. i+w/2]); odd: mean(x[i-half .. i+half])
    const int klo = (window & 1) ? -half : 1 - half;
    for (int i = half; i < t - half; ++i) {
      double a = 0;
      for (int k = klo; k <= half; ++k) a += (double)x[(int64_t)(i + k) * npatch] * cw;
      y[(int64_t)i * npatch] = (float)a;
    }
    // edges: least-squares line through the first / last `window` samples
    for (int side = 0; side < 2; ++side) {
      const int start = side == 0 ? 0 : t - window;
      double xm = 0;
      for (int j = 0; j < window; ++j) xm += (double)x[(int64_t)(start + j) * npatch];
      xm /= window;
      const double jm = 0.5 * (window - 1);
      double num = 0, den = 0;
      for (int j = 0; j < window; ++j) {
        num += (j - jm) * ((double)x[(int64_t)(start + j) * npatch] - xm);
        den += (j - jm) * (j - jm);
      }
      const double slope = num / den;
      if (side == 0) {
        for (int i = 0; i < half; ++i) y[(int64_t)i * npatch] = (float)(xm + slope * (i - jm));
      } else {
        for (int i = t - half; i < t; ++i)
          y[(int64_t)i * npatch] = (float)(xm + slope * ((i - start) - jm));
      }
    }
  }
  __syncthreads();
  if (subtract_mean) {
    const int64_t n = (int64_t)2 * t * npatch;
    double a = 0;
    for (int64_t i = tid; i < n; i += blockDim.x) a += out[i];
    const float mean = (float)(block_sum(a, red) / (double)n);
    __syncthreads();
    for (int64_t i = tid; i < n; i += blockDim.x) out[i] = out[i] - mean;
  }
}

extern "C" {

int mc_xc_ref_mean_except_current(const void* U, const void* V, const int* sched_ptr,
                                  const int* sched_idx, const uint8_t* sched_rebuild, void* REF,
                                  int t, int npatch, int64_t len, float inv_count, void* stream) {
  if (!U || !V || !sched_ptr || !sched_idx || !sched_rebuild || !REF || t < 2 || npatch < 1 ||
      len < 1)
    return MC_ERR_ARG;
  const int64_t n = (int64_t)npatch * len;
  hipLaunchKernelGGL(ref_mean_except_current, dim3((unsigned)((n + 255) / 256)), dim3(256), 0,
                     (hipStream_t)stream, (const float2*)U, (const float2*)V, sched_ptr, sched_idx,
                     sched_rebuild, (float2*)REF, t, n, inv_count);
  return mc_check_launch();
}

int mc_field_accumulate(const int* peaks, const float* nb, const int* frames, int nf, int npatch,
                        int P, int t, float pixel_spacing, float outlier_threshold, int flags,
                        float* field, void* stream) {
  if (!peaks || !frames || !field || nf < 1 || npatch < 1 || npatch > MC_MAX_PATCHES || P < 2)
    return MC_ERR_ARG;
  if ((flags & 1) && !nb) return MC_ERR_ARG;
  hipLaunchKernelGGL(field_accumulate, dim3(nf), dim3(256), 0, (hipStream_t)stream, peaks, nb,
                     frames, npatch, P, t, pixel_spacing, outlier_threshold, flags, field);
  return mc_check_launch();
}

int mc_field_smooth_center(const float* field_in, float* field_out, int t, int npatch, int window,
                           int subtract_mean, void* stream) {
  if (!field_in || !field_out || t < 1 || npatch < 1) return MC_ERR_ARG;
  if (window >= 3 && (window > t || field_in == field_out)) return MC_ERR_ARG;
  hipLaunchKernelGGL(field_smooth_center, dim3(1), dim3(256), 0, (hipStream_t)stream, field_in,
                     field_out, t, npatch, window, subtract_mean);
  return mc_check_launch();
}

}  // extern "C"
