// Pruned, separable 2-D real FFT engine for transform lengths that are not powers of two: the xcg_*
// kernels (direct mixed-radix lines for the K3 formats, Bluestein chirp-z for everything else) with the
// same pruning, layouts (T1, S, T2) and fused prologues / epilogues as xc_fft.hip.  Split from xc_fft.hip
// in round 3 (one translation unit took five minutes to compile).
#include "xc_common.h"

// The entry points of this file are compiled as four objects (XCG_PART = 0 .. 3, _build.py): each part
// instantiates one kernel family (rows forward + peak neighbourhood / columns forward / columns inverse /
// rows inverse), 75 s each side by side instead of five minutes in a row.  XCG_PART < 0: everything.
#ifndef XCG_PART
#define XCG_PART -1
#endif

// K6 for any width: the nine map values around a peak as direct sums over the kept columns,
//   cc(y, x) = sum_kx h(kx) Re(T2[p][kx][y] exp(+2 pi i kx x / W)),
// h = 1 for kx = 0 (and the Nyquist column of an even width), 2 otherwise; the imaginary parts of
// those self-conjugate columns are dropped as a c2r transform drops them.  nkx terms per value:
// nothing to transform for 9 values.  kx x is reduced mod W in integers before the sine.
#if XCG_PART == 0 || XCG_PART < 0
__global__ __launch_bounds__(MC_WG) void xcg_peak_nbhd(const cfloat* __restrict__ T2, const int* __restrict__ peaks,
                                                       float* __restrict__ nb, XcGeom g) {
  const int tid = threadIdx.x;
  const int p = blockIdx.y, dy = (int)blockIdx.x - 1;
  const int pk = peaks[p];
  const int py = pk / g.W, px = pk - py * g.W;
  const int y = py + dy;
  float* o = nb + (int64_t)p * 9 + (dy + 1) * 3;
  if (y < 0 || y >= g.H) {
    if (tid < 3) o[tid] = __builtin_nanf("");
    return;
  }
  const cfloat* in = T2 + (int64_t)p * g.nkx * g.H + y;
  float acc[3] = {0.f, 0.f, 0.f};
  const float invw = 1.0f / (float)g.W;
  for (int kx = tid; kx < g.nkx; kx += MC_WG) {
    cfloat v = in[(int64_t)kx * g.H];
    const bool self = kx == 0 || (!(g.W & 1) && kx == g.W / 2);
    if (self) v.y = 0.f;
    const float hk = self ? 1.f : 2.f;
#pragma unroll
    for (int i = 0; i < 3; ++i) {
      const int x = px + i - 1;
      if (x < 0 || x >= g.W) continue;
      const int r = (int)(((int64_t)kx * x) % g.W);
      float sn, cs;
      sincospif(2.0f * (float)r * invw, &sn, &cs);
      acc[i] += hk * (v.x * cs - v.y * sn);
    }
  }
  __shared__ float part[3][MC_WG / 64];
#pragma unroll
  for (int i = 0; i < 3; ++i) {
    float a = acc[i];
    for (int off = 32; off > 0; off >>= 1) a += __shfl_down(a, off);
    if ((tid & 63) == 0) part[i][tid >> 6] = a;
  }
  __syncthreads();
  if (tid < 3) {
    const int x = px + tid - 1;
    float v = __builtin_nanf("");
    if (x >= 0 && x < g.W) {
      v = 0.f;
      for (int w = 0; w < MC_WG / 64; ++w) v += part[tid][w];
    }
    o[tid] = v;
  }
}
#endif

// =====================================================================================
// Generic transform lengths (Bluestein chirp-z on the power-of-two workgroup FFT).
// Rows: any EVEN W (real rows are packed into W/2 complex points as in the power-of-two
// path); columns: any H.  Same pruning, same fused prologues/epilogues, same layouts
// (T1, S, T2); one line per workgroup pass, two length-M transforms per line, M the power
// of two >= 2n-1.  Used for whole-frame transforms of non-power-of-two detectors (K3:
// 4092 x 5760).  Tables per (n, direction): chirp[n] = exp(-+ i pi j^2 / n),
// bspec[M] = FFT_M(wrapped conj chirp) / M  (host, double precision, plan.py).
// =====================================================================================
// Line-engine codes of the xcg_* kernels' template parameter: 5..14 = chirp-z on M = 2^code points;
// 20 / 21 = chirp-z on M = 5120 / 10240 (2^k 5); 22 .. 25 = NO chirp, a direct mixed-radix transform
// of the n points themselves: 2880 / 5760 = 2^a 3^2 5 (rows of 5760 / 11520 columns), 4092 / 8184 =
// 2^a 3 11 31 (their columns; radix 31 and 11 passes, mc_fft.h).
__host__ __device__ constexpr int mc_line_m(int code) {
  return code < 20 ? (1 << code) : code == 20 ? 5120 : code == 21 ? 10240 : code == 22 ? 2880 : code == 23 ? 5760
       : code == 24 ? 4092 : code == 25 ? 8184 : 0;
}
__host__ __device__ constexpr bool mc_line_direct(int code) { return code >= 22; }
static int mc_line_code(int M, bool direct) {
  if (direct) return M == 2880 ? 22 : M == 5760 ? 23 : M == 4092 ? 24 : M == 8184 ? 25 : -1;
  if (M == 5120) return 20;
  if (M == 10240) return 21;
  return mc_is_pow2(M) ? mc_ilog2(M) : -1;
}

struct XcLine {
  const cfloat* tw_m;   // exp(-2 pi i k / M), M entries
  const cfloat* chirp;  // n entries
  const cfloat* bspec;  // M entries
  int n;                // transform length (W/2 for rows, H for columns)
  int keep;             // > 0: output-pruned plan (wg_bluestein), only xcg_rows_fwd takes it
};

// One line transform of the xcg_* kernels: chirp-z on M points (tables of `ln`), or -- for the
// direct codes -- the mixed-radix transform of the n = M points themselves (ln.tw_m then holds
// exp(-2 pi i k / n)); DIR only matters for the direct form (the chirp tables carry the direction).
template <int CODE, int DIR, typename Load, typename Store>
__device__ __forceinline__ void xcg_line_fft(cfloat* line, int tid, const XcLine& ln, int n, Load load,
                                             Store store, int keep = 0) {
  constexpr int M = mc_line_m(CODE);
  if constexpr (mc_line_direct(CODE)) {
    // the lane index made opaque per line: twiddles and addresses of the mixed-radix passes are then
    // re-derived for every line instead of being hoisted out of the row loops into 100+ registers
    int t = tid;
    asm volatile("" : "+v"(t));
    wg_fft_any<M, DIR>(line, t, ln.tw_m, 1, load, store);
  }
  else wg_bluestein<M>(line, tid, ln.tw_m, ln.chirp, ln.bspec, n, load, store, keep);
}


// RAW (N2): 1 = u8, 2 = i16 samples conditioned on the fly as raw * gain - job_sub[job] (`gain` has the
// frames' row pitch: whole-frame jobs; mc_raw_movie_stats supplies job_sub and mean_rstd[1])
template <int LOGM, int RAW = 0>
__global__ __launch_bounds__(MC_WG) void xcg_rows_fwd(
    const void* __restrict__ src_any, const int64_t* __restrict__ job_off, int64_t row_stride,
    const int* __restrict__ job_expo, const float* __restrict__ mask,
    const float* __restrict__ mean_rstd, cfloat* __restrict__ T1,
    const cfloat* __restrict__ tw_row, XcLine ln, XcGeom g, const float* __restrict__ gain,
    const float* __restrict__ job_sub) {
  constexpr int M = mc_line_m(LOGM);
  // direct (mixed-radix) lines: the transform's outputs go back into the line itself and the unpack
  // reads Z[k], Z[n-k] from it -- no zlo / zhi copies: 48 instead of 57 KB of LDS for 5760-column
  // frames, i.e. three workgroups per CU instead of two (the kernel is latency-bound)
  constexpr bool DIRECT = mc_line_direct(LOGM);
  extern __shared__ __attribute__((aligned(16))) char smem[];
  cfloat* line = reinterpret_cast<cfloat*>(smem);
  cfloat* zlo = line + lds_len(M);       // Z[k], k < nkx
  cfloat* zhi = zlo + (g.nkx + 1);       // Z[n-k] at index k, 1 <= k <= nkx
  cfloat* stg = DIRECT ? line + lds_len(M) : zhi + (g.nkx + 1);
  const int tid = threadIdx.x;
  const int job = blockIdx.x, grp = blockIdx.y;
  const int RG = g.RG, n = ln.n;
  const float mean = RAW ? job_sub[job] : (mean_rstd ? mean_rstd[0] : 0.f);
  const float rstd = mean_rstd ? mean_rstd[1] : 1.f;
  const int expo = job_expo ? job_expo[job] : (mask ? 1 : 0);
  constexpr int SB = RAW == 1 ? 1 : RAW == 2 ? 2 : 4;
  const char* base = static_cast<const char*>(src_any) + job_off[job] * SB;
  for (int r = 0; r < RG; ++r) {
    const int y = g.y0 + grp * RG + r;
    const char* rowb = base + (int64_t)y * row_stride * SB;
    const float* grow = RAW ? gain + (int64_t)y * row_stride : nullptr;
    // one sample as the estimator sees it: the fp32 frame, or raw * gain (the mean comes off below)
    auto row_at = [&](int x) -> float {
      if constexpr (RAW == 1) return (float)reinterpret_cast<const unsigned char*>(rowb)[x] * grow[x];
      else if constexpr (RAW == 2) return (float)reinterpret_cast<const short*>(rowb)[x] * grow[x];
      else return reinterpret_cast<const float*>(rowb)[x];
    };
    const float* mrow = mask + (int64_t)y * g.W;
    if (g.W & 1) {
      // odd width: no two-samples-per-point packing; the row is a length-W complex line with zero
      // imaginary parts and the wanted bins are the first nkx outputs as they are
      auto load1 = [&](int x) {
        cfloat v = cmake(0.f, 0.f);
        if (x >= g.x0 && x < g.x1) {
          v.x = (row_at(x) - mean) * rstd;
          if (expo > 0) {
            const float m0 = mrow[x];
            for (int e = 0; e < expo; ++e) v.x *= m0;
          }
        }
        return v;
      };
      auto store1 = [&](int k, cfloat v) {
        if (k < g.nkx) stg[k * (RG + 1) + r] = v;
      };
      xcg_line_fft<LOGM, -1>(line, tid, ln, n, load1, store1, ln.keep);
      __syncthreads();
      continue;
    }
    auto load = [&](int j) {
      const int x = 2 * j;
      cfloat v = cmake(0.f, 0.f);
      if (x >= g.x0 && x < g.x1) {
        v = cmake((row_at(x) - mean) * rstd, (row_at(x + 1) - mean) * rstd);
        if (expo > 0) {
          const float m0 = mrow[x], m1 = mrow[x + 1];
          for (int e = 0; e < expo; ++e) {
            v.x *= m0;
            v.y *= m1;
          }
        }
      }
      return v;
    };
    auto store = [&](int k, cfloat v) {
      if constexpr (DIRECT) {
        line[lpad(k)] = v;  // the last pass has read all its inputs before it stores (smooth_rec)
      } else {
        if (k < g.nkx) zlo[k] = v;
        if (k > 0 && n - k <= g.nkx) zhi[n - k] = v;
      }
    };
    xcg_line_fft<LOGM, -1>(line, tid, ln, n, load, store, ln.keep);
    __syncthreads();
    for (int k = tid; k < g.nkx; k += MC_WG) {
      cfloat zk, zm;
      if constexpr (DIRECT) {
        zk = line[lpad(k < n ? k : 0)];                       // Z[n] == Z[0]
        zm = cconj(line[lpad((k == 0 || k == n) ? 0 : n - k)]);  // Z[n-k]
      } else {
        zk = (k < n) ? zlo[k] : zlo[0];
        zm = cconj((k == 0 || k == n) ? zlo[0] : zhi[k]);
      }
      const cfloat sm = cadd(zk, zm), d = csub(zk, zm);
      const cfloat w = (k < n) ? tw_row[k] : cmake(-1.f, 0.f);
      const cfloat wd = cmul(w, d);
      stg[k * (RG + 1) + r] = cmake(0.5f * (sm.x + wd.y), 0.5f * (sm.y - wd.x));
    }
    __syncthreads();
  }
  cfloat* out = T1 + (int64_t)job * g.nkx * g.ny + (int64_t)grp * RG;
  for (int i = tid; i < g.nkx * RG; i += MC_WG) {
    const int kx = i / RG, r = i - kx * RG;
    out[(int64_t)kx * g.ny + r] = stg[kx * (RG + 1) + r];
  }
}


template <int LOGM>
__global__ __launch_bounds__(MC_WG) void xcg_cols_fwd(const cfloat* __restrict__ T1,
                                                      const float* __restrict__ filt,
                                                      cfloat* __restrict__ S, XcLine ln, XcGeom g) {
  constexpr int M = mc_line_m(LOGM);
  extern __shared__ __attribute__((aligned(16))) char smem[];
  cfloat* line = reinterpret_cast<cfloat*>(smem);
  const int tid = threadIdx.x;
  const int kx = blockIdx.x, job = blockIdx.y;
  const int H = g.H, nky = g.kyp + g.kyn;
  const cfloat* col = T1 + ((int64_t)job * g.nkx + kx) * g.ny;
  cfloat* out = S + ((int64_t)job * g.nkx + kx) * nky;
  const float* f = filt ? filt + (int64_t)kx * nky : nullptr;
  auto load = [&](int y) {
    const int yy = y - g.y0;
    return (yy >= 0 && yy < g.ny) ? col[yy] : cmake(0.f, 0.f);
  };
  auto store = [&](int ky, cfloat v) {
    const int kyi = kept_index(ky, H, g.kyp, g.kyn);
    if (kyi >= 0) out[kyi] = f ? cscale(v, f[kyi]) : v;
  };
  xcg_line_fft<LOGM, -1>(line, tid, ln, H, load, store);
}

// MODE 0: conj(ref)*cur; MODE 1: cur * exp(-2 pi i (fy sy + fx sx)) (Fourier shift)
template <int LOGM, int MODE>
__global__ __launch_bounds__(MC_WG) void xcg_cols_inv(
    const cfloat* __restrict__ S_cur, const int* __restrict__ cur_idx,
    const cfloat* __restrict__ S_ref, const int* __restrict__ ref_idx,
    const float* __restrict__ shifts, cfloat* __restrict__ T2, float scale, XcLine ln, XcGeom g) {
  constexpr int M = mc_line_m(LOGM);
  extern __shared__ __attribute__((aligned(16))) char smem[];
  cfloat* line = reinterpret_cast<cfloat*>(smem);
  const int tid = threadIdx.x;
  const int kx = blockIdx.x, p = blockIdx.y;
  const int H = g.H, nky = g.kyp + g.kyn;
  const cfloat* cur = S_cur + ((int64_t)cur_idx[p] * g.nkx + kx) * nky;
  const cfloat* ref = MODE == 0 ? S_ref + ((int64_t)ref_idx[p] * g.nkx + kx) * nky : nullptr;
  cfloat* out = T2 + ((int64_t)p * g.nkx + kx) * H;
  float sy = 0.f, sx = 0.f, fx = 0.f;
  if (MODE == 1) {
    sy = shifts[2 * p];
    sx = shifts[2 * p + 1];
    fx = (float)kx * (float)(1.0 / (double)g.W);  // torch.fft.rfftfreq: k * (1/n)
  }
  auto load = [&](int ky) {
    const int kyi = kept_index(ky, H, g.kyp, g.kyn);
    if (kyi < 0) return cmake(0.f, 0.f);
    cfloat v;
    if (MODE == 0) {
      v = cmulc(ref[kyi], cur[kyi]);
    } else {
      const int kk = (ky < (H + 1) / 2) ? ky : ky - H;
      const float fy = (float)kk * (float)(1.0 / (double)H);
      const float m2pi = -6.283185307179586f;
      const float ang = (m2pi * fy) * sy + (m2pi * fx) * sx;
      float sn, cs;
      sincosf(ang, &sn, &cs);
      v = cmul(cur[kyi], cmake(cs, sn));
    }
    return cscale(v, scale);
  };
  auto store = [&](int y, cfloat v) { out[y] = v; };
  xcg_line_fft<LOGM, +1>(line, tid, ln, H, load, store);
}

template <int LOGM, int EPI>
__global__ __launch_bounds__(MC_WG) void xcg_rows_inv(
    const cfloat* __restrict__ T2, const float* __restrict__ bounds, int* __restrict__ best,
    float* __restrict__ part_val, int* __restrict__ part_idx, float* __restrict__ out_real,
    const int64_t* __restrict__ out_off, int64_t out_stride, const cfloat* __restrict__ tw_row,
    XcLine ln, XcGeom g, int near, int phase) {
  constexpr int M = mc_line_m(LOGM);
  extern __shared__ __attribute__((aligned(16))) char smem[];
  cfloat* line = reinterpret_cast<cfloat*>(smem);
  cfloat* stg = line + lds_len(M);
  const int tid = threadIdx.x;
  const int p = blockIdx.y;
  const int RG = g.RG, n = ln.n;
  const int ngrp = g.H / RG;
  int grp = blockIdx.x;
  if (EPI == 0) grp = phase == 0 ? ((int)blockIdx.x < near ? (int)blockIdx.x : ngrp - 2 * near + (int)blockIdx.x)
                                 : near + (int)blockIdx.x;
  if constexpr (EPI == 0) {
    if (phase == 1) {
      float b = 0.f;
      for (int r = 0; r < RG; ++r) b = fmaxf(b, bounds[(int64_t)p * g.H + grp * RG + r]);
      b = b * 1.0001f + 1e-30f;
      if (float_order(b) < __hip_atomic_load(&best[p], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) {
        if (tid == 0) {
          part_val[(int64_t)p * ngrp + grp] = -INFINITY;
          part_idx[(int64_t)p * ngrp + grp] = 0x7fffffff;
        }
        return;
      }
    }
  }
  const cfloat* in = T2 + (int64_t)p * g.nkx * g.H + (int64_t)grp * RG;
  for (int i = tid; i < g.nkx * RG; i += MC_WG) {
    const int kx = i / RG, r = i - kx * RG;
    stg[kx * (RG + 1) + r] = in[(int64_t)kx * g.H + r];
  }
  __syncthreads();
  float bv = -INFINITY;
  int bi = 0x7fffffff;
  for (int r = 0; r < RG; ++r) {
    const int y = grp * RG + r;
    if (g.W & 1) {
      // odd width: the full Hermitian line of W points, X[W - k] = conj(X[k]); outputs are real
      auto load1 = [&](int k) {
        cfloat v = cmake(0.f, 0.f);
        if (k < g.nkx) {
          v = stg[k * (RG + 1) + r];
          if (k == 0) v.y = 0.f;
        } else if (n - k < g.nkx) {
          v = cconj(stg[(n - k) * (RG + 1) + r]);
        }
        return v;
      };
      if constexpr (EPI == 0) {
        auto store1 = [&](int j, cfloat v) { cand_merge(bv, bi, v.x, y * g.W + j); };
        xcg_line_fft<LOGM, +1>(line, tid, ln, n, load1, store1);
      } else {
        float* orow = out_real + out_off[p] + (int64_t)y * out_stride;
        auto store1 = [&](int j, cfloat v) { orow[j] = v.x; };
        xcg_line_fft<LOGM, +1>(line, tid, ln, n, load1, store1);
      }
      __syncthreads();
      continue;
    }
    // c2r pack for a real row of even length W = 2n (same identity as the 2^k path)
    auto load = [&](int k) {
      const int km = n - k;  // in [1, n]
      cfloat xk = (k < g.nkx) ? stg[k * (RG + 1) + r] : cmake(0.f, 0.f);
      cfloat xm = (km < g.nkx) ? cconj(stg[km * (RG + 1) + r]) : cmake(0.f, 0.f);
      if (k == 0) {
        xk.y = 0.f;
        xm.y = 0.f;
      }
      const cfloat sm = cadd(xk, xm), d = csub(xk, xm);
      cfloat w = tw_row[k];
      w.y = -w.y;
      const cfloat wd = cmul(w, d);
      return cmake(sm.x - wd.y, sm.y + wd.x);
    };
    if constexpr (EPI == 0) {
      auto store = [&](int j, cfloat v) {
        const int flat = y * g.W + 2 * j;
        cand_merge(bv, bi, v.x, flat);
        cand_merge(bv, bi, v.y, flat + 1);
      };
      xcg_line_fft<LOGM, +1>(line, tid, ln, n, load, store);
    } else {
      float* orow = out_real + out_off[p] + (int64_t)y * out_stride;
      auto store = [&](int j, cfloat v) {
        orow[2 * j] = v.x;
        orow[2 * j + 1] = v.y;
      };
      xcg_line_fft<LOGM, +1>(line, tid, ln, n, load, store);
    }
    __syncthreads();
  }
  if constexpr (EPI == 0) {
    for (int off = 32; off > 0; off >>= 1) {
      const float ov = __shfl_down(bv, off);
      const int oi = __shfl_down(bi, off);
      cand_merge(bv, bi, ov, oi);
    }
    __shared__ float wv[MC_WG / 64];
    __shared__ int wi[MC_WG / 64];
    if ((tid & 63) == 0) {
      wv[tid >> 6] = bv;
      wi[tid >> 6] = bi;
    }
    __syncthreads();
    if (tid == 0) {
      for (int w = 1; w < MC_WG / 64; ++w) cand_merge(bv, bi, wv[w], wi[w]);
      part_val[(int64_t)p * ngrp + grp] = bv;
      part_idx[(int64_t)p * ngrp + grp] = bi;
      atomicMax(&best[p], float_order(bv));
    }
  }
}

#define MC_DISPATCH_LOGM(LOGV, ...)          \
  switch (LOGV) {                            \
    MC_DISPATCH_CASE(5, __VA_ARGS__)         \
    MC_DISPATCH_CASE(6, __VA_ARGS__)         \
    MC_DISPATCH_CASE(7, __VA_ARGS__)         \
    MC_DISPATCH_CASE(8, __VA_ARGS__)         \
    MC_DISPATCH_CASE(9, __VA_ARGS__)         \
    MC_DISPATCH_CASE(10, __VA_ARGS__)        \
    MC_DISPATCH_CASE(11, __VA_ARGS__)        \
    MC_DISPATCH_CASE(12, __VA_ARGS__)        \
    MC_DISPATCH_CASE(13, __VA_ARGS__)        \
    MC_DISPATCH_CASE(14, __VA_ARGS__)        \
    MC_DISPATCH_CASE(20, __VA_ARGS__)        \
    MC_DISPATCH_CASE(21, __VA_ARGS__)        \
    MC_DISPATCH_CASE(22, __VA_ARGS__)        \
    MC_DISPATCH_CASE(23, __VA_ARGS__)        \
    MC_DISPATCH_CASE(24, __VA_ARGS__)        \
    MC_DISPATCH_CASE(25, __VA_ARGS__)        \
    default:                                 \
      return MC_ERR_UNSUPPORTED;             \
  }

// geometry check without the power-of-two requirement
static int geom_from_g(const mc_xc_geom* q, XcGeom* g) {
  if (!q) return MC_ERR_ARG;
  // chirp-z lines of up to M = 16384 points (139 KB of LDS): W / 2 and H up to 8192
  // (odd widths: one real sample per point of the line, so at most 8191 columns)
  if (q->W < 4 || q->W > 16384 || ((q->W & 1) && q->W > 8191) || q->H < 2 || q->H > 8192)
    return MC_ERR_UNSUPPORTED;
  if (q->nkx < 1 || q->nkx > q->W / 2 + 1) return MC_ERR_ARG;
  if (q->kyp < 0 || q->kyn < 0 || q->kyp + q->kyn < 1 || q->kyp + q->kyn > q->H) return MC_ERR_ARG;
  if (q->RG < 1 || q->ny < 1 || q->ny % q->RG || q->H % q->RG) return MC_ERR_ARG;
  if (q->y0 < 0 || q->y0 + q->ny > q->H) return MC_ERR_ARG;
  if (q->x0 < 0 || q->x1 > q->W || q->x0 >= q->x1) return MC_ERR_ARG;
  if (!(q->W & 1) && ((q->x0 & 1) || (q->x1 & 1))) return MC_ERR_ARG;  // packed pairs: whole pairs in or out
  g->W = q->W; g->H = q->H; g->nkx = q->nkx; g->kyp = q->kyp; g->kyn = q->kyn;
  g->y0 = q->y0; g->ny = q->ny; g->x0 = q->x0; g->x1 = q->x1; g->RG = q->RG;
  return MC_OK;
}

// allow_keep: the caller's kernel understands output-pruned plans (keep > 0, M >= n + 2 keep - 1)
static int line_from(const mc_xc_line* l, int n, XcLine* out, int* logm, bool allow_keep = false,
                     int need_keep = 0) {
  if (!l || !l->tw_m || !l->chirp || !l->bspec) return MC_ERR_ARG;
  // direct plan: M == n and n is one of the mixed-radix lengths -- no chirp, tw_m = exp(-2 pi i k / n)
  const bool direct = l->M == n && l->keep == 0 && mc_line_code(n, true) >= 0;
  const int code = mc_line_code(l->M, direct);
  if (code < 0 || l->M < 32 || l->M > 16384) return MC_ERR_UNSUPPORTED;
  if (l->keep < 0 || (l->keep > 0 && !allow_keep)) return MC_ERR_ARG;
  if (!direct) {
    if (l->keep > 0) {
      if (l->keep < need_keep || l->M < n + 2 * l->keep - 1) return MC_ERR_ARG;
    } else if (l->M < 2 * n - 1) {
      return MC_ERR_UNSUPPORTED;
    }
  }
  out->tw_m = (const cfloat*)l->tw_m; out->chirp = (const cfloat*)l->chirp;
  out->bspec = (const cfloat*)l->bspec; out->n = n; out->keep = l->keep;
  *logm = code;
  return MC_OK;
}

#define MC_SET_LDS(k, bytes) \
  (void)hipFuncSetAttribute((const void*)(k), hipFuncAttributeMaxDynamicSharedMemorySize, (int)(bytes))

extern "C" {

#if XCG_PART == 0 || XCG_PART < 0
int mc_xcg_peak_neighbourhood(const void* T2, const int* peaks, float* nb, int npairs, const mc_xc_geom* q,
                              void* stream) {
  XcGeom g;
  int rc = geom_from_g(q, &g);
  if (rc) return rc;
  if (!T2 || !peaks || !nb || npairs < 1) return MC_ERR_ARG;
  hipLaunchKernelGGL(xcg_peak_nbhd, dim3(3, npairs), dim3(MC_WG), 0, (hipStream_t)stream, (const cfloat*)T2, peaks,
                     nb, g);
  return mc_check_launch();
}
#endif

#if XCG_PART == 0 || XCG_PART < 0
int mc_xcg_rows_forward(const float* src, const int64_t* job_off, int64_t row_stride,
                        const int* job_expo, const float* mask, const float* mean_rstd, void* T1,
                        const void* tw_row, const mc_xc_line* line, int njobs, const mc_xc_geom* q,
                        void* stream) {
  XcGeom g; XcLine ln; int logm;
  int rc = geom_from_g(q, &g);
  if (rc) return rc;
  // rows forward needs Z[k] for k < nkx and Z[n - k] for 1 <= k <= nkx: keep >= nkx + 1
  if ((rc = line_from(line, (g.W & 1) ? g.W : g.W / 2, &ln, &logm, true, (g.W & 1) ? g.nkx : g.nkx + 1))) return rc;
  if (!src || !job_off || !T1 || !tw_row || njobs < 1) return MC_ERR_ARG;
  const size_t lds = sizeof(cfloat) * ((size_t)lds_len(line->M) + (mc_line_direct(logm) ? 0 : 2 * (g.nkx + 1)) +
                                       (size_t)g.nkx * (g.RG + 1));
  if (lds > 160 * 1024) return MC_ERR_ARG;
  dim3 grid(njobs, g.ny / g.RG);
  MC_DISPATCH_LOGM(logm, {
    auto k = xcg_rows_fwd<L>;
    MC_SET_LDS(k, lds);
    hipLaunchKernelGGL(k, grid, dim3(MC_WG), lds, (hipStream_t)stream, (const void*)src, job_off, row_stride,
                       job_expo, mask, mean_rstd, (cfloat*)T1, (const cfloat*)tw_row, ln, g, (const float*)nullptr,
                       (const float*)nullptr);
  });
  return mc_check_launch();
}

// N2: the same row pass from the raw bytes of a u8 / i16 movie (whole-frame jobs), for the K3 formats: rows of
// 5760 / 11520 samples (direct mixed-radix lines of 2880 / 5760 points).  Other lengths: MC_ERR_UNSUPPORTED.
int mc_xcg_rows_forward_raw(const void* raw, int storage, const float* gain, const int64_t* job_off,
                            int64_t row_stride, const float* mask, const float* job_sub, const float* mean_rstd,
                            void* T1, const void* tw_row, const mc_xc_line* line, int njobs, const mc_xc_geom* q,
                            void* stream) {
  if (storage != MC_STORE_U8 && storage != MC_STORE_I16) return MC_ERR_UNSUPPORTED;
  XcGeom g; XcLine ln; int logm;
  int rc = geom_from_g(q, &g);
  if (rc) return rc;
  if (g.W & 1) return MC_ERR_UNSUPPORTED;
  if ((rc = line_from(line, g.W / 2, &ln, &logm, true, g.nkx + 1))) return rc;
  if (!raw || !gain || !job_off || !mask || !job_sub || !mean_rstd || !T1 || !tw_row || njobs < 1) return MC_ERR_ARG;
  if (logm != 22 && logm != 23) return MC_ERR_UNSUPPORTED;
  const size_t lds = sizeof(cfloat) * ((size_t)lds_len(line->M) + (size_t)g.nkx * (g.RG + 1));
  if (lds > 160 * 1024) return MC_ERR_ARG;
  dim3 grid(njobs, g.ny / g.RG);
#define MC_XCG_RAW(L, R)                                                                                         \
  do {                                                                                                           \
    auto k = xcg_rows_fwd<L, R>;                                                                                 \
    MC_SET_LDS(k, lds);                                                                                          \
    hipLaunchKernelGGL(k, grid, dim3(MC_WG), lds, (hipStream_t)stream, raw, job_off, row_stride, (const int*)nullptr, \
                       mask, mean_rstd, (cfloat*)T1, (const cfloat*)tw_row, ln, g, gain, job_sub);               \
  } while (0)
  if (logm == 22) { if (storage == MC_STORE_U8) MC_XCG_RAW(22, 1); else MC_XCG_RAW(22, 2); }
  else { if (storage == MC_STORE_U8) MC_XCG_RAW(23, 1); else MC_XCG_RAW(23, 2); }
#undef MC_XCG_RAW
  return mc_check_launch();
}
#endif

#if XCG_PART == 1 || XCG_PART < 0
int mc_xcg_cols_forward(const void* T1, const float* filt, void* S, const mc_xc_line* line,
                        int njobs, const mc_xc_geom* q, void* stream) {
  XcGeom g; XcLine ln; int logm;
  int rc = geom_from_g(q, &g);
  if (rc) return rc;
  if ((rc = line_from(line, g.H, &ln, &logm))) return rc;
  if (!T1 || !S || njobs < 1) return MC_ERR_ARG;
  const size_t lds = sizeof(cfloat) * (size_t)lds_len(line->M);
  dim3 grid(g.nkx, njobs);
  MC_DISPATCH_LOGM(logm, {
    auto k = xcg_cols_fwd<L>;
    MC_SET_LDS(k, lds);
    hipLaunchKernelGGL(k, grid, dim3(MC_WG), lds, (hipStream_t)stream, (const cfloat*)T1, filt,
                       (cfloat*)S, ln, g);
  });
  return mc_check_launch();
}
#endif

#if XCG_PART == 2 || XCG_PART < 0
int mc_xcg_cols_inverse(const void* S_cur, const int* cur_idx, const void* S_ref,
                        const int* ref_idx, const float* shifts, void* T2, const mc_xc_line* line,
                        float scale, int npairs, const mc_xc_geom* q, void* stream) {
  XcGeom g; XcLine ln; int logm;
  int rc = geom_from_g(q, &g);
  if (rc) return rc;
  if ((rc = line_from(line, g.H, &ln, &logm))) return rc;
  if (!S_cur || !cur_idx || !T2 || npairs < 1 || (!shifts && (!S_ref || !ref_idx))) return MC_ERR_ARG;
  const size_t lds = sizeof(cfloat) * (size_t)lds_len(line->M);
  dim3 grid(g.nkx, npairs);
  MC_DISPATCH_LOGM(logm, {
    if (shifts) {
      auto k = xcg_cols_inv<L, 1>;
      MC_SET_LDS(k, lds);
      hipLaunchKernelGGL(k, grid, dim3(MC_WG), lds, (hipStream_t)stream, (const cfloat*)S_cur, cur_idx,
                         (const cfloat*)nullptr, (const int*)nullptr, shifts, (cfloat*)T2, scale, ln, g);
    } else {
      auto k = xcg_cols_inv<L, 0>;
      MC_SET_LDS(k, lds);
      hipLaunchKernelGGL(k, grid, dim3(MC_WG), lds, (hipStream_t)stream, (const cfloat*)S_cur, cur_idx,
                         (const cfloat*)S_ref, ref_idx, (const float*)nullptr, (cfloat*)T2, scale, ln, g);
    }
  });
  return mc_check_launch();
}
#endif

#if XCG_PART == 3 || XCG_PART < 0
int mc_xcg_rows_inverse(const void* T2, float* part_val, int* part_idx, int* peaks, float* shifts,
                        float* out, const int64_t* out_off, int64_t out_stride, const void* tw_row,
                        const mc_xc_line* line, int npairs, const mc_xc_geom* q, void* stream) {
  XcGeom g; XcLine ln; int logm;
  int rc = geom_from_g(q, &g);
  if (rc) return rc;
  if ((rc = line_from(line, (g.W & 1) ? g.W : g.W / 2, &ln, &logm))) return rc;
  if (!T2 || !tw_row || npairs < 1) return MC_ERR_ARG;
  const bool store = out != nullptr;
  if (store ? !out_off : (!part_val || !part_idx || !peaks || !shifts)) return MC_ERR_ARG;
  const size_t lds = sizeof(cfloat) * ((size_t)lds_len(line->M) + (size_t)g.nkx * (g.RG + 1));
  if (lds > 160 * 1024) return MC_ERR_ARG;
  const int ngrp = g.H / g.RG;
  if (store) {
    MC_DISPATCH_LOGM(logm, {
      auto k = xcg_rows_inv<L, 1>;
      MC_SET_LDS(k, lds);
      hipLaunchKernelGGL(k, dim3(ngrp, npairs), dim3(MC_WG), lds, (hipStream_t)stream, (const cfloat*)T2,
                         (const float*)nullptr, (int*)nullptr, (float*)nullptr, (int*)nullptr, out, out_off,
                         out_stride, (const cfloat*)tw_row, ln, g, 0, 0);
    });
    return mc_check_launch();
  }
  int* best = part_idx + (int64_t)npairs * ngrp;
  {
    const float ninf = -INFINITY;
    int pat;
    memcpy(&pat, &ninf, 4);
    pat = pat >= 0 ? pat : pat ^ 0x7fffffff;
    hipError_t e = hipMemsetD32Async((hipDeviceptr_t)best, pat, npairs, (hipStream_t)stream);
    if (e != hipSuccess) return (int)e;
  }
  int near = (64 + g.RG - 1) / g.RG;
  if (2 * near > ngrp) near = ngrp / 2;
  float* bounds = part_val + (int64_t)npairs * ngrp;
  if (ngrp - 2 * near > 0)
    mc_launch_row_bounds((const cfloat*)T2, bounds, g.nkx, g.H, npairs, (hipStream_t)stream);
  MC_DISPATCH_LOGM(logm, {
    auto k = xcg_rows_inv<L, 0>;
    MC_SET_LDS(k, lds);
    if (near > 0)
      hipLaunchKernelGGL(k, dim3(2 * near, npairs), dim3(MC_WG), lds, (hipStream_t)stream, (const cfloat*)T2,
                         (const float*)bounds, best, part_val, part_idx, (float*)nullptr,
                         (const int64_t*)nullptr, (int64_t)0, (const cfloat*)tw_row, ln, g, near, 0);
    if (ngrp - 2 * near > 0)
      hipLaunchKernelGGL(k, dim3(ngrp - 2 * near, npairs), dim3(MC_WG), lds, (hipStream_t)stream,
                         (const cfloat*)T2, (const float*)bounds, best, part_val, part_idx, (float*)nullptr,
                         (const int64_t*)nullptr, (int64_t)0, (const cfloat*)tw_row, ln, g, near, 1);
  });
  rc = mc_check_launch();
  if (rc) return rc;
  mc_launch_peak_final(part_val, part_idx, ngrp, g.H, g.W, peaks, shifts, nullptr, npairs, (hipStream_t)stream);
  return mc_check_launch();
}
#endif

}  // extern "C"
