// Full-spectrum 2-D real transforms with a ROW-MAJOR spectrum: what
// correct_motion_fast (correct_motion.py:484-496: rfftn -> fourier_shift_dft_2d -> irfftn) and the
// exposure-filtered frame sum (examples/ttMotion.py:331-351: rfft2 -> dose_weight_movie -> irfft2 ->
// sum) run on.
//
// The pruned engine of xc_fft.hip hands its row pass output to the column pass TRANSPOSED
// (T1[job][kx][y]); with all nkx = W/2 + 1 bins kept that needs an LDS stage of nkx x RG bins per
// workgroup (147 KB for W = 4096: one workgroup per CU, 1.1 TB/s -- 1.9 ms per 15 frames, two thirds
// of correct_motion_fast's 12 ms per 40 x 4096^2 stack).  Here the spectrum stays row-major,
//     S[job][y][pitch]   complex, pitch = nkx rounded up to 16 (whole 128-byte lines per 16 columns),
// so a row pass reads or writes whole rows and needs no stage, and the COLUMN pass does the strided
// access instead: a workgroup owns two adjacent columns (16 bytes per row), stages them in two LDS
// lines, transforms forward, applies the pointwise step and transforms back IN PLACE -- one read and
// one write of S for what were two kernels and a round trip.  The eight workgroups whose column
// pairs share 128-byte lines are dispatched next to each other on one XCD (blockIdx mapping below),
// so every line is fetched from HBM once and written back whole.
//
//   full_rows_fwd   rows:  samples -> real FFT(W) -> S[job][y][0..W/2]
//   full_cols_shift cols:  S column pair -> FFT(H) -> * exp(-2 pi i (fy sy + fx sx)) / (H W) -> IFFT(H) -> S
//   full_cols_dose  cols:  sum_f q_f(k) FFT_H(S_f column) accumulated in registers over the frames
//                          of a chunk (+ A) -> A; on the last chunk / sqrt(sum q^2), IFFT(H), / (H W)
//   full_rows_inv   rows:  S[job][y][0..W/2] -> c2r pack -> IFFT(W/2) -> real rows
//
// Sizes: rows of W = 64 .. 8192 (powers of two), 5760 and 11520 columns (W / 2 = 2^a 3^2 5);
// columns of H = 256 .. 4096 (powers of two), 4092 and 8184 rows (2^a 3 11 31: radix-31 and radix-11
// passes, mc_fft.h) -- the K3 detector's two frame formats (BASELINE configs 3 and 5) run here
// without chirp-z.  Columns of more than 4096 rows go one column per workgroup (NC = 1).
#include "mc_fft.h"
#include "mcorr.h"

// blockIdx.x -> column pair: the 8 pairs of one 128-byte line group on one XCD, consecutively
// (single columns, NC = 1: the same with 16 columns per group -- npairs is then the column count / 2
// and the caller maps block b to column 2 * full_pair_of_block(b >> 1, ..) + (b & 1))
__device__ __forceinline__ int full_pair_of_block(int b, int npairs) {
  const int ngroups = (npairs + 7) / 8;
  if (ngroups < 8) return b;  // tiny widths: no mapping
  const int xcd = b & 7, i = b >> 3;
  const int grp = i >> 3, within = i & 7;
  const int G = grp * 8 + xcd;
  // groups beyond the last multiple of 8 keep the plain order
  const int full = (ngroups / 8) * 8;
  if (b >= full * 8) return b;
  return G * 8 + within;
}

// The lane index, made opaque: every transform of a kernel derives its addresses and twiddle indices
// from its own copy, so the compiler cannot keep one transform's twiddles and addresses alive for the
// next (common-subexpression elimination across the unrolled column / direction loops cost 380
// registers for a 4092-point column pair).
__device__ __forceinline__ int full_opaque(int t) {
  asm volatile("" : "+v"(t));
  return t;
}

template <int N>
__global__ __launch_bounds__(MC_WG) void full_rows_fwd(const float* __restrict__ src,
                                                       const int64_t* __restrict__ job_off,
                                                       int64_t row_stride, cfloat* __restrict__ S, int H,
                                                       int pitch, const cfloat* __restrict__ tw_row,
                                                       int rows_per_wg) {
  // N complex points = W / 2
  __shared__ __attribute__((aligned(16))) cfloat line[lds_len(N)];
  const int tid = threadIdx.x;
  const int job = blockIdx.y;
  const float* base = src + job_off[job];
  for (int r = 0; r < rows_per_wg; ++r) {
    const int y = blockIdx.x * rows_per_wg + r;
    if (y >= H) break;  // workgroup-uniform
    const float* row = base + (int64_t)y * row_stride;
    auto load = [&](int j) {
      const float2 v = *reinterpret_cast<const float2*>(row + 2 * j);
      return cmake(v.x, v.y);
    };
    auto keep = [&](int k, cfloat v) { line[lpad(k)] = v; };
    wg_fft_any<N, -1>(line, (N & (N - 1)) ? full_opaque(tid) : tid, tw_row, 2, load, keep);
    __syncthreads();
    // real-FFT unpack: X[k] = (Z[k] + conj(Z[N-k]))/2 - i/2 * w^k * (Z[k] - conj(Z[N-k])), k = 0..N
    cfloat* out = S + ((int64_t)job * H + y) * pitch;
    for (int k = tid; k <= N; k += MC_WG) {
      const cfloat zk = line[lpad(k == N ? 0 : k)];
      const cfloat zm = cconj(line[lpad(k == 0 ? 0 : N - k)]);
      const cfloat sm = cadd(zk, zm), d = csub(zk, zm);
      const cfloat w = (k < N) ? tw_row[k] : cmake(-1.f, 0.f);
      const cfloat wd = cmul(w, d);  // -i*wd = (wd.y, -wd.x)
      out[k] = cmake(0.5f * (sm.x + wd.y), 0.5f * (sm.y - wd.x));
    }
    __syncthreads();  // the next row's first pass writes the line
  }
}

template <int N>
__global__ __launch_bounds__(MC_WG) void full_rows_inv(const cfloat* __restrict__ S, float* __restrict__ out,
                                                       const int64_t* __restrict__ out_off, int64_t out_stride,
                                                       int H, int pitch, const cfloat* __restrict__ tw_row,
                                                       int rows_per_wg) {
  extern __shared__ __attribute__((aligned(16))) char smem_fr[];
  cfloat* line = reinterpret_cast<cfloat*>(smem_fr);  // lds_len(N)
  cfloat* xs = line + lds_len(N) + 1;                 // N + 1 bins of the row
  const int tid = threadIdx.x;
  const int job = blockIdx.y;
  for (int r = 0; r < rows_per_wg; ++r) {
    const int y = blockIdx.x * rows_per_wg + r;
    if (y >= H) break;
    const cfloat* in = S + ((int64_t)job * H + y) * pitch;
    for (int k = tid; k <= N; k += MC_WG) xs[k] = in[k];
    __syncthreads();
    // c2r pack: Z[k] = (X[k] + conj(X[N-k])) + i * conj(w^k) * (X[k] - conj(X[N-k]))
    auto load = [&](int k) {
      cfloat xk = xs[k];
      cfloat xm = cconj(xs[N - k]);
      if (k == 0) {  // c2r ignores the imaginary part of the DC and Nyquist bins (pocketfft)
        xk.y = 0.f;
        xm.y = 0.f;
      }
      const cfloat sm = cadd(xk, xm), d = csub(xk, xm);
      cfloat w = tw_row[k];
      w.y = -w.y;
      const cfloat wd = cmul(w, d);  // i*wd = (-wd.y, wd.x)
      return cmake(sm.x - wd.y, sm.y + wd.x);
    };
    float* orow = out + out_off[job] + (int64_t)y * out_stride;
    auto store = [&](int n, cfloat v) { *reinterpret_cast<float2*>(orow + 2 * n) = make_float2(v.x, v.y); };
    wg_fft_any<N, +1>(line, (N & (N - 1)) ? full_opaque(tid) : tid, tw_row, 2, load, store);
    __syncthreads();  // xs and the line are rewritten by the next row
  }
}

// signed frequency of row ky of an H-point transform (torch.fft.fftfreq)
__device__ __forceinline__ float full_fy(int ky, int H) {
  const int kk = (ky < (H + 1) / 2) ? ky : ky - H;
  return (float)kk * (float)(1.0 / (double)H);
}

// first column of workgroup b (NC columns per workgroup), ncols = pitch
template <int NC>
__device__ __forceinline__ int full_col_of_block(int b, int pitch) {
  if constexpr (NC == 2) return 2 * full_pair_of_block(b, pitch / 2);
  else return 2 * full_pair_of_block(b >> 1, pitch / 2) + (b & 1);
}

// stage NC adjacent columns of S (rows `pitch` apart) into NC LDS lines / write them back.  All of a
// thread's loads are issued before the first LDS write (a rolled loop waits for every load in turn:
// H / WG memory round trips per column instead of one).
template <int H, int NC, int WG>
__device__ __forceinline__ void full_cols_load(cfloat* const* lines, const cfloat* base, int64_t pitch, int tid) {
  constexpr int IT = (H + WG - 1) / WG;
  if constexpr (NC == 2) {
    float4 v[IT];
#pragma unroll
    for (int it = 0; it < IT; ++it) {
      const int i = tid + it * WG;
      if (i < H) v[it] = *reinterpret_cast<const float4*>(base + (int64_t)i * pitch);
    }
#pragma unroll
    for (int it = 0; it < IT; ++it) {
      const int i = tid + it * WG;
      if (i < H) {
        lines[0][lpad(i)] = cmake(v[it].x, v[it].y);
        lines[1][lpad(i)] = cmake(v[it].z, v[it].w);
      }
    }
  } else {
    cfloat v[IT];
#pragma unroll
    for (int it = 0; it < IT; ++it) {
      const int i = tid + it * WG;
      if (i < H) v[it] = base[(int64_t)i * pitch];
    }
#pragma unroll
    for (int it = 0; it < IT; ++it) {
      const int i = tid + it * WG;
      if (i < H) lines[0][lpad(i)] = v[it];
    }
  }
}
template <int H, int NC, int WG>
__device__ __forceinline__ void full_cols_store(cfloat* const* lines, cfloat* base, int pitch, int tid) {
  for (int i = tid; i < H; i += WG) {
    if constexpr (NC == 2) {
      const cfloat a = lines[0][lpad(i)], b = lines[1][lpad(i)];
      *reinterpret_cast<float4*>(base + (int64_t)i * pitch) = make_float4(a.x, a.y, b.x, b.y);
    } else {
      base[(int64_t)i * pitch] = lines[0][lpad(i)];
    }
  }
}

template <int H, int NC, int WG>
__global__ __launch_bounds__(WG) void full_cols_shift(cfloat* __restrict__ S, int W, int pitch,
                                                         const cfloat* __restrict__ tw_col,
                                                         const float* __restrict__ shifts, float scale) {
  extern __shared__ __attribute__((aligned(16))) char smem_fc[];
  cfloat* lines[2] = {reinterpret_cast<cfloat*>(smem_fc), reinterpret_cast<cfloat*>(smem_fc) + lds_len(H)};
  const int tid = threadIdx.x;
  const int kx0 = full_col_of_block<NC>(blockIdx.x, pitch);
  if (kx0 > W / 2) return;  // padding columns of the pitch (workgroup-uniform)
  const int job = blockIdx.y;
  cfloat* base = S + (int64_t)job * H * pitch + kx0;
  full_cols_load<H, NC, WG>(lines, base, pitch, tid);
  __syncthreads();
  const float sy = shifts[2 * job], sx = shifts[2 * job + 1];
  const float m2pi = -6.283185307179586f;
#pragma unroll
  for (int c = 0; c < NC; ++c) {
    cfloat* line = lines[c];
    const float fx = (float)(kx0 + c) * (float)(1.0 / (double)W);  // torch.fft.rfftfreq: k * (1/n)
    auto rd = [&](int i) { return line[lpad(i)]; };
    auto ramp = [&](int ky, cfloat v) {
      const float ang = (m2pi * full_fy(ky, H)) * sy + (m2pi * fx) * sx;
      float sn, cs;
      mc_sincos(ang, &sn, &cs);
      line[lpad(ky)] = cscale(cmul(v, cmake(cs, sn)), scale);
    };
    wg_fft_any_inplace<H, -1, WG>(line, full_opaque(tid), tw_col, 1, rd, ramp);
    __syncthreads();
    auto back = [&](int y, cfloat v) { line[lpad(y)] = v; };
    wg_fft_any_inplace<H, +1, WG>(line, full_opaque(tid), tw_col, 1, rd, back);
    __syncthreads();
  }
  full_cols_store<H, NC, WG>(lines, base, pitch, tid);
}

// Exposure filter of examples/ttMotion.py:331-351 (crit_exposure_bfactor = -1), as dose_accumulate_kernel
// (plan_stats.hip) defines it: q_f(k) = exp(-0.5 N_f / N_c(|k|)), N_c = (0.24499 |k|^-1.6649 + 2.8141)
// vscale, N_f = pre + dose_per_frame (f + 1), |k| in 1/Angstrom clamped at 1e-6.
__device__ __forceinline__ float full_dose_mh(int kx, int ky, int W, int H, float pixel_size, float vscale) {
  const float fy = full_fy(ky, H);
  const float fx = (float)kx * (float)(1.0 / (double)W);
  const float f = fmaxf(sqrtf(fy * fy + fx * fx) / pixel_size, 1e-6f);
  const float ncrit = (0.24499f * powf(f, -1.6649f) + 2.8141f) * vscale;
  return -0.5f / ncrit;
}

// radix of the last pass of a mixed-radix length-H transform; outputs the last pass hands to one
// thread (per column)
template <int H>
__host__ __device__ constexpr int full_last_radix() {
  int ns = 1, r = 1;
  while (ns < H) {
    r = smooth_radix(H / ns);
    ns *= r;
  }
  return r;
}
template <int H, int WG>
__host__ __device__ constexpr int full_last_slots() {
  if ((H & (H - 1)) == 0) return (H / MC_WG) > 8 ? (H / MC_WG) : 8;  // 512 / 256 points: radix 8 / 4 on 64 threads
  constexpr int r = full_last_radix<H>();
  return ((H / r + WG - 1) / WG) * r;  // iterations of the last pass x its radix
}

// Input strides (in complex elements): element (frame j, row y, column kx) of S sits at
// j sf + y sr + kx sc -- row-major spectra: (H pitch, pitch, 1); column-major copies made by
// full_transpose: (ncols H, 1, H), read with NC = 1 as contiguous columns.
template <int H, int NC, int WG>
__global__ __launch_bounds__(WG) void full_cols_dose(const cfloat* __restrict__ S, int nframes, int frame0,
                                                        int total_frames, cfloat* __restrict__ A, int W,
                                                        int pitch, const cfloat* __restrict__ tw_col,
                                                        float pixel_size, float pre_exposure,
                                                        float dose_per_frame, float vscale, int first, int last,
                                                        float scale, int64_t sf, int64_t sr, int64_t sc) {
  constexpr int SLOTS = full_last_slots<H, WG>();
  extern __shared__ __attribute__((aligned(16))) char smem_fc[];
  cfloat* lines[2] = {reinterpret_cast<cfloat*>(smem_fc), reinterpret_cast<cfloat*>(smem_fc) + lds_len(H)};
  const int tid = threadIdx.x;
  const int kx0 = full_col_of_block<NC>(blockIdx.x, pitch);
  if (kx0 > W / 2) return;  // padding columns of the pitch (workgroup-uniform)
  // mixed-radix lines (one column per workgroup): the exposure exponents of the column's rows sit in
  // LDS behind the line instead of in 24-33 registers per thread
  constexpr bool MH_LDS = (H & (H - 1)) != 0;
  static_assert(!MH_LDS || NC == 1, "mixed-radix exposure pass: one column per workgroup");
  float* mhl = reinterpret_cast<float*>(lines[0] + NC * lds_len(H));
  if constexpr (MH_LDS) {
    for (int ky = tid; ky < H; ky += WG) mhl[ky] = full_dose_mh(kx0, ky, W, H, pixel_size, vscale);
  }
  cfloat acc[NC][SLOTS];
  float mh[NC][MH_LDS ? 1 : SLOTS];
  int kys[SLOTS];  // output row of a slot (power-of-two lines: recorded; mixed radix: computed, see below)
  int nslots = 0;
#pragma unroll
  for (int s = 0; s < SLOTS; ++s) kys[s] = 0;
#pragma unroll
  for (int c = 0; c < NC; ++c)
#pragma unroll
    for (int s = 0; s < SLOTS; ++s) {
      acc[c][s] = cmake(0.f, 0.f);
      if constexpr (!MH_LDS) mh[c][s] = 0.f;
    }
  for (int j = 0; j < nframes; ++j) {
    full_cols_load<H, NC, WG>(lines, S + (int64_t)j * sf + (int64_t)kx0 * sc, sr, tid);
    __syncthreads();
    const float dose = pre_exposure + dose_per_frame * (float)(frame0 + j + 1);
#pragma unroll
    for (int c = 0; c < NC; ++c) {
      cfloat* line = lines[c];
      auto rd = [&](int i) { return line[lpad(i)]; };
      auto take3 = [&](int ky, cfloat v, int slot) {
        float m;
        if constexpr (MH_LDS) {
          m = mhl[ky];
        } else {
          if (j == 0) {
            kys[slot] = ky;
            mh[c][slot] = full_dose_mh(kx0 + c, ky, W, H, pixel_size, vscale);
          }
          m = mh[c][slot];
        }
        const float q = expf(m * dose);
        acc[c][slot].x += q * v.x;
        acc[c][slot].y += q * v.y;
      };
      if constexpr ((H & (H - 1)) == 0) {
        int slot = 0;  // the last pass calls `take` SLOTS times per thread, in a fixed (unrolled) order
        auto take = [&](int ky, cfloat v) {
          take3(ky, v, slot);
          ++slot;
        };
        wg_fft_any_inplace<H, -1, WG>(line, full_opaque(tid), tw_col, 1, rd, take);
        nslots = slot;
      } else {
        // mixed radix: the pass itself names the slot (iteration x radix + output), a compile-time
        // constant at every call site; the last iteration only runs on the threads that have a butterfly
        wg_fft_any_inplace<H, -1, WG>(line, full_opaque(tid), tw_col, 1, rd, take3);
        constexpr int R = full_last_radix<H>();
        nslots = (tid + (SLOTS / R - 1) * WG < H / R) ? SLOTS : SLOTS - R;
      }
    }
    __syncthreads();  // the next frame overwrites the lines
  }
  // accumulator columns: add what earlier chunks left in A, on the last chunk "restore the power"
  // (/ sqrt(sum_f q_f^2) over ALL frames), transform back and scale
  cfloat* abase = A + kx0;
#pragma unroll
  for (int c = 0; c < NC; ++c) {
#pragma unroll
    for (int s = 0; s < SLOTS; ++s) {
      if (s >= nslots) continue;
      cfloat a = acc[c][s];
      int ky = kys[s];
      if constexpr ((H & (H - 1)) != 0) {  // last mixed-radix pass: output j + m H/R of butterfly j = tid + it WG
        constexpr int R = full_last_radix<H>();
        ky = tid + (s / R) * WG + (s % R) * (H / R);
      }
      if (!first) {
        const cfloat prev = abase[(int64_t)ky * pitch + c];
        a.x += prev.x;
        a.y += prev.y;
      }
      if (last) {
        const float m = MH_LDS ? mhl[ky] : mh[c][MH_LDS ? 0 : s];
        float qq = 0.f;
        for (int f = 0; f < total_frames; ++f) {
          const float q = expf(m * (pre_exposure + dose_per_frame * (float)(f + 1)));
          qq += q * q;
        }
        const float r = scale / sqrtf(qq);
        a.x *= r;
        a.y *= r;
      }
      lines[c][lpad(ky)] = a;
    }
  }
  __syncthreads();
  if (last) {
#pragma unroll
    for (int c = 0; c < NC; ++c) {
      cfloat* line = lines[c];
      auto rd = [&](int i) { return line[lpad(i)]; };
      auto back = [&](int y, cfloat v) { line[lpad(y)] = v; };
      wg_fft_any_inplace<H, +1, WG>(line, full_opaque(tid), tw_col, 1, rd, back);
      __syncthreads();
    }
  }
  full_cols_store<H, NC, WG>(lines, abase, pitch, tid);
}

// ---- H = 4096: the register-resident radix-16 transform (mc_fft.h: 16 x 16 x 16, three passes,
// two exchanges through ONE 32 KiB line, 4 barriers).  Thread tid owns inputs 256 n1 + tid and
// outputs tid + 256 k3 -- the same rows -- so a column pair goes global -> registers -> forward ->
// pointwise -> inverse -> global without ever being staged: 32 KiB of LDS per workgroup instead of
// 70 (3-4 workgroups per CU instead of 2) and a third of the barriers.
__global__ __launch_bounds__(MC_WG) void full_cols_shift_r16(cfloat* __restrict__ S, int W, int pitch,
                                                             const cfloat* __restrict__ tw_col,
                                                             const float* __restrict__ shifts, float scale) {
  constexpr int H = 4096;
  __shared__ __attribute__((aligned(16))) cfloat line[H];
  const int tid = threadIdx.x;
  const int kx0 = 2 * full_pair_of_block(blockIdx.x, pitch / 2);
  if (kx0 > W / 2) return;  // padding columns of the pitch (workgroup-uniform)
  const int job = blockIdx.y;
  cfloat* base = S + (int64_t)job * H * pitch + kx0;
  cfloat v[2][16];
#pragma unroll
  for (int n1 = 0; n1 < 16; ++n1) {
    const float4 q = *reinterpret_cast<const float4*>(base + (int64_t)(256 * n1 + tid) * pitch);
    v[0][n1] = cmake(q.x, q.y);
    v[1][n1] = cmake(q.z, q.w);
  }
  const float sy = shifts[2 * job], sx = shifts[2 * job + 1];
  const float m2pi = -6.283185307179586f;
#pragma unroll
  for (int c = 0; c < 2; ++c) {
    const float fx = (float)(kx0 + c) * (float)(1.0 / (double)W);
    auto in = [&](int n1, int) { return v[c][n1]; };
    auto ramp = [&](int k, cfloat x) {
      const float ang = (m2pi * full_fy(k, H)) * sy + (m2pi * fx) * sx;
      float sn, cs;
      mc_sincos(ang, &sn, &cs);
      v[c][(k - tid) >> 8] = cscale(cmul(x, cmake(cs, sn)), scale);
    };
    wg_fft4096_r16<-1, 8, 8>(line, tid, tw_col, in, ramp);
    __syncthreads();
    auto back = [&](int k, cfloat x) { v[c][(k - tid) >> 8] = x; };
    wg_fft4096_r16<+1, 8, 8>(line, tid, tw_col, in, back);
    __syncthreads();
  }
#pragma unroll
  for (int n1 = 0; n1 < 16; ++n1)
    *reinterpret_cast<float4*>(base + (int64_t)(256 * n1 + tid) * pitch) =
        make_float4(v[0][n1].x, v[0][n1].y, v[1][n1].x, v[1][n1].y);
}

template <int NC>
__global__ __launch_bounds__(MC_WG) void full_cols_dose_r16(const cfloat* __restrict__ S, int nframes, int frame0,
                                                            int total_frames, cfloat* __restrict__ A, int W,
                                                            int pitch, const cfloat* __restrict__ tw_col,
                                                            float pixel_size, float pre_exposure,
                                                            float dose_per_frame, float vscale, int first, int last,
                                                            float scale, int64_t sf, int64_t sr, int64_t sc) {
  // NC = 1: one column per workgroup (8-byte loads; 130 registers instead of 256 + spills to AGPRs:
  // three wavefronts per SIMD instead of one)
  constexpr int H = 4096;
  __shared__ __attribute__((aligned(16))) cfloat line[H];
  const int tid = threadIdx.x;
  const int kx0 = full_col_of_block<NC>(blockIdx.x, pitch);
  if (kx0 > W / 2) return;  // padding columns of the pitch (workgroup-uniform)
  cfloat acc[NC][16];
  float mh[NC][16];
#pragma unroll
  for (int c = 0; c < NC; ++c)
#pragma unroll
    for (int k3 = 0; k3 < 16; ++k3) {
      acc[c][k3] = cmake(0.f, 0.f);
      mh[c][k3] = full_dose_mh(kx0 + c, tid + 256 * k3, W, H, pixel_size, vscale);
    }
  for (int j = 0; j < nframes; ++j) {
    const cfloat* base = S + (int64_t)j * sf + (int64_t)kx0 * sc;
    cfloat v[NC][16];
#pragma unroll
    for (int n1 = 0; n1 < 16; ++n1) {
      if constexpr (NC == 2) {
        const float4 q = *reinterpret_cast<const float4*>(base + (int64_t)(256 * n1 + tid) * sr);
        v[0][n1] = cmake(q.x, q.y);
        v[1][n1] = cmake(q.z, q.w);
      } else {
        v[0][n1] = base[(int64_t)(256 * n1 + tid) * sr];
      }
    }
    const float dose = pre_exposure + dose_per_frame * (float)(frame0 + j + 1);
#pragma unroll
    for (int c = 0; c < NC; ++c) {
      auto in = [&](int n1, int) { return v[c][n1]; };
      auto take = [&](int k, cfloat x) {
        const int k3 = (k - tid) >> 8;
        const float q = expf(mh[c][k3] * dose);
        acc[c][k3].x += q * x.x;
        acc[c][k3].y += q * x.y;
      };
      wg_fft4096_r16<-1, 8, 8>(line, tid, tw_col, in, take);
      __syncthreads();
    }
  }
  cfloat* abase = A + kx0;
#pragma unroll
  for (int c = 0; c < NC; ++c)
#pragma unroll
    for (int k3 = 0; k3 < 16; ++k3) {
      cfloat a = acc[c][k3];
      if (!first) {
        const cfloat prev = abase[(int64_t)(tid + 256 * k3) * pitch + c];
        a.x += prev.x;
        a.y += prev.y;
      }
      if (last) {
        float qq = 0.f;
        for (int f = 0; f < total_frames; ++f) {
          const float q = expf(mh[c][k3] * (pre_exposure + dose_per_frame * (float)(f + 1)));
          qq += q * q;
        }
        const float r = scale / sqrtf(qq);
        a.x *= r;
        a.y *= r;
      }
      acc[c][k3] = a;
    }
  if (last) {
#pragma unroll
    for (int c = 0; c < NC; ++c) {
      auto in = [&](int n1, int) { return acc[c][n1]; };
      auto back = [&](int k, cfloat x) { acc[c][(k - tid) >> 8] = x; };
      wg_fft4096_r16<+1, 8, 8>(line, tid, tw_col, in, back);
      __syncthreads();
    }
  }
#pragma unroll
  for (int n1 = 0; n1 < 16; ++n1) {
    if constexpr (NC == 2)
      *reinterpret_cast<float4*>(abase + (int64_t)(256 * n1 + tid) * pitch) =
          make_float4(acc[0][n1].x, acc[0][n1].y, acc[1][n1].x, acc[1][n1].y);
    else
      abase[(int64_t)(256 * n1 + tid) * pitch] = acc[0][n1];
  }
}

// S[job][y][pitch] (row-major) -> ST[job][kx][y] (column-major, kx <= W/2) through 64 x 64 LDS tiles:
// whole 512-byte row pieces in, whole 512-byte column pieces out.  The column passes use 8 or 16
// bytes of every 128-byte line of a row-major spectrum (L2 -> L1 traffic 8-16x the data: what
// bounds them); the exposure-weighted pass, which only READS the spectra of a chunk of frames,
// is fed from this copy instead: contiguous columns, one read + write pass more, less time.
__global__ __launch_bounds__(256) void full_transpose(const cfloat* __restrict__ S, cfloat* __restrict__ ST, int H,
                                                      int ncols, int pitch) {
  __shared__ __attribute__((aligned(16))) cfloat tile[64][66];
  const int job = blockIdx.z;
  const int y0 = blockIdx.y * 64, x0 = blockIdx.x * 64;
  const cfloat* src = S + (int64_t)job * H * pitch;
  cfloat* dst = ST + (int64_t)job * ncols * H;
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;  // 32 x 8: a thread moves 2 bins at a time
#pragma unroll
  for (int r = ty; r < 64; r += 8) {
    const int y = y0 + r, x = x0 + 2 * tx;
    if (y < H && x < pitch) {  // pitch is a multiple of 16: whole pairs inside the row
      const float4 v = *reinterpret_cast<const float4*>(src + (int64_t)y * pitch + x);
      tile[r][2 * tx] = cmake(v.x, v.y);
      tile[r][2 * tx + 1] = cmake(v.z, v.w);
    }
  }
  __syncthreads();
#pragma unroll
  for (int c = ty; c < 64; c += 8) {
    const int x = x0 + c, y = y0 + 2 * tx;
    if (x < ncols && y < H) {  // H is even: whole pairs inside the column
      const cfloat a = tile[2 * tx][c], b = tile[2 * tx + 1][c];
      *reinterpret_cast<float4*>(dst + (int64_t)x * H + y) = make_float4(a.x, a.y, b.x, b.y);
    }
  }
}

static bool full_rows_ok(int W) {
  return (mc_is_pow2(W) && W >= 64 && W <= 8192) || W == 5760 || W == 11520;
}
static bool full_cols_ok(int H) { return (mc_is_pow2(H) && H >= 256 && H <= 4096) || H == 4092 || H == 8184; }
static bool full_sizes_ok(int H, int W, int pitch) {
  return full_rows_ok(W) && full_cols_ok(H) && pitch >= W / 2 + 1 && (pitch % 16) == 0;
}

#define MC_FULL_SET_LDS(k, bytes) \
  (void)hipFuncSetAttribute((const void*)(k), hipFuncAttributeMaxDynamicSharedMemorySize, (int)(bytes))

#define MC_FULL_CASE(V, ...) \
  case V: {                  \
    constexpr int L = V;     \
    __VA_ARGS__              \
  } break;
// complex points of a row line (W / 2)
#define MC_FULL_DISPATCH_ROWS(NV, ...)                                                              \
  switch (NV) {                                                                                     \
    MC_FULL_CASE(32, __VA_ARGS__) MC_FULL_CASE(64, __VA_ARGS__) MC_FULL_CASE(128, __VA_ARGS__)      \
    MC_FULL_CASE(256, __VA_ARGS__) MC_FULL_CASE(512, __VA_ARGS__) MC_FULL_CASE(1024, __VA_ARGS__)   \
    MC_FULL_CASE(2048, __VA_ARGS__) MC_FULL_CASE(4096, __VA_ARGS__) MC_FULL_CASE(2880, __VA_ARGS__) \
    MC_FULL_CASE(5760, __VA_ARGS__)                                                                 \
    default: return MC_ERR_UNSUPPORTED;                                                             \
  }
// rows of a column line (H) taken by the staged kernels (4096: the register-resident kernels)
#define MC_FULL_DISPATCH_COLS(HV, ...)                                                             \
  switch (HV) {                                                                                    \
    MC_FULL_CASE(256, __VA_ARGS__) MC_FULL_CASE(512, __VA_ARGS__) MC_FULL_CASE(1024, __VA_ARGS__)  \
    MC_FULL_CASE(2048, __VA_ARGS__) MC_FULL_CASE(4092, __VA_ARGS__) MC_FULL_CASE(8184, __VA_ARGS__) \
    default: return MC_ERR_UNSUPPORTED;                                                            \
  }
// columns per workgroup: pairs while two lines fit twice into a CU's LDS, single columns above;
// threads per workgroup: 512 for 8184 rows (264 radix-31 butterflies per column)
template <int H>
constexpr int full_nc() { return H > 4096 ? 1 : 2; }
template <int H>
constexpr int full_wg() { return H > 4096 ? 512 : MC_WG; }

extern "C" {

int mc_full_spectrum_pitch(int W) { return ((W / 2 + 1) + 15) & ~15; }

int mc_full_rows_forward(const float* src, const int64_t* job_off, int64_t row_stride, void* S,
                         const void* tw_row, int njobs, int H, int W, int pitch, void* stream) {
  if (!src || !job_off || !S || !tw_row || njobs < 1) return MC_ERR_ARG;
  if (!full_sizes_ok(H, W, pitch) || (reinterpret_cast<uintptr_t>(src) & 7) || (row_stride & 1)) return MC_ERR_UNSUPPORTED;
  const int rows = 8;
  dim3 grid((H + rows - 1) / rows, njobs);
  MC_FULL_DISPATCH_ROWS(W / 2, {
    hipLaunchKernelGGL(full_rows_fwd<L>, grid, dim3(MC_WG), 0, (hipStream_t)stream, src, job_off, row_stride,
                       (cfloat*)S, H, pitch, (const cfloat*)tw_row, rows);
  });
  return mc_check_launch();
}

int mc_full_rows_inverse(const void* S, float* out, const int64_t* out_off, int64_t out_stride,
                         const void* tw_row, int njobs, int H, int W, int pitch, void* stream) {
  if (!S || !out || !out_off || !tw_row || njobs < 1) return MC_ERR_ARG;
  if (!full_sizes_ok(H, W, pitch) || (reinterpret_cast<uintptr_t>(out) & 7) || (out_stride & 1)) return MC_ERR_UNSUPPORTED;
  const int rows = 8;
  dim3 grid((H + rows - 1) / rows, njobs);
  MC_FULL_DISPATCH_ROWS(W / 2, {
    auto k = full_rows_inv<L>;
    const size_t lds = sizeof(cfloat) * ((size_t)lds_len(L) + 1 + L + 2);
    MC_FULL_SET_LDS(k, lds);
    hipLaunchKernelGGL(k, grid, dim3(MC_WG), lds, (hipStream_t)stream, (const cfloat*)S, out, out_off,
                       out_stride, H, pitch, (const cfloat*)tw_row, rows);
  });
  return mc_check_launch();
}

int mc_full_cols_shift(void* S, const float* shifts, const void* tw_col, float scale, int njobs, int H, int W,
                       int pitch, void* stream) {
  if (!S || !shifts || !tw_col || njobs < 1) return MC_ERR_ARG;
  if (!full_sizes_ok(H, W, pitch)) return MC_ERR_UNSUPPORTED;
  if (H == 4096) {
    hipLaunchKernelGGL(full_cols_shift_r16, dim3(pitch / 2, njobs), dim3(MC_WG), 0, (hipStream_t)stream, (cfloat*)S,
                       W, pitch, (const cfloat*)tw_col, shifts, scale);
    return mc_check_launch();
  }
  MC_FULL_DISPATCH_COLS(H, {
    constexpr int NC = full_nc<L>(), WG = full_wg<L>();
    auto k = full_cols_shift<L, NC, WG>;
    const size_t lds = NC * sizeof(cfloat) * (size_t)lds_len(L);
    MC_FULL_SET_LDS(k, lds);
    hipLaunchKernelGGL(k, dim3(pitch / NC, njobs), dim3(WG), lds, (hipStream_t)stream, (cfloat*)S, W, pitch,
                       (const cfloat*)tw_col, shifts, scale);
  });
  return mc_check_launch();
}

int mc_full_transpose(const void* S, void* ST, int njobs, int H, int W, int pitch, void* stream) {
  if (!S || !ST || njobs < 1) return MC_ERR_ARG;
  if (!full_sizes_ok(H, W, pitch) || (H & 1)) return MC_ERR_UNSUPPORTED;
  const int ncols = W / 2 + 1;
  hipLaunchKernelGGL(full_transpose, dim3((ncols + 63) / 64, (H + 63) / 64, njobs), dim3(256), 0, (hipStream_t)stream,
                     (const cfloat*)S, (cfloat*)ST, H, ncols, pitch);
  return mc_check_launch();
}

static int full_cols_dose_impl(const void* S, bool colmajor, int nframes, int frame0, int total_frames, void* A,
                               const void* tw_col, int H, int W, int pitch, float pixel_size, float pre_exposure,
                               float dose_per_frame, float voltage, int first, int last, float scale, void* stream);

int mc_full_cols_dose(const void* S, int nframes, int frame0, int total_frames, void* A, const void* tw_col,
                      int H, int W, int pitch, float pixel_size, float pre_exposure, float dose_per_frame,
                      float voltage, int first, int last, float scale, void* stream) {
  return full_cols_dose_impl(S, false, nframes, frame0, total_frames, A, tw_col, H, W, pitch, pixel_size, pre_exposure,
                             dose_per_frame, voltage, first, last, scale, stream);
}

int mc_full_cols_dose_cm(const void* ST, int nframes, int frame0, int total_frames, void* A, const void* tw_col,
                         int H, int W, int pitch, float pixel_size, float pre_exposure, float dose_per_frame,
                         float voltage, int first, int last, float scale, void* stream) {
  if (H != 4096 && H != 4092 && H != 8184) return MC_ERR_UNSUPPORTED;  // the one-column-per-workgroup kernels
  return full_cols_dose_impl(ST, true, nframes, frame0, total_frames, A, tw_col, H, W, pitch, pixel_size, pre_exposure,
                             dose_per_frame, voltage, first, last, scale, stream);
}

static int full_cols_dose_impl(const void* S, bool colmajor, int nframes, int frame0, int total_frames, void* A,
                               const void* tw_col, int H, int W, int pitch, float pixel_size, float pre_exposure,
                               float dose_per_frame, float voltage, int first, int last, float scale, void* stream) {
  if (!S || !A || !tw_col || nframes < 1 || frame0 < 0 || total_frames < frame0 + nframes || !(pixel_size > 0.f) ||
      !(dose_per_frame >= 0.f))
    return MC_ERR_ARG;
  if (!full_sizes_ok(H, W, pitch)) return MC_ERR_UNSUPPORTED;
  const float vscale = voltage >= 300.f ? 1.0f : (voltage >= 200.f ? 0.8f : 0.75f);
  const int64_t sf = colmajor ? (int64_t)(W / 2 + 1) * H : (int64_t)H * pitch;
  const int64_t sr = colmajor ? 1 : pitch, sc = colmajor ? H : 1;
  if (H == 4096) {
    hipLaunchKernelGGL(full_cols_dose_r16<1>, dim3(pitch), dim3(MC_WG), 0, (hipStream_t)stream, (const cfloat*)S,
                       nframes, frame0, total_frames, (cfloat*)A, W, pitch, (const cfloat*)tw_col, pixel_size,
                       pre_exposure, dose_per_frame, vscale, first, last, scale, sf, sr, sc);
    return mc_check_launch();
  }
  MC_FULL_DISPATCH_COLS(H, {
    // mixed-radix columns: one column per workgroup (two columns' accumulators and exposure exponents
    // on top of the radix-31 pass need 300 registers: one wavefront per SIMD)
    constexpr int NC = (L & (L - 1)) ? 1 : full_nc<L>(), WG = full_wg<L>();
    auto k = full_cols_dose<L, NC, WG>;
    const size_t lds = NC * sizeof(cfloat) * (size_t)lds_len(L) + ((L & (L - 1)) ? sizeof(float) * (size_t)L : 0);
    MC_FULL_SET_LDS(k, lds);
    hipLaunchKernelGGL(k, dim3(pitch / NC), dim3(WG), lds, (hipStream_t)stream, (const cfloat*)S, nframes, frame0,
                       total_frames, (cfloat*)A, W, pitch, (const cfloat*)tw_col, pixel_size, pre_exposure,
                       dose_per_frame, vscale, first, last, scale, sf, sr, sc);
  });
  return mc_check_launch();
}

}  // extern "C"
