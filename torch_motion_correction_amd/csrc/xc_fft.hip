// Pruned, separable 2-D real FFT engine with fused prologues/epilogues for the
// cross-correlation shift search (reference: estimate_motion_xc.py:76-123 for whole
// frames, :338-355 for patches).
//
// The reference materialises full spectra and full correlation maps.  Here the
// binary band-pass (utils.py:104-112) and the finite support of the circular mask
// (xc.py:69-74) are exploited exactly: spectrum bins the band-pass zeroes are never
// produced (only nkx columns and kyp+kyn rows are kept) and image rows/columns the
// mask zeroes are never read.  Zero contributions are skipped, nothing is
// approximated.
//
//   K1 xc_rows_fwd   rows:  gather + (x-mean)*rstd*mask^e -> real FFT(W) -> first nkx bins
//                           -> T1[job][kx][ysupport]            (transposed via LDS)
//   K2 xc_cols_fwd   cols:  T1 column -> FFT(H) -> kept ky rows * filter -> S[job][kx][kyi]
//   K3 xc_cols_inv   cols:  conj(S_ref)*S_cur -> inverse FFT(H) -> T2[pair][kx][y]
//   K4 xc_rows_inv   rows:  T2 rows -> inverse real FFT(W) -> fused arg-max | store
//   K5 xc_peak_final       reduce K4's per-workgroup candidates, decode wrap-around
//   K6 xc_peak_nbhd        re-evaluate rows y-1,y,y+1 of one map for the parabola fit
#include <stdlib.h>
#include <string.h>
#include <type_traits>
#include "xc_common.h"
#include "mc_wave_fft.h"


// ------------------------------------------------------------------ K1: rows forward
// Two LDS lines (ping-pong: one barrier per pass), twiddles in registers for the whole
// row loop, and the next row's samples + mask values already in flight (registers)
// while the current row is transformed.


// Sub-groups: fft_threads(N) threads cooperate on one row, MC_WG / that many rows are in
// flight per workgroup (N = 2048: the whole workgroup on one row; N = 512: one wavefront
// per row, four rows at a time).  Each sub-group owns a pair of LDS lines (ping-pong: one
// barrier per pass); twiddles live in registers for the whole row loop.
// RAW (N2): 1 = u8, 2 = i16 samples conditioned on the fly as raw * gain - job_sub[job] (whole-frame jobs:
// `gain` has the frames' row pitch); no statistics then.
template <int LOGN, bool STATS, int RAW = 0>
__global__ __launch_bounds__(MC_WG) void xc_rows_fwd(
    const void* __restrict__ src_any, const int64_t* __restrict__ job_off, int64_t row_stride,
    const int* __restrict__ job_expo, const float* __restrict__ mask,
    const float* __restrict__ mean_rstd, cfloat* __restrict__ T1,
    const cfloat* __restrict__ tw_row, XcGeom g, XcBox box, double* __restrict__ stats_acc,
    const float* __restrict__ gain, const float* __restrict__ job_sub) {
  constexpr int N = 1 << LOGN;  // complex length = W/2
  constexpr int NT = fft_threads(N), SG = MC_WG / NT;
  constexpr int R0 = FftPlan<N>::radix(0), NB0 = N / R0, IT0 = (NB0 + NT - 1) / NT;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x;
  const int lt = tid & (NT - 1), sg = tid / NT;
  cfloat* l0 = reinterpret_cast<cfloat*>(smem) + sg * 2 * lds_len(N);
  cfloat* l1 = l0 + lds_len(N);
  cfloat* stg = reinterpret_cast<cfloat*>(smem) + SG * 2 * lds_len(N);
  // job is the fastest grid dimension: the workgroups that share a row group's mask rows
  // (one per job) are dispatched together and find them in L2
  const int job = blockIdx.x;
  const int grp = blockIdx.y;
  const int RG = g.RG;
  const float mean = RAW ? job_sub[job] : (mean_rstd ? mean_rstd[0] : 0.f);
  const float rstd = mean_rstd ? mean_rstd[1] : 1.f;
  const int expo = job_expo ? job_expo[job] : (mask ? 1 : 0);
  constexpr int SB = RAW == 1 ? 1 : RAW == 2 ? 2 : 4;
  const char* base = static_cast<const char*>(src_any) + job_off[job] * SB;
  FftTwiddles<N> T;
  T.template init<-1>(lt, tw_row, 2);

  int s = 0;
  float st_s = 0.f, st_q = 0.f;  // sum and sum of squares of (x - mean_rstd[0]) inside the box
  for (int r = sg; r < RG; r += SG) {  // RG % SG == 0: every sub-group runs the same trip count
    const int y = g.y0 + grp * RG + r;
    const char* rowb = base + (int64_t)y * row_stride * SB;
    const float* grow = RAW ? gain + (int64_t)y * row_stride : nullptr;
    auto row_at = [&](int xx) -> float {
      if constexpr (RAW == 1) return (float)reinterpret_cast<const unsigned char*>(rowb)[xx] * grow[xx];
      else if constexpr (RAW == 2) return (float)reinterpret_cast<const short*>(rowb)[xx] * grow[xx];
      else return reinterpret_cast<const float*>(rowb)[xx];
    };
    const float* mrow = mask + (int64_t)y * g.W;
    cfloat px[IT0][R0], mk[IT0][R0];
#pragma unroll
    for (int it = 0; it < IT0; ++it) {
      const int j = lt + it * NT;
#pragma unroll
      for (int q = 0; q < R0; ++q) {
        const int x = 2 * (j + q * NB0);
        const bool on = (NB0 % NT == 0 || j < NB0) && x >= g.x0 && x < g.x1;
        px[it][q] = on ? cmake(row_at(x), row_at(x + 1)) : cmake(mean, mean);
        mk[it][q] = (on && expo > 0) ? cmake(mrow[x], mrow[x + 1]) : cmake(on ? 1.f : 0.f, on ? 1.f : 0.f);
      }
    }
    if constexpr (STATS) {
      if (y >= box.hl && y < box.hu) {
#pragma unroll
        for (int it = 0; it < IT0; ++it)
#pragma unroll
          for (int q = 0; q < R0; ++q) {
            const int x = 2 * (lt + it * NT + q * NB0);
            if ((NB0 % NT == 0 || lt + it * NT < NB0) && x >= box.wl && x < box.wu) {
              // box.wl/wu are even (host guarantees), so x+1 is inside too
              const float a = px[it][q].x - mean, b = px[it][q].y - mean;
              st_s += a + b;
              st_q += a * a + b * b;
            }
          }
      }
    }
    auto load = [&](int, int it, int q) {
      cfloat v = cmake((px[it][q].x - mean) * rstd, (px[it][q].y - mean) * rstd);
      const cfloat mm = mk[it][q];
      v.x *= mm.x;
      v.y *= mm.y;
      for (int e = 1; e < expo; ++e) {
        v.x *= mm.x;
        v.y *= mm.y;
      }
      return v;
    };
    auto nostore = [](int, cfloat) {};
    const int res = wg_fft_pp<N, -1, false>(l0, l1, s, lt, T, load, nostore);
    const cfloat* Z = res ? l1 : l0;
    // real-FFT unpack: X[k] = (Z[k] + conj(Z[N-k]))/2 - i/2 * w^k * (Z[k] - conj(Z[N-k]))
    for (int k = lt; k < g.nkx; k += NT) {
      const cfloat zk = Z[lpad(k & (N - 1))];
      const cfloat zm = cconj(Z[lpad((N - k) & (N - 1))]);
      const cfloat sm = cadd(zk, zm), d = csub(zk, zm);
      const cfloat w = (k < N) ? tw_row[k] : cmake(-1.f, 0.f);
      const cfloat wd = cmul(w, d);  // -i*wd = (wd.y, -wd.x)
      stg[k * (RG + 1) + r] = cmake(0.5f * (sm.x + wd.y), 0.5f * (sm.y - wd.x));
    }
    s = res ^ 1;  // next row must not start in the line that is still being unpacked
  }
  __syncthreads();
  cfloat* out = T1 + (int64_t)job * g.nkx * g.ny + (int64_t)grp * RG;
  for (int i = tid; i < g.nkx * RG; i += MC_WG) {
    const int kx = i / RG, r = i - kx * RG;
    out[(int64_t)kx * g.ny + r] = stg[kx * (RG + 1) + r];
  }
  if constexpr (STATS) {
    double ds = st_s, dq = st_q;
    for (int off = 32; off > 0; off >>= 1) {
      ds += __shfl_down(ds, off);
      dq += __shfl_down(dq, off);
    }
    __shared__ double rs[MC_WG / 64], rq[MC_WG / 64];
    if ((tid & 63) == 0) {
      rs[tid >> 6] = ds;
      rq[tid >> 6] = dq;
    }
    __syncthreads();
    if (tid == 0) {
      for (int w = 1; w < MC_WG / 64; ++w) {
        ds += rs[w];
        dq += rq[w];
      }
      if (ds != 0.0 || dq != 0.0) {
        atomicAdd(&stats_acc[0], ds);
        atomicAdd(&stats_acc[1], dq);
      }
    }
  }
}

// ------------------------------------------------------------------ K1, wave per row
// W = 4096 rows (N = 2048 complex points) with nkx <= 512: one wavefront transforms one
// row on its own (mc_wave_fft.h): no workgroup barrier anywhere in the row loop, 8 KiB of
// LDS per wave, so 12-16 independent row streams per CU keep their HBM loads in flight.
// A wave takes 4 consecutive rows (two pairs), a workgroup 16: bins of a row pair leave
// as one 16-byte store per (kx, pair) and a workgroup completes whole 128-byte lines of
// T1[job][kx][y].  Same arithmetic as xc_rows_fwd up to the summation order of the FFT.
__device__ __forceinline__ void wf_sync() {
  // wave-private LDS hand-off: DS operations of one wave execute in order, so this only
  // has to stop the compiler from moving a lane's reads above other lanes' writes
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// Scheduling pin: `v` (an index / offset every later address is derived from) becomes
// opaque at the point where `dep` has been computed, so the loads that use it cannot be
// hoisted above that point (the compiler otherwise issues every table read at the top of
// the row and pays for it with ~60 registers each).
__device__ __forceinline__ void wf_pin(int& v, float dep) { asm volatile("" : "+v"(v) : "v"(dep)); }

#define XC_STAT_SLOTS 64  // stats_acc = XC_STAT_SLOTS x {sum, sumsq} doubles

// Twiddles come from two LDS tables shared by the workgroup's four waves (filled once from
// tw_row, exact table values):
//   twA[s][q]            = W_2048^{q 2^s}        s = 0..3, q < 128     (4 KiB)
//   twB[g][k2 - 1][h]    = W_128^{(2 g + h) k2}  k2 = 1..15            (960 B)
// Lane t reads twA[s][2t..2t+1] and twB[t>>4][k2-1][0..1] as one 16-byte LDS read each
// (the latter a broadcast within 16 lanes).  Pass A's fifteen twiddles W^{q k1} are the four exact
// bases k1 = 1, 2, 4, 8 and eleven products of them (wf_twiddle16, as the 1024- and 4096-point column
// engines do): the table of all fifteen was 15 KiB, and with it a workgroup's 51 KB of LDS allowed three
// workgroups per CU; 40 KB allow four.  The kernel is bound by how many waves are there to issue (49 %
// VALU, 37 % LDS, 62 % of the HBM ceiling; a wave issues at most one VALU instruction per ~6 cycles,
// scripts/ubench/pk_rate.hip), so the fourth wave per SIMD is worth more than the 44 extra instructions
// per row.
#define WF_TWA (4 * 128)
#define WF_TWB (4 * 15 * 2)
#ifndef WF_ROWS_PER_WG
#define WF_ROWS_PER_WG 32  // rounds of 8 rows: a wave takes rows 2 wv and 2 wv + 1 of every round (16: the prologue is 22 % of a wave's life; 32: K1 445 -> 430 us)
#endif
#ifndef WF_PREFETCH_DEFAULT
#define WF_PREFETCH_DEFAULT 1  // 1: next row's samples, 2: and mask row, loaded under the current transform
#endif
#ifndef WF_MIN_WG
#define WF_MIN_WG 2  // workgroups per CU the register allocation aims at (2: 256 VGPRs per lane, 3: 168)
#endif

// N1LO / N1HI: only the 256-sample chunks [N1LO, N1HI) of a row can touch the mask support,
// the others are zero and are not loaded.  CLAMP_ALL = false: the chunks strictly between
// N1LO and N1HI - 1 lie wholly inside the support (host checks) and load unclamped.
// Statistics: box.wl / box.wu are multiples of 256 (host checks), so a chunk is inside
// the box or outside it as a whole.
template <int N1LO, int N1HI, bool CLAMP_ALL, bool HALF = false, int RAW = 0>
__device__ __forceinline__ void wf_load_px(const void* __restrict__ row_any, int t, int xlo, int xhi,
                                           float4 (&px)[16]) {
  if constexpr (RAW == 1) {
    // u8 storage (N2): the lane's four samples are ONE dword, kept raw in px[n1].x and widened where the
    // row is consumed (wf_row), exactly as the fp16 form does
    const unsigned char* row = static_cast<const unsigned char*>(row_any);
#pragma unroll
    for (int n1 = N1LO; n1 < N1HI; ++n1) {
      const int x = 256 * n1 + 4 * t;
      const int xs = (CLAMP_ALL || n1 == N1LO || n1 == N1HI - 1) ? min(max(x, xlo), xhi) : x;
      px[n1].x = __builtin_nontemporal_load(reinterpret_cast<const float*>(row + xs));
    }
    return;
  }
  if constexpr (HALF || RAW == 2) {
    // fp16 storage: the lane's four samples are 8 bytes; they stay RAW in px[n1].x / .y (converting here
    // would make the prefetch wait for its own loads) and are widened where the row is consumed
    const _Float16* row = static_cast<const _Float16*>(row_any);  // (or int16: the same 8 raw bytes)
#pragma unroll
    for (int n1 = N1LO; n1 < N1HI; ++n1) {
      const int x = 256 * n1 + 4 * t;
      typedef float f2 __attribute__((ext_vector_type(2)));
      const int xs = (CLAMP_ALL || n1 == N1LO || n1 == N1HI - 1) ? min(max(x, xlo), xhi) : x;
      const f2 q = __builtin_nontemporal_load(reinterpret_cast<const f2*>(row + xs));
      px[n1].x = q.x;
      px[n1].y = q.y;
    }
    return;
  }
  const float* row = static_cast<const float*>(row_any);
  // Branch-free: a lane whose quad lies outside [xlo, xhi + 4) -- the support box, or with
  // CLAMP_ALL and a chord table this row's own chord of the mask disk -- reads the nearest quad
  // inside it instead (a line its neighbours fetch anyway: no extra HBM traffic); the value
  // is later multiplied by the mask's exact zero.
#pragma unroll
  for (int n1 = N1LO; n1 < N1HI; ++n1) {
    const int x = 256 * n1 + 4 * t;
    // read-once stream: non-temporal, so that it does not push the mask rows out of L2
    typedef float f4 __attribute__((ext_vector_type(4)));
    const int xs = (CLAMP_ALL || n1 == N1LO || n1 == N1HI - 1) ? min(max(x, xlo), xhi) : x;
    const f4 q = __builtin_nontemporal_load(reinterpret_cast<const f4*>(row + xs));
    px[n1] = make_float4(q.x, q.y, q.z, q.w);
  }
}

template <int N1LO, int N1HI>
__device__ __forceinline__ void wf_load_mask(const float* __restrict__ mrow, int t, float4 (&mk)[16]) {
  // mask rows (L2-resident) are read as they are: exact zeros outside the support
#pragma unroll
#ifdef MC_K1_NOMASK  // timing experiment only (wrong results): what the mask loads cost
  for (int n1 = N1LO; n1 < N1HI; ++n1) mk[n1] = make_float4(1.f, 1.f, 1.f, 1.f);
#else
  for (int n1 = N1LO; n1 < N1HI; ++n1) mk[n1] = *reinterpret_cast<const float4*>(mrow + 256 * n1 + 4 * t);
#endif
}

// One row.  PREFETCH 0: px / mk are loaded here.  1: px holds this row's samples on entry
// and the next row's on exit (loaded right after the current ones were consumed, so the
// HBM latency of row i+1 hides behind the transform of row i); 2: the same for mk too.
// next_row / next_mrow are null after the last row (wave-uniform).
// raw bits of two fp16 samples -> two floats
__device__ __forceinline__ wf2 wf_unpack_h2(float bits) {
  const unsigned v = __float_as_uint(bits);
  return wf2{(float)__builtin_bit_cast(_Float16, (unsigned short)(v & 0xffffu)),
             (float)__builtin_bit_cast(_Float16, (unsigned short)(v >> 16))};
}

// raw bits of four u8 samples (one dword) / two i16 samples -> floats
__device__ __forceinline__ void wf_unpack_u8x4(float bits, wf2& a01, wf2& a23) {
  const unsigned v = __float_as_uint(bits);
  a01 = wf2{(float)(v & 0xffu), (float)((v >> 8) & 0xffu)};
  a23 = wf2{(float)((v >> 16) & 0xffu), (float)(v >> 24)};
}
__device__ __forceinline__ wf2 wf_unpack_i16x2(float bits) {
  const unsigned v = __float_as_uint(bits);
  return wf2{(float)(short)(v & 0xffffu), (float)((int)v >> 16)};
}

template <int KEEP, bool STATS, int N1LO, int N1HI, bool CLAMP_ALL, int PREFETCH, bool HALF = false, int RAW = 0>
__device__ __forceinline__ void wf_row(float4 (&px)[16], float4 (&mk)[16],
                                       const void* __restrict__ row, const float* __restrict__ mrow,
                                       const void* __restrict__ next_row,
                                       const float* __restrict__ next_mrow, int t, wf2* slab,
                                       const cfloat* twA, const cfloat* twB, const cfloat* twK,
                                       const XcGeom& g, int box_lo, int box_hi, float mean, float rstd,
                                       float& st_s, float& st_q, wf2 (&X)[4][KEEP], int xlo, int xhi,
                                       int nxlo, int nxhi, const float* __restrict__ grow = nullptr) {
  wf2 A0[16], A1[16];
  if (PREFETCH < 1) wf_load_px<N1LO, N1HI, CLAMP_ALL, HALF, RAW>(row, t, xlo, xhi, px);
  if constexpr (RAW != 0) {
    // N2: A = (raw * gain - sub_f) * rstd * mask, `mean` = sub_f = frame mean + box mean (mc_raw_movie_stats).
    // Gain and mask values are fetched and consumed in two half-row groups: all 32 float4 of a row at
    // once would be 128 registers next to the 64 of A0 / A1.
    auto half_row = [&](auto lo_tag) {
      constexpr int LO = decltype(lo_tag)::value, HI = LO + 8;
      float4 gq[8], mq[8];
#pragma unroll
      for (int n1 = LO; n1 < HI; ++n1) {
        if (n1 >= N1LO && n1 < N1HI) {
          const int x = 256 * n1 + 4 * t;
          const int xs = (CLAMP_ALL || n1 == N1LO || n1 == N1HI - 1) ? min(max(x, xlo), xhi) : x;
          gq[n1 - LO] = *reinterpret_cast<const float4*>(grow + xs);  // the sample's own (clamped) column
          mq[n1 - LO] = *reinterpret_cast<const float4*>(mrow + x);
        }
      }
#pragma unroll
      for (int n1 = LO; n1 < HI; ++n1) {
        if (n1 >= N1LO && n1 < N1HI) {
          wf2 r01, r23;
          if constexpr (RAW == 1) wf_unpack_u8x4(px[n1].x, r01, r23);
          else { r01 = wf_unpack_i16x2(px[n1].x); r23 = wf_unpack_i16x2(px[n1].y); }
          const float4 gv = gq[n1 - LO], mv = mq[n1 - LO];
          const wf2 a01 = __builtin_elementwise_fma(r01, wf2{gv.x, gv.y}, wf2{-mean, -mean});
          const wf2 a23 = __builtin_elementwise_fma(r23, wf2{gv.z, gv.w}, wf2{-mean, -mean});
          A0[n1] = (a01 * rstd) * wf2{mv.x, mv.y};
          A1[n1] = (a23 * rstd) * wf2{mv.z, mv.w};
        } else {
          A0[n1] = wf2{0.f, 0.f};
          A1[n1] = wf2{0.f, 0.f};
        }
      }
    };
    half_row(std::integral_constant<int, 0>{});
    {
      int tp = t;
      wf_pin(tp, A1[7].y);  // the second group's loads start once the first group has been consumed
      t = tp;
    }
    half_row(std::integral_constant<int, 8>{});
  }
  if (PREFETCH < 2 && RAW == 0) wf_load_mask<N1LO, N1HI>(mrow, t, mk);
  auto condition = [&](auto in_box) {
    constexpr bool INBOX = decltype(in_box)::value;
    wf2 acc_s = {0.f, 0.f}, acc_q = {0.f, 0.f};
#pragma unroll
    for (int n1 = 0; n1 < 16; ++n1) {
      if (n1 >= N1LO && n1 < N1HI) {
        const wf2 a01 = (HALF ? wf_unpack_h2(px[n1].x) : wf2{px[n1].x, px[n1].y}) - mean;
        const wf2 a23 = (HALF ? wf_unpack_h2(px[n1].y) : wf2{px[n1].z, px[n1].w}) - mean;
        if (INBOX) {  // chunk weight 1 inside the box, 0 outside (scalar): no branch per chunk
          const float cw = (n1 >= box_lo && n1 < box_hi) ? 1.f : 0.f;
          const wf2 sa = a01 + a23;
          const wf2 sq = __builtin_elementwise_fma(a01, a01, a23 * a23);
          acc_s = __builtin_elementwise_fma(sa, wf2{cw, cw}, acc_s);
          acc_q = __builtin_elementwise_fma(sq, wf2{cw, cw}, acc_q);
        }
        A0[n1] = (a01 * rstd) * wf2{mk[n1].x, mk[n1].y};
        A1[n1] = (a23 * rstd) * wf2{mk[n1].z, mk[n1].w};
      } else {
        A0[n1] = wf2{0.f, 0.f};
        A1[n1] = wf2{0.f, 0.f};
      }
    }
    if (INBOX) {
      st_s += acc_s.x + acc_s.y;
      st_q += acc_q.x + acc_q.y;
    }
  };
  if constexpr (RAW == 0) {
    if (STATS && box_hi > box_lo) condition(std::true_type{});  // wave-uniform: a row of the box
    else condition(std::false_type{});
  }
  int tl = t;  // lane index as the tables see it (re-pinned before each table)
  wf_pin(tl, A0[N1HI - 1].x);
  const WfLane L = wf_lane(tl);  // slab addresses: derived here, not carried across rows
  if (PREFETCH >= 1) {
    if (next_row) {  // issued once this row's samples have been consumed, not earlier
      int tp = t;
      wf_pin(tp, A1[N1HI - 1].y);
      wf_load_px<N1LO, N1HI, CLAMP_ALL, HALF, RAW>(next_row, tp, nxlo, nxhi, px);
      if (PREFETCH >= 2 && RAW == 0) wf_load_mask<N1LO, N1HI>(next_mrow, tp, mk);
    }
  }
  wf_dft16(A0);
  wf_pin(tl, A0[15].y);  // table reads fly under the second butterfly
  {
    const float4* twa = reinterpret_cast<const float4*>(twA) + tl;  // [s][64 lanes] of 16 B: q = 2 t, 2 t + 1
    float4 w[4];
#pragma unroll
    for (int sb = 0; sb < 4; ++sb) w[sb] = twa[sb * 64];
    wf_dft16(A1);
    wf_twiddle16(A0, wf2{w[0].x, w[0].y}, wf2{w[1].x, w[1].y}, wf2{w[2].x, w[2].y}, wf2{w[3].x, w[3].y});
    wf_twiddle16(A1, wf2{w[0].z, w[0].w}, wf2{w[1].z, w[1].w}, wf2{w[2].z, w[2].w}, wf2{w[3].z, w[3].w});
  }

  wf2 B0[16], B1[16];
#pragma unroll
  for (int k1 = 0; k1 < 16; ++k1) slab[L.x1w_base + (k1 ^ L.x1w_mask)] = A0[k1];
  wf_sync();
#pragma unroll
  for (int n2 = 0; n2 < 16; ++n2) B0[n2] = slab[L.x1r[n2 & 3] + 64 * n2];
  wf_sync();
#pragma unroll
  for (int k1 = 0; k1 < 16; ++k1) slab[L.x1w_base + (k1 ^ L.x1w_mask)] = A1[k1];
  wf_sync();
  wf_dft16(B0);
#pragma unroll
  for (int n2 = 0; n2 < 16; ++n2) B1[n2] = slab[L.x1r[n2 & 3] + 64 * n2];
  wf_sync();
  wf_pin(tl, B0[15].y);
  {
    const float4* twb = reinterpret_cast<const float4*>(twB) + (tl >> 4) * 15;
    float4 w[15];
#pragma unroll
    for (int k2 = 1; k2 < 16; ++k2) w[k2 - 1] = twb[k2 - 1];
    wf_dft16(B1);
#pragma unroll
    for (int k2 = 1; k2 < 16; ++k2) {
      B0[k2] = wf_cmul(B0[k2], wf2{w[k2 - 1].x, w[k2 - 1].y});
      B1[k2] = wf_cmul(B1[k2], wf2{w[k2 - 1].z, w[k2 - 1].w});
    }
  }
#pragma unroll
  for (int k2 = 0; k2 < 16; ++k2) slab[L.x2w + 16 * k2] = B0[k2];
  wf_sync();
  wf2 Ce[4][4], Co[4][4];
#pragma unroll
  for (int s = 0; s < 4; ++s)
#pragma unroll
    for (int n3h = 0; n3h < 4; ++n3h) Ce[s][n3h] = slab[L.x2r[s] + 256 * n3h];
  wf_sync();
#pragma unroll
  for (int k2 = 0; k2 < 16; ++k2) slab[L.x2w + 16 * k2] = B1[k2];
  wf_sync();
#pragma unroll
  for (int s = 0; s < 4; ++s)
#pragma unroll
    for (int n3h = 0; n3h < 4; ++n3h) Co[s][n3h] = slab[L.x2r[s] + 256 * n3h];
  wf_sync();
  wf_pin(tl, Ce[0][0].x);
  wf2 z[4][8], wk[4];
#pragma unroll
  for (int s = 0; s < 4; ++s) {
    wk[s] = wf_from(twK[64 * s + tl]);
    wf_dft8_pruned<KEEP>(Ce[s], Co[s], z[s]);
  }
  wf_unpack_lane<KEEP>(z, wk, L.self != 0, X);
}

// RAW (N2): 1 = u8, 2 = i16 samples conditioned on the fly: `gain` is the (h, row_stride) gain reference
// (same row pitch as the frames: whole-frame jobs), `job_sub[job]` the per-frame offset, mean_rstd[1]
// the scale (mc_raw_movie_stats); no statistics are gathered.
#ifdef MC_K1_STAMP  // in-kernel phase timing of the wave-per-row K1 (s_memtime deltas summed over waves; experiments)
__device__ unsigned long long g_k1_stamps[16];
#define KSTAMP(v) do { __builtin_amdgcn_sched_barrier(0); v = __builtin_amdgcn_s_memtime(); __builtin_amdgcn_s_waitcnt(0xC07F); __builtin_amdgcn_sched_barrier(0); } while (0)
#define KSTAMP_ADD(acc, a, b) (acc) += (b) - (a)
#else
#define KSTAMP(v) do { } while (0)
#define KSTAMP_ADD(acc, a, b) do { } while (0)
#endif
template <int KEEP, bool STATS, int N1LO, int N1HI, bool CLAMP_ALL, int WF_PREFETCH, bool HALF = false, int RAW = 0>
__global__ __launch_bounds__(256, WF_MIN_WG) void xc_rows_fwd_wave(
    const void* __restrict__ src, const int64_t* __restrict__ job_off, int64_t row_stride,
    const float* __restrict__ mask, const float* __restrict__ mean_rstd, cfloat* __restrict__ T1,
    const cfloat* __restrict__ tw_row, XcGeom g, XcBox box, double* __restrict__ stats_acc,
    const int2* __restrict__ chord, int lines16, const float* __restrict__ gain,
    const float* __restrict__ job_sub) {
  extern __shared__ __attribute__((aligned(16))) float4 park0[];  // lines16: [4 waves][nkx]
  __shared__ __attribute__((aligned(16))) cfloat slabs[4][WF_SLAB];
  __shared__ __attribute__((aligned(16))) cfloat tab[WF_TWA + WF_TWB + 256];
  const cfloat* twA = tab;
  const cfloat* twB = tab + WF_TWA;
  const cfloat* twK = tab + WF_TWA + WF_TWB;  // [slot][lane] = w^kbin
  const int t = threadIdx.x & 63;
  const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  wf2* slab = reinterpret_cast<wf2*>(slabs[wv]);
#ifdef MC_K1_STAMP
  unsigned long long K0 = 0, K1 = 0, Ka = 0, Kb = 0, Kc = 0, Kd = 0, Ke = 0, Kf = 0, kt_row = 0, kt_park = 0, kt_b1 = 0, kt_st = 0, kt_b2 = 0;
#endif
  KSTAMP(K0);
  // Workgroup -> (job, row group): workgroups are dealt round-robin over the 8 XCDs (speed
  // only, MI355X guide), so every job of one row group is sent to the same XCD: its L2 then
  // fetches the group's mask rows once for all jobs instead of once per XCD.
  const int njobs = gridDim.y;
  const int b = blockIdx.x + gridDim.x * blockIdx.y;  // gridDim.x = 8 * ceil(groups / 8)... see host
  const int grp = 8 * (b / (8 * njobs)) + (b & 7);
  const int job = (b >> 3) % njobs;
  if (grp * WF_ROWS_PER_WG >= g.ny) return;  // padding of the last eight groups (uniform)
  {  // tables from tw_row[k] = exp(-2 pi i k / 4096): W_2048^m = tw_row[2 m], W_128^m =
     // tw_row[32 m]; all loads issued before the first LDS write
    constexpr int NTAB = WF_TWA + WF_TWB + 256, PER = (NTAB + 255) / 256;
    cfloat tv[PER];
#pragma unroll
    for (int j = 0; j < PER; ++j) {
      int i = threadIdx.x + 256 * j;
      i = i < NTAB ? i : NTAB - 1;
      int src_k;
      if (i < WF_TWA) {
        src_k = (2 * (i & 127)) << (i >> 7);  // W_2048^{q 2^s} = tw_row[2 q 2^s]
      } else if (i < WF_TWA + WF_TWB) {
        const int e = i - WF_TWA, h = e & 1, k2 = ((e >> 1) % 15) + 1, gq = e / 30;
        src_k = 32 * (2 * gq + h) * k2;
      } else {
        const int e = i - WF_TWA - WF_TWB, sl = e >> 6, l = e & 63;
        src_k = sl == 0 ? l : (sl == 1 ? (l == 0 ? 128 : 256 - l) : (sl == 2 ? 64 + l : 192 - l));
      }
      tv[j] = tw_row[src_k];
    }
#pragma unroll
    for (int j = 0; j < PER; ++j) {
      const int i = threadIdx.x + 256 * j;
      if (i < NTAB) tab[i] = tv[j];
    }
  }
  const float mean = RAW ? job_sub[job] : (mean_rstd ? mean_rstd[0] : 0.f);
  const float rstd = mean_rstd ? mean_rstd[1] : 1.f;
  // frames in their storage type: fp32, or (HALF) fp16 / (RAW) u8, i16 read as they are; job_off / row_stride in samples
  constexpr int SB = RAW == 1 ? 1 : (HALF || RAW == 2) ? 2 : 4;
  const char* base = static_cast<const char*>(src) + job_off[job] * SB;
  auto row_at = [&](int y) -> const void* { return base + (int64_t)y * row_stride * SB; };
  float st_s = 0.f, st_q = 0.f;
  cfloat* out = T1 + (int64_t)job * g.nkx * g.ny;
  // A workgroup takes 16 rows in two rounds of 8 consecutive rows; in a round wave wv
  // transforms rows 2 wv and 2 wv + 1, parks their bins in its own slab as [kx][2 rows] and
  // the workgroup then writes T1[job][kx][8 rows] as whole 64-byte pieces (every byte of T1
  // written once; scattered 16-byte stores cost 2.8x the bytes at the memory side).
  const int r16 = grp * WF_ROWS_PER_WG;
  auto row_of = [&](int i) { return r16 + (i >> 1) * 8 + 2 * wv + (i & 1); };  // i = 0..3
  const int rounds_left = (g.ny - r16) >> 3;  // ny % 8 == 0
  const int nrows = 2 * (rounds_left < WF_ROWS_PER_WG / 8 ? rounds_left : WF_ROWS_PER_WG / 8);
  float4 px[16], mk[16];
  // clamp bounds of a row's sample loads: the support box, or (CLAMP_ALL with a table) the row's
  // own chord of the mask disk -- the corners of the box, 21 % of it, are then never fetched
  const int bxlo = g.x0 & ~3, bxhi = ((g.x1 + 3) & ~3) - 4;
  auto bounds = [&](int y) { return (CLAMP_ALL && chord) ? chord[y] : make_int2(bxlo, bxhi); };
  if (WF_PREFETCH >= 1 && nrows > 0) {
    const int2 c0 = bounds(g.y0 + row_of(0));
    wf_load_px<N1LO, N1HI, CLAMP_ALL, HALF, RAW>(row_at(g.y0 + row_of(0)), t, c0.x, c0.y, px);
  }
  if (WF_PREFETCH >= 2 && RAW == 0 && nrows > 0)
    wf_load_mask<N1LO, N1HI>(mask + (int64_t)(g.y0 + row_of(0)) * g.W, t, mk);
  __syncthreads();
  KSTAMP(K1);
  wf2 Xe[4][KEEP];  // bins of the even row of the current pair
#pragma unroll 1
  for (int rr = 0; rr < nrows; ++rr) {
    const int y = g.y0 + row_of(rr);
    const void* row = row_at(y);
    const float* mrow = mask + (int64_t)y * g.W;
    const int yn = g.y0 + row_of(rr + 1);
    const void* next_row = rr + 1 < nrows ? row_at(yn) : nullptr;
    const float* next_mrow = mask + (int64_t)yn * g.W;
    const bool in_box_row = STATS && y >= box.hl && y < box.hu;
    wf2 X[4][KEEP];
    const int2 cb = bounds(y), cn = bounds(rr + 1 < nrows ? yn : y);
    KSTAMP(Ka);
    wf_row<KEEP, STATS, N1LO, N1HI, CLAMP_ALL, WF_PREFETCH, HALF, RAW>(
        px, mk, row, mrow, next_row, next_mrow, t, slab, twA, twB, twK, g, box.wl >> 8,
        in_box_row ? (box.wu >> 8) : 0, mean, rstd, st_s, st_q, X, cb.x, cb.y, cn.x, cn.y,
        RAW ? gain + (int64_t)y * row_stride : nullptr);
    KSTAMP(Kb);
    KSTAMP_ADD(kt_row, Ka, Kb);
    if (rr & 1) {
      int ts = t;
      wf_pin(ts, X[0][0].x);  // addresses: computed here, not carried across rows
      const WfLane L = wf_lane(ts);
      // lines16: the first round's bins wait in their own LDS area (park0, dynamic) until the second
      // round is done, and T1[job][kx][16 rows] goes out as whole 128-byte lines (two 64-byte halves
      // written 10 us apart merged in L2 only most of the time: 0.55 GB written for a 0.40 GB T1)
      const bool hold = lines16 && nrows == 4 && rr == 1;
      float4* park = hold ? park0 + wv * g.nkx : reinterpret_cast<float4*>(slab);  // [kx] = {even row, odd row}
#pragma unroll
      for (int s = 0; s < 4; ++s)
#pragma unroll
        for (int k3 = 0; k3 < KEEP; ++k3) {
          const int k = L.kbin[s] + 256 * k3;
          if (k < g.nkx) park[k] = make_float4(Xe[s][k3].x, Xe[s][k3].y, X[s][k3].x, X[s][k3].y);
        }
      KSTAMP(Kc);
      KSTAMP_ADD(kt_park, Kb, Kc);
      if (!hold) {  // workgroup-uniform
        __syncthreads();
        KSTAMP(Kd);
        KSTAMP_ADD(kt_b1, Kc, Kd);
        int tj = threadIdx.x;
        wf_pin(tj, X[0][0].y);
        const float4* parked = reinterpret_cast<const float4*>(&slabs[0][0]);
        if (lines16 && nrows == 4) {
          for (int j = tj; j < 8 * g.nkx; j += 256) {  // 8 lanes = the 128 bytes of one kx
            const int kx = j >> 3, pc = j & 7, w = pc & 3;
            const float4 v = (pc >> 2) ? parked[w * (WF_SLAB / 2) + kx] : park0[w * g.nkx + kx];
            *reinterpret_cast<float4*>(out + (int64_t)kx * g.ny + r16 + 2 * pc) = v;
          }
        } else {
          const int r8 = r16 + (rr >> 1) * 8;
          // 4 lanes = the 64 bytes of one kx; nkx <= 256 KEEP, so at most 4 KEEP pieces per thread: all the
          // LDS reads first, then the stores (one LDS latency per round instead of one per piece)
          typedef float f4 __attribute__((ext_vector_type(4)));
#pragma unroll
          for (int half = 0; half < KEEP; ++half) {  // four pieces (16 registers) at a time
            float4 pv[4];
#pragma unroll
            for (int it = 0; it < 4; ++it) {
              const int j = tj + 256 * (4 * half + it);
              if (j < 4 * g.nkx) pv[it] = parked[(j & 3) * (WF_SLAB / 2) + (j >> 2)];
            }
#pragma unroll
            for (int it = 0; it < 4; ++it) {
              const int j = tj + 256 * (4 * half + it);
              if (j < 4 * g.nkx) {
                const int kx = j >> 2, w = j & 3;
#ifndef MC_K1_PLAIN_STORES
                // T1 is written once here and read once by K2, 0.4 GB later: non-temporal (K1 0.465 -> 0.455 ms)
                const f4 v = {pv[it].x, pv[it].y, pv[it].z, pv[it].w};
                __builtin_nontemporal_store(v, reinterpret_cast<f4*>(out + (int64_t)kx * g.ny + r8 + 2 * w));
#else
                *reinterpret_cast<float4*>(out + (int64_t)kx * g.ny + r8 + 2 * w) = pv[it];
#endif
              }
            }
          }
        }
        KSTAMP(Ke);
        KSTAMP_ADD(kt_st, Kd, Ke);
        __syncthreads();
        KSTAMP(Kf);
        KSTAMP_ADD(kt_b2, Ke, Kf);
      }
    } else {
#pragma unroll
      for (int s = 0; s < 4; ++s)
#pragma unroll
        for (int k3 = 0; k3 < KEEP; ++k3) Xe[s][k3] = X[s][k3];
    }
  }
#ifdef MC_K1_STAMP
  KSTAMP(Kf);
  if (t == 0) {
    atomicAdd(&g_k1_stamps[0], K1 - K0); atomicAdd(&g_k1_stamps[1], kt_row); atomicAdd(&g_k1_stamps[2], kt_park);
    atomicAdd(&g_k1_stamps[3], kt_b1); atomicAdd(&g_k1_stamps[4], kt_st); atomicAdd(&g_k1_stamps[5], kt_b2);
    atomicAdd(&g_k1_stamps[6], Kf - K0); atomicAdd(&g_k1_stamps[7], 1ull); atomicAdd(&g_k1_stamps[8], (unsigned long long)nrows);
  }
#endif
  if constexpr (STATS) {
    double ds = st_s, dq = st_q;
    for (int o = 32; o > 0; o >>= 1) {
      ds += __shfl_down(ds, o);
      dq += __shfl_down(dq, o);
    }
    __shared__ double rs[4], rq[4];
    if (t == 0) {
      rs[wv] = ds;
      rq[wv] = dq;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
      ds = (rs[0] + rs[1]) + (rs[2] + rs[3]);
      dq = (rq[0] + rq[1]) + (rq[2] + rq[3]);
      if (ds != 0.0 || dq != 0.0) {
        const int slot = (blockIdx.x + 7 * blockIdx.y) & (XC_STAT_SLOTS - 1);
        atomicAdd(&stats_acc[2 * slot], ds);
        atomicAdd(&stats_acc[2 * slot + 1], dq);
      }
    }
  }
}

// ------------------------------------------------------------------ K1, wave per 1024-sample row
// Patch rows (W = 1024, N = 512 complex points, nkx <= 128): the 8 x 8 x 8 variant of the
// wave engine (mc_wave_fft.h, second half) -- eight complex values per lane, one radix-8
// butterfly per lane and pass, a 4 KiB slab per wave.  DUAL: the same samples are transformed
// twice, with mask^ea and mask^eb (the U and V spectra of the mean-except-current reference,
// estimate_motion_xc.py:315-346), so the patch rows are read once instead of twice.
// Exponents must be >= 1 (the mask's exact zeros outside its support do the windowing).
#define WF5_TWA (7 * 64)
#define WF5_TWB (8 * 7)
#define WF5_TWK 128
#ifndef WF5_MIN_WAVES
#define WF5_MIN_WAVES 5  // waves per SIMD the register allocation aims at (DUAL fp32 sits at 97 VGPRs without it: 4)
#endif
#define WF5_ROWS_PER_WG 32  // rounds of 8 rows: wave wv takes rows 2 wv and 2 wv + 1 of a round

__device__ __forceinline__ wf2 wf5_ld2(const float* p) {  // 4-byte aligned 8-byte load
  wf2 v;
  __builtin_memcpy(&v, p, 8);
  return v;
}
// two adjacent samples of fp16 storage (a 2-byte aligned 4-byte load), widened to fp32: the
// reference has no fp16 path at all (rfftn rejects Half on the CPU); the result is what
// it computes on the fp32 up-cast of the same stack
typedef _Float16 wf_h2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ wf2 wf5_ld2h(const _Float16* p) {
  wf_h2 v;
  __builtin_memcpy(&v, p, 4);
  return wf2{(float)v.x, (float)v.y};
}

__device__ __forceinline__ void wf5_fft(wf2 (&A)[8], int t, wf2* slab, const wf2* twA, const wf2* twB,
                                        const wf2* twK, wf2 (&X)[2]) {
  const int lo = t & 7, hi = t >> 3;
  wf_dft8(A);
#pragma unroll
  for (int k1 = 1; k1 < 8; ++k1) A[k1] = wf_cmul(A[k1], twA[(k1 - 1) * 64 + t]);
#pragma unroll
  for (int k1 = 0; k1 < 8; ++k1) slab[wf5_x1(k1, hi, lo)] = A[k1];  // source: n2 = t >> 3, n3 = t & 7
  wf_sync();
  wf2 B[8];
#pragma unroll
  for (int n2 = 0; n2 < 8; ++n2) B[n2] = slab[wf5_x1(lo, n2, hi)];  // dest: k1 = t & 7, n3 = t >> 3
  wf_sync();
  wf_dft8(B);
#pragma unroll
  for (int k2 = 1; k2 < 8; ++k2) B[k2] = wf_cmul(B[k2], twB[hi * 7 + k2 - 1]);
#pragma unroll
  for (int k2 = 0; k2 < 8; ++k2) slab[wf5_x2(k2, hi, lo)] = B[k2];
  wf_sync();
  wf2 e[4], o[4], z[8];
#pragma unroll
  for (int n3 = 0; n3 < 8; ++n3) {  // dest: c = t = k1 + 8 k2
    const wf2 v = slab[wf5_x2(hi, n3, lo)];
    if (n3 & 1) o[n3 >> 1] = v; else e[n3 >> 1] = v;
  }
  wf_sync();
  wf_dft8_pruned<2>(e, o, z);
  // bin 512 - k lives in lane (64 - t) & 63 at 7 - k3 (lane 0: itself at (8 - k3) & 7)
  const int p = (64 - t) & 63;
  const wf2 zp7 = wf2{__shfl(z[7].x, p), __shfl(z[7].y, p)};
  const wf2 zp6 = wf2{__shfl(z[6].x, p), __shfl(z[6].y, p)};
  const wf2 m0 = t == 0 ? z[0] : zp7, m1 = t == 0 ? zp7 : zp6;
  X[0] = wf_unpack(z[0], m0, twK[t]);
  X[1] = wf_unpack(z[1], m1, twK[t + 64]);
}

template <bool DUAL, bool HALF>
__global__ __launch_bounds__(256, WF5_MIN_WAVES) void xc_rows_fwd_wave512(
    const void* __restrict__ src_any, const int64_t* __restrict__ job_off, int64_t row_stride,
    const int* __restrict__ expo_a, const int* __restrict__ expo_b, const float* __restrict__ mask,
    const float* __restrict__ mean_rstd, cfloat* __restrict__ T1a, cfloat* __restrict__ T1b,
    const cfloat* __restrict__ tw_row, XcGeom g, const int2* __restrict__ chord) {
  __shared__ __attribute__((aligned(16))) wf2 slabs[4][WF5_SLAB];
  __shared__ __attribute__((aligned(16))) wf2 tab[WF5_TWA + WF5_TWB + WF5_TWK];
  const wf2* twA = tab;
  const wf2* twB = tab + WF5_TWA;
  const wf2* twK = tab + WF5_TWA + WF5_TWB;
  const int t = threadIdx.x & 63;
  const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  wf2* slab = slabs[wv];
  const int job = blockIdx.x, grp = blockIdx.y;
  {  // tables from tw_row[k] = exp(-2 pi i k / 1024): W_512^m = tw_row[2 m], W_64^m = tw_row[16 m]
    constexpr int NTAB = WF5_TWA + WF5_TWB + WF5_TWK, PER = (NTAB + 255) / 256;
    cfloat tv[PER];
#pragma unroll
    for (int j = 0; j < PER; ++j) {
      int i = threadIdx.x + 256 * j;
      i = i < NTAB ? i : NTAB - 1;
      int src_k;
      if (i < WF5_TWA) src_k = 2 * (i & 63) * ((i >> 6) + 1);
      else if (i < WF5_TWA + WF5_TWB) src_k = 16 * ((i - WF5_TWA) / 7) * ((i - WF5_TWA) % 7 + 1);
      else src_k = i - WF5_TWA - WF5_TWB;
      tv[j] = tw_row[src_k];
    }
#pragma unroll
    for (int j = 0; j < PER; ++j) {
      const int i = threadIdx.x + 256 * j;
      if (i < NTAB) tab[i] = wf_from(tv[j]);
    }
  }
  const float mean = mean_rstd ? mean_rstd[0] : 0.f;
  const float rstd = mean_rstd ? mean_rstd[1] : 1.f;
  const int ea = expo_a[job], eb = DUAL ? expo_b[job] : 1;
  const float* base = HALF ? nullptr : static_cast<const float*>(src_any) + job_off[job];
  const _Float16* base_h = HALF ? static_cast<const _Float16*>(src_any) + job_off[job] : nullptr;
  cfloat* outa = T1a + (int64_t)job * g.nkx * g.ny;
  cfloat* outb = DUAL ? T1b + (int64_t)job * g.nkx * g.ny : nullptr;
  const int r16 = grp * WF5_ROWS_PER_WG;
  const int rounds = min(WF5_ROWS_PER_WG / 8, (g.ny - r16) / 8);  // ny % 8 == 0
  const int nrows = rounds > 0 ? 2 * rounds : 0;
  const int bxlo = g.x0, bxhi = g.x1 - 2;  // both even: a pair of samples is in or out as a whole
  __syncthreads();
  wf2 Xae[2], Xbe[2];  // bins of the even row of the current pair
#pragma unroll 1
  for (int rr = 0; rr < nrows; ++rr) {
    const int r = r16 + (rr >> 1) * 8 + 2 * wv + (rr & 1);
    const int y = g.y0 + r;
    // with a chord table: only this row's chord of the mask disk is fetched (21 % fewer samples)
    const int xlo = chord ? chord[y].x : bxlo, xhi = chord ? chord[y].y + 2 : bxhi;
    int tl = t;
    asm volatile("" : "+v"(tl));  // per-row addresses are re-derived, not carried (registers)
    const float* row = HALF ? nullptr : base + (int64_t)y * row_stride;
    const _Float16* row_h = HALF ? base_h + (int64_t)y * row_stride : nullptr;
    const float* mrow = mask + (int64_t)y * g.W;
    wf2 A[8], Bv[8], mk[8];
#pragma unroll
    for (int n1 = 0; n1 < 8; ++n1) {  // all sixteen loads in flight before anything is used
      const int x = 128 * n1 + 2 * tl;
      const int xc = min(max(x, xlo), xhi);  // outside the support: mask == 0
      A[n1] = HALF ? wf5_ld2h(row_h + xc) : wf5_ld2(row + xc);
      mk[n1] = *reinterpret_cast<const wf2*>(mrow + x);
    }
#pragma unroll
    for (int n1 = 0; n1 < 8; ++n1) A[n1] = (A[n1] - mean) * rstd;
    if (DUAL && ea == 1 && eb == 2) {  // the leave-one-out schedule's only pair: mask and mask^2, no power loop
#pragma unroll
      for (int n1 = 0; n1 < 8; ++n1) {
        A[n1] = A[n1] * mk[n1];
        Bv[n1] = A[n1] * mk[n1];
      }
    } else {  // mask^ea and mask^eb: wave-uniform trip counts, kept out of the load loop
      wf2 pw[8];
#pragma unroll
      for (int n1 = 0; n1 < 8; ++n1) pw[n1] = mk[n1];
      for (int e = 1; e < (DUAL ? eb : ea); ++e) {
#pragma unroll
        for (int n1 = 0; n1 < 8; ++n1) pw[n1] *= mk[n1];
      }
      if (DUAL) {
#pragma unroll
        for (int n1 = 0; n1 < 8; ++n1) Bv[n1] = A[n1] * pw[n1];
#pragma unroll
        for (int n1 = 0; n1 < 8; ++n1) pw[n1] = mk[n1];
        for (int e = 1; e < ea; ++e) {
#pragma unroll
          for (int n1 = 0; n1 < 8; ++n1) pw[n1] *= mk[n1];
        }
      }
#pragma unroll
      for (int n1 = 0; n1 < 8; ++n1) A[n1] = A[n1] * pw[n1];
    }
    wf2 Xa[2], Xb[2];
    wf5_fft(A, tl, slab, twA, twB, twK, Xa);
    if (DUAL) wf5_fft(Bv, tl, slab, twA, twB, twK, Xb);
    if (rr & 1) {
      float4* park = reinterpret_cast<float4*>(slab);  // [spectrum][128 kx] = {even row, odd row}
      park[tl] = make_float4(Xae[0].x, Xae[0].y, Xa[0].x, Xa[0].y);
      park[tl + 64] = make_float4(Xae[1].x, Xae[1].y, Xa[1].x, Xa[1].y);
      if (DUAL) {
        park[128 + tl] = make_float4(Xbe[0].x, Xbe[0].y, Xb[0].x, Xb[0].y);
        park[128 + tl + 64] = make_float4(Xbe[1].x, Xbe[1].y, Xb[1].x, Xb[1].y);
      }
      __syncthreads();
      {
        const int r8 = r16 + (rr >> 1) * 8;
        const float4* parked = reinterpret_cast<const float4*>(&slabs[0][0]);
        const int per = 4 * g.nkx;  // 4 lanes = the 64 bytes (8 rows) of one kx
        for (int j = threadIdx.x; j < (DUAL ? 2 : 1) * per; j += 256) {
          const int sp = j >= per, jj = j - sp * per;
          const int kx = jj >> 2, w = jj & 3;
          cfloat* out = sp ? outb : outa;
          *reinterpret_cast<float4*>(out + (int64_t)kx * g.ny + r8 + 2 * w) =
              parked[w * (WF5_SLAB / 2) + sp * 128 + kx];
        }
      }
      __syncthreads();
    } else {
      Xae[0] = Xa[0]; Xae[1] = Xa[1];
      if (DUAL) { Xbe[0] = Xb[0]; Xbe[1] = Xb[1]; }
    }
  }
}

// Provisional mean of the fused-statistics path: m0 = {mean of n samples, 1, 1} by ONE
// workgroup (any value near the true mean keeps the linear fix-up free of cancellation).
template <typename T>
__global__ __launch_bounds__(256) void xc_provisional_mean_kernel(const T* __restrict__ x, int n,
                                                                  float* __restrict__ m0) {
  double s = 0.0;
  for (int i = threadIdx.x; i < n; i += 256) s += (double)(float)x[i];
  for (int off = 32; off > 0; off >>= 1) s += __shfl_down(s, off);
  __shared__ double part[4];
  if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) {
    m0[0] = (float)(((part[0] + part[1]) + (part[2] + part[3])) / (double)n);
    m0[1] = 1.f;
    m0[2] = 1.f;
  }
}

// (sum, sumsq) of (x - m0) over `count` samples, spread over XC_STAT_SLOTS accumulators
// -> fix = {mean - m0, 1/std}, out3 = {mean, 1/std, std}  (unbiased std, as
// torch.std_mean, utils.py:81)
__global__ void xc_stats_finalize(const double* __restrict__ acc_slots, double count,
                                  const float* __restrict__ m0, float* __restrict__ fix,
                                  float* __restrict__ out3) {
  double acc[2] = {0.0, 0.0};
  for (int s = 0; s < XC_STAT_SLOTS; ++s) {
    acc[0] += acc_slots[2 * s];
    acc[1] += acc_slots[2 * s + 1];
  }
  const double dm = acc[0] / count;
  double var = (acc[1] - acc[0] * acc[0] / count) / (count - 1.0);
  if (var < 0) var = 0;
  const float stdf = (float)sqrt(var);
  fix[0] = (float)dm;
  fix[1] = 1.0f / stdf;
  out3[0] = (float)((double)m0[0] + dm);
  out3[1] = 1.0f / stdf;
  out3[2] = stdf;
}

// ------------------------------------------------------------------ K2: columns forward
// fix (optional): {dmean, rstd} and Mhat = pruned spectrum of the mask: the spectrum of
// ((x - m0) - dmean) * rstd * mask is (Y - dmean * Mhat) * rstd by linearity.
#ifndef XC_FWD_COLS
#define XC_FWD_COLS 2  // columns per workgroup in the radix-16 K2, the second one fetched under the first (1: 107, 2: 101, 4: 105 us)
#endif
// R16 (H = 4096, kyp and kyn <= 512): the register-resident radix-16 transform of mc_fft.h
// with the unwanted output rows pruned at compile time.
template <int LOGH, bool R16 = false>
__global__ __launch_bounds__(MC_WG) void xc_cols_fwd(const cfloat* __restrict__ T1,
                                                     const float* __restrict__ filt,
                                                     cfloat* __restrict__ S,
                                                     const cfloat* __restrict__ tw_col, XcGeom g,
                                                     const float* __restrict__ fix,
                                                     const cfloat* __restrict__ Mhat) {
  constexpr int H = 1 << LOGH;
  __shared__ __attribute__((aligned(16))) cfloat line[R16 ? H : lds_len(H)];  // radix 16: unpadded, 5 workgroups / CU
  const int tid = threadIdx.x;
  const int kx = blockIdx.x, job = blockIdx.y;
  const cfloat* col = T1 + ((int64_t)job * g.nkx + kx) * g.ny;
  const int nky = g.kyp + g.kyn;
  cfloat* out = S + ((int64_t)job * g.nkx + kx) * nky;
  const float* f = filt ? filt + (int64_t)kx * nky : nullptr;
  const cfloat* mh = fix ? Mhat + (int64_t)kx * nky : nullptr;
  const float dmean = fix ? fix[0] : 0.f, rstd = fix ? fix[1] : 1.f;
  auto load = [&](int y) {
    const int yy = y - g.y0;
    return (yy >= 0 && yy < g.ny) ? col[yy] : cmake(0.f, 0.f);
  };
  auto store = [&](int ky, cfloat v) {
    int kyi = -1;
    if (ky < g.kyp) kyi = ky;
    else if (ky >= H - g.kyn) kyi = ky - (H - g.kyn) + g.kyp;
    if (kyi >= 0) {
      if (fix) {
        const cfloat m = mh[kyi];
        v = cmake((v.x - dmean * m.x) * rstd, (v.y - dmean * m.y) * rstd);
      }
      out[kyi] = f ? cscale(v, f[kyi]) : v;
    }
  };
  if constexpr (R16) {
    // XC_FWD_COLS consecutive kx columns per workgroup (blockIdx.x counts column groups): the next
    // column's samples are in flight (registers) while the current one is transformed
    const int kx0 = blockIdx.x * XC_FWD_COLS;
    auto fetch = [&](int kxc, cfloat (&v)[16], int tq) {
      const cfloat* c = T1 + ((int64_t)job * g.nkx + kxc) * g.ny;
#pragma unroll
      for (int n1 = 0; n1 < 16; ++n1) {
        const int yy = 256 * n1 + tq - g.y0;
        v[n1] = (yy >= 0 && yy < g.ny) ? c[yy] : cmake(0.f, 0.f);
      }
    };
    cfloat curv[16], nxtv[16];
    fetch(kx0 < g.nkx ? kx0 : g.nkx - 1, curv, tid);
    // twiddle bases and (below) the filter / mask-spectrum values of the 4 rows this thread stores: all issued with
    // the column's samples, instead of waiting for them in the middle and at the end of the transform
    const R16Tw TW = r16_twiddles(tid, tw_col);
    int kyo[4];
#pragma unroll
    for (int s4 = 0; s4 < 4; ++s4) kyo[s4] = kept_index(tid + 256 * (s4 < 2 ? s4 : s4 + 12), H, g.kyp, g.kyn);
#pragma unroll 1
    for (int cc = 0; cc < XC_FWD_COLS; ++cc) {
      const int kxc = kx0 + cc;
      if (kxc >= g.nkx) break;  // workgroup-uniform
      int tcol = tid;  // opaque per column: nothing derived from it is hoisted (registers)
      asm volatile("" : "+v"(tcol));
      if (cc + 1 < XC_FWD_COLS && kxc + 1 < g.nkx) fetch(kxc + 1, nxtv, tcol);
      cfloat* outc = S + ((int64_t)job * g.nkx + kxc) * nky;
      const float* fc = filt ? filt + (int64_t)kxc * nky : nullptr;
      const cfloat* mhc = fix ? Mhat + (int64_t)kxc * nky : nullptr;
      auto loadr = [&](int n1, int) { return curv[n1]; };
      float fpre[4];
      cfloat mpre[4];
#pragma unroll
      for (int s4 = 0; s4 < 4; ++s4) {
        fpre[s4] = (fc && kyo[s4] >= 0) ? fc[kyo[s4]] : 1.f;
        mpre[s4] = (fix && kyo[s4] >= 0) ? mhc[kyo[s4]] : cmake(0.f, 0.f);
      }
      auto storer = [&](int k3, int, cfloat v) {  // k3 in {0, 1, 14, 15}, a compile-time constant at every call
        const int s4 = k3 < 2 ? k3 : k3 - 12;
        if (kyo[s4] >= 0) {
          if (fix) v = cmake((v.x - dmean * mpre[s4].x) * rstd, (v.y - dmean * mpre[s4].y) * rstd);
          outc[kyo[s4]] = fc ? cscale(v, fpre[s4]) : v;
        }
      };
      wg_fft4096_r16_tw<-1, 8, 2>(line, tcol, TW, loadr, storer);
      __syncthreads();
#pragma unroll
      for (int n1 = 0; n1 < 16; ++n1) curv[n1] = nxtv[n1];
    }
  } else {
    wg_fft<H, -1>(line, tid, tw_col, 1, load, store);
  }
}

// ------------------------------------------------------------------ K2 / K3, wave per 1024-point column
// Columns of 1024 x 1024 patches (H = 1024, at most 128 kept rows at either end of the
// spectrum): one wavefront per column, four columns per workgroup, the 16 x 8 x 8 transform of
// mc_wave_fft.h (third part) in registers, no workgroup barrier.  The workgroup-per-column
// kernels spend a 1024-point column on 256 threads (4 values each) and 7 barriers.
// K2: only the kept output rows are produced (k3 in {0, 7} of the last radix-8 pass).
// K3: only the kept input rows are fetched (n1 in {0, 1, 14, 15} of the first radix-16 pass);
//     the inverse runs the forward kernel on conjugated data.
#ifndef XC_NEAR_MIN_WAVES
#define XC_NEAR_MIN_WAVES 4  // register target of the near-window column passes (5: 96 VGPRs, spills)
#endif
#define XC_FWDW_COLS 1  // columns per wavefront in xc_cols_fwd_wave1024 (4 measured slower: the kernel streams T1 at 3.2 TB/s)
// The six table entries a lane needs (they depend on the lane only): loaded once, up front -- behind the
// acquire fence of wf_sync the compiler cannot start them early, and a wave waited for the L2 in the middle of
// every column.
struct Wf10Tw {
  wf2 w1, w2, w4, w8, b[2];
};
__device__ __forceinline__ Wf10Tw wf10_twiddles(int t, const cfloat* __restrict__ tw) {
  Wf10Tw T;
  T.w1 = wf_from(tw[t]); T.w2 = wf_from(tw[2 * t]); T.w4 = wf_from(tw[4 * t]); T.w8 = wf_from(tw[8 * t]);
  T.b[0] = wf_from(tw[16 * (t >> 4)]);
  T.b[1] = wf_from(tw[16 * ((t >> 4) + 4)]);
  return T;
}
__device__ __forceinline__ void wf10_passes_ab(wf2 (&a)[16], int t, wf2* slab, const Wf10Tw& T,
                                               wf2 (&B)[2][8]) {
  wf_twiddle16(a, T.w1, T.w2, T.w4, T.w8);
#pragma unroll
  for (int k1 = 0; k1 < 16; ++k1) slab[wf10_x1(k1, t)] = a[k1];
  wf_sync();
  const int k1 = t & 15;
#pragma unroll
  for (int b = 0; b < 2; ++b) {
    const int n3 = (t >> 4) + 4 * b;
#pragma unroll
    for (int n2 = 0; n2 < 8; ++n2) B[b][n2] = slab[wf10_x1(k1, 8 * n2 + n3)];
  }
  wf_sync();
#pragma unroll
  for (int b = 0; b < 2; ++b) {
    wf_dft8(B[b]);
    wf_twiddle8(B[b], T.b[b]);  // W_64^{n3 k2}
#pragma unroll
    for (int k2 = 0; k2 < 8; ++k2) slab[wf10_x2(k1, k2, (t >> 4) + 4 * b)] = B[b][k2];
  }
  wf_sync();
}

__global__ __launch_bounds__(256) void xc_cols_fwd_wave1024(const cfloat* __restrict__ T1,
                                                            const float* __restrict__ filt,
                                                            cfloat* __restrict__ S,
                                                            const cfloat* __restrict__ tw_col, XcGeom g,
                                                            const float* __restrict__ fix,
                                                            const cfloat* __restrict__ Mhat) {
  constexpr int H = 1024;
  __shared__ __attribute__((aligned(16))) wf2 slabs[4][WF10_N];
  const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int kx0 = (blockIdx.x * 4 + wv) * XC_FWDW_COLS, job = blockIdx.y;
  if (kx0 >= g.nkx) return;  // no workgroup barrier below
  wf2* slab = slabs[wv];
  const int nky = g.kyp + g.kyn;
  const float dmean = fix ? fix[0] : 0.f, rstd = fix ? fix[1] : 1.f;
  const Wf10Tw TW = wf10_twiddles(threadIdx.x & 63, tw_col);
#pragma unroll 1
  for (int cc = 0; cc < XC_FWDW_COLS; ++cc) {
  const int kx = kx0 + cc;
  if (kx >= g.nkx) break;  // wave-uniform
  int tq = threadIdx.x & 63;  // opaque per column: nothing derived from it is hoisted (registers)
  asm volatile("" : "+v"(tq));
  const int t = tq;
  const cfloat* col = T1 + ((int64_t)job * g.nkx + kx) * g.ny;
  cfloat* out = S + ((int64_t)job * g.nkx + kx) * nky;
  const float* f = filt ? filt + (int64_t)kx * nky : nullptr;
  const cfloat* mh = fix ? Mhat + (int64_t)kx * nky : nullptr;
  wf2 a[16], B[2][8];
#pragma unroll
  for (int n1 = 0; n1 < 16; ++n1) {
    const int yy = 64 * n1 + t - g.y0;
    a[n1] = (yy >= 0 && yy < g.ny) ? wf_from(col[yy]) : wf2{0.f, 0.f};
  }
  // filter and mask-spectrum values of the (at most four) rows this lane stores: fetched with the samples
  int kyo[2][2];
  float fpre[2][2];
  cfloat mpre[2][2];
#pragma unroll
  for (int b = 0; b < 2; ++b)
#pragma unroll
    for (int sel = 0; sel < 2; ++sel) {
      const int ky = (t & 15) + 16 * ((t >> 4) + 4 * b) + (sel ? 896 : 0);
      int kyi = -1;
      if (ky < g.kyp) kyi = ky;
      else if (ky >= H - g.kyn) kyi = ky - (H - g.kyn) + g.kyp;
      kyo[b][sel] = kyi;
      fpre[b][sel] = (f && kyi >= 0) ? f[kyi] : 1.f;
      mpre[b][sel] = (fix && kyi >= 0) ? mh[kyi] : cmake(0.f, 0.f);
    }
  wf_dft16(a);
  wf10_passes_ab(a, t, slab, TW, B);
#pragma unroll
  for (int b = 0; b < 2; ++b) {
    const int k1 = t & 15, k2 = (t >> 4) + 4 * b;
    wf2 e[4], o[4], z[8];
#pragma unroll
    for (int n3 = 0; n3 < 8; ++n3) {
      const wf2 v = slab[wf10_x2(k1, k2, n3)];
      if (n3 & 1) o[n3 >> 1] = v; else e[n3 >> 1] = v;
    }
    wf_sync();
    wf_dft8_pruned<1>(e, o, z);  // k3 = 0 and 7
#pragma unroll
    for (int sel = 0; sel < 2; ++sel) {
      const int kyi = kyo[b][sel];
      if (kyi >= 0) {
        cfloat v = wf_to(z[sel ? 7 : 0]);
        if (fix) v = cmake((v.x - dmean * mpre[b][sel].x) * rstd, (v.y - dmean * mpre[b][sel].y) * rstd);
        out[kyi] = f ? cscale(v, fpre[b][sel]) : v;
      }
    }
  }
  }
}

__global__ __launch_bounds__(256) void xc_cols_inv_wave1024(
    const cfloat* __restrict__ S_cur, const int* __restrict__ cur_idx,
    const cfloat* __restrict__ S_ref, const int* __restrict__ ref_idx, cfloat* __restrict__ T2,
    const cfloat* __restrict__ tw_col, float scale, XcGeom g) {
  constexpr int H = 1024;
  __shared__ __attribute__((aligned(16))) wf2 slabs[4][WF10_N];
  const int t = threadIdx.x & 63;
  const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int kx = blockIdx.x * 4 + wv, p = blockIdx.y;
  if (kx >= g.nkx) return;  // no workgroup barrier below
  wf2* slab = slabs[wv];
  const int nky = g.kyp + g.kyn;
  const cfloat* cur = S_cur + ((int64_t)cur_idx[p] * g.nkx + kx) * nky;
  const cfloat* ref = S_ref + ((int64_t)ref_idx[p] * g.nkx + kx) * nky;
  cfloat* out = T2 + ((int64_t)p * g.nkx + kx) * H;
  wf2 a[16], B[2][8];
#pragma unroll
  for (int n1 = 0; n1 < 16; ++n1) {
    a[n1] = wf2{0.f, 0.f};
    if (n1 < 2 || n1 >= 14) {  // the only input rows a band-limited spectrum can hold
      const int kyi = kept_index(64 * n1 + t, H, g.kyp, g.kyn);
      if (kyi >= 0) {
        const cfloat v = cscale(cmulc(ref[kyi], cur[kyi]), scale);
        a[n1] = wf2{v.x, -v.y};  // conjugate in, conjugate out: inverse transform
      }
    }
  }
  const Wf10Tw TW = wf10_twiddles(t, tw_col);
  wf_dft16(a);
  wf10_passes_ab(a, t, slab, TW, B);
#pragma unroll
  for (int b = 0; b < 2; ++b) {
    const int k1 = t & 15, k2 = (t >> 4) + 4 * b;
    wf2 c[8];
#pragma unroll
    for (int n3 = 0; n3 < 8; ++n3) c[n3] = slab[wf10_x2(k1, k2, n3)];
    wf_dft8(c);
#pragma unroll
    for (int k3 = 0; k3 < 8; ++k3) out[k1 + 16 * k2 + 128 * k3] = cmake(c[k3].x, -c[k3].y);
  }
}

// Near-window form of the same (see xc_cols_inv_near below): a wavefront runs XC_NEAR_COLS
// columns of one pair, keeps the stored window's rows and its share of the row bounds (16 rows
// per lane, in registers over the column loop).
#define XC_NEAR_COLS_W 8
__global__ __launch_bounds__(256, XC_NEAR_MIN_WAVES) void xc_cols_inv_near_wave1024(
    const cfloat* __restrict__ S_cur, const int* __restrict__ cur_idx,
    const cfloat* __restrict__ S_ref, const int* __restrict__ ref_idx, cfloat* __restrict__ T2n,
    float* __restrict__ bounds, const cfloat* __restrict__ tw_col, float scale, XcGeom g, int nstore) {
  constexpr int H = 1024;
  __shared__ __attribute__((aligned(16))) wf2 slabs[4][WF10_N];
  const int t = threadIdx.x & 63;
  const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int p = blockIdx.y;
  const int kx0 = (blockIdx.x * 4 + wv) * XC_NEAR_COLS_W;
  if (kx0 >= g.nkx) return;  // no workgroup barrier below
  wf2* slab = slabs[wv];
  const int nky = g.kyp + g.kyn;
  float acc[16];
#pragma unroll
  for (int i = 0; i < 16; ++i) acc[i] = 0.f;
  // twiddles once per wave; a column's eight input values are fetched while the previous column is
  // transformed (as xc_cols_inv_near does: the kernel waited, exposed, at the head of every column)
  const Wf10Tw TW = wf10_twiddles(t, tw_col);
  const int64_t cur_base = (int64_t)cur_idx[p] * g.nkx, ref_base = (int64_t)ref_idx[p] * g.nkx;
  int kyi4[4];
#pragma unroll
  for (int s4 = 0; s4 < 4; ++s4) kyi4[s4] = kept_index(64 * (s4 < 2 ? s4 : s4 + 12) + t, H, g.kyp, g.kyn);
  cfloat pc[4], pr[4];
  auto fetch4 = [&](int kx, cfloat (&c4)[4], cfloat (&r4)[4]) {
    const cfloat* cur = S_cur + (cur_base + kx) * nky;
    const cfloat* ref = S_ref + (ref_base + kx) * nky;
#pragma unroll
    for (int s4 = 0; s4 < 4; ++s4) {
      c4[s4] = kyi4[s4] >= 0 ? cur[kyi4[s4]] : cmake(0.f, 0.f);
      r4[s4] = kyi4[s4] >= 0 ? ref[kyi4[s4]] : cmake(0.f, 0.f);
    }
  };
  fetch4(kx0, pc, pr);
#pragma unroll 1
  for (int cc = 0; cc < XC_NEAR_COLS_W; ++cc) {
    const int kx = kx0 + cc;
    if (kx >= g.nkx) break;  // wave-uniform
    int tl = t;  // opaque per column: nothing derived from it is hoisted (registers)
    asm volatile("" : "+v"(tl));
    cfloat* outn = T2n + ((int64_t)p * g.nkx + kx) * (2 * nstore);
    const float wgt = kx == 0 ? 1.f : 2.f;
    wf2 a[16], B[2][8];
#pragma unroll
    for (int n1 = 0; n1 < 16; ++n1) {
      a[n1] = wf2{0.f, 0.f};
      if (n1 < 2 || n1 >= 14) {
        const int s4 = n1 < 2 ? n1 : n1 - 12;
        const cfloat v = cscale(cmulc(pr[s4], pc[s4]), scale);  // zero where the row is not kept
        a[n1] = wf2{v.x, -v.y};
      }
    }
    if (cc + 1 < XC_NEAR_COLS_W && kx + 1 < g.nkx) fetch4(kx + 1, pc, pr);
    wf_dft16_lo2(a);  // entries 2..13 are zero
    wf10_passes_ab(a, tl, slab, TW, B);
#pragma unroll
    for (int b = 0; b < 2; ++b) {
      const int k1 = tl & 15, k2 = (tl >> 4) + 4 * b;
      wf2 c[8];
#pragma unroll
      for (int n3 = 0; n3 < 8; ++n3) c[n3] = slab[wf10_x2(k1, k2, n3)];
      wf_sync();
      wf_dft8(c);
#pragma unroll
      for (int k3 = 0; k3 < 8; ++k3) {
        const int y = k1 + 16 * k2 + 128 * k3;
        const int yn = y < nstore ? y : y - (H - 2 * nstore);
        if (yn >= 0 && yn < 2 * nstore && (y < nstore || y >= H - nstore)) outn[yn] = cmake(c[k3].x, -c[k3].y);
        acc[8 * b + k3] += wgt * __builtin_amdgcn_sqrtf(c[k3].x * c[k3].x + c[k3].y * c[k3].y);
      }
    }
  }
#pragma unroll
  for (int b = 0; b < 2; ++b)
#pragma unroll
    for (int k3 = 0; k3 < 8; ++k3)
      atomicAdd(&bounds[(int64_t)p * H + (t & 15) + 16 * ((t >> 4) + 4 * b) + 128 * k3], acc[8 * b + k3]);
}

// ------------------------------------------------------------------ K3: columns inverse
// pair p: cur spectrum index cur_idx[p] in S_cur, ref spectrum index ref_idx[p] in S_ref.
// MODE 0: conj(ref)*cur (cross-correlation); MODE 1: cur * phase ramp (Fourier shift,
// correct_motion.py:488-494) -- phase computed in K3 from shifts[p] = (sy, sx).
template <int LOGH, int MODE>
__global__ __launch_bounds__(MC_WG) void xc_cols_inv(
    const cfloat* __restrict__ S_cur, const int* __restrict__ cur_idx,
    const cfloat* __restrict__ S_ref, const int* __restrict__ ref_idx,
    const float* __restrict__ shifts, cfloat* __restrict__ T2, const cfloat* __restrict__ tw_col,
    float scale, XcGeom g, const int* __restrict__ gate) {
  constexpr int H = 1 << LOGH;
  __shared__ __attribute__((aligned(16))) cfloat line[lds_len(H)];
  if (gate && gate[0] == 0) return;  // the near window settled every pair: nothing to do
  const int tid = threadIdx.x;
  const int kx = blockIdx.x, p = blockIdx.y;
  const int nky = g.kyp + g.kyn;
  const cfloat* cur = S_cur + ((int64_t)cur_idx[p] * g.nkx + kx) * nky;
  const cfloat* ref = MODE == 0 ? S_ref + ((int64_t)ref_idx[p] * g.nkx + kx) * nky : nullptr;
  cfloat* out = T2 + ((int64_t)p * g.nkx + kx) * H;
  float sy = 0.f, sx = 0.f, fx = 0.f;
  if (MODE == 1) {
    sy = shifts[2 * p];
    sx = shifts[2 * p + 1];
    fx = (float)kx / (float)g.W;  // rfftfreq
  }
  auto load = [&](int ky) {
    int kyi = -1;
    if (ky < g.kyp) kyi = ky;
    else if (ky >= H - g.kyn) kyi = ky - (H - g.kyn) + g.kyp;
    if (kyi < 0) return cmake(0.f, 0.f);
    cfloat v;
    if (MODE == 0) {
      v = cmulc(ref[kyi], cur[kyi]);
    } else {
      // torch.fft.fftfreq: k/H for k < (H+1)/2 else (k-H)/H; angle = sum(-2*pi*f*s)
      const int kk = (ky < (H + 1) / 2) ? ky : ky - H;
      const float fy = (float)kk / (float)H;
      const float m2pi = -6.283185307179586f;
      const float ang = (m2pi * fy) * sy + (m2pi * fx) * sx;
      float sn, cs;
      sincosf(ang, &sn, &cs);
      v = cmul(cur[kyi], cmake(cs, sn));
    }
    return cscale(v, scale);
  };
  auto store = [&](int y, cfloat v) { out[y] = v; };
  wg_fft<H, +1>(line, tid, tw_col, 1, load, store);
}

// K3 for the arg-max search without the full T2: a workgroup runs XC_NEAR_COLS columns of
// one pair through the inverse column FFT, keeps only the rows of the near window
// (rows [0, nstore) and [H - nstore, H), nstore = near rows + guard rows for the sub-pixel
// neighbourhood of a peak on the window's edge) in T2n[p][kx][2 nstore] and adds
// its share of the triangle-inequality row bounds (see K4) to bounds[p][y] -- per thread in
// registers over its columns, then one float atomic per (thread, row).  The full map is
// only ever materialised (xc_cols_inv, gated by `need_full`) when some far row's bound
// reaches the maximum found in the near window.
#ifndef XC_NEAR_COLS
#define XC_NEAR_COLS 8
#endif
#define XC_NEAR_GUARD 8  // extra stored rows per end: neighbourhood of a peak on the window's edge
template <int LOGH, bool R16 = false>
__global__ __launch_bounds__(MC_WG, XC_NEAR_MIN_WAVES) void xc_cols_inv_near(
    const cfloat* __restrict__ S_cur, const int* __restrict__ cur_idx,
    const cfloat* __restrict__ S_ref, const int* __restrict__ ref_idx, cfloat* __restrict__ T2n,
    float* __restrict__ bounds, const cfloat* __restrict__ tw_col, float scale, XcGeom g, int nstore) {
  constexpr int H = 1 << LOGH;
  __shared__ __attribute__((aligned(16))) cfloat line[R16 ? H : lds_len(H)];
  constexpr int NOUT = H / MC_WG;  // rows per thread in the last pass (H >= 1024: all threads busy)
  const int tid = threadIdx.x;
  const int p = blockIdx.y;
  const int nky = g.kyp + g.kyn;
  // this thread's share of the row bounds: the last pass of every column hands a thread the
  // same NOUT rows in the same order, so the sums stay in registers over the column loop
  float acc[NOUT];
#pragma unroll
  for (int c = 0; c < NOUT; ++c) acc[c] = 0.f;
  const int64_t cur_base = (int64_t)cur_idx[p] * g.nkx, ref_base = (int64_t)ref_idx[p] * g.nkx;
  // R16: a thread's pass-A inputs are rows tid + 256 n1, n1 in {0, 1, 14, 15} (the band-pass keeps |ky| < 512);
  // the next column's eight values are fetched while this column is transformed (its loads used to sit, exposed,
  // at the head of every column: the kernel is neither VALU- nor HBM-bound, profiles/r03_k3n_pmc.txt)
  int kyi4[4];
  cfloat pc[4], pr[4];
  auto fetch4 = [&](int kx, cfloat (&c4)[4], cfloat (&r4)[4]) {
    const cfloat* cur = S_cur + (cur_base + kx) * nky;
    const cfloat* ref = S_ref + (ref_base + kx) * nky;
#pragma unroll
    for (int s4 = 0; s4 < 4; ++s4) {
      c4[s4] = kyi4[s4] >= 0 ? cur[kyi4[s4]] : cmake(0.f, 0.f);
      r4[s4] = kyi4[s4] >= 0 ? ref[kyi4[s4]] : cmake(0.f, 0.f);
    }
  };
  R16Tw TW;  // twiddle bases: once per workgroup, not eight global loads inside every column
  if constexpr (R16) {
    TW = r16_twiddles(tid, tw_col);
#pragma unroll
    for (int s4 = 0; s4 < 4; ++s4) kyi4[s4] = kept_index(tid + 256 * (s4 < 2 ? s4 : s4 + 12), H, g.kyp, g.kyn);
    if ((int)blockIdx.x * XC_NEAR_COLS < g.nkx) fetch4(blockIdx.x * XC_NEAR_COLS, pc, pr);
  }
#pragma unroll 1
  for (int cc = 0; cc < XC_NEAR_COLS; ++cc) {
    const int kx = blockIdx.x * XC_NEAR_COLS + cc;
    if (kx >= g.nkx) break;  // workgroup-uniform
    const cfloat* cur = S_cur + (cur_base + kx) * nky;
    const cfloat* ref = S_ref + (ref_base + kx) * nky;
    cfloat* outn = T2n + ((int64_t)p * g.nkx + kx) * (2 * nstore);
    const float wgt = kx == 0 ? 1.f : 2.f;
    auto load = [&](int ky) {
      const int kyi = kept_index(ky, H, g.kyp, g.kyn);
      if (kyi < 0) return cmake(0.f, 0.f);
      return cscale(cmulc(ref[kyi], cur[kyi]), scale);
    };
    cfloat qc[4], qr[4];  // this column's values; pc / pr receive the next column's
    if constexpr (R16) {
#pragma unroll
      for (int s4 = 0; s4 < 4; ++s4) {
        qc[s4] = pc[s4];
        qr[s4] = pr[s4];
      }
      if (cc + 1 < XC_NEAR_COLS && kx + 1 < g.nkx) fetch4(kx + 1, pc, pr);
    }
    auto load16 = [&](int n1, int) { return cscale(cmulc(qr[n1 < 2 ? n1 : n1 - 12], qc[n1 < 2 ? n1 : n1 - 12]), scale); };
    int c = 0;
    auto store = [&](int y, cfloat v) {
      const int yn = y < nstore ? y : y - (H - 2 * nstore);  // position in the stored window
      if (yn >= 0 && yn < 2 * nstore && (y < nstore || y >= H - nstore)) outn[yn] = v;
      // hardware square root (1 ulp): the bound test carries a 1e-4 relative slack.  (The cheaper upper bound
      // max + (sqrt 2 - 1) min, up to 8 % above |z|, does not make the kernel faster and opens the fall-back on
      // noisier movies: scripts/far_margin.py, noise 4: largest far bound 0.95 of the maximum, 1.001 with it.)
      acc[c++] += wgt * __builtin_amdgcn_sqrtf(v.x * v.x + v.y * v.y);
    };
    // opaque per column: everything derived from the thread index (kept-row indices, near
    // positions, LDS addresses of every pass) is loop-invariant and would otherwise be
    // hoisted out of the column loop into ~90 registers (one workgroup less per CU)
    int tcol = tid;
    asm volatile("" : "+v"(tcol));
    // (what the opaque copy hides and the stored-window tests need: with nstore <= 256, checked by the host, only the
    // first and the last of a thread's 16 rows tid + 256 k3 can lie in the window -- 14 tests fold away)
    __builtin_assume(tcol >= 0 && tcol < MC_WG);
    __builtin_assume(nstore > 0 && nstore <= 256);
    if constexpr (R16) wg_fft4096_r16_tw<+1, 2, 8>(line, tcol, TW, load16, store);
    else wg_fft<H, +1>(line, tcol, tw_col, 1, load, store);
    __syncthreads();  // the next column's first pass overwrites the line
  }
  // rows of the last pass (fft_pass with NS * R == H): y = tid + it * MC_WG + m * (H / R), in
  // the order it-major, m-minor
  if constexpr (R16) {  // wg_fft4096_r16 stores y = tid + 256 k3 in the order k3 = 0..15
#pragma unroll
    for (int k3 = 0; k3 < 16; ++k3) atomicAdd(&bounds[(int64_t)p * H + tid + 256 * k3], acc[k3]);
  } else {
    constexpr int R = (H >= 4096) ? 8 : (H == 2048 ? 4 : 2);  // last radix of FftPlan<H>: 8 8 8 {8,4,2}
    constexpr int NB = H / R, IT = NB / MC_WG;
    static_assert(IT * R == NOUT, "row ownership of the last pass");
#pragma unroll
    for (int it = 0; it < IT; ++it)
#pragma unroll
      for (int m = 0; m < R; ++m)
        atomicAdd(&bounds[(int64_t)p * H + tid + it * MC_WG + m * NB], acc[it * R + m]);
  }
}

// best[p] = order(-inf), gate = 0, bounds = 0 in one launch
__global__ void xc_search_init(int* __restrict__ best, int* __restrict__ gate, float* __restrict__ bounds,
                               int npairs, int nbounds, float* __restrict__ shift_table, int nshift) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < nbounds) bounds[i] = 0.f;
  if (i < nshift) shift_table[i] = 0.f;  // rows no pair writes (the reference frame) stay zero
  if (i < npairs) best[i] = (int)0x807fffffu;  // float_order(-INFINITY) = 0xff800000 ^ 0x7fffffff
  if (i == 0) gate[0] = 0;
}

// After the near-window phase: a far row group must be evaluated iff its bound can reach
// the maximum attained so far.  Initialises the far groups' candidates and raises
// need_full[0] when any such group exists.
__global__ void xc_far_needed(const float* __restrict__ bounds, const int* __restrict__ best,
                              float* __restrict__ part_val, int* __restrict__ part_idx,
                              int* __restrict__ need_full, int H, int RG, int near) {
  const int ngrp = H / RG;
  const int p = blockIdx.y;
  const int grp = near + blockIdx.x * blockDim.x + threadIdx.x;
  if (grp >= ngrp - near) return;
  float b = 0.f;
  for (int r = 0; r < RG; ++r) b = fmaxf(b, bounds[(int64_t)p * H + grp * RG + r]);
  b = b * 1.0001f + 1e-30f;
  part_val[(int64_t)p * ngrp + grp] = -INFINITY;
  part_idx[(int64_t)p * ngrp + grp] = 0x7fffffff;
  const int fb = __float_as_int(b);
  if ((fb >= 0 ? fb : fb ^ 0x7fffffff) >= best[p]) atomicOr(need_full, 1);
}

// ------------------------------------------------------------------ K4: rows inverse

// EPI 0: arg-max over the whole map (partials per workgroup); EPI 1: store real rows.
// Branch and bound for the arg-max (EPI 0): every value of row y obeys
//   |cc(y,x)| <= |X[0]| + 2 * sum_{k>=1} |X[k]|          (X = T2[.][y], triangle inequality)
// so a row group whose bound is below a value some other workgroup has already
// *attained* cannot hold the maximum (nor tie with it) and is skipped; `best[p]` carries
// that running maximum (monotone atomic max, stale reads only cost skipped work).
// Groups are visited nearest-to-zero-shift first, where the peak usually is.

// bounds[p][y] = |X[0]| + 2 * sum_{k>=1} |X[k]|,  X = T2[p][.][y].  Workgroup = 64 rows x
// 4 interleaved kx slices (coalesced 512-byte reads over y, 4 loads in flight per thread).
__global__ __launch_bounds__(256) void xc_row_bounds(const cfloat* __restrict__ T2,
                                                     float* __restrict__ bounds, int nkx, int H) {
  __shared__ float part[4][64];
  const int ly = threadIdx.x & 63, slice = threadIdx.x >> 6;
  const int y = blockIdx.x * 64 + ly;
  const int p = blockIdx.y;
  float b0 = 0.f, b1 = 0.f, b2 = 0.f, b3 = 0.f;
  if (y < H) {
    const cfloat* in = T2 + (int64_t)p * nkx * H + y;
    int kx = slice;
    for (; kx + 12 < nkx; kx += 16) {
      const cfloat v0 = in[(int64_t)kx * H], v1 = in[(int64_t)(kx + 4) * H];
      const cfloat v2 = in[(int64_t)(kx + 8) * H], v3 = in[(int64_t)(kx + 12) * H];
      b0 += (kx == 0 ? 1.f : 2.f) * sqrtf(v0.x * v0.x + v0.y * v0.y);
      b1 += 2.f * sqrtf(v1.x * v1.x + v1.y * v1.y);
      b2 += 2.f * sqrtf(v2.x * v2.x + v2.y * v2.y);
      b3 += 2.f * sqrtf(v3.x * v3.x + v3.y * v3.y);
    }
    for (; kx < nkx; kx += 4) {
      const cfloat v = in[(int64_t)kx * H];
      b0 += (kx == 0 ? 1.f : 2.f) * sqrtf(v.x * v.x + v.y * v.y);
    }
  }
  part[slice][ly] = (b0 + b1) + (b2 + b3);
  __syncthreads();
  if (slice == 0 && y < H)
    bounds[(int64_t)p * H + y] = (part[0][ly] + part[1][ly]) + (part[2][ly] + part[3][ly]);
}

template <int LOGN, int EPI>
__global__ __launch_bounds__(MC_WG) void xc_rows_inv(const cfloat* __restrict__ T2,
                                                     const float* __restrict__ bounds,
                                                     int* __restrict__ best,
                                                     float* __restrict__ part_val,
                                                     int* __restrict__ part_idx,
                                                     float* __restrict__ out_real,
                                                     const int64_t* __restrict__ out_off,
                                                     int64_t out_stride,
                                                     const cfloat* __restrict__ tw_row, XcGeom g,
                                                     int near, int phase, int compact,
                                                     const int* __restrict__ gate) {
  constexpr int N = 1 << LOGN;
  if (gate && gate[0] == 0) return;  // far phase not needed (xc_far_needed filled the candidates)
  constexpr int NT = fft_threads(N), SG = MC_WG / NT;
  constexpr int R0 = FftPlan<N>::radix(0), NB0 = N / R0, IT0 = (NB0 + NT - 1) / NT;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x;
  const int lt = tid & (NT - 1), sg = tid / NT;
  cfloat* l0 = reinterpret_cast<cfloat*>(smem) + sg * 2 * lds_len(N);
  cfloat* l1 = l0 + lds_len(N);
  cfloat* stg = reinterpret_cast<cfloat*>(smem) + SG * 2 * lds_len(N);
  const int p = blockIdx.y;
  const int RG = g.RG;
  const int ngrp = g.H / RG;
  // phase 0: the `near` groups at each end of the map (small |shift|), evaluated
  // unconditionally -> best[p]; phase 1: all other groups, with the skip test.
  int grp = blockIdx.x;
  if (EPI == 0) grp = phase == 0 ? ((int)blockIdx.x < near ? (int)blockIdx.x : ngrp - 2 * near + (int)blockIdx.x)
                                 : near + (int)blockIdx.x;
  if constexpr (EPI == 0) {
    if (phase == 1) {  // wave-uniform early exit: nothing of T2 is read for a skipped group
      float b = 0.f;
      for (int r = 0; r < RG; ++r) b = fmaxf(b, bounds[(int64_t)p * g.H + grp * RG + r]);
      b = b * 1.0001f + 1e-30f;  // rounding slack of the transform itself
      if (float_order(b) < __hip_atomic_load(&best[p], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) {
        if (tid == 0) {
          part_val[(int64_t)p * ngrp + grp] = -INFINITY;
          part_idx[(int64_t)p * ngrp + grp] = 0x7fffffff;
        }
        return;
      }
    }
  }
  // compact > 0: T2 holds only the stored window, [p][kx][2 nstore], nstore = near RG + guard,
  // guard = compact - 1 rows (xc_cols_inv_near); the negative-shift groups start at nstore + guard
  const int nstore = near * RG + (compact > 0 ? compact - 1 : 0);
  const int cstride = compact ? 2 * nstore : g.H;
  const int coff = (int)blockIdx.x < near ? (int)blockIdx.x * RG
                                          : nstore + (compact - 1) + ((int)blockIdx.x - near) * RG;
  const cfloat* in = T2 + (int64_t)p * g.nkx * cstride + (int64_t)(compact ? coff : grp * RG);
  // eight loads in flight per thread (the one-at-a-time loop waited for the L2 up to 13 times in a row)
  for (int i0 = tid; i0 < g.nkx * RG; i0 += 8 * MC_WG) {
    cfloat v8[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const int i = i0 + u * MC_WG;
      const int kx = i / RG, r = i - kx * RG;
      if (i < g.nkx * RG) v8[u] = in[(int64_t)kx * cstride + r];
    }
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const int i = i0 + u * MC_WG;
      const int kx = i / RG, r = i - kx * RG;
      if (i < g.nkx * RG) stg[kx * (RG + 1) + r] = v8[u];
    }
  }
  FftTwiddles<N> T;
  T.template init<+1>(lt, tw_row, 2);
  // c2r pack twiddles conj(w^k) for this thread's first-pass elements
  cfloat wk[IT0][R0];
#pragma unroll
  for (int it = 0; it < IT0; ++it)
#pragma unroll
    for (int q = 0; q < R0; ++q) {
      const int j = lt + it * NT;
      const int k = (NB0 % NT == 0 || j < NB0) ? j + q * NB0 : 0;
      cfloat w = tw_row[k];
      w.y = -w.y;
      wk[it][q] = w;
    }
  __syncthreads();
  float bv = -INFINITY;
  int bi = 0x7fffffff;
  int s = 0;
  for (int r = sg; r < RG; r += SG) {  // RG % SG == 0
    const int y = grp * RG + r;
    // c2r pack: Z[k] = (X[k] + conj(X[N-k])) + i * conj(w^k) * (X[k] - conj(X[N-k]))
    auto load = [&](int k, int it, int q) {
      const int km = N - k;  // in [1, N]
      cfloat xk = (k < g.nkx) ? stg[k * (RG + 1) + r] : cmake(0.f, 0.f);
      cfloat xm = (km < g.nkx) ? cconj(stg[km * (RG + 1) + r]) : cmake(0.f, 0.f);
      if (k == 0) {  // c2r ignores the imaginary part of the DC and Nyquist bins (pocketfft)
        xk.y = 0.f;
        xm.y = 0.f;
      }
      const cfloat sm = cadd(xk, xm), d = csub(xk, xm);
      const cfloat wd = cmul(wk[it][q], d);  // i*wd = (-wd.y, wd.x)
      return cmake(sm.x - wd.y, sm.y + wd.x);
    };
    int res;
    if constexpr (EPI == 0) {
      auto store = [&](int n, cfloat v) {
        const int flat = y * g.W + 2 * n;
        cand_merge(bv, bi, v.x, flat);
        cand_merge(bv, bi, v.y, flat + 1);
      };
      res = wg_fft_pp<N, +1, true>(l0, l1, s, lt, T, load, store);
    } else {
      float* orow = out_real + out_off[p] + (int64_t)y * out_stride;
      auto store = [&](int n, cfloat v) {
        orow[2 * n] = v.x;
        orow[2 * n + 1] = v.y;
      };
      res = wg_fft_pp<N, +1, true>(l0, l1, s, lt, T, load, store);
    }
    s = res ^ 1;  // the last pass still reads line[res] while the next row starts
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

// ------------------------------------------------------------------ K5: final peak
// peaks[p] = flat index of the first maximum; shifts[p] = wrapped (y, x) as float
// (xc.py:116-121: p if p <= n//2 else p - n).
__global__ void xc_peak_final(const float* __restrict__ part_val, const int* __restrict__ part_idx,
                              int ngrp, int H, int W, int* __restrict__ peaks,
                              float* __restrict__ shifts, const int* __restrict__ shift_rows) {
  const int p = blockIdx.x;
  float bv = -INFINITY;
  int bi = 0x7fffffff;
  for (int i = threadIdx.x; i < ngrp; i += blockDim.x)
    cand_merge(bv, bi, part_val[(int64_t)p * ngrp + i], part_idx[(int64_t)p * ngrp + i]);
  for (int off = 32; off > 0; off >>= 1) {
    const float ov = __shfl_down(bv, off);
    const int oi = __shfl_down(bi, off);
    cand_merge(bv, bi, ov, oi);
  }
  if (threadIdx.x == 0) {
    if (bi == 0x7fffffff) bi = 0;
    peaks[p] = bi;
    const int py = bi / W, px = bi - py * W;
    const int row = shift_rows ? shift_rows[p] : p;  // scatter into a per-frame table when asked
    shifts[2 * row] = (float)(py <= H / 2 ? py : py - H);
    shifts[2 * row + 1] = (float)(px <= W / 2 ? px : px - W);
  }
}

// ------------------------------------------------------------------ K6: neighbourhood
// nb[p][dy][dx] (3x3 floats) = correlation values around peaks[p], produced by the
// same inverse-row arithmetic as K4 so the values are the ones the arg-max saw.
// Entries outside the map are NaN.
// T2n / gate / nstore (optional): while gate[0] == 0 the map rows live in the compact
// near-window buffer T2n[p][kx][2 nstore] of xc_cols_inv_near, otherwise in the full T2.
template <int LOGN>
__global__ __launch_bounds__(MC_WG) void xc_peak_nbhd(const cfloat* __restrict__ T2,
                                                      const int* __restrict__ peaks,
                                                      float* __restrict__ nb,
                                                      const cfloat* __restrict__ tw_row, XcGeom g,
                                                      const cfloat* __restrict__ T2n,
                                                      const int* __restrict__ gate, int nstore) {
  constexpr int N = 1 << LOGN;
  __shared__ __attribute__((aligned(16))) cfloat line[lds_len(N)];
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
  int64_t cs = g.H;
  if (T2n && gate[0] == 0) {  // workgroup-uniform
    const int yn = y < nstore ? y : y - (g.H - 2 * nstore);
    if (yn < 0 || yn >= 2 * nstore || (y >= nstore && y < g.H - nstore)) {  // not stored (cannot
      if (tid < 3) o[tid] = __builtin_nanf("");  // happen for a peak inside the near window)
      return;
    }
    cs = 2 * nstore;
    in = T2n + (int64_t)p * g.nkx * cs + yn;
  }
  auto X = [&](int k) { return in[(int64_t)k * cs]; };
  auto load = [&](int k) {
    const int km = N - k;
    cfloat xk = (k < g.nkx) ? X(k) : cmake(0.f, 0.f);
    cfloat xm = (km < g.nkx) ? cconj(X(km)) : cmake(0.f, 0.f);
    if (k == 0) {
      xk.y = 0.f;
      xm.y = 0.f;
    }
    const cfloat s = cadd(xk, xm), d = csub(xk, xm);
    cfloat w = tw_row[k];
    w.y = -w.y;
    const cfloat wd = cmul(w, d);
    return cmake(s.x - wd.y, s.y + wd.x);
  };
  auto store = [&](int n, cfloat v) { line[lpad(n)] = v; };
  wg_fft<N, +1>(line, tid, tw_row, 2, load, store);
  __syncthreads();
  if (tid < 3) {
    const int x = px + tid - 1;
    float v = __builtin_nanf("");
    if (x >= 0 && x < g.W) {
      const cfloat z = line[lpad(x >> 1)];
      v = (x & 1) ? z.y : z.x;
    }
    o[tid] = v;
  }
}

// ------------------------------------------------------------------ host dispatch
void mc_launch_row_bounds(const cfloat* T2, float* bounds, int nkx, int H, int npairs, hipStream_t stream) {
  hipLaunchKernelGGL(xc_row_bounds, dim3((H + 63) / 64, npairs), dim3(256), 0, stream, T2, bounds, nkx, H);
}
void mc_launch_peak_final(const float* part_val, const int* part_idx, int ngrp, int H, int W, int* peaks,
                          float* shifts, const int* shift_rows, int npairs, hipStream_t stream) {
  hipLaunchKernelGGL(xc_peak_final, dim3(npairs), dim3(64), 0, stream, part_val, part_idx, ngrp, H, W, peaks, shifts,
                     shift_rows);
}
#define MC_DISPATCH_LOG(LOGV, ...)          \
  switch (LOGV) {                           \
    MC_DISPATCH_CASE(4, __VA_ARGS__)        \
    MC_DISPATCH_CASE(5, __VA_ARGS__)        \
    MC_DISPATCH_CASE(6, __VA_ARGS__)        \
    MC_DISPATCH_CASE(7, __VA_ARGS__)        \
    MC_DISPATCH_CASE(8, __VA_ARGS__)        \
    MC_DISPATCH_CASE(9, __VA_ARGS__)        \
    MC_DISPATCH_CASE(10, __VA_ARGS__)       \
    MC_DISPATCH_CASE(11, __VA_ARGS__)       \
    MC_DISPATCH_CASE(12, __VA_ARGS__)       \
    default:                                \
      return MC_ERR_UNSUPPORTED;            \
  }


// mc_xc_row_engine(): 0 = automatic (wave-per-row kernel whenever the shape fits),
// 1 = always the workgroup-per-row kernels (A/B timing and cross-checks of the engines).
static int g_col_engine = 0;  // mc_xc_col_engine(): 0 = automatic, 1 = always the radix-8 Stockham columns
static int g_row_engine = 0;
static bool mc_force_wg_rows() { return g_row_engine == 1; }
// experiments: NAME=1 in the environment switches a default-off feature on
static bool mc_env_on(const char* name) {
  const char* v = getenv(name);
  return v && v[0] == '1';
}

// The wave-per-row kernel reads samples and mask rows with 16-byte loads.  job_off[] lives
// on the device: callers of the C ABI keep it a multiple of 4 floats whenever W == 4096
// (whole frames: f * h * w; documented in mcorr.h).
static bool wave_rows_aligned(const float* src, const float* mask, int64_t row_stride) {
  return ((reinterpret_cast<uintptr_t>(src) | reinterpret_cast<uintptr_t>(mask)) & 15) == 0 &&
         (row_stride & 3) == 0;
}

static size_t rows_lds_bytes(int N, const XcGeom& g) {
  const int sgroups = MC_WG / fft_threads(N);
  return sizeof(cfloat) * ((size_t)sgroups * 2 * lds_len(N) + (size_t)g.nkx * (g.RG + 1));
}

extern "C" {

static void* g_after_k3n_event = nullptr;
int mc_xc_after_k3n_event(void* event) {
  g_after_k3n_event = event;
  return MC_OK;
}

int mc_xc_col_engine(int mode) {
  if (mode < 0 || mode > 1) return MC_ERR_ARG;
  g_col_engine = mode;
  return MC_OK;
}

int mc_xc_row_engine(int mode) {
  if (mode < 0 || mode > 1) return MC_ERR_ARG;
  g_row_engine = mode;
  return MC_OK;
}

int mc_xc_rows_lds_bytes(const mc_xc_geom* q) {
  XcGeom g;
  int rc = geom_from(q, &g, true, false);
  if (rc) return rc;
  return (int)rows_lds_bytes(g.W / 2, g);
}

static int rows_forward_impl(const float* src, const int64_t* job_off, int64_t row_stride,
                             const int* job_expo, const float* mask, const float* mean_rstd,
                             void* T1, const void* tw_row, int njobs, const mc_xc_geom* q,
                             const XcBox* box, double* stats_acc, void* stream,
                             const int* row_chord = nullptr, bool half = false) {
  XcGeom g;
  int rc = geom_from(q, &g, true, false);
  if (rc) return rc;
  if (!src || !job_off || !T1 || !tw_row || njobs < 1) return MC_ERR_ARG;
  // fp16 samples: only the wave-per-row engine reads them (4096-column frames); anything else is
  // MC_ERR_UNSUPPORTED and the caller widens the stack once
  if (half && !(g.W == 2 * WF_N && g.nkx <= 512 && (g.ny % 8) == 0 && mask && !job_expo &&
                wave_rows_aligned(src, mask, row_stride) &&
                (!stats_acc || (((box ? box->wl : 0) | (box ? box->wu : 0)) & 255) == 0)))
    return MC_ERR_UNSUPPORTED;
  const int logn = mc_ilog2(g.W) - 1;
  XcBox b = box ? *box : XcBox{0, 0, 0, 0};
  if (g.W == 2 * WF_N && g.nkx <= 512 && (g.ny % 8) == 0 && mask && !job_expo &&
      wave_rows_aligned(src, mask, row_stride) && (!stats_acc || ((b.wl | b.wu) & 255) == 0) &&
      !mc_force_wg_rows()) {
    // wave-per-row engine (mc_wave_fft.h); misaligned jobs take its element-wise loads
    const int ngroups = (g.ny + WF_ROWS_PER_WG - 1) / WF_ROWS_PER_WG;
    dim3 grid((ngroups + 7) / 8 * 8, njobs);  // linear id = x + gridDim.x * y, decoded in the kernel
    // whole 128-byte lines of T1 (16 rows per kx) when the first round's bins fit next to the slabs
    // with two workgroups per CU still resident, and every 16-row piece is line-aligned.  OFF by
    // default (MC_K1_LINES16=1 enables it): it saves the 0.15 GB of write amplification (one stream:
    // 2.00 -> 1.97 ms per 40 x 4096^2 step) but its 26 KB of extra LDS per workgroup keeps the warp's
    // tiles of the other stream out of the CU: 1.74 -> 1.84 ms per step under the two-stream overlap.
    const size_t park_bytes = (size_t)4 * g.nkx * 16;
    const int lines16 = (WF_ROWS_PER_WG == 16 && mc_env_on("MC_K1_LINES16") && (g.ny % 16) == 0 && park_bytes <= 27 * 1024 &&
                         ((reinterpret_cast<uintptr_t>(T1) & 127) == 0)) ? 1 : 0;
    const size_t dyn = lines16 ? park_bytes : 0;
#define MC_WAVE_LAUNCH(KEEP, ST, LO, HI, CL, PF)                                                  \
  do {                                                                                            \
    auto kw = xc_rows_fwd_wave<KEEP, ST, LO, HI, CL, PF>;                                         \
    if (dyn) (void)hipFuncSetAttribute((const void*)kw, hipFuncAttributeMaxDynamicSharedMemorySize, (int)dyn); \
    hipLaunchKernelGGL(kw, grid, dim3(256), dyn, (hipStream_t)stream, src, job_off, row_stride, mask, \
                       mean_rstd, (cfloat*)T1, (const cfloat*)tw_row, g, b, stats_acc,             \
                       (const int2*)row_chord, lines16, (const float*)nullptr, (const float*)nullptr); \
  } while (0)
#define MC_WAVE_LAUNCH_H(KEEP, ST)                                                                \
  do {                                                                                            \
    auto kw = xc_rows_fwd_wave<KEEP, ST, 0, 16, true, WF_PREFETCH_DEFAULT, true>;                 \
    if (dyn) (void)hipFuncSetAttribute((const void*)kw, hipFuncAttributeMaxDynamicSharedMemorySize, (int)dyn); \
    hipLaunchKernelGGL(kw, grid, dim3(256), dyn, (hipStream_t)stream, (const void*)src, job_off, row_stride, mask, \
                       mean_rstd, (cfloat*)T1, (const cfloat*)tw_row, g, b, stats_acc,             \
                       (const int2*)row_chord, lines16, (const float*)nullptr, (const float*)nullptr); \
  } while (0)
    if (half) {  // fp16 storage: the general variant (all 16 chunks, per-row chord clamp when given)
      if (g.nkx <= 256) {
        if (stats_acc) MC_WAVE_LAUNCH_H(1, true); else MC_WAVE_LAUNCH_H(1, false);
      } else {
        if (stats_acc) MC_WAVE_LAUNCH_H(2, true); else MC_WAVE_LAUNCH_H(2, false);
      }
      return mc_check_launch();
    }
#define MC_WAVE_PICK(KEEP, ST)                                                              \
  do {                                                                                      \
    if (g.x0 >= 256 && g.x0 <= 512 && g.x1 >= 3584 && g.x1 <= 3840 && row_chord)            \
      MC_WAVE_LAUNCH(KEEP, ST, 1, 15, true, WF_PREFETCH_DEFAULT); /* per-row chord clamp */  \
    else if (g.x0 >= 256 && g.x0 <= 512 && g.x1 >= 3584 && g.x1 <= 3840)                    \
      MC_WAVE_LAUNCH(KEEP, ST, 1, 15, false, WF_PREFETCH_DEFAULT); /* square 4096 frames */ \
    else                                                                                    \
      MC_WAVE_LAUNCH(KEEP, ST, 0, 16, true, WF_PREFETCH_DEFAULT);                           \
  } while (0)
#ifdef MC_K1_STAMP
    unsigned long long kz[16] = {0};
    (void)hipMemcpyToSymbol(HIP_SYMBOL(g_k1_stamps), kz, sizeof kz);
#endif
    if (g.nkx <= 256) {
      if (stats_acc) MC_WAVE_PICK(1, true); else MC_WAVE_PICK(1, false);
    } else {
      if (stats_acc) MC_WAVE_PICK(2, true); else MC_WAVE_PICK(2, false);
    }
#ifdef MC_K1_STAMP
    (void)hipStreamSynchronize((hipStream_t)stream);
    (void)hipMemcpyFromSymbol(kz, HIP_SYMBOL(g_k1_stamps), sizeof kz);
    if (kz[7] && njobs > 1) {
      const double wv = (double)kz[7], rows = (double)kz[8];
      fprintf(stderr, "K1 stamps (memtime ticks): per wave: prologue %.0f total %.0f | per row: transform %.0f | per row pair: "
              "park %.0f barrier %.0f store %.0f barrier %.0f | waves %.0f rows/wave %.2f\n",
              kz[0] / wv, kz[6] / wv, kz[1] / rows, 2 * kz[2] / rows, 2 * kz[3] / rows, 2 * kz[4] / rows, 2 * kz[5] / rows, wv, rows / wv);
    }
#endif
#undef MC_WAVE_PICK
#undef MC_WAVE_LAUNCH
    return mc_check_launch();
  }
  const size_t lds = rows_lds_bytes(g.W / 2, g);
  if (lds > 160 * 1024) return MC_ERR_ARG;
  dim3 grid(njobs, g.ny / g.RG);
  MC_DISPATCH_LOG(logn, {
    auto k = stats_acc ? xc_rows_fwd<L, true> : xc_rows_fwd<L, false>;
    if (lds > 64 * 1024)
      (void)hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipLaunchKernelGGL(k, grid, dim3(MC_WG), lds, (hipStream_t)stream, (const void*)src, job_off, row_stride,
                       job_expo, mask, mean_rstd, (cfloat*)T1, (const cfloat*)tw_row, g, b, stats_acc,
                       (const float*)nullptr, (const float*)nullptr);
  });
  return mc_check_launch();
}

int mc_xc_rows_forward_dual(const float* src, const int64_t* job_off, int64_t row_stride,
                            const int* expo_a, const int* expo_b, const float* mask,
                            const float* mean_rstd, void* T1a, void* T1b, const void* tw_row,
                            int njobs, const mc_xc_geom* q, const int* row_chord, void* stream) {
  return mc_xc_rows_forward_dual_t(src, MC_STORE_F32, job_off, row_stride, expo_a, expo_b, mask, mean_rstd,
                                   T1a, T1b, tw_row, njobs, q, row_chord, stream);
}

int mc_xc_rows_forward_dual_t(const void* src, int storage, const int64_t* job_off, int64_t row_stride,
                              const int* expo_a, const int* expo_b, const float* mask,
                              const float* mean_rstd, void* T1a, void* T1b, const void* tw_row,
                              int njobs, const mc_xc_geom* q, const int* row_chord, void* stream) {
  if (storage != MC_STORE_F32 && storage != MC_STORE_F16) return MC_ERR_UNSUPPORTED;
  const bool half = storage == MC_STORE_F16;
  XcGeom g;
  int rc = geom_from(q, &g, true, false);
  if (rc) return rc;
  if (!src || !job_off || !expo_a || !mask || !T1a || !tw_row || njobs < 1 || (expo_b && !T1b))
    return MC_ERR_ARG;
  if (g.W != 2 * WF5_N || g.nkx > 128 || (g.ny % 8) || (reinterpret_cast<uintptr_t>(mask) & 7))
    return MC_ERR_UNSUPPORTED;
  dim3 grid(njobs, (g.ny + WF5_ROWS_PER_WG - 1) / WF5_ROWS_PER_WG);
#define MC_W512_GO(D, H)                                                                              \
  hipLaunchKernelGGL((xc_rows_fwd_wave512<D, H>), grid, dim3(256), 0, (hipStream_t)stream, src, job_off, \
                     row_stride, expo_a, D ? expo_b : (const int*)nullptr, mask, mean_rstd, (cfloat*)T1a, \
                     D ? (cfloat*)T1b : (cfloat*)nullptr, (const cfloat*)tw_row, g, (const int2*)row_chord)
  if (expo_b) {
    if (half) MC_W512_GO(true, true);
    else MC_W512_GO(true, false);
  } else {
    if (half) MC_W512_GO(false, true);
    else MC_W512_GO(false, false);
  }
#undef MC_W512_GO
  return mc_check_launch();
}

int mc_xc_rows_forward(const float* src, const int64_t* job_off, int64_t row_stride,
                       const int* job_expo, const float* mask, const float* mean_rstd,
                       void* T1, const void* tw_row, int njobs, const mc_xc_geom* q,
                       void* stream) {
  return rows_forward_impl(src, job_off, row_stride, job_expo, mask, mean_rstd, T1, tw_row, njobs, q,
                           nullptr, nullptr, stream);
}

int mc_xc_provisional_mean(const float* x, int n, float* m0, void* stream) {
  if (!x || !m0 || n < 1) return MC_ERR_ARG;
  hipLaunchKernelGGL(xc_provisional_mean_kernel<float>, dim3(1), dim3(256), 0, (hipStream_t)stream, x, n, m0);
  return mc_check_launch();
}

int mc_xc_provisional_mean_t(const void* x, int storage, int n, float* m0, void* stream) {
  if (!x || !m0 || n < 1) return MC_ERR_ARG;
  if (storage == MC_STORE_F32) return mc_xc_provisional_mean(static_cast<const float*>(x), n, m0, stream);
  if (storage != MC_STORE_F16) return MC_ERR_UNSUPPORTED;
  hipLaunchKernelGGL(xc_provisional_mean_kernel<_Float16>, dim3(1), dim3(256), 0, (hipStream_t)stream,
                     static_cast<const _Float16*>(x), n, m0);
  return mc_check_launch();
}

int mc_xc_rows_forward_stats(const float* src, const int64_t* job_off, int64_t row_stride,
                             const float* mask, const float* m0, void* T1, const void* tw_row,
                             int njobs, const mc_xc_geom* q, int hl, int hu, int wl, int wu,
                             double* acc, float* fix, float* out3, const int* row_chord, void* stream) {
  return mc_xc_rows_forward_stats_t(src, MC_STORE_F32, job_off, row_stride, mask, m0, T1, tw_row, njobs, q, hl, hu,
                                    wl, wu, acc, fix, out3, row_chord, stream);
}

int mc_xc_rows_forward_stats_t(const void* src_any, int storage, const int64_t* job_off, int64_t row_stride,
                               const float* mask, const float* m0, void* T1, const void* tw_row,
                               int njobs, const mc_xc_geom* q, int hl, int hu, int wl, int wu,
                               double* acc, float* fix, float* out3, const int* row_chord, void* stream) {
  if (storage != MC_STORE_F32 && storage != MC_STORE_F16) return MC_ERR_UNSUPPORTED;
  const float* src = static_cast<const float*>(src_any);
  const bool half = storage == MC_STORE_F16;
  if (!m0 || !acc || !fix || !out3 || !q) return MC_ERR_ARG;
  if (hl < q->y0 || hu > q->y0 + q->ny || wl < q->x0 || wu > q->x1 || (wl & 1) || (wu & 1) ||
      hl >= hu || wl >= wu)
    return MC_ERR_ARG;  // the box must lie inside the region K1 reads
  hipError_t e = hipMemsetAsync(acc, 0, 2 * XC_STAT_SLOTS * sizeof(double), (hipStream_t)stream);
  if (e != hipSuccess) return (int)e;
  XcBox box{hl, hu, wl, wu};
  int rc = rows_forward_impl(src, job_off, row_stride, nullptr, mask, m0, T1, tw_row, njobs, q, &box,
                             acc, stream, row_chord, half);
  if (rc) return rc;
  const double count = (double)njobs * (hu - hl) * (wu - wl);
  hipLaunchKernelGGL(xc_stats_finalize, dim3(1), dim3(1), 0, (hipStream_t)stream, acc, count, m0, fix,
                     out3);
  return mc_check_launch();
}

// N2: K1 straight from the raw bytes of a u8 / i16 movie: A = (raw * gain - sub[job]) * mean_rstd[1] * mask.
// Whole-frame jobs: 4096-column frames on the wave-per-row engine, any other power-of-two width on the
// workgroup engine (mc_xcg_rows_forward_raw has the K3 formats); anything else is MC_ERR_UNSUPPORTED and the
// caller conditions the movie into an fp32 copy first (mc_condition_movie).
int mc_xc_rows_forward_raw(const void* raw, int storage, const float* gain, const int64_t* job_off,
                           int64_t row_stride, const float* mask, const float* job_sub, const float* mean_rstd,
                           void* T1, const void* tw_row, int njobs, const mc_xc_geom* q, const int* row_chord,
                           void* stream) {
  if (storage != MC_STORE_U8 && storage != MC_STORE_I16) return MC_ERR_UNSUPPORTED;
  XcGeom g;
  int rc = geom_from(q, &g, true, false);
  if (rc) return rc;
  if (!raw || !gain || !job_off || !mask || !job_sub || !mean_rstd || !T1 || !tw_row || njobs < 1) return MC_ERR_ARG;
  const uintptr_t al = reinterpret_cast<uintptr_t>(raw) | reinterpret_cast<uintptr_t>(gain) |
                       reinterpret_cast<uintptr_t>(mask);
  if (!(g.W == 2 * WF_N && g.nkx <= 512 && (g.ny % 8) == 0 && (al & 15) == 0 && (row_stride & 7) == 0)) {
    // any other power-of-two width: the workgroup-per-row engine, element-wise loads of raw and gain
    const size_t lds = rows_lds_bytes(g.W / 2, g);
    if (lds > 160 * 1024) return MC_ERR_ARG;
    const int logn = mc_ilog2(g.W) - 1;
    dim3 gridw(njobs, g.ny / g.RG);
    XcBox bb{0, 0, 0, 0};
    MC_DISPATCH_LOG(logn, {
      if (storage == MC_STORE_U8) {
        auto k = xc_rows_fwd<L, false, 1>;
        if (lds > 64 * 1024) (void)hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        hipLaunchKernelGGL(k, gridw, dim3(MC_WG), lds, (hipStream_t)stream, raw, job_off, row_stride, (const int*)nullptr,
                           mask, mean_rstd, (cfloat*)T1, (const cfloat*)tw_row, g, bb, (double*)nullptr, gain, job_sub);
      } else {
        auto k = xc_rows_fwd<L, false, 2>;
        if (lds > 64 * 1024) (void)hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        hipLaunchKernelGGL(k, gridw, dim3(MC_WG), lds, (hipStream_t)stream, raw, job_off, row_stride, (const int*)nullptr,
                           mask, mean_rstd, (cfloat*)T1, (const cfloat*)tw_row, g, bb, (double*)nullptr, gain, job_sub);
      }
    });
    return mc_check_launch();
  }
  const int ngroups = (g.ny + WF_ROWS_PER_WG - 1) / WF_ROWS_PER_WG;
  dim3 grid((ngroups + 7) / 8 * 8, njobs);
  XcBox b{0, 0, 0, 0};
#define MC_WAVE_RAW(KEEP, R)                                                                                   \
  hipLaunchKernelGGL((xc_rows_fwd_wave<KEEP, false, 0, 16, true, WF_PREFETCH_DEFAULT, false, R>), grid, dim3(256), 0, \
                     (hipStream_t)stream, raw, job_off, row_stride, mask, mean_rstd, (cfloat*)T1,               \
                     (const cfloat*)tw_row, g, b, (double*)nullptr, (const int2*)row_chord, 0, gain, job_sub)
  if (storage == MC_STORE_U8) {
    if (g.nkx <= 256) MC_WAVE_RAW(1, 1); else MC_WAVE_RAW(2, 1);
  } else {
    if (g.nkx <= 256) MC_WAVE_RAW(1, 2); else MC_WAVE_RAW(2, 2);
  }
#undef MC_WAVE_RAW
  return mc_check_launch();
}

int mc_xc_cols_forward(const void* T1, const float* filt, void* S, const void* tw_col, int njobs,
                       const mc_xc_geom* q, void* stream) {
  return mc_xc_cols_forward_fix(T1, filt, S, tw_col, njobs, q, nullptr, nullptr, stream);
}

int mc_xc_cols_forward_fix(const void* T1, const float* filt, void* S, const void* tw_col,
                           int njobs, const mc_xc_geom* q, const float* fix, const void* Mhat,
                           void* stream) {
  XcGeom g;
  int rc = geom_from(q, &g, false, true);
  if (rc) return rc;
  if (!T1 || !S || !tw_col || njobs < 1 || (fix && !Mhat)) return MC_ERR_ARG;
  dim3 grid(g.nkx, njobs);
  if (g.H == 1024 && g.kyp <= 128 && g.kyn <= 128 && g_col_engine == 0) {
    hipLaunchKernelGGL(xc_cols_fwd_wave1024, dim3((g.nkx + 4 * XC_FWDW_COLS - 1) / (4 * XC_FWDW_COLS), njobs),
                       dim3(256), 0, (hipStream_t)stream,
                       (const cfloat*)T1, filt, (cfloat*)S, (const cfloat*)tw_col, g, fix,
                       (const cfloat*)Mhat);
    return mc_check_launch();
  }
  if (g.H == 4096 && g.kyp <= 512 && g.kyn <= 512 && g_col_engine == 0) {
    hipLaunchKernelGGL((xc_cols_fwd<12, true>), dim3((g.nkx + XC_FWD_COLS - 1) / XC_FWD_COLS, njobs),
                       dim3(MC_WG), 0, (hipStream_t)stream,
                       (const cfloat*)T1, filt, (cfloat*)S, (const cfloat*)tw_col, g, fix,
                       (const cfloat*)Mhat);
    return mc_check_launch();
  }
  MC_DISPATCH_LOG(mc_ilog2(g.H), {
    hipLaunchKernelGGL(xc_cols_fwd<L>, grid, dim3(MC_WG), 0, (hipStream_t)stream,
                       (const cfloat*)T1, filt, (cfloat*)S, (const cfloat*)tw_col, g, fix,
                       (const cfloat*)Mhat);
  });
  return mc_check_launch();
}

int mc_xc_cols_inverse(const void* S_cur, const int* cur_idx, const void* S_ref,
                       const int* ref_idx, void* T2, const void* tw_col, float scale, int npairs,
                       const mc_xc_geom* q, void* stream) {
  XcGeom g;
  int rc = geom_from(q, &g, false, true);
  if (rc) return rc;
  if (!S_cur || !cur_idx || !S_ref || !ref_idx || !T2 || !tw_col || npairs < 1) return MC_ERR_ARG;
  if (g.H == 1024 && g.kyp <= 128 && g.kyn <= 128 && g_col_engine == 0) {
    hipLaunchKernelGGL(xc_cols_inv_wave1024, dim3((g.nkx + 3) / 4, npairs), dim3(256), 0, (hipStream_t)stream,
                       (const cfloat*)S_cur, cur_idx, (const cfloat*)S_ref, ref_idx, (cfloat*)T2,
                       (const cfloat*)tw_col, scale, g);
    return mc_check_launch();
  }
  dim3 grid(g.nkx, npairs);
  MC_DISPATCH_LOG(mc_ilog2(g.H), {
    hipLaunchKernelGGL((xc_cols_inv<L, 0>), grid, dim3(MC_WG), 0, (hipStream_t)stream,
                       (const cfloat*)S_cur, cur_idx, (const cfloat*)S_ref, ref_idx,
                       (const float*)nullptr, (cfloat*)T2, (const cfloat*)tw_col, scale, g,
                       (const int*)nullptr);
  });
  return mc_check_launch();
}

int mc_fourier_shift_cols_inverse(const void* S, const int* idx, const float* shifts, void* T2,
                                  const void* tw_col, float scale, int nframes,
                                  const mc_xc_geom* q, void* stream) {
  XcGeom g;
  int rc = geom_from(q, &g, false, true);
  if (rc) return rc;
  if (!S || !idx || !shifts || !T2 || !tw_col || nframes < 1) return MC_ERR_ARG;
  dim3 grid(g.nkx, nframes);
  MC_DISPATCH_LOG(mc_ilog2(g.H), {
    hipLaunchKernelGGL((xc_cols_inv<L, 1>), grid, dim3(MC_WG), 0, (hipStream_t)stream,
                       (const cfloat*)S, idx, (const cfloat*)nullptr, (const int*)nullptr, shifts,
                       (cfloat*)T2, (const cfloat*)tw_col, scale, g, (const int*)nullptr);
  });
  return mc_check_launch();
}

int mc_xc_rows_inverse_argmax(const void* T2, float* part_val, int* part_idx, int* peaks,
                              float* shifts, const void* tw_row, int npairs, const mc_xc_geom* q,
                              void* stream) {
  // part_idx holds npairs*(H/RG) candidates followed by npairs running maxima
  XcGeom g;
  int rc = geom_from(q, &g, true, false);
  if (rc) return rc;
  if (!T2 || !part_val || !part_idx || !peaks || !shifts || !tw_row || npairs < 1)
    return MC_ERR_ARG;
  const int logn = mc_ilog2(g.W) - 1;
  const size_t lds = rows_lds_bytes(g.W / 2, g);
  if (lds > 160 * 1024) return MC_ERR_ARG;
  const int ngrp = g.H / g.RG;
  int* best = part_idx + (int64_t)npairs * ngrp;
  {  // best[p] = order(-inf)
    const float ninf = -INFINITY;
    int pat;
    memcpy(&pat, &ninf, 4);
    pat = pat >= 0 ? pat : pat ^ 0x7fffffff;
    hipError_t e = hipMemsetD32Async((hipDeviceptr_t)best, pat, npairs, (hipStream_t)stream);
    if (e != hipSuccess) return (int)e;
  }
  int near = (64 + g.RG - 1) / g.RG;  // groups covering |shift_y| <= 64 px at each end
  if (2 * near > ngrp) near = ngrp / 2;
  float* bounds = part_val + (int64_t)npairs * ngrp;  // npairs * H row bounds
  if (ngrp - 2 * near > 0)
    hipLaunchKernelGGL(xc_row_bounds, dim3((g.H + 63) / 64, npairs), dim3(256), 0, (hipStream_t)stream,
                       (const cfloat*)T2, bounds, g.nkx, g.H);
  MC_DISPATCH_LOG(logn, {
    auto k = xc_rows_inv<L, 0>;
    if (lds > 64 * 1024)
      (void)hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (near > 0)
      hipLaunchKernelGGL(k, dim3(2 * near, npairs), dim3(MC_WG), lds, (hipStream_t)stream,
                         (const cfloat*)T2, (const float*)bounds, best, part_val, part_idx, (float*)nullptr,
                         (const int64_t*)nullptr, (int64_t)0, (const cfloat*)tw_row, g, near, 0, 0,
                         (const int*)nullptr);
    if (ngrp - 2 * near > 0)
      hipLaunchKernelGGL(k, dim3(ngrp - 2 * near, npairs), dim3(MC_WG), lds, (hipStream_t)stream,
                         (const cfloat*)T2, (const float*)bounds, best, part_val, part_idx, (float*)nullptr,
                         (const int64_t*)nullptr, (int64_t)0, (const cfloat*)tw_row, g, near, 1, 0,
                         (const int*)nullptr);
  });
  rc = mc_check_launch();
  if (rc) return rc;
  hipLaunchKernelGGL(xc_peak_final, dim3(npairs), dim3(64), 0, (hipStream_t)stream, part_val,
                     part_idx, ngrp, g.H, g.W, peaks, shifts, (const int*)nullptr);
  return mc_check_launch();
}

int mc_xc_near_rows(const mc_xc_geom* q) {
  XcGeom g;
  int rc = geom_from(q, &g, true, true);
  if (rc) return rc;
  int near = (64 + g.RG - 1) / g.RG;
  if (2 * near > g.H / g.RG) near = (g.H / g.RG) / 2;
  return near * g.RG + XC_NEAR_GUARD;  // searched rows + guard rows, per end of the map
}

int mc_xc_correlate_argmax(const void* S_cur, const int* cur_idx, const void* S_ref,
                           const int* ref_idx, void* T2_full, void* T2_near, float* part_val,
                           int* part_idx, int* peaks, float* shifts, const int* shift_rows,
                           int n_shift_rows, float* nb, const void* tw_col, const void* tw_row,
                           float scale, int npairs, const mc_xc_geom* q, void* stream) {
  XcGeom g;
  int rc = geom_from(q, &g, true, true);
  if (rc) return rc;
  if (!S_cur || !cur_idx || !S_ref || !ref_idx || !T2_full || !T2_near || !part_val || !part_idx ||
      !peaks || !shifts || !tw_col || !tw_row || npairs < 1)
    return MC_ERR_ARG;
  if (g.H < 1024) return MC_ERR_UNSUPPORTED;  // every thread must own H / 256 outputs of the last pass
  hipStream_t st = (hipStream_t)stream;
  const int logn = mc_ilog2(g.W) - 1;
  const size_t lds = rows_lds_bytes(g.W / 2, g);
  if (lds > 160 * 1024) return MC_ERR_ARG;
  const int ngrp = g.H / g.RG;
  int near = (64 + g.RG - 1) / g.RG;
  if (2 * near > ngrp) near = ngrp / 2;
  if (near < 1) return MC_ERR_UNSUPPORTED;
  const int nstore = near * g.RG + XC_NEAR_GUARD;
  if (nstore > 256) return MC_ERR_UNSUPPORTED;  // xc_cols_inv_near: the stored window lies in a thread's first / last row
  int* best = part_idx + (int64_t)npairs * ngrp;  // npairs running maxima, then the gate word
  int* gate = best + npairs;
  float* bounds = part_val + (int64_t)npairs * ngrp;  // npairs * H row bounds
  if (shift_rows && n_shift_rows < 1) return MC_ERR_ARG;
  hipLaunchKernelGGL(xc_search_init, dim3((npairs * g.H + 255) / 256), dim3(256), 0, st, best, gate, bounds,
                     npairs, npairs * g.H, shifts, shift_rows ? 2 * n_shift_rows : 0);
  if (g.H == 1024 && g.kyp <= 128 && g.kyn <= 128 && g_col_engine == 0) {
    hipLaunchKernelGGL(xc_cols_inv_near_wave1024, dim3((g.nkx + 4 * XC_NEAR_COLS_W - 1) / (4 * XC_NEAR_COLS_W), npairs),
                       dim3(256), 0, st, (const cfloat*)S_cur, cur_idx, (const cfloat*)S_ref, ref_idx,
                       (cfloat*)T2_near, bounds, (const cfloat*)tw_col, scale, g, nstore);
  } else if (g.H == 4096 && g.kyp <= 512 && g.kyn <= 512 && g_col_engine == 0) {
    hipLaunchKernelGGL((xc_cols_inv_near<12, true>), dim3((g.nkx + XC_NEAR_COLS - 1) / XC_NEAR_COLS, npairs),
                       dim3(MC_WG), 0, st, (const cfloat*)S_cur, cur_idx, (const cfloat*)S_ref, ref_idx,
                       (cfloat*)T2_near, bounds, (const cfloat*)tw_col, scale, g, nstore);
  } else
  MC_DISPATCH_LOG(mc_ilog2(g.H), {
    if constexpr (L >= 10) {
      hipLaunchKernelGGL(xc_cols_inv_near<L>, dim3((g.nkx + XC_NEAR_COLS - 1) / XC_NEAR_COLS, npairs),
                         dim3(MC_WG), 0, st, (const cfloat*)S_cur, cur_idx, (const cfloat*)S_ref, ref_idx,
                         (cfloat*)T2_near, bounds, (const cfloat*)tw_col, scale, g, nstore);
    } else {
      return MC_ERR_UNSUPPORTED;
    }
  });
  rc = mc_check_launch();
  if (rc) return rc;
  if (g_after_k3n_event) (void)hipEventRecord((hipEvent_t)g_after_k3n_event, st);  // pipeline schedules (mc_xc_after_k3n_event)
  const int nfar = ngrp - 2 * near;
  MC_DISPATCH_LOG(logn, {
    auto k = xc_rows_inv<L, 0>;
    if (lds > 64 * 1024)
      (void)hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipLaunchKernelGGL(k, dim3(2 * near, npairs), dim3(MC_WG), lds, st, (const cfloat*)T2_near,
                       (const float*)bounds, best, part_val, part_idx, (float*)nullptr,
                       (const int64_t*)nullptr, (int64_t)0, (const cfloat*)tw_row, g, near, 0,
                       1 + XC_NEAR_GUARD, (const int*)nullptr);
  });
  rc = mc_check_launch();
  if (rc) return rc;
  if (nfar > 0) {
    hipLaunchKernelGGL(xc_far_needed, dim3((nfar + 63) / 64, npairs), dim3(64), 0, st,
                       (const float*)bounds, (const int*)best, part_val, part_idx, gate, g.H, g.RG, near);
    // fallback, skipped on the device unless a far row can still win: full map + far phase
    MC_DISPATCH_LOG(mc_ilog2(g.H), {
      hipLaunchKernelGGL((xc_cols_inv<L, 0>), dim3(g.nkx, npairs), dim3(MC_WG), 0, st,
                         (const cfloat*)S_cur, cur_idx, (const cfloat*)S_ref, ref_idx,
                         (const float*)nullptr, (cfloat*)T2_full, (const cfloat*)tw_col, scale, g,
                         (const int*)gate);
    });
    MC_DISPATCH_LOG(logn, {
      auto k = xc_rows_inv<L, 0>;
      hipLaunchKernelGGL(k, dim3(nfar, npairs), dim3(MC_WG), lds, st, (const cfloat*)T2_full,
                         (const float*)bounds, best, part_val, part_idx, (float*)nullptr,
                         (const int64_t*)nullptr, (int64_t)0, (const cfloat*)tw_row, g, near, 1, 0,
                         (const int*)gate);
    });
    rc = mc_check_launch();
    if (rc) return rc;
  }
  hipLaunchKernelGGL(xc_peak_final, dim3(npairs), dim3(64), 0, st, part_val, part_idx, ngrp, g.H, g.W,
                     peaks, shifts, shift_rows);
  if (nb) {  // 3 x 3 values around every peak (sub-pixel refinement), from whichever buffer holds the rows
    MC_DISPATCH_LOG(logn, {
      hipLaunchKernelGGL(xc_peak_nbhd<L>, dim3(3, npairs), dim3(MC_WG), 0, st, (const cfloat*)T2_full,
                         (const int*)peaks, nb, (const cfloat*)tw_row, g, (const cfloat*)T2_near,
                         (const int*)gate, nstore);
    });
  }
  return mc_check_launch();
}

int mc_xc_rows_inverse_store(const void* T2, float* out, const int64_t* out_off,
                             int64_t out_stride, const void* tw_row, int nframes,
                             const mc_xc_geom* q, void* stream) {
  XcGeom g;
  int rc = geom_from(q, &g, true, false);
  if (rc) return rc;
  if (!T2 || !out || !out_off || !tw_row || nframes < 1) return MC_ERR_ARG;
  const int logn = mc_ilog2(g.W) - 1;
  const size_t lds = rows_lds_bytes(g.W / 2, g);
  if (lds > 160 * 1024) return MC_ERR_ARG;
  dim3 grid(g.H / g.RG, nframes);
  MC_DISPATCH_LOG(logn, {
    auto k = xc_rows_inv<L, 1>;
    if (lds > 64 * 1024)
      (void)hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipLaunchKernelGGL(k, grid, dim3(MC_WG), lds, (hipStream_t)stream, (const cfloat*)T2,
                       (const float*)nullptr, (int*)nullptr, (float*)nullptr, (int*)nullptr, out, out_off, out_stride,
                       (const cfloat*)tw_row, g, 0, 0, 0, (const int*)nullptr);
  });
  return mc_check_launch();
}

int mc_xc_peak_neighbourhood(const void* T2, const int* peaks, float* nb, const void* tw_row,
                             int npairs, const mc_xc_geom* q, void* stream) {
  XcGeom g;
  int rc = geom_from(q, &g, true, false);
  if (rc) return rc;
  if (!T2 || !peaks || !nb || !tw_row || npairs < 1) return MC_ERR_ARG;
  const int logn = mc_ilog2(g.W) - 1;
  dim3 grid(3, npairs);
  MC_DISPATCH_LOG(logn, {
    hipLaunchKernelGGL(xc_peak_nbhd<L>, grid, dim3(MC_WG), 0, (hipStream_t)stream,
                       (const cfloat*)T2, peaks, nb, (const cfloat*)tw_row, g, (const cfloat*)nullptr,
                       (const int*)nullptr, 0);
  });
  return mc_check_launch();
}

}  // extern "C"

