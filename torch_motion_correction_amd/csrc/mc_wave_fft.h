// Wavefront-private 2048-point complex FFT (= one real row of 4096 samples) for gfx950.
//
// One wave64 transforms one line on its own: no workgroup barriers, the data stay in
// registers (32 complex per lane) and only cross lanes through an 8 KiB LDS slab that
// belongs to the wave.  Decomposition N = 16 x 16 x 8 (decimation in time, natural order
// in, natural order out):
//
//   n = 128 n1 + 8 n2 + n3,   k = k1 + 16 k2 + 256 k3
//   pass A  radix 16 over n1, twiddle W_2048^{q k1}   (q = 8 n2 + n3; lane t owns q = 2t, 2t+1)
//   pass B  radix 16 over n2, twiddle W_128^{n3 k2}   (lane t owns (k1, n3) = (t & 15, 2 (t >> 4) + h))
//   pass C  radix  8 over n3                          (lane t owns c = k1 + 16 k2 in
//                                                      {t | 128, 256 - t, 64 + t, 192 - t})
//
// Each exchange moves one half of the line at a time (h = parity of n3) through the same
// 1024-entry slab, which is what keeps the slab at 8 KiB per wave.  The pass-C ownership
// puts bin k and bin N - k in the same lane, so the real-FFT unpack
//   X[k] = (Z[k] + conj Z[N-k]) / 2 - i/2 w^k (Z[k] - conj Z[N-k]),   w = exp(-2 pi i / 4096)
// needs no third exchange.  Only the bins k3 in {0..KEEP-1} and {8-KEEP..7} of every
// radix-8 butterfly are used (band-pass pruning): the rest is dead code.
//
// Arithmetic is written on 2-float vectors (re, im) so that it maps onto the packed fp32
// VALU instructions (v_pk_add_f32 / v_pk_mul_f32 / v_pk_fma_f32: one instruction per
// complex add, two per complex multiply); the rotations by -i and the (-im, re) operand of
// a complex multiply are expressed through the instructions' op_sel / neg modifiers
// (inline asm), which cost nothing.  The kernel is VALU-issue bound, so the instruction
// count is what matters.
//
// The lane-level functions are __host__ __device__ so that tests/host_wave_fft.cpp can
// run the identical index algebra lane by lane on the CPU (clang++, no GPU needed).
#pragma once

#ifdef __HIPCC__
#include "mc_common.h"
#define MC_HD __host__ __device__ __forceinline__
#else
#include <math.h>
#include <stdint.h>
struct cfloat {
  float x, y;
};
#define MC_HD static inline
#endif

typedef float wf2 __attribute__((ext_vector_type(2)));  // (re, im)

#define WF_N 2048       // complex points per line
#define WF_SLAB 1024    // complex entries of a wave's LDS slab (8 KiB)

MC_HD wf2 wf_make(float re, float im) { return wf2{re, im}; }
MC_HD wf2 wf_from(cfloat c) { return wf2{c.x, c.y}; }
MC_HD cfloat wf_to(wf2 v) { return cfloat{v.x, v.y}; }

// a + (-i) b = (a.re + b.im, a.im - b.re)   and   a - (-i) b
MC_HD wf2 wf_add_mi(wf2 a, wf2 b) {
#if defined(__HIP_DEVICE_COMPILE__)
  wf2 r;
  asm("v_pk_add_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,0] neg_hi:[0,1]" : "=v"(r) : "v"(a), "v"(b));
  return r;
#else
  return wf2{a.x + b.y, a.y - b.x};
#endif
}
MC_HD wf2 wf_sub_mi(wf2 a, wf2 b) {
#if defined(__HIP_DEVICE_COMPILE__)
  wf2 r;
  asm("v_pk_add_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,0] neg_lo:[0,1]" : "=v"(r) : "v"(a), "v"(b));
  return r;
#else
  return wf2{a.x - b.y, a.y + b.x};
#endif
}
// complex product a * w = a.re * (w.re, w.im) + a.im * (-w.im, w.re)
MC_HD wf2 wf_cmul(wf2 a, wf2 w) {
#if defined(__HIP_DEVICE_COMPILE__)
  wf2 t, r;
  asm("v_pk_mul_f32 %0, %1, %2 op_sel_hi:[0,1]" : "=v"(t) : "v"(a), "v"(w));
  asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[1,1,0] op_sel_hi:[1,0,1] neg_lo:[0,1,0]"
      : "=v"(r)
      : "v"(a), "v"(w), "v"(t));
  return r;
#else
  return wf2{a.x * w.x - a.y * w.y, a.x * w.y + a.y * w.x};
#endif
}
// a + w * b  and  a - w * b  with a compile-time constant w = (cr, ci): the butterflies'
// internal twiddles folded into the following add (two packed FMAs each)
MC_HD wf2 wf_fma_c(wf2 a, wf2 b, float cr, float ci) {
  const wf2 c = {cr, ci}, cs = {-ci, cr};
  return __builtin_elementwise_fma(b.yy, cs, __builtin_elementwise_fma(b.xx, c, a));
}
MC_HD wf2 wf_cmulc(wf2 a, float cr, float ci) {
  const wf2 c = {cr, ci}, cs = {-ci, cr};
  return __builtin_elementwise_fma(a.yy, cs, a.xx * c);
}

// forward radix-4 butterfly, natural order out
MC_HD void wf_bfly4(wf2& a0, wf2& a1, wf2& a2, wf2& a3) {
  const wf2 t0 = a0 + a2, t1 = a0 - a2;
  const wf2 t2 = a1 + a3, d = a1 - a3;
  a0 = t0 + t2;
  a2 = t0 - t2;
  a1 = wf_add_mi(t1, d);
  a3 = wf_sub_mi(t1, d);
}

// forward 16-point DFT in place, natural order in and out:  n = j + 4 m,  k = p + 4 r
MC_HD void wf_dft16(wf2 (&a)[16]) {
  const float C1 = 0.92387953251128673848f, S1 = 0.38268343236508978178f;
  const float H = 0.70710678118654752440f;
#pragma unroll
  for (int j = 0; j < 4; ++j) wf_bfly4(a[j], a[j + 4], a[j + 8], a[j + 12]);
  // a[j + 4 p] now holds the p-th output of column j; twiddle W_16^{j p}
  a[1 + 4] = wf_cmulc(a[1 + 4], C1, -S1);   // W^1
  a[1 + 8] = wf_cmulc(a[1 + 8], H, -H);     // W^2
  a[1 + 12] = wf_cmulc(a[1 + 12], S1, -C1); // W^3
  a[2 + 4] = wf_cmulc(a[2 + 4], H, -H);     // W^2
  // a[2 + 8] * W^4 = -i a[2 + 8]: folded into the second stage below
  a[2 + 12] = wf_cmulc(a[2 + 12], -H, -H);  // W^6
  a[3 + 4] = wf_cmulc(a[3 + 4], S1, -C1);   // W^3
  a[3 + 8] = wf_cmulc(a[3 + 8], -H, -H);    // W^6
  a[3 + 12] = wf_cmulc(a[3 + 12], -C1, S1); // W^9
#pragma unroll
  for (int p = 0; p < 4; ++p) {
    wf2 &b0 = a[4 * p], &b1 = a[4 * p + 1], &b2 = a[4 * p + 2], &b3 = a[4 * p + 3];
    if (p == 2) {  // b2 stands for -i b2
      const wf2 t0 = wf_add_mi(b0, b2), t1 = wf_sub_mi(b0, b2);
      const wf2 t2 = b1 + b3, d = b1 - b3;
      b0 = t0 + t2;
      b2 = t0 - t2;
      b1 = wf_add_mi(t1, d);
      b3 = wf_sub_mi(t1, d);
    } else {
      wf_bfly4(b0, b1, b2, b3);
    }
  }
  // a[4 p + r] = X[p + 4 r]  ->  transpose to natural order (register renaming only)
#pragma unroll
  for (int p = 0; p < 4; ++p)
#pragma unroll
    for (int r = p + 1; r < 4; ++r) {
      const wf2 s = a[4 * p + r];
      a[4 * p + r] = a[4 * r + p];
      a[4 * r + p] = s;
    }
}

// wf_dft16 of an input whose entries 2..13 are zero (a column of a band-passed spectrum: of the 16
// interleaved sub-sequences of pass A only the rows |ky| < 512 of 4096 carry data).  Every column j of
// the 4 x 4 decomposition then has ONE non-zero input, so its 4-point DFT is that input times W_4^{m p}
// (m = 0 for j = 0, 1: all four outputs equal it; m = 3 for j = 2, 3: x, i x, -x, -i x) and the whole
// first stage folds into the W_16 twiddles: 9 constant complex multiplies instead of 4 butterflies + 9.
MC_HD void wf_dft16_lo2(wf2 (&a)[16]) {
  const float C1 = 0.92387953251128673848f, S1 = 0.38268343236508978178f;
  const float H = 0.70710678118654752440f;
  const wf2 x0 = a[0], x1 = a[1], x2 = a[14], x3 = a[15];
  a[0] = x0; a[4] = x0; a[8] = x0; a[12] = x0;
  a[1] = x1;
  a[1 + 4] = wf_cmulc(x1, C1, -S1);   // W^1
  a[1 + 8] = wf_cmulc(x1, H, -H);     // W^2
  a[1 + 12] = wf_cmulc(x1, S1, -C1);  // W^3
  a[2] = x2;
  a[2 + 4] = wf_cmulc(x2, H, H);      // i W^2
  a[2 + 8] = -x2;                     // (-1); W^4 = -i is folded into the second stage as in wf_dft16
  a[2 + 12] = wf_cmulc(x2, -H, H);    // -i W^6
  a[3] = x3;
  a[3 + 4] = wf_cmulc(x3, C1, S1);    // i W^3
  a[3 + 8] = wf_cmulc(x3, H, H);      // -W^6
  a[3 + 12] = wf_cmulc(x3, S1, C1);   // -i W^9
#pragma unroll
  for (int p = 0; p < 4; ++p) {
    wf2 &b0 = a[4 * p], &b1 = a[4 * p + 1], &b2 = a[4 * p + 2], &b3 = a[4 * p + 3];
    if (p == 2) {  // b2 stands for -i b2
      const wf2 t0 = wf_add_mi(b0, b2), t1 = wf_sub_mi(b0, b2);
      const wf2 t2 = b1 + b3, d = b1 - b3;
      b0 = t0 + t2;
      b2 = t0 - t2;
      b1 = wf_add_mi(t1, d);
      b3 = wf_sub_mi(t1, d);
    } else {
      wf_bfly4(b0, b1, b2, b3);
    }
  }
#pragma unroll
  for (int p = 0; p < 4; ++p)
#pragma unroll
    for (int r = p + 1; r < 4; ++r) {
      const wf2 s = a[4 * p + r];
      a[4 * p + r] = a[4 * r + p];
      a[4 * r + p] = s;
    }
}

// forward 8-point DFT, natural order in (e = even inputs n3 = 0,2,4,6; o = odd inputs
// 1,3,5,7); only outputs 0, 1, 6, 7 (KEEP = 2) or 0, 7 (KEEP = 1) are produced
template <int KEEP>
MC_HD void wf_dft8_pruned(const wf2 (&e)[4], const wf2 (&o)[4], wf2 (&z)[8]) {
  const float H = 0.70710678118654752440f;
  wf2 e0 = e[0], e1 = e[1], e2 = e[2], e3 = e[3];
  wf2 o0 = o[0], o1 = o[1], o2 = o[2], o3 = o[3];
  wf_bfly4(e0, e1, e2, e3);  // E[0..3]
  wf_bfly4(o0, o1, o2, o3);  // O[0..3]
  // X[k] = E[k & 3] + W_8^k O[k & 3];  W^1 = (H,-H), W^2 = -i, W^3 = (-H,-H), W^{k+4} = -W^k
  z[0] = e0 + o0;
  z[7] = wf_fma_c(e3, o3, H, H);  // W^7 = -W^3
  if (KEEP >= 2) {
    z[1] = wf_fma_c(e1, o1, H, -H);
    z[6] = wf_sub_mi(e2, o2);  // W^6 = -W^2 = +i
  }
  if (KEEP >= 4) {
    z[2] = wf_add_mi(e2, o2);
    z[3] = wf_fma_c(e3, o3, -H, -H);
    z[4] = e0 - o0;
    z[5] = wf_fma_c(e1, o1, -H, H);
  }
}

// ------------------------------------------------------------------ lane geometry
// Slab addresses (complex index, < 1024) of the two exchanges for lane t.
struct WfLane {
  int x1w_base, x1w_mask;  // exchange 1 write: slab[x1w_base + (k1 ^ x1w_mask)]
  int x1r[4];              // exchange 1 read:  slab[x1r[n2 & 3] + 64 * n2]
  int x2w;                 // exchange 2 write: slab[x2w + 16 * k2]
  int x2r[4];              // exchange 2 read:  slab[x2r[s] + 256 * n3h], s = butterfly slot
  int kbin[4];             // spectrum bin c of butterfly slot s (a0, a1, b0, b1); + 256 k3
  int self;                // lane 0: slots a0 / a1 pair with themselves
};

MC_HD WfLane wf_lane(int t) {
  WfLane L;
  {  // source of exchange 1: q = 2 t + h  ->  n2 = t >> 2, n3h = t & 3
    const int n2 = t >> 2, h1 = (t >> 1) & 1, h0 = t & 1;
    L.x1w_base = (n2 * 2 + h1) * 32 + h0 * 16;
    L.x1w_mask = (n2 & 3) | (h1 << 2) | (h0 << 3);
  }
  {  // destination of exchange 1 / source of exchange 2: k1 = t & 15, n3h = t >> 4
    const int k1 = t & 15, g2 = t >> 4, h1 = g2 >> 1, h0 = g2 & 1;
    const int lanepart = h1 * 32 + h0 * 16, lo = k1 ^ ((h1 << 2) | (h0 << 3));
    for (int v = 0; v < 4; ++v) L.x1r[v] = lanepart + (lo ^ v);
    L.x2w = g2 * 256 + k1;
  }
  L.kbin[0] = t;
  L.kbin[1] = t == 0 ? 128 : 256 - t;
  L.kbin[2] = 64 + t;
  L.kbin[3] = 192 - t;
  for (int s = 0; s < 4; ++s) L.x2r[s] = L.kbin[s];
  L.self = t == 0;
  return L;
}

// Real-FFT unpack of one bin: zk = Z[k], zmr = Z[N - k] (not yet conjugated), wk = w^k.
//   X = 0.5 * ((zk + conj zm) - i * wk * (zk - conj zm))
MC_HD wf2 wf_unpack(wf2 zk, wf2 zmr, wf2 wk) {
  const wf2 zm = {zmr.x, -zmr.y};
  const wf2 sm = zk + zm, d = zk - zm;
  const wf2 wd = wf_cmul(d, wk);
  return wf_add_mi(sm, wd) * 0.5f;
}

// From the pruned outputs of a lane's four radix-8 butterflies to its 4 * KEEP real-FFT
// bins:  X[s][k3] is bin kbin[s] + 256 k3.  wk[s] = w^{kbin[s]}.
template <int KEEP>
MC_HD void wf_unpack_lane(const wf2 (&z)[4][8], const wf2 (&wk)[4], bool self, wf2 (&X)[4][KEEP]) {
  const float C1 = 0.92387953251128673848f, S1 = 0.38268343236508978178f;  // w^256
#pragma unroll
  for (int k3 = 0; k3 < KEEP; ++k3) {
    // general lanes: slot s pairs with slot s ^ 1 at 7 - k3; lane 0: a0 pairs with a0 at
    // (8 - k3) & 7, a1 with a1 at 7 - k3
    const wf2 m_a0 = self ? z[0][(8 - k3) & 7] : z[1][7 - k3];
    const wf2 m_a1 = self ? z[1][7 - k3] : z[0][7 - k3];
    wf2 w[4] = {wk[0], wk[1], wk[2], wk[3]};
    if (k3 == 1) {
#pragma unroll
      for (int s = 0; s < 4; ++s) w[s] = wf_cmulc(w[s], C1, -S1);
    }
    X[0][k3] = wf_unpack(z[0][k3], m_a0, w[0]);
    X[1][k3] = wf_unpack(z[1][k3], m_a1, w[1]);
    X[2][k3] = wf_unpack(z[2][k3], z[3][7 - k3], w[2]);
    X[3][k3] = wf_unpack(z[3][k3], z[2][7 - k3], w[3]);
  }
}

// =====================================================================================
// 512-point variant (= one real row of 1024 samples: the rows of 1024 x 1024 patches).
// N = 8 x 8 x 8, eight complex values per lane, ONE radix-8 butterfly per lane and pass:
//   n = 64 n1 + 8 n2 + n3,   k = k1 + 8 k2 + 64 k3
//   pass A  radix 8 over n1, twiddle W_512^{q k1}   lane t owns q = 8 n2 + n3 = t
//   pass B  radix 8 over n2, twiddle W_64^{n3 k2}   lane t owns (k1, n3) = (t & 7, t >> 3)
//   pass C  radix 8 over n3                          lane t owns c = k1 + 8 k2 = t
// Two whole-line exchanges through a 512-entry (4 KiB) wave-private slab.  Bin k = t + 64 k3
// pairs with bin 512 - k, which lives in lane (64 - t) & 63 at 7 - k3 (lane 0: itself at
// (8 - k3) & 7): the real-FFT unpack fetches it with a lane permute instead of a third
// exchange.  Only k3 in {0, 1} (k < 128) and their partners {7, 6} are produced.
// =====================================================================================
#define WF5_N 512
#define WF5_SLAB 512

MC_HD void wf_dft8(wf2 (&a)[8]) {  // full forward 8-point DFT, natural order in and out
  const float H = 0.70710678118654752440f;
  wf2 e0 = a[0], e1 = a[2], e2 = a[4], e3 = a[6];
  wf2 o0 = a[1], o1 = a[3], o2 = a[5], o3 = a[7];
  wf_bfly4(e0, e1, e2, e3);
  wf_bfly4(o0, o1, o2, o3);
  a[0] = e0 + o0;
  a[4] = e0 - o0;
  a[1] = wf_fma_c(e1, o1, H, -H);
  a[5] = wf_fma_c(e1, o1, -H, H);
  a[2] = wf_add_mi(e2, o2);
  a[6] = wf_sub_mi(e2, o2);
  a[3] = wf_fma_c(e3, o3, -H, -H);
  a[7] = wf_fma_c(e3, o3, H, H);
}

// Slab addresses (complex index < 512), conflict-free for ds_write_b64 (16-lane groups) and
// ds_read_b64 (32-lane groups):
//   exchange 1: entry (k1, n2, n3) at ((n2 * 8 + (k1 ^ n2)) * 8) + (n3 ^ (4 * (k1 >> 2)))
//   exchange 2: entry (k2, n3, k1) at ((n3 * 8 + (k2 ^ n3)) * 8) + k1
MC_HD int wf5_x1(int k1, int n2, int n3) { return ((n2 * 8 + (k1 ^ n2)) * 8) + (n3 ^ (4 * (k1 >> 2))); }
MC_HD int wf5_x2(int k2, int n3, int k1) { return ((n3 * 8 + (k2 ^ n3)) * 8) + k1; }

// =====================================================================================
// 1024-point variant for the COLUMNS of 1024 x 1024 patches (complex in, complex out).
// N = 16 x 8 x 8, sixteen values per lane:  n = 64 n1 + 8 n2 + n3,  k = k1 + 16 k2 + 128 k3
//   pass A  radix 16 over n1, twiddle W_1024^{q k1}   lane t owns q = 8 n2 + n3 = t
//   pass B  radix  8 over n2, twiddle W_64^{n3 k2}    lane t owns (k1, n3) = (t & 15, (t >> 4) + 4 b), b = 0, 1
//   pass C  radix  8 over n3                           lane t owns (k1, k2) = (t & 15, (t >> 4) + 4 b)
// Two whole-line exchanges through a 1024-entry (8 KiB) wave-private slab:
//   exchange 1: entry (k1, q)       at k1 * 64 + (q ^ ((2 * k1) & 31))
//   exchange 2: entry (k1, k2, n3)  at n3 * 128 + k2 * 16 + k1
// (conflict-free for 8-byte writes in 16-lane groups and 8-byte reads in 32-lane groups).
// =====================================================================================
#define WF10_N 1024
MC_HD int wf10_x1(int k1, int q) { return k1 * 64 + (q ^ ((2 * k1) & 31)); }
MC_HD int wf10_x2(int k1, int k2, int n3) { return n3 * 128 + k2 * 16 + k1; }

// w^1 .. w^7 from w^1 (depth <= 3 products), applied to a[1..7]
MC_HD void wf_twiddle8(wf2 (&a)[8], wf2 w1) {
  const wf2 w2 = wf_cmul(w1, w1), w3 = wf_cmul(w2, w1), w4 = wf_cmul(w2, w2);
  a[1] = wf_cmul(a[1], w1);
  a[2] = wf_cmul(a[2], w2);
  a[3] = wf_cmul(a[3], w3);
  a[4] = wf_cmul(a[4], w4);
  a[5] = wf_cmul(a[5], wf_cmul(w4, w1));
  a[6] = wf_cmul(a[6], wf_cmul(w4, w2));
  a[7] = wf_cmul(a[7], wf_cmul(w4, w3));
}
// w^1 .. w^15 from the exact w^1, w^2, w^4, w^8, applied to a[1..15]
MC_HD void wf_twiddle16(wf2 (&a)[16], wf2 w1, wf2 w2, wf2 w4, wf2 w8) {
  const wf2 w3 = wf_cmul(w2, w1), w5 = wf_cmul(w4, w1), w6 = wf_cmul(w4, w2), w7 = wf_cmul(w4, w3);
  a[1] = wf_cmul(a[1], w1); a[2] = wf_cmul(a[2], w2); a[3] = wf_cmul(a[3], w3); a[4] = wf_cmul(a[4], w4);
  a[5] = wf_cmul(a[5], w5); a[6] = wf_cmul(a[6], w6); a[7] = wf_cmul(a[7], w7); a[8] = wf_cmul(a[8], w8);
  a[9] = wf_cmul(a[9], wf_cmul(w8, w1)); a[10] = wf_cmul(a[10], wf_cmul(w8, w2));
  a[11] = wf_cmul(a[11], wf_cmul(w8, w3)); a[12] = wf_cmul(a[12], wf_cmul(w8, w4));
  a[13] = wf_cmul(a[13], wf_cmul(w8, w5)); a[14] = wf_cmul(a[14], wf_cmul(w8, w6));
  a[15] = wf_cmul(a[15], wf_cmul(w8, w7));
}
