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
// The lane-level functions below are __host__ __device__ so that tests/host_wave_fft.cpp
// can run the identical index algebra lane by lane on the CPU (g++, no GPU needed).
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
static inline cfloat cmake(float a, float b) { return cfloat{a, b}; }
static inline cfloat cadd(cfloat a, cfloat b) { return cfloat{a.x + b.x, a.y + b.y}; }
static inline cfloat csub(cfloat a, cfloat b) { return cfloat{a.x - b.x, a.y - b.y}; }
static inline cfloat cmul(cfloat a, cfloat b) {
  return cfloat{a.x * b.x - a.y * b.y, a.x * b.y + a.y * b.x};
}
static inline cfloat cconj(cfloat a) { return cfloat{a.x, -a.y}; }
#define MC_HD static inline
#endif

#define WF_N 2048       // complex points per line
#define WF_SLAB 1024    // complex entries of a wave's LDS slab (8 KiB)

// multiply by the compile-time constant (cr, ci)
MC_HD cfloat wf_cmulc(cfloat a, float cr, float ci) {
  return cmake(a.x * cr - a.y * ci, a.x * ci + a.y * cr);
}
// multiply by -i (forward transforms only need this one)
MC_HD cfloat wf_mul_mi(cfloat a) { return cmake(a.y, -a.x); }

// forward radix-4 butterfly, natural order out
MC_HD void wf_bfly4(cfloat& a0, cfloat& a1, cfloat& a2, cfloat& a3) {
  const cfloat t0 = cadd(a0, a2), t1 = csub(a0, a2);
  const cfloat t2 = cadd(a1, a3), t3 = wf_mul_mi(csub(a1, a3));
  a0 = cadd(t0, t2);
  a1 = cadd(t1, t3);
  a2 = csub(t0, t2);
  a3 = csub(t1, t3);
}

// forward 16-point DFT in place, natural order in and out:  n = j + 4 m,  k = p + 4 r
MC_HD void wf_dft16(cfloat (&a)[16]) {
  const float C1 = 0.92387953251128673848f, S1 = 0.38268343236508978178f;
  const float H = 0.70710678118654752440f;
#pragma unroll
  for (int j = 0; j < 4; ++j) wf_bfly4(a[j], a[j + 4], a[j + 8], a[j + 12]);
  // a[j + 4 p] now holds the p-th output of column j; twiddle W_16^{j p}
  a[1 + 4] = wf_cmulc(a[1 + 4], C1, -S1);   // W^1
  a[1 + 8] = wf_cmulc(a[1 + 8], H, -H);     // W^2
  a[1 + 12] = wf_cmulc(a[1 + 12], S1, -C1); // W^3
  a[2 + 4] = wf_cmulc(a[2 + 4], H, -H);     // W^2
  a[2 + 8] = wf_mul_mi(a[2 + 8]);           // W^4
  a[2 + 12] = wf_cmulc(a[2 + 12], -H, -H);  // W^6
  a[3 + 4] = wf_cmulc(a[3 + 4], S1, -C1);   // W^3
  a[3 + 8] = wf_cmulc(a[3 + 8], -H, -H);    // W^6
  a[3 + 12] = wf_cmulc(a[3 + 12], -C1, S1); // W^9
#pragma unroll
  for (int p = 0; p < 4; ++p) wf_bfly4(a[4 * p], a[4 * p + 1], a[4 * p + 2], a[4 * p + 3]);
  // a[4 p + r] = X[p + 4 r]  ->  transpose to natural order
#pragma unroll
  for (int p = 0; p < 4; ++p)
#pragma unroll
    for (int r = p + 1; r < 4; ++r) {
      const cfloat s = a[4 * p + r];
      a[4 * p + r] = a[4 * r + p];
      a[4 * r + p] = s;
    }
}

// forward 8-point DFT, natural order in (e = even inputs n3 = 0,2,4,6; o = odd inputs
// 1,3,5,7); only outputs 0, 1, 6, 7 (KEEP = 2) or 0, 7 (KEEP = 1) are produced
template <int KEEP>
MC_HD void wf_dft8_pruned(const cfloat (&e)[4], const cfloat (&o)[4], cfloat (&z)[8]) {
  const float H = 0.70710678118654752440f;
  cfloat e0 = e[0], e1 = e[1], e2 = e[2], e3 = e[3];
  cfloat o0 = o[0], o1 = o[1], o2 = o[2], o3 = o[3];
  wf_bfly4(e0, e1, e2, e3);  // E[0..3]
  wf_bfly4(o0, o1, o2, o3);  // O[0..3]
  // X[k] = E[k & 3] + W_8^k O[k & 3]
  z[0] = cadd(e0, o0);
  z[7] = csub(e3, wf_cmulc(o3, -H, -H));  // W^7 = -W^3, W^3 = (-H, -H)
  if (KEEP >= 2) {
    z[1] = cadd(e1, wf_cmulc(o1, H, -H));  // W^1
    z[6] = csub(e2, wf_mul_mi(o2));        // W^6 = -W^2, W^2 = -i
  }
  if (KEEP >= 4) {
    z[2] = cadd(e2, wf_mul_mi(o2));
    z[3] = cadd(e3, wf_cmulc(o3, -H, -H));
    z[4] = csub(e0, o0);
    z[5] = csub(e1, wf_cmulc(o1, H, -H));
  }
}

// w^1 .. w^15 from w^1 (depth <= 4 products), applied to a[1..15]
MC_HD void wf_twiddle16(cfloat (&a)[16], cfloat w1) {
  const cfloat w2 = cmul(w1, w1), w4 = cmul(w2, w2), w8 = cmul(w4, w4);
  const cfloat w3 = cmul(w2, w1), w5 = cmul(w4, w1), w6 = cmul(w4, w2), w7 = cmul(w4, w3);
  a[1] = cmul(a[1], w1);
  a[2] = cmul(a[2], w2);
  a[3] = cmul(a[3], w3);
  a[4] = cmul(a[4], w4);
  a[5] = cmul(a[5], w5);
  a[6] = cmul(a[6], w6);
  a[7] = cmul(a[7], w7);
  a[8] = cmul(a[8], w8);
  a[9] = cmul(a[9], cmul(w8, w1));
  a[10] = cmul(a[10], cmul(w8, w2));
  a[11] = cmul(a[11], cmul(w8, w3));
  a[12] = cmul(a[12], cmul(w8, w4));
  a[13] = cmul(a[13], cmul(w8, w5));
  a[14] = cmul(a[14], cmul(w8, w6));
  a[15] = cmul(a[15], cmul(w8, w7));
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

// Real-FFT unpack of one bin: Zk = Z[k], Zm = Z[N - k] (not yet conjugated), wk = w^k.
MC_HD cfloat wf_unpack(cfloat zk, cfloat zmr, cfloat wk) {
  const cfloat zm = cconj(zmr);
  const cfloat sm = cadd(zk, zm), d = csub(zk, zm);
  const cfloat wd = cmul(wk, d);  // -i * wd = (wd.y, -wd.x)
  return cmake(0.5f * (sm.x + wd.y), 0.5f * (sm.y - wd.x));
}

// From the pruned outputs of a lane's four radix-8 butterflies to its 4 * KEEP real-FFT
// bins:  X[s][k3] is bin kbin[s] + 256 k3.  wk[s] = w^{kbin[s]}.
template <int KEEP>
MC_HD void wf_unpack_lane(const cfloat (&z)[4][8], const cfloat (&wk)[4], bool self,
                          cfloat (&X)[4][KEEP]) {
  const float C1 = 0.92387953251128673848f, S1 = 0.38268343236508978178f;  // w^256
#pragma unroll
  for (int k3 = 0; k3 < KEEP; ++k3) {
    // general lanes: slot s pairs with slot s ^ 1 at 7 - k3; lane 0: a0 pairs with a0 at
    // (8 - k3) & 7, a1 with a1 at 7 - k3
    const cfloat m_a0 = self ? z[0][(8 - k3) & 7] : z[1][7 - k3];
    const cfloat m_a1 = self ? z[1][7 - k3] : z[0][7 - k3];
    cfloat w[4] = {wk[0], wk[1], wk[2], wk[3]};
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
