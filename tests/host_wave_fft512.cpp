// CPU check of the 512-point wave transform's index algebra (csrc/mc_wave_fft.h, second
// half): 64 lanes executed one after the other over a shared 512-entry slab, exactly as
// xc_rows_fwd_wave512 does; the 2 bins every lane ends up with are compared with a
// double-precision DFT of the same real row.
//
//   /opt/rocm/lib/llvm/bin/clang++ -O1 -std=c++17 -I torch_motion_correction_amd/csrc tests/host_wave_fft512.cpp -o /tmp/host_wave_fft512 -lm && /tmp/host_wave_fft512
#include <stdio.h>
#include <stdlib.h>

#include <vector>

#include "mc_wave_fft.h"

static wf2 tw1024(int k) {
  const double a = -2.0 * M_PI * (double)k / 1024.0;
  return wf_make((float)cos(a), (float)sin(a));
}

int main() {
  std::vector<float> row(1024);
  srand(7);
  for (int i = 0; i < 1024; ++i) row[i] = (i >= 126 && i < 900) ? (float)(rand() % 65536) / 65536.f - 0.5f : 0.f;
  static wf2 A[64][8], B[64][8], C[64][8], slab[WF5_SLAB];
  for (int t = 0; t < 64; ++t) {  // pass A: q = t
    for (int n1 = 0; n1 < 8; ++n1) A[t][n1] = wf_make(row[128 * n1 + 2 * t], row[128 * n1 + 2 * t + 1]);
    wf_dft8(A[t]);
    for (int k1 = 1; k1 < 8; ++k1) A[t][k1] = wf_cmul(A[t][k1], tw1024(2 * t * k1));  // W_512^{q k1}
  }
  for (int i = 0; i < WF5_SLAB; ++i) slab[i] = wf_make(NAN, NAN);
  for (int t = 0; t < 64; ++t)
    for (int k1 = 0; k1 < 8; ++k1) slab[wf5_x1(k1, t >> 3, t & 7)] = A[t][k1];
  for (int t = 0; t < 64; ++t) {  // pass B: (k1, n3) = (t & 7, t >> 3)
    for (int n2 = 0; n2 < 8; ++n2) B[t][n2] = slab[wf5_x1(t & 7, n2, t >> 3)];
    wf_dft8(B[t]);
    for (int k2 = 1; k2 < 8; ++k2) B[t][k2] = wf_cmul(B[t][k2], tw1024(16 * (t >> 3) * k2));  // W_64^{n3 k2}
  }
  for (int i = 0; i < WF5_SLAB; ++i) slab[i] = wf_make(NAN, NAN);
  for (int t = 0; t < 64; ++t)
    for (int k2 = 0; k2 < 8; ++k2) slab[wf5_x2(k2, t >> 3, t & 7)] = B[t][k2];
  static wf2 Z[64][8];
  for (int t = 0; t < 64; ++t) {  // pass C: c = t = k1 + 8 k2
    wf2 e[4], o[4];
    for (int n3 = 0; n3 < 8; ++n3) ((n3 & 1) ? o : e)[n3 >> 1] = slab[wf5_x2(t >> 3, n3, t & 7)];
    for (int i = 0; i < 8; ++i) Z[t][i] = wf_make(NAN, NAN);
    wf_dft8_pruned<2>(e, o, Z[t]);
  }
  double worst = 0.0, scale = 0.0;
  for (int t = 0; t < 64; ++t) {
    const int p = (64 - t) & 63;
    const wf2 zp7 = Z[p][7], zp6 = Z[p][6];  // the lane permute
    const wf2 m0 = t == 0 ? Z[t][0] : zp7, m1 = t == 0 ? zp7 : zp6;
    const wf2 w0 = tw1024(t), w1 = tw1024(t + 64);
    const wf2 X[2] = {wf_unpack(Z[t][0], m0, w0), wf_unpack(Z[t][1], m1, w1)};
    for (int k3 = 0; k3 < 2; ++k3) {
      const int k = t + 64 * k3;
      double re = 0, im = 0;
      for (int n = 0; n < 1024; ++n) {
        const double a = -2.0 * M_PI * (double)((k * n) % 1024) / 1024.0;
        re += row[n] * cos(a);
        im += row[n] * sin(a);
      }
      const double e = sqrt((X[k3].x - re) * (X[k3].x - re) + (X[k3].y - im) * (X[k3].y - im));
      if (!(e <= worst)) worst = e;
      scale = fmax(scale, sqrt(re * re + im * im));
    }
  }
  printf("512-point wave transform: relative error %.3g\n", worst / scale);
  const int bad = !(worst / scale < 2e-6);
  printf(bad ? "FAIL\n" : "OK\n");
  return bad;
}
