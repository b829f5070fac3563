// CPU check of the 1024-point wave column transform's index algebra (csrc/mc_wave_fft.h, third
// part): 64 lanes executed one after the other over a shared 1024-entry slab; every output is
// compared with a double-precision DFT of the same complex line.
//
//   /opt/rocm/lib/llvm/bin/clang++ -O1 -std=c++17 -I torch_motion_correction_amd/csrc tests/host_wave_fft1024.cpp -o /tmp/host_wave_fft1024 -lm && /tmp/host_wave_fft1024
#include <stdio.h>
#include <stdlib.h>

#include <vector>

#include "mc_wave_fft.h"

static wf2 tw1024(int k) {
  const double a = -2.0 * M_PI * (double)k / 1024.0;
  return wf_make((float)cos(a), (float)sin(a));
}

int main() {
  std::vector<wf2> x(1024);
  srand(11);
  for (int i = 0; i < 1024; ++i)
    x[i] = wf_make((float)(rand() % 65536) / 65536.f - 0.5f, (float)(rand() % 65536) / 65536.f - 0.5f);
  static wf2 A[64][16], B[64][2][8], slab[1024], X[1024];
  for (int t = 0; t < 64; ++t) {  // pass A: q = t
    for (int n1 = 0; n1 < 16; ++n1) A[t][n1] = x[64 * n1 + t];
    wf_dft16(A[t]);
    wf_twiddle16(A[t], tw1024(t), tw1024(2 * t), tw1024(4 * t), tw1024(8 * t));  // W_1024^{q k1}
  }
  for (int i = 0; i < 1024; ++i) slab[i] = wf_make(NAN, NAN);
  for (int t = 0; t < 64; ++t)
    for (int k1 = 0; k1 < 16; ++k1) slab[wf10_x1(k1, t)] = A[t][k1];
  for (int t = 0; t < 64; ++t)  // pass B: (k1, n3) = (t & 15, (t >> 4) + 4 b)
    for (int b = 0; b < 2; ++b) {
      const int k1 = t & 15, n3 = (t >> 4) + 4 * b;
      for (int n2 = 0; n2 < 8; ++n2) B[t][b][n2] = slab[wf10_x1(k1, 8 * n2 + n3)];
      wf_dft8(B[t][b]);
      wf_twiddle8(B[t][b], tw1024(16 * n3));  // W_64^{n3 k2}
    }
  for (int i = 0; i < 1024; ++i) slab[i] = wf_make(NAN, NAN);
  for (int t = 0; t < 64; ++t)
    for (int b = 0; b < 2; ++b)
      for (int k2 = 0; k2 < 8; ++k2) slab[wf10_x2(t & 15, k2, (t >> 4) + 4 * b)] = B[t][b][k2];
  for (int i = 0; i < 1024; ++i) X[i] = wf_make(NAN, NAN);
  for (int t = 0; t < 64; ++t)  // pass C: (k1, k2) = (t & 15, (t >> 4) + 4 b)
    for (int b = 0; b < 2; ++b) {
      const int k1 = t & 15, k2 = (t >> 4) + 4 * b;
      wf2 c[8];
      for (int n3 = 0; n3 < 8; ++n3) c[n3] = slab[wf10_x2(k1, k2, n3)];
      wf_dft8(c);
      for (int k3 = 0; k3 < 8; ++k3) X[k1 + 16 * k2 + 128 * k3] = c[k3];
    }
  double worst = 0.0, scale = 0.0;
  for (int k = 0; k < 1024; ++k) {
    double re = 0, im = 0;
    for (int n = 0; n < 1024; ++n) {
      const double a = -2.0 * M_PI * (double)((k * n) % 1024) / 1024.0;
      re += x[n].x * cos(a) - x[n].y * sin(a);
      im += x[n].x * sin(a) + x[n].y * cos(a);
    }
    const double e = sqrt((X[k].x - re) * (X[k].x - re) + (X[k].y - im) * (X[k].y - im));
    if (!(e <= worst)) worst = e;
    scale = fmax(scale, sqrt(re * re + im * im));
  }
  printf("1024-point wave column transform: relative error %.3g\n", worst / scale);
  const int bad = !(worst / scale < 2e-6);
  printf(bad ? "FAIL\n" : "OK\n");
  return bad;
}
