// CPU check of the index algebra of csrc/mc_wave_fft.h (clang++ on the host, no GPU): the 64 lanes of
// a wavefront are executed one after the other, phase by phase, over a shared 1024-entry
// slab exactly as xc_rows_fwd_wave does, and the 4 * KEEP real-FFT bins every lane ends
// up with are compared with a double-precision DFT of the same real row.
//
//   /opt/rocm/lib/llvm/bin/clang++ -O1 -std=c++17 -I torch_motion_correction_amd/csrc tests/host_wave_fft.cpp -o /tmp/host_wave_fft -lm && /tmp/host_wave_fft
#include <stdio.h>
#include <stdlib.h>

#include <vector>

#include "mc_wave_fft.h"

static wf2 tw4096(int k) {
  const double a = -2.0 * M_PI * (double)k / 4096.0;
  return wf_make((float)cos(a), (float)sin(a));
}

template <int KEEP>
static double run(unsigned seed, int x0, int x1) {
  std::vector<float> row(4096);
  srand(seed);
  for (int i = 0; i < 4096; ++i) row[i] = (i >= x0 && i < x1) ? (float)(rand() % 65536) / 65536.f - 0.5f : 0.f;
  static wf2 A0[64][16], A1[64][16], B0[64][16], B1[64][16], Ce[64][4][4], Co[64][4][4];
  static wf2 slab[WF_SLAB];
  WfLane L[64];
  for (int t = 0; t < 64; ++t) L[t] = wf_lane(t);
  // pass A
  for (int t = 0; t < 64; ++t) {
    for (int n1 = 0; n1 < 16; ++n1) {
      const int x = 256 * n1 + 4 * t;
      A0[t][n1] = wf_make(row[x], row[x + 1]);
      A1[t][n1] = wf_make(row[x + 2], row[x + 3]);
    }
    wf_dft16(A0[t]);
    wf_dft16(A1[t]);
    // twA[s][q] = W_2048^{q 2^s} = tw4096(2 q 2^s): the four exact bases, the other eleven are products
    wf_twiddle16(A0[t], tw4096(2 * (2 * t)), tw4096(4 * (2 * t)), tw4096(8 * (2 * t)), tw4096(16 * (2 * t)));
    wf_twiddle16(A1[t], tw4096(2 * (2 * t + 1)), tw4096(4 * (2 * t + 1)), tw4096(8 * (2 * t + 1)), tw4096(16 * (2 * t + 1)));
  }
  // exchange 1, pass B
  for (int h = 0; h < 2; ++h) {
    for (int i = 0; i < WF_SLAB; ++i) slab[i] = wf_make(NAN, NAN);
    for (int t = 0; t < 64; ++t)
      for (int k1 = 0; k1 < 16; ++k1) slab[L[t].x1w_base + (k1 ^ L[t].x1w_mask)] = (h ? A1 : A0)[t][k1];
    for (int t = 0; t < 64; ++t) {
      wf2(&B)[16] = (h ? B1 : B0)[t];
      for (int n2 = 0; n2 < 16; ++n2) B[n2] = slab[L[t].x1r[n2 & 3] + 64 * n2];
      wf_dft16(B);
      // twB[g][k2 - 1][h] = W_128^{(2 g + h) k2} = tw4096(32 (2 g + h) k2)
      for (int k2 = 1; k2 < 16; ++k2) B[k2] = wf_cmul(B[k2], tw4096(32 * (2 * (t >> 4) + h) * k2));
    }
  }
  // exchange 2
  for (int h = 0; h < 2; ++h) {
    for (int i = 0; i < WF_SLAB; ++i) slab[i] = wf_make(NAN, NAN);
    for (int t = 0; t < 64; ++t)
      for (int k2 = 0; k2 < 16; ++k2) slab[L[t].x2w + 16 * k2] = (h ? B1 : B0)[t][k2];
    for (int t = 0; t < 64; ++t)
      for (int s = 0; s < 4; ++s)
        for (int n3h = 0; n3h < 4; ++n3h) (h ? Co : Ce)[t][s][n3h] = slab[L[t].x2r[s] + 256 * n3h];
  }
  // pass C, unpack, compare
  std::vector<double> re(512, 0.0), im(512, 0.0);
  for (int k = 0; k < 256 * KEEP; ++k)
    for (int n = x0; n < x1; ++n) {
      const double a = -2.0 * M_PI * (double)((int64_t)k * n % 4096) / 4096.0;
      re[k] += row[n] * cos(a);
      im[k] += row[n] * sin(a);
    }
  double worst = 0.0, scale = 0.0;
  std::vector<int> seen(512, 0);
  for (int t = 0; t < 64; ++t) {
    wf2 z[4][8], wk[4], X[4][KEEP];
    for (int s = 0; s < 4; ++s) {
      for (int i = 0; i < 8; ++i) z[s][i] = wf_make(NAN, NAN);
      wk[s] = tw4096(L[t].kbin[s]);
      wf_dft8_pruned<KEEP>(Ce[t][s], Co[t][s], z[s]);
    }
    wf_unpack_lane<KEEP>(z, wk, L[t].self != 0, X);
    for (int s = 0; s < 4; ++s)
      for (int k3 = 0; k3 < KEEP; ++k3) {
        const int k = L[t].kbin[s] + 256 * k3;
        seen[k]++;
        const double dr = X[s][k3].x - re[k], di = X[s][k3].y - im[k];
        const double e = sqrt(dr * dr + di * di);
        if (!(e <= worst)) worst = e;  // NaN-propagating
        scale = fmax(scale, sqrt(re[k] * re[k] + im[k] * im[k]));
      }
  }
  for (int k = 0; k < 256 * KEEP; ++k)
    if (seen[k] != 1) {
      printf("bin %d produced %d times\n", k, seen[k]);
      return 1e9;
    }
  return worst / scale;
}

// wf_dft16_lo2 (inputs 2..13 zero, the band-passed columns of the search) against wf_dft16 on the same data
static double run_lo2(unsigned seed) {
  srand(seed);
  double worst = 0.0, scale = 0.0;
  for (int trial = 0; trial < 64; ++trial) {
    wf2 a[16], b[16];
    for (int i = 0; i < 16; ++i) {
      const bool kept = i < 2 || i >= 14;
      a[i] = kept ? wf_make((float)(rand() % 65536) / 65536.f - 0.5f, (float)(rand() % 65536) / 65536.f - 0.5f)
                  : wf_make(0.f, 0.f);
      b[i] = a[i];
    }
    wf_dft16(a);
    wf_dft16_lo2(b);
    for (int i = 0; i < 16; ++i) {
      worst = fmax(worst, fmax(fabs((double)a[i].x - b[i].x), fabs((double)a[i].y - b[i].y)));
      scale = fmax(scale, fmax(fabs((double)a[i].x), fabs((double)a[i].y)));
    }
  }
  return worst / scale;
}

int main() {
  int bad = 0;
  const double e0 = run_lo2(7);
  printf("wf_dft16_lo2 against wf_dft16: %.3g\n", e0);
  if (!(e0 < 5e-7)) bad = 1;
  const double e1 = run<1>(1, 0, 4096), e2 = run<2>(2, 0, 4096), e3 = run<2>(3, 510, 3586);
  printf("relative error KEEP=1: %.3g  KEEP=2: %.3g  KEEP=2 (support 510..3586): %.3g\n", e1, e2, e3);
  if (!(e1 < 2e-6) || !(e2 < 2e-6) || !(e3 < 2e-6)) bad = 1;
  printf(bad ? "FAIL\n" : "OK\n");
  return bad;
}
