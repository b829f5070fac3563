// Workgroup-cooperative power-of-two complex FFT for gfx950.
//
// One 256-thread workgroup (4 wave64) transforms one length-N line that lives in an
// LDS buffer of interleaved complex values.  The algorithm is Stockham autosort
// (natural order in, natural order out) with radix-8 passes and one trailing
// radix-4/2 pass; each pass keeps its butterflies in registers between the
// "read everything" and "write everything" halves so a single LDS line suffices.
// The first pass pulls its inputs through a caller-supplied functor (fused
// prologue: global load, normalise, mask, conj-multiply ...) and the last pass
// pushes its outputs through a caller-supplied functor (fused epilogue: filter,
// store, arg-max ...), so neither end of a transform needs an extra LDS round trip.
//
// LDS layout: cfloat line[lds_len(N)], element i at lpad(i) = i + (i >> 4)
// (one pad element per 16: the strided writes of a radix-8 pass then fall on
// distinct banks for ds_write_b64's 16-lane groups).
#pragma once
#include <type_traits>
#include "mc_common.h"

__device__ __forceinline__ int lpad(int i) { return i + (i >> 4); }
__host__ __device__ constexpr int lds_len(int n) { return n + (n >> 4) + 1; }

#define MC_SQRT1_2 0.70710678118654752440f

template <int DIR>
__device__ __forceinline__ void bfly2(cfloat* a) {
  cfloat t = a[0];
  a[0] = cadd(t, a[1]);
  a[1] = csub(t, a[1]);
}

template <int DIR>
__device__ __forceinline__ void bfly4(cfloat& a0, cfloat& a1, cfloat& a2, cfloat& a3) {
  cfloat t0 = cadd(a0, a2), t1 = csub(a0, a2);
  cfloat t2 = cadd(a1, a3), t3 = cmul_i<DIR>(csub(a1, a3));
  a0 = cadd(t0, t2);
  a1 = cadd(t1, t3);
  a2 = csub(t0, t2);
  a3 = csub(t1, t3);
}

template <int DIR>
__device__ __forceinline__ void bfly8(cfloat* a) {
  bfly4<DIR>(a[0], a[2], a[4], a[6]);  // E[k] in a0,a2,a4,a6
  bfly4<DIR>(a[1], a[3], a[5], a[7]);  // O[k] in a1,a3,a5,a7
  const float c = MC_SQRT1_2;
  cfloat o1, o2, o3;
  if (DIR < 0) {
    o1 = cmake(c * (a[3].x + a[3].y), c * (a[3].y - a[3].x));
    o3 = cmake(c * (a[7].y - a[7].x), -c * (a[7].x + a[7].y));
  } else {
    o1 = cmake(c * (a[3].x - a[3].y), c * (a[3].x + a[3].y));
    o3 = cmake(-c * (a[7].x + a[7].y), c * (a[7].x - a[7].y));
  }
  o2 = cmul_i<DIR>(a[5]);
  cfloat e0 = a[0], e1 = a[2], e2 = a[4], e3 = a[6], o0 = a[1];
  a[0] = cadd(e0, o0);
  a[4] = csub(e0, o0);
  a[1] = cadd(e1, o1);
  a[5] = csub(e1, o1);
  a[2] = cadd(e2, o2);
  a[6] = csub(e2, o2);
  a[3] = cadd(e3, o3);
  a[7] = csub(e3, o3);
}

template <int R, int DIR>
__device__ __forceinline__ void bfly(cfloat* a) {
  if constexpr (R == 8) bfly8<DIR>(a);
  else if constexpr (R == 4) bfly4<DIR>(a[0], a[1], a[2], a[3]);
  else bfly2<DIR>(a);
}

// ---------------------------------------------------------------------------------
// Pass plan of a length-N transform: radix 8 while possible, then one 4 or 2.
template <int N>
struct FftPlan {
  static constexpr int npass() {
    int n = N, p = 0;
    while (n > 1) {
      n /= (n >= 8 ? 8 : n);
      ++p;
    }
    return p;
  }
  static constexpr int radix(int pass) {
    int n = N, r = 1;
    for (int p = 0; p <= pass; ++p) {
      r = n >= 8 ? 8 : n;
      n /= r;
    }
    return r;
  }
  static constexpr int ns(int pass) {  // product of the radices before `pass`
    int n = 1;
    for (int p = 0; p < pass; ++p) n *= radix(p);
    return n;
  }
  static constexpr int iters(int pass) { return (N / radix(pass) + MC_WG - 1) / MC_WG; }
  static constexpr int max_iters() {
    int m = 1;
    for (int p = 0; p < npass(); ++p) m = iters(p) > m ? iters(p) : m;
    return m;
  }
};

// Threads that cooperate on one length-N transform in the ping-pong kernels: N/8 (one
// radix-8 butterfly each), at least one wavefront, at most the workgroup.  A workgroup
// then runs MC_WG / fft_threads(N) transforms side by side (patch rows: 512 points ->
// one wavefront per row, four rows at a time).
__host__ __device__ constexpr int fft_threads(int n) {
  return (n / 8) < 64 ? 64 : ((n / 8) > MC_WG ? MC_WG : (n / 8));
}

// Per-thread twiddle bases: the same for every line a thread transforms, so kernels
// that loop over many lines load them once.  w[pass][it] = exp(-+2 pi i k / (NS*R)).
template <int N>
struct FftTwiddles {
  static constexpr int P = FftPlan<N>::npass();
  static constexpr int NT = fft_threads(N);
  static constexpr int max_iters() {
    int m = 1;
    for (int p = 0; p < P; ++p) {
      const int it = (N / FftPlan<N>::radix(p) + NT - 1) / NT;
      m = it > m ? it : m;
    }
    return m;
  }
  cfloat w[P > 1 ? P - 1 : 1][max_iters()];
  // lt = thread index within the transform's sub-group, 0 <= lt < NT
  template <int DIR>
  __device__ __forceinline__ void init(int lt, const cfloat* __restrict__ tw, int tw_stride) {
#pragma unroll
    for (int p = 1; p < P; ++p) {
      const int R = FftPlan<N>::radix(p), NS = FftPlan<N>::ns(p), NB = N / R;
      const int IT = (NB + NT - 1) / NT;
#pragma unroll
      for (int it = 0; it < IT; ++it) {
        const int j = lt + it * NT;
        const int k = j & (NS - 1);
        cfloat v = tw[(j < NB ? k : 0) * (N / (NS * R)) * tw_stride];
        if (DIR > 0) v.y = -v.y;
        w[p - 1][it] = v;
      }
    }
  }
};

// One Stockham pass (index PASS of the plan) by the NT threads of a sub-group.
template <int N, int PASS, int DIR, typename Load, typename Store>
__device__ __forceinline__ void fft_pass2(int lt, const FftTwiddles<N>& T, Load load, Store store) {
  constexpr int R = FftPlan<N>::radix(PASS);
  constexpr int NS = FftPlan<N>::ns(PASS);
  constexpr int NB = N / R;
  constexpr int NT = fft_threads(N);
  constexpr int IT = (NB + NT - 1) / NT;
  cfloat v[IT][R];
#pragma unroll
  for (int it = 0; it < IT; ++it) {
    const int j = lt + it * NT;
    if (NB % NT == 0 || j < NB) {
#pragma unroll
      for (int m = 0; m < R; ++m) v[it][m] = load(j + m * NB, it, m);
      if constexpr (NS > 1) {
        const cfloat w1 = T.w[PASS - 1][it];
        v[it][1] = cmul(v[it][1], w1);
        if constexpr (R >= 4) {
          const cfloat w2 = cmul(w1, w1), w3 = cmul(w2, w1);
          v[it][2] = cmul(v[it][2], w2);
          v[it][3] = cmul(v[it][3], w3);
          if constexpr (R == 8) {
            const cfloat w4 = cmul(w2, w2), w5 = cmul(w4, w1), w6 = cmul(w3, w3), w7 = cmul(w4, w3);
            v[it][4] = cmul(v[it][4], w4);
            v[it][5] = cmul(v[it][5], w5);
            v[it][6] = cmul(v[it][6], w6);
            v[it][7] = cmul(v[it][7], w7);
          }
        }
      }
      bfly<R, DIR>(v[it]);
      const int k = j & (NS - 1);
      const int base = (j - k) * R + k;
#pragma unroll
      for (int m = 0; m < R; ++m) store(base + m * NS, v[it][m]);
    }
  }
}

// Ping-pong transform over two LDS lines (l0, l1: this sub-group's own pair): pass p reads
// line (s+p-1)&1 (p>0) and writes line (s+p)&1; one workgroup barrier after every
// LDS-writing pass, none between a pass's reads and writes.  Every sub-group of the
// workgroup must run the same sequence (the barriers are workgroup-wide).  The first pass
// reads through `load(i, it, m)` (it, m = position of element i in this thread's
// first-pass registers); the last pass writes through `store(i, v)` when
// LAST_TO_FUNCTOR, else to LDS.  Returns the index of the line that was read or written
// last; the next transform of a loop must start at (that ^ 1) -- see xc_fft.hip.
template <int N, int DIR, bool LAST_TO_FUNCTOR, int PASS, typename Load, typename Store>
__device__ __forceinline__ int fft_pp_rec(cfloat* l0, cfloat* l1, int s, int lt,
                                          const FftTwiddles<N>& T, Load load, Store store) {
  constexpr int P = FftPlan<N>::npass();
  cfloat* src = ((s + PASS + 1) & 1) ? l1 : l0;
  cfloat* dst = ((s + PASS) & 1) ? l1 : l0;
  auto lds_load = [src](int i, int, int) { return src[lpad(i)]; };
  auto lds_store = [dst](int i, cfloat v) { dst[lpad(i)] = v; };
  constexpr bool LAST = (PASS == P - 1);
  if constexpr (PASS == 0 && LAST) {
    if constexpr (LAST_TO_FUNCTOR) fft_pass2<N, PASS, DIR>(lt, T, load, store);
    else { fft_pass2<N, PASS, DIR>(lt, T, load, lds_store); __syncthreads(); }
    return (s + PASS) & 1;
  } else if constexpr (PASS == 0) {
    fft_pass2<N, PASS, DIR>(lt, T, load, lds_store);
    __syncthreads();
    return fft_pp_rec<N, DIR, LAST_TO_FUNCTOR, PASS + 1>(l0, l1, s, lt, T, load, store);
  } else if constexpr (LAST) {
    if constexpr (LAST_TO_FUNCTOR) {
      fft_pass2<N, PASS, DIR>(lt, T, lds_load, store);
      return (s + PASS + 1) & 1;
    } else {
      fft_pass2<N, PASS, DIR>(lt, T, lds_load, lds_store);
      __syncthreads();
      return (s + PASS) & 1;
    }
  } else {
    fft_pass2<N, PASS, DIR>(lt, T, lds_load, lds_store);
    __syncthreads();
    return fft_pp_rec<N, DIR, LAST_TO_FUNCTOR, PASS + 1>(l0, l1, s, lt, T, load, store);
  }
}

template <int N, int DIR, bool LAST_TO_FUNCTOR, typename Load, typename Store>
__device__ __forceinline__ int wg_fft_pp(cfloat* l0, cfloat* l1, int s, int lt,
                                         const FftTwiddles<N>& T, Load load, Store store) {
  return fft_pp_rec<N, DIR, LAST_TO_FUNCTOR, 0>(l0, l1, s, lt, T, load, store);
}

// One Stockham pass of radix R at sub-transform length NS (product of earlier
// radices).  tw = table of exp(-2*pi*i*k/L), L = N * tw_stride.
template <int N, int R, int NS, int DIR, bool SYNC_MID, bool SYNC_END, typename Load, typename Store>
__device__ __forceinline__ void fft_pass(int tid, const cfloat* __restrict__ tw, int tw_stride,
                                         Load load, Store store) {
  constexpr int NB = N / R;
  constexpr int IT = (NB + MC_WG - 1) / MC_WG;
  cfloat v[IT][R];
#pragma unroll
  for (int it = 0; it < IT; ++it) {
    const int j = tid + it * MC_WG;
    if (NB >= MC_WG || j < NB) {
#pragma unroll
      for (int m = 0; m < R; ++m) v[it][m] = load(j + m * NB);
      if constexpr (NS > 1) {
        const int k = j & (NS - 1);
        cfloat w1 = tw[k * (N / (NS * R)) * tw_stride];
        if (DIR > 0) w1.y = -w1.y;
        v[it][1] = cmul(v[it][1], w1);
        if constexpr (R >= 4) {
          cfloat w2 = cmul(w1, w1), w3 = cmul(w2, w1);
          v[it][2] = cmul(v[it][2], w2);
          v[it][3] = cmul(v[it][3], w3);
          if constexpr (R == 8) {
            cfloat w4 = cmul(w2, w2), w5 = cmul(w4, w1), w6 = cmul(w3, w3), w7 = cmul(w4, w3);
            v[it][4] = cmul(v[it][4], w4);
            v[it][5] = cmul(v[it][5], w5);
            v[it][6] = cmul(v[it][6], w6);
            v[it][7] = cmul(v[it][7], w7);
          }
        }
      }
      bfly<R, DIR>(v[it]);
    }
  }
  if (SYNC_MID) __syncthreads();
#pragma unroll
  for (int it = 0; it < IT; ++it) {
    const int j = tid + it * MC_WG;
    if (NB >= MC_WG || j < NB) {
      const int k = j & (NS - 1);
      const int base = (j - k) * R + k;
#pragma unroll
      for (int m = 0; m < R; ++m) store(base + m * NS, v[it][m]);
    }
  }
  if (SYNC_END) __syncthreads();
}

// FIRST_LDS: the first pass's `load` reads the line itself, so its reads must be fenced
// from its writes like any later pass.
template <int N, int NS, int DIR, bool FIRST, bool FIRST_LDS = false, typename Load, typename Store>
__device__ __forceinline__ void fft_rec(cfloat* line, int tid, const cfloat* __restrict__ tw,
                                        int tw_stride, Load load, Store store) {
  constexpr int REM = N / NS;
  constexpr int R = REM >= 8 ? 8 : REM;
  constexpr bool LAST = (NS * R == N);
  auto lds_load = [line](int i) { return line[lpad(i)]; };
  auto lds_store = [line](int i, cfloat v) { line[lpad(i)] = v; };
  if constexpr (FIRST && LAST) {
    fft_pass<N, R, NS, DIR, FIRST_LDS, false>(tid, tw, tw_stride, load, store);
  } else if constexpr (FIRST) {
    fft_pass<N, R, NS, DIR, FIRST_LDS, true>(tid, tw, tw_stride, load, lds_store);
    fft_rec<N, NS * R, DIR, false>(line, tid, tw, tw_stride, load, store);
  } else if constexpr (LAST) {
    fft_pass<N, R, NS, DIR, true, false>(tid, tw, tw_stride, lds_load, store);
  } else {
    fft_pass<N, R, NS, DIR, true, true>(tid, tw, tw_stride, lds_load, lds_store);
    fft_rec<N, NS * R, DIR, false>(line, tid, tw, tw_stride, load, store);
  }
}

// Length-N transform by the whole workgroup.  Preconditions: every thread of the
// 256-thread workgroup calls it; nobody still reads `line` from an earlier use
// (caller barriers).  DIR=-1 forward (exp(-i..)), DIR=+1 inverse, unscaled.
// `load(i)` supplies input element i (each i exactly once, by some thread);
// `store(i, v)` receives output element i.  If `store` writes `line` itself the
// caller must barrier before reading it.
template <int N, int DIR, typename Load, typename Store>
__device__ __forceinline__ void wg_fft(cfloat* line, int tid, const cfloat* __restrict__ tw,
                                       int tw_stride, Load load, Store store) {
  fft_rec<N, 1, DIR, true>(line, tid, tw, tw_stride, load, store);
}

// Same, but the inputs already sit in `line` (natural order): load(i) must read line[lpad(i)].
template <int N, int DIR, typename Load, typename Store>
__device__ __forceinline__ void wg_fft_inplace(cfloat* line, int tid, const cfloat* __restrict__ tw,
                                               int tw_stride, Load load, Store store) {
  fft_rec<N, 1, DIR, true, true>(line, tid, tw, tw_stride, load, store);
}

// =====================================================================================
// Lengths N = 2^a 3^b 5^c 7^d 11^e 13^f 31^g ("smooth": 2880 and 5760 = half the 5760- and
// 11520-column rows of K3 detectors, 4092 and 8184 = their row counts, 5120 and 10240 as chirp-z
// lengths): the same one-line Stockham scheme as above with odd-radix passes and sub-transform
// lengths that are no longer powers of two (k = j mod NS instead of a mask).  Radices: the primes
// above 5 first (largest first), then 8 while N allows, then 4, 9, 10, 5, 3, 2.
// tw = exp(-2 pi i k / (N tw_stride)).
// =====================================================================================
template <int DIR>
__device__ __forceinline__ void bfly3(cfloat* a) {
  const float s = 0.86602540378443864676f;
  const cfloat t1 = cadd(a[1], a[2]);
  const cfloat t2 = cmake(a[0].x - 0.5f * t1.x, a[0].y - 0.5f * t1.y);
  const cfloat d = cmul_i<DIR>(cscale(csub(a[1], a[2]), s));  // (-/+ i) s (a1 - a2)
  a[0] = cadd(a[0], t1);
  a[1] = cadd(t2, d);
  a[2] = csub(t2, d);
}

template <int DIR>
__device__ __forceinline__ void bfly5(cfloat* a) {
  const float c1 = 0.30901699437494742410f, c2 = -0.80901699437494742410f;
  const float s1 = 0.95105651629515357212f, s2 = 0.58778525229247312917f;
  const cfloat a1 = cadd(a[1], a[4]), a2 = cadd(a[2], a[3]);
  const cfloat b1 = csub(a[1], a[4]), b2 = csub(a[2], a[3]);
  const cfloat m1 = cmake(a[0].x + c1 * a1.x + c2 * a2.x, a[0].y + c1 * a1.y + c2 * a2.y);
  const cfloat m2 = cmake(a[0].x + c2 * a1.x + c1 * a2.x, a[0].y + c2 * a1.y + c1 * a2.y);
  const cfloat n1 = cmul_i<DIR>(cmake(s1 * b1.x + s2 * b2.x, s1 * b1.y + s2 * b2.y));
  const cfloat n2 = cmul_i<DIR>(cmake(s2 * b1.x - s1 * b2.x, s2 * b1.y - s1 * b2.y));
  a[0] = cadd(a[0], cadd(a1, a2));
  a[1] = cadd(m1, n1);
  a[4] = csub(m1, n1);
  a[2] = cadd(m2, n2);
  a[3] = csub(m2, n2);
}

// cos / sin (2 pi j / R) at compile time (double-precision Taylor series on an argument reduced to
// [-pi, pi]; 34 terms: the remainder is below 1e-17)
constexpr double mc_cx_sincos(double x, bool want_sin) {
  double term = want_sin ? x : 1.0, sum = term;
  for (int k = want_sin ? 3 : 2; k < 70; k += 2) {
    term *= -x * x / (double)((k - 1) * k);
    sum += term;
  }
  return sum;
}
template <int R>
struct PrimeTab {
  float c[R], s[R];
  constexpr PrimeTab() : c{}, s{} {
    for (int j = 0; j < R; ++j) {
      const int jj = j <= R / 2 ? j : j - R;  // angle in (-pi, pi]
      const double x = 6.283185307179586476925286766559 * (double)jj / (double)R;
      c[j] = (float)mc_cx_sincos(x, false);
      s[j] = (float)mc_cx_sincos(x, true);
    }
  }
};

// Butterfly of an odd prime radix R (7, 11, 13, 31: 4092 = 2^2 3 11 31 and 8184 = 2^3 3 11 31 are the
// row counts of K3 frames) as a direct DFT in its symmetric form: with s_m = a_m + a_{R-m},
// d_m = a_m - a_{R-m},
//   X_0 = a_0 + sum s_m,   X_{k}, X_{R-k} = (a_0 + sum_m cos(2 pi m k / R) s_m) -/+ DIR... i (sum_m sin(2 pi m k / R) d_m)
// i.e. (R-1)^2 / 2 real-times-complex products instead of (R-1)^2 complex ones; the cos / sin values
// are compile-time constants after unrolling.
template <int R, int DIR>
__device__ __forceinline__ void bfly_prime(cfloat* a) {
  constexpr int HF = (R - 1) / 2;
  constexpr PrimeTab<R> T{};
  cfloat s[HF], d[HF];
#pragma unroll
  for (int m = 1; m <= HF; ++m) {
    s[m - 1] = cadd(a[m], a[R - m]);
    d[m - 1] = csub(a[m], a[R - m]);
  }
  const cfloat x0 = a[0];
  cfloat tot = x0;
#pragma unroll
  for (int m = 0; m < HF; ++m) tot = cadd(tot, s[m]);
  a[0] = tot;
#pragma unroll
  for (int k = 1; k <= HF; ++k) {
    cfloat p = x0, q = cmake(0.f, 0.f);
#pragma unroll
    for (int m = 1; m <= HF; ++m) {
      const float c = T.c[(m * k) % R], sn = T.s[(m * k) % R];
      p.x = __builtin_fmaf(c, s[m - 1].x, p.x);
      p.y = __builtin_fmaf(c, s[m - 1].y, p.y);
      q.x = __builtin_fmaf(sn, d[m - 1].x, q.x);
      q.y = __builtin_fmaf(sn, d[m - 1].y, q.y);
    }
    // sum_n a_n exp(DIR 2 pi i n k / R) = p + DIR i q
    const cfloat iq = cmul_i<DIR>(q);
    a[k] = cadd(p, iq);
    a[R - k] = csub(p, iq);
  }
}

// 9 = 3 x 3 (Cooley-Tukey inside the registers: three radix-3 butterflies over n1, the three twiddles
// W_9^{n2 k1}, three radix-3 butterflies over n2) and 10 = 2 x 5 (prime-factor map, no twiddles):
// 2880 = 8 8 9 5 and 5760 = 8 8 9 10 take four passes instead of five / six.
template <int DIR>
__device__ __forceinline__ void bfly9(cfloat* a) {
  // W_9^m = exp(DIR 2 pi i m / 9)
  const float c1 = 0.76604444311897803520f, s1 = 0.64278760968653932632f;   // m = 1
  const float c2 = 0.17364817766693034885f, s2 = 0.98480775301220805937f;   // m = 2
  const float c4 = -0.93969262078590838405f, s4 = 0.34202014332566873304f;  // m = 4
  cfloat t[3][3];  // [n2][k1]
#pragma unroll
  for (int n2 = 0; n2 < 3; ++n2) {
    cfloat u[3] = {a[n2], a[3 + n2], a[6 + n2]};
    bfly3<DIR>(u);
    t[n2][0] = u[0]; t[n2][1] = u[1]; t[n2][2] = u[2];
  }
  const float sg = DIR < 0 ? -1.f : 1.f;
  t[1][1] = cmul(t[1][1], cmake(c1, sg * s1));
  t[1][2] = cmul(t[1][2], cmake(c2, sg * s2));
  t[2][1] = cmul(t[2][1], cmake(c2, sg * s2));
  t[2][2] = cmul(t[2][2], cmake(c4, sg * s4));
#pragma unroll
  for (int k1 = 0; k1 < 3; ++k1) {
    cfloat u[3] = {t[0][k1], t[1][k1], t[2][k1]};
    bfly3<DIR>(u);
    a[k1] = u[0]; a[k1 + 3] = u[1]; a[k1 + 6] = u[2];
  }
}

template <int DIR>
__device__ __forceinline__ void bfly10(cfloat* a) {
  // n = (5 n1 + 2 n2) mod 10, k = (5 k1 + 6 k2) mod 10
  cfloat y0[5] = {a[0], a[2], a[4], a[6], a[8]};   // n1 = 0: n = 2 n2
  cfloat y1[5] = {a[5], a[7], a[9], a[1], a[3]};   // n1 = 1: n = 5 + 2 n2 mod 10
  bfly5<DIR>(y0);
  bfly5<DIR>(y1);
#pragma unroll
  for (int k2 = 0; k2 < 5; ++k2) {
    a[(6 * k2) % 10] = cadd(y0[k2], y1[k2]);      // k1 = 0
    a[(5 + 6 * k2) % 10] = csub(y0[k2], y1[k2]);  // k1 = 1
  }
}

// 12 = 3 x 4 and 24 = 3 x 8 by the prime-factor map (no twiddles): the tails of 4092 = 31 11 12 and
// 8184 = 31 11 24, three passes per column instead of four.
//   12: n = (4 n1 + 3 n2) mod 12, k = (4 k1 + 9 k2) mod 12;  24: n = (8 n1 + 3 n2) mod 24, k = (16 k1 + 9 k2) mod 24
template <int R2, int DIR>
__device__ __forceinline__ void bfly_pfa3(cfloat* a) {
  constexpr int R = 3 * R2;                    // R2 = 4 or 8
  constexpr int KA = R2 == 4 ? 4 : 16;         // k = (KA k1 + 9 k2) mod R
  cfloat y[3][R2];                             // [n1][n2] -> after the R2-point transforms [n1][k2]
#pragma unroll
  for (int n1 = 0; n1 < 3; ++n1) {
#pragma unroll
    for (int n2 = 0; n2 < R2; ++n2) y[n1][n2] = a[(R2 * n1 + 3 * n2) % R];
    bfly<R2, DIR>(y[n1]);
  }
#pragma unroll
  for (int k2 = 0; k2 < R2; ++k2) {
    cfloat u[3] = {y[0][k2], y[1][k2], y[2][k2]};
    bfly3<DIR>(u);
#pragma unroll
    for (int k1 = 0; k1 < 3; ++k1) a[(KA * k1 + 9 * k2) % R] = u[k1];
  }
}

template <int R, int DIR>
__device__ __forceinline__ void bfly_any(cfloat* a) {
  if constexpr (R == 12) bfly_pfa3<4, DIR>(a);
  else if constexpr (R == 24) bfly_pfa3<8, DIR>(a);
  else if constexpr (R == 9) bfly9<DIR>(a);
  else if constexpr (R == 10) bfly10<DIR>(a);
  else if constexpr (R == 5) bfly5<DIR>(a);
  else if constexpr (R == 3) bfly3<DIR>(a);
  else if constexpr (R == 7 || R == 11 || R == 13 || R == 31) bfly_prime<R, DIR>(a);
  else bfly<R, DIR>(a);
}

// radix of the next pass: the large primes first (the first pass has no twiddles), then 8, 4, 5, 3, 2
__host__ __device__ constexpr int smooth_radix(int rem) {
  return rem % 31 == 0 ? 31 : rem % 13 == 0 ? 13 : rem % 11 == 0 ? 11 : rem % 7 == 0 ? 7
       : rem == 24 ? 24 : rem == 12 ? 12 : rem % 8 == 0 ? 8
       : rem % 4 == 0 ? 4 : rem % 9 == 0 ? 9 : rem % 10 == 0 ? 10 : rem % 5 == 0 ? 5 : rem % 3 == 0 ? 3
       : rem % 2 == 0 ? 2 : 0;
}
__host__ __device__ constexpr bool is_smooth(int n) {
  while (n % 2 == 0) n /= 2;
  while (n % 3 == 0) n /= 3;
  while (n % 5 == 0) n /= 5;
  while (n % 7 == 0) n /= 7;
  while (n % 11 == 0) n /= 11;
  while (n % 13 == 0) n /= 13;
  while (n % 31 == 0) n /= 31;
  return n == 1;
}

template <int N, int R, int NS, int DIR, bool SYNC_MID, bool SYNC_END, int WG = MC_WG, typename Load, typename Store>
__device__ __forceinline__ void smooth_pass(int tid, const cfloat* __restrict__ tw, int tw_stride,
                                            Load load, Store store) {
  constexpr int NB = N / R;
  constexpr int IT = (NB + WG - 1) / WG;
  cfloat v[IT][R];
#pragma unroll
  for (int it = 0; it < IT; ++it) {
    const int j = tid + it * WG;
    if (j < NB) {
#pragma unroll
      for (int m = 0; m < R; ++m) v[it][m] = load(j + m * NB);
      if constexpr (NS > 1) {
        const int k = j % NS;
        if constexpr (R > 8) {
          // many powers: every twiddle from the table (k m < NS R, so the index stays below N)
#pragma unroll
          for (int m = 1; m < R; ++m) {
            cfloat wm = tw[k * m * (N / (NS * R)) * tw_stride];
            if (DIR > 0) wm.y = -wm.y;
            v[it][m] = cmul(v[it][m], wm);
          }
        } else {
          cfloat w1 = tw[k * (N / (NS * R)) * tw_stride];
          if (DIR > 0) w1.y = -w1.y;
          cfloat wm = w1;
#pragma unroll
          for (int m = 1; m < R; ++m) {
            v[it][m] = cmul(v[it][m], wm);
            if (m + 1 < R) wm = cmul(wm, w1);
          }
        }
      }
      bfly_any<R, DIR>(v[it]);
    }
  }
  if (SYNC_MID) __syncthreads();
#pragma unroll
  for (int it = 0; it < IT; ++it) {
    const int j = tid + it * WG;
    if (j < NB) {
      const int k = j % NS;
      const int base = (j - k) * R + k;
#pragma unroll
      for (int m = 0; m < R; ++m) {
        // a store functor may ask for the compile-time position (iteration, output) of the value
        // among this thread's outputs of the pass: register accumulators indexed without scratch
        if constexpr (std::is_invocable_v<Store, int, cfloat, int>) store(base + m * NS, v[it][m], it * R + m);
        else store(base + m * NS, v[it][m]);
      }
    }
  }
  if (SYNC_END) __syncthreads();
}

template <int N, int NS, int DIR, bool FIRST, bool FIRST_LDS = false, int WG = MC_WG, typename Load, typename Store>
__device__ __forceinline__ void smooth_rec(cfloat* line, int tid, const cfloat* __restrict__ tw,
                                           int tw_stride, Load load, Store store) {
  constexpr int R = smooth_radix(N / NS);
  static_assert(R > 0, "length is not of the form 2^a 3^b 5^c");
  constexpr bool LAST = (NS * R == N);
  auto lds_load = [line](int i) { return line[lpad(i)]; };
  auto lds_store = [line](int i, cfloat v) { line[lpad(i)] = v; };
  if constexpr (FIRST && LAST) {
    smooth_pass<N, R, NS, DIR, FIRST_LDS, false, WG>(tid, tw, tw_stride, load, store);
  } else if constexpr (FIRST) {
    smooth_pass<N, R, NS, DIR, FIRST_LDS, true, WG>(tid, tw, tw_stride, load, lds_store);
    smooth_rec<N, NS * R, DIR, false, false, WG>(line, tid, tw, tw_stride, load, store);
  } else if constexpr (LAST) {
    smooth_pass<N, R, NS, DIR, true, false, WG>(tid, tw, tw_stride, lds_load, store);
  } else {
    smooth_pass<N, R, NS, DIR, true, true, WG>(tid, tw, tw_stride, lds_load, lds_store);
    smooth_rec<N, NS * R, DIR, false, false, WG>(line, tid, tw, tw_stride, load, store);
  }
}

// Length-N transform (any N = 2^a 3^b 5^c; powers of two take the plan-driven passes above) by
// the whole workgroup; contract as wg_fft.
// WG: threads of the workgroup (mixed-radix lengths only; 512 for the 8184-point columns, whose
// radix-31 pass has 264 butterflies).
template <int N, int DIR, int WG = MC_WG, typename Load, typename Store>
__device__ __forceinline__ void wg_fft_any(cfloat* line, int tid, const cfloat* __restrict__ tw,
                                           int tw_stride, Load load, Store store) {
  if constexpr ((N & (N - 1)) == 0) {
    static_assert(WG == MC_WG, "power-of-two lines run on 256 threads");
    fft_rec<N, 1, DIR, true>(line, tid, tw, tw_stride, load, store);
  } else {
    smooth_rec<N, 1, DIR, true, false, WG>(line, tid, tw, tw_stride, load, store);
  }
}
template <int N, int DIR, int WG = MC_WG, typename Load, typename Store>
__device__ __forceinline__ void wg_fft_any_inplace(cfloat* line, int tid, const cfloat* __restrict__ tw,
                                                   int tw_stride, Load load, Store store) {
  if constexpr ((N & (N - 1)) == 0) {
    static_assert(WG == MC_WG, "power-of-two lines run on 256 threads");
    fft_rec<N, 1, DIR, true, true>(line, tid, tw, tw_stride, load, store);
  } else {
    smooth_rec<N, 1, DIR, true, true, WG>(line, tid, tw, tw_stride, load, store);
  }
}

// Bluestein chirp-z: a length-n DFT (any n, 2n-1 <= M = power of two) of x as
//   X[k] = c[k] * ( (x .* c) (*) conj(c) )[k],   c[j] = exp(DIR * i*pi*j^2/n)
// with the convolution done by two length-M transforms in `line`.  chirp = c for this
// direction (n entries), bspec = FFT_M of the wrapped conj(c) (M entries, already scaled
// by 1/M).  load(j) -> x[j] for j < n (each j once); store(k, X[k]) is called for every
// k < M, the caller keeps the k it needs (k < n).  Every thread of the workgroup must call
// it; `line` must be free on entry; on exit a barrier is still needed before reusing `line`.
// keep > 0: output-pruned form.  Only the outputs k in [0, keep) and (n - keep, n) are wanted, so
// the circular convolution only has to be exact there: M >= n + 2 keep - 1 is enough (instead
// of 2n - 1), provided bspec was built for that wrap (plan.line_plan(keep=...)): output k < keep
// sits at position k, output n - m (1 <= m < keep) at position M - m.
template <int M, typename Load, typename Store>
__device__ __forceinline__ void wg_bluestein(cfloat* line, int tid, const cfloat* __restrict__ tw_m,
                                             const cfloat* __restrict__ chirp,
                                             const cfloat* __restrict__ bspec, int n, Load load,
                                             Store store, int keep = 0) {
  auto in1 = [&](int j) { return j < n ? cmul(load(j), chirp[j]) : cmake(0.f, 0.f); };
  auto out1 = [&](int j, cfloat v) { line[lpad(j)] = cmul(v, bspec[j]); };
  wg_fft_any<M, -1>(line, tid, tw_m, 1, in1, out1);
  __syncthreads();
  auto in2 = [&](int j) { return line[lpad(j)]; };
  auto out2 = [&](int p, cfloat v) {
    int k = p;
    if (keep > 0) {
      if (p > M - keep) {
        k = p - M + n;
        if (n & 1) v = cmake(-v.x, -v.y);  // the convolution ran at offset k - n: chirp(k - n) = (-1)^n chirp(k)
      } else if (p >= keep) {
        return;
      }
    }
    if (k < n) store(k, cmul(v, chirp[k]));
  };
  wg_fft_any_inplace<M, +1>(line, tid, tw_m, 1, in2, out2);
}

// =====================================================================================
// 4096-point transform by one 256-thread workgroup, sixteen values per thread in registers:
// N = 16 x 16 x 16, three register-resident radix-16 passes, two exchanges through ONE
// 32 KiB LDS line (4 barriers per transform instead of the 7 of the radix-8 Stockham
// passes above).  n = 256 n1 + 16 n2 + n3,  k = k1 + 16 k2 + 256 k3:
//   pass A  thread q = 16 n2 + n3 = tid:  radix 16 over n1, twiddle W_4096^{q k1}
//   pass B  thread (k1, n3) = (tid >> 4, tid & 15): radix 16 over n2, twiddle W_256^{n3 k2}
//   pass C  thread c = k1 + 16 k2 = tid:  radix 16 over n3, output k = c + 256 k3
// LDS addresses (complex index), conflict-free for 8-byte writes (16-lane groups) and reads
// (32-lane groups):
//   exchange 1: (k1, q)       at k1 * 256 + (q ^ (16 * (k1 & 1)))
//   exchange 2: (k1, k2, n3)  at k1 * 256 + k2 * 16 + (n3 ^ k1)
// Pruning is resolved at compile time: with IN_KEEP < 8 only the inputs n1 in [0, IN_KEEP) and
// [16 - IN_KEEP, 16) are fetched (the others are literal zeros: band-limited spectra), with
// OUT_KEEP < 8 only the outputs k3 in [0, OUT_KEEP) and [16 - OUT_KEEP, 16) are produced;
// the dead butterfly arithmetic disappears.  DIR = +1 runs the forward kernel on conjugated
// data.  load(n1, n) -> input n = 256 n1 + tid (n1 a compile-time constant after unrolling);
// store(k, v) <- output k.  Every thread of
// the workgroup must call it; `line` (4096 entries) must be free on entry and is free again
// after a __syncthreads() on exit.  tw = exp(-2 pi i k / 4096), 4096 entries.
// =====================================================================================
#include "mc_wave_fft.h"

// The eight twiddle bases a thread needs (exact table entries; the other 22 are products): they depend on
// the thread index only, so a kernel that transforms many lines per workgroup loads them once
// (r16_twiddles) instead of waiting for eight global loads in the middle of every line.
struct R16Tw {
  wf2 a1, a2, a4, a8, b1, b2, b4, b8;
};
__device__ __forceinline__ R16Tw r16_twiddles(int tid, const cfloat* __restrict__ tw) {
  const int n3 = tid & 15;
  R16Tw t;
  t.a1 = wf_from(tw[tid]); t.a2 = wf_from(tw[2 * tid]); t.a4 = wf_from(tw[4 * tid]); t.a8 = wf_from(tw[8 * tid]);
  t.b1 = wf_from(tw[16 * n3]); t.b2 = wf_from(tw[32 * n3]); t.b4 = wf_from(tw[64 * n3]); t.b8 = wf_from(tw[128 * n3]);
  return t;
}

template <int DIR, int IN_KEEP, int OUT_KEEP, typename Load, typename Store>
__device__ __forceinline__ void wg_fft4096_r16_tw(cfloat* line_c, int tid, const R16Tw& TW, Load load, Store store);

template <int DIR, int IN_KEEP, int OUT_KEEP, typename Load, typename Store>
__device__ __forceinline__ void wg_fft4096_r16(cfloat* line_c, int tid, const cfloat* __restrict__ tw,
                                               Load load, Store store) {
  const R16Tw TW = r16_twiddles(tid, tw);
  wg_fft4096_r16_tw<DIR, IN_KEEP, OUT_KEEP>(line_c, tid, TW, load, store);
}

template <int DIR, int IN_KEEP, int OUT_KEEP, typename Load, typename Store>
__device__ __forceinline__ void wg_fft4096_r16_tw(cfloat* line_c, int tid, const R16Tw& TW, Load load, Store store) {
  wf2* line = reinterpret_cast<wf2*>(line_c);
  auto cj = [](wf2 v) { return DIR > 0 ? wf2{v.x, -v.y} : v; };
  wf2 a[16];
  // ---- pass A
#pragma unroll
  for (int n1 = 0; n1 < 16; ++n1) {
    if (n1 < IN_KEEP || n1 >= 16 - IN_KEEP) a[n1] = cj(wf_from(load(n1, 256 * n1 + tid)));
    else a[n1] = wf2{0.f, 0.f};
  }
  if constexpr (IN_KEEP == 2) wf_dft16_lo2(a);  // 12 of the 16 inputs are zero
  else wf_dft16(a);
  {
    const wf2 w1 = TW.a1, w2 = TW.a2, w4 = TW.a4, w8 = TW.a8;  // exact bases
    const wf2 w3 = wf_cmul(w2, w1), w5 = wf_cmul(w4, w1), w6 = wf_cmul(w4, w2), w7 = wf_cmul(w4, w3);
    a[1] = wf_cmul(a[1], w1); a[2] = wf_cmul(a[2], w2); a[3] = wf_cmul(a[3], w3); a[4] = wf_cmul(a[4], w4);
    a[5] = wf_cmul(a[5], w5); a[6] = wf_cmul(a[6], w6); a[7] = wf_cmul(a[7], w7); a[8] = wf_cmul(a[8], w8);
    a[9] = wf_cmul(a[9], wf_cmul(w8, w1)); a[10] = wf_cmul(a[10], wf_cmul(w8, w2));
    a[11] = wf_cmul(a[11], wf_cmul(w8, w3)); a[12] = wf_cmul(a[12], wf_cmul(w8, w4));
    a[13] = wf_cmul(a[13], wf_cmul(w8, w5)); a[14] = wf_cmul(a[14], wf_cmul(w8, w6));
    a[15] = wf_cmul(a[15], wf_cmul(w8, w7));
  }
#pragma unroll
  for (int k1 = 0; k1 < 16; ++k1) line[k1 * 256 + (tid ^ (16 * (k1 & 1)))] = a[k1];
  __syncthreads();
  // ---- pass B: (k1, n3) = (tid >> 4, tid & 15), entries (k1, q = 16 n2 + n3)
  {
    const int k1 = tid >> 4, n3 = tid & 15;
#pragma unroll
    for (int n2 = 0; n2 < 16; ++n2) a[n2] = line[k1 * 256 + ((16 * n2 + n3) ^ (16 * (k1 & 1)))];
    __syncthreads();  // everyone has read exchange 1 before exchange 2 overwrites the line
    wf_dft16(a);
    const wf2 w1 = TW.b1, w2 = TW.b2, w4 = TW.b4, w8 = TW.b8;
    const wf2 w3 = wf_cmul(w2, w1), w5 = wf_cmul(w4, w1), w6 = wf_cmul(w4, w2), w7 = wf_cmul(w4, w3);
    a[1] = wf_cmul(a[1], w1); a[2] = wf_cmul(a[2], w2); a[3] = wf_cmul(a[3], w3); a[4] = wf_cmul(a[4], w4);
    a[5] = wf_cmul(a[5], w5); a[6] = wf_cmul(a[6], w6); a[7] = wf_cmul(a[7], w7); a[8] = wf_cmul(a[8], w8);
    a[9] = wf_cmul(a[9], wf_cmul(w8, w1)); a[10] = wf_cmul(a[10], wf_cmul(w8, w2));
    a[11] = wf_cmul(a[11], wf_cmul(w8, w3)); a[12] = wf_cmul(a[12], wf_cmul(w8, w4));
    a[13] = wf_cmul(a[13], wf_cmul(w8, w5)); a[14] = wf_cmul(a[14], wf_cmul(w8, w6));
    a[15] = wf_cmul(a[15], wf_cmul(w8, w7));
#pragma unroll
    for (int k2 = 0; k2 < 16; ++k2) line[k1 * 256 + k2 * 16 + (n3 ^ k1)] = a[k2];
  }
  __syncthreads();
  // ---- pass C: c = k1 + 16 k2 = tid
  {
    const int k1 = tid & 15, k2 = tid >> 4;
#pragma unroll
    for (int n3 = 0; n3 < 16; ++n3) a[n3] = line[k1 * 256 + k2 * 16 + (n3 ^ k1)];
    wf_dft16(a);
#pragma unroll
    for (int k3 = 0; k3 < 16; ++k3)
      if (k3 < OUT_KEEP || k3 >= 16 - OUT_KEEP) {
        // a store that takes the (compile-time) butterfly output index first can keep per-output data in registers
        if constexpr (std::is_invocable_v<Store, int, int, cfloat>) store(k3, tid + 256 * k3, wf_to(cj(a[k3])));
        else store(tid + 256 * k3, wf_to(cj(a[k3])));
      }
  }
}
