// libmcorr -- Fourier shift of frames too wide for one row line: x-polyphase form.
//
// correct_motion_fast (correct_motion.py:484-496) is  irfftn(rfftn(frame) * phi),
// phi = exp(-2 pi i (fy sy + fx sx)).  The FULL spectrum is needed, and the row kernels stage a
// line plus all kept bins in LDS: that stops at about 8190 columns.  A frame of W = 2 N2 columns is
// therefore split into its even and odd columns (two frames of N2 columns, which the existing
// kernels transform), and with E, O their spectra (kx' = 0..N2/2 stored) the bins of the full
// spectrum are the radix-2 butterfly
//     X[ky, k']      = E + w O          (w = exp(-2 pi i k' / W))
//     X[ky, k' + N2] = E - w O          (= the conjugate mirror of the stored bin (-ky, N2 - k'):
//                                         by the Hermitian symmetry of E and O it needs the SAME two
//                                         values E[ky,k'], O[ky,k'])
// Each is multiplied by phi at its own frequency -- k'/W for the first, (k' - N2)/W for the second,
// except k' = 0 where the second one is the Nyquist column itself and takes +1/2 as rfftfreq does,
// and the Nyquist row of a mirrored bin, which takes +1/2 too (checked against numpy's
// irfft2(rfft2(x) phi) to 1e-15 for even and odd heights) --
// and the inverse butterfly gives the spectra of the even / odd columns of the shifted frame:
//     YE = (YA + YB) / 2,   YO = conj(w) (YA - YB) / 2.
// So the whole step is pointwise on (E, O): one pass over the two spectra, in place.
// Layout: S[(job)][kx'][ky] complex as mc_xc_cols_forward writes it for the full geometry of an
// (H, N2) frame (all H rows kept, natural order); job j = even columns of frame j, job j + n = odd.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "mc_common.h"
#include "mcorr.h"

__global__ __launch_bounds__(256) void polyphase_shift_kernel(float2* __restrict__ S, const float* __restrict__ shifts,
                                                              int n, int nkx, int H, int W) {
  const int j = blockIdx.z, kx = blockIdx.y;
  const int ky = blockIdx.x * 256 + threadIdx.x;
  if (ky >= H) return;
  const int64_t plane = (int64_t)nkx * H;
  float2* pe = S + (int64_t)j * plane + (int64_t)kx * H + ky;
  float2* po = S + (int64_t)(j + n) * plane + (int64_t)kx * H + ky;
  const float sy = shifts[2 * j], sx = shifts[2 * j + 1];
  const float fy = (float)(ky < (H + 1) / 2 ? ky : ky - H) * (1.0f / (float)H);  // fftfreq(H)
  const float fxa = (float)kx * (1.0f / (float)W);
  const float fxb = kx == 0 ? 0.5f : (float)(kx - W / 2) * (1.0f / (float)W);
  float ws, wc, as, ac, bs, bc;
  sincospif(-2.0f * fxa, &ws, &wc);  // w = exp(-2 pi i kx / W)
  sincospif(-2.0f * (fy * sy + fxa * sx), &as, &ac);
  // a mirrored bin sees the Nyquist ROW of an even-height frame with the opposite sign: it is the
  // conjugate of the stored bin (-ky, .), and fftfreq gives -1/2 for both ky = H/2 and -ky
  const float fyb = (kx != 0 && (H & 1) == 0 && ky == H / 2) ? 0.5f : fy;
  sincospif(-2.0f * (fyb * sy + fxb * sx), &bs, &bc);
  const float2 E = *pe, O = *po;
  const float2 wO = make_float2(wc * O.x - ws * O.y, wc * O.y + ws * O.x);
  const float2 A = make_float2(E.x + wO.x, E.y + wO.y), B = make_float2(E.x - wO.x, E.y - wO.y);
  const float2 YA = make_float2(A.x * ac - A.y * as, A.x * as + A.y * ac);
  const float2 YB = make_float2(B.x * bc - B.y * bs, B.x * bs + B.y * bc);
  const float2 P = make_float2(0.5f * (YA.x + YB.x), 0.5f * (YA.y + YB.y));
  const float2 D = make_float2(0.5f * (YA.x - YB.x), 0.5f * (YA.y - YB.y));
  *pe = P;
  *po = make_float2(wc * D.x + ws * D.y, wc * D.y - ws * D.x);  // conj(w) D
}

// Exposure-weighted accumulation (mc_dose_accumulate, examples/ttMotion.py:331-351) in the same
// polyphase form: YA += q_f(|f_A|) (E_f + w O_f), YB += q_f(|f_B|) (E_f - w O_f) over the frames of a
// chunk (the weights are real and depend on |f| only, so the mirrored bin needs no conjugation
// care); on the last chunk both are normalised by 1 / sqrt(sum_f q_f^2) at their own frequency
// and folded back into the spectra of the even / odd columns of the weighted sum.
__device__ __forceinline__ float dose_mh(float fy, float fx, float pixel_size, float vscale) {
  const float f = fmaxf(sqrtf(fy * fy + fx * fx) / pixel_size, 1e-6f);
  return -0.5f / ((0.24499f * powf(f, -1.6649f) + 2.8141f) * vscale);
}

__global__ __launch_bounds__(256) void polyphase_dose_kernel(const float2* __restrict__ S, int n, int frame0,
                                                             int total_frames, float2* __restrict__ A, int nkx,
                                                             int H, int W, float pixel_size, float pre_exposure,
                                                             float dose_per_frame, float vscale, int first,
                                                             int last) {
  const int kx = blockIdx.y;
  const int ky = blockIdx.x * 256 + threadIdx.x;
  if (ky >= H) return;
  const int64_t plane = (int64_t)nkx * H, i = (int64_t)kx * H + ky;
  const float fy = (float)(ky < (H + 1) / 2 ? ky : ky - H) * (1.0f / (float)H);
  const float fxa = (float)kx * (1.0f / (float)W);
  const float fxb = kx == 0 ? 0.5f : (float)(kx - W / 2) * (1.0f / (float)W);
  const float mha = dose_mh(fy, fxa, pixel_size, vscale), mhb = dose_mh(fy, fxb, pixel_size, vscale);
  float ws, wc;
  sincospif(-2.0f * fxa, &ws, &wc);
  float2 ya = first ? make_float2(0.f, 0.f) : A[i], yb = first ? make_float2(0.f, 0.f) : A[plane + i];
  for (int j = 0; j < n; ++j) {
    const float dose = pre_exposure + dose_per_frame * (float)(frame0 + j + 1);
    const float qa = expf(mha * dose), qb = expf(mhb * dose);
    const float2 E = S[(int64_t)j * plane + i], O = S[(int64_t)(j + n) * plane + i];
    const float2 wO = make_float2(wc * O.x - ws * O.y, wc * O.y + ws * O.x);
    ya.x += qa * (E.x + wO.x); ya.y += qa * (E.y + wO.y);
    yb.x += qb * (E.x - wO.x); yb.y += qb * (E.y - wO.y);
  }
  if (last) {
    float qqa = 0.f, qqb = 0.f;
    for (int j = 0; j < total_frames; ++j) {
      const float dose = pre_exposure + dose_per_frame * (float)(j + 1);
      const float qa = expf(mha * dose), qb = expf(mhb * dose);
      qqa += qa * qa; qqb += qb * qb;
    }
    const float ra = 1.0f / sqrtf(qqa), rb = 1.0f / sqrtf(qqb);
    ya.x *= ra; ya.y *= ra; yb.x *= rb; yb.y *= rb;
    const float2 P = make_float2(0.5f * (ya.x + yb.x), 0.5f * (ya.y + yb.y));
    const float2 D = make_float2(0.5f * (ya.x - yb.x), 0.5f * (ya.y - yb.y));
    ya = P;
    yb = make_float2(wc * D.x + ws * D.y, wc * D.y - ws * D.x);
  }
  A[i] = ya;
  A[plane + i] = yb;
}

extern "C" int mc_polyphase_dose_accumulate(const void* S, int nframes, int frame0, int total_frames, void* A,
                                            int nkx, int H, int W, float pixel_size, float pre_exposure,
                                            float dose_per_frame, float voltage, int first, int last,
                                            void* stream) {
  if (!S || !A || nframes < 1 || frame0 < 0 || total_frames < frame0 + nframes || H < 2 || W < 8 || (W & 3) ||
      nkx != W / 4 + 1 || nkx > 65535 || !(pixel_size > 0.f) || !(dose_per_frame >= 0.f))
    return MC_ERR_ARG;
  const float vscale = voltage >= 300.f ? 1.0f : (voltage >= 200.f ? 0.8f : 0.75f);
  dim3 grid((H + 255) / 256, nkx);
  hipLaunchKernelGGL(polyphase_dose_kernel, grid, dim3(256), 0, (hipStream_t)stream, (const float2*)S, nframes,
                     frame0, total_frames, (float2*)A, nkx, H, W, pixel_size, pre_exposure, dose_per_frame,
                     vscale, first, last);
  return mc_check_launch();
}

extern "C" int mc_polyphase_fourier_shift(void* S, const float* shifts_px, int nframes, int nkx, int H, int W,
                                          void* stream) {
  if (!S || !shifts_px || nframes < 1 || nframes > 65535 || H < 2 || W < 8 || (W & 3) || nkx != W / 4 + 1 ||
      nkx > 65535)
    return MC_ERR_ARG;
  dim3 grid((H + 255) / 256, nkx, nframes);
  hipLaunchKernelGGL(polyphase_shift_kernel, grid, dim3(256), 0, (hipStream_t)stream, (float2*)S, shifts_px,
                     nframes, nkx, H, W);
  return mc_check_launch();
}
