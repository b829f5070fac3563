/* libmcorr -- MI355X (gfx950) motion-correction kernels, C ABI.
 *
 * Drop-in boundary for the estimate -> correct hot path of
 * teamtomo/torch-motion-correction.  The reference has no FFI of its own (it is pure
 * Python on torch ops), so these entry points are what a
 * binding of that path needs: plain device pointers, sizes and a hipStream_t passed
 * as void*.  No torch types appear here.  Every function is asynchronous on
 * `stream`, allocates nothing, never synchronises, and returns 0 (MC_OK), a negative
 * MC_ERR_* code, or a positive hipError_t from the launch.
 *
 * All pointers are DEVICE pointers unless marked (host).  Complex values are
 * interleaved float pairs (re, im).  The Python host layer that mirrors the
 * reference's function signatures on top of this ABI is
 * torch_motion_correction_amd/ (ctypes); INTEGRATION.md shows the binding.
 *
 * Reference citations are relative to /root/reference/src/torch_motion_correction/.
 */
#ifndef MCORR_H
#define MCORR_H
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MCORR_ABI_VERSION 1
int mc_abi_version(void);

/* Geometry of one pruned 2-D real transform (host struct, passed by pointer).
 * W,H: transform size (powers of two: 32<=W<=8192, 16<=H<=4096).
 * nkx: kept rfft columns [0,nkx).  kyp,kyn: kept ky rows [0,kyp) U [H-kyn,H).
 * [y0,y0+ny) x [x0,x1): window region that can be non-zero (support of the mask);
 * ny and H must be multiples of RG (rows per workgroup, 1..16); x0,x1 even. */
typedef struct mc_xc_geom {
  int W, H, nkx, kyp, kyn, y0, ny, x0, x1, RG;
} mc_xc_geom;

/* ---- plan constants ------------------------------------------------------------ */

/* Soft-edged disk mask (h*w floats): replaces torch_grid_utils.circle as called at
 * estimate_motion_xc.py:69-74 and :262-264 -- 1 where |p-c| < radius (c = (h//2,w//2)),
 * cos(pi/2 * d/smoothing_radius) of the exact Euclidean distance transform d to the
 * disk for 0 < d <= smoothing_radius, else 0.  halfw: scratch of h ints. */
int mc_circle_mask(float* mask, int* halfw, int h, int w, float radius, float smoothing_radius,
                   void* stream);

/* Combined band-pass * B-factor envelope on the pruned grid, filt[kx][kyi]
 * (nkx * (kyp+kyn) floats): replaces b_envelope (estimate_motion_xc.py:81-88,:266-273)
 * and prepare_bandpass_filter/bandpass_filter (utils.py:87-114) -- value
 * exp(-B*(f/pixel_size)^2/4) where low < f <= high (cycles/px), else 0. */
int mc_xc_filter(float* filt, const mc_xc_geom* geom, float low, float high, float b_factor,
                 float pixel_size, void* stream);

/* storage types of frame data (the values of mc_condition_movie's `kind`) */
#define MC_STORE_U8 0
#define MC_STORE_I16 1
#define MC_STORE_F16 2
#define MC_STORE_F32 3

/* ---- a2: normalize_image statistics (utils.py:49-84) ---------------------------- */
/* mean and unbiased std of stack[:, hl:hu, wl:wu] over all t frames jointly.
 * acc: 2 doubles of scratch; out3 = {mean, 1/std, std} as floats. */
int mc_central_box_stats(const float* stack, int t, int h, int w, int hl, int hu, int wl, int wu,
                         double* acc, float* out3, void* stream);
/* The same over frames in their storage type (MC_STORE_F32 / MC_STORE_F16): statistics of the
 * fp32 up-cast, read straight from the fp16 bytes. */
int mc_central_box_stats_t(const void* stack, int storage, int t, int h, int w, int hl, int hu, int wl,
                           int wu, double* acc, float* out3, void* stream);
/* dst = (src - mean) * (1/std), n elements (normalize_image's elementwise pass). */
int mc_normalize(const float* src, float* dst, int64_t n, const float* mean_rstd, void* stream);

/* ---- a1/a6/a8: cross-correlation shift search ----------------------------------- */
/* Dynamic LDS bytes K1/K4 need for this geometry (host helper). */
int mc_xc_rows_lds_bytes(const mc_xc_geom* geom);

/* K1.  For each job j: window origin src + job_off[j], rows row_stride floats apart;
 * each sample becomes (x-mean)*rstd*mask^expo[j] (mask: H*W floats or NULL;
 * mean_rstd NULL = no normalisation; job_expo NULL = exponent 1); real FFT along x;
 * first nkx bins -> T1[j][kx][ny] (complex).  estimate_motion_xc.py:66,77-78 / :339-345. */
int mc_xc_rows_forward(const float* src, const int64_t* job_off, int64_t row_stride,
                       const int* job_expo, const float* mask, const float* mean_rstd, void* T1,
                       const void* tw_row, int njobs, const mc_xc_geom* geom, void* stream);

/* N2, fused conditioning (examples/ttMotion.py:90-121 gain_correct, :180-199 set_frames_mean_zero, done on
 * the fly by the kernels that read the raw bytes: no conditioned fp32 movie exists).  One pass over a raw
 * (t,h,w) movie of storage `kind` (MC_STORE_*) and its optional (h,w) gain reference:
 *   stats[3 f ..] = { sum v, sum_box v, sum_box v^2 },  v = raw * gain,  box = [hl,hu) x [wl,wu) -- the central box of
 *   normalize_image (utils.py:76-81);   mu[f] = mean of frame f (0 when mean_zero == 0);
 *   mean_rstd = {mean, 1/std} of the conditioned box values of ALL frames jointly (unbiased, utils.py:82-83);
 *   sub[f] = mu[f] + mean: what mc_xc_rows_forward_raw subtracts from raw * gain. */
int mc_raw_movie_stats(const void* raw, int kind, const float* gain, int nframes, int h, int w, int hl, int hu,
                       int wl, int wu, int mean_zero, double* stats, float* mu, float* sub, float* mean_rstd,
                       void* stream);

/* Input conditioning of raw detector frames (caller-side steps of the reference's pipeline,
 * examples/ttMotion.py:90-121 gain multiply and :174-199 per-frame mean-zero):
 * out[f] = raw[f] * gain - mean(raw[f] * gain), fp32 out.  kind: storage type of raw, 0 = u8,
 * 1 = i16, 2 = f16, 3 = f32; gain: (h*w) floats or NULL; mean_zero != 0 needs `sums`
 * (nframes doubles of scratch).  hw = pixels per frame. */
int mc_condition_movie(const void* raw, int kind, const float* gain, int nframes, int64_t hw,
                       int mean_zero, double* sums, float* out, void* stream);
/* The same with the example pipeline's hot-pixel step between the gain multiply and the mean-zero
 * step (examples/ttMotion.py:127-172): a pixel of v = raw * gain is hot when it lies more than
 * `threshold` population standard deviations from its frame's mean -- the example's detection,
 * reproduced exactly.  Replacement: the example draws a RANDOM neighbour; here (documented as
 * ours) the mean of the up-to-8 neighbours that are not hot themselves, from the frame before any
 * replacement (the frame mean if all are hot).  The mean subtracted afterwards is the mean AFTER
 * replacement.  stats: 3 * nframes doubles of scratch (sum, sum of squares, replacement delta);
 * hot_count: nframes ints (number of hot pixels per frame) or NULL. */
int mc_condition_movie_hot(const void* raw, int kind, const float* gain, int nframes, int h, int w,
                           int mean_zero, float threshold, double* stats, int* hot_count, float* out,
                           void* stream);

/* Dose-weighted accumulation in Fourier space (the caller-side exposure filter of the
 * reference's pipeline, examples/ttMotion.py:331-351, crit_exposure_bfactor = -1):
 * A[kx][ky] (+)= sum_j q_{frame0+j}(k) * S[j][kx][ky] over the nframes full spectra in S
 * ([j][W/2+1][H] complex, as K1+K2 with the full geometry produce them); first != 0 starts A
 * from zero, last != 0 applies 1/sqrt(sum over all total_frames of q^2).  An inverse
 * transform of A (mc_fourier_shift_cols_inverse with zero shifts + mc_xc_rows_inverse_store)
 * then gives sum_f irfft2(q_f * rfft2(frame_f)).  q_f = exp(-0.5 N_f / N_c(k)), Grant &
 * Grigorieff 2015; third-party semantics, parity unpinned (oracle/thirdparty_semantics.py). */
int mc_dose_accumulate(const void* S, int nframes, int frame0, int total_frames, void* A, int W,
                       int H, float pixel_size, float pre_exposure, float dose_per_frame,
                       float voltage, int first, int last, void* stream);

/* K1 for rows of 1024 samples (1024 x 1024 patches; needs W == 1024, nkx <= 128, ny % 8 == 0,
 * a mask, every exponent >= 1; MC_ERR_UNSUPPORTED otherwise): one wavefront per row, and with
 * expo_b != NULL the same samples are transformed twice, with mask^expo_a[j] into T1a and
 * mask^expo_b[j] into T1b -- the U and V spectra of the mean-except-current reference
 * (estimate_motion_xc.py:315-346) from one read of the patch rows.  Layouts as
 * mc_xc_rows_forward. */
int mc_xc_rows_forward_dual(const float* src, const int64_t* job_off, int64_t row_stride,
                            const int* expo_a, const int* expo_b, const float* mask,
                            const float* mean_rstd, void* T1a, void* T1b, const void* tw_row,
                            int njobs, const mc_xc_geom* geom, const int* row_chord /* NULL or as in
                            mc_xc_rows_forward_stats */, void* stream);
/* The same reading the samples in their storage type (MC_STORE_F32 or MC_STORE_F16; job_off and
 * row_stride in elements): an fp16 movie is transformed without an fp32 copy of it.  Results are
 * those of the fp32 up-cast (the reference cannot run Half on the CPU at all). */
int mc_xc_rows_forward_dual_t(const void* src, int storage, const int64_t* job_off, int64_t row_stride,
                              const int* expo_a, const int* expo_b, const float* mask,
                              const float* mean_rstd, void* T1a, void* T1b, const void* tw_row,
                              int njobs, const mc_xc_geom* geom, const int* row_chord, void* stream);

/* Column-transform engine of K2 / the near-window K3: 0 = automatic (H == 4096 with at most
 * 512 kept rows at either end of the spectrum -> register-resident radix-16 transform;
 * H == 1024 with at most 128 -> one wavefront per column, mc_wave_fft.h),
 * 1 = always the radix-8 Stockham passes.  Process-wide; results agree to fp32 rounding. */
int mc_xc_col_engine(int mode);

/* Scheduling aid for a caller that overlaps movies on several streams (pipeline.py): while `event`
 * (a hipEvent_t) is non-NULL, mc_xc_correlate_argmax records it on its stream right after the
 * near-window column pass -- the last kernel of the search that fills the machine.  Process-wide;
 * NULL switches it off.  No effect on results. */
int mc_xc_after_k3n_event(void* event);

/* Row-transform engine of K1: 0 = automatic (W == 4096, nkx <= 512, no per-job exponents,
 * 16-byte aligned src/mask and row_stride % 4 == 0 -> one wavefront per row, mc_wave_fft.h;
 * job_off[] must then be multiples of 4 floats, as whole-frame offsets f*h*w are),
 * 1 = always one workgroup per row.  Process-wide; results agree to fp32 rounding. */
int mc_xc_row_engine(int mode);

/* m0 = {mean of the n floats at x, 1, 1}: the provisional mean mc_xc_rows_forward_stats takes
 * (one small workgroup; any value near the true mean serves). */
int mc_xc_provisional_mean(const float* x, int n, float* m0, void* stream);
int mc_xc_provisional_mean_t(const void* x, int storage, int n, float* m0, void* stream);

/* K1 with the normalisation statistics fused in (whole-frame jobs only): samples become
 * (x - m0[0]) * mask (m0: device float[3] = {provisional mean, 1, 1}); while reading, the
 * sums of (x-m0) and (x-m0)^2 over the central box rows [hl,hu) x cols [wl,wu) (window
 * coordinates, inside the mask support, wl/wu even) of every job are accumulated;
 * afterwards fix = {mean - m0, 1/std} and out3 = {mean, 1/std, std} (unbiased std over
 * all jobs jointly, utils.py:76-84).  acc: 128 doubles scratch (64 x {sum, sumsq}).  Feed `fix` to
 * mc_xc_cols_forward_fix, which finishes the normalisation by linearity.
 * row_chord (NULL, or H x 2 ints per WINDOW row y: first and last 4-aligned column whose quad
 * holds a non-zero mask value): samples outside a row's chord of the mask disk are not fetched
 * (they are multiplied by the mask's exact zero) -- 21 % of the support box of a circular mask.
 * The statistics box must lie inside the chords (the caller checks; plan.row_chords). */
int mc_xc_rows_forward_stats(const float* src, const int64_t* job_off, int64_t row_stride,
                             const float* mask, const float* m0, void* T1, const void* tw_row,
                             int njobs, const mc_xc_geom* geom, int hl, int hu, int wl, int wu,
                             double* acc, float* fix, float* out3, const int* row_chord,
                             void* stream);
/* The same reading the frames in their storage type (MC_STORE_F32 / MC_STORE_F16; job_off and row_stride
 * in samples): fp16 frames are read as they are by the wavefront-per-row kernel (4096-column frames;
 * MC_ERR_UNSUPPORTED otherwise: widen the stack and call mc_xc_rows_forward_stats). */
int mc_xc_rows_forward_stats_t(const void* src, int storage, const int64_t* job_off, int64_t row_stride,
                               const float* mask, const float* m0, void* T1, const void* tw_row,
                               int njobs, const mc_xc_geom* q, int hl, int hu, int wl, int wu,
                               double* acc, float* fix, float* out3, const int* row_chord, void* stream);

/* N2: K1 from the raw bytes of a u8 / i16 movie, conditioned on the fly (no fp32 movie):
 *   rows of (raw * gain - job_sub[job]) * mean_rstd[1] * mask  ->  T1, as mc_xc_rows_forward writes it.
 * `gain` has the frames' row pitch (whole-frame jobs); job_sub / mean_rstd come from mc_raw_movie_stats;
 * follow with mc_xc_cols_forward / mc_xcg_cols_forward (no fix-up: the statistics are known before this pass).
 * mc_xc_rows_forward_raw: power-of-two widths (4096 columns with 16-byte aligned buffers and row_stride % 8 == 0 on
 * the wave-per-row engine, the others on the workgroup engine); mc_xcg_rows_forward_raw: rows of 5760 / 11520
 * samples (the K3 formats; `line` = the direct 2880 / 5760-point plan).  Anything else: MC_ERR_UNSUPPORTED. */
int mc_xc_rows_forward_raw(const void* raw, int storage, const float* gain, const int64_t* job_off,
                           int64_t row_stride, const float* mask, const float* job_sub, const float* mean_rstd,
                           void* T1, const void* tw_row, int njobs, const mc_xc_geom* q, const int* row_chord,
                           void* stream);

/* K2.  Column FFT of T1, kept ky rows, times filt (or NULL) -> S[j][kx][kyi].
 * estimate_motion_xc.py:78,98 / :340-346. */
int mc_xc_cols_forward(const void* T1, const float* filt, void* S, const void* tw_col, int njobs,
                       const mc_xc_geom* geom, void* stream);
/* K2 with the linear normalisation fix-up: S = filt * fix[1] * (Y - fix[0] * Mhat), Mhat =
 * pruned spectrum of the mask ([kx][kyi] complex, i.e. K1+K2 of an all-ones window). */
int mc_xc_cols_forward_fix(const void* T1, const float* filt, void* S, const void* tw_col,
                           int njobs, const mc_xc_geom* geom, const float* fix, const void* Mhat,
                           void* stream);

/* K3.  pair p: conj(S_ref[ref_idx[p]]) * S_cur[cur_idx[p]] * scale, inverse column FFT
 * -> T2[p][kx][H].  estimate_motion_xc.py:112-113 / :349-350. */
int mc_xc_cols_inverse(const void* S_cur, const int* cur_idx, const void* S_ref,
                       const int* ref_idx, void* T2, const void* tw_col, float scale, int npairs,
                       const mc_xc_geom* geom, void* stream);

/* Rows stored per end of the map by the near-window search below (searched rows + guard rows). */
int mc_xc_near_rows(const mc_xc_geom* geom);

/* K3+K4+K5 for the arg-max search without materialising the map or T2 (power-of-two
 * W and H >= 256; estimate_motion_xc.py:106-123).  Rows [0, n) and [H-n, H) of the
 * correlation map (n = mc_xc_near_rows(geom)) are transformed from T2_near
 * ([p][kx][2n] complex) while the column pass accumulates the per-row triangle-inequality
 * bounds of ALL rows; only if a far row's bound can still reach the near-window maximum are
 * the full column pass (T2_full, [p][kx][H]) and the far row groups evaluated, a decision
 * taken on the device (the fallback kernels are enqueued and return at once).  The result
 * is the exact arg-max of the full map either way.  n counts the searched rows plus a few
 * guard rows, so that nb (optional, [p][3][3] floats: the map around every peak as
 * mc_xc_peak_neighbourhood gives it, for the sub-pixel parabola fit of
 * estimate_motion_xc.py:414-483) can be taken from T2_near too.  part_val: npairs*(H/RG) +
 * npairs*H floats; part_idx: npairs*(H/RG) + npairs + 1 ints.
 * shift_rows (optional): pair p's shift goes to row shift_rows[p] of a (n_shift_rows, 2) table
 * that is zeroed first (rows no pair writes -- the reference frame -- stay exactly zero,
 * estimate_motion_xc.py:102); NULL: shifts is (npairs, 2), row p. */
int mc_xc_correlate_argmax(const void* S_cur, const int* cur_idx, const void* S_ref,
                           const int* ref_idx, void* T2_full, void* T2_near, float* part_val,
                           int* part_idx, int* peaks, float* shifts, const int* shift_rows,
                           int n_shift_rows, float* nb, const void* tw_col, const void* tw_row,
                           float scale, int npairs, const mc_xc_geom* geom, void* stream);

/* K4+K5.  Inverse real row FFT of T2 fused with the arg-max (first maximum, as
 * torch.argmax): peaks[p] = flat index y*W+x, shifts[p] = (sy,sx) after the
 * wrap-around rule `p if p <= n//2 else p-n`.  part_val: scratch of npairs*(H/RG) +
 * npairs*H floats; part_idx: scratch of npairs*(H/RG) + npairs ints.  The search is an exact
 * branch and bound: row groups whose triangle-inequality bound cannot reach the best
 * value found so far are skipped.  estimate_motion_xc.py:113-121 / :350-355,368-369. */
int mc_xc_rows_inverse_argmax(const void* T2, float* part_val, int* part_idx, int* peaks,
                              float* shifts, const void* tw_row, int npairs,
                              const mc_xc_geom* geom, void* stream);

/* K6.  nb[p][3][3]: correlation values at (py-1..py+1, px-1..px+1) around peaks[p]
 * (NaN outside the map), same arithmetic as K4.  Feeds the parabola fit of
 * _apply_sub_pixel_refinement, estimate_motion_xc.py:414-483. */
int mc_xc_peak_neighbourhood(const void* T2, const int* peaks, float* nb, const void* tw_row,
                             int npairs, const mc_xc_geom* geom, void* stream);

/* Reference spectra for reference_strategy="mean_except_current"
 * (estimate_motion_xc.py:310-328 + the in-place mask aliasing):
 * REF[f][g] = inv_count * sum_{o != f} (o in S_f ? V[o][g] : U[o][g]), spectra of `len`
 * complex values, g in [0,npatch).  The sets S_f come as a schedule the
 * host derives from the memo replay: frame f either adds sched_idx[sched_ptr[f] ..
 * sched_ptr[f+1]) to the running set (sched_rebuild[f] == 0) or replaces the set by that
 * list (sched_rebuild[f] == 1). */
int mc_xc_ref_mean_except_current(const void* U, const void* V, const int* sched_ptr,
                                  const int* sched_idx, const uint8_t* sched_rebuild, void* REF,
                                  int t, int npatch, int64_t len, float inv_count, void* stream);

/* ---- a11/a12/a13: per-frame shift post-processing ------------------------------- */
/* Sub-pixel parabola (rules Q4/Q5), wrap-around, outlier rejection
 * (estimate_motion_xc.py:538-627) and accumulation into field (2,t,gh,gw) [Angstrom]
 * for `nf` frames: pair index p = fi*npatch + g maps to frame frames[fi].
 * flags bit0 = sub_pixel_refinement, bit1 = outlier_rejection. */
int mc_field_accumulate(const int* peaks, const float* nb, const int* frames, int nf, int npatch,
                        int P, int t, float pixel_spacing, float outlier_threshold, int flags,
                        float* field, void* stream);
/* Savitzky-Golay (polyorder 1, mode "interp") along t for each (c,gy,gx), as
 * scipy.signal.savgol_filter at estimate_motion_xc.py:528-529; then (optional)
 * subtraction of the single global mean (xc.py:410).  Any window 3 <= window <= t, odd or
 * even (the reference reaches an even one as min(window|1, t) for even t: scipy then centres
 * the interior mean on x[i-w/2+1 .. i+w/2]).  field_out may alias field_in only if window < 3. */
int mc_field_smooth_center(const float* field_in, float* field_out, int t, int npatch, int window,
                           int subtract_mean, void* stream);

/* ---- a14/a16: cubic spline grids ------------------------------------------------- */
/* Evaluate a (c,nt,nh,nw) uniform cubic spline grid on the tensor-product lattice
 * given by per-axis tap tables: for lattice coordinate i of an axis, 4 sample indices
 * idx_*[4*i+k] and 4 weights w_*[4*i+k] (basis weights of the Catmull-Rom or B-spline
 * matrix with the library's linear-extrapolation edge samples folded in by the host).
 * out[c][NT][NY][NX].  Replaces CubicCatmullRomGrid3d / CubicBSplineGrid3d as used by
 * deformation_field_utils.py:9-39,42-93,96-126. */
int mc_spline_lattice(const float* data, int c, int nt, int nh, int nw, const int* idx_t,
                      const float* w_t, int NT, const int* idx_y, const float* w_y, int NY,
                      const int* idx_x, const float* w_x, int NX, float* out, void* stream);

/* The same grid at npoints scattered (t, y, x) points -- evaluate_deformation_field on arbitrary
 * points (deformation_field_utils.py:9-39): per point 4 taps per axis, idx_*[4*i+k] / w_*[4*i+k];
 * out[npoints][c]. */
int mc_spline_points(const float* data, int c, int nt, int nh, int nw, const int* idx_t, const float* w_t,
                     const int* idx_y, const float* w_y, const int* idx_x, const float* w_x, int64_t npoints,
                     float* out, void* stream);

/* ---- a15/a17/a18: deformation-field warp ------------------------------------------ */
/* lattice: (nframes, 2, GH, GW) Angstrom shifts on the 10x-oversampled lattice
 * (evaluate_deformation_field_at_t, correct_motion.py:67-72).  For every frame:
 * bicubic/reflection upsample to per-pixel shifts (get_pixel_shifts,
 * correct_motion.py:132-185), then bicubic/border resample of the frame at
 * pixel + shift/pixel_spacing with zero outside (_correct_frame + sample_image_2d,
 * correct_motion.py:81-129).  scratch: mc_warp_scratch_bytes() bytes, 16-byte aligned.
 * out_frames (nframes*h*w) and/or out_sum (h*w, OVERWRITTEN with the sum over the frames: the
 * caller need not clear it) may be NULL (not both). */
int mc_warp_scratch_bytes(int nframes, int h, int w, int GH, int GW, int64_t* bytes /*host*/);
int mc_warp_frames(const float* frames, int nframes, int h, int w, const float* lattice, int GH,
                   int GW, float pixel_spacing, float* scratch, float* out_frames, float* out_sum,
                   void* stream);
/* The same with the frames in their storage type (MC_STORE_F32 / MC_STORE_F16; outputs stay fp32).
 * fp16 frames are resampled straight from the fp16 bytes (half the HBM read) when w % 8 == 0, the
 * frames are 16-byte aligned and the lattice is sparse (32 pixel rows span <= 1.5 lattice cells);
 * otherwise MC_ERR_UNSUPPORTED: widen the stack and call mc_warp_frames. */
int mc_warp_frames_t(const void* frames, int storage, int nframes, int h, int w, const float* lattice,
                     int GH, int GW, float pixel_spacing, float* scratch, float* out_frames, float* out_sum,
                     void* stream);

/* N2: the rigid warp (correct_motion for a (2,t,1,1) field, correct_motion.py:18-78) fed from the RAW movie:
 * every sample is conditioned as raw * gain - mu[f] on its way to the resampler (examples/ttMotion.py:90-121,
 * 180-199), so the result equals mc_warp_rigid on the output of mc_condition_movie without that fp32 movie.
 * storage: MC_STORE_U8 or MC_STORE_I16; gain (h,w) fp32; mu from mc_raw_movie_stats; scratch / phase as
 * mc_warp_rigid_phase.  w % 4 == 0 and 16-byte aligned buffers, else MC_ERR_UNSUPPORTED. */
int mc_warp_rigid_raw(const void* raw, int storage, const float* gain, const float* mu, int nframes, int h, int w,
                      const float* shifts_px, float* scratch, float* out_frames, float* out_sum, int phase,
                      void* stream);

/* The tail of the rigid movie pipeline in two launches: integer-peak shifts (t,2) px of estimate_global_motion ->
 * field (2,t) Angstrom (image_shifts_to_deformation_field, deformation_field_utils.py:129-162), the per-frame
 * shifts the corrector uses (the field's spline at the frame times, correct_motion.py:57-72, lattice point
 * (0,0), / pixel_spacing) and the weight tables of mc_warp_rigid_phase(phase 1) in `scratch`.  Results are
 * bit for bit those of mc_spline_lattice + mc_warp_rigid_phase. */
int mc_rigid_tables_from_shifts(const float* shifts, float pixel_spacing, const int* idx_t, const float* w_t,
                                const float* w_y, const float* w_x, int nframes, int h, int w, float* field,
                                float* shifts_px, float* scratch, void* stream);

/* Rigid special case of mc_warp_frames: a (2,nt,1,1) field gives each frame ONE shift,
 * shifts_px[f] = (sy, sx) in pixels (device).  The coordinate chain is then separable
 * and the resample is a regular 5x5 separable correlation (see warp.hip).  Same
 * outputs/contract as mc_warp_frames.  scratch: mc_warp_rigid_scratch_bytes(). */
int mc_warp_rigid_scratch_bytes(int nframes, int h, int w, int64_t* bytes /*host*/);
int mc_warp_rigid(const float* frames, int nframes, int h, int w, const float* shifts_px,
                  float* scratch, float* out_frames, float* out_sum, void* stream);
/* The same in two steps, for callers that want to time or schedule the resampling kernel on its
 * own: phase 1 fills the per-frame weight tables in `scratch`, phase 2 (same arguments) resamples;
 * phase 0 = mc_warp_rigid. */
int mc_warp_rigid_phase(const float* frames, int nframes, int h, int w, const float* shifts_px,
                        float* scratch, float* out_frames, float* out_sum, int phase, void* stream);
/* The same with the frames in their storage type (MC_STORE_F32 / MC_STORE_F16; outputs stay fp32): fp16
 * frames are DMA'd to LDS as they are and widened on the way to the registers (half the read bytes).
 * fp16 needs w % 8 == 0 and 16-byte aligned frames / out_frames, else MC_ERR_UNSUPPORTED (widen and
 * call mc_warp_rigid). */
int mc_warp_rigid_phase_t(const void* frames, int storage, int nframes, int h, int w, const float* shifts_px,
                          float* scratch, float* out_frames, float* out_sum, int phase, void* stream);

/* get_pixel_shifts (correct_motion.py:132-185) for one (2,GH,GW) lattice: out (h,w,2)
 * shifts in px.  scratch: mc_warp_scratch_bytes(1,h,w,GH,GW). */
int mc_pixel_shifts(const float* lattice, int GH, int GW, int h, int w, float pixel_spacing,
                    float* scratch, float* out, void* stream);
/* The same at caller-supplied coordinates -- the `pixel_grid` argument of get_pixel_shifts
 * (correct_motion.py:136,167-168) when it is not the identity grid: coords_yx (n,2) pixel
 * coordinates (y,x) of an (h,w) frame -> out (n,2) shifts in px. */
int mc_pixel_shifts_at(const float* lattice, int GH, int GW, int h, int w, float pixel_spacing,
                       const float* coords_yx, int64_t n, float* out, void* stream);

/* ---- full-spectrum transforms with a row-major spectrum -------------------------------------
 * What correct_motion_fast (correct_motion.py:484-496) and the exposure-filtered sum
 * (examples/ttMotion.py:331-351) run on for rows of W = 64 .. 8192 (powers of two), 5760 or 11520
 * columns and columns of H = 256 .. 4096 (powers of two), 4092 or 8184 rows -- any combination;
 * the K3 formats 4092 x 5760 and 8184 x 11520 are transformed by mixed-radix passes (radix 31, 11,
 * 8, 5, 3), no chirp-z.  MC_ERR_UNSUPPORTED otherwise: use the pruned-engine entry points below.
 * S[job][y][pitch] complex, pitch = mc_full_spectrum_pitch(W) = W/2 + 1 rounded up to 16.
 *   mc_full_rows_forward   rfft along x of njobs frames (origin src + job_off[j], rows row_stride
 *                          floats apart) -> S
 *   mc_full_cols_shift     per job: fft along y, * exp(-2 pi i (fy sy + fx sx)) * scale,
 *                          shifts[j] = (sy, sx) px, ifft along y, in place
 *   mc_full_cols_dose      A += sum_f q_f(k) fft_y(S_f) over the nframes frames of S (frames frame0..
 *                          of total_frames; first: A starts at zero); last: A *= scale / sqrt(sum_f
 *                          q_f^2) over ALL frames and is transformed back along y (then
 *                          mc_full_rows_inverse(A) is the exposure-filtered sum).  q_f as in
 *                          mc_dose_accumulate.  A: H * pitch complex.
 *   mc_full_rows_inverse   irfft along x (c2r, unscaled) of S -> real rows at out + out_off[j] */
int mc_full_spectrum_pitch(int W);
int mc_full_rows_forward(const float* src, const int64_t* job_off, int64_t row_stride, void* S,
                         const void* tw_row, int njobs, int H, int W, int pitch, void* stream);
int mc_full_cols_shift(void* S, const float* shifts, const void* tw_col, float scale, int njobs, int H, int W,
                       int pitch, void* stream);
int mc_full_cols_dose(const void* S, int nframes, int frame0, int total_frames, void* A, const void* tw_col,
                      int H, int W, int pitch, float pixel_size, float pre_exposure, float dose_per_frame,
                      float voltage, int first, int last, float scale, void* stream);
/* Column-major copy of a chunk of row-major spectra, ST[job][kx][y] (kx <= W/2), and the exposure-weighted
 * pass reading it (H = 4096, 4092 or 8184: contiguous columns instead of 8 bytes of every 128-byte line;
 * A stays row-major; otherwise as mc_full_cols_dose). */
int mc_full_transpose(const void* S, void* ST, int njobs, int H, int W, int pitch, void* stream);
int mc_full_cols_dose_cm(const void* ST, int nframes, int frame0, int total_frames, void* A, const void* tw_col,
                         int H, int W, int pitch, float pixel_size, float pre_exposure, float dose_per_frame,
                         float voltage, int first, int last, float scale, void* stream);
int mc_full_rows_inverse(const void* S, float* out, const int64_t* out_off, int64_t out_stride,
                         const void* tw_row, int njobs, int H, int W, int pitch, void* stream);

/* correct_motion_fast (correct_motion.py:430-498): K3 variant multiplying spectrum
 * idx[p] by exp(-2*pi*i*(fy*sy+fx*sx)), shifts[p]=(sy,sx) px, then inverse columns. */
int mc_fourier_shift_cols_inverse(const void* S, const int* idx, const float* shifts, void* T2,
                                  const void* tw_col, float scale, int nframes,
                                  const mc_xc_geom* geom, void* stream);
/* K4 variant storing real rows: out + out_off[p] + y*out_stride. */
int mc_xc_rows_inverse_store(const void* T2, float* out, const int64_t* out_off,
                             int64_t out_stride, const void* tw_row, int nframes,
                             const mc_xc_geom* geom, void* stream);

/* ---- generic transform lengths (Bluestein): any even W (rows), any H (columns) ------ */
/* One line plan (host struct of device pointers): tw_m = exp(-2 pi i k/M) (M entries),
 * chirp = exp(-+ i pi j^2/n) (n entries; - for forward, + for inverse transforms),
 * bspec = FFT_M(wrapped conj(chirp)) / M (M entries); M = power of two >= 2n-1, <= 16384.
 * n is W/2 for the row functions and H for the column functions. */
typedef struct mc_xc_line {
  const void* tw_m;
  const void* chirp;
  const void* bspec;
  int M;
  int keep; /* 0: classic plan.  > 0 (mc_xcg_rows_forward only): output-pruned plan -- only
               outputs k < keep and k > n - keep are exact, which needs M >= n + 2 keep - 1 only;
               bspec = FFT_M of conj(chirp) wrapped over the offsets (-(n-1) - (keep-1) .. keep-1) / M */
} mc_xc_line;

/* Same contracts and layouts as mc_xc_rows_forward / mc_xc_cols_forward /
 * mc_xc_cols_inverse (+ Fourier-shift mode when `shifts` != NULL) /
 * mc_xc_rows_inverse_argmax (out == NULL) and mc_xc_rows_inverse_store (out != NULL),
 * without the power-of-two restriction: 4 <= W <= 16384 even or W <= 8191 odd (odd widths: the
 * row line plan has n = W points, one real sample each, instead of W / 2), 2 <= H <= 8192 (a row line plus
 * its staged bins must fit 160 KB of LDS: MC_ERR_ARG otherwise).
 * tw_row = exp(-2 pi i k / W), W entries. */
int mc_xcg_rows_forward(const float* src, const int64_t* job_off, int64_t row_stride,
                        const int* job_expo, const float* mask, const float* mean_rstd, void* T1,
                        const void* tw_row, const mc_xc_line* line, int njobs,
                        const mc_xc_geom* geom, void* stream);
/* N2: the K3-format rows (5760 / 11520 samples) from raw u8 / i16 bytes; see mc_xc_rows_forward_raw. */
int mc_xcg_rows_forward_raw(const void* raw, int storage, const float* gain, const int64_t* job_off,
                            int64_t row_stride, const float* mask, const float* job_sub, const float* mean_rstd,
                            void* T1, const void* tw_row, const mc_xc_line* line, int njobs, const mc_xc_geom* q,
                            void* stream);
int mc_xcg_cols_forward(const void* T1, const float* filt, void* S, const mc_xc_line* line,
                        int njobs, const mc_xc_geom* geom, void* stream);
int mc_xcg_cols_inverse(const void* S_cur, const int* cur_idx, const void* S_ref,
                        const int* ref_idx, const float* shifts, void* T2, const mc_xc_line* line,
                        float scale, int npairs, const mc_xc_geom* geom, void* stream);
int mc_xcg_rows_inverse(const void* T2, float* part_val, int* part_idx, int* peaks, float* shifts,
                        float* out, const int64_t* out_off, int64_t out_stride, const void* tw_row,
                        const mc_xc_line* line, int npairs, const mc_xc_geom* geom, void* stream);

/* K6 for any size (same contract as mc_xc_peak_neighbourhood, no twiddle table): the nine values
 * as direct sums over the kept columns of T2. */
int mc_xcg_peak_neighbourhood(const void* T2, const int* peaks, float* nb, int npairs,
                              const mc_xc_geom* geom, void* stream);

/* caller-side frame sum (examples/ttMotion.py:398): sum[h*w] = sum_t frames[t]. */
int mc_sum_frames(const float* frames, int nframes, int64_t hw, float* sum, void* stream);

/* ---- estimate_local_motion (estimate_motion_optimizer.py:28-439): loss + gradient ------
 * The reference rebuilds, every iteration and for every patch, rfftn(patch * mask), the
 * Fourier shift by the spline-predicted shifts, the band-pass and B-factor filters, the
 * leave-one-out reference and the loss (:361-417, :466-514, :611-671), and differentiates
 * it with autograd.  Here the masked + filtered patch spectra are computed once (pruned,
 * layout of mc_xc_cols_forward: (npatch * t jobs, nkx, nky) complex, job = patch * t +
 * frame) and an iteration is two passes over them.
 * mc_local_loss_sums: for every patch b, bin tile and frame f (partial: (npatch, ntiles,
 * t, 6) floats, to be summed over the tiles by the caller):
 *   [0] sum h fy Im(conj(S) G_f)   [1] same with fx   [2] sum h |t G_f - S|^2
 *   [3] sum h Re(G_f conj(S - G_f))   [4] sum h |S - G_f|^2   [5] sum h |G_f|^2
 * with G_f = P_f exp(-2 pi i (fy sy_f + fx sx_f)), S = sum_f G_f, shifts_px (npatch, t,
 * 2) = (y, x) pixels, fy (nky) / fx (nkx) = the bins' frequencies in cycles/pixel, hx
 * (nkx) = weight of column kx (NULL = 1; Hermitian multiplicities for the real-space
 * losses).  Loss values and shift gradients of "mse", "cc" follow from these sums
 * (csrc/local_motion.hip header); "ncc" needs mc_local_ncc_grad with ab (npatch, t, 2) =
 * (dL/dn_f, dL/dey_f) -> partial (npatch, ntiles, t, 2) = sum h f{y,x} Im(V_f G_f).
 * 1 <= t <= 512. */
int mc_local_loss_tiles(int nkx, int nky, int* ntiles);
int mc_local_loss_sums(const void* spectra, const float* shifts_px, const float* fy, const float* fx,
                       const float* hx, int npatch, int t, int nkx, int nky, float* partial,
                       void* stream);
int mc_local_ncc_grad(const void* spectra, const float* shifts_px, const float* fy, const float* fx,
                      const float* hx, const float* ab, int npatch, int t, int nkx, int nky,
                      float* partial, void* stream);

/* correct_motion_fast (correct_motion.py:484-496) for frames whose full spectrum does not fit one
 * row line (more than ~8190 columns): the frame's even and odd columns are transformed as two
 * (H, W/2) frames (mc_xcg_rows_forward / mc_xcg_cols_forward, full geometry) into S -- (2 nframes,
 * W/4 + 1, H) complex, job j = even columns of frame j, job j + nframes = odd columns -- and this
 * call applies, in place, the radix-2 butterfly to the bins of the full spectrum, the phase ramp
 * exp(-2 pi i (fy sy + fx sx)) at each bin's own frequency (shifts_px (nframes, 2) = (y, x)), and the
 * inverse butterfly; the two halves are then inverse-transformed with zero shifts and interleaved.
 * W % 4 == 0; nkx must be W/4 + 1. */
int mc_polyphase_fourier_shift(void* S, const float* shifts_px, int nframes, int nkx, int H, int W,
                               void* stream);
/* mc_dose_accumulate in the same form: S holds the spectra of the even (jobs 0..nframes-1) and odd
 * (jobs nframes..2 nframes-1) columns of a chunk of frames; A (2, W/4 + 1, H) complex accumulates
 * the weighted butterfly outputs and, on the last chunk, becomes the spectra of the even / odd
 * columns of the exposure-weighted sum (inverse-transform both with zero shifts and interleave). */
int mc_polyphase_dose_accumulate(const void* S, int nframes, int frame0, int total_frames, void* A,
                                 int nkx, int H, int W, float pixel_size, float pre_exposure,
                                 float dose_per_frame, float voltage, int first, int last,
                                 void* stream);

#ifdef __cplusplus
}
#endif
#endif /* MCORR_H */
