"""Device-side pipelines on top of libmcorr (all tensors already on the GPU).

These functions only allocate buffers (torch caching allocator), build small index
tables and enqueue libmcorr kernels on the current stream; they never synchronise.
"""

from __future__ import annotations

import ctypes as C

import numpy as np
import torch

from . import _lib, lattice, plan as planmod, spline
from ._lib import check, ptr, stream_ptr

WORKSPACE_BYTES = 2 << 30  # soft cap for one transposed intermediate (T1 / T2)
FUSED_SEARCH = True  # near-window arg-max (mc_xc_correlate_argmax); False: full T2 + separate kernels


_CONST: dict = {}


def _cached(key, build):
    """Small device-resident index/tap tables, built once per (shape, device)."""
    v = _CONST.get(key)
    if v is None:
        if len(_CONST) > 512:
            _CONST.clear()
        v = _CONST[key] = build()
    return v


def _i32(a, device):
    return torch.as_tensor(np.asarray(a, dtype=np.int32), device=device)


def _i64(a, device):
    return torch.as_tensor(np.asarray(a, dtype=np.int64), device=device)


# ------------------------------------------------------------------ statistics


STORE_F16, STORE_F32 = 2, 3  # MC_STORE_* of include/mcorr.h


def storage_of(img: torch.Tensor) -> int:
    """Frame storage type tag of the *_t entry points: fp32, or fp16 read straight from its bytes."""
    if img.dtype == torch.float32:
        return STORE_F32
    if img.dtype == torch.float16:
        return STORE_F16
    raise TypeError(f"frames must be float32 or float16 on the device, got {img.dtype}")


def central_box_stats(img: torch.Tensor, frac_low=0.25, frac_high=0.75) -> torch.Tensor:
    """(mean, 1/std, std) of the central box over all frames (utils.py:49-84); fp16 frames are read
    as they are (statistics of the fp32 up-cast)."""
    lib = _lib.load()
    t, h, w = img.shape
    hl, hu, wl, wu = int(frac_low * h), int(frac_high * h), int(frac_low * w), int(frac_high * w)
    acc = torch.empty(2, dtype=torch.float64, device=img.device)
    out3 = torch.empty(3, dtype=torch.float32, device=img.device)
    check(lib.mc_central_box_stats_t(ptr(img), storage_of(img), t, h, w, hl, hu, wl, wu, ptr(acc), ptr(out3),
                                     stream_ptr(img.device)), "mc_central_box_stats")
    return out3


def normalize(img: torch.Tensor, stats: torch.Tensor) -> torch.Tensor:
    lib = _lib.load()
    out = torch.empty_like(img)
    check(lib.mc_normalize(ptr(img), ptr(out), img.numel(), ptr(stats), stream_ptr(img.device)),
          "mc_normalize")
    return out


# ------------------------------------------------------------------ spectra


def _k1(lib, g, dev, src, off, row_stride, expo, mask, stats, T1, tw_row, n, st):
    if planmod.native_rows(g):
        return lib.mc_xc_rows_forward(ptr(src), ptr(off), row_stride, ptr(expo), ptr(mask), ptr(stats),
                                      ptr(T1), ptr(tw_row), n, g, st)
    line, _ = planmod.line_plan(planmod.row_line_length(g.W), -1, dev,
                                keep=planmod.row_line_keep(g.W, g.nkx))  # output-pruned when that shrinks M
    return lib.mc_xcg_rows_forward(ptr(src), ptr(off), row_stride, ptr(expo), ptr(mask), ptr(stats),
                                   ptr(T1), ptr(tw_row), line, n, g, st)


def _k2(lib, g, dev, T1, filt, S, tw_col, n, st):
    if planmod.native_height(g.H):
        return lib.mc_xc_cols_forward(ptr(T1), ptr(filt), ptr(S), ptr(tw_col), n, g, st)
    line, _ = planmod.line_plan(g.H, -1, dev)
    return lib.mc_xcg_cols_forward(ptr(T1), ptr(filt), ptr(S), line, n, g, st)


def _wave512_ok(g, job_expo, use_mask, min_expo):
    """Patch rows of 1024 samples can take the wave-per-row K1 (mc_xc_rows_forward_dual):
    needs the mask and per-job exponents that are all >= 1 (`min_expo`, known on the host)."""
    return (g.W == 1024 and g.nkx <= 128 and g.ny % 8 == 0 and use_mask and job_expo is not None
            and min_expo is not None and min_expo >= 1)


def _forward_spectra(src, job_off, row_stride, job_expo, pl, stats, use_mask=True, use_filter=True,
                     job_expo_b=None, min_expo=None):
    """K1+K2 for a list of jobs -> S (njobs, nkx, nky, 2).  With `job_expo_b` the rows are read
    once and transformed twice (mask^job_expo and mask^job_expo_b): returns (S_a, S_b)."""
    lib = _lib.load()
    g, dev = pl.geom, src.device
    njobs = int(job_off.numel())
    dual = job_expo_b is not None
    wave = _wave512_ok(g, job_expo, use_mask, min_expo)
    if src.dtype != torch.float32 and not wave:
        raise _lib.McorrUnsupported("only the wave-per-row patch kernel reads fp16 frames: widen the stack first")
    if dual and not wave:  # no fused kernel for this shape: two ordinary passes
        return (_forward_spectra(src, job_off, row_stride, job_expo, pl, stats, use_mask, use_filter),
                _forward_spectra(src, job_off, row_stride, job_expo_b, pl, stats, use_mask, use_filter))
    S = torch.empty((njobs, g.nkx, g.nky, 2), dtype=torch.float32, device=dev)
    Sb = torch.empty_like(S) if dual else None
    per_job = g.nkx * g.ny * 8 * (2 if dual else 1)
    chunk = max(1, min(njobs, WORKSPACE_BYTES // per_job))
    T1 = torch.empty((chunk, g.nkx, g.ny, 2), dtype=torch.float32, device=dev)
    T1b = torch.empty_like(T1) if dual else None
    st = stream_ptr(dev)
    for a in range(0, njobs, chunk):
        n = min(chunk, njobs - a)
        off = job_off[a : a + n]
        expo = None if job_expo is None else job_expo[a : a + n]
        if wave:
            expo_b = job_expo_b[a : a + n] if dual else None
            check(lib.mc_xc_rows_forward_dual_t(ptr(src), storage_of(src), ptr(off), row_stride, ptr(expo),
                                                ptr(expo_b), ptr(pl.mask), ptr(stats), ptr(T1), ptr(T1b),
                                                ptr(pl.tw_row), n, g,
                                                ptr(pl.chord) if USE_ROW_CHORDS else None, st),
                  "mc_xc_rows_forward_dual")
        else:
            check(_k1(lib, g, dev, src, off, row_stride, expo, pl.mask if use_mask else None, stats, T1,
                      pl.tw_row, n, st), "xc rows forward")
        check(_k2(lib, g, dev, T1, pl.filt if use_filter else None, S[a : a + n], pl.tw_col, n, st),
              "xc cols forward")
        if dual:
            check(_k2(lib, g, dev, T1b, pl.filt if use_filter else None, Sb[a : a + n], pl.tw_col, n, st),
                  "xc cols forward")
    return (S, Sb) if dual else S


def _peaks(S_cur, cur_idx, S_ref, ref_idx, pl, want_nbhd, shift_rows=None, n_shift_rows=0):
    """K3+K4(+K6) for all pairs -> peaks (npairs,) int32, shifts (npairs,2), nb or None.
    With `shift_rows` (int32, one row index per pair) the shifts are scattered into a zeroed
    (n_shift_rows, 2) table instead (rows no pair writes stay exactly zero)."""
    lib = _lib.load()
    g, dev = pl.geom, S_cur.device
    npairs = int(cur_idx.numel())
    peaks = torch.empty(npairs, dtype=torch.int32, device=dev)
    shifts = torch.empty((n_shift_rows if shift_rows is not None else npairs, 2), dtype=torch.float32,
                         device=dev)
    nb = torch.empty((npairs, 3, 3), dtype=torch.float32, device=dev) if want_nbhd else None
    per_pair = g.nkx * g.H * 8
    chunk = max(1, min(npairs, WORKSPACE_BYTES // per_pair))
    T2 = torch.empty((chunk, g.nkx, g.H, 2), dtype=torch.float32, device=dev)
    ngrp = g.H // g.RG
    pv = torch.empty(chunk * ngrp + chunk * g.H, dtype=torch.float32, device=dev)
    pi = torch.empty(chunk * ngrp + chunk + 1, dtype=torch.int32, device=dev)
    st = stream_ptr(dev)
    scale = 1.0 / (g.H * g.W)
    # near-window search (+ the 3x3 neighbourhood of the peak when asked): the full map (T2) is
    # a device-side fallback that normally never runs (mc_xc_correlate_argmax)
    fused = FUSED_SEARCH and planmod.native_rows(g) and planmod.native_height(g.H) and g.H >= 1024
    scatter = shift_rows is not None and fused and chunk >= npairs  # one call zeroes + fills the table
    if shift_rows is not None and not scatter:
        table, shifts = shifts, torch.empty((npairs, 2), dtype=torch.float32, device=dev)
    if fused:
        T2n = torch.empty((chunk, g.nkx, 2 * lib.mc_xc_near_rows(g), 2), dtype=torch.float32, device=dev)
    for a in range(0, npairs, chunk):
        n = min(chunk, npairs - a)
        if fused:
            check(lib.mc_xc_correlate_argmax(ptr(S_cur), ptr(cur_idx[a : a + n]), ptr(S_ref),
                                             ptr(ref_idx[a : a + n]), ptr(T2), ptr(T2n), ptr(pv), ptr(pi),
                                             ptr(peaks[a : a + n]),
                                             ptr(shifts) if scatter else ptr(shifts[a : a + n]),
                                             ptr(shift_rows) if scatter else None,
                                             n_shift_rows if scatter else 0,
                                             ptr(nb[a : a + n]) if want_nbhd else None, ptr(pl.tw_col),
                                             ptr(pl.tw_row), scale, n, g, st), "mc_xc_correlate_argmax")
            continue
        if planmod.native_height(g.H):
            check(lib.mc_xc_cols_inverse(ptr(S_cur), ptr(cur_idx[a : a + n]), ptr(S_ref),
                                         ptr(ref_idx[a : a + n]), ptr(T2), ptr(pl.tw_col), scale, n, g,
                                         st), "mc_xc_cols_inverse")
        else:
            line, _ = planmod.line_plan(g.H, +1, dev)
            check(lib.mc_xcg_cols_inverse(ptr(S_cur), ptr(cur_idx[a : a + n]), ptr(S_ref),
                                          ptr(ref_idx[a : a + n]), None, ptr(T2), line, scale, n, g, st),
                  "mc_xcg_cols_inverse")
        if planmod.native_rows(g):
            check(lib.mc_xc_rows_inverse_argmax(ptr(T2), ptr(pv), ptr(pi), ptr(peaks[a : a + n]),
                                                ptr(shifts[a : a + n]), ptr(pl.tw_row), n, g, st),
                  "mc_xc_rows_inverse_argmax")
        else:
            line, _ = planmod.line_plan(planmod.row_line_length(g.W), +1, dev)
            check(lib.mc_xcg_rows_inverse(ptr(T2), ptr(pv), ptr(pi), ptr(peaks[a : a + n]),
                                          ptr(shifts[a : a + n]), None, None, 0, ptr(pl.tw_row), line, n,
                                          g, st), "mc_xcg_rows_inverse")
        if want_nbhd and planmod.native_rows(g):
            check(lib.mc_xc_peak_neighbourhood(ptr(T2), ptr(peaks[a : a + n]), ptr(nb[a : a + n]),
                                               ptr(pl.tw_row), n, g, st),
                  "mc_xc_peak_neighbourhood")
        elif want_nbhd:  # any other width: nine direct sums over the kept columns
            check(lib.mc_xcg_peak_neighbourhood(ptr(T2), ptr(peaks[a : a + n]), ptr(nb[a : a + n]), n, g, st),
                  "mc_xcg_peak_neighbourhood")
    if shift_rows is not None and not scatter:  # general path: scatter with torch
        table.zero_()
        table[shift_rows.long()] = shifts
        shifts = table
    return peaks, shifts, nb


# ------------------------------------------------------------------ a1: global shifts


def mask_spectrum(pl, dev):
    """Pruned spectrum of the plan's mask, (nkx, nky, 2): K1+K2 of an all-ones window.
    Built once per plan; used by the fused-statistics path (normalisation by linearity)."""
    if getattr(pl, "mhat", None) is None:
        g = pl.geom
        ones = torch.ones((g.H, g.W), dtype=torch.float32, device=dev)
        off = torch.zeros(1, dtype=torch.int64, device=dev)
        pl.mhat = _forward_spectra(ones, off, g.W, None, pl, None, use_filter=False)[0].contiguous()
    return pl.mhat


def _global_spectra(img, pl, split=False):
    """Filtered pruned spectra of all frames, (t, nkx, nky, 2), with normalize_image's
    statistics gathered inside K1 whenever the central box lies in the region K1 reads
    (always for near-square frames); otherwise a separate statistics pass."""
    lib = _lib.load()
    t, h, w = img.shape
    dev, g = img.device, pl.geom
    hl, hu, wl, wu = int(0.25 * h), int(0.75 * h), int(0.25 * w), int(0.75 * w)
    job_off = _cached(("frame_off", str(dev), t, h, w),
                      lambda: torch.arange(t, device=dev, dtype=torch.int64) * (h * w))
    fused = (planmod.native_rows(g) and planmod.native_height(g.H) and hl >= g.y0 and hu <= g.y0 + g.ny and wl >= g.x0
             and wu <= g.x1 and wl % 2 == 0 and wu % 2 == 0 and hu > hl and wu > wl)
    # fp16 frames are read as they are by the wave-per-row K1 (4096 columns); any other shape widens once
    half_ok = (img.dtype == torch.float16 and fused and w == 4096 and g.nkx <= 512 and g.ny % 8 == 0
               and wl % 256 == 0 and wu % 256 == 0 and img.data_ptr() % 16 == 0)
    if img.dtype != torch.float32 and not half_ok:
        img = img.float()
    if not fused:
        return _forward_spectra(img, job_off, w, None, pl, central_box_stats(img))
    st = stream_ptr(dev)
    mhat = mask_spectrum(pl, dev)
    # provisional mean m0 keeps the linear fix-up free of cancellation; any value near the
    # true mean does, so one row of frame 0's box is enough (one small workgroup)
    acc = torch.empty(128, dtype=torch.float64, device=dev)  # 64 x {sum, sumsq}
    m0 = torch.empty(3, dtype=torch.float32, device=dev)
    check(lib.mc_xc_provisional_mean_t(C.c_void_p(img.data_ptr() + img.element_size() * (hl * w + wl)),
                                       storage_of(img), wu - wl, ptr(m0), st), "mc_xc_provisional_mean")
    fix = torch.empty(2, dtype=torch.float32, device=dev)
    out3 = torch.empty(3, dtype=torch.float32, device=dev)
    T1 = torch.empty((t, g.nkx, g.ny, 2), dtype=torch.float32, device=dev)
    S = torch.empty((t, g.nkx, g.nky, 2), dtype=torch.float32, device=dev)
    try:
        check(lib.mc_xc_rows_forward_stats_t(ptr(img), storage_of(img), ptr(job_off), w, ptr(pl.mask), ptr(m0),
                                             ptr(T1), ptr(pl.tw_row), t, g, hl, hu, wl, wu, ptr(acc), ptr(fix),
                                             ptr(out3), ptr(_box_chords(pl, hl, hu, wl, wu)), st),
              "mc_xc_rows_forward_stats")
    except _lib.McorrUnsupported:
        # the C side decides which shapes read fp16 natively (row engine, alignment, geometry); the
        # predicate above is only a shortcut -- on disagreement widen once, as the warps do
        if img.dtype == torch.float32:
            raise
        del T1, S
        return _global_spectra(img.float(), pl, split)
    if AFTER_K1_HOOK is not None and not HOOK_AFTER_K2:
        AFTER_K1_HOOK()
    if split:  # the caller enqueues the column pass itself, possibly on another stream (global_stage_b)
        return ("k2_pending", T1, S, fix, mhat)
    check(lib.mc_xc_cols_forward_fix(ptr(T1), ptr(pl.filt), ptr(S), ptr(pl.tw_col), t, g, ptr(fix),
                                     ptr(mhat), st), "mc_xc_cols_forward_fix")
    if AFTER_K1_HOOK is not None and HOOK_AFTER_K2:
        AFTER_K1_HOOK()
    return S


def _os_env_flag(name, default):
    import os

    v = os.environ.get(name)
    return default if v is None else v not in ("0", "false", "no")


def _box_chords(pl, hl, hu, wl, wu):
    """The plan's per-row chord table if the statistics box lies inside the chords (it does for
    the reference's circular mask: the box corner is at 0.35 n from the centre, the soft edge ends
    at 0.375 n), else None: K1 then clamps to the support box only."""
    if pl.chord is None or not USE_ROW_CHORDS:
        return None
    key = ("chord_ok", id(pl), hl, hu, wl, wu)

    def build():
        c = pl.chord[hl:hu]
        return bool((c[:, 0] <= wl).all()) and bool((c[:, 1] + 4 >= wu).all())

    return pl.chord if _cached(key, build) else None


USE_ROW_CHORDS = _os_env_flag("MC_ROW_CHORDS", True)


def global_shifts(img, reference_frame, pixel_spacing, b_factor, frequency_range):
    """Integer-pixel (t,2) shifts of every frame against `reference_frame`
    (estimate_motion_xc.py:57-123); the reference frame's row is exactly zero.
    `reference_frame` follows the reference's Python indexing (xc.py:101,107): a negative value
    selects from the end but never equals a loop index, so that frame is NOT skipped (it is
    correlated with itself); anything outside [-t, t) raises IndexError."""
    t, h, w = img.shape
    dev = img.device
    pl = planmod.get_xc_plan(h, w, pixel_spacing, b_factor, frequency_range, dev)
    S = _global_spectra(img, pl)
    return _shifts_from_spectra(S, t, reference_frame, pl)


def global_stage_a(img, pixel_spacing, b_factor, frequency_range):
    """First stage of global_shifts for the three-stage movie pipeline: the plan and K1 (the HBM-bound row
    pass over the frames) on the current stream.  Returns an opaque state for ``global_stage_b``; its
    tensors must be made known to the stream stage b runs on (``stage_tensors``)."""
    t, h, w = img.shape
    pl = planmod.get_xc_plan(h, w, pixel_spacing, b_factor, frequency_range, img.device)
    return (t, pl, _global_spectra(img, pl, split=True))


def stage_tensors(state):
    r = state[2]
    return [x for x in (r[1:] if isinstance(r, tuple) else (r,)) if isinstance(x, torch.Tensor)]


def global_stage_b(state, reference_frame):
    """Second stage: the column pass (when stage a left it pending), K3, K4 -> (t, 2) shifts, on the current stream."""
    lib = _lib.load()
    t, pl, r = state
    if isinstance(r, tuple):
        _, T1, S, fix, mhat = r
        g = pl.geom
        check(lib.mc_xc_cols_forward_fix(ptr(T1), ptr(pl.filt), ptr(S), ptr(pl.tw_col), t, g, ptr(fix),
                                         ptr(mhat), stream_ptr(S.device)), "mc_xc_cols_forward_fix")
    else:
        S = r
    return _shifts_from_spectra(S, t, reference_frame, pl)


def _shifts_from_spectra(S, t, reference_frame, pl):
    """K3 + K4 of the global estimate: every frame's filtered spectrum against the reference frame's."""
    dev = S.device
    ref = _lib.normalize_frame_index(reference_frame, t)
    skip = int(reference_frame) >= 0
    cur = [f for f in range(t) if not (skip and f == ref)]
    if not cur:
        return torch.zeros((t, 2), dtype=torch.float32, device=dev)
    cur_idx, ref_idx = _cached(
        ("global_pairs", str(dev), t, ref, skip),
        lambda: (_i32(cur, dev), _i32([ref] * len(cur), dev)))
    # pair p = frame cur[p]: its shift goes to row cur[p] of the (t, 2) table; the reference
    # frame's row is written by nobody and stays exactly zero
    _, shifts, _ = _peaks(S, cur_idx, S, ref_idx, pl, want_nbhd=False, shift_rows=cur_idx, n_shift_rows=t)
    return shifts


# ------------------------------------------------------------------ a8: patch field


def patch_field(img, stats, pixel_spacing, reference_frame, reference_strategy, b_factor,
                frequency_range, patch_sidelength, sub_pixel_refinement, temporal_smoothing,
                smoothing_window_size, field0, outlier_rejection, outlier_threshold):
    """Per-patch shift field (2,t,gh,gw) in Angstrom, mean-subtracted, plus the patch
    centres (estimate_motion_xc.py:250-411).  `img` is the (already pre-corrected)
    stack; `stats` = central-box statistics to normalise with inside K1, or None when
    `img` is already normalised.  `field0` = prior field resampled to (2,t,gh,gw) or None."""
    lib = _lib.load()
    t, h, w = img.shape
    dev = img.device
    p = int(patch_sidelength)
    if reference_strategy not in ("middle_frame", "mean_except_current"):
        raise ValueError(f"Unknown reference_strategy: {reference_strategy}")
    if p > h or p > w:
        raise ValueError(f"patch_sidelength {p} exceeds the frame size {h}x{w}")
    pl = planmod.get_xc_plan(p, p, pixel_spacing, b_factor, frequency_range, dev)
    g = pl.geom
    if img.dtype != torch.float32 and not (g.W == 1024 and g.nkx <= 128 and g.ny % 8 == 0):
        # fp16 frames are read natively by the 1024-px patch kernel only (BASELINE C5); any other
        # patch size goes through the workgroup / chirp-z kernels on a widened copy
        if stats is not None:
            pass  # the statistics were taken from the fp16 bytes: identical values
        img = img.float()
    cy, cx = lattice.patch_grid_centers(t, h, w, p)
    gh, gw = len(cy), len(cx)
    npatch = gh * gw
    origin = ((cy[:, None] - p // 2) * w + (cx[None, :] - p // 2)).reshape(-1)  # (npatch,)
    # the memo key is the caller's raw value (a negative key is an entry of its own in the
    # reference's memo, patch_grid/_patch_grid.py:283-294); the data come from the wrapped index
    ref_key = int(reference_frame)
    reference_frame = _lib.normalize_frame_index(ref_key, t)
    ref_expo, cur_expo, processed, ref_read = lattice.mask_schedule(t, reference_strategy, ref_key,
                                                                    with_ref_reads=True)
    field = torch.zeros((2, t, gh, gw), dtype=torch.float32, device=dev) if field0 is None \
        else field0.contiguous().clone()
    st = stream_ptr(dev)

    def jobs(frames, expos):
        off = (np.asarray(frames, dtype=np.int64)[:, None] * (h * w) + origin[None, :]).reshape(-1)
        ex = np.repeat(np.asarray(expos, dtype=np.int32), npatch)
        return _i64(off, dev), _i32(ex, dev)

    nproc = len(processed)
    if nproc > 0:
        if reference_strategy == "mean_except_current":
            if t < 2:
                raise ValueError("mean_except_current needs at least 2 frames")
            if ref_expo.max() > 1 or cur_expo.max() > 0:
                raise NotImplementedError("unexpected mask schedule")
            off, ex1 = jobs(range(t), [1] * t)
            try:
                U, V = _forward_spectra(img, off, w, ex1, pl, stats, job_expo_b=ex1 * 2, min_expo=1)
            except _lib.McorrUnsupported:
                if img.dtype == torch.float32:
                    raise
                img = img.float()  # the C side has no fp16 kernel for this case after all: widen once
                U, V = _forward_spectra(img, off, w, ex1, pl, stats, job_expo_b=ex1 * 2, min_expo=1)
            sp, si, sr = lattice.leave_one_out_schedule(ref_expo)
            sp, si, sr = _i32(sp, dev), _i32(si, dev), torch.as_tensor(sr, device=dev)
            REF = torch.empty_like(U)
            check(lib.mc_xc_ref_mean_except_current(ptr(U), ptr(V), ptr(sp), ptr(si), ptr(sr), ptr(REF),
                                                    t, npatch, g.nkx * g.nky, 1.0 / (t - 1), st),
                  "mc_xc_ref_mean_except_current")
            del V
            S_cur, S_ref = U, REF
        else:
            exl = [int(cur_expo[f]) + 1 for f in processed]
            off, ex = jobs(processed, exl)
            try:
                S_cur = _forward_spectra(img, off, w, ex, pl, stats, min_expo=min(exl))
            except _lib.McorrUnsupported:
                if img.dtype == torch.float32:
                    raise
                img = img.float()
                S_cur = _forward_spectra(img, off, w, ex, pl, stats, min_expo=min(exl))
            exl = [int(ref_read[f]) + 1 for f in processed]
            off, ex = jobs([reference_frame] * nproc, exl)
            S_ref = _forward_spectra(img, off, w, ex, pl, stats, min_expo=min(exl))
        pair_idx = torch.arange(nproc * npatch, device=dev, dtype=torch.int32)
        peaks, _, nb = _peaks(S_cur, pair_idx, S_ref, pair_idx, pl, want_nbhd=sub_pixel_refinement)
        flags = (1 if sub_pixel_refinement else 0) | (2 if outlier_rejection else 0)
        check(lib.mc_field_accumulate(ptr(peaks), ptr(nb), ptr(_i32(processed, dev)), nproc, npatch,
                                      p, t, float(pixel_spacing), float(outlier_threshold), flags,
                                      ptr(field), st), "mc_field_accumulate")
    window = 0
    if temporal_smoothing:
        window = int(smoothing_window_size)
        if window % 2 == 0:
            window += 1
        window = min(window, t)
        if window < 3:
            window = 0  # (an even window = t for even t < window|1 is what scipy gets, xc.py:506-529)
    out = torch.empty_like(field)
    check(lib.mc_field_smooth_center(ptr(field), ptr(out), t, npatch, window, 1, st),
          "mc_field_smooth_center")
    return out, lattice.centers_tensor(t, cy, cx)


# ------------------------------------------------------------------ a14/a16: spline lattice


def spline_lattice(field, ut, uy, ux, grid_type):
    """Evaluate the (c,nt,nh,nw) spline grid `field` on the tensor-product lattice
    ut x uy x ux (CPU float32 coordinate vectors in [0,1]) -> (c, NT, NY, NX)."""
    lib = _lib.load()
    dev = field.device
    c, nt, nh, nw = field.shape
    tabs = []
    for n, u in ((nt, ut), (nh, uy), (nw, ux)):
        u = u.detach().to(torch.float32).cpu().contiguous()

        def build(n=n, u=u):
            idx, wts = spline.axis_taps(n, u, grid_type)
            return idx.to(dev), wts.to(dev), int(u.numel())

        tabs.append(_cached(("taps", str(dev), n, grid_type, u.numpy().tobytes()), build))
    out = torch.empty((c, tabs[0][2], tabs[1][2], tabs[2][2]), dtype=torch.float32, device=dev)
    f = field.contiguous()
    check(lib.mc_spline_lattice(ptr(f), c, nt, nh, nw, ptr(tabs[0][0]), ptr(tabs[0][1]), tabs[0][2],
                                ptr(tabs[1][0]), ptr(tabs[1][1]), tabs[1][2], ptr(tabs[2][0]),
                                ptr(tabs[2][1]), tabs[2][2], ptr(out), stream_ptr(dev)),
          "mc_spline_lattice")
    return out


def spline_points(field, tyx, grid_type):
    """Evaluate the (c,nt,nh,nw) spline grid at (n, 3) points (t, y, x in [0,1], CPU float32) ->
    (n, c) on the device: ONE launch (tap tables per point built on the host)."""
    lib = _lib.load()
    dev = field.device
    c, nt, nh, nw = field.shape
    pts = tyx.detach().to(torch.float32).cpu().reshape(-1, 3)
    n = int(pts.shape[0])
    out = torch.empty((n, c), dtype=torch.float32, device=dev)
    if n == 0:
        return out
    tabs = [spline.axis_taps(size, pts[:, a].contiguous(), grid_type) for a, size in enumerate((nt, nh, nw))]
    dt = [(i.to(dev), w.to(dev)) for i, w in tabs]
    f = field.contiguous()
    check(lib.mc_spline_points(ptr(f), c, nt, nh, nw, ptr(dt[0][0]), ptr(dt[0][1]), ptr(dt[1][0]), ptr(dt[1][1]),
                               ptr(dt[2][0]), ptr(dt[2][1]), n, ptr(out), stream_ptr(dev)), "mc_spline_points")
    return out


def frame_lattices(field, t, grid_type):
    """(t, 2, 10gh, 10gw) Angstrom lattices, one per frame time linspace(0,1,t)
    (correct_motion.py:57,67-72)."""
    _, _, gh, gw = field.shape
    lin = lambda n: _cached(("linspace", n), lambda: torch.linspace(0, 1, steps=n))
    lat = spline_lattice(field, lin(t), lin(10 * gh), lin(10 * gw), grid_type)
    return lat.permute(1, 0, 2, 3).contiguous()


# ------------------------------------------------------------------ a15/a17/a18: warp


RIGID_KERNEL_HOOK = None  # callable(fn) -> calls fn(); set by bench.py to time warp_rigid_dma alone
HOOK_AFTER_K2 = False
AFTER_K1_HOOK = None  # callable() invoked right after the global estimate's K1 has been enqueued (pipeline schedules)


def rigid_tables(img, lattices, pixel_spacing):
    """The per-frame weight tables of the rigid warp (rigid_base + rigid_weights: phase 1 of
    mc_warp_rigid_phase_t) enqueued on the CURRENT stream -> an opaque handle for
    ``warp(..., rigid=True, tables=handle)``.  The movie pipeline builds them on the estimator's
    stream, so that the warp stream carries nothing but the resampling launch."""
    lib = _lib.load()
    t, h, w = img.shape
    dev = img.device
    shifts_px = (lattices[:, :, 0, 0] / pixel_spacing).contiguous()  # shifts_angstroms / ps
    nbytes = C.c_int64(0)
    check(lib.mc_warp_rigid_scratch_bytes(t, h, w, C.byref(nbytes)), "mc_warp_rigid_scratch_bytes")
    scratch = torch.empty((nbytes.value + 3) // 4, dtype=torch.float32, device=dev)
    # phase 1 never touches the frames (only their geometry matters): tagged fp32 so that fp16 stacks of
    # any row length get their tables here, whichever kernel resamples them later
    check(lib.mc_warp_rigid_phase_t(ptr(img), STORE_F32, t, h, w, ptr(shifts_px), ptr(scratch), None, None, 1,
                                    stream_ptr(dev)), "mc_warp_rigid_phase")
    return shifts_px, scratch


def rigid_tables_from_shifts(shifts, img_shape, pixel_spacing, grid_type):
    """The movie pipeline's tail in two launches (mc_rigid_tables_from_shifts): (t,2) px shifts of the global
    estimate -> ((2,t,1,1) Angstrom field, handle for ``warp(..., rigid=True, tables=handle)``); the same
    numbers as image_shifts_to_deformation_field + frame_lattices + rigid_tables, bit for bit."""
    lib = _lib.load()
    t, h, w = img_shape
    dev = shifts.device

    def build():
        lin = torch.linspace(0, 1, steps=t)
        it, wt = spline.axis_taps(t, lin, grid_type)
        _, w1 = spline.axis_taps(1, torch.linspace(0, 1, steps=10)[:1], grid_type)  # lattice point 0 of a 1-sample axis
        return it.to(dev), wt.to(dev), w1[0].contiguous().to(dev)

    idx_t, w_t, w1 = _cached(("rigid_tail_taps", str(dev), t, grid_type), build)
    field = torch.empty((2, t), dtype=torch.float32, device=dev)
    shifts_px = torch.empty((t, 2), dtype=torch.float32, device=dev)
    nbytes = C.c_int64(0)
    check(lib.mc_warp_rigid_scratch_bytes(t, h, w, C.byref(nbytes)), "mc_warp_rigid_scratch_bytes")
    scratch = torch.empty((nbytes.value + 3) // 4, dtype=torch.float32, device=dev)
    check(lib.mc_rigid_tables_from_shifts(ptr(shifts.contiguous()), float(pixel_spacing), ptr(idx_t), ptr(w_t), ptr(w1),
                                          ptr(w1), t, h, w, ptr(field), ptr(shifts_px), ptr(scratch), stream_ptr(dev)),
          "mc_rigid_tables_from_shifts")
    return field[:, :, None, None], (shifts_px, scratch)


def warp(img, lattices, pixel_spacing, want_frames=True, want_sum=False, rigid=False, tables=None):
    """Resample every frame through its lattice; returns (frames or None, sum or None).
    rigid=True: the lattices come from a (2,nt,1,1) field, i.e. one shift per frame ->
    the separable rigid kernel (`tables`: the handle of an earlier ``rigid_tables`` call for
    the same stack and lattices; `lattices` may then be None)."""
    lib = _lib.load()
    t, h, w = img.shape
    dev = img.device
    if rigid and img.dtype == torch.float16 and (w % 8 or img.data_ptr() % 16):
        img = img.float()  # fp16 rows that are not whole 8-sample units: widened once
    frames = torch.empty((t, h, w), dtype=torch.float32, device=dev) if want_frames else None
    total = torch.empty((h, w), dtype=torch.float32, device=dev) if want_sum else None  # the kernels store it
    nbytes = C.c_int64(0)
    if rigid:
        if tables is None and RIGID_KERNEL_HOOK is not None:
            tables = rigid_tables(img, lattices, pixel_spacing)
        if tables is None:
            shifts_px = (lattices[:, :, 0, 0] / pixel_spacing).contiguous()  # shifts_angstroms / ps
            check(lib.mc_warp_rigid_scratch_bytes(t, h, w, C.byref(nbytes)), "mc_warp_rigid_scratch_bytes")
            scratch = torch.empty((nbytes.value + 3) // 4, dtype=torch.float32, device=dev)
            args = (ptr(img), storage_of(img), t, h, w, ptr(shifts_px), ptr(scratch), ptr(frames), ptr(total))
            check(lib.mc_warp_rigid_phase_t(*args, 0, stream_ptr(dev)), "mc_warp_rigid")
            return frames, total
        shifts_px, scratch = tables
        args = (ptr(img), storage_of(img), t, h, w, ptr(shifts_px), ptr(scratch), ptr(frames), ptr(total))
        run = lambda: check(lib.mc_warp_rigid_phase_t(*args, 2, stream_ptr(dev)), "mc_warp_rigid_phase")
        if RIGID_KERNEL_HOOK is None:
            run()
        else:  # instrumentation: the hook brackets the resampling kernel alone (bench.py)
            RIGID_KERNEL_HOOK(run)
        return frames, total
    _, _, GH, GW = lattices.shape
    check(lib.mc_warp_scratch_bytes(t, h, w, GH, GW, C.byref(nbytes)), "mc_warp_scratch_bytes")
    scratch = torch.empty((nbytes.value + 3) // 4, dtype=torch.float32, device=dev)
    rc = lib.mc_warp_frames_t(ptr(img), storage_of(img), t, h, w, ptr(lattices), GH, GW, float(pixel_spacing),
                              ptr(scratch), ptr(frames), ptr(total), stream_ptr(dev))
    if rc == -2 and img.dtype != torch.float32:
        # fp16 frames outside the LDS-staged kernel's shapes (row length not a multiple of 8, dense
        # lattice): widen once and take the fp32 kernels
        rc = lib.mc_warp_frames(ptr(img.float()), t, h, w, ptr(lattices), GH, GW, float(pixel_spacing),
                                ptr(scratch), ptr(frames), ptr(total), stream_ptr(dev))
    check(rc, "mc_warp_frames")
    return frames, total


def pixel_shifts(lattice, h, w, pixel_spacing):
    """(h,w,2) px shifts from one (2,GH,GW) Angstrom lattice (correct_motion.py:132-185)."""
    lib = _lib.load()
    dev = lattice.device
    _, GH, GW = lattice.shape
    nbytes = C.c_int64(0)
    check(lib.mc_warp_scratch_bytes(1, h, w, GH, GW, C.byref(nbytes)), "mc_warp_scratch_bytes")
    scratch = torch.empty((nbytes.value + 3) // 4, dtype=torch.float32, device=dev)
    out = torch.empty((h, w, 2), dtype=torch.float32, device=dev)
    check(lib.mc_pixel_shifts(ptr(lattice.contiguous()), GH, GW, h, w, float(pixel_spacing),
                              ptr(scratch), ptr(out), stream_ptr(dev)), "mc_pixel_shifts")
    return out


def pixel_shifts_at(lattice, h, w, pixel_spacing, coords):
    """get_pixel_shifts at arbitrary (..., 2) yx pixel coordinates (the reference's `pixel_grid`
    argument, correct_motion.py:167-168) -> (..., 2) px."""
    lib = _lib.load()
    dev = lattice.device
    _, GH, GW = lattice.shape
    pts = coords.reshape(-1, 2).contiguous()
    out = torch.empty_like(pts)
    if pts.shape[0]:
        check(lib.mc_pixel_shifts_at(ptr(lattice.contiguous()), GH, GW, h, w, float(pixel_spacing), ptr(pts),
                                     pts.shape[0], ptr(out), stream_ptr(dev)), "mc_pixel_shifts_at")
    return out.reshape(coords.shape)


# ------------------------------------------------------------------ a19: Fourier shift


POLYPHASE_FOURIER_SHIFT = False  # tests: force the x-polyphase form on frames that do not need it


def _fourier_shift_polyphase(img, shifts):
    """fourier_shift for frames whose full spectrum does not fit one row line (csrc/polyphase.hip):
    even and odd columns are transformed as two (h, w/2) frames, one pointwise pass applies the
    radix-2 butterfly + phase ramp + inverse butterfly, the halves go back and are interleaved."""
    lib = _lib.load()
    t, h, w = img.shape
    if w % 4:
        raise NotImplementedError(f"frames of {w} columns: the polyphase Fourier shift needs a width divisible by 4")
    dev = img.device
    w2 = w // 2
    g = planmod.full_geometry(h, w2)
    tw_row, tw_col = planmod.get_twiddles(w2, dev), planmod.get_twiddles(h, dev)
    out = torch.empty_like(img)
    per_frame = 2 * g.nkx * g.H * 8
    chunk = max(1, min(t, WORKSPACE_BYTES // (2 * per_frame)))
    T1 = torch.empty((2 * chunk, g.nkx, g.ny, 2), dtype=torch.float32, device=dev)
    S = torch.empty((2 * chunk, g.nkx, g.nky, 2), dtype=torch.float32, device=dev)
    st = stream_ptr(dev)
    shifts = shifts.to(dev, torch.float32).contiguous()
    zero = torch.zeros((2 * chunk, 2), dtype=torch.float32, device=dev)
    for a in range(0, t, chunk):
        n = min(chunk, t - a)
        sub = torch.cat([img[a:a + n, :, 0::2], img[a:a + n, :, 1::2]], dim=0).contiguous()  # (2n, h, w2)
        off = torch.arange(2 * n, device=dev, dtype=torch.int64) * (h * w2)
        idx = torch.arange(2 * n, device=dev, dtype=torch.int32)
        check(_k1(lib, g, dev, sub, off, w2, None, None, None, T1, tw_row, 2 * n, st), "xc rows forward")
        check(_k2(lib, g, dev, T1, None, S, tw_col, 2 * n, st), "xc cols forward")
        check(lib.mc_polyphase_fourier_shift(ptr(S), ptr(shifts[a:a + n]), n, g.nkx, h, w, st),
              "mc_polyphase_fourier_shift")
        if planmod.native_height(g.H):
            check(lib.mc_fourier_shift_cols_inverse(ptr(S), ptr(idx), ptr(zero), ptr(T1), ptr(tw_col),
                                                    1.0 / (h * w2), 2 * n, g, st), "mc_fourier_shift_cols_inverse")
        else:
            line, _ = planmod.line_plan(g.H, +1, dev)
            check(lib.mc_xcg_cols_inverse(ptr(S), ptr(idx), None, None, ptr(zero), ptr(T1), line,
                                          1.0 / (h * w2), 2 * n, g, st), "mc_xcg_cols_inverse")
        res = torch.empty_like(sub)
        if planmod.native_rows(g):
            check(lib.mc_xc_rows_inverse_store(ptr(T1), ptr(res), ptr(off), w2, ptr(tw_row), 2 * n, g, st),
                  "mc_xc_rows_inverse_store")
        else:
            line, _ = planmod.line_plan(planmod.row_line_length(g.W), +1, dev)
            check(lib.mc_xcg_rows_inverse(ptr(T1), None, None, None, None, ptr(res), ptr(off), w2,
                                          ptr(tw_row), line, 2 * n, g, st), "mc_xcg_rows_inverse")
        out[a:a + n, :, 0::2] = res[:n]
        out[a:a + n, :, 1::2] = res[n:]
    return out


FULL_ROW_MAJOR = True  # tests: False forces the pruned engine's transposed layout on power-of-two frames
DOSE_COLUMN_MAJOR = True  # tests / A-B: False feeds the exposure-weighted pass from the row-major spectra


def _full_row_major_ok(h, w):
    """Frames the row-major full-spectrum kernels (csrc/full_fft.hip) take: power-of-two rows and
    columns, and the K3 detector's 5760 / 11520 columns and 4092 / 8184 rows (mixed radix)."""
    pow2 = lambda n: n > 0 and (n & (n - 1)) == 0
    rows_ok = (pow2(w) and 64 <= w <= 8192) or w in (5760, 11520)
    cols_ok = (pow2(h) and 256 <= h <= 4096) or h in (4092, 8184)
    return FULL_ROW_MAJOR and rows_ok and cols_ok


def _fourier_shift_row_major(img, shifts):
    """fourier_shift on power-of-two frames: rows forward -> (columns forward, phase ramp, columns
    inverse) in one in-place kernel -> rows inverse, the spectrum row-major throughout."""
    lib = _lib.load()
    t, h, w = img.shape
    dev = img.device
    pitch = lib.mc_full_spectrum_pitch(w)
    tw_row, tw_col = planmod.get_twiddles(w, dev), planmod.get_twiddles(h, dev)
    out = torch.empty_like(img)
    per_frame = h * pitch * 8
    chunk = max(1, min(t, WORKSPACE_BYTES // per_frame))
    S = torch.empty((chunk, h, pitch, 2), dtype=torch.float32, device=dev)
    st = stream_ptr(dev)
    shifts = shifts.to(dev, torch.float32).contiguous()
    for a in range(0, t, chunk):
        n = min(chunk, t - a)
        off = torch.arange(a, a + n, device=dev, dtype=torch.int64) * (h * w)
        check(lib.mc_full_rows_forward(ptr(img), ptr(off), w, ptr(S), ptr(tw_row), n, h, w, pitch, st),
              "mc_full_rows_forward")
        check(lib.mc_full_cols_shift(ptr(S), ptr(shifts[a:a + n]), ptr(tw_col), 1.0 / (h * w), n, h, w, pitch, st),
              "mc_full_cols_shift")
        check(lib.mc_full_rows_inverse(ptr(S), ptr(out), ptr(off), w, ptr(tw_row), n, h, w, pitch, st),
              "mc_full_rows_inverse")
    return out


def _dose_weighted_sum_row_major(img, pixel_spacing, dose_per_frame, pre_exposure, voltage, frames_of=None,
                                 shape=None):
    """dose_weighted_sum on the row-major kernels: rows forward per chunk, the exposure-weighted
    accumulation inside the forward column pass (frame loop in registers), one inverse per movie.
    `frames_of(a, n)` -> the (n, h, w) fp32 frames a .. a+n-1 (default: slices of `img`); a caller
    that produces the frames on the fly (motion_correct_sum: warp a chunk, transform it, drop it)
    never holds more than one chunk of corrected frames."""
    lib = _lib.load()
    t, h, w = img.shape if shape is None else shape
    dev = img.device
    pitch = lib.mc_full_spectrum_pitch(w)
    tw_row, tw_col = planmod.get_twiddles(w, dev), planmod.get_twiddles(h, dev)
    per_frame = h * pitch * 8
    # 4096 / 4092 rows: the exposure-weighted pass reads a column-major copy of the chunk's spectra
    # (mc_full_transpose): contiguous columns instead of 8 bytes of every 128-byte line -- 4.4 -> 3.6 ms
    # per 40 x 4096^2, 8.6 -> 7.4 ms per 40 x 4092 x 5760 with the copy's own read + write pass paid.
    # Not for 8184 rows (14.4 -> 15.2 ms per 12 frames: that column kernel is bound by its radix-31
    # pass on 512-thread workgroups, not by how it is fed).
    colmajor = DOSE_COLUMN_MAJOR and h in (4096, 4092)
    chunk = max(1, min(t, WORKSPACE_BYTES // ((2 if colmajor else 1) * per_frame)))
    S = torch.empty((chunk, h, pitch, 2), dtype=torch.float32, device=dev)
    ST = torch.empty((chunk, w // 2 + 1, h, 2), dtype=torch.float32, device=dev) if colmajor else None
    A = torch.empty((h, pitch, 2), dtype=torch.float32, device=dev)
    st = stream_ptr(dev)
    for a in range(0, t, chunk):
        n = min(chunk, t - a)
        if frames_of is None:
            src, first = img, a
        else:
            src, first = frames_of(a, n), 0
        off = torch.arange(first, first + n, device=dev, dtype=torch.int64) * (h * w)
        check(lib.mc_full_rows_forward(ptr(src), ptr(off), w, ptr(S), ptr(tw_row), n, h, w, pitch, st),
              "mc_full_rows_forward")
        dose_args = (n, a, t, ptr(A), ptr(tw_col), h, w, pitch, float(pixel_spacing), float(pre_exposure),
                     float(dose_per_frame), float(voltage), 1 if a == 0 else 0, 1 if a + n >= t else 0,
                     1.0 / (h * w), st)
        if colmajor:
            check(lib.mc_full_transpose(ptr(S), ptr(ST), n, h, w, pitch, st), "mc_full_transpose")
            check(lib.mc_full_cols_dose_cm(ptr(ST), *dose_args), "mc_full_cols_dose_cm")
        else:
            check(lib.mc_full_cols_dose(ptr(S), *dose_args), "mc_full_cols_dose")
        del src
    out = torch.empty((h, w), dtype=torch.float32, device=dev)
    off0 = torch.zeros(1, device=dev, dtype=torch.int64)
    check(lib.mc_full_rows_inverse(ptr(A), ptr(out), ptr(off0), w, ptr(tw_row), 1, h, w, pitch, st),
          "mc_full_rows_inverse")
    return out


def warp_dose_weighted_sum(img, lattices, pixel_spacing, rigid, dose_per_frame, pre_exposure, voltage):
    """correct_motion -> dose_weight -> sum (examples/ttMotion.py:318-351, 398) without the corrected
    movie in memory: on the row-major sizes (powers of two, the K3 formats) the frames are warped,
    transformed and weighted a chunk at a time (BASELINE C5: 60 x 8184 x 11520 -- 22.6 GB of
    corrected fp32 frames are never allocated); other sizes warp everything first."""
    t, h, w = img.shape
    if POLYPHASE_FOURIER_SHIFT or not _full_row_major_ok(h, w):
        frames, _ = warp(img, lattices, pixel_spacing, want_frames=True, want_sum=False, rigid=rigid)
        return dose_weighted_sum(frames, pixel_spacing, dose_per_frame, pre_exposure, voltage)

    def frames_of(a, n):
        return warp(img[a:a + n], lattices[a:a + n], pixel_spacing, want_frames=True, want_sum=False,
                    rigid=rigid)[0]

    return _dose_weighted_sum_row_major(img, pixel_spacing, dose_per_frame, pre_exposure, voltage,
                                        frames_of=frames_of, shape=(t, h, w))


def fourier_shift(img, shifts):
    """irfft2(rfft2(img) * exp(-2 pi i (fy sy + fx sx))) per frame; shifts (t,2) px
    (correct_motion.py:484-496)."""
    lib = _lib.load()
    t, h, w = img.shape
    dev = img.device
    if POLYPHASE_FOURIER_SHIFT:
        return _fourier_shift_polyphase(img, shifts)
    if _full_row_major_ok(h, w):
        return _fourier_shift_row_major(img, shifts)
    try:
        g = planmod.full_geometry(h, w)
    except NotImplementedError:
        if w % 4 == 0 and w <= 16384 and h <= 8192:
            return _fourier_shift_polyphase(img, shifts)  # too wide for one row line: even / odd columns
        raise
    tw_row, tw_col = planmod.get_twiddles(w, dev), planmod.get_twiddles(h, dev)
    out = torch.empty_like(img)
    per_frame = g.nkx * g.H * 8
    chunk = max(1, min(t, WORKSPACE_BYTES // (2 * per_frame)))
    T1 = torch.empty((chunk, g.nkx, g.ny, 2), dtype=torch.float32, device=dev)
    S = torch.empty((chunk, g.nkx, g.nky, 2), dtype=torch.float32, device=dev)
    idx = torch.arange(chunk, device=dev, dtype=torch.int32)
    st = stream_ptr(dev)
    shifts = shifts.to(dev, torch.float32).contiguous()
    for a in range(0, t, chunk):
        n = min(chunk, t - a)
        off = torch.arange(a, a + n, device=dev, dtype=torch.int64) * (h * w)
        check(_k1(lib, g, dev, img, off, w, None, None, None, T1, tw_row, n, st), "xc rows forward")
        check(_k2(lib, g, dev, T1, None, S, tw_col, n, st), "xc cols forward")
        # T1 is dead now and has the same footprint as T2: reuse it
        if planmod.native_height(g.H):
            check(lib.mc_fourier_shift_cols_inverse(ptr(S), ptr(idx), ptr(shifts[a : a + n]), ptr(T1),
                                                    ptr(tw_col), 1.0 / (h * w), n, g, st),
                  "mc_fourier_shift_cols_inverse")
        else:
            line, _ = planmod.line_plan(g.H, +1, dev)
            check(lib.mc_xcg_cols_inverse(ptr(S), ptr(idx), None, None, ptr(shifts[a : a + n]), ptr(T1),
                                          line, 1.0 / (h * w), n, g, st), "mc_xcg_cols_inverse")
        if planmod.native_rows(g):
            check(lib.mc_xc_rows_inverse_store(ptr(T1), ptr(out), ptr(off), w, ptr(tw_row), n, g, st),
                  "mc_xc_rows_inverse_store")
        else:
            line, _ = planmod.line_plan(planmod.row_line_length(g.W), +1, dev)
            check(lib.mc_xcg_rows_inverse(ptr(T1), None, None, None, None, ptr(out), ptr(off), w,
                                          ptr(tw_row), line, n, g, st), "mc_xcg_rows_inverse")
    return out


def _inverse_frames(lib, g, S, n, h, w, dev, st):
    """irfft2 of n full spectra S (n, nkx, H) -> (n, h, w) frames (zero-shift Fourier-shift path)."""
    tw_row, tw_col = planmod.get_twiddles(w, dev), planmod.get_twiddles(h, dev)
    out = torch.empty((n, h, w), dtype=torch.float32, device=dev)
    idx = torch.arange(n, device=dev, dtype=torch.int32)
    zero = torch.zeros((n, 2), device=dev, dtype=torch.float32)
    off = torch.arange(n, device=dev, dtype=torch.int64) * (h * w)
    T2 = torch.empty((n, g.nkx, g.H, 2), dtype=torch.float32, device=dev)
    if planmod.native_height(g.H):
        check(lib.mc_fourier_shift_cols_inverse(ptr(S), ptr(idx), ptr(zero), ptr(T2), ptr(tw_col),
                                                1.0 / (h * w), n, g, st), "mc_fourier_shift_cols_inverse")
    else:
        line, _ = planmod.line_plan(g.H, +1, dev)
        check(lib.mc_xcg_cols_inverse(ptr(S), ptr(idx), None, None, ptr(zero), ptr(T2), line,
                                      1.0 / (h * w), n, g, st), "mc_xcg_cols_inverse")
    if planmod.native_rows(g):
        check(lib.mc_xc_rows_inverse_store(ptr(T2), ptr(out), ptr(off), w, ptr(tw_row), n, g, st),
              "mc_xc_rows_inverse_store")
    else:
        line, _ = planmod.line_plan(planmod.row_line_length(g.W), +1, dev)
        check(lib.mc_xcg_rows_inverse(ptr(T2), None, None, None, None, ptr(out), ptr(off), w,
                                      ptr(tw_row), line, n, g, st), "mc_xcg_rows_inverse")
    return out


def _dose_weighted_sum_polyphase(img, pixel_spacing, dose_per_frame, pre_exposure, voltage):
    """dose_weighted_sum for frames too wide for one row line: even / odd columns (csrc/polyphase.hip)."""
    lib = _lib.load()
    t, h, w = img.shape
    if w % 4:
        raise NotImplementedError(f"frames of {w} columns: the polyphase form needs a width divisible by 4")
    dev = img.device
    w2 = w // 2
    g = planmod.full_geometry(h, w2)
    tw_row, tw_col = planmod.get_twiddles(w2, dev), planmod.get_twiddles(h, dev)
    per_frame = 2 * g.nkx * g.H * 8
    chunk = max(1, min(t, WORKSPACE_BYTES // (2 * per_frame)))
    T1 = torch.empty((2 * chunk, g.nkx, g.ny, 2), dtype=torch.float32, device=dev)
    S = torch.empty((2 * chunk, g.nkx, g.nky, 2), dtype=torch.float32, device=dev)
    A = torch.empty((2, g.nkx, g.nky, 2), dtype=torch.float32, device=dev)
    st = stream_ptr(dev)
    for a in range(0, t, chunk):
        n = min(chunk, t - a)
        sub = torch.cat([img[a:a + n, :, 0::2], img[a:a + n, :, 1::2]], dim=0).contiguous()
        off = torch.arange(2 * n, device=dev, dtype=torch.int64) * (h * w2)
        check(_k1(lib, g, dev, sub, off, w2, None, None, None, T1, tw_row, 2 * n, st), "xc rows forward")
        check(_k2(lib, g, dev, T1, None, S, tw_col, 2 * n, st), "xc cols forward")
        check(lib.mc_polyphase_dose_accumulate(ptr(S), n, a, t, ptr(A), g.nkx, h, w, float(pixel_spacing),
                                               float(pre_exposure), float(dose_per_frame), float(voltage),
                                               1 if a == 0 else 0, 1 if a + n >= t else 0, st),
              "mc_polyphase_dose_accumulate")
    halves = _inverse_frames(lib, g, A, 2, h, w2, dev, st)
    out = torch.empty((h, w), dtype=torch.float32, device=dev)
    out[:, 0::2] = halves[0]
    out[:, 1::2] = halves[1]
    return out


def dose_weighted_sum(img, pixel_spacing, dose_per_frame, pre_exposure=0.0, voltage=300.0):
    """sum_f irfft2(q_f * rfft2(frame_f)): the exposure-filtered frame sum of the reference's
    example pipeline (examples/ttMotion.py:331-351, 398) with ONE inverse transform per movie:
    full spectra of a chunk of frames (K1+K2, no mask / filter) -> mc_dose_accumulate -> inverse
    column and row passes.  Semantics of the absent third-party filter: parity unpinned."""
    lib = _lib.load()
    t, h, w = img.shape
    dev = img.device
    if POLYPHASE_FOURIER_SHIFT:
        return _dose_weighted_sum_polyphase(img, pixel_spacing, dose_per_frame, pre_exposure, voltage)
    if _full_row_major_ok(h, w):
        return _dose_weighted_sum_row_major(img, pixel_spacing, dose_per_frame, pre_exposure, voltage)
    try:
        g = planmod.full_geometry(h, w)
    except NotImplementedError:
        if w % 4 == 0 and w <= 16384 and h <= 8192:
            return _dose_weighted_sum_polyphase(img, pixel_spacing, dose_per_frame, pre_exposure, voltage)
        raise
    tw_row, tw_col = planmod.get_twiddles(w, dev), planmod.get_twiddles(h, dev)
    per_frame = g.nkx * g.H * 8
    chunk = max(1, min(t, WORKSPACE_BYTES // (2 * per_frame)))
    T1 = torch.empty((chunk, g.nkx, g.ny, 2), dtype=torch.float32, device=dev)
    S = torch.empty((chunk, g.nkx, g.nky, 2), dtype=torch.float32, device=dev)
    A = torch.empty((1, g.nkx, g.nky, 2), dtype=torch.float32, device=dev)
    st = stream_ptr(dev)
    for a in range(0, t, chunk):
        n = min(chunk, t - a)
        off = torch.arange(a, a + n, device=dev, dtype=torch.int64) * (h * w)
        check(_k1(lib, g, dev, img, off, w, None, None, None, T1, tw_row, n, st), "xc rows forward")
        check(_k2(lib, g, dev, T1, None, S, tw_col, n, st), "xc cols forward")
        check(lib.mc_dose_accumulate(ptr(S), n, a, t, ptr(A), w, h, float(pixel_spacing), float(pre_exposure),
                                     float(dose_per_frame), float(voltage), 1 if a == 0 else 0,
                                     1 if a + n >= t else 0, st), "mc_dose_accumulate")
    # inverse of the single accumulated spectrum: Fourier-shift path with a zero shift
    out = torch.empty((1, h, w), dtype=torch.float32, device=dev)
    idx = torch.zeros(1, device=dev, dtype=torch.int32)
    zero = torch.zeros((1, 2), device=dev, dtype=torch.float32)
    off0 = torch.zeros(1, device=dev, dtype=torch.int64)
    T2 = T1[:1] if g.ny == g.H else torch.empty((1, g.nkx, g.H, 2), dtype=torch.float32, device=dev)
    if planmod.native_height(g.H):
        check(lib.mc_fourier_shift_cols_inverse(ptr(A), ptr(idx), ptr(zero), ptr(T2), ptr(tw_col),
                                                1.0 / (h * w), 1, g, st), "mc_fourier_shift_cols_inverse")
    else:
        line, _ = planmod.line_plan(g.H, +1, dev)
        check(lib.mc_xcg_cols_inverse(ptr(A), ptr(idx), None, None, ptr(zero), ptr(T2), line,
                                      1.0 / (h * w), 1, g, st), "mc_xcg_cols_inverse")
    if planmod.native_rows(g):
        check(lib.mc_xc_rows_inverse_store(ptr(T2), ptr(out), ptr(off0), w, ptr(tw_row), 1, g, st),
              "mc_xc_rows_inverse_store")
    else:
        line, _ = planmod.line_plan(planmod.row_line_length(g.W), +1, dev)
        check(lib.mc_xcg_rows_inverse(ptr(T2), None, None, None, None, ptr(out), ptr(off0), w,
                                      ptr(tw_row), line, 1, g, st), "mc_xcg_rows_inverse")
    return out[0]


_RAW_KINDS = {torch.uint8: 0, torch.int16: 1, torch.float16: 2, torch.float32: 3}


# ------------------------------------------------------------------ N2: the rigid path straight from raw frames


class RawMovie:
    """A raw detector movie with what the fused kernels need to condition it on the fly
    (c = raw * gain - mu_f, examples/ttMotion.py:90-121, 180-199): the (t,h,w) u8 / i16 stack, the (h,w)
    fp32 gain reference and, from ONE pass over the raw bytes (mc_raw_movie_stats), the frame means `mu`,
    the per-frame offsets `sub` = mu + box mean and `mean_rstd` of the conditioned central box.  No fp32
    movie is ever allocated."""

    def __init__(self, raw, gain, mean_zero=True):
        lib = _lib.load()
        if raw.dtype not in (torch.uint8, torch.int16):
            raise TypeError(f"the fused raw path reads uint8 or int16 frames, got {raw.dtype}")
        t, h, w = raw.shape
        dev = raw.device
        self.raw = raw.contiguous()
        self.gain = (torch.ones((h, w), dtype=torch.float32, device=dev) if gain is None
                     else gain.detach().to(device=dev, dtype=torch.float32).contiguous())
        if tuple(self.gain.shape) != (h, w):
            raise ValueError(f"gain reference {tuple(self.gain.shape)} does not match the frames {(h, w)}")
        self.kind = _RAW_KINDS[raw.dtype]
        self.shape = (t, h, w)
        hl, hu, wl, wu = int(0.25 * h), int(0.75 * h), int(0.25 * w), int(0.75 * w)  # utils.py:76-81
        self.stats = torch.empty((t, 3), dtype=torch.float64, device=dev)
        self.mu = torch.empty(t, dtype=torch.float32, device=dev)
        self.sub = torch.empty(t, dtype=torch.float32, device=dev)
        self.mean_rstd = torch.empty(2, dtype=torch.float32, device=dev)
        check(lib.mc_raw_movie_stats(ptr(self.raw), self.kind, ptr(self.gain), t, h, w, hl, hu, wl, wu,
                                     1 if mean_zero else 0, ptr(self.stats), ptr(self.mu), ptr(self.sub),
                                     ptr(self.mean_rstd), stream_ptr(dev)), "mc_raw_movie_stats")


def raw_fused_supported(raw, pl):
    """Shapes the fused raw kernels take (a shortcut: the C side is the authority and answers
    MC_ERR_UNSUPPORTED for anything else; the caller then conditions the movie into an fp32 copy).  K1 from
    raw bytes exists for power-of-two widths (4096: the wave-per-row engine) and the K3 formats' rows of 5760 /
    11520 samples; the raw warp needs rows of whole quads and at most 256 frames."""
    t, h, w = raw.shape
    g = pl.geom
    rows_ok = planmod.native_rows(g) or (w % 2 == 0 and planmod.row_line_length(w) in (2880, 5760)
                                        and planmod.USE_DIRECT_LINES)
    return raw.dtype in (torch.uint8, torch.int16) and rows_ok and w % 4 == 0 and t <= 256 and raw.data_ptr() % 16 == 0


def global_shifts_raw(rm: RawMovie, reference_frame, pixel_spacing, b_factor, frequency_range):
    """global_shifts for a RawMovie: K1 reads the raw bytes (mc_xc_rows_forward_raw / mc_xcg_rows_forward_raw), the
    statistics are known beforehand, so the plain column pass follows.  Raises McorrUnsupported for shapes
    without a fused kernel."""
    lib = _lib.load()
    t, h, w = rm.shape
    dev = rm.raw.device
    _lib.normalize_frame_index(reference_frame, t)  # IndexError before any launch, as the fp32 path
    pl = planmod.get_xc_plan(h, w, pixel_spacing, b_factor, frequency_range, dev)
    g = pl.geom
    if not raw_fused_supported(rm.raw, pl):
        raise _lib.McorrUnsupported("no fused raw kernel for this frame shape")
    st = stream_ptr(dev)
    job_off = _cached(("frame_off", str(dev), t, h, w),
                      lambda: torch.arange(t, device=dev, dtype=torch.int64) * (h * w))
    S = torch.empty((t, g.nkx, g.nky, 2), dtype=torch.float32, device=dev)
    # row pass in chunks of frames when the transposed intermediate would be large (K3 formats: 0.4 GB per 10 frames)
    per_job = g.nkx * g.ny * 8
    chunk = max(1, min(t, WORKSPACE_BYTES // per_job))
    T1 = torch.empty((chunk, g.nkx, g.ny, 2), dtype=torch.float32, device=dev)
    chord = ptr(pl.chord) if (pl.chord is not None and USE_ROW_CHORDS) else None
    for a in range(0, t, chunk):
        n = min(chunk, t - a)
        off, sub = job_off[a:a + n], rm.sub[a:a + n]
        if planmod.native_rows(g):
            check(lib.mc_xc_rows_forward_raw(ptr(rm.raw), rm.kind, ptr(rm.gain), ptr(off), w, ptr(pl.mask), ptr(sub),
                                             ptr(rm.mean_rstd), ptr(T1), ptr(pl.tw_row), n, g, chord, st),
                  "mc_xc_rows_forward_raw")
        else:
            line, _ = planmod.line_plan(planmod.row_line_length(g.W), -1, dev, keep=planmod.row_line_keep(g.W, g.nkx))
            check(lib.mc_xcg_rows_forward_raw(ptr(rm.raw), rm.kind, ptr(rm.gain), ptr(off), w, ptr(pl.mask), ptr(sub),
                                              ptr(rm.mean_rstd), ptr(T1), ptr(pl.tw_row), line, n, g, st),
                  "mc_xcg_rows_forward_raw")
        if a + n >= t and AFTER_K1_HOOK is not None:
            AFTER_K1_HOOK()
        check(_k2(lib, g, dev, T1, pl.filt, S[a:a + n], pl.tw_col, n, st), "xc cols forward")
    del T1
    return _shifts_from_spectra(S, t, reference_frame, pl)


def warp_rigid_raw(rm: RawMovie, lattices, pixel_spacing, want_frames=True, want_sum=False, tables=None):
    """``warp(..., rigid=True)`` of the conditioned movie without materialising it (mc_warp_rigid_raw)."""
    lib = _lib.load()
    t, h, w = rm.shape
    dev = rm.raw.device
    frames = torch.empty((t, h, w), dtype=torch.float32, device=dev) if want_frames else None
    total = torch.empty((h, w), dtype=torch.float32, device=dev) if want_sum else None
    if tables is None:
        shifts_px = (lattices[:, :, 0, 0] / pixel_spacing).contiguous()
        nbytes = C.c_int64(0)
        check(lib.mc_warp_rigid_scratch_bytes(t, h, w, C.byref(nbytes)), "mc_warp_rigid_scratch_bytes")
        scratch = torch.empty((nbytes.value + 3) // 4, dtype=torch.float32, device=dev)
        phase = 0
    else:
        shifts_px, scratch = tables
        phase = 2
    run = lambda: check(lib.mc_warp_rigid_raw(ptr(rm.raw), rm.kind, ptr(rm.gain), ptr(rm.mu), t, h, w, ptr(shifts_px),
                                              ptr(scratch), ptr(frames), ptr(total), phase, stream_ptr(dev)),
                        "mc_warp_rigid_raw")
    if RIGID_KERNEL_HOOK is not None and phase == 2:
        RIGID_KERNEL_HOOK(run)
    else:
        run()
    return frames, total


def condition_movie(raw, gain=None, mean_zero=True, hot_pixel_threshold=None, return_hot_counts=False):
    """raw (t,h,w) u8 / i16 / f16 / f32 on the GPU -> fp32 frames: x * gain, minus the frame's
    own mean (examples/ttMotion.py:90-121, 174-199), in two passes over the raw bytes.  With
    `hot_pixel_threshold` (the example uses 10.0) the hot-pixel step of examples/ttMotion.py:127-172
    runs in between: the example's detection, a deterministic replacement (mc_condition_movie_hot)."""
    lib = _lib.load()
    if raw.dtype not in _RAW_KINDS:
        raise TypeError(f"unsupported raw frame type {raw.dtype}; use uint8, int16, float16 or float32")
    t, h, w = raw.shape
    dev = raw.device
    raw = raw.contiguous()
    if gain is not None:
        if tuple(gain.shape) != (h, w):
            raise ValueError(f"gain reference has shape {tuple(gain.shape)}, frames are {(h, w)}")
        gain = gain.to(device=dev, dtype=torch.float32).contiguous()
    out = torch.empty((t, h, w), dtype=torch.float32, device=dev)
    if hot_pixel_threshold is not None:
        stats = torch.empty(3 * t, dtype=torch.float64, device=dev)
        counts = torch.empty(t, dtype=torch.int32, device=dev)
        check(lib.mc_condition_movie_hot(ptr(raw), _RAW_KINDS[raw.dtype], ptr(gain), t, h, w,
                                         1 if mean_zero else 0, float(hot_pixel_threshold), ptr(stats),
                                         ptr(counts), ptr(out), stream_ptr(dev)), "mc_condition_movie_hot")
        return (out, counts) if return_hot_counts else out
    sums = torch.empty(t, dtype=torch.float64, device=dev) if mean_zero else None
    check(lib.mc_condition_movie(ptr(raw), _RAW_KINDS[raw.dtype], ptr(gain), t, h * w, 1 if mean_zero else 0,
                                 ptr(sums), ptr(out), stream_ptr(dev)), "mc_condition_movie")
    return (out, torch.zeros(t, dtype=torch.int32, device=dev)) if return_hot_counts else out


def sum_frames(frames):
    lib = _lib.load()
    t, h, w = frames.shape
    total = torch.empty((h, w), dtype=torch.float32, device=frames.device)
    check(lib.mc_sum_frames(ptr(frames), t, h * w, ptr(total), stream_ptr(frames.device)),
          "mc_sum_frames")
    return total
