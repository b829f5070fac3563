"""Reference-compatible Python entry points (same names, argument order, defaults,
shapes, units and error behaviour as torch_motion_correction's public API,
src/torch_motion_correction/__init__.py:12-44), executing on the MI355X through
libmcorr.  No CPU fallback exists: without a ROCm device these raise McorrError.

Device rule (reference: ``device=None`` -> ``image.device``): results are returned on
the device the reference would have returned them on; when that device is the CPU,
inputs are staged to the current GPU, computed there and copied back.

``BUG_COMPATIBLE`` (default True) keeps the reference's behavioural accidents that
change numbers or mutate arguments: the in-place negation of
the caller's field in ``correct_motion_fast`` (Q1) and the mask-exponent schedule
caused by the lazy-patch memo aliasing (Q2/Q3), and the RuntimeError the reference raises
for sub_pixel_refinement=False with outlier_rejection=True (Q12: torch.std on int64
peaks).  Q4-Q9 are plain semantics and are always reproduced.
"""

from __future__ import annotations

import functools
import inspect

import torch

from . import engine
from ._lib import McorrUnsupported, device_scope, normalize_frame_index, require_gpu

BUG_COMPATIBLE = True
RIGID_FAST_PATH = True  # (2,nt,1,1) fields use the separable rigid warp kernel
VERBOSE = False  # the reference prints progress lines; opt in with VERBOSE = True


def _say(msg: str):
    if VERBOSE:
        print(msg)


def _is_rigid(field: torch.Tensor) -> bool:
    return tuple(field.shape[-2:]) == (1, 1)


def _out_device(image: torch.Tensor, device):
    return image.device if device is None else torch.device(device)


def _stage(x: torch.Tensor, dev, keep_half: bool = False) -> torch.Tensor:
    """The tensor on the GPU as contiguous fp32 -- or, `keep_half`, an fp16 stack as it is: the patch
    estimator and the deformation-field warp read fp16 frames straight from their bytes (BASELINE
    C5; no fp32 copy of the movie, half the HBM read)."""
    if keep_half and x.dtype == torch.float16:
        return x.detach().to(device=dev).contiguous()
    return x.detach().to(device=dev, dtype=torch.float32).contiguous()


def _on_gpu(fn):
    """Run `fn` with the GPU it computes on as the CURRENT HIP device (libmcorr launches on the
    current device; the reference takes ``device=`` / the first tensor's device per call).  The
    device is chosen exactly as the body does: ``device`` if given, else the first argument's."""
    sig = inspect.signature(fn)

    @functools.wraps(fn)
    def scoped(*args, **kwargs):
        bound = sig.bind(*args, **kwargs)
        first = next(iter(bound.arguments.values()))
        device = bound.arguments.get("device")
        out_dev = first.device if device is None and isinstance(first, torch.Tensor) else device
        with device_scope(require_gpu(out_dev)):
            return fn(*args, **kwargs)

    return scoped


# ------------------------------------------------------------------ field utilities


def image_shifts_to_deformation_field(shifts, pixel_spacing, device=None):
    """(t,2) px shifts -> (2,t,1,1) Angstrom field, no sign flip
    (deformation_field_utils.py:129-162)."""
    if device is not None:
        shifts = shifts.to(device)
    return (shifts * pixel_spacing).transpose(0, 1)[:, :, None, None]


@_on_gpu
def evaluate_deformation_field(deformation_field, tyx, grid_type="catmull_rom"):
    """(c,nt,nh,nw) spline grid evaluated at (...,3) tyx points in [0,1] -> (...,c)
    (deformation_field_utils.py:9-39): one launch over all points (mc_spline_points)."""
    out_dev = deformation_field.device
    dev = require_gpu(out_dev)
    field = _stage(deformation_field, dev)
    vals = engine.spline_points(field, tyx.reshape(-1, 3), grid_type)
    return vals.reshape(*tyx.shape[:-1], field.shape[0]).to(out_dev)


@_on_gpu
def evaluate_deformation_field_at_t(deformation_field, t, grid_shape, grid_type="catmull_rom"):
    """(c, H, W) shifts on the linspace(0,1) lattice at time t
    (deformation_field_utils.py:42-93)."""
    out_dev = deformation_field.device
    dev = require_gpu(out_dev)
    H, W = grid_shape
    lat = engine.spline_lattice(_stage(deformation_field, dev),
                                torch.as_tensor([float(t)], dtype=torch.float32),
                                torch.linspace(0, 1, steps=H), torch.linspace(0, 1, steps=W), grid_type)
    return lat[:, 0].to(out_dev)


@_on_gpu
def resample_deformation_field(deformation_field, target_resolution):
    """Catmull-Rom resample to (nt,nh,nw) (deformation_field_utils.py:96-126)."""
    out_dev = deformation_field.device
    dev = require_gpu(out_dev)
    nt, nh, nw = target_resolution
    lat = engine.spline_lattice(_stage(deformation_field, dev), torch.linspace(0, 1, steps=nt),
                                torch.linspace(0, 1, steps=nh), torch.linspace(0, 1, steps=nw),
                                "catmull_rom")
    return lat.to(out_dev)


# ------------------------------------------------------------------ estimators


@_on_gpu
def estimate_global_motion(image, pixel_spacing, reference_frame=None, b_factor=500,
                           frequency_range=(300, 10), device=None):
    """Whole-frame cross-correlation shift estimate (estimate_motion_xc.py:21-135).
    Returns the (2,t,1,1) float32 deformation field in Angstrom (integer px shifts x
    pixel_spacing; the reference frame's entry is exactly 0)."""
    out_dev = _out_device(image, device)
    dev = require_gpu(out_dev)
    img = _stage(image, dev, keep_half=True)  # fp16 stacks: K1 reads the 16-bit samples (4096-column frames)
    t = img.shape[0]
    ref = t // 2 if reference_frame is None else reference_frame
    normalize_frame_index(ref, t)  # IndexError outside [-t, t), as filtered_fft[ref] (xc.py:101)
    _say(f"Cross-correlation whole image: using frame {ref} as reference")
    shifts = engine.global_shifts(img, ref, float(pixel_spacing), float(b_factor), frequency_range)
    if VERBOSE:
        _say(f"Estimated shifts range: y=[{shifts[:, 0].min():.1f}, {shifts[:, 0].max():.1f}], "
             f"x=[{shifts[:, 1].min():.1f}, {shifts[:, 1].max():.1f}]")
    return image_shifts_to_deformation_field(shifts, pixel_spacing).to(out_dev)


@_on_gpu
def estimate_motion_cross_correlation_patches(
    image, pixel_spacing, reference_frame=None, reference_strategy="mean_except_current",
    b_factor=500, frequency_range=(300, 10), patch_sidelength=1024, sub_pixel_refinement=True,
    temporal_smoothing=True, smoothing_window_size=5, deformation_field=None,
    outlier_rejection=True, outlier_threshold=3.0, device=None,
):
    """Per-patch cross-correlation shift estimate (estimate_motion_xc.py:138-411).
    Returns ((2,t,gh,gw) Angstrom field with its global mean subtracted,
    (t,gh,gw,3) int64 patch centres)."""
    out_dev = _out_device(image, device)
    dev = require_gpu(out_dev)
    # an fp16 stack stays fp16 unless a prior field has to be applied first (that path normalises
    # and resamples in fp32)
    img = _stage(image, dev, keep_half=deformation_field is None)
    t, h, w = img.shape
    ref = t // 2 if reference_frame is None else reference_frame
    if reference_strategy not in ("middle_frame", "mean_except_current"):
        raise ValueError(f"Unknown reference_strategy: {reference_strategy}")
    if reference_strategy == "middle_frame":
        normalize_frame_index(ref, t)  # lazy_patch_grid[ref] (xc.py:306): IndexError outside [-t, t)
    else:
        ref = t // 2  # mean_except_current never reads reference_frame (xc.py:310-328)
    if BUG_COMPATIBLE and outlier_rejection and not sub_pixel_refinement:
        # Q12: integer peak coordinates reach torch.std at estimate_motion_xc.py:577
        raise RuntimeError("std and var only support floating point and complex dtypes")
    stats = engine.central_box_stats(img)  # statistics of the *uncorrected* stack (Q9)
    field0 = None
    if deformation_field is not None:
        if deformation_field.device != out_dev:
            deformation_field = deformation_field.clone()  # the reference's .to(device) copy
        norm = engine.normalize(img, stats)
        stats = None
        if tuple(deformation_field.shape[-2:]) == (1, 1):
            _say("Applying single patch deformation field using correct_motion_fast")
            img = _correct_motion_fast_impl(norm, deformation_field, dev, mutate=BUG_COMPATIBLE)
        else:
            _say("Applying full deformation field using correct_motion")
            lat = engine.frame_lattices(_stage(deformation_field, dev), t, "bspline")
            img, _ = engine.warp(norm, lat, float(pixel_spacing))
        # the prior field (after Q1's in-place negation, if any) is the accumulator base
        from .lattice import patch_grid_centers

        cy, cx = patch_grid_centers(t, h, w, int(patch_sidelength))
        field0 = resample_deformation_field(_stage(deformation_field, dev), (t, len(cy), len(cx)))
    field, centers = engine.patch_field(
        img, stats, float(pixel_spacing), ref, reference_strategy, float(b_factor), frequency_range,
        patch_sidelength, bool(sub_pixel_refinement), bool(temporal_smoothing),
        int(smoothing_window_size), field0, bool(outlier_rejection), float(outlier_threshold))
    return field.to(out_dev), centers.to(out_dev)


def estimate_motion(image, pixel_spacing, patch_sidelength=None, **kwargs):
    """Convenience alias (ours, not the reference's): global estimate when
    ``patch_sidelength`` is None, else the patch estimate (field only)."""
    if patch_sidelength is None:
        return estimate_global_motion(image, pixel_spacing, **kwargs)
    return estimate_motion_cross_correlation_patches(
        image, pixel_spacing, patch_sidelength=patch_sidelength, **kwargs)[0]


# ------------------------------------------------------------------ correctors


@_on_gpu
def correct_motion(image, deformation_grid, pixel_spacing, grad=False, grid_type="catmull_rom",
                   device=None):
    """Apply a (2,nt,gh,gw) Angstrom deformation field (correct_motion.py:18-78).
    Returns the (t,h,w) corrected frames, detached.  ``grad=True`` (autograd through
    the resampling) is not available on the HIP path."""
    if grad:
        raise NotImplementedError("grad=True is not supported by the HIP path (forward only)")
    out_dev = _out_device(image, device)
    dev = require_gpu(out_dev)
    img = _stage(image, dev, keep_half=True)
    lat = engine.frame_lattices(_stage(deformation_grid, dev), img.shape[0], grid_type)
    frames, _ = engine.warp(img, lat, float(pixel_spacing), want_frames=True, want_sum=False,
                            rigid=RIGID_FAST_PATH and _is_rigid(deformation_grid))
    return frames.to(out_dev)


def _grid_data_and_type(grid, default="catmull_rom"):
    """A (c,nt,nh,nw) tensor, or a spline-grid object of the reference's dependency
    (torch_cubic_spline_grids: `.data` holds the control points, the class name the basis)."""
    if isinstance(grid, torch.Tensor):
        return grid, default
    data = getattr(grid, "data", None)
    if not isinstance(data, torch.Tensor):
        raise TypeError("expected a (2, nt, nh, nw) tensor or a cubic spline grid object with a .data tensor")
    return data, ("bspline" if "bspline" in type(grid).__name__.lower() else "catmull_rom")


def _wants_grad(grid) -> bool:
    if isinstance(grid, torch.Tensor):
        return grid.requires_grad
    params = getattr(grid, "parameters", None)
    return any(p.requires_grad for p in params()) if callable(params) else False


@_on_gpu
def correct_motion_two_grids(image, new_deformation_grid, base_deformation_grid, pixel_spacing, grad=True,
                             device=None):
    """correct_motion.py:188-299 -- the frames resampled through the SUM of two spline grids
    (an optimisable one and a frozen base), each evaluated on the (10 gh, 10 gw) lattice of the
    new grid.  The grids are (2,nt,nh,nw) tensors (Catmull-Rom) or grid objects of the reference's
    spline package; they may differ in resolution and basis.  Forward only: with ``grad=True`` (the
    reference's default) and a grid that requires gradients the result is attached to that grid, as
    in the reference, but calling ``backward`` through it raises NotImplementedError."""
    out_dev = _out_device(image, device)
    dev = require_gpu(out_dev)
    img = _stage(image, dev)
    new, new_type = _grid_data_and_type(new_deformation_grid)
    base, base_type = _grid_data_and_type(base_deformation_grid)
    t = img.shape[0]
    _, _, gh, gw = new.shape
    lin = lambda n: torch.linspace(0, 1, steps=n)
    lat = (engine.spline_lattice(_stage(new, dev), lin(t), lin(10 * gh), lin(10 * gw), new_type)
           + engine.spline_lattice(_stage(base, dev), lin(t), lin(10 * gh), lin(10 * gw), base_type))
    frames, _ = engine.warp(img, lat.permute(1, 0, 2, 3).contiguous(), float(pixel_spacing),
                            want_frames=True, want_sum=False)
    frames = frames.to(out_dev)
    if grad and _wants_grad(new_deformation_grid):
        # the reference returns frames attached to the new grid's graph; the forward here is the
        # same, the backward is refused where it would be needed
        params = ([new_deformation_grid] if isinstance(new_deformation_grid, torch.Tensor)
                  else [p for p in new_deformation_grid.parameters() if p.requires_grad])
        frames = _ForwardOnly.apply(frames, *params)
    return frames


class _ForwardOnly(torch.autograd.Function):
    """Marks a HIP result as depending on `params`; differentiating through it fails loudly."""

    @staticmethod
    def forward(ctx, frames, *params):
        return frames.view_as(frames)

    @staticmethod
    def backward(ctx, gout):
        raise NotImplementedError("gradients through the HIP resampling kernels are not available "
                                  "(forward only); use grad=False")


@_on_gpu
def correct_motion_slow(image, deformation_grid, grad=False, device=None):
    """correct_motion.py:302-427 -- the (2,nt,nh,nw) field evaluated (Catmull-Rom) at EVERY pixel
    (t_i, y/(h-1), x/(w-1)) and used as PIXEL shifts (no pixel spacing), then bicubic resampling.
    One frame at a time: the per-pixel shifts are a tensor-product spline lattice of the frame's
    own size, which the general warp kernel consumes directly (its bicubic lattice upsample is the
    identity at the lattice nodes up to 1e-4 of the shift difference between adjacent pixels)."""
    if grad:
        raise NotImplementedError("grad=True is not supported by the HIP path (forward only)")
    out_dev = _out_device(image, device)
    dev = require_gpu(out_dev)
    img = _stage(image, dev)
    field = _stage(deformation_grid, dev)
    t, h, w = img.shape
    times = torch.linspace(0, 1, steps=t)
    uy = torch.arange(h, dtype=torch.float32) / float(h - 1)
    ux = torch.arange(w, dtype=torch.float32) / float(w - 1)
    out = torch.empty_like(img)
    for f in range(t):
        lat = engine.spline_lattice(field, times[f:f + 1], uy, ux, "catmull_rom")  # (2, 1, h, w)
        frames, _ = engine.warp(img[f:f + 1], lat.permute(1, 0, 2, 3).contiguous(), 1.0,
                                want_frames=True, want_sum=False)
        out[f] = frames[0]
    return out.to(out_dev)


@_on_gpu
def motion_correct_sum(image, deformation_grid, pixel_spacing, grid_type="catmull_rom", device=None,
                       return_frames=False, dose_per_frame=None, pre_exposure=0.0, voltage=300.0):
    """Fused correct_motion + the caller-side ``torch.sum(movie, dim=0)`` of the
    reference's pipeline (examples/ttMotion.py:398): returns the (h,w) aligned sum
    (and the frames when asked) without a second pass over the stack.  With
    ``dose_per_frame`` (e/A^2) the sum is exposure-filtered as in the reference's
    ``dose_weight`` step (examples/ttMotion.py:331-351; see ``dose_weighted_sum``)."""
    out_dev = _out_device(image, device)
    dev = require_gpu(out_dev)
    img = _stage(image, dev, keep_half=True)
    lat = engine.frame_lattices(_stage(deformation_grid, dev), img.shape[0], grid_type)
    rigid = RIGID_FAST_PATH and _is_rigid(deformation_grid)
    if dose_per_frame is None:
        frames, total = engine.warp(img, lat, float(pixel_spacing), want_frames=return_frames, want_sum=True,
                                    rigid=rigid)
    elif return_frames:
        frames, _ = engine.warp(img, lat, float(pixel_spacing), want_frames=True, want_sum=False, rigid=rigid)
        total = engine.dose_weighted_sum(frames, float(pixel_spacing), float(dose_per_frame),
                                         float(pre_exposure), float(voltage))
    else:  # the corrected movie is only an intermediate: warped and transformed a chunk at a time
        frames = None
        total = engine.warp_dose_weighted_sum(img, lat, float(pixel_spacing), rigid, float(dose_per_frame),
                                              float(pre_exposure), float(voltage))
    return (total.to(out_dev), frames.to(out_dev)) if return_frames else total.to(out_dev)


@_on_gpu
def condition_movie(movie, gain=None, mean_zero=True, device=None, hot_pixel_threshold=None,
                    return_hot_counts=False):
    """Raw detector frames -> the fp32 stack the estimators expect: ``movie * gain`` (a (h,w)
    multiplicative gain reference, already flipped / rotated as needed) and, per frame,
    minus its own mean -- the ``gain_correct`` and ``set_frames_mean_zero`` steps of the
    reference's pipeline (examples/ttMotion.py:90-121, 174-199) done on the device straight
    from uint8 / int16 / float16 / float32 storage.  `hot_pixel_threshold` (the example's
    ``remove_hot_pixels`` uses 10.0; None = off) adds that step between the two: a pixel more than
    `threshold` standard deviations from its frame's mean is hot -- the example's detection -- and
    is replaced by the mean of its neighbours that are not hot (the example draws a RANDOM
    neighbour: that part has no deterministic counterpart and the rule here is ours).
    `return_hot_counts`: also the (t,) int32 number of hot pixels found per frame."""
    out_dev = _out_device(movie, device)
    dev = require_gpu(out_dev)
    res = engine.condition_movie(movie.detach().to(dev), gain, bool(mean_zero), hot_pixel_threshold,
                                 bool(return_hot_counts))
    if return_hot_counts:
        return res[0].to(out_dev), res[1].to(out_dev)
    return res.to(out_dev)


@_on_gpu
def motion_correct_raw(movie, gain, pixel_spacing, reference_frame=None, b_factor=500, frequency_range=(300, 10),
                       grid_type="catmull_rom", mean_zero=True, return_frames=False, device=None):
    """The reference pipeline's gain_correct -> set_frames_mean_zero -> estimate_global_motion -> correct_motion
    -> sum (examples/ttMotion.py:90-121, 180-199, 286-398) for a RAW uint8 / int16 movie, with the conditioning
    fused into the kernels that read the raw bytes: one statistics pass, then the estimator's row transform and
    the rigid warp compute ``raw * gain - frame mean`` on the fly.  No conditioned fp32 movie is allocated.
    Returns ``(field (2,t,1,1) Angstrom, sum (h,w)[, frames (t,h,w)])`` -- what ``condition_movie`` followed by
    ``estimate_global_motion`` and ``motion_correct_sum`` return.  Fused kernels exist for power-of-two frame
    widths and the K3 formats (5760 / 11520 columns), rows of whole quads, at most 256 frames; any other shape
    takes exactly that route, on an fp32 copy."""
    out_dev = _out_device(movie, device)
    dev = require_gpu(out_dev)
    raw = movie.detach().to(dev)
    t = raw.shape[0]
    ref = t // 2 if reference_frame is None else int(reference_frame)
    ps = float(pixel_spacing)
    try:
        rm = engine.RawMovie(raw, None if gain is None else gain.to(dev), mean_zero=bool(mean_zero))
        shifts = engine.global_shifts_raw(rm, ref, ps, float(b_factor), tuple(frequency_range))
        field = image_shifts_to_deformation_field(shifts, ps)
        lat = engine.frame_lattices(field.contiguous(), t, grid_type)
        frames, total = engine.warp_rigid_raw(rm, lat, ps, want_frames=bool(return_frames), want_sum=True)
    except (McorrUnsupported, TypeError):
        img = engine.condition_movie(raw, None if gain is None else gain.to(dev), bool(mean_zero))
        shifts = engine.global_shifts(img, ref, ps, float(b_factor), tuple(frequency_range))
        field = image_shifts_to_deformation_field(shifts, ps)
        lat = engine.frame_lattices(field.contiguous(), t, grid_type)
        frames, total = engine.warp(img, lat, ps, want_frames=bool(return_frames), want_sum=True, rigid=True)
    if return_frames:
        return field.to(out_dev), total.to(out_dev), frames.to(out_dev)
    return field.to(out_dev), total.to(out_dev)


@_on_gpu
def dose_weighted_sum(movie, pixel_spacing, dose_per_frame, pre_exposure=0.0, voltage=300.0, device=None):
    """``sum_f irfft2(q_f * rfft2(frame_f))`` with the Grant & Grigorieff exposure filter
    ``q_f = exp(-0.5 N_f / N_c(|k|))`` normalised by ``sqrt(sum_f q_f^2)``: the reference
    pipeline's ``dose_weight(movie)`` (examples/ttMotion.py:331-351: rfft2(norm='ortho') ->
    torch_fourier_filter dose_weight_movie(crit_exposure_bfactor=-1) -> irfft2 -> sum).  The
    third-party filter is not part of the reference tree and untested there: parity unpinned."""
    out_dev = _out_device(movie, device)
    dev = require_gpu(out_dev)
    return engine.dose_weighted_sum(_stage(movie, dev), float(pixel_spacing), float(dose_per_frame),
                                    float(pre_exposure), float(voltage)).to(out_dev)


@_on_gpu
def estimate_local_motion(image, pixel_spacing, patch_shape, deformation_field_resolution,
                          initial_deformation_field=None, device=None, n_iterations=100, b_factor=500,
                          frequency_range=(300, 10), optimizer_type="adam", grid_type="catmull_rom",
                          loss_type="mse", optimizer_kwargs=None, return_trajectory=False,
                          trajectory_kwargs=None):
    """Refine a (2, nt, nh, nw) spline deformation field by gradient descent on the agreement of
    Fourier-shifted patches with the mean of the other frames (estimate_motion_optimizer.py:28-439;
    same arguments, defaults, return values and error messages).  The loss and its analytic
    gradient are HIP kernels over patch spectra that are transformed once (local_motion.py)."""
    from . import local_motion
    from .optimization_state import OptimizationTracker

    out_dev = _out_device(image, device)
    dev = require_gpu(out_dev)  # CPU tensors are staged to the GPU and the field comes back on `out_dev`
    img = _stage(image, dev)
    if grid_type not in ("catmull_rom", "bspline"):
        raise ValueError(f"Invalid grid type: {grid_type}. Must be 'catmull_rom' or 'bspline'.")
    res = tuple(int(r) for r in deformation_field_resolution)
    trajectory = None
    if return_trajectory:
        tk = trajectory_kwargs if trajectory_kwargs is not None else {}
        tk.setdefault("sample_every_n_steps", 1)
        tk.setdefault("total_steps", n_iterations)
        trajectory = OptimizationTracker(**tk)
    if initial_deformation_field is None:
        init = torch.zeros((2, *res), dtype=torch.float32, device=dev)
    else:
        init = resample_deformation_field(_stage(initial_deformation_field.detach(), dev), res).contiguous()
        init = init - torch.mean(init)
    final = local_motion.estimate_local_motion(img, float(pixel_spacing), patch_shape, res, init, n_iterations,
                                               b_factor, frequency_range, optimizer_type, grid_type, loss_type,
                                               optimizer_kwargs, trajectory)
    final = final.to(out_dev)
    return (final, trajectory) if return_trajectory else final


def _correct_motion_fast_impl(img_dev, deformation_grid, dev, mutate):
    if tuple(deformation_grid.shape[-2:]) != (1, 1):
        raise ValueError(
            f"Expected single patch deformation field with shape (2, t, 1, 1), "
            f"but got shape {deformation_grid.shape}. "
            f"Final two dimensions must be (1, 1) for single patch correction."
        )
    shifts = -deformation_grid.detach()[:, :, 0, 0].transpose(0, 1).to(torch.float32)
    if mutate:
        deformation_grid.mul_(-1)  # Q1: correct_motion.py:473-474 negates the caller's tensor
    return engine.fourier_shift(img_dev, shifts.to(dev))


@_on_gpu
def correct_motion_fast(image, deformation_grid, device=None):
    """Rigid correction by a Fourier phase ramp (correct_motion.py:430-498); field
    values are used as pixels.  With BUG_COMPATIBLE the caller's `deformation_grid` is
    negated in place exactly when the reference would do so (grid already on the
    target device)."""
    out_dev = _out_device(image, device)
    dev = require_gpu(out_dev)
    mutate = BUG_COMPATIBLE and (device is None or deformation_grid.device == torch.device(device))
    out = _correct_motion_fast_impl(_stage(image, dev), deformation_grid, dev, mutate)
    return out.to(out_dev)


@_on_gpu
def get_pixel_shifts(frame, pixel_spacing, frame_deformation_grid, pixel_grid=None):
    """(h,w,2) per-pixel shifts in px from a (2,G_h,G_w) Angstrom lattice
    (correct_motion.py:132-185).  `pixel_grid` (..., 2) holds the (y, x) pixel coordinates to
    evaluate at (correct_motion.py:167-168); the reference always passes the identity grid
    coordinate_grid((h,w)), which -- like None -- takes the tabulated kernel; any other grid is
    evaluated point by point (mc_pixel_shifts_at), result shape = pixel_grid's."""
    out_dev = frame.device
    dev = require_gpu(out_dev)
    h, w = frame.shape[-2:]
    lat = _stage(frame_deformation_grid, dev)
    if pixel_grid is not None:
        if pixel_grid.shape[-1] != 2:
            raise ValueError(f"pixel_grid must have shape (..., 2), got {tuple(pixel_grid.shape)}")
        grid = _stage(pixel_grid, dev)
        identity = tuple(grid.shape) == (h, w, 2) and bool(
            (grid[..., 0] == torch.arange(h, device=dev, dtype=torch.float32)[:, None]).all()
            and (grid[..., 1] == torch.arange(w, device=dev, dtype=torch.float32)[None, :]).all())
        if not identity:
            return engine.pixel_shifts_at(lat, h, w, float(pixel_spacing), grid).to(out_dev)
    out = engine.pixel_shifts(lat, h, w, float(pixel_spacing))
    return out.to(out_dev)
