"""CPU oracle of the estimate -> correct hot path (torch-CPU op sequence).

TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).  Each function cites the
reference lines it follows (paths relative to /root/reference/src/
torch_motion_correction/).  Behavioural accidents of the reference that change
numbers are reproduced on purpose.

Third-party numerics come from oracle/thirdparty_semantics.py (parity unpinned).
"""

from __future__ import annotations

import torch
import torch.nn.functional as F
from scipy.signal import savgol_filter

from oracle import thirdparty_semantics as tp
from oracle.patch_grid import patch_grid_lazy

QUIET = True


def _say(*a):
    if not QUIET:
        print(*a)


# ------------------------------------------------------------------- utils.py


def normalize_image(image, frac_low=0.25, frac_high=0.75):
    """utils.py:49-84 -- one scalar mean / unbiased std of the central box over all
    frames jointly."""
    _, h, w = image.shape
    box = image[:, int(frac_low * h) : int(frac_high * h), int(frac_low * w) : int(frac_high * w)]
    std, mean = torch.std_mean(box, dim=(-3, -2, -1))
    return (image - mean) / std


def prepare_bandpass_filter(frequency_range, patch_shape, pixel_spacing, refinement_fraction=1.0,
                            device=None):
    """utils.py:87-114 -- binary band between pixel_spacing/cuton and
    pixel_spacing/cutoff (cycles/px), falloff 0."""
    cuton, cutoff_max = torch.as_tensor(frequency_range).float()
    cutoff = torch.lerp(cuton, cutoff_max, refinement_fraction)
    low = torch.as_tensor(1 / cuton, dtype=torch.float32) * pixel_spacing
    high = torch.as_tensor(1 / cutoff, dtype=torch.float32) * pixel_spacing
    return tp.bandpass_filter(low=low, high=high, falloff=0, image_shape=patch_shape, rfft=True,
                              fftshift=False, device=device)


# -------------------------------------------------- deformation_field_utils.py


def image_shifts_to_deformation_field(shifts, pixel_spacing, device=None):
    """deformation_field_utils.py:129-162 -- (t,2) px -> (2,t,1,1) Angstrom, no sign flip."""
    if device is not None:
        shifts = shifts.to(device)
    return (shifts * pixel_spacing).transpose(0, 1)[:, :, None, None]


def evaluate_deformation_field(deformation_field, tyx, grid_type="catmull_rom"):
    """deformation_field_utils.py:9-39 -- (c,nt,nh,nw) spline grid at (...,3) -> (...,c)."""
    return tp.cubic_spline_grid_3d(deformation_field, tyx, grid_type)


def evaluate_deformation_field_at_t(deformation_field, t, grid_shape, grid_type="catmull_rom"):
    """deformation_field_utils.py:42-93 -- (2, H, W) shifts on a linspace(0,1) lattice."""
    H, W = grid_shape
    yy, xx = torch.meshgrid(torch.linspace(0, 1, steps=H), torch.linspace(0, 1, steps=W),
                            indexing="ij")
    tyx = torch.stack([torch.full_like(yy, float(t)), yy, xx], dim=-1).reshape(-1, 3)
    vals = evaluate_deformation_field(deformation_field, tyx, grid_type)  # (H*W, c)
    return vals.reshape(H, W, -1).permute(2, 0, 1)


def resample_deformation_field(deformation_field, target_resolution):
    """deformation_field_utils.py:96-126 -- Catmull-Rom (the default) resample."""
    nt, nh, nw = target_resolution
    tt, yy, xx = torch.meshgrid(torch.linspace(0, 1, steps=nt), torch.linspace(0, 1, steps=nh),
                                torch.linspace(0, 1, steps=nw), indexing="ij")
    tyx = torch.stack([tt, yy, xx], dim=-1)
    return evaluate_deformation_field(deformation_field, tyx).permute(3, 0, 1, 2)


# ------------------------------------------------------------ correct_motion.py


def get_pixel_shifts(frame, pixel_spacing, frame_deformation_grid, pixel_grid):
    """correct_motion.py:132-185 -- bicubic/reflection/align_corners upsample of the
    (2, G_h, G_w) Angstrom lattice to per-pixel shifts in px, (h, w, 2)."""
    h, w = frame.shape
    _, gh, gw = frame_deformation_grid.shape
    img_len = torch.as_tensor([h - 1, w - 1], dtype=torch.float32)
    grid_len = torch.as_tensor([gh - 1, gw - 1], dtype=torch.float32)
    interp = (pixel_grid / img_len) * grid_len
    interp = tp.array_to_grid_sample(interp, array_shape=(gh, gw))
    shifts = F.grid_sample(frame_deformation_grid[None], interp[None], mode="bicubic",
                           padding_mode="reflection", align_corners=True)
    return (shifts / pixel_spacing)[0].permute(1, 2, 0)


def _correct_frame(frame, pixel_spacing, frame_deformation_grid):
    """correct_motion.py:81-129."""
    h, w = frame.shape
    pixel_grid = tp.coordinate_grid((h, w))
    shifts = get_pixel_shifts(frame, pixel_spacing, frame_deformation_grid, pixel_grid)
    return tp.sample_image_2d(frame, pixel_grid + shifts, interpolation="bicubic")


def correct_motion(image, deformation_grid, pixel_spacing, grad=False, grid_type="catmull_rom",
                   device=None):
    """correct_motion.py:18-78 -- per frame: spline at t_i on a (10gh, 10gw) lattice,
    bicubic upsample, bicubic resample.  Returns (t,h,w); no sum, no dose weight."""
    image = image.float().cpu()
    deformation_grid = deformation_grid.float().cpu()
    t = image.shape[0]
    _, _, gh, gw = deformation_grid.shape
    times = torch.linspace(0, 1, steps=t)
    out = []
    with torch.no_grad():
        for frame, ft in zip(image, times):
            lattice = evaluate_deformation_field_at_t(deformation_grid, ft, (10 * gh, 10 * gw),
                                                      grid_type)
            out.append(_correct_frame(frame, pixel_spacing, lattice))
    return torch.stack(out, dim=0)


def correct_motion_two_grids(image, new_data, base_data, pixel_spacing, new_type="catmull_rom",
                             base_type="catmull_rom"):
    """correct_motion.py:188-299 -- both grids evaluated on the (10 gh, 10 gw) lattice of the NEW
    grid at t_i, summed, then _correct_frame.  The grids are given by their control data + basis."""
    image = image.float().cpu()
    t = image.shape[0]
    _, _, gh, gw = new_data.shape
    out = []
    with torch.no_grad():
        for frame, ft in zip(image, torch.linspace(0, 1, steps=t)):
            lat = (evaluate_deformation_field_at_t(new_data.float().cpu(), ft, (10 * gh, 10 * gw), new_type)
                   + evaluate_deformation_field_at_t(base_data.float().cpu(), ft, (10 * gh, 10 * gw), base_type))
            out.append(_correct_frame(frame, pixel_spacing, lat))
    return torch.stack(out, dim=0)


def correct_motion_slow(image, deformation_grid, grad=False, device=None):
    """correct_motion.py:302-427 -- the field at every pixel (t_i, y/(h-1), x/(w-1)), Catmull-Rom,
    used as PIXEL shifts; bicubic sample_image_2d of pixel + shift."""
    image = image.float().cpu()
    field = deformation_grid.float().cpu()
    t, h, w = image.shape
    pixel_grid = tp.coordinate_grid((h, w))
    norm = pixel_grid / torch.as_tensor([h - 1, w - 1], dtype=torch.float32)
    out = []
    with torch.no_grad():
        for frame, ft in zip(image, torch.linspace(0, 1, steps=t)):
            tyx = F.pad(norm, pad=(1, 0), value=float(ft))
            shifts = evaluate_deformation_field(field, tyx)
            out.append(tp.sample_image_2d(frame, pixel_grid + shifts, interpolation="bicubic"))
    return torch.stack(out, dim=0)


def correct_motion_fast(image, deformation_grid, device=None):
    """correct_motion.py:430-498 -- Fourier phase-ramp shift by -field (used as pixels).
    Q1: negates the caller's grid IN PLACE (cm.py:473-474)."""
    if deformation_grid.shape[-2:] != (1, 1):
        raise ValueError(
            f"Expected single patch deformation field with shape (2, t, 1, 1), "
            f"but got shape {deformation_grid.shape}. "
            f"Final two dimensions must be (1, 1) for single patch correction."
        )
    t, h, w = image.shape
    shifts = deformation_grid[:, :, 0, 0].transpose(0, 1)  # view (t, 2)
    shifts *= -1
    dft = torch.fft.rfftn(image, dim=(-2, -1))
    dft = tp.fourier_shift_dft_2d(dft, (h, w), shifts, rfft=True, fftshifted=False)
    return torch.fft.irfftn(dft, s=(h, w))


# -------------------------------------------------------- estimate_motion_xc.py


def _filters(shape, pixel_spacing, b_factor, frequency_range):
    """xc.py:69-95 / :262-280 -- soft disk mask, B envelope, binary bandpass."""
    h, w = shape
    mask = tp.circle(radius=min(h, w) / 4, image_shape=(h, w), smoothing_radius=min(h, w) / 8)
    benv = tp.b_envelope(B=b_factor, image_shape=(h, w), pixel_size=pixel_spacing, rfft=True,
                         fftshift=False)
    band = prepare_bandpass_filter(frequency_range, (h, w), pixel_spacing)
    return mask, benv, band


def estimate_global_motion(image, pixel_spacing, reference_frame=None, b_factor=500,
                           frequency_range=(300, 10), device=None, return_cc=False):
    """xc.py:21-135 -- integer-pixel rigid shifts vs the reference frame."""
    image = image.float().cpu()
    t, h, w = image.shape
    ref = t // 2 if reference_frame is None else reference_frame
    image = normalize_image(image)
    mask, benv, band = _filters((h, w), pixel_spacing, b_factor, frequency_range)
    spec = torch.fft.rfftn(image * mask, dim=(-2, -1)) * band * benv
    shifts = torch.zeros((t, 2))
    ccs = {}
    for f in range(t):
        if f == ref:
            continue
        cc = torch.fft.irfftn(torch.conj(spec[ref]) * spec[f], s=(h, w))
        if return_cc:
            ccs[f] = cc
        py, px = divmod(int(torch.argmax(cc.flatten())), w)
        shifts[f, 0] = py if py <= h // 2 else py - h
        shifts[f, 1] = px if px <= w // 2 else px - w
    field = image_shifts_to_deformation_field(shifts, pixel_spacing)
    return (field, ccs) if return_cc else field


def _sub_pixel(cc3, peak, ph, pw):
    """xc.py:414-483 -- independent 1-D parabolas; none when the peak touches a border
    (Q4); an axis is skipped when its two outer samples are equal (Q5)."""
    py = (peak // pw).float()
    px = (peak % pw).float()
    for i in range(cc3.shape[0]):
        y, x = int(peak[i]) // pw, int(peak[i]) % pw
        if 1 <= y < ph - 1 and 1 <= x < pw - 1:
            v = cc3[i, y - 1 : y + 2, x]
            if v[2] != v[0]:
                py[i] += 0.5 * (v[0] - v[2]) / (v[0] - 2 * v[1] + v[2])
            v = cc3[i, y, x - 1 : x + 2]
            if v[2] != v[0]:
                px[i] += 0.5 * (v[0] - v[2]) / (v[0] - 2 * v[1] + v[2])
    return py, px


def _reject_outliers(sy, sx, thr):
    """xc.py:538-627 -- z = |s - lower median| / max(unbiased std, 1e-6); a patch with
    either axis beyond thr gets both axes replaced by the mean of the valid patches
    (median when none is valid)."""
    fy, fx = sy.flatten(), sx.flatten()
    my, mx = torch.median(fy), torch.median(fx)
    dy = torch.max(torch.std(fy), torch.tensor(1e-6))
    dx = torch.max(torch.std(fx), torch.tensor(1e-6))
    bad = (torch.abs(fy - my) / dy > thr) | (torch.abs(fx - mx) / dx > thr)
    ok_y, ok_x = fy[~bad], fx[~bad]
    ry = torch.mean(ok_y) if len(ok_y) > 0 else my
    rx = torch.mean(ok_x) if len(ok_x) > 0 else mx
    fy, fx = fy.clone(), fx.clone()
    fy[bad] = ry
    fx[bad] = rx
    return fy.view(sy.shape), fx.view(sx.shape)


def _smooth_time(field, window):
    """xc.py:486-535 -- scipy savgol_filter(window, polyorder=1) per patch per axis."""
    if window % 2 == 0:
        window += 1
    window = min(window, field.shape[1])
    if window < 3:
        return field
    out = field.clone()
    for gy in range(field.shape[2]):
        for gx in range(field.shape[3]):
            for c in (0, 1):
                series = field[c, :, gy, gx].numpy()
                if len(series) >= window:
                    out[c, :, gy, gx] = torch.from_numpy(savgol_filter(series, window, 1))
    return out


def estimate_motion_cross_correlation_patches(
    image, pixel_spacing, reference_frame=None, reference_strategy="mean_except_current",
    b_factor=500, frequency_range=(300, 10), patch_sidelength=1024, sub_pixel_refinement=True,
    temporal_smoothing=True, smoothing_window_size=5, deformation_field=None,
    outlier_rejection=True, outlier_threshold=3.0, device=None,
):
    """xc.py:138-411.  Returns ((2,t,gh,gw) Angstrom mean-subtracted field (Q6),
    (t,gh,gw,3) int64 centres).  The lazy patch memo aliasing (Q2/Q3) is reproduced
    by running the same in-place ops on oracle.patch_grid.LazyPatches."""
    image = image.float().cpu()
    t, h, w = image.shape
    ref = t // 2 if reference_frame is None else reference_frame
    image = normalize_image(image)  # Q9: before the optional pre-correction
    if deformation_field is not None:
        deformation_field = deformation_field.cpu()
        if deformation_field.shape[-2:] == (1, 1):
            image = correct_motion_fast(image, deformation_field)  # Q1 side effect kept
        else:
            image = correct_motion(image, deformation_field, pixel_spacing, grid_type="bspline")
    p = patch_sidelength
    lazy, centers = patch_grid_lazy(image, (1, p, p), (1, p // 2, p // 2), True)
    gh, gw = centers.shape[1:3]
    mask, benv, band = _filters((p, p), pixel_spacing, b_factor, frequency_range)
    if deformation_field is None:
        field = torch.zeros((2, t, gh, gw))
    else:
        field = resample_deformation_field(deformation_field, (t, gh, gw))

    for f in range(t):
        if reference_strategy == "middle_frame":
            if f == ref:
                continue
            ref_p = lazy[ref][0, :, :, 0]  # view of the memo entry
        elif reference_strategy == "mean_except_current":
            ref_p, n = None, 0
            for o in range(t):
                if o == f:
                    continue
                other = lazy[o][0, :, :, 0]
                if ref_p is None:
                    ref_p = other.clone()
                else:
                    ref_p += other
                n += 1
            ref_p = ref_p / n
        else:
            raise ValueError(f"Unknown reference_strategy: {reference_strategy}")
        cur = lazy[f][0, :, :, 0]  # view of the memo entry
        ref_p *= mask  # in place: mutates the memo for middle_frame (Q2)
        rs = torch.fft.rfftn(ref_p, dim=(-2, -1)) * band * benv
        cur *= mask  # in place: mutates the memo (Q2)
        cs = torch.fft.rfftn(cur, dim=(-2, -1)) * band * benv
        cc = torch.fft.irfftn(torch.conj(rs) * cs, s=(p, p)).reshape(gh * gw, p * p)
        peak = torch.argmax(cc, dim=1)
        if sub_pixel_refinement:
            py, px = _sub_pixel(cc.view(gh * gw, p, p), peak, p, p)
        else:
            py, px = peak // p, peak % p
        sy = torch.where(py <= p // 2, py, py - p).view(gh, gw)
        sx = torch.where(px <= p // 2, px, px - p).view(gh, gw)
        if outlier_rejection:
            sy, sx = _reject_outliers(sy, sx, outlier_threshold)
        field[0, f] += sy * pixel_spacing
        field[1, f] += sx * pixel_spacing

    if temporal_smoothing:
        field = _smooth_time(field, smoothing_window_size)
    field = field - torch.mean(field)  # Q6: one scalar for both channels
    return field, centers


def dose_weighted_sum(movie, pixel_spacing, dose_per_frame, pre_exposure=0.0, voltage=300.0):
    """The reference pipeline's ``dose_weight(movie)`` (examples/ttMotion.py:331-351): rfft2
    (ortho) -> dose_weight_movie -> irfft2 (ortho) -> sum over frames.  Third-party filter
    semantics from thirdparty_semantics.dose_weight_movie: parity unpinned."""
    shape = (movie.shape[-2], movie.shape[-1])
    dft = torch.fft.rfft2(movie.float(), dim=(-2, -1), norm="ortho")
    dw = tp.dose_weight_movie(dft, shape, pixel_spacing, pre_exposure, dose_per_frame, voltage)
    return torch.fft.irfft2(dw, s=shape, dim=(-2, -1), norm="ortho").sum(dim=0)
