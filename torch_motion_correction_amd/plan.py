"""Host-side plans: pruned-transform geometry, twiddle tables, cached mask / filter.

Everything here is small scalar/1-D host logic plus calls into libmcorr for the
device-resident constants (mask via an exact EDT kernel, filter table).  Plans are
cached per (device, shape, parameters) so steady-state calls launch no plan work
(the reference rebuilds mask and filters on every call, estimate_motion_xc.py:69-95).
"""

from __future__ import annotations

import math
from dataclasses import dataclass

import numpy as np
import torch

from . import _lib
from ._lib import XcGeom, XcLine, check, ptr, stream_ptr

LDS_BUDGET = 80 * 1024  # bytes per workgroup for the row kernels (2 workgroups / CU)


def _is_pow2(n: int) -> bool:
    return n > 0 and (n & (n - 1)) == 0


def band_limits(frequency_range, pixel_spacing):
    """(low, high) in cycles/pixel as float32 scalars, computed with the same torch ops
    as the reference's prepare_bandpass_filter (utils.py:97-101)."""
    cuton, cutoff_max = torch.as_tensor(frequency_range).float()
    cutoff = torch.lerp(cuton, cutoff_max, 1.0)
    low = torch.as_tensor(1 / cuton, dtype=torch.float32) * pixel_spacing
    high = torch.as_tensor(1 / cutoff, dtype=torch.float32) * pixel_spacing
    return float(low), float(high)


def xc_geometry(h: int, w: int, high: float, radius: float, smoothing: float) -> XcGeom:
    """Pruning bounds for an (h, w) transform: which rfft columns / fft rows can be
    non-zero under the band-pass `f <= high`, and which window rows/columns can be
    non-zero under the soft disk mask.  All bounds are conservative supersets."""
    if not (4 <= w <= 16384 and (w % 2 == 0 or w <= 8191) and 2 <= h <= 8192):
        raise NotImplementedError(
            f"transform size {h}x{w}: libmcorr handles even widths up to 16384 (odd widths up to 8191) and "
            "heights up to 8192 (power-of-two lengths up to 8192 x 4096 natively, everything else by chirp-z)"
        )
    hi32 = np.float32(high)
    fx = np.arange(w // 2 + 1, dtype=np.float32) * np.float32(1.0 / w)
    nkx = int(np.nonzero(fx <= hi32)[0].max()) + 1 if (fx <= hi32).any() else 1
    ky = np.arange(h)
    kk = np.where(ky < (h + 1) // 2, ky, ky - h)
    fy = np.abs(kk.astype(np.float32) * np.float32(1.0 / h))
    keep = fy <= hi32
    if not keep.any():
        keep[0] = True
    if keep.all():
        kyp, kyn = h, 0
    else:
        kyp = int(np.argmin(keep))  # first False
        kyn = int(np.argmin(keep[::-1]))  # trailing Trues
    # mask support
    cy, cx = h // 2, w // 2
    reach = int(math.ceil(radius + smoothing)) + 2
    y0, y1 = max(0, cy - reach), min(h, cy + reach + 1)
    x0, x1 = max(0, cx - reach), min(w, cx + reach + 1)
    if w % 2 == 0:  # two samples per point of the row line: whole pairs are in or out
        x0 -= x0 & 1
        x1 += x1 & 1
    n_line = row_line_length(w)
    native_w = native_width(w)
    if native_w:  # the power-of-two row kernels transform `subgroups` rows side by side: RG must be a
        # multiple of that and divide h; heights that allow no such grouping (odd heights with
        # narrow frames) take the chirp-z row kernels, which have no such constraint
        sg = row_subgroups(w)
        rg0 = 16
        while rg0 > sg and h % rg0:
            rg0 //= 2
        if rg0 < sg or h % rg0:
            native_w = False
    if native_w:
        subgroups = row_subgroups(w)  # rows transformed side by side (mc_fft.h)
        lines = subgroups * 2 * (n_line + (n_line >> 4) + 1)
        rg = 16
    else:  # chirp-z rows: one line of M = pow2 >= 2*(w/2)-1 points + two small side buffers
        subgroups = 1
        m = min(bluestein_size(n_line), bluestein_size_for(n_line + 2 * row_line_keep(w, nkx) - 1))  # line_plan(keep=...)
        lines = (m + (m >> 4) + 1) + 2 * (nkx + 1)
        rg = 4
    while rg > subgroups and 8 * (lines + nkx * (rg + 1)) > (LDS_BUDGET if native_w else 150 * 1024):
        rg //= 2
    if not native_w:  # the inverse row pass holds a classic (unpruned) line + nkx * (rg + 1) staged bins
        mi = bluestein_size(n_line)
        while rg > 1 and 8 * ((mi + (mi >> 4) + 1) + nkx * (rg + 1)) > 160 * 1024:
            rg //= 2
        if max(8 * (lines + nkx * (rg + 1)), 8 * ((mi + (mi >> 4) + 1) + nkx * (rg + 1))) > 160 * 1024:
            raise NotImplementedError(
                f"transform size {h}x{w} with {nkx} kept columns: a chirp-z row line plus its staged bins "
                "exceeds the 160 KB of LDS (band-limited transforms of frames this wide fit; the full "
                "spectrum that correct_motion_fast needs does not)")
    while rg > 1 and h % rg:
        rg //= 2
    rg = min(rg, h)
    if rg < subgroups or h % rg:
        raise NotImplementedError(f"transform size {h}x{w}: no valid row grouping")
    y0 -= y0 % rg
    y1 = min(h, ((y1 + rg - 1) // rg) * rg)
    return XcGeom(W=w, H=h, nkx=nkx, kyp=kyp, kyn=kyn, y0=y0, ny=y1 - y0, x0=x0, x1=x1, RG=rg)


def full_geometry(h: int, w: int) -> XcGeom:
    """No pruning at all (correct_motion_fast needs the full spectrum)."""
    return xc_geometry(h, w, high=10.0, radius=float(max(h, w)), smoothing=0.0)


def row_line_length(w: int) -> int:
    """Points of one chirp-z row line: two real samples per complex point for even widths, one for
    odd widths (no packing identity there)."""
    return w // 2 if w % 2 == 0 else w


def row_line_keep(w: int, nkx: int) -> int:
    """Outputs the forward row pass needs on either side of zero (plan.line_plan(keep=...)): the
    packed form also needs the mirror Z[n - k] of every kept bin."""
    return nkx + 1 if w % 2 == 0 else nkx


def row_subgroups(w: int) -> int:
    """Rows the power-of-two row kernels transform side by side in one workgroup (mc_fft.h)."""
    return 256 // min(256, max(64, (w // 2) // 8))


def native_rows(g) -> bool:
    """This geometry's row passes run on the power-of-two kernels (xc_geometry falls back to the
    chirp-z rows when the height admits no row grouping for them: RG then is not a multiple of the
    side-by-side row count)."""
    return native_width(g.W) and g.RG % row_subgroups(g.W) == 0


def native_width(w: int) -> bool:
    """Row transforms of this width run on the power-of-two kernels (else chirp-z)."""
    return _is_pow2(w) and 32 <= w <= 8192


def native_height(h: int) -> bool:
    return _is_pow2(h) and 16 <= h <= 4096


DIRECT_LINE_LENGTHS = (2880, 5760, 4092, 8184)   # mixed-radix lines (2^a 3^2 5; 2^a 3 11 31) transformed as they are: no chirp-z
SMOOTH_CHIRP_LENGTHS = (5120, 10240)  # chirp-z lengths 2^k 5 next to the powers of two


def bluestein_size_for(length: int) -> int:
    """Smallest chirp-z length libmcorr has kernels for: a power of two, or 5120 / 10240."""
    m = 32
    while m < length:
        m *= 2
    for c in SMOOTH_CHIRP_LENGTHS:
        if length <= c < m:
            m = c
    return m


def bluestein_size(n: int) -> int:
    return bluestein_size_for(2 * n - 1)


_LINES: dict = {}
USE_DIRECT_LINES = True  # tests: False forces chirp-z on the mixed-radix lengths too


def line_plan(n: int, direction: int, device, keep: int = 0):
    """Chirp-z tables for a length-n transform (any n), direction -1 forward / +1 inverse:
    chirp[j] = exp(direction * i*pi*j^2/n) with j^2 reduced mod 2n in integers, bspec =
    FFT_M(wrapped conj(chirp)) / M.  Returns (XcLine, keep-alive tensors).

    keep > 0: output-pruned plan for the forward row pass.  Only outputs k in [0, keep) and
    (n - keep, n) are wanted, i.e. signed k in (-keep, keep); with inputs j in [0, n) the
    convolution only touches offsets k - j in (-(n-1) - (keep-1) .. keep-1), so a circular
    length M >= n + 2 keep - 1 suffices (4096 instead of 8192 for 5760-wide frames)."""
    keep = int(keep)
    if n in DIRECT_LINE_LENGTHS and USE_DIRECT_LINES:
        # the line length itself factors into 2, 3 and 5: transformed directly (xcg_line_fft's direct
        # codes), tw_m = exp(-2 pi i k / n); chirp / bspec are unused
        key = (str(device), n, 0, 0)
        if key not in _LINES:
            tw = get_twiddles(n, device)
            _LINES[key] = (XcLine(tw_m=tw.data_ptr(), chirp=tw.data_ptr(), bspec=tw.data_ptr(), M=n, keep=0), (tw,))
        return _LINES[key]
    if keep > 0 and bluestein_size_for(n + 2 * keep - 1) >= bluestein_size(n):
        keep = 0  # nothing to gain: same circular length as the classic plan
    key = (str(device), n, direction, keep)
    if key in _LINES:
        return _LINES[key]
    m = bluestein_size_for(n + 2 * keep - 1) if keep > 0 else bluestein_size(n)
    if m > 16384:
        raise NotImplementedError(f"transform length {n}: chirp-z needs M={m} > 16384")
    chirp_at = lambda j: np.exp(1j * direction * np.pi * ((j * j) % (2 * n)).astype(np.float64) / n)
    chirp = chirp_at(np.arange(n, dtype=np.int64))
    bw = np.zeros(m, dtype=np.complex128)
    if keep > 0:
        off = np.arange(-(n - 1) - (keep - 1), keep, dtype=np.int64)  # every offset k - j that occurs
        bw[off % m] = np.conj(chirp_at(off))
    else:
        b = np.conj(chirp)
        bw[:n] = b
        if n > 1:
            bw[m - n + 1:] = b[1:][::-1]
    bspec = np.fft.fft(bw) / m
    to_dev = lambda z: torch.from_numpy(np.stack([z.real, z.imag], -1).astype(np.float32)).to(device)
    tw_m, ch, bs = get_twiddles(m, device), to_dev(chirp), to_dev(bspec)
    line = XcLine(tw_m=tw_m.data_ptr(), chirp=ch.data_ptr(), bspec=bs.data_ptr(), M=m, keep=keep)
    _LINES[key] = (line, (tw_m, ch, bs))
    return _LINES[key]


def twiddles(n: int, device) -> torch.Tensor:
    """exp(-2 pi i k / n), k in [0, n), complex64 stored as (n, 2) float32."""
    k = np.arange(n, dtype=np.float64)
    ang = -2.0 * np.pi * k / n
    tw = np.stack([np.cos(ang), np.sin(ang)], axis=-1).astype(np.float32)
    return torch.from_numpy(tw).to(device)


@dataclass
class XcPlan:
    geom: XcGeom
    mask: torch.Tensor  # (H, W) float32
    filt: torch.Tensor  # (nkx, nky) float32
    tw_row: torch.Tensor  # (W, 2)
    tw_col: torch.Tensor  # (H, 2)
    low: float
    high: float
    mhat: object = None  # pruned spectrum of the mask (engine.mask_spectrum), built lazily
    chord: object = None  # (H, 2) int32: per-row [first, last] 4-aligned column with a non-zero mask quad


def row_chords(mask: torch.Tensor) -> torch.Tensor:
    """Per window row, the first and the last 4-aligned column whose quad holds a non-zero mask
    value (rows without any: the middle quad).  K1 clamps its sample loads to that range."""
    h, w = mask.shape
    nz = (mask != 0).reshape(h, w // 4, 4).any(dim=2)  # (h, w/4) quads
    any_ = nz.any(dim=1)
    first = torch.argmax(nz.to(torch.int32), dim=1)
    last = (w // 4 - 1) - torch.argmax(nz.flip(1).to(torch.int32), dim=1)
    mid = torch.full_like(first, (w // 8))
    first = torch.where(any_, first, mid)
    last = torch.where(any_, last, mid)
    return (torch.stack([first, last], dim=1) * 4).to(torch.int32).contiguous()


_PLANS: dict = {}
_TWIDDLES: dict = {}


def get_twiddles(n: int, device) -> torch.Tensor:
    key = (str(device), n)
    if key not in _TWIDDLES:
        _TWIDDLES[key] = twiddles(n, device)
    return _TWIDDLES[key]


def circle_mask(h: int, w: int, radius: float, smoothing: float, device) -> torch.Tensor:
    lib = _lib.load()
    mask = torch.empty((h, w), dtype=torch.float32, device=device)
    halfw = torch.empty((h,), dtype=torch.int32, device=device)
    check(lib.mc_circle_mask(ptr(mask), ptr(halfw), h, w, radius, smoothing, stream_ptr(device)),
          "mc_circle_mask")
    return mask


def get_xc_plan(h: int, w: int, pixel_spacing: float, b_factor: float, frequency_range,
                device, mask_radius=None, mask_smoothing=None) -> XcPlan:
    """Plan for cross-correlating (h, w) windows: mask radius min(h,w)/4, soft edge
    min(h,w)/8 (estimate_motion_xc.py:69-74 / :262-264) unless given (estimate_local_motion
    uses pw/4 and pw/4, estimate_motion_optimizer.py:162-167)."""
    radius = min(h, w) / 4 if mask_radius is None else float(mask_radius)
    smoothing = min(h, w) / 8 if mask_smoothing is None else float(mask_smoothing)
    key = (str(device), h, w, float(pixel_spacing), float(b_factor), tuple(map(float, frequency_range)),
           radius, smoothing)
    if key in _PLANS:
        return _PLANS[key]
    lib = _lib.load()
    low, high = band_limits(frequency_range, pixel_spacing)
    geom = xc_geometry(h, w, high, radius, smoothing)
    mask = circle_mask(h, w, radius, smoothing, device)
    filt = torch.empty((geom.nkx, geom.nky), dtype=torch.float32, device=device)
    check(lib.mc_xc_filter(ptr(filt), geom, low, high, float(b_factor), float(pixel_spacing),
                           stream_ptr(device)), "mc_xc_filter")
    plan = XcPlan(geom, mask, filt, get_twiddles(w, device), get_twiddles(h, device), low, high)
    if w % 4 == 0:
        plan.chord = row_chords(mask)
    _PLANS[key] = plan
    return plan


def clear_plan_cache():
    _PLANS.clear()
    _TWIDDLES.clear()
    _LINES.clear()
