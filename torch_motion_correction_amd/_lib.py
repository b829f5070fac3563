"""ctypes binding of libmcorr.so (the C ABI declared in include/mcorr.h).

There is deliberately no fallback: if the library is missing or a call fails the
caller gets an exception.  Nothing here touches the oracle.
"""

from __future__ import annotations

import ctypes as C
import os

import torch

_PKG_DIR = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_PKG_DIR, "libmcorr.so")

vp, i32, i64, f32 = C.c_void_p, C.c_int, C.c_int64, C.c_float


class XcGeom(C.Structure):
    """mirror of ``mc_xc_geom``"""

    _fields_ = [(n, C.c_int) for n in ("W", "H", "nkx", "kyp", "kyn", "y0", "ny", "x0", "x1", "RG")]

    @property
    def nky(self) -> int:
        return self.kyp + self.kyn


class XcLine(C.Structure):
    """mirror of ``mc_xc_line`` (Bluestein line plan: device pointers + M)"""

    _fields_ = [("tw_m", C.c_void_p), ("chirp", C.c_void_p), ("bspec", C.c_void_p), ("M", C.c_int),
                ("keep", C.c_int)]


GP = C.POINTER(XcGeom)
LP = C.POINTER(XcLine)

# name -> argtypes; every function returns int
SIGNATURES = {
    "mc_abi_version": [],
    "mc_circle_mask": [vp, vp, i32, i32, f32, f32, vp],
    "mc_xc_filter": [vp, GP, f32, f32, f32, f32, vp],
    "mc_central_box_stats": [vp, i32, i32, i32, i32, i32, i32, i32, vp, vp, vp],
    "mc_central_box_stats_t": [vp, i32, i32, i32, i32, i32, i32, i32, i32, vp, vp, vp],
    "mc_normalize": [vp, vp, i64, vp, vp],
    "mc_xc_rows_lds_bytes": [GP],
    "mc_xc_row_engine": [i32],
    "mc_xc_col_engine": [i32],
    "mc_xc_after_k3n_event": [vp],
    "mc_xc_rows_forward": [vp, vp, i64, vp, vp, vp, vp, vp, i32, GP, vp],
    "mc_xc_rows_forward_dual": [vp, vp, i64, vp, vp, vp, vp, vp, vp, vp, i32, GP, vp, vp],
    "mc_xc_rows_forward_dual_t": [vp, i32, vp, i64, vp, vp, vp, vp, vp, vp, vp, i32, GP, vp, vp],
    "mc_xc_rows_forward_stats": [vp, vp, i64, vp, vp, vp, vp, i32, GP, i32, i32, i32, i32, vp, vp, vp, vp, vp],
    "mc_xc_rows_forward_stats_t": [vp, i32, vp, i64, vp, vp, vp, vp, i32, GP, i32, i32, i32, i32, vp, vp, vp, vp, vp],
    "mc_xc_cols_forward": [vp, vp, vp, vp, i32, GP, vp],
    "mc_xc_cols_forward_fix": [vp, vp, vp, vp, i32, GP, vp, vp, vp],
    "mc_xc_cols_inverse": [vp, vp, vp, vp, vp, vp, f32, i32, GP, vp],
    "mc_xc_rows_inverse_argmax": [vp, vp, vp, vp, vp, vp, i32, GP, vp],
    "mc_xc_near_rows": [GP],
    "mc_xc_correlate_argmax": [vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, i32, vp, vp, vp, f32, i32, GP, vp],
    "mc_xc_provisional_mean": [vp, i32, vp, vp],
    "mc_xc_provisional_mean_t": [vp, i32, i32, vp, vp],
    "mc_xc_peak_neighbourhood": [vp, vp, vp, vp, i32, GP, vp],
    "mc_xc_ref_mean_except_current": [vp, vp, vp, vp, vp, vp, i32, i32, i64, f32, vp],
    "mc_field_accumulate": [vp, vp, vp, i32, i32, i32, i32, f32, f32, i32, vp, vp],
    "mc_field_smooth_center": [vp, vp, i32, i32, i32, i32, vp],
    "mc_spline_lattice": [vp, i32, i32, i32, i32, vp, vp, i32, vp, vp, i32, vp, vp, i32, vp, vp],
    "mc_warp_scratch_bytes": [i32, i32, i32, i32, i32, C.POINTER(C.c_int64)],
    "mc_warp_frames": [vp, i32, i32, i32, vp, i32, i32, f32, vp, vp, vp, vp],
    "mc_warp_frames_t": [vp, i32, i32, i32, i32, vp, i32, i32, f32, vp, vp, vp, vp],
    "mc_warp_rigid_scratch_bytes": [i32, i32, i32, C.POINTER(C.c_int64)],
    "mc_warp_rigid": [vp, i32, i32, i32, vp, vp, vp, vp, vp],
    "mc_warp_rigid_phase": [vp, i32, i32, i32, vp, vp, vp, vp, i32, vp],
    "mc_warp_rigid_phase_t": [vp, i32, i32, i32, i32, vp, vp, vp, vp, i32, vp],
    "mc_pixel_shifts": [vp, i32, i32, i32, i32, f32, vp, vp, vp],
    "mc_pixel_shifts_at": [vp, i32, i32, i32, i32, f32, vp, i64, vp, vp],
    "mc_full_spectrum_pitch": [i32],
    "mc_full_rows_forward": [vp, vp, i64, vp, vp, i32, i32, i32, i32, vp],
    "mc_full_cols_shift": [vp, vp, vp, f32, i32, i32, i32, i32, vp],
    "mc_full_cols_dose": [vp, i32, i32, i32, vp, vp, i32, i32, i32, f32, f32, f32, f32, i32, i32, f32, vp],
    "mc_full_cols_dose_cm": [vp, i32, i32, i32, vp, vp, i32, i32, i32, f32, f32, f32, f32, i32, i32, f32, vp],
    "mc_full_transpose": [vp, vp, i32, i32, i32, i32, vp],
    "mc_full_rows_inverse": [vp, vp, vp, i64, vp, i32, i32, i32, i32, vp],
    "mc_fourier_shift_cols_inverse": [vp, vp, vp, vp, vp, f32, i32, GP, vp],
    "mc_xc_rows_inverse_store": [vp, vp, vp, i64, vp, i32, GP, vp],
    "mc_xcg_rows_forward": [vp, vp, i64, vp, vp, vp, vp, vp, LP, i32, GP, vp],
    "mc_xcg_cols_forward": [vp, vp, vp, LP, i32, GP, vp],
    "mc_xcg_cols_inverse": [vp, vp, vp, vp, vp, vp, LP, f32, i32, GP, vp],
    "mc_xcg_rows_inverse": [vp, vp, vp, vp, vp, vp, vp, i64, vp, LP, i32, GP, vp],
    "mc_xcg_peak_neighbourhood": [vp, vp, vp, i32, GP, vp],
    "mc_sum_frames": [vp, i32, i64, vp, vp],
    "mc_condition_movie": [vp, i32, vp, i32, i64, i32, vp, vp, vp],
    "mc_raw_movie_stats": [vp, i32, vp, i32, i32, i32, i32, i32, i32, i32, i32, vp, vp, vp, vp, vp],
    "mc_xc_rows_forward_raw": [vp, i32, vp, vp, i64, vp, vp, vp, vp, vp, i32, GP, vp, vp],
    "mc_xcg_rows_forward_raw": [vp, i32, vp, vp, i64, vp, vp, vp, vp, vp, LP, i32, GP, vp],
    "mc_rigid_tables_from_shifts": [vp, f32, vp, vp, vp, vp, i32, i32, i32, vp, vp, vp, vp],
    "mc_warp_rigid_raw": [vp, i32, vp, vp, i32, i32, i32, vp, vp, vp, vp, i32, vp],
    "mc_condition_movie_hot": [vp, i32, vp, i32, i32, i32, i32, f32, vp, vp, vp, vp],
    "mc_spline_points": [vp, i32, i32, i32, i32, vp, vp, vp, vp, vp, vp, i64, vp, vp],
    "mc_dose_accumulate": [vp, i32, i32, i32, vp, i32, i32, f32, f32, f32, f32, i32, i32, vp],
    "mc_polyphase_fourier_shift": [vp, vp, i32, i32, i32, i32, vp],
    "mc_polyphase_dose_accumulate": [vp, i32, i32, i32, vp, i32, i32, i32, f32, f32, f32, f32, i32, i32, vp],
    "mc_local_loss_tiles": [i32, i32, vp],
    "mc_local_loss_sums": [vp, vp, vp, vp, vp, i32, i32, i32, i32, vp, vp],
    "mc_local_ncc_grad": [vp, vp, vp, vp, vp, vp, i32, i32, i32, i32, vp, vp],
}

_lib = None


class McorrError(RuntimeError):
    pass


def load():
    """Load libmcorr.so; raises McorrError (never falls back) when it is absent."""
    global _lib
    if _lib is not None:
        return _lib
    path = os.environ.get("MCORR_LIB", LIB_PATH)  # experiments: an A/B build from scripts/build_variant.sh
    if not os.path.exists(path):
        raise McorrError(
            f"{LIB_PATH} not found: build it with `python -m torch_motion_correction_amd._build` "
            "(hipcc --offload-arch=gfx950). There is no CPU fallback."
        )
    lib = C.CDLL(path)
    for name, argtypes in SIGNATURES.items():
        fn = getattr(lib, name)  # AttributeError if the symbol is missing
        fn.argtypes = argtypes
        fn.restype = C.c_int
    if lib.mc_abi_version() != 1:
        raise McorrError("libmcorr ABI version mismatch; rebuild the library")
    _lib = lib
    return lib


class McorrUnsupported(McorrError):
    """An entry point answered MC_ERR_UNSUPPORTED (-2): this shape / storage type has no kernel built in.
    For fp16 stacks the C side is the single authority on which shapes are read natively; callers catch
    this, widen the stack once and take the fp32 entry point."""


def check(rc: int, what: str):
    if rc == -2:
        raise McorrUnsupported(f"{what} failed: unsupported size/mode")
    if rc != 0:
        kind = {-1: "bad argument"}.get(rc, f"hipError {rc}")
        raise McorrError(f"{what} failed: {kind}")


def ptr(t):
    """device pointer of a tensor (or None)"""
    if t is None:
        return None
    assert t.is_contiguous(), "libmcorr needs contiguous buffers"
    return C.c_void_p(t.data_ptr())


def stream_ptr(device) -> C.c_void_p:
    """Current stream of `device` as a hipStream_t.  libmcorr launches on the CURRENT HIP device,
    so the buffers' device must be the current one: every entry point of the package runs inside
    ``device_scope`` (torch.cuda.device); anything that slipped past it fails here instead of
    launching on the wrong GPU."""
    d = torch.device(device)
    if d.type != "cuda":
        raise McorrError(f"libmcorr buffers must live on a ROCm device, got {d}")
    cur = torch.cuda.current_device()
    if d.index is not None and d.index != cur:
        raise McorrError(f"buffers are on cuda:{d.index} but the current device is cuda:{cur}: "
                         "enter torch_motion_correction_amd._lib.device_scope(device) first")
    return C.c_void_p(torch.cuda.current_stream(d).cuda_stream)


def device_scope(device):
    """Context manager making `device` the current HIP device for the kernels enqueued inside
    (the reference takes ``device=`` / ``image.device`` per call and has no notion of a current
    device: estimate_motion_xc.py:52-55, correct_motion.py:49-53)."""
    return torch.cuda.device(torch.device(device))


def normalize_frame_index(reference_frame, t: int) -> int:
    """The reference indexes ``filtered_fft[reference_frame]`` / ``lazy_patch_grid[reference_frame]``
    (estimate_motion_xc.py:101, :306): Python semantics, negative values count from the end and
    anything outside [-t, t) raises IndexError.  Returns the wrapped index; no raw user value
    ever reaches a device-side table."""
    import operator

    r = operator.index(reference_frame)
    if not -t <= r < t:
        raise IndexError(f"index {r} is out of bounds for dimension 0 with size {t}")
    return r % t


def require_gpu(device=None) -> torch.device:
    """The product path runs on the MI355X only."""
    if not torch.cuda.is_available():
        raise McorrError(
            "torch_motion_correction_amd needs a ROCm GPU (gfx950); no GPU is visible and there "
            "is no CPU fallback"
        )
    if device is None or torch.device(device).type != "cuda":
        return torch.device("cuda", torch.cuda.current_device())
    d = torch.device(device)
    return d if d.index is not None else torch.device("cuda", torch.cuda.current_device())
