"""Generate tests/golden/*.npz.  TEST INFRASTRUCTURE ONLY.

Run in the build container (where /root/reference exists):

    python -m oracle.make_goldens

Two kinds of vectors are written:

* ``patch_grid_reference.npz`` -- produced by the REFERENCE's own code: its
  ``patch_grid`` sub-package is importable on its own (it only needs torch+einops;
  the package ``__init__`` is bypassed because it eagerly imports the five absent
  teamtomo dependencies).  Contents: 1-D patch centres for the BASELINE.json
  shapes, a full lazy gather on a small stack, and the mask-exponent tables that
  result from replaying the reference's per-frame loop (xc.py:297-346) on the
  reference's LazyPatchGrid with a scalar "mask" of 2.0 (log2 of the value read =
  number of times that memo entry had been multiplied in place).
* ``reference_helpers.npz`` -- also produced by the REFERENCE's own code: utils.py,
  estimate_motion_xc.py, correct_motion.py and deformation_field_utils.py are imported with
  the five absent packages bound to stubs that raise when called, and the reference's
  torch/scipy-only helpers are run on seeded inputs: normalize_image, array_to_grid_sample,
  image_shifts_to_deformation_field, _apply_sub_pixel_refinement, _apply_outlier_rejection,
  _apply_temporal_smoothing, get_pixel_shifts.  Nothing third-party is emulated.
* ``oracle_*.npz`` -- outputs of the oracle itself (parity unpinned at the
  third-party boundaries) on the reference's test fixtures and on the SURVEY
  section 8d synthetic drift stack; used as regression pins and as the expected
  values for the GPU parity tests.
"""

from __future__ import annotations

import importlib
import os
import sys
import types

import numpy as np
import torch

GOLD = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden")
REF_SRC = "/root/reference/src/torch_motion_correction"


def _reference_patch_grid():
    pkg = types.ModuleType("torch_motion_correction")
    pkg.__path__ = [REF_SRC]
    sys.modules["torch_motion_correction"] = pkg
    return importlib.import_module("torch_motion_correction.patch_grid")


def reference_vectors():
    pg = _reference_patch_grid()
    centers_mod = importlib.import_module("torch_motion_correction.patch_grid._patch_grid_centers")
    out = {}
    # 1-D centres for every (dim_length, patch, step) the configs need
    cases = [(64, 32), (512, 128), (256, 64), (4096, 1024), (4092, 1024), (5760, 1024),
             (8184, 1024), (11520, 1024), (100, 32), (33, 32), (30, 32), (959, 256), (927, 256)]
    for n, p in cases:
        c = centers_mod._patch_centers_1d(dim_length=n, patch_length=p, patch_step=p // 2,
                                          distribute_patches=True)
        out[f"centers_{n}_{p}"] = c.numpy()
    # full lazy gather on a small stack
    g = torch.Generator().manual_seed(7)
    img = torch.randn(3, 40, 52, generator=g)
    lazy, centers = pg.patch_grid_lazy(images=img, patch_shape=(1, 16, 16), patch_step=(1, 8, 8),
                                       distribute_patches=True)
    out["gather_img"] = img.numpy()
    out["gather_centers"] = centers.numpy()
    out["gather_frame1"] = lazy[1].numpy()

    # memo aliasing replay (Q2/Q3): exponent tables
    def replay(t, strategy):
        imgs = torch.ones(t, 8, 8)
        lz, _ = pg.patch_grid_lazy(images=imgs, patch_shape=(1, 4, 4), patch_step=(1, 2, 2),
                                   distribute_patches=True)
        ref = t // 2
        table = np.full((t, t), -1, dtype=np.int64)  # [frame, other] exponent when read
        cur_exp = np.full((t,), -1, dtype=np.int64)
        for f in range(t):
            if strategy == "middle_frame":
                if f == ref:
                    continue
                r = lz[ref]
                table[f, ref] = int(torch.log2(r.flatten()[0]).item())
                r = r.reshape(r.shape[1], r.shape[2], 4, 4)
            else:
                r = None
                for o in range(t):
                    if o == f:
                        continue
                    other = lz[o]
                    table[f, o] = int(torch.log2(other.flatten()[0]).item())
                    r = other.clone() if r is None else r + other
            cur = lz[f]
            cur_exp[f] = int(torch.log2(cur.flatten()[0]).item())
            if strategy == "middle_frame":
                r *= 2.0  # reshape of a contiguous tensor is a view, as einops.rearrange is
            cur *= 2.0
        return table, cur_exp

    for t in (5, 8, 40, 51, 60):
        for s in ("middle_frame", "mean_except_current"):
            tab, cur = replay(t, s)
            out[f"exp_{s}_{t}"] = tab
            out[f"cur_{s}_{t}"] = cur
    np.savez_compressed(os.path.join(GOLD, "patch_grid_reference.npz"), **out)
    print("wrote patch_grid_reference.npz", len(out), "arrays")


THIRD_PARTY = {  # absent packages -> names the reference imports from them at module level
    "torch_fourier_filter": [], "torch_fourier_filter.envelopes": ["b_envelope"],
    "torch_fourier_filter.bandpass": ["bandpass_filter"],
    "torch_grid_utils": ["circle", "coordinate_grid"],
    "torch_cubic_spline_grids": ["CubicBSplineGrid3d", "CubicCatmullRomGrid3d"],
    "torch_fourier_shift": ["fourier_shift_dft_2d"],
    "torch_image_interpolation": ["sample_image_2d"],
    "torch_image_interpolation.grid_sample_utils": ["array_to_grid_sample"],
}


def _reference_modules_behind_inert_stubs():
    """Import the reference's OWN modules (utils, estimate_motion_xc, correct_motion,
    deformation_field_utils).  Their top-level imports name five absent third-party packages;
    those names are bound to stubs that RAISE when called, so nothing third-party is emulated:
    only reference functions that use torch / scipy / einops alone can run (and only those are
    captured below).  The one exception is documented at get_pixel_shifts."""
    def refuse(name):
        def f(*a, **k):
            raise NotImplementedError(f"{name} is a third-party function that is absent here")
        return f

    for mod, names in THIRD_PARTY.items():
        m = types.ModuleType(mod)
        m.__path__ = []
        for n in names:
            if n[0].isupper():  # class names appear in type annotations: an inert class
                setattr(m, n, type(n, (), {"__init__": refuse(f"{mod}.{n}"),
                                           "from_grid_data": staticmethod(refuse(f"{mod}.{n}.from_grid_data"))}))
            else:
                setattr(m, n, refuse(f"{mod}.{n}"))
        sys.modules[mod] = m
    _reference_patch_grid()
    mods = {n: importlib.import_module(f"torch_motion_correction.{n}")
            for n in ("utils", "deformation_field_utils", "correct_motion", "estimate_motion_xc")}
    return mods


def reference_helper_vectors():
    """tests/golden/reference_helpers.npz: outputs of the reference's own helper functions
    (the code under /root/reference, imported, not restated) on seeded inputs."""
    import contextlib
    import io

    m = _reference_modules_behind_inert_stubs()
    xc, utils, dfu, cm = m["estimate_motion_xc"], m["utils"], m["deformation_field_utils"], m["correct_motion"]
    g = torch.Generator().manual_seed(2024)
    out = {}
    with contextlib.redirect_stdout(io.StringIO()):
        # a2 normalize_image (utils.py:49-84)
        img = torch.randn(4, 40, 56, generator=g) * 3.0 + 7.0
        out["norm_in"], out["norm_out"] = img.numpy(), utils.normalize_image(img).numpy()
        # array_to_grid_sample (utils.py:9-30)
        coords = torch.rand(5, 7, 2, generator=g) * torch.tensor([39.0, 55.0])
        out["a2g_in"], out["a2g_out"] = coords.numpy(), utils.array_to_grid_sample(coords, (40, 56)).numpy()
        # a7 image_shifts_to_deformation_field (deformation_field_utils.py:129-162)
        sh = torch.randn(6, 2, generator=g)
        out["s2f_in"] = sh.numpy()
        out["s2f_out"] = dfu.image_shifts_to_deformation_field(sh, pixel_spacing=1.7).numpy()
        # a11 _apply_sub_pixel_refinement (estimate_motion_xc.py:414-483): interior peaks, a
        # border peak (Q4) and a flat axis (Q5)
        ph, pw, n = 12, 16, 6
        cc = torch.randn(n, ph * pw, generator=g)
        peaks = torch.tensor([5 * pw + 7, 0 * pw + 3, 6 * pw + 15, 11 * pw + 8, 4 * pw + 4, 7 * pw + 9])
        cc[torch.arange(n), peaks] += 6.0
        cc[4].view(ph, pw)[3, 4] = cc[4].view(ph, pw)[5, 4]  # equal outer samples in y
        py, px = xc._apply_sub_pixel_refinement(cc, peaks, ph, pw)
        out["sp_cc"], out["sp_peaks"] = cc.numpy(), peaks.numpy()
        out["sp_y"], out["sp_x"] = py.numpy(), px.numpy()
        # a12 _apply_outlier_rejection (estimate_motion_xc.py:538-627): one outlier, none, all equal
        cases = []
        sy, sx = torch.randn(4, 5, generator=g) * 0.3, torch.randn(4, 5, generator=g) * 0.3
        sy[1, 2] = 25.0
        cases.append((sy, sx, 3.0))
        cases.append((torch.randn(3, 3, generator=g), torch.randn(3, 3, generator=g), 3.0))
        cases.append((torch.full((2, 3), 1.5), torch.full((2, 3), -0.5), 2.0))
        a, b = torch.randn(2, 2, generator=g), torch.randn(2, 2, generator=g)
        cases.append((a, b, 0.1))  # tiny threshold: everything is rejected -> median fallback
        for i, (a, b, thr) in enumerate(cases):
            ry, rx = xc._apply_outlier_rejection(a.clone(), b.clone(), thr, 0)
            out[f"or{i}_y"], out[f"or{i}_x"], out[f"or{i}_thr"] = a.numpy(), b.numpy(), np.float32(thr)
            out[f"or{i}_oy"], out[f"or{i}_ox"] = ry.numpy(), rx.numpy()
        # a13 _apply_temporal_smoothing (estimate_motion_xc.py:486-535): odd, even and too-long windows
        fld = torch.randn(2, 9, 2, 3, generator=g)
        out["ts_in"] = fld.numpy()
        for wdw in (5, 4, 3, 15, 2):
            out[f"ts_out_{wdw}"] = xc._apply_temporal_smoothing(fld.clone(), wdw, torch.device("cpu")).numpy()
        # even EFFECTIVE window: min(window | 1, t) with an even t below it (scipy accepts an even
        # window_length in mode="interp": the result is the least-squares line over the whole series)
        for tt, wdw in ((4, 5), (6, 7), (6, 9), (8, 11), (6, 5)):
            fe = torch.randn(2, tt, 2, 2, generator=g)
            out[f"tse_in_{tt}_{wdw}"] = fe.numpy()
            out[f"tse_out_{tt}_{wdw}"] = xc._apply_temporal_smoothing(fe.clone(), wdw, torch.device("cpu")).numpy()
        # a18 get_pixel_shifts (correct_motion.py:132-185).  It calls array_to_grid_sample under
        # the name it imports from torch_image_interpolation; the reference keeps an identical
        # copy of that function in its own utils.py:9-30, which is what is bound here.
        cm.array_to_grid_sample = utils.array_to_grid_sample
        frame = torch.zeros(37, 53)
        lattice = torch.randn(2, 20, 30, generator=g) * 2.0
        yy, xx = torch.meshgrid(torch.arange(37, dtype=torch.float32), torch.arange(53, dtype=torch.float32),
                                indexing="ij")
        pixel_grid = torch.stack([yy, xx], dim=-1)
        out["gps_lattice"] = lattice.numpy()
        out["gps_out"] = cm.get_pixel_shifts(frame, 1.3, lattice, pixel_grid).numpy()
        # a caller-supplied pixel_grid that is NOT the identity: a sub-grid with fractional
        # coordinates, a few of them outside the frame (reflection padding of the lattice)
        sub = torch.rand(9, 11, 2, generator=g) * torch.tensor([44.0, 60.0]) - torch.tensor([4.0, 4.0])
        out["gps_sub_grid"] = sub.numpy()
        out["gps_sub_out"] = cm.get_pixel_shifts(frame, 1.3, lattice, sub).numpy()
    np.savez_compressed(os.path.join(GOLD, "reference_helpers.npz"), **out)
    print("wrote reference_helpers.npz", len(out), "arrays")


def reference_body_vectors():
    """tests/golden/reference_bodies_with_standins.npz: the reference's own FUNCTION BODIES --
    estimate_global_motion (xc.py:21-135), estimate_motion_cross_correlation_patches (:138-411,
    both strategies, t = 52 so that the memo eviction runs, rigid and full prior fields),
    correct_motion (cm.py:18-78, both bases) and correct_motion_fast (:430-498) -- executed from
    /root/reference with the names they import from the five absent packages bound to
    ``oracle.thirdparty_semantics``.

    THIS FILE DOES NOT PIN THIRD-PARTY SEMANTICS: the stand-ins are this repo's own restatement,
    so agreement says nothing about circle / b_envelope / bandpass_filter / fourier_shift_dft_2d /
    the spline grids / sample_image_2d ("parity unpinned" stays).  What it does pin is the
    RESTATEMENT of the reference's own control flow in oracle/motion.py (ordering of
    normalisation, pre-correction, memo aliasing Q1/Q2/Q3, accumulation, smoothing, mean
    subtraction): the oracle must reproduce these outputs bit for bit."""
    import contextlib
    import io

    from oracle import thirdparty_semantics as tp

    m = _reference_modules_behind_inert_stubs()
    xc, utils, dfu, cm = m["estimate_motion_xc"], m["utils"], m["deformation_field_utils"], m["correct_motion"]

    def grid_class(kind):
        class StandInGrid:
            def __init__(self, data):
                self.data = data

            @classmethod
            def from_grid_data(cls, data):
                return cls(data)

            def to(self, device):
                return self

            def __call__(self, tyx):
                return tp.cubic_spline_grid_3d(self.data, tyx, kind)

        return StandInGrid

    saved = {}

    def bind(mod, name, value):
        saved[(mod, name)] = getattr(mod, name)
        setattr(mod, name, value)

    bind(xc, "circle", tp.circle)
    bind(xc, "b_envelope", tp.b_envelope)
    bind(utils, "bandpass_filter", tp.bandpass_filter)
    bind(cm, "fourier_shift_dft_2d", tp.fourier_shift_dft_2d)
    bind(cm, "coordinate_grid", tp.coordinate_grid)
    bind(cm, "sample_image_2d", tp.sample_image_2d)
    bind(cm, "array_to_grid_sample", utils.array_to_grid_sample)  # the reference's own copy
    for mod in (dfu, cm):
        bind(mod, "CubicBSplineGrid3d", grid_class("bspline"))
        bind(mod, "CubicCatmullRomGrid3d", grid_class("catmull_rom"))
    out = {}
    try:
        with contextlib.redirect_stdout(io.StringIO()):
            mov, stat = blob_stack(True), blob_stack(False)
            out["blob_global"] = xc.estimate_global_motion(mov, 1.0).numpy()
            out["blob_global_ref0"] = xc.estimate_global_motion(mov, 1.0, reference_frame=0).numpy()
            out["blob_global_refm1"] = xc.estimate_global_motion(mov, 1.0, reference_frame=-1).numpy()
            for s in ("mean_except_current", "middle_frame"):
                fld, pos = xc.estimate_motion_cross_correlation_patches(mov, 1.0, patch_sidelength=32,
                                                                        reference_strategy=s)
                out[f"blob_patches_{s}"] = fld.numpy()
            out["blob_patch_pos"] = pos.numpy()
            f22, f11 = torch.zeros(2, 5, 2, 2), torch.zeros(2, 5, 1, 1)
            for f in range(5):
                f22[0, f], f22[1, f] = 0.1 * f, 0.05 * f
                f11[0, f], f11[1, f] = 0.1 * f, 0.05 * f
            out["blob_correct_cr"] = cm.correct_motion(stat, f22, 1.0).numpy()
            out["blob_correct_bs"] = cm.correct_motion(stat, f22, 1.0, grid_type="bspline").numpy()
            out["blob_correct_rigid"] = cm.correct_motion(stat, f11, 1.0).numpy()
            g11 = f11.clone()
            out["blob_fast"] = cm.correct_motion_fast(stat, g11).numpy()
            out["blob_fast_grid_after"] = g11.numpy()  # Q1: negated in place

            st, dy, dx = drift_stack(8, 256, 256)
            fld = xc.estimate_global_motion(st, 1.0)
            out["drift_global"] = fld.numpy()
            out["drift_corrected_sum"] = cm.correct_motion(st, fld, 1.0).sum(0).numpy()
            pf, pos = xc.estimate_motion_cross_correlation_patches(st, 1.0, patch_sidelength=64)
            out["drift_patch_field"], out["drift_patch_pos"] = pf.numpy(), pos.numpy()
            out["drift_patch_corrected_sum"] = cm.correct_motion(st, pf, 1.0, grid_type="bspline").sum(0).numpy()
            # options: no sub-pixel step / rejection / smoothing, other strategy, window, threshold
            for i, kw in enumerate((
                    {"reference_strategy": "middle_frame"},
                    {"reference_strategy": "middle_frame", "reference_frame": 1},
                    {"reference_strategy": "middle_frame", "reference_frame": -1},
                    {"sub_pixel_refinement": False, "outlier_rejection": False},
                    {"temporal_smoothing": False}, {"smoothing_window_size": 3},
                    {"outlier_threshold": 1.0}, {"b_factor": 1000, "frequency_range": (200, 20)})):
                f_, _ = xc.estimate_motion_cross_correlation_patches(st, 1.3, patch_sidelength=64, **kw)
                out[f"drift_opt{i}"] = f_.numpy()
            # even effective smoothing window: t = 4 < 5
            f_, _ = xc.estimate_motion_cross_correlation_patches(st[:4], 1.0, patch_sidelength=64)
            out["drift_t4"] = f_.numpy()
            f_, _ = xc.estimate_motion_cross_correlation_patches(st[:6], 1.0, patch_sidelength=64,
                                                                 smoothing_window_size=7)
            out["drift_t6_w7"] = f_.numpy()
            # prior fields (Q1 in-place negation, Q9 order): rigid -> correct_motion_fast, full -> bspline
            prior = fld.clone()
            f_, _ = xc.estimate_motion_cross_correlation_patches(st, 1.0, patch_sidelength=64,
                                                                 deformation_field=prior)
            out["drift_prior_rigid"], out["drift_prior_rigid_after"] = f_.numpy(), prior.numpy()
            prior = pf.clone()
            f_, _ = xc.estimate_motion_cross_correlation_patches(st, 1.0, patch_sidelength=64,
                                                                 deformation_field=prior)
            out["drift_prior_full"], out["drift_prior_full_after"] = f_.numpy(), prior.numpy()
            # t = 52: the 50-entry memo evicts (Q3), both strategies
            st52, _, _ = drift_stack(52, 96, 96, seed=99, pad=16)
            for s in ("mean_except_current", "middle_frame"):
                f_, _ = xc.estimate_motion_cross_correlation_patches(st52, 1.0, patch_sidelength=32,
                                                                     reference_strategy=s)
                out[f"t52_{s}"] = f_.numpy()
    finally:
        for (mod, name), v in saved.items():
            setattr(mod, name, v)
    np.savez_compressed(os.path.join(GOLD, "reference_bodies_with_standins.npz"), **out)
    print("wrote reference_bodies_with_standins.npz", len(out), "arrays")


def blob_stack(moving: bool):
    """tests/test_estimate_motion.py:13-33 and tests/test_correct_motion.py:15-32."""
    t, h, w = 5, 64, 64
    yy, xx = torch.meshgrid(torch.arange(h, dtype=torch.float32),
                            torch.arange(w, dtype=torch.float32), indexing="ij")
    img = torch.zeros(t, h, w)
    for f in range(t):
        cy = (h // 2 + (2 * f if moving else 0)) % h
        cx = (w // 2 + (f if moving else 0)) % w
        img[f] = torch.exp(-((yy - cy) ** 2 + (xx - cx) ** 2) / (2 * 10**2))
    return img


def drift_stack(t, h, w, seed=1234, noise=1.0, pad=64):
    """SURVEY.md section 8d synthetic recipe: white-noise texture, integer drift."""
    g = torch.Generator().manual_seed(seed)
    base = torch.randn(h + 2 * pad, w + 2 * pad, generator=g)
    dy = torch.round(torch.linspace(-6, 8, t)).long()
    dx = torch.round(torch.linspace(5, -4, t)).long()
    frames = [
        base[pad - dy[f] : pad - dy[f] + h, pad - dx[f] : pad - dx[f] + w]
        + noise * torch.randn(h, w, generator=g)
        for f in range(t)
    ]
    return torch.stack(frames), dy, dx


def oracle_vectors():
    import oracle

    out = {}
    mov, stat = blob_stack(True), blob_stack(False)
    out["blob_global"] = oracle.estimate_global_motion(mov, 1.0).numpy()
    for s in ("mean_except_current", "middle_frame"):
        fld, pos = oracle.estimate_motion_cross_correlation_patches(
            mov, 1.0, patch_sidelength=32, reference_strategy=s)
        out[f"blob_patches_{s}"] = fld.numpy()
    out["blob_patch_pos"] = pos.numpy()
    f22 = torch.zeros(2, 5, 2, 2)
    f11 = torch.zeros(2, 5, 1, 1)
    for f in range(5):
        f22[0, f], f22[1, f] = 0.1 * f, 0.05 * f
        f11[0, f], f11[1, f] = 0.1 * f, 0.05 * f
    out["blob_correct_cr"] = oracle.correct_motion(stat, f22, 1.0).numpy()
    out["blob_correct_bs"] = oracle.correct_motion(stat, f22, 1.0, grid_type="bspline").numpy()
    out["blob_fast"] = oracle.correct_motion_fast(stat, f11.clone()).numpy()
    np.savez_compressed(os.path.join(GOLD, "oracle_blob.npz"), **out)

    out = {}
    st, dy, dx = drift_stack(8, 256, 256)
    fld = oracle.estimate_global_motion(st, 1.0)
    out["dy"], out["dx"] = dy.numpy(), dx.numpy()
    out["global_field"] = fld.numpy()
    out["corrected_sum"] = oracle.correct_motion(st, fld, 1.0).sum(0).numpy()
    pf, pos = oracle.estimate_motion_cross_correlation_patches(st, 1.0, patch_sidelength=64)
    out["patch_field"], out["patch_pos"] = pf.numpy(), pos.numpy()
    out["patch_corrected_sum"] = oracle.correct_motion(st, pf, 1.0, grid_type="bspline").sum(0).numpy()
    np.savez_compressed(os.path.join(GOLD, "oracle_drift_8x256.npz"), **out)
    print("wrote oracle_*.npz")


def reference_local_motion_vectors():
    """tests/golden/reference_local_helpers.npz: the torch-only pieces of the reference's
    estimate_local_motion path, run from the reference's own modules (inert stubs for the absent
    packages, as above): _compute_loss (estimate_motion_optimizer.py:611-671), the optimiser
    defaults of _setup_optimizer (:517-608), ImagePatchIterator's patches and normalised centres
    in lattice order (patch_utils.py:47-192) and OptimizationTracker (optimization_state.py)."""
    import json

    _reference_modules_behind_inert_stubs()
    emo = importlib.import_module("torch_motion_correction.estimate_motion_optimizer")
    pu = importlib.import_module("torch_motion_correction.patch_utils")
    ost = importlib.import_module("torch_motion_correction.optimization_state")
    pgm = importlib.import_module("torch_motion_correction.patch_grid")
    g = torch.Generator().manual_seed(77)
    out = {}
    b, t, ph, pw = 2, 3, 8, 10
    x = torch.fft.rfftn(torch.randn(b, t, ph, pw, generator=g), dim=(-2, -1))
    y = torch.fft.rfftn(torch.randn(b, t, ph, pw, generator=g), dim=(-2, -1))
    out["loss_x"], out["loss_y"] = torch.view_as_real(x).numpy(), torch.view_as_real(y).numpy()
    for lt in ("mse", "ncc", "cc"):
        out[f"loss_{lt}"] = emo._compute_loss(x, y, ph, pw, loss_type=lt).numpy()
    defaults = {}
    for name in ("adam", "sgd", "rmsprop", "lbfgs"):
        opt = emo._setup_optimizer(name, [torch.zeros(2, requires_grad=True)])
        defaults[name] = {k: (list(v) if isinstance(v, tuple) else v) for k, v in opt.defaults.items()
                          if isinstance(v, (int, float, bool, str, tuple, type(None)))}
        defaults[name]["class"] = type(opt).__name__
    out["optimizer_defaults_json"] = np.frombuffer(json.dumps(defaults, sort_keys=True).encode(), dtype=np.uint8)
    img = torch.randn(3, 40, 52, generator=g)
    pts = pgm.patch_grid_centers(image_shape=(3, 40, 52), patch_shape=(1, 16, 20), patch_step=(1, 8, 10),
                                 distribute_patches=True)
    it = pu.ImagePatchIterator(image=img, patch_size=(16, 20), control_points=pts)
    patches, centers = [], []
    for pb, cb in it.get_iterator(batch_size=4, randomized=False):
        patches.append(pb)
        centers.append(cb)
    out["ipi_image"], out["ipi_points"] = img.numpy(), pts.numpy()
    out["ipi_patches"] = torch.cat(patches, 0).numpy()  # (npatch, t, ph, pw)
    out["ipi_centers"] = torch.cat(centers, 1).numpy()  # (t, npatch, 3) normalised
    out["ipi_batch_sizes"] = np.asarray([p.shape[0] for p in patches])
    # the reference's shuffled patch order (random.shuffle on the global state, patch_utils.py:163-164):
    # two consecutive passes after random.seed(2024), as normalised centres of frame 0
    import random

    random.seed(2024)
    for k in range(2):
        cs = [cb for _, cb in it.get_iterator(batch_size=8, randomized=True)]
        out[f"ipi_shuffled_centers_{k}"] = torch.cat(cs, 1)[0].numpy()  # (npatch, 3)
    tr = ost.OptimizationTracker(sample_every_n_steps=3, total_steps=8)
    for step in range(8):
        if tr.sample_this_step(step):
            tr.add_checkpoint(torch.full((2, 1, 1, 2), float(step)), 0.5 * step, step)
    out["tracker_json"] = np.frombuffer(json.dumps(tr.as_dict(), sort_keys=True).encode(), dtype=np.uint8)
    np.savez_compressed(os.path.join(GOLD, "reference_local_helpers.npz"), **out)
    print("wrote reference_local_helpers.npz", len(out), "arrays")


if __name__ == "__main__":
    os.makedirs(GOLD, exist_ok=True)
    if os.path.isdir(REF_SRC):
        reference_vectors()
        reference_helper_vectors()
        reference_local_motion_vectors()
        reference_body_vectors()
    else:
        print("reference not present: patch_grid_reference.npz not regenerated")
    oracle_vectors()
