"""CPU tests of the oracle: pinned against the reference-produced goldens, against the
assertions of the reference's own test-suite, and against its own committed outputs."""

import os

import numpy as np
import pytest
import torch

import oracle
from oracle import patch_grid as opg
from oracle import thirdparty_semantics as tp
from conftest import blob_stack, drift_stack, ramp_field


# ---------------------------------------------------------------- pinned by the reference


def test_patch_centres_match_reference(golden):
    g = golden("patch_grid_reference.npz")
    n = 0
    for key in g.files:
        if key.startswith("centers_"):
            _, dim, p = key.split("_")
            got = opg.centers_1d(int(dim), int(p), int(p) // 2, True).numpy()
            assert np.array_equal(got, g[key]), key
            n += 1
    assert n >= 10


def test_lazy_gather_matches_reference(golden):
    g = golden("patch_grid_reference.npz")
    img = torch.from_numpy(g["gather_img"])
    lazy, centers = opg.patch_grid_lazy(img, (1, 16, 16), (1, 8, 8), True)
    assert np.array_equal(centers.numpy(), g["gather_centers"])
    assert np.array_equal(lazy[1].numpy(), g["gather_frame1"])
    assert lazy[1] is lazy[1]  # memoised: the same tensor object is handed out again


@pytest.mark.parametrize("t", [5, 8, 40, 51, 60])
@pytest.mark.parametrize("strategy", ["middle_frame", "mean_except_current"])
def test_memo_aliasing_schedule_matches_reference(golden, t, strategy):
    """Replay xc.py:297-346's access pattern on the oracle's LazyPatches with a scalar
    mask of 2.0: the exponents must equal what the reference's own class produced."""
    g = golden("patch_grid_reference.npz")
    imgs = torch.ones(t, 8, 8)
    lz, _ = opg.patch_grid_lazy(imgs, (1, 4, 4), (1, 2, 2), True)
    ref = t // 2
    table = np.full((t, t), -1, dtype=np.int64)
    cur = np.full((t,), -1, dtype=np.int64)
    for f in range(t):
        r = None
        if strategy == "middle_frame":
            if f == ref:
                continue
            r = lz[ref][0, :, :, 0]
            table[f, ref] = int(torch.log2(r.flatten()[0]))
        else:
            for o in range(t):
                if o != f:
                    table[f, o] = int(torch.log2(lz[o].flatten()[0]))
        c = lz[f][0, :, :, 0]
        cur[f] = int(torch.log2(c.flatten()[0]))
        if r is not None:
            r *= 2.0
        c *= 2.0
    assert np.array_equal(table, g[f"exp_{strategy}_{t}"])
    assert np.array_equal(cur, g[f"cur_{strategy}_{t}"])


# ------------------------------------------- the reference's own test assertions


def test_ref_suite_global_shapes():
    mov = blob_stack(True)
    for kw in ({}, {"reference_frame": 0}, {"b_factor": 1000}, {"frequency_range": (200, 20)}):
        f = oracle.estimate_global_motion(mov, 1.0, **kw)
        assert isinstance(f, torch.Tensor) and f.shape == (2, 5, 1, 1)


@pytest.mark.parametrize("kw", [
    {}, {"reference_strategy": "middle_frame"},
    {"sub_pixel_refinement": False, "outlier_rejection": False},
    {"smoothing_window_size": 3}, {"outlier_threshold": 2.0}, {"temporal_smoothing": False},
    {"outlier_rejection": False},
])
def test_ref_suite_patches_shapes(kw):
    mov = blob_stack(True)
    f, pos = oracle.estimate_motion_cross_correlation_patches(mov, 1.0, patch_sidelength=32, **kw)
    assert f.ndim == 4 and f.shape[0] == 2 and f.shape[1] == 5
    assert pos.ndim == 4 and pos.shape[0] == 5 and pos.dtype == torch.int64


def test_q12_integer_peaks_with_outlier_rejection_raise():
    """Reference accident: without sub-pixel refinement the shifts are int64 and
    torch.std (xc.py:577) rejects them -- the combination raises in the reference."""
    with pytest.raises(RuntimeError, match="floating point"):
        oracle.estimate_motion_cross_correlation_patches(blob_stack(True), 1.0, patch_sidelength=32,
                                                         sub_pixel_refinement=False)


def test_ref_suite_patches_with_initial_field():
    mov = blob_stack(True)
    f, _ = oracle.estimate_motion_cross_correlation_patches(
        mov, 1.0, patch_sidelength=32, deformation_field=torch.zeros(2, 5, 1, 1))
    assert f.shape[0] == 2 and torch.isfinite(f).all()


def test_ref_suite_unknown_strategy():
    with pytest.raises(ValueError, match="Unknown reference_strategy"):
        oracle.estimate_motion_cross_correlation_patches(blob_stack(True), 1.0, patch_sidelength=32,
                                                         reference_strategy="nope")


@pytest.mark.parametrize("grid_type", ["catmull_rom", "bspline"])
def test_ref_suite_correct_motion(grid_type):
    stat = blob_stack(False)
    out = oracle.correct_motion(stat, ramp_field(), 1.0, grid_type=grid_type)
    assert out.shape == stat.shape and not out.requires_grad
    zero = oracle.correct_motion(stat, torch.zeros(2, 5, 2, 2), 1.0, grid_type=grid_type)
    assert torch.allclose(zero, stat, atol=0.1)  # tests/test_correct_motion.py:132-145
    assert torch.allclose(zero, stat, atol=1e-5)  # and much tighter in fact


def test_ref_suite_correct_motion_fast():
    stat = blob_stack(False)
    out = oracle.correct_motion_fast(stat, ramp_field(g=1))
    assert out.shape == stat.shape
    zero = oracle.correct_motion_fast(stat, torch.zeros(2, 5, 1, 1))
    assert torch.allclose(zero, stat, atol=1e-5)  # tests/test_correct_motion.py:188-199
    with pytest.raises(ValueError, match="Expected single patch deformation field"):
        oracle.correct_motion_fast(stat, ramp_field())


def test_ref_suite_global_then_correct_is_finite():
    mov = blob_stack(True)
    fld = oracle.estimate_global_motion(mov, 1.0)
    assert torch.isfinite(oracle.correct_motion(mov, fld, 1.0)).all()  # size-1 spline axes
    assert torch.isfinite(oracle.correct_motion_fast(mov, fld.clone())).all()


def test_q1_fast_negates_callers_field():
    f = ramp_field(g=1)
    before = f.clone()
    oracle.correct_motion_fast(blob_stack(False), f)
    assert torch.equal(f, -before)


# --------------------------------------------------------------- own committed outputs


def test_oracle_blob_goldens(golden):
    g = golden("oracle_blob.npz")
    mov, stat = blob_stack(True), blob_stack(False)
    assert np.array_equal(oracle.estimate_global_motion(mov, 1.0).numpy(), g["blob_global"])
    for s in ("mean_except_current", "middle_frame"):
        f, pos = oracle.estimate_motion_cross_correlation_patches(mov, 1.0, patch_sidelength=32,
                                                                  reference_strategy=s)
        assert np.allclose(f.numpy(), g[f"blob_patches_{s}"], atol=2e-5)
        assert np.array_equal(pos.numpy(), g["blob_patch_pos"])
    assert np.allclose(oracle.correct_motion(stat, ramp_field(), 1.0).numpy(), g["blob_correct_cr"],
                       atol=1e-6)
    assert np.allclose(oracle.correct_motion(stat, ramp_field(), 1.0, grid_type="bspline").numpy(),
                       g["blob_correct_bs"], atol=1e-6)
    assert np.allclose(oracle.correct_motion_fast(stat, ramp_field(g=1)).numpy(), g["blob_fast"],
                       atol=1e-6)


def test_oracle_recovers_known_integer_drift(golden):
    """Known-answer test: the synthetic stack's drift is known exactly."""
    st, dy, dx = drift_stack(8, 256, 256)
    f = oracle.estimate_global_motion(st, 1.0)
    assert torch.equal(f[0, :, 0, 0], (dy - dy[4]).float())
    assert torch.equal(f[1, :, 0, 0], (dx - dx[4]).float())
    g = golden("oracle_drift_8x256.npz")
    assert np.array_equal(f.numpy(), g["global_field"])
    s = oracle.correct_motion(st, f, 1.0).sum(0)
    assert np.allclose(s.numpy(), g["corrected_sum"], atol=1e-4)


# ------------------------------------------------------------------ semantics spot checks


def test_spline_properties():
    g = torch.Generator().manual_seed(0)
    data = torch.randn(2, 4, 3, 5, generator=g)
    knots = torch.stack(torch.meshgrid(torch.linspace(0, 1, 4), torch.linspace(0, 1, 3),
                                       torch.linspace(0, 1, 5), indexing="ij"), -1)
    cr = tp.cubic_spline_grid_3d(data, knots, "catmull_rom")
    assert torch.allclose(cr.permute(3, 0, 1, 2), data, atol=1e-5)  # Catmull-Rom interpolates
    const = torch.full((2, 4, 3, 5), 3.25)
    pts = torch.rand(50, 3, generator=g)
    for kind in ("catmull_rom", "bspline"):
        assert torch.allclose(tp.cubic_spline_grid_3d(const, pts, kind), torch.full((50, 2), 3.25),
                              atol=1e-5)
    one = torch.randn(2, 5, 1, 1, generator=g)  # single-sample axes are constant
    v = tp.cubic_spline_grid_3d(one, torch.tensor([[0.25, 0.3, 0.9], [0.25, 0.7, 0.1]]), "catmull_rom")
    assert torch.allclose(v[0], v[1], atol=1e-6) and torch.allclose(v[0], one[:, 1, 0, 0], atol=1e-5)


def test_mask_and_filters():
    m = tp.circle(16, (64, 64), smoothing_radius=8)
    assert m[32, 32] == 1 and m[32, 47] == 1 and m[32, 48] < 1 and m[0, 0] == 0
    assert float(m[32, 56]) == pytest.approx(0.0, abs=1e-6) and m.min() >= -1e-7 and m.max() == 1
    band = oracle.prepare_bandpass_filter((300, 10), (64, 64), 1.0)
    assert band[0, 0] == 0 and set(band.unique().tolist()) <= {0.0, 1.0}
    f = tp.fftfreq_grid((64, 64), rfft=True, norm=True)
    assert torch.equal(band, ((f > 1 / 300) & (f <= 0.1)).float())
    env = tp.b_envelope(500, (64, 64), 1.0)
    assert torch.allclose(env, torch.exp(-500 * f**2 / 4))


def test_sub_pixel_rules():
    """Q4: no refinement when the peak touches a border; Q5: axis skipped when symmetric."""
    from oracle.motion import _sub_pixel

    cc = torch.zeros(1, 8, 8)
    cc[0, 0, 3] = 1.0
    cc[0, 1, 3] = 0.5
    py, px = _sub_pixel(cc, torch.tensor([3]), 8, 8)
    assert py.item() == 0 and px.item() == 3  # border: untouched
    cc = torch.zeros(1, 8, 8)
    cc[0, 3, 3], cc[0, 2, 3], cc[0, 4, 3] = 1.0, 0.2, 0.6
    py, px = _sub_pixel(cc, torch.tensor([3 * 8 + 3]), 8, 8)
    assert px.item() == 3  # symmetric in x: skipped
    assert py.item() == pytest.approx(3 + 0.5 * (0.2 - 0.6) / (0.2 - 2.0 + 0.6))


def test_outlier_rejection_rules():
    from oracle.motion import _reject_outliers

    sy = torch.tensor([[0.0, 0.1, 0.0], [0.1, 9.0, 0.0], [0.1, 0.0, 0.1]])
    sx = torch.zeros(3, 3)
    ry, rx = _reject_outliers(sy, sx, 2.0)
    assert ry[1, 1].item() == pytest.approx(sy.flatten()[[0, 1, 2, 3, 5, 6, 7, 8]].mean().item())
    same = torch.full((2, 2), 1.5)
    ry, rx = _reject_outliers(same, same, 3.0)  # zero spread: nothing rejected
    assert torch.equal(ry, same)


def test_dose_weighted_sum_properties():
    """Exposure-filtered frame sum of the reference's example pipeline
    (examples/ttMotion.py:331-351); third-party filter semantics: parity unpinned, so only
    the properties that hold for any exp(-0.5 N/N_c) filter with power restoration."""
    g = torch.Generator().manual_seed(3)
    m = torch.randn(6, 48, 64, generator=g) + 2.0
    # zero dose: every weight is 1 -> plain sum / sqrt(t)
    z = oracle.dose_weighted_sum(m, 1.0, 0.0)
    assert torch.allclose(z, m.sum(0) / 6**0.5, atol=1e-4)
    # the DC term keeps weight ~1 for any dose: mean(sum) = sum of means / sqrt(t)
    d = oracle.dose_weighted_sum(m, 1.3, 2.0, pre_exposure=1.0)
    assert float(d.mean()) == pytest.approx(float(m.sum(0).mean()) / 6**0.5, rel=1e-4)
    # high frequencies are damped more than low ones, later frames more than early ones
    f = tp.fftfreq_grid((48, 64), rfft=True, norm=True)
    w = tp.dose_weight_movie(torch.ones(6, 48, 33, dtype=torch.complex64), (48, 64), 1.3, 1.0, 2.0).real
    assert float(w[-1][f > 0.4].mean()) < float(w[0][f > 0.4].mean())
    assert torch.allclose((w**2).sum(0), torch.ones(48, 33), atol=1e-5)  # power restored
    # linear in the movie
    assert torch.allclose(oracle.dose_weighted_sum(3 * m, 1.3, 2.0), 3 * oracle.dose_weighted_sum(m, 1.3, 2.0),
                          atol=1e-4)


# ------------------------------------------------------------------ pinned by the reference's own helpers


@pytest.fixture(scope="module")
def refh():
    """tests/golden/reference_helpers.npz: produced by the REFERENCE's own functions (imported
    from /root/reference with the absent third-party packages bound to stubs that raise;
    oracle/make_goldens.py::reference_helper_vectors)."""
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "reference_helpers.npz")
    return {k: torch.from_numpy(v) if v.ndim else v for k, v in np.load(path, allow_pickle=False).items()}


def test_oracle_normalize_and_field_helpers_match_the_reference(refh):
    from oracle import motion

    assert torch.equal(motion.normalize_image(refh["norm_in"]), refh["norm_out"])  # utils.py:49-84
    assert torch.equal(tp.array_to_grid_sample(refh["a2g_in"], (40, 56)), refh["a2g_out"])  # utils.py:9-30
    got = motion.image_shifts_to_deformation_field(refh["s2f_in"], 1.7)  # dfu.py:129-162
    assert got.shape == refh["s2f_out"].shape and torch.equal(got, refh["s2f_out"])


def test_oracle_sub_pixel_refinement_matches_the_reference(refh):
    """estimate_motion_xc.py:414-483 incl. the border rule (Q4) and the equal-samples rule (Q5)."""
    from oracle import motion

    cc, peaks = refh["sp_cc"], refh["sp_peaks"]
    py, px = motion._sub_pixel(cc.view(cc.shape[0], 12, 16), peaks, 12, 16)
    assert torch.equal(py, refh["sp_y"]) and torch.equal(px, refh["sp_x"])
    assert float(py[1]) == 0.0 and float(px[2]) == 15.0  # border peaks stay integer


def test_oracle_outlier_rejection_matches_the_reference(refh):
    """estimate_motion_xc.py:538-627: one outlier, none, a constant field, everything rejected."""
    from oracle import motion

    for i in range(4):
        oy, ox = motion._reject_outliers(refh[f"or{i}_y"], refh[f"or{i}_x"], float(refh[f"or{i}_thr"]))
        assert torch.equal(oy, refh[f"or{i}_oy"]) and torch.equal(ox, refh[f"or{i}_ox"]), i
    assert not torch.equal(refh["or0_oy"], refh["or0_y"])  # the planted outlier was replaced


def test_oracle_temporal_smoothing_matches_the_reference(refh):
    """estimate_motion_xc.py:486-535: odd, even (-> next odd), minimal, too long (-> t) and no-op windows."""
    from oracle import motion

    for wdw in (5, 4, 3, 15, 2):
        assert torch.equal(motion._smooth_time(refh["ts_in"].clone(), wdw), refh[f"ts_out_{wdw}"]), wdw


def test_oracle_even_effective_smoothing_window_matches_the_reference(refh):
    """min(window | 1, t) is EVEN for an even t below it (xc.py:506-512): scipy accepts it."""
    from oracle import motion

    n = 0
    for key in refh:
        if key.startswith("tse_in_"):
            _, _, tt, wdw = key.split("_")
            got = motion._smooth_time(refh[key].clone(), int(wdw))
            assert torch.equal(got, refh[f"tse_out_{tt}_{wdw}"]), key
            n += 1
    assert n == 5


def test_oracle_pixel_shifts_on_a_caller_grid_match_the_reference(refh):
    """correct_motion.py:167-168: `pixel_grid` is used, not assumed to be the identity grid."""
    from oracle import motion

    got = motion.get_pixel_shifts(torch.zeros(37, 53), 1.3, refh["gps_lattice"], refh["gps_sub_grid"])
    assert torch.equal(got, refh["gps_sub_out"])


# ---------------------------------- the reference's own function bodies (stand-ins below them)


@pytest.fixture(scope="module")
def refb():
    """tests/golden/reference_bodies_with_standins.npz: the reference's estimate_global_motion,
    estimate_motion_cross_correlation_patches, correct_motion and correct_motion_fast EXECUTED from
    /root/reference with the absent third-party names bound to oracle.thirdparty_semantics
    (oracle/make_goldens.py:reference_body_vectors).  Pins the restatement of the reference's own
    control flow; does NOT pin third-party semantics (parity stays unpinned there)."""
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden",
                        "reference_bodies_with_standins.npz")
    return {k: torch.from_numpy(v) for k, v in np.load(path).items()}


def test_oracle_equals_reference_bodies_blob(refb):
    mov, stat = blob_stack(True), blob_stack(False)
    assert torch.equal(oracle.estimate_global_motion(mov, 1.0), refb["blob_global"])
    assert torch.equal(oracle.estimate_global_motion(mov, 1.0, reference_frame=0), refb["blob_global_ref0"])
    # a negative reference frame indexes from the end but is never "the current frame": not skipped
    assert torch.equal(oracle.estimate_global_motion(mov, 1.0, reference_frame=-1), refb["blob_global_refm1"])
    with pytest.raises(IndexError):
        oracle.estimate_global_motion(mov, 1.0, reference_frame=5)
    for s in ("mean_except_current", "middle_frame"):
        fld, pos = oracle.estimate_motion_cross_correlation_patches(mov, 1.0, patch_sidelength=32,
                                                                    reference_strategy=s)
        assert torch.equal(fld, refb[f"blob_patches_{s}"]), s
        assert torch.equal(pos, refb["blob_patch_pos"])
    f22, f11 = ramp_field(5, 2), ramp_field(5, 1)
    assert torch.equal(oracle.correct_motion(stat, f22, 1.0), refb["blob_correct_cr"])
    assert torch.equal(oracle.correct_motion(stat, f22, 1.0, grid_type="bspline"), refb["blob_correct_bs"])
    assert torch.equal(oracle.correct_motion(stat, f11, 1.0), refb["blob_correct_rigid"])
    g11 = f11.clone()
    assert torch.equal(oracle.correct_motion_fast(stat, g11), refb["blob_fast"])
    assert torch.equal(g11, refb["blob_fast_grid_after"]) and torch.equal(g11, -f11)  # Q1


def test_oracle_equals_reference_bodies_drift(refb):
    st, _, _ = drift_stack(8, 256, 256)
    fld = oracle.estimate_global_motion(st, 1.0)
    assert torch.equal(fld, refb["drift_global"])
    assert torch.equal(oracle.correct_motion(st, fld, 1.0).sum(0), refb["drift_corrected_sum"])
    pf, pos = oracle.estimate_motion_cross_correlation_patches(st, 1.0, patch_sidelength=64)
    assert torch.equal(pf, refb["drift_patch_field"]) and torch.equal(pos, refb["drift_patch_pos"])
    assert torch.equal(oracle.correct_motion(st, pf, 1.0, grid_type="bspline").sum(0),
                       refb["drift_patch_corrected_sum"])
    for i, kw in enumerate((
            {"reference_strategy": "middle_frame"},
            {"reference_strategy": "middle_frame", "reference_frame": 1},
            {"reference_strategy": "middle_frame", "reference_frame": -1},
            {"sub_pixel_refinement": False, "outlier_rejection": False},
            {"temporal_smoothing": False}, {"smoothing_window_size": 3},
            {"outlier_threshold": 1.0}, {"b_factor": 1000, "frequency_range": (200, 20)})):
        f_, _ = oracle.estimate_motion_cross_correlation_patches(st, 1.3, patch_sidelength=64, **kw)
        assert torch.equal(f_, refb[f"drift_opt{i}"]), kw
    f_, _ = oracle.estimate_motion_cross_correlation_patches(st[:4], 1.0, patch_sidelength=64)
    assert torch.equal(f_, refb["drift_t4"])  # even effective smoothing window (t = 4 < 5)
    f_, _ = oracle.estimate_motion_cross_correlation_patches(st[:6], 1.0, patch_sidelength=64,
                                                             smoothing_window_size=7)
    assert torch.equal(f_, refb["drift_t6_w7"])
    prior = fld.clone()
    f_, _ = oracle.estimate_motion_cross_correlation_patches(st, 1.0, patch_sidelength=64,
                                                             deformation_field=prior)
    assert torch.equal(f_, refb["drift_prior_rigid"]) and torch.equal(prior, refb["drift_prior_rigid_after"])
    prior = pf.clone()
    f_, _ = oracle.estimate_motion_cross_correlation_patches(st, 1.0, patch_sidelength=64,
                                                             deformation_field=prior)
    assert torch.equal(f_, refb["drift_prior_full"]) and torch.equal(prior, refb["drift_prior_full_after"])


@pytest.mark.parametrize("strategy", ["mean_except_current", "middle_frame"])
def test_oracle_equals_reference_bodies_t52(refb, strategy):
    """t = 52 > 50: the reference's memo evicts half its entries (Q3); both strategies."""
    st52, _, _ = drift_stack(52, 96, 96, seed=99, pad=16)
    f_, _ = oracle.estimate_motion_cross_correlation_patches(st52, 1.0, patch_sidelength=32,
                                                             reference_strategy=strategy)
    assert torch.equal(f_, refb[f"t52_{strategy}"])


def test_oracle_pixel_shifts_match_the_reference(refh):
    """correct_motion.py:132-185 (bicubic / reflection / align_corners upsample of the lattice)."""
    from oracle import motion

    yy, xx = torch.meshgrid(torch.arange(37, dtype=torch.float32), torch.arange(53, dtype=torch.float32),
                            indexing="ij")
    got = motion.get_pixel_shifts(torch.zeros(37, 53), 1.3, refh["gps_lattice"], torch.stack([yy, xx], dim=-1))
    assert torch.equal(got, refh["gps_out"])


# ------------------------------------------------------------------ estimate_local_motion


@pytest.fixture(scope="module")
def refl():
    """tests/golden/reference_local_helpers.npz: produced by the REFERENCE's own torch-only
    functions of the estimate_local_motion path (oracle/make_goldens.py:reference_local_motion_vectors)."""
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "reference_local_helpers.npz")
    return np.load(path)


def test_oracle_local_losses_match_the_reference(refl):
    from oracle import local_motion as olm

    x = torch.view_as_complex(torch.from_numpy(refl["loss_x"]))
    y = torch.view_as_complex(torch.from_numpy(refl["loss_y"]))
    for lt in ("mse", "ncc", "cc"):
        got = olm.compute_loss(x, y, 8, 10, lt)
        assert torch.equal(got, torch.from_numpy(refl[f"loss_{lt}"])), lt


def test_oracle_and_product_optimiser_defaults_match_the_reference(refl):
    import json

    from oracle import local_motion as olm
    from torch_motion_correction_amd import local_motion as plm

    want = json.loads(bytes(refl["optimizer_defaults_json"]).decode())
    for mod in (olm, plm):
        for name, ref in want.items():
            opt = mod.setup_optimizer(name, [torch.zeros(2, requires_grad=True)])
            assert type(opt).__name__ == ref["class"]
            for k, v in ref.items():
                if k == "class":
                    continue
                have = opt.defaults[k]
                assert (list(have) if isinstance(have, tuple) else have) == v, (name, k, have, v)
        with pytest.raises(ValueError, match="Unsupported optimizer"):
            mod.setup_optimizer("adagrad", [torch.zeros(2, requires_grad=True)])


def test_oracle_patches_and_centres_match_the_reference_iterator(refl):
    """ImagePatchIterator in lattice order (patch_utils.py): same pixels, same normalised centres."""
    from oracle import patch_grid as pg

    img = torch.from_numpy(refl["ipi_image"])
    t, h, w = img.shape
    ph, pw = 16, 20
    centers = pg.centers_3d((t, h, w), (1, ph, pw), (1, ph // 2, pw // 2), True)
    assert torch.equal(centers, torch.from_numpy(refl["ipi_points"]))
    cn = centers.clone().float()
    cn[..., 0] /= float(t - 1)
    cn[..., 1] /= float(h - 1)
    cn[..., 2] /= float(w - 1)
    assert torch.equal(cn.reshape(t, -1, 3), torch.from_numpy(refl["ipi_centers"]))
    pts = centers[0].reshape(-1, 3)
    patches = torch.stack([img[:, int(c[1]) - ph // 2:int(c[1]) - ph // 2 + ph,
                               int(c[2]) - pw // 2:int(c[2]) - pw // 2 + pw] for c in pts])
    assert torch.equal(patches, torch.from_numpy(refl["ipi_patches"]))
    n = patches.shape[0]
    assert list(refl["ipi_batch_sizes"]) == [min(4, n - a) for a in range(0, n, 4)]
    # the shuffled order of two consecutive passes after random.seed(2024): one random.shuffle of
    # range(npatch) per pass reproduces the reference's (patch_utils.py:161-166)
    import random

    random.seed(2024)
    lattice_order = torch.from_numpy(refl["ipi_centers"])[0]
    for k in range(2):
        order = list(range(n))
        random.shuffle(order)
        assert torch.equal(lattice_order[order], torch.from_numpy(refl[f"ipi_shuffled_centers_{k}"])), k


def test_trackers_match_the_reference(refl):
    import json

    from oracle import local_motion as olm
    from torch_motion_correction_amd import optimization_state as pos

    want = json.loads(bytes(refl["tracker_json"]).decode())
    for cls in (olm.OptimizationTracker, pos.OptimizationTracker):
        tr = cls(sample_every_n_steps=3, total_steps=8)
        for step in range(8):
            if tr.sample_this_step(step):
                tr.add_checkpoint(torch.full((2, 1, 1, 2), float(step)), 0.5 * step, step)
        assert json.loads(json.dumps(tr.as_dict(), sort_keys=True)) == want


def test_oracle_local_motion_recovers_a_small_drift():
    g = torch.Generator().manual_seed(21)
    base = torch.randn(96 + 16, 96 + 16, generator=g)
    dy, dx = [-2, -1, 0, 1, 2], [1, 1, 0, -1, -1]
    st = torch.stack([base[8 - dy[f]:8 - dy[f] + 96, 8 - dx[f]:8 - dx[f] + 96]
                      + 0.5 * torch.randn(96, 96, generator=g) for f in range(5)])
    f, tr = oracle.estimate_local_motion(st, 1.0, (64, 64), (5, 1, 1), None, n_iterations=120,
                                         optimizer_kwargs={"lr": 0.05}, return_trajectory=True,
                                         trajectory_kwargs={"sample_every_n_steps": 40})
    mean = float(np.mean(dy + dx))
    assert float((f[0, :, 0, 0] - (torch.tensor(dy, dtype=torch.float32) - mean)).abs().max()) < 0.4
    assert float((f[1, :, 0, 0] - (torch.tensor(dx, dtype=torch.float32) - mean)).abs().max()) < 0.4
    assert [c.step for c in tr.checkpoints] == [0, 40, 80, 119]
    assert tr.checkpoints[-1].loss < tr.checkpoints[0].loss
    with pytest.raises(ValueError, match="Invalid grid type"):
        oracle.estimate_local_motion(st, 1.0, (64, 64), (5, 1, 1), grid_type="linear", n_iterations=1)
