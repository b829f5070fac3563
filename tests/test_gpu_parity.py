"""GPU parity tests: the HIP path (Python mirror -> ctypes -> libmcorr C ABI) against
the CPU oracle on the same seeded inputs, against the committed golden fixtures, and --
at BASELINE.json's full 40 x 4096 x 4096 size -- through size-independent properties.

Tolerances (north star: 1e-4 relative on float32):
  * shift vectors from the integer-peak search: EXACT equality;
  * sub-pixel patch fields: 1e-4 px absolute (values are O(1) px);
  * corrected frames / sums: max |a-b| <= 1e-4 * max|b|, evaluated on all pixels
    except "knife-edge" pixels -- pixels whose sampling coordinate lies within 1e-3 px
    of the frame border [0, n-1], where the reference's zero-outside rule is a
    discontinuity and a 1-ulp difference in the interpolated shift flips the result
    (the reference is not reproducible against itself there across CPU ISAs, DESIGN.md
    section 6).  The excluded fraction is asserted to be small.
"""

import os

import numpy as np
import pytest
import torch

import oracle
from oracle import thirdparty_semantics as tp
from conftest import blob_stack, drift_stack, ramp_field

pytestmark = pytest.mark.gpu

REL = 1e-4


@pytest.fixture(scope="module")
def mc():
    import torch_motion_correction_amd as m

    return m


def rel_err(a, b):
    a, b = a.detach().cpu().double(), b.detach().cpu().double()
    return float((a - b).abs().max() / max(float(b.abs().max()), 1e-30))


def knife_edge_mask(stack, field, pixel_spacing, grid_type, eps=1e-3):
    """(t,h,w) bool: oracle sampling coordinate within eps of the frame border."""
    t, h, w = stack.shape
    _, _, gh, gw = field.shape
    grid = tp.coordinate_grid((h, w))
    out = torch.zeros(t, h, w, dtype=torch.bool)
    for i, ft in enumerate(torch.linspace(0, 1, steps=t)):
        lat = oracle.evaluate_deformation_field_at_t(field, ft, (10 * gh, 10 * gw), grid_type)
        c = grid + oracle.get_pixel_shifts(stack[i], pixel_spacing, lat, grid)
        near = lambda v, n: (v.abs() < eps) | ((v - (n - 1)).abs() < eps)
        out[i] = near(c[..., 0], h) | near(c[..., 1], w)
    return out


def assert_frames_close(got, ref, knife, max_excluded=0.02):
    got, ref = got.detach().cpu(), ref.detach().cpu()
    frac = float(knife.float().mean())
    assert frac <= max_excluded, f"too many knife-edge pixels excluded: {frac}"
    d = (got - ref).abs()
    d[knife] = 0
    assert float(d.max()) <= REL * float(ref.abs().max()), float(d.max() / ref.abs().max())


def assert_frames_close_large(got, ref, knife, max_excluded=0.02, max_flipped=5e-3):
    """Frames larger than 4096 px with a high-gradient texture: the reference computes the sampling
    coordinate `pixel + shift` in fp32, whose spacing is 2^-11 px = 4.9e-4 px for pixel indices >=
    4096.  A last-bit difference in the interpolated shift (FMA contraction in the bicubic upsample;
    ATen's own AVX2 and AVX-512 kernels differ there too) flips that rounding for a small share of
    the pixels and moves the sample by one coordinate ulp, i.e. the value by ulp x |gradient| --
    above 1e-4 of the range on white-noise-like data.  So: (a) all but `max_flipped` of the pixels
    within REL; (b) EVERY pixel within REL + one coordinate ulp per axis x the largest neighbour
    difference in its 4 x 4 footprint (x 1.5: bicubic overshoot)."""
    import torch.nn.functional as F

    got, ref = got.detach().cpu(), ref.detach().cpu()
    frac = float(knife.float().mean())
    assert frac <= max_excluded, f"too many knife-edge pixels excluded: {frac}"
    h, w = ref.shape[-2:]
    ulp = 2.0 ** (int(np.ceil(np.log2(max(h, w)))) - 1 - 23)  # spacing of fp32 just below max(h, w)
    tol = REL * float(ref.abs().max())
    d = (got - ref).abs()
    d[knife] = 0
    flipped = float((d > tol).float().mean())
    assert flipped <= max_flipped, f"{flipped:.2e} of the pixels beyond {REL}"
    gy = (ref[..., 1:, :] - ref[..., :-1, :]).abs()
    gx = (ref[..., :, 1:] - ref[..., :, :-1]).abs()
    g = torch.maximum(F.pad(gy, (0, 0, 0, 1)), F.pad(gx, (0, 1, 0, 0)))
    g = F.max_pool2d(g.reshape(-1, 1, h, w), kernel_size=5, stride=1, padding=2).reshape(ref.shape)
    bound = tol + 2 * 1.5 * ulp * g
    assert bool((d <= bound).all()), float((d - bound).max())
    return flipped


# ------------------------------------------------------------------ plan constants


@pytest.mark.parametrize("shape,r,s", [((32, 32), 8, 4), ((64, 64), 16, 8), ((256, 256), 64, 32),
                                       ((1024, 1024), 256, 128), ((64, 128), 16, 8), ((64, 64), 16, 0)])
def test_circle_mask(dev, shape, r, s):
    from torch_motion_correction_amd import plan

    got = plan.circle_mask(shape[0], shape[1], float(r), float(s), dev).cpu()
    ref = tp.circle(float(r), shape, smoothing_radius=float(s))
    assert float((got - ref).abs().max()) <= 2e-7


@pytest.mark.parametrize("n,ps,fr,B", [(64, 1.0, (300, 10), 500), (256, 1.0, (300, 10), 500),
                                       (512, 0.83, (300, 10), 500), (128, 1.5, (200, 20), 1000)])
def test_filter_table(dev, n, ps, fr, B):
    from torch_motion_correction_amd import plan

    pl = plan.get_xc_plan(n, n, ps, float(B), fr, dev)
    g = pl.geom
    full = oracle.prepare_bandpass_filter(fr, (n, n), ps) * tp.b_envelope(B, (n, n), ps)
    rows = list(range(g.kyp)) + list(range(n - g.kyn, n))
    assert float((pl.filt.cpu() - full[rows][:, : g.nkx].T).abs().max()) <= 3e-7


def test_central_box_stats(dev):
    from torch_motion_correction_amd import engine

    g = torch.Generator().manual_seed(0)
    img = torch.randn(5, 64, 96, generator=g) * 3 + 7
    s = engine.central_box_stats(img.to(dev)).cpu()
    std, mean = torch.std_mean(img[:, 16:48, 24:72])
    assert float(s[0]) == pytest.approx(float(mean), rel=1e-6)
    assert float(s[2]) == pytest.approx(float(std), rel=1e-6)
    n = engine.normalize(img.to(dev), engine.central_box_stats(img.to(dev))).cpu()
    assert rel_err(n, oracle.normalize_image(img)) <= 1e-6


@pytest.mark.parametrize("t,n", [(3, 64), (2, 256)])
def test_pruned_spectrum_equals_full_spectrum_on_kept_bins(dev, t, n):
    """K1+K2 against torch.fft of the oracle's normalised, masked, filtered frames."""
    from torch_motion_correction_amd import engine, plan

    g = torch.Generator().manual_seed(1)
    img = torch.randn(t, n, n, generator=g)
    pl = plan.get_xc_plan(n, n, 1.0, 500.0, (300, 10), dev)
    gm = pl.geom
    d = img.to(dev)
    off = torch.arange(t, device=dev, dtype=torch.int64) * (n * n)
    S = torch.view_as_complex(engine._forward_spectra(d, off, n, None, pl, engine.central_box_stats(d)).cpu())
    spec = (torch.fft.rfftn(oracle.normalize_image(img) * tp.circle(n / 4, (n, n), smoothing_radius=n / 8),
                            dim=(-2, -1)) * oracle.prepare_bandpass_filter((300, 10), (n, n), 1.0)
            * tp.b_envelope(500, (n, n), 1.0))
    rows = list(range(gm.kyp)) + list(range(n - gm.kyn, n))
    sub = spec[:, rows][:, :, : gm.nkx].transpose(1, 2)
    assert float((S - sub).abs().max() / sub.abs().max()) <= 2e-6


# ------------------------------------------------------------------ a1: global estimate


def test_global_blob_fixture(mc, dev, golden):
    mov = blob_stack(True)
    f = mc.estimate_global_motion(mov.to(dev), 1.0)
    assert f.shape == (2, 5, 1, 1) and f.device.type == "cuda" and f.dtype == torch.float32
    assert np.array_equal(f.cpu().numpy(), golden("oracle_blob.npz")["blob_global"])
    assert torch.equal(f.cpu(), oracle.estimate_global_motion(mov, 1.0))


@pytest.mark.parametrize("kw", [{"reference_frame": 0}, {"b_factor": 1000}, {"frequency_range": (200, 20)},
                                {"reference_frame": 4}])
def test_global_options(mc, dev, kw):
    mov = blob_stack(True)
    assert torch.equal(mc.estimate_global_motion(mov.to(dev), 1.0, **kw).cpu(),
                       oracle.estimate_global_motion(mov, 1.0, **kw))


@pytest.mark.parametrize("t,h,w,ps", [(8, 256, 256, 1.0), (8, 512, 512, 1.0), (6, 256, 512, 1.0),
                                      (5, 512, 256, 0.83), (3, 128, 128, 2.5)])
def test_global_drift_stacks_exact(mc, dev, t, h, w, ps):
    st, dy, dx = drift_stack(t, h, w, seed=t * 1000 + h)
    got = mc.estimate_global_motion(st.to(dev), ps).cpu()
    ref = oracle.estimate_global_motion(st, ps)
    assert torch.equal(got, ref)
    if ps == 1.0 and min(h, w) >= 512:  # the recipe's drift is recovered exactly from 512^2 up
        assert torch.equal(got[0, :, 0, 0], (dy - dy[t // 2]).float())
        assert torch.equal(got[1, :, 0, 0], (dx - dx[t // 2]).float())


def test_global_cpu_tensors_round_trip(mc):
    """reference tests pass CPU tensors with device=cpu: results come back on the CPU."""
    mov = blob_stack(True)
    f = mc.estimate_global_motion(mov, 1.0, device=torch.device("cpu"))
    assert f.device.type == "cpu" and torch.equal(f, oracle.estimate_global_motion(mov, 1.0))


def test_global_single_frame(mc, dev):
    f = mc.estimate_global_motion(blob_stack(True)[:1].to(dev), 1.0)
    assert f.shape == (2, 1, 1, 1) and float(f.abs().max()) == 0.0


# ------------------------------------------------------------------ a14/a16: spline fields


@pytest.mark.parametrize("kind", ["catmull_rom", "bspline"])
def test_spline_lattice_and_points(mc, dev, kind):
    g = torch.Generator().manual_seed(5)
    fld = torch.randn(2, 4, 3, 5, generator=g)
    a = mc.evaluate_deformation_field_at_t(fld.to(dev), 0.37, (30, 50), kind)
    assert rel_err(a, oracle.evaluate_deformation_field_at_t(fld, 0.37, (30, 50), kind)) <= 2e-6
    pts = torch.rand(4, 6, 3, generator=g)
    a = mc.evaluate_deformation_field(fld.to(dev), pts, kind)
    assert a.shape == (4, 6, 2)
    assert rel_err(a, oracle.evaluate_deformation_field(fld, pts, kind)) <= 2e-6
    one = torch.randn(2, 5, 1, 1, generator=g)
    a = mc.evaluate_deformation_field_at_t(one.to(dev), 0.25, (10, 10), kind)
    assert rel_err(a, oracle.evaluate_deformation_field_at_t(one, 0.25, (10, 10), kind)) <= 2e-6


def test_resample_and_pixel_shifts(mc, dev):
    g = torch.Generator().manual_seed(6)
    fld = torch.randn(2, 4, 3, 5, generator=g)
    assert rel_err(mc.resample_deformation_field(fld.to(dev), (6, 4, 7)),
                   oracle.resample_deformation_field(fld, (6, 4, 7))) <= 2e-6
    lat = torch.randn(2, 20, 30, generator=g)
    a = mc.get_pixel_shifts(torch.zeros(64, 96, device=dev), 1.7, lat.to(dev))
    b = oracle.get_pixel_shifts(torch.zeros(64, 96), 1.7, lat, tp.coordinate_grid((64, 96)))
    assert a.shape == (64, 96, 2) and rel_err(a, b) <= 2e-6


# ------------------------------------------------------------------ a15/a17/a18: correct_motion


@pytest.mark.parametrize("kind", ["catmull_rom", "bspline"])
def test_correct_blob_fixture(mc, dev, golden, kind):
    stat = blob_stack(False)
    out = mc.correct_motion(stat.to(dev), ramp_field().to(dev), 1.0, grid_type=kind)
    assert out.shape == stat.shape and not out.requires_grad
    key = "blob_correct_cr" if kind == "catmull_rom" else "blob_correct_bs"
    assert float((out.cpu() - torch.from_numpy(golden("oracle_blob.npz")[key])).abs().max()) <= REL
    zero = mc.correct_motion(stat.to(dev), torch.zeros(2, 5, 2, 2, device=dev), 1.0, grid_type=kind)
    assert torch.allclose(zero.cpu(), stat, atol=1e-5)  # reference asserts atol=0.1


@pytest.mark.parametrize("kind", ["catmull_rom", "bspline"])
@pytest.mark.parametrize("shape,ps", [((6, 96, 160), 1.3), ((3, 100, 130), 1.0), ((2, 64, 300), 0.7)])
def test_correct_random_field_general_kernel(mc, dev, kind, shape, ps):
    g = torch.Generator().manual_seed(sum(shape))
    img = torch.randn(*shape, generator=g)
    fld = torch.randn(2, 4, 3, 5, generator=g) * 3
    got = mc.correct_motion(img.to(dev), fld.to(dev), ps, grid_type=kind)
    ref = oracle.correct_motion(img, fld, ps, grid_type=kind)
    assert_frames_close(got, ref, knife_edge_mask(img, fld, ps, kind))


@pytest.mark.parametrize("ps", [1.0, 1.37])
def test_correct_rigid_field_fast_kernel(mc, dev, ps):
    """(2,t,1,1) fields take the separable rigid kernel; integer and fractional shifts."""
    st, dy, dx = drift_stack(8, 256, 256)
    for fld in (oracle.estimate_global_motion(st, ps),
                torch.randn(2, 8, 1, 1, generator=torch.Generator().manual_seed(9)) * 5):
        got = mc.correct_motion(st.to(dev), fld.to(dev), ps)
        ref = oracle.correct_motion(st, fld, ps)
        knife = knife_edge_mask(st, fld, ps, "catmull_rom", eps=2e-3)
        assert_frames_close(got, ref, knife)
        total = mc.motion_correct_sum(st.to(dev), fld.to(dev), ps).cpu()
        d = (total - ref.sum(0)).abs()
        d[knife.any(0)] = 0
        assert float(d.max()) <= REL * float(ref.sum(0).abs().max())


@pytest.mark.parametrize("shape", [(6, 256, 512), (3, 300, 1024), (2, 512, 1032), (4, 96, 264), (2, 128, 260),
                                   (3, 1024, 4096)])
def test_rigid_warp_reads_fp16_frames_natively(mc, dev, shape):
    """An fp16 stack through the rigid kernel (mc_warp_rigid_phase_t, MC_STORE_F16: the window is DMA'd
    as raw 16-bit samples and widened between LDS and the registers) gives EXACTLY the frames and the
    sum of its fp32 up-cast through the fp32 kernel (the conversion is exact, the arithmetic the same):
    integer and fractional shifts of both signs, shifts larger than the tile margins, tiles cut by
    every border, odd and even window start columns, a width that is not a multiple of 8 (fallback)."""
    from torch_motion_correction_amd import engine

    t, h, w = shape
    g = torch.Generator().manual_seed(h * 7 + w)
    st16 = (torch.randn(t, h, w, generator=g) * 3 + 1).half().to(dev)
    for fld in (torch.randn(2, t, 1, 1, generator=g) * 6,
                torch.round(torch.randn(2, t, 1, 1, generator=g) * 9),
                torch.tensor([[40.5] * t, [-77.25] * t])[:, :, None, None]):
        total16, frames16 = mc.motion_correct_sum(st16, fld.to(dev), 1.0, return_frames=True)
        total32, frames32 = mc.motion_correct_sum(st16.float(), fld.to(dev), 1.0, return_frames=True)
        assert torch.equal(frames16, frames32)
        assert torch.equal(total16, total32)
        assert torch.equal(mc.correct_motion(st16, fld.to(dev), 1.3), mc.correct_motion(st16.float(), fld.to(dev), 1.3))
        assert torch.equal(mc.motion_correct_sum(st16, fld.to(dev), 1.0), total32)
    # and against the oracle on the up-cast (the reference itself cannot run fp16)
    fld = torch.randn(2, t, 1, 1, generator=g) * 4
    if h * w <= 512 * 1032:
        ref = oracle.correct_motion(st16.float().cpu(), fld, 1.0)
        knife = knife_edge_mask(st16.float().cpu(), fld, 1.0, "catmull_rom", eps=2e-3)
        assert_frames_close(mc.correct_motion(st16, fld.to(dev), 1.0), ref, knife)


def test_rigid_and_general_kernels_agree(mc, dev):
    """The rigid kernel is a specialisation: same field through both kernels."""
    from torch_motion_correction_amd import api

    st, _, _ = drift_stack(6, 256, 384, seed=3)
    fld = torch.randn(2, 6, 1, 1, generator=torch.Generator().manual_seed(4)) * 4
    a = mc.correct_motion(st.to(dev), fld.to(dev), 1.1)
    api.RIGID_FAST_PATH = False
    try:
        b = mc.correct_motion(st.to(dev), fld.to(dev), 1.1)
    finally:
        api.RIGID_FAST_PATH = True
    assert_frames_close(a, b.cpu(), knife_edge_mask(st, fld, 1.1, "catmull_rom", eps=2e-3))


def test_global_then_correct_matches_golden_sum(mc, dev, golden):
    g = golden("oracle_drift_8x256.npz")
    st, _, _ = drift_stack(8, 256, 256)
    fld = mc.estimate_global_motion(st.to(dev), 1.0)
    assert np.array_equal(fld.cpu().numpy(), g["global_field"])
    total, frames = mc.motion_correct_sum(st.to(dev), fld, 1.0, return_frames=True)
    ref = torch.from_numpy(g["corrected_sum"])
    knife = knife_edge_mask(st, fld.cpu(), 1.0, "catmull_rom", eps=2e-3).any(0)
    d = (total.cpu() - ref).abs()
    d[knife] = 0
    assert float(d.max()) <= REL * float(ref.abs().max())
    assert torch.allclose(frames.sum(0), total, atol=1e-4)


def test_correct_grad_unsupported(mc, dev):
    with pytest.raises(NotImplementedError, match="grad=True"):
        mc.correct_motion(blob_stack(False).to(dev), ramp_field().to(dev), 1.0, grad=True)


# ------------------------------------------------------------------ a19: correct_motion_fast


def test_fast_matches_oracle_and_golden(mc, dev, golden):
    stat = blob_stack(False)
    got = mc.correct_motion_fast(stat.to(dev), ramp_field(g=1).to(dev))
    assert float((got.cpu() - torch.from_numpy(golden("oracle_blob.npz")["blob_fast"])).abs().max()) <= 1e-5
    zero = mc.correct_motion_fast(stat.to(dev), torch.zeros(2, 5, 1, 1, device=dev))
    assert torch.allclose(zero.cpu(), stat, atol=1e-5)  # tests/test_correct_motion.py:188-199
    g = torch.Generator().manual_seed(3)
    img = torch.randn(3, 64, 128, generator=g)
    sh = torch.randn(2, 3, 1, 1, generator=g) * 4
    assert rel_err(mc.correct_motion_fast(img.to(dev), sh.clone().to(dev)),
                   oracle.correct_motion_fast(img, sh.clone())) <= 1e-5


def test_fast_error_and_q1_side_effect(mc, dev):
    stat = blob_stack(False).to(dev)
    with pytest.raises(ValueError, match="Expected single patch deformation field"):
        mc.correct_motion_fast(stat, ramp_field().to(dev))
    f = ramp_field(g=1).to(dev)
    before = f.clone()
    mc.correct_motion_fast(stat, f)
    assert torch.equal(f, -before)  # Q1: correct_motion.py:473-474 negates the caller's tensor


# ------------------------------------------------------------------ a8: patch estimate


@pytest.mark.parametrize("strategy", ["mean_except_current", "middle_frame"])
def test_patches_blob_fixture(mc, dev, golden, strategy):
    mov = blob_stack(True)
    f, pos = mc.estimate_motion_cross_correlation_patches(mov.to(dev), 1.0, patch_sidelength=32,
                                                          reference_strategy=strategy)
    g = golden("oracle_blob.npz")
    assert f.shape == (2, 5, 2, 2) and pos.shape == (5, 2, 2, 3) and pos.dtype == torch.int64
    assert np.array_equal(pos.cpu().numpy(), g["blob_patch_pos"])
    assert float((f.cpu() - torch.from_numpy(g[f"blob_patches_{strategy}"])).abs().max()) <= REL


@pytest.mark.parametrize("kw", [
    {}, {"reference_strategy": "middle_frame"}, {"sub_pixel_refinement": False, "outlier_rejection": False},
    {"smoothing_window_size": 3}, {"outlier_threshold": 1.0}, {"temporal_smoothing": False},
    {"outlier_rejection": False}, {"reference_frame": 1}, {"b_factor": 200, "frequency_range": (150, 12)},
])
def test_patches_drift_options(mc, dev, kw):
    st, _, _ = drift_stack(8, 256, 256)
    got, pos = mc.estimate_motion_cross_correlation_patches(st.to(dev), 1.0, patch_sidelength=64, **kw)
    ref, rpos = oracle.estimate_motion_cross_correlation_patches(st, 1.0, patch_sidelength=64, **kw)
    assert torch.equal(pos.cpu(), rpos)
    assert float((got.cpu() - ref).abs().max()) <= REL


def test_patches_golden_and_bspline_correct(mc, dev, golden):
    g = golden("oracle_drift_8x256.npz")
    st, _, _ = drift_stack(8, 256, 256)
    fld, pos = mc.estimate_motion_cross_correlation_patches(st.to(dev), 1.0, patch_sidelength=64)
    assert float((fld.cpu() - torch.from_numpy(g["patch_field"])).abs().max()) <= REL
    assert np.array_equal(pos.cpu().numpy(), g["patch_pos"])
    total = mc.motion_correct_sum(st.to(dev), fld, 1.0, grid_type="bspline").cpu()
    ref = torch.from_numpy(g["patch_corrected_sum"])
    knife = knife_edge_mask(st, fld.cpu(), 1.0, "bspline").any(0)
    d = (total - ref).abs()
    d[knife] = 0
    assert float(d.max()) <= 2 * REL * float(ref.abs().max())  # field itself differs by <= 1e-4 px


@pytest.mark.parametrize("shape", [(1, 1), (3, 3)])
def test_patches_with_prior_field(mc, dev, shape):
    """cumulative estimate incl. the pre-correction and Q1's negated accumulator base"""
    st, _, _ = drift_stack(6, 256, 256, seed=21)
    prior = torch.randn(2, 6, *shape, generator=torch.Generator().manual_seed(2)) * 1.5
    a = prior.clone().to(dev)
    b = prior.clone()
    got, _ = mc.estimate_motion_cross_correlation_patches(st.to(dev), 1.0, patch_sidelength=64,
                                                          deformation_field=a)
    ref, _ = oracle.estimate_motion_cross_correlation_patches(st, 1.0, patch_sidelength=64,
                                                              deformation_field=b)
    assert torch.equal(a.cpu(), b)  # same side effect on the caller's tensor
    assert float((got.cpu() - ref).abs().max()) <= 5 * REL


def test_patches_errors(mc, dev):
    mov = blob_stack(True).to(dev)
    with pytest.raises(ValueError, match="Unknown reference_strategy"):
        mc.estimate_motion_cross_correlation_patches(mov, 1.0, patch_sidelength=32, reference_strategy="x")
    with pytest.raises(RuntimeError, match="floating point"):  # Q12, as the reference
        mc.estimate_motion_cross_correlation_patches(mov, 1.0, patch_sidelength=32, sub_pixel_refinement=False)
    with pytest.raises(ValueError, match="exceeds the frame size"):
        mc.estimate_motion_cross_correlation_patches(mov, 1.0, patch_sidelength=128)


def test_long_movie_eviction_schedule(mc, dev):
    """t > 50 exercises the memo-eviction exponent table (Q3) end to end."""
    g = torch.Generator().manual_seed(8)
    base = torch.randn(64, 64, generator=g)
    st = base[None] + 0.3 * torch.randn(52, 64, 64, generator=g)
    got, _ = mc.estimate_motion_cross_correlation_patches(st.to(dev), 1.0, patch_sidelength=32)
    ref, _ = oracle.estimate_motion_cross_correlation_patches(st, 1.0, patch_sidelength=32)
    assert float((got.cpu() - ref).abs().max()) <= REL


# ------------------------------------------------------------------ full size (BASELINE C2)


@pytest.fixture(scope="module")
def big_stack(dev):
    import bench

    stack, dy, dx = bench.synth_stack(40, 4096, 4096, 1234, dev)
    return stack, dy, dx


def test_full_size_known_drift(mc, big_stack):
    stack, dy, dx = big_stack
    f = mc.estimate_global_motion(stack, 1.0).cpu()
    assert f[0, :, 0, 0].tolist() == [float(d - dy[20]) for d in dy]
    assert f[1, :, 0, 0].tolist() == [float(d - dx[20]) for d in dx]


def test_full_size_fp16_storage_is_read_natively(mc, big_stack, dev):
    """BASELINE C2's stack stored as fp16 (N2): K1 and the rigid warp read the 16-bit samples as they are
    (mc_xc_rows_forward_stats_t, mc_warp_rigid_phase_t).  The conversion is exact and the arithmetic the
    same, so every output EQUALS that of the up-cast stack through the fp32 kernels: shifts (= the known
    drift), normalisation statistics, frames, sum; the movie pipeline takes fp16 movies too."""
    from torch_motion_correction_amd import engine, pipeline

    stack, dy, dx = big_stack
    st16 = stack[:12].half()
    up = st16.float()
    f16 = mc.estimate_global_motion(st16, 1.0)
    f32 = mc.estimate_global_motion(up, 1.0)
    assert torch.equal(f16, f32)
    assert f16[0, :, 0, 0].tolist() == [float(d - dy[6]) for d in dy[:12]]
    assert f16[1, :, 0, 0].tolist() == [float(d - dx[6]) for d in dx[:12]]
    # the filtered, normalised spectra themselves (fused statistics included)
    pl = engine.planmod.get_xc_plan(4096, 4096, 1.0, 500.0, (300, 10), dev)
    assert torch.equal(engine._global_spectra(st16, pl), engine._global_spectra(up, pl))
    t16, fr16 = mc.motion_correct_sum(st16, f16, 1.0, return_frames=True)
    t32, fr32 = mc.motion_correct_sum(up, f32, 1.0, return_frames=True)
    assert torch.equal(fr16, fr32) and torch.equal(t16, t32)
    res = pipeline.motion_correct_movies([st16, up], 1.0, device=dev)
    assert torch.equal(res[0].field, res[1].field) and torch.equal(res[0].total, res[1].total)
    assert res[0].field.cpu()[0, :, 0, 0].tolist() == [float(d - dy[6]) for d in dy[:12]]


def test_full_size_warp_properties(mc, big_stack, dev):
    stack, dy, dx = big_stack
    field = mc.estimate_global_motion(stack, 1.0)
    total, frames = mc.motion_correct_sum(stack, field, 1.0, return_frames=True)
    # (i) fused sum == sum of the written frames
    from torch_motion_correction_amd import engine

    assert float((engine.sum_frames(frames) - total).abs().max()) <= 1e-4 * float(total.abs().max())
    # (ii) aligned: away from the borders every corrected frame is the same texture
    inner = (slice(64, -64), slice(64, -64))
    resid = frames[0][inner] - frames[39][inner]
    assert float(resid.std()) < 1.6 and float(frames[0][inner].std()) > 1.3  # only the 2 noise terms
    assert float(total[inner].std()) > 35  # 40 coherent copies of the sigma=1 texture
    # (iii) linearity of the warp
    scaled, _ = engine.warp(stack * 3.0, engine.frame_lattices(field.contiguous(), 40, "catmull_rom"), 1.0,
                            want_frames=True, rigid=True)
    assert float((scaled - 3.0 * frames).abs().max()) <= 1e-4 * float(frames.abs().max()) * 3
    # (iv) zero field is the identity to fp32 rounding of the coordinate chain
    ident = mc.correct_motion(stack[:2], torch.zeros(2, 2, 1, 1, device=dev), 1.0)
    assert float((ident - stack[:2]).abs().max()) <= 2e-3 * float(stack[:2].abs().max())
    # (v) a slab of the full-size result against the oracle run on a padded crop
    f = 3
    sy, sx = int(field[0, f, 0, 0]), int(field[1, f, 0, 0])
    assert torch.allclose(frames[f, 1000:1016, 2000:2016].cpu(),
                          stack[f, 1000 + sy : 1016 + sy, 2000 + sx : 2016 + sx].cpu(), atol=5e-3)


# ------------------------------------------------------------------ round-1 additions


def test_fused_statistics_match_separate_pass(dev):
    """K1's in-flight sums + linear fix-up in K2 == separate statistics pass + direct
    normalisation (both against the oracle's normalize_image numbers)."""
    from torch_motion_correction_amd import engine, plan

    g = torch.Generator().manual_seed(12)
    img = (torch.randn(4, 256, 256, generator=g) * 2.5 + 40.0)  # |mean| >> std on purpose
    d = img.to(dev)
    pl = plan.get_xc_plan(256, 256, 1.0, 500.0, (300, 10), dev)
    fused = torch.view_as_complex(engine._global_spectra(d, pl).cpu())
    off = torch.arange(4, device=dev, dtype=torch.int64) * (256 * 256)
    sep = torch.view_as_complex(engine._forward_spectra(d, off, 256, None, pl, engine.central_box_stats(d)).cpu())
    assert float((fused - sep).abs().max() / sep.abs().max()) <= 2e-5
    st, _, _ = drift_stack(6, 512, 512, seed=5)
    import torch_motion_correction_amd as m

    assert torch.equal(m.estimate_global_motion((st * 3 + 100).to(dev), 1.0).cpu(),
                       oracle.estimate_global_motion(st * 3 + 100, 1.0))


def test_large_shifts_beyond_the_near_window(mc, dev):
    """Peaks far from zero shift (|dy| up to 200 px, negative and positive) must be found
    by the second, bounded phase of the arg-max."""
    g = torch.Generator().manual_seed(77)
    base = torch.randn(512 + 512, 512 + 512, generator=g)
    offs = [(0, 0), (200, -150), (-180, 90), (100, 230), (-70, -240)]
    st = torch.stack([base[256 - dy : 768 - dy, 256 - dx : 768 - dx] + 0.5 * torch.randn(512, 512, generator=g)
                      for dy, dx in offs])
    got = mc.estimate_global_motion(st.to(dev), 1.0, reference_frame=0).cpu()
    ref = oracle.estimate_global_motion(st, 1.0, reference_frame=0)
    assert torch.equal(got, ref)
    assert got[0, :, 0, 0].tolist() == [float(o[0]) for o in offs]
    assert got[1, :, 0, 0].tolist() == [float(o[1]) for o in offs]


def test_pure_noise_has_no_peak_and_still_matches(mc, dev):
    """Nothing can be skipped by the bound here; the (arbitrary) arg-max must still agree."""
    g = torch.Generator().manual_seed(123)
    st = torch.randn(4, 256, 256, generator=g)
    got = mc.estimate_global_motion(st.to(dev), 1.0).cpu()
    ref, ccs = oracle.estimate_global_motion(st, 1.0, return_cc=True)
    for f, cc in ccs.items():  # only compare where the oracle's own maximum is not a near-tie
        top = torch.topk(cc.flatten(), 2).values
        if float(top[0] - top[1]) > 1e-5 * float(top[0].abs()):
            assert torch.equal(got[:, f], ref[:, f])


@pytest.mark.parametrize("shape", [(3, 130, 250), (2, 96, 134)])
def test_rigid_warp_odd_widths(mc, dev, shape):
    """w % 4 != 0 takes the register-tile rigid kernel (no LDS-DMA); edge tiles everywhere."""
    g = torch.Generator().manual_seed(sum(shape))
    img = torch.randn(*shape, generator=g)
    fld = torch.randn(2, shape[0], 1, 1, generator=g) * 6
    got = mc.correct_motion(img.to(dev), fld.to(dev), 1.2)
    ref = oracle.correct_motion(img, fld, 1.2)
    assert_frames_close(got, ref, knife_edge_mask(img, fld, 1.2, "catmull_rom", eps=2e-3), max_excluded=0.05)
    total = mc.motion_correct_sum(img.to(dev), fld.to(dev), 1.2).cpu()
    assert float((total - got.cpu().sum(0)).abs().max()) <= 1e-4


# ------------------------------------------------------------------ non-power-of-two frames


@pytest.mark.parametrize("t,h,w,ps", [(6, 96, 120, 1.0), (5, 100, 64, 1.0), (4, 64, 100, 1.3),
                                      (5, 250, 372, 1.0), (3, 124, 126, 0.9), (3, 1000, 4096, 1.0),
                                      (3, 200, 1440, 1.0), (3, 4100, 128, 1.0), (3, 128, 8200, 1.0),
                                      (3, 121, 128, 1.0), (2, 959, 1024, 1.0)])
def test_global_estimate_on_arbitrary_even_sizes(mc, dev, t, h, w, ps):
    """chirp-z rows and/or columns: integer shifts must equal the oracle's exactly.
    (3, 200, 1440): the output-pruned chirp-z row plan (M = 1024 instead of 2048).
    (3, 4100, 128) / (3, 128, 8200): columns / inverse rows beyond 4096 points: chirp-z lines of
    M = 16384 (frames up to 8192 x 16384, e.g. 8184 x 11520 super-resolution movies).
    (3, 121, 128) / (2, 959, 1024): power-of-two widths whose odd height admits no row grouping for
    the power-of-two row kernels: chirp-z rows instead.
    (3, 1000, 4096): wave-per-row K1 with a mask support of 760 rows (47 full 16-row
    workgroups + a tail of 8) feeding chirp-z columns."""
    st, _, _ = drift_stack(t, h, w, seed=h * 7 + w)
    got = mc.estimate_global_motion(st.to(dev), ps).cpu()
    ref, ccs = oracle.estimate_global_motion(st, ps, return_cc=True)
    for f, cc in ccs.items():
        top = torch.topk(cc.flatten(), 2).values
        if float(top[0] - top[1]) > 1e-5 * float(top[0].abs()):  # skip oracle near-ties
            assert torch.equal(got[:, f], ref[:, f]), (f, got[:, f].flatten(), ref[:, f].flatten())


@pytest.mark.parametrize("h,w,pruned_m", [(100, 120, False), (64, 1440, True), (64, 1442, True), (72, 2880, True)])
def test_pruned_spectrum_on_arbitrary_sizes(dev, h, w, pruned_m):
    """pruned_m: the forward row pass runs the output-pruned chirp-z plan (half the circular length)."""
    from torch_motion_correction_amd import engine, plan

    g = torch.Generator().manual_seed(31)
    img = torch.randn(3, h, w, generator=g)
    pl = plan.get_xc_plan(h, w, 1.0, 500.0, (300, 10), dev)
    gm = pl.geom
    line, _ = plan.line_plan(w // 2, -1, dev, keep=gm.nkx + 1)
    assert (line.keep > 0 and line.M < plan.bluestein_size(w // 2)) == pruned_m
    d = img.to(dev)
    off = torch.arange(3, device=dev, dtype=torch.int64) * (h * w)
    S = torch.view_as_complex(engine._forward_spectra(d, off, w, None, pl, engine.central_box_stats(d)).cpu())
    spec = (torch.fft.rfftn(oracle.normalize_image(img) * tp.circle(min(h, w) / 4, (h, w), smoothing_radius=min(h, w) / 8),
                            dim=(-2, -1)) * oracle.prepare_bandpass_filter((300, 10), (h, w), 1.0)
            * tp.b_envelope(500, (h, w), 1.0))
    rows = list(range(gm.kyp)) + list(range(h - gm.kyn, h))
    sub = spec[:, rows][:, :, : gm.nkx].transpose(1, 2)
    assert float((S - sub).abs().max() / sub.abs().max()) <= 5e-6


@pytest.mark.parametrize("t,h,w", [(3, 96, 5760), (2, 64, 11520), (3, 2880, 128), (2, 96, 7000), (2, 5760, 64)])
def test_mixed_radix_lines(mc, dev, t, h, w):
    """Rows of 5760 / 11520 columns (K3 detectors) and columns of 2880 / 5760 rows are transformed
    DIRECTLY by radix-8/5/3 passes (n = 2880 = 2^6 3^2 5, 5760 = 2^7 3^2 5) instead of chirp-z; 7000
    columns take a chirp-z line of M = 5120 = 2^10 5.  The integer shifts must equal the oracle's,
    the pruned spectra the chirp-z path's and torch's."""
    from torch_motion_correction_amd import engine, plan

    st, _, _ = drift_stack(t, h, w, seed=h + w)
    got = mc.estimate_global_motion(st.to(dev), 1.0).cpu()
    ref, ccs = oracle.estimate_global_motion(st, 1.0, return_cc=True)
    for f, cc in ccs.items():
        top = torch.topk(cc.flatten(), 2).values
        if float(top[0] - top[1]) > 1e-5 * float(top[0].abs()):
            assert torch.equal(got[:, f], ref[:, f]), (f, got[:, f].flatten(), ref[:, f].flatten())
    pl = plan.get_xc_plan(h, w, 1.0, 500.0, (300, 10), dev)
    gm = pl.geom
    d = st.to(dev)
    off = torch.arange(t, device=dev, dtype=torch.int64) * (h * w)
    S = torch.view_as_complex(engine._forward_spectra(d, off, w, None, pl, engine.central_box_stats(d)).cpu())
    spec = (torch.fft.rfftn(oracle.normalize_image(st) * tp.circle(min(h, w) / 4, (h, w), smoothing_radius=min(h, w) / 8),
                            dim=(-2, -1)) * oracle.prepare_bandpass_filter((300, 10), (h, w), 1.0)
            * tp.b_envelope(500, (h, w), 1.0))
    rows = list(range(gm.kyp)) + list(range(h - gm.kyn, h))
    sub = spec[:, rows][:, :, : gm.nkx].transpose(1, 2)
    assert float((S - sub).abs().max() / sub.abs().max()) <= 5e-6
    if w // 2 in plan.DIRECT_LINE_LENGTHS or h in plan.DIRECT_LINE_LENGTHS:
        try:
            plan.USE_DIRECT_LINES = False
            plan._LINES.clear()
            S2 = torch.view_as_complex(engine._forward_spectra(d, off, w, None, pl, engine.central_box_stats(d)).cpu())
        finally:
            plan.USE_DIRECT_LINES = True
            plan._LINES.clear()
        assert float((S - S2).abs().max() / sub.abs().max()) <= 5e-6


@pytest.mark.parametrize("shape", [(3, 100, 66), (2, 96, 120), (2, 64, 90), (2, 77, 64), (2, 64, 5760),
                                   (2, 2880, 64)])
def test_correct_motion_fast_on_arbitrary_sizes(mc, dev, shape):
    g = torch.Generator().manual_seed(sum(shape))
    img = torch.randn(*shape, generator=g)
    sh = torch.randn(2, shape[0], 1, 1, generator=g) * 3
    assert rel_err(mc.correct_motion_fast(img.to(dev), sh.clone().to(dev)),
                   oracle.correct_motion_fast(img, sh.clone())) <= 2e-5


def test_odd_widths_run_on_unpacked_row_lines(mc, dev):
    """Odd widths (the reference's example movie is 959 x 927): the chirp-z row kernels take one real
    sample per line point instead of the two-per-point packing.  Global estimate, Fourier shift and
    dose-weighted sum against the oracle."""
    st, _, _ = drift_stack(5, 121, 135, seed=4)
    got = mc.estimate_global_motion(st.to(dev), 1.0).cpu()
    ref, ccs = oracle.estimate_global_motion(st, 1.0, return_cc=True)
    for f, cc in ccs.items():
        top = torch.topk(cc.flatten(), 2).values
        if float(top[0] - top[1]) > 1e-5 * float(top[0].abs()):
            assert torch.equal(got[:, f], ref[:, f]), (f, got[:, f].flatten(), ref[:, f].flatten())
    g = torch.Generator().manual_seed(8)
    img = torch.randn(3, 64, 75, generator=g)
    sh = torch.randn(2, 3, 1, 1, generator=g) * 3
    assert rel_err(mc.correct_motion_fast(img.to(dev), sh.clone().to(dev)), oracle.correct_motion_fast(img, sh.clone())) <= 2e-5
    m = img * 2.0 + 5.0
    d = mc.dose_weighted_sum(m.to(dev), 1.0, 1.2).cpu()
    r = oracle.dose_weighted_sum(m, 1.0, 1.2)
    assert float((d - r).abs().max()) <= 1e-4 * float(r.abs().max())
    with pytest.raises(NotImplementedError, match="even widths"):
        mc.estimate_global_motion(torch.randn(2, 64, 8193, device=dev), 1.0)


# ------------------------------------------------------------------ wave-per-row K1


@pytest.mark.parametrize("h,fr", [(4096, (300, 10)), (512, (300, 10)), (1024, (300, 20))])
def test_wave_row_engine_matches_workgroup_engine(dev, h, fr):
    """The wavefront-per-row K1 (mc_wave_fft.h; W = 4096, nkx <= 512) against the
    workgroup-per-row K1 on the same frames: T1, the fused box statistics, and the
    final filtered spectra.  h = 512 puts the mask support inside chunks 7..8 of the row
    (clamped sample loads + exact mask zeros), h = 4096 is the benchmark geometry, the
    (300, 20) band keeps nkx <= 256 (one kept radix-8 output pair per butterfly)."""
    from torch_motion_correction_amd import _lib, engine, plan
    from torch_motion_correction_amd._lib import check, ptr, stream_ptr

    lib = _lib.load()
    w, t = 4096, 2
    g = torch.Generator().manual_seed(h)
    img = (torch.randn(t, h, w, generator=g) * 1.7 + 11.0).to(dev)
    pl = plan.get_xc_plan(h, w, 1.0, 500.0, fr, dev)
    gm = pl.geom
    assert gm.nkx <= 512 and gm.ny % 8 == 0 and (gm.nkx <= 256) == (fr[1] == 20)
    off = torch.arange(t, device=dev, dtype=torch.int64) * (h * w)
    hl, hu, wl, wu = int(0.25 * h), int(0.75 * h), int(0.25 * w), int(0.75 * w)
    if h == 512:  # the statistics box must lie inside the region K1 reads; unaligned to the
        wl, wu = 1900, 2200  # 256-px chunks here: the fused statistics stay on the workgroup kernel
    elif h == 1024:
        wl, wu = 1792, 2304  # chunk-aligned: fused statistics on the wave kernel
    assert gm.x0 <= wl and wu <= gm.x1 and gm.y0 <= hl and hu <= gm.y0 + gm.ny
    m0 = torch.tensor([11.0, 1.0, 1.0], device=dev)
    res = {}
    try:
        for mode in (1, 0):
            check(lib.mc_xc_row_engine(mode), "mc_xc_row_engine")
            T1 = torch.zeros((t, gm.nkx, gm.ny, 2), device=dev)
            acc = torch.empty(128, dtype=torch.float64, device=dev)
            fix = torch.empty(2, device=dev)
            out3 = torch.empty(3, device=dev)
            check(lib.mc_xc_rows_forward_stats(ptr(img), ptr(off), w, ptr(pl.mask), ptr(m0), ptr(T1),
                                               ptr(pl.tw_row), t, gm, hl, hu, wl, wu, ptr(acc), ptr(fix),
                                               ptr(out3), ptr(pl.chord) if mode == 0 else None,
                                               stream_ptr(dev)), "rows_forward_stats")
            plain = torch.zeros_like(T1)
            st = torch.tensor([11.0, 0.5], device=dev)
            check(lib.mc_xc_rows_forward(ptr(img), ptr(off), w, None, ptr(pl.mask), ptr(st), ptr(plain),
                                         ptr(pl.tw_row), t, gm, stream_ptr(dev)), "rows_forward")
            S = engine._global_spectra(img, pl)
            res[mode] = [x.cpu() for x in (T1, out3, plain, S)]
    finally:
        lib.mc_xc_row_engine(0)
    for a, b, tol in zip(res[0], res[1], (3e-6, 1e-6, 3e-6, 1e-5)):
        assert torch.isfinite(a).all()
        assert float((a - b).abs().max()) <= tol * float(b.abs().max())
    # statistics against torch on the central box
    std, mean = torch.std_mean(img[:, hl:hu, wl:wu].double())
    assert float(res[0][1][0]) == pytest.approx(float(mean), rel=1e-6)
    assert float(res[0][1][2]) == pytest.approx(float(std), rel=1e-5)


# ------------------------------------------------------------------ movie pipeline


@pytest.mark.parametrize("overlap", [True, False])
def test_movie_pipeline_equals_sequential_calls(mc, dev, overlap):
    """motion_correct_movies (two-stream pipeline) returns, movie by movie, exactly what
    estimate_global_motion followed by motion_correct_sum returns, and matches the oracle."""
    movies, truth = [], []
    for i, (t, n) in enumerate([(6, 512), (5, 256), (8, 512), (6, 512)]):
        st, dy, dx = drift_stack(t, n, n, seed=100 + i)
        movies.append((st * (1.0 + i) + 3.0 * i).to(dev))
        truth.append((dy - dy[t // 2], dx - dx[t // 2]))
    res = mc.motion_correct_movies(movies, 1.0, return_frames=True, overlap=overlap)
    torch.cuda.synchronize()
    assert len(res) == len(movies)
    for m, r, (ty, tx) in zip(movies, res, truth):
        field = mc.estimate_global_motion(m, 1.0)
        total, frames = mc.motion_correct_sum(m, field, 1.0, return_frames=True)
        assert torch.equal(r.field, field)
        assert torch.equal(r.total, total) and torch.equal(r.frames, frames)
        assert r.field[0, :, 0, 0].cpu().tolist() == [float(v) for v in ty]
        assert r.field[1, :, 0, 0].cpu().tolist() == [float(v) for v in tx]
    o = oracle.estimate_global_motion(movies[0].cpu(), 1.0)
    assert torch.equal(res[0].field.cpu(), o)
    # sum-only mode, and the generator form used by bench.py
    only = mc.motion_correct_movies(movies[:2], 1.0, overlap=overlap)
    assert only[0].frames is None and torch.equal(only[1].total, res[1].total)
    pipe = mc.MoviePipeline(dev, 1.0, return_frames=False, overlap=overlap)
    last = None
    for last in pipe.iterate([movies[2]] * 3):
        pass
    torch.cuda.synchronize()
    assert torch.equal(last.total, res[2].total)
    with pytest.raises(ValueError):
        mc.motion_correct_movies([movies[0][0]], 1.0)


# ------------------------------------------------------------------ near-window search + fallback


def test_near_window_search_and_device_side_fallback(mc, dev):
    """H >= 1024 goes through mc_xc_correlate_argmax: near-window rows first, the full map
    only when a far row can still win.  (i) small drift: settled in the near window;
    (ii) shifts of several hundred pixels: the gated full pass must find them;
    (iii) pure noise: nothing can be skipped, the arg-max must still be the oracle's."""
    n = 1024
    g = torch.Generator().manual_seed(31)
    base = torch.randn(n + 1024, n + 1024, generator=g)
    offs = [(0, 0), (3, -5), (300, -210), (-410, 95), (60, 500), (-64, -65)]
    st = torch.stack([base[512 - dy : 512 - dy + n, 512 - dx : 512 - dx + n] + 0.5 * torch.randn(n, n, generator=g)
                      for dy, dx in offs])
    got = mc.estimate_global_motion(st.to(dev), 1.0, reference_frame=0).cpu()
    assert got[0, :, 0, 0].tolist() == [float(o[0]) for o in offs]
    assert got[1, :, 0, 0].tolist() == [float(o[1]) for o in offs]
    assert torch.equal(got, oracle.estimate_global_motion(st, 1.0, reference_frame=0))
    noise = torch.randn(3, n, n, generator=g)
    gotn = mc.estimate_global_motion(noise.to(dev), 1.0).cpu()
    refn, ccs = oracle.estimate_global_motion(noise, 1.0, return_cc=True)
    for f, cc in ccs.items():  # compare where the oracle's own maximum is not a near-tie
        top = torch.topk(cc.flatten(), 2).values
        if float(top[0] - top[1]) > 1e-4 * float(top[0].abs()):
            assert torch.equal(gotn[:, f], refn[:, f])


# ------------------------------------------------------------------ a20 / N1: dose-weighted sum


@pytest.mark.parametrize("shape,ps,dose,pre,kv", [((6, 256, 256), 1.0, 1.5, 0.0, 300.0),
                                                  ((5, 64, 96), 1.3, 0.8, 2.0, 200.0),
                                                  ((3, 100, 72), 0.83, 2.5, 0.5, 300.0)])
def test_dose_weighted_sum_matches_oracle(mc, dev, shape, ps, dose, pre, kv):
    """sum_f irfft2(q_f rfft2(frame_f)) accumulated in Fourier space (one inverse transform)
    against the oracle's per-frame restatement of examples/ttMotion.py:331-351; power-of-two
    and chirp-z sizes.  Third-party filter semantics: parity unpinned."""
    g = torch.Generator().manual_seed(sum(shape))
    m = torch.randn(*shape, generator=g) * 2.0 + 5.0
    got = mc.dose_weighted_sum(m.to(dev), ps, dose, pre_exposure=pre, voltage=kv).cpu()
    ref = oracle.dose_weighted_sum(m, ps, dose, pre_exposure=pre, voltage=kv)
    assert got.shape == ref.shape
    assert float((got - ref).abs().max()) <= 1e-4 * float(ref.abs().max())


def test_motion_correct_sum_with_dose_weighting(mc, dev):
    st, dy, dx = drift_stack(6, 256, 256, seed=9)
    field = mc.estimate_global_motion(st.to(dev), 1.0)
    total, frames = mc.motion_correct_sum(st.to(dev), field, 1.0, return_frames=True, dose_per_frame=1.2,
                                          pre_exposure=0.3)
    plain = mc.correct_motion(st.to(dev), field, 1.0)
    assert torch.equal(frames, plain)
    ref = oracle.dose_weighted_sum(plain.cpu(), 1.0, 1.2, pre_exposure=0.3)
    assert float((total.cpu() - ref).abs().max()) <= 1e-4 * float(ref.abs().max())
    # zero dose = plain sum / sqrt(t)
    z = mc.motion_correct_sum(st.to(dev), field, 1.0, dose_per_frame=0.0).cpu()
    assert float((z - plain.sum(0).cpu() / 6**0.5).abs().max()) <= 1e-4 * float(z.abs().max())


@pytest.mark.parametrize("rigid", [True, False])
def test_motion_correct_sum_dose_weighting_streams_chunks(mc, dev, rigid):
    """motion_correct_sum(dose_per_frame=...) without return_frames warps, transforms and weights the
    movie a chunk of frames at a time (the corrected movie is never held): same sum as the
    all-frames form and as the oracle's dose weighting of the corrected frames, with the workspace
    squeezed to two frames per chunk (7 frames: a last chunk of one)."""
    from torch_motion_correction_amd import engine

    st, dy, dx = drift_stack(7, 256, 512, seed=21)
    if rigid:
        field = mc.estimate_global_motion(st.to(dev), 1.0)
    else:
        g = torch.Generator().manual_seed(5)
        field = (torch.randn(2, 7, 3, 4, generator=g) * 1.5).to(dev)
    whole, frames = mc.motion_correct_sum(st.to(dev), field, 1.0, return_frames=True, dose_per_frame=0.9,
                                          pre_exposure=0.5, voltage=200.0)
    try:
        ws, engine.WORKSPACE_BYTES = engine.WORKSPACE_BYTES, 2 * 256 * (512 // 2 + 16) * 8
        streamed = mc.motion_correct_sum(st.to(dev), field, 1.0, dose_per_frame=0.9, pre_exposure=0.5,
                                         voltage=200.0)
    finally:
        engine.WORKSPACE_BYTES = ws
    ref = oracle.dose_weighted_sum(frames.cpu(), 1.0, 0.9, pre_exposure=0.5, voltage=200.0)
    assert float((streamed.cpu() - ref).abs().max()) <= 1e-4 * float(ref.abs().max())
    assert float((streamed - whole).abs().max()) <= 2e-5 * float(ref.abs().max())


# ------------------------------------------------------------------ wave-per-row K1, patch rows


def test_wave512_patch_rows_match_workgroup_engine(dev):
    """mc_xc_rows_forward_dual (one wavefront per 1024-sample row, U and V from one read)
    against the workgroup-per-row K1 run twice, on patch jobs at odd and even offsets."""
    from torch_motion_correction_amd import _lib, plan
    from torch_motion_correction_amd._lib import check, ptr, stream_ptr

    lib = _lib.load()
    p, h, w = 1024, 1500, 2200
    g = torch.Generator().manual_seed(8)
    img = (torch.randn(2, h, w, generator=g) * 1.3 + 4.0).to(dev)
    pl = plan.get_xc_plan(p, p, 1.0, 500.0, (300, 10), dev)
    gm = pl.geom
    assert gm.W == 1024 and gm.nkx <= 128 and gm.ny % 8 == 0
    origins = [0, 7 * w + 13, 300 * w + 1101, h * w + 476 * w + 1176]  # odd and even x offsets
    off = torch.tensor(origins, dtype=torch.int64, device=dev)
    n = len(origins)
    ea = torch.tensor([1, 2, 1, 3], dtype=torch.int32, device=dev)
    eb = torch.tensor([2, 4, 2, 1], dtype=torch.int32, device=dev)
    stats = torch.tensor([4.0, 0.7], device=dev)
    st = stream_ptr(dev)
    Ua, Ub, Ra, Rb, Sa = (torch.zeros((n, gm.nkx, gm.ny, 2), device=dev) for _ in range(5))
    check(lib.mc_xc_rows_forward_dual(ptr(img), ptr(off), w, ptr(ea), ptr(eb), ptr(pl.mask), ptr(stats), ptr(Ua),
                                      ptr(Ub), ptr(pl.tw_row), n, gm, ptr(pl.chord), st), "dual")  # chord-clamped loads
    check(lib.mc_xc_rows_forward_dual(ptr(img), ptr(off), w, ptr(ea), None, ptr(pl.mask), ptr(stats), ptr(Sa), None,
                                      ptr(pl.tw_row), n, gm, None, st), "single")  # box-clamped loads
    check(lib.mc_xc_rows_forward(ptr(img), ptr(off), w, ptr(ea), ptr(pl.mask), ptr(stats), ptr(Ra), ptr(pl.tw_row),
                                 n, gm, st), "wg a")
    check(lib.mc_xc_rows_forward(ptr(img), ptr(off), w, ptr(eb), ptr(pl.mask), ptr(stats), ptr(Rb), ptr(pl.tw_row),
                                 n, gm, st), "wg b")
    for got, ref in ((Ua, Ra), (Ub, Rb), (Sa, Ra)):
        got, ref = got.cpu(), ref.cpu()
        assert torch.isfinite(got).all()
        assert float((got - ref).abs().max()) <= 3e-6 * float(ref.abs().max())


def test_radix16_column_engine_matches_stockham_columns(mc, dev):
    """H = 4096 columns: the register-resident radix-16 transform (K2 with pruned outputs, the
    near-window K3 with pruned inputs) against the radix-8 Stockham passes: filtered spectra,
    and the arg-max search end to end (near window, bounds, device-side fallback)."""
    from torch_motion_correction_amd import _lib, engine, plan
    from torch_motion_correction_amd._lib import check

    lib = _lib.load()
    g = torch.Generator().manual_seed(41)
    base = torch.randn(4096 + 256, 4096 + 256, generator=g)
    offs = [(0, 0), (5, -3), (-2, 7), (100, -90)]  # the last one lies beyond the near window
    img = torch.stack([base[128 - dy : 128 - dy + 4096, 128 - dx : 128 - dx + 4096] for dy, dx in offs])
    img = (img + 0.5 * torch.randn(len(offs), 4096, 4096, generator=g)).to(dev)
    pl = plan.get_xc_plan(4096, 4096, 1.0, 500.0, (300, 10), dev)
    res = {}
    try:
        for mode in (1, 0):
            check(lib.mc_xc_col_engine(mode), "mc_xc_col_engine")
            S = engine._global_spectra(img, pl)
            f = mc.estimate_global_motion(img, 1.0, reference_frame=0)
            res[mode] = (S.cpu(), f.cpu())
    finally:
        lib.mc_xc_col_engine(0)
    assert torch.isfinite(res[0][0]).all()
    assert float((res[0][0] - res[1][0]).abs().max()) <= 1e-5 * float(res[1][0].abs().max())
    assert torch.equal(res[0][1], res[1][1])
    assert res[0][1][0, :, 0, 0].tolist() == [float(o[0]) for o in offs]
    assert res[0][1][1, :, 0, 0].tolist() == [float(o[1]) for o in offs]


def test_wave1024_patch_columns_match_workgroup_engine(dev):
    """H = 1024 columns (patches): one wavefront per column (K2 with pruned outputs, K3 with
    pruned inputs) against the workgroup-per-column Stockham kernels."""
    from torch_motion_correction_amd import _lib, engine, plan
    from torch_motion_correction_amd._lib import check, ptr, stream_ptr

    lib = _lib.load()
    p, h, w = 1024, 1300, 2100
    g = torch.Generator().manual_seed(10)
    img = (torch.randn(3, h, w, generator=g) * 0.9 + 2.0).to(dev)
    pl = plan.get_xc_plan(p, p, 1.0, 500.0, (300, 10), dev)
    gm = pl.geom
    assert gm.H == 1024 and gm.kyp <= 128 and gm.kyn <= 128
    off = torch.tensor([0, 201 * w + 77, h * w + 100 * w + 1000, 2 * h * w + 276 * w + 1076], dtype=torch.int64,
                       device=dev)
    ex = torch.tensor([1, 2, 1, 2], dtype=torch.int32, device=dev)
    stats = torch.tensor([2.0, 1.1], device=dev)
    n = int(off.numel())
    cur = torch.arange(n, dtype=torch.int32, device=dev)
    ref = torch.tensor([1, 0, 3, 2], dtype=torch.int32, device=dev)
    st = stream_ptr(dev)
    res = {}
    try:
        for mode in (1, 0):
            check(lib.mc_xc_col_engine(mode), "mc_xc_col_engine")
            S = engine._forward_spectra(img, off, w, ex, pl, stats, min_expo=1)
            T2 = torch.zeros((n, gm.nkx, gm.H, 2), device=dev)
            check(lib.mc_xc_cols_inverse(ptr(S), ptr(cur), ptr(S), ptr(ref), ptr(T2), ptr(pl.tw_col),
                                         1.0 / (p * p), n, gm, st), "mc_xc_cols_inverse")
            res[mode] = (S.cpu(), T2.cpu())
    finally:
        lib.mc_xc_col_engine(0)
    for a, b in zip(res[0], res[1]):
        assert torch.isfinite(a).all()
        assert float((a - b).abs().max()) <= 5e-6 * float(b.abs().max())


def test_wide_band_on_4096_frames_takes_the_general_kernels(mc, dev):
    """frequency_range (300, 4) keeps 1025 rfft columns and 2049 fft rows of a 4096^2 frame:
    beyond what the wave-per-row K1 (nkx <= 512) and the radix-16 columns (<= 512 kept rows per
    end) are built for, so the workgroup kernels run; the drift must still be recovered."""
    st, dy, dx = drift_stack(3, 4096, 4096, seed=21)
    f = mc.estimate_global_motion(st.to(dev), 1.0, frequency_range=(300, 4)).cpu()
    assert f[0, :, 0, 0].tolist() == [float(d - dy[1]) for d in dy]
    assert f[1, :, 0, 0].tolist() == [float(d - dx[1]) for d in dx]


# ------------------------------------------------------------------ against the reference's own helpers


def test_product_matches_reference_helper_vectors(mc, dev):
    """tests/golden/reference_helpers.npz holds outputs of the reference's OWN functions
    (oracle/make_goldens.py::reference_helper_vectors): normalize_image (utils.py:49-84) and
    get_pixel_shifts (correct_motion.py:132-185) are compared with the HIP path directly."""
    from torch_motion_correction_amd import engine

    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "reference_helpers.npz")
    ref = {k: torch.from_numpy(v) for k, v in np.load(path, allow_pickle=False).items() if v.ndim}
    img = ref["norm_in"].to(dev)
    got = engine.normalize(img, engine.central_box_stats(img)).cpu()
    assert float((got - ref["norm_out"]).abs().max()) <= 2e-6 * float(ref["norm_out"].abs().max())
    yy, xx = torch.meshgrid(torch.arange(37, dtype=torch.float32), torch.arange(53, dtype=torch.float32),
                            indexing="ij")
    shifts = mc.get_pixel_shifts(torch.zeros(37, 53, device=dev), 1.3, ref["gps_lattice"].to(dev),
                                 torch.stack([yy, xx], dim=-1).to(dev)).cpu()
    assert shifts.shape == ref["gps_out"].shape
    assert float((shifts - ref["gps_out"]).abs().max()) <= 1e-5 * float(ref["gps_out"].abs().max())


@pytest.mark.parametrize("strategy", ["mean_except_current", "middle_frame"])
def test_patches_1024_near_window_search_with_sub_pixel(mc, dev, strategy):
    """p = 1024 patches go through the wave kernels (rows 512-point dual, columns 1024-point)
    and the near-window search incl. the 3x3 neighbourhood read from the compact buffer; the
    result must equal the full-map path and the oracle."""
    from torch_motion_correction_amd import engine

    st, dy, dx = drift_stack(5, 1100, 1600, seed=17)  # 1 x 2 patches
    got = {}
    try:
        for fused in (False, True):
            engine.FUSED_SEARCH = fused
            f, pos = mc.estimate_motion_cross_correlation_patches(st.to(dev), 1.0, patch_sidelength=1024,
                                                                  reference_strategy=strategy)
            got[fused] = (f.cpu(), pos.cpu())
    finally:
        engine.FUSED_SEARCH = True
    assert torch.equal(got[True][1], got[False][1])
    assert float((got[True][0] - got[False][0]).abs().max()) <= 1e-4
    of, opos = oracle.estimate_motion_cross_correlation_patches(st, 1.0, patch_sidelength=1024,
                                                                reference_strategy=strategy)
    assert torch.equal(got[True][1], opos)
    assert float((got[True][0] - of).abs().max()) <= 1e-4


# ------------------------------------------------------------------ N2: input conditioning


@pytest.mark.parametrize("dtype", [torch.uint8, torch.int16, torch.float16, torch.float32])
@pytest.mark.parametrize("hw", [(70, 90), (72, 96), (64, 2056)])
def test_condition_movie_matches_the_example_pipeline(mc, dev, dtype, hw):
    """gain multiply + per-frame mean-zero (examples/ttMotion.py:90-121, 174-199) straight from
    the storage type, against the example's numpy arithmetic.  (70, 90): h*w % 8 != 0, the
    per-frame kernels; the others: the tiled kernels (gain tile held in registers over the frames),
    (64, 2056) with more than one workgroup and a partial last one."""
    g = torch.Generator().manual_seed(3)
    raw = (torch.rand(3, *hw, generator=g) * 40).to(dtype)
    gain = torch.rand(*hw, generator=g) * 0.4 + 0.8
    got = mc.condition_movie(raw.to(dev), gain.to(dev)).cpu()
    x = raw.float().numpy().astype(np.float64) * gain.numpy().astype(np.float64)
    ref = torch.from_numpy((x - x.mean(axis=(1, 2), keepdims=True)).astype(np.float32))
    assert got.dtype == torch.float32 and got.shape == ref.shape
    assert float((got - ref).abs().max()) <= 2e-5 * float(ref.abs().max())
    assert float(got.mean(dim=(1, 2)).abs().max()) < 1e-4
    plain = mc.condition_movie(raw.to(dev), None, mean_zero=False).cpu()
    assert torch.equal(plain, raw.float())
    with pytest.raises(TypeError):
        mc.condition_movie(raw.double().to(dev))
    with pytest.raises(ValueError):
        mc.condition_movie(raw.to(dev), gain[:10].to(dev))


# ------------------------------------------------------------------ estimate_local_motion


def _local_case(t=5, h=96, w=112, seed=5):
    st, _, _ = drift_stack(t, h, w, seed=seed)
    return st


@pytest.mark.parametrize("loss_type", ["mse", "cc", "ncc"])
@pytest.mark.parametrize("patch,res,grid_type", [((32, 32), (5, 2, 2), "catmull_rom"), ((32, 48), (3, 2, 3), "bspline")])
def test_local_motion_loss_and_gradient_match_autograd_oracle(dev, loss_type, patch, res, grid_type):
    """The HIP loss and its analytic gradient w.r.t. the spline parameters against the oracle's
    autograd through rfftn -> Fourier shift -> filters -> leave-one-out reference -> loss
    (estimate_motion_optimizer.py:361-417), at a non-trivial parameter point.  Tolerance: 2e-4 of
    the largest gradient entry (fp32 phase of up to ~30 rad on both sides)."""
    from torch_motion_correction_amd import local_motion

    st = _local_case()
    g = torch.Generator().manual_seed(11)
    new = (torch.randn(2, *res, generator=g) * 1.5).requires_grad_(True)
    init = torch.randn(2, *res, generator=g) * 0.7
    oprob = oracle.LocalMotionProblem(st, 1.2, patch)
    total = None
    for a in range(0, oprob.npatch, 8):
        l = oprob.batch_loss(new, init, grid_type, list(range(a, min(a + 8, oprob.npatch))), loss_type)
        total = l if total is None else total + l
    total.backward()
    prob = local_motion.LocalMotionProblem(st.to(dev), 1.2, patch, res, grid_type)
    assert (prob.gh, prob.gw) == (oprob.gh, oprob.gw)
    sizes = np.minimum(8, prob.npatch - (np.arange(prob.npatch) // 8) * 8)
    wb = torch.from_numpy(1.0 / sizes.astype(np.float64)).to(dev)
    nd = new.detach().to(dev).requires_grad_(True)
    loss = local_motion._Loss.apply(prob.shifts_px(nd, init.to(dev)), prob, wb, loss_type)
    loss.backward()
    lv, tv = loss.item(), total.item()
    assert abs(lv - tv) <= 2e-4 * abs(tv), (lv, tv)
    err = (nd.grad.cpu() - new.grad).abs().max() / new.grad.abs().max()
    assert float(err) <= 2e-4, float(err)


@pytest.mark.parametrize("optimizer_type,loss_type,n_it", [("adam", "mse", 4), ("sgd", "cc", 3), ("lbfgs", "mse", 3),
                                                          ("rmsprop", "ncc", 3)])
def test_local_motion_short_runs_follow_the_oracle(mc, dev, optimizer_type, loss_type, n_it):
    """A few optimiser steps end to end, with an initial field and a trajectory.  Adam / RMSprop
    normalise the gradient, which turns 1e-5 relative gradient differences into visible parameter
    differences where a gradient entry is near zero, hence the looser bound (5 % of the step size)."""
    st = _local_case(t=5, h=96, w=96, seed=9)
    init = torch.randn(2, 5, 1, 1, generator=torch.Generator().manual_seed(2)) * 0.5
    kw = dict(patch_shape=(32, 32), deformation_field_resolution=(5, 2, 2), initial_deformation_field=init,
              n_iterations=n_it, optimizer_type=optimizer_type, loss_type=loss_type, return_trajectory=True)
    import random

    # 25 patches in batches of 8: which patch is alone in the last batch of a pass is decided by the
    # reference's random.shuffle (patch_utils.py:163-164); same seed, same passes
    random.seed(7)
    ref, rtr = oracle.estimate_local_motion(st, 1.0, **kw)
    after_oracle = random.random()
    random.seed(7)
    got, gtr = mc.estimate_local_motion(st.to(dev), 1.0, **kw)
    assert random.random() == after_oracle  # the global random state advanced identically
    assert got.shape == ref.shape == (2, 5, 2, 2) and got.device.type == "cuda"
    step = float((ref - (oracle.resample_deformation_field(init, (5, 2, 2)) - 0)).abs().max())
    assert float((got.cpu() - ref).abs().max()) <= 0.05 * max(step, 1e-3)
    assert [c.step for c in gtr.checkpoints] == [c.step for c in rtr.checkpoints]
    for a, b in zip(gtr.checkpoints, rtr.checkpoints):
        assert abs(a.loss - b.loss) <= 1e-3 * abs(b.loss) + 1e-9


def test_local_motion_lbfgs_subsample_uses_the_shuffled_patches(mc, dev):
    """lbfgs_patch_subsample keeps the FIRST n patches of every closure evaluation's shuffled order
    (estimate_motion_optimizer.py:295-303): a random subset per evaluation, the reference's for the seed."""
    import random

    st = _local_case(t=4, h=96, w=96, seed=3)
    kw = dict(patch_shape=(32, 32), deformation_field_resolution=(4, 2, 2), n_iterations=2,
              optimizer_type="lbfgs", loss_type="mse", optimizer_kwargs={"lbfgs_patch_subsample": 6})
    random.seed(11)
    ref = oracle.estimate_local_motion(st, 1.0, **kw)
    random.seed(11)
    got = mc.estimate_local_motion(st.to(dev), 1.0, **kw)
    scale = max(float(ref.abs().max()), 1e-3)
    assert float((got.cpu() - ref).abs().max()) <= 0.05 * scale
    random.seed(12)  # another seed, other subsets: a different (but again matching) answer
    other = mc.estimate_local_motion(st.to(dev), 1.0, **kw)
    assert float((other - got).abs().max()) > 0


def test_local_motion_recovers_a_known_drift(mc, dev):
    """Frames displaced by a known small per-frame shift (inside the basin of the 10-pixel band
    limit): refinement from zero must find it.  frame_f(x) = base(x - d_f)  =>  field ~ d_f - mean."""
    g = torch.Generator().manual_seed(21)
    base = torch.randn(128 + 16, 128 + 16, generator=g)
    dy, dx = [-2, -1, 0, 0, 1, 2], [1, 1, 0, 0, -1, -1]
    st = torch.stack([base[8 - dy[f]:8 - dy[f] + 128, 8 - dx[f]:8 - dx[f] + 128]
                      + 0.5 * torch.randn(128, 128, generator=g) for f in range(6)])
    field = mc.estimate_local_motion(st.to(dev), 1.0, (64, 64), (6, 1, 1), None, n_iterations=150,
                                     optimizer_kwargs={"lr": 0.05}).cpu()
    mean = float(np.mean(dy + dx))
    assert float((field[0, :, 0, 0] - (torch.tensor(dy, dtype=torch.float32) - mean)).abs().max()) < 0.35
    assert float((field[1, :, 0, 0] - (torch.tensor(dx, dtype=torch.float32) - mean)).abs().max()) < 0.35


def test_local_motion_argument_errors(mc, dev):
    st = torch.randn(4, 64, 64, device=dev)
    with pytest.raises(ValueError, match="Invalid grid type"):
        mc.estimate_local_motion(st, 1.0, (32, 32), (4, 1, 1), grid_type="linear", n_iterations=1)
    with pytest.raises(ValueError, match="Unsupported optimizer"):
        mc.estimate_local_motion(st, 1.0, (32, 32), (4, 1, 1), optimizer_type="adagrad", n_iterations=1)


# ------------------------------------------------------------------ correct_motion_two_grids / _slow


class _FakeSplineGrid:
    """Stands for a grid object of the reference's spline dependency: `.data` + class name."""

    def __init__(self, data):
        self.data = data


class CubicBSplineGrid3d(_FakeSplineGrid):
    pass


class CubicCatmullRomGrid3d(torch.nn.Module):
    """Module form, as the reference's tests build it: `.data` is a Parameter."""

    def __init__(self, data):
        super().__init__()
        self._data = torch.nn.Parameter(data)

    @property
    def data(self):
        return self._data


def test_reference_suite_two_grids_and_slow(mc, dev):
    """tests/test_correct_motion.py:202-253 (slow) and :304-553 (two grids) of the reference, on its
    fixtures: shapes, devices, finiteness, zero-field identity within atol 0.1, detached output for
    grad=False, attached output for grad=True."""
    img = blob_stack(False)
    t = img.shape[0]
    field = ramp_field(t, 2)
    new = CubicCatmullRomGrid3d(field.clone().to(dev))
    zero = CubicCatmullRomGrid3d(torch.zeros(2, t, 2, 2, device=dev))
    out = mc.correct_motion_two_grids(image=img, new_deformation_grid=new, base_deformation_grid=zero,
                                      pixel_spacing=1.0, device=dev)
    assert out.shape == img.shape and out.device.type == "cuda" and torch.isfinite(out).all()
    assert out.requires_grad  # test_gradient_preservation, first half
    det = mc.correct_motion_two_grids(img, new, zero, 1.0, grad=False, device=dev)
    assert not det.requires_grad and torch.equal(det, out.detach())
    bs = mc.correct_motion_two_grids(img, CubicBSplineGrid3d(field.clone().to(dev)),
                                     CubicBSplineGrid3d(torch.zeros(2, t, 2, 2, device=dev)), 1.0, device=dev)
    assert bs.shape == img.shape
    ident = mc.correct_motion_two_grids(img, zero, zero, 1.0, device=dev)
    assert torch.allclose(ident.detach().cpu(), img, atol=0.1)
    slow = mc.correct_motion_slow(image=img, deformation_grid=field, device=dev)
    assert slow.shape == img.shape and slow.device.type == "cuda" and not slow.requires_grad
    assert torch.allclose(mc.correct_motion_slow(img, torch.zeros(2, t, 2, 2), device=dev).cpu(), img, atol=0.1)


def test_reference_suite_local_motion(mc, dev):
    """tests/test_estimate_motion.py:197-304 of the reference on its fixture (5 x 64 x 64 blob,
    32-px patches, (t, 2, 2) grid, 2 iterations): shapes for every optimiser / basis / loss it
    exercises, trajectory returned, optimiser kwargs accepted."""
    img = blob_stack(True)
    t = img.shape[0]
    kw = dict(pixel_spacing=1.0, patch_shape=(32, 32), deformation_field_resolution=(t, 2, 2),
              initial_deformation_field=None, device=dev, n_iterations=2)
    for extra in ({"optimizer_type": "adam"}, {"optimizer_type": "sgd"}, {"grid_type": "bspline"},
                  {"loss_type": "ncc"}, {"optimizer_type": "adam", "optimizer_kwargs": {"lr": 0.001}},
                  {"initial_deformation_field": torch.zeros(2, t, 2, 2)}):
        out = mc.estimate_local_motion(image=img, **{**kw, **extra})
        assert out.shape == (2, t, 2, 2) and isinstance(out, torch.Tensor) and torch.isfinite(out).all()
    out, traj = mc.estimate_local_motion(image=img, **kw, return_trajectory=True)
    assert out.shape == (2, t, 2, 2) and traj is not None and len(traj.checkpoints) == 2
    # as the reference's tests call it: CPU image, device=cpu -> staged to the GPU, field returned on the CPU
    cpu_out = mc.estimate_local_motion(image=img, **{**kw, "device": torch.device("cpu")})
    assert cpu_out.device.type == "cpu" and cpu_out.shape == (2, t, 2, 2)


def _coord_knife(coords, h, w, eps=2e-3):
    near = lambda v, n: (v.abs() < eps) | ((v - (n - 1)).abs() < eps)
    return near(coords[..., 0], h) | near(coords[..., 1], w)


def test_correct_motion_two_grids_matches_oracle(mc, dev):
    g = torch.Generator().manual_seed(41)
    img = torch.randn(4, 72, 88, generator=g)
    new = torch.randn(2, 4, 2, 3, generator=g) * 2.0
    base = torch.randn(2, 3, 3, 2, generator=g) * 1.5  # other resolution, B-spline basis
    got = mc.correct_motion_two_grids(img.to(dev), new.to(dev), CubicBSplineGrid3d(base.to(dev)), 1.3,
                                      grad=False)
    ref = oracle.correct_motion_two_grids(img, new, base, 1.3, "catmull_rom", "bspline")
    grid = tp.coordinate_grid((72, 88))
    knife = torch.zeros(4, 72, 88, dtype=torch.bool)
    for i, ft in enumerate(torch.linspace(0, 1, steps=4)):
        lat = (oracle.evaluate_deformation_field_at_t(new, ft, (20, 30), "catmull_rom")
               + oracle.evaluate_deformation_field_at_t(base, ft, (20, 30), "bspline"))
        knife[i] = _coord_knife(grid + oracle.get_pixel_shifts(img[i], 1.3, lat, grid), 72, 88)
    assert_frames_close(got, ref, knife, max_excluded=0.05)
    # grad=True with a grid that requires gradients: forward as in the reference (attached result),
    # backward refused loudly
    att = mc.correct_motion_two_grids(img.to(dev), new.to(dev).requires_grad_(True), base.to(dev), 1.3)
    assert att.requires_grad and torch.equal(att.detach(), mc.correct_motion_two_grids(
        img.to(dev), new.to(dev), base.to(dev), 1.3, grad=False))
    with pytest.raises(NotImplementedError, match="forward only"):
        att.sum().backward()
    # same two tensors, default grad=True but nothing requires gradients: allowed
    again = mc.correct_motion_two_grids(img.to(dev), new.to(dev), CubicBSplineGrid3d(base.to(dev)), 1.3)
    assert torch.equal(again, got)


def test_correct_motion_slow_matches_oracle(mc, dev):
    """Per-pixel spline evaluation, shifts in PIXELS (correct_motion.py:302-427)."""
    g = torch.Generator().manual_seed(43)
    img = torch.randn(3, 64, 80, generator=g)
    field = torch.randn(2, 3, 3, 2, generator=g) * 2.5
    got = mc.correct_motion_slow(img.to(dev), field.to(dev))
    ref = oracle.correct_motion_slow(img, field)
    grid = tp.coordinate_grid((64, 80))
    norm = grid / torch.tensor([63.0, 79.0])
    knife = torch.zeros(3, 64, 80, dtype=torch.bool)
    for i, ft in enumerate(torch.linspace(0, 1, steps=3)):
        tyx = torch.nn.functional.pad(norm, (1, 0), value=float(ft))
        knife[i] = _coord_knife(grid + oracle.evaluate_deformation_field(field, tyx), 64, 80)
    assert_frames_close(got, ref, knife, max_excluded=0.05)
    with pytest.raises(NotImplementedError):
        mc.correct_motion_slow(img.to(dev), field.to(dev), grad=True)


def test_local_motion_1024_patches_take_the_wave_engine(dev):
    """p = 1024 patches run the wavefront-per-row / per-column transforms (xc_rows_fwd_wave512,
    xc_cols_fwd_wave1024); loss and gradient against the oracle's autograd on 2 x 2 patches."""
    from torch_motion_correction_amd import local_motion

    st, _, _ = drift_stack(3, 2048, 2048, seed=77)
    res, patch = (3, 2, 2), (1024, 1024)
    g = torch.Generator().manual_seed(5)
    new = (torch.randn(2, *res, generator=g) * 1.0).requires_grad_(True)
    init = torch.zeros(2, *res)
    oprob = oracle.LocalMotionProblem(st, 1.0, patch)
    total = oprob.batch_loss(new, init, "catmull_rom", list(range(oprob.npatch)), "mse")
    total.backward()
    prob = local_motion.LocalMotionProblem(st.to(dev), 1.0, patch, res, "catmull_rom")
    assert prob.npatch == oprob.npatch == 4
    wb = torch.full((4,), 0.25, dtype=torch.float64, device=dev)
    nd = new.detach().to(dev).requires_grad_(True)
    loss = local_motion._Loss.apply(prob.shifts_px(nd, init.to(dev)), prob, wb, "mse")
    loss.backward()
    assert abs(loss.item() - total.item()) <= 2e-4 * abs(total.item())
    assert float((nd.grad.cpu() - new.grad).abs().max() / new.grad.abs().max()) <= 2e-4


@pytest.mark.parametrize("shape", [(3, 64, 128), (2, 100, 132), (2, 96, 120), (2, 33, 72)])
def test_polyphase_fourier_shift_matches_direct_path_and_oracle(mc, dev, shape):
    """correct_motion_fast through the x-polyphase form (even / odd columns transformed separately,
    csrc/polyphase.hip) -- the path frames wider than ~8190 columns take -- forced on small frames."""
    from torch_motion_correction_amd import engine

    g = torch.Generator().manual_seed(sum(shape))
    img = torch.randn(*shape, generator=g)
    sh = torch.randn(2, shape[0], 1, 1, generator=g) * 3
    ref = oracle.correct_motion_fast(img, sh.clone())
    direct = mc.correct_motion_fast(img.to(dev), sh.clone().to(dev))
    engine.POLYPHASE_FOURIER_SHIFT = True
    try:
        poly = mc.correct_motion_fast(img.to(dev), sh.clone().to(dev))
    finally:
        engine.POLYPHASE_FOURIER_SHIFT = False
    assert rel_err(poly, ref) <= 2e-5 and rel_err(poly, direct) <= 2e-5


@pytest.mark.parametrize("shape,ps,dose", [((5, 64, 128), 1.0, 1.5), ((4, 90, 132), 0.83, 0.8), ((3, 33, 72), 1.3, 2.0)])
def test_polyphase_dose_weighted_sum_matches_direct_path_and_oracle(mc, dev, shape, ps, dose):
    """dose_weighted_sum through the x-polyphase form (the path of frames wider than ~8190 columns),
    forced on small frames: equal to the direct path and to the oracle."""
    from torch_motion_correction_amd import engine

    g = torch.Generator().manual_seed(sum(shape))
    m = torch.randn(*shape, generator=g) * 2.0 + 5.0
    ref = oracle.dose_weighted_sum(m, ps, dose, pre_exposure=0.4, voltage=300.0)
    direct = mc.dose_weighted_sum(m.to(dev), ps, dose, pre_exposure=0.4).cpu()
    engine.POLYPHASE_FOURIER_SHIFT = True
    try:
        poly = mc.dose_weighted_sum(m.to(dev), ps, dose, pre_exposure=0.4).cpu()
    finally:
        engine.POLYPHASE_FOURIER_SHIFT = False
    scale = float(ref.abs().max())
    assert float((poly - ref).abs().max()) <= 1e-4 * scale and float((poly - direct).abs().max()) <= 2e-5 * scale


def test_example_movie_shape_runs_the_reference_flow(mc, dev):
    """The reference's example movie is 40 x 959 x 927 (examples/example.ipynb: odd height and odd
    width).  Its flow -- global estimate, patch estimate with the global field as prior (which
    pre-corrects with correct_motion_fast), correct_motion -- on a stack of that frame size; the
    global shifts against the oracle."""
    st, dy, dx = drift_stack(8, 959, 927, seed=6)
    d = st.to(dev)
    glob = mc.estimate_global_motion(d, 1.35)
    ref, ccs = oracle.estimate_global_motion(st, 1.35, return_cc=True)
    for f, cc in ccs.items():
        top = torch.topk(cc.flatten(), 2).values
        if float(top[0] - top[1]) > 1e-5 * float(top[0].abs()):
            assert torch.equal(glob.cpu()[:, f], ref[:, f]), f
    field, centres = mc.estimate_motion_cross_correlation_patches(d, 1.35, patch_sidelength=256, deformation_field=glob)
    assert field.shape[:2] == (2, 8) and centres.shape[-1] == 3 and torch.isfinite(field).all()
    out = mc.correct_motion(d, field, 1.35)
    assert out.shape == st.shape and torch.isfinite(out).all()


@pytest.mark.parametrize("p,strategy", [(48, "mean_except_current"), (63, "mean_except_current"), (80, "middle_frame")])
def test_patches_with_patch_sizes_that_are_not_powers_of_two(mc, dev, p, strategy):
    """Patch transforms by chirp-z (even and odd side lengths) with the sub-pixel refinement: the
    3 x 3 neighbourhood of every peak comes from direct sums over the kept columns
    (mc_xcg_peak_neighbourhood).  Field within 1e-4 px of the oracle, same patch centres."""
    st, _, _ = drift_stack(5, 170, 190, seed=3)
    got, gc = mc.estimate_motion_cross_correlation_patches(st.to(dev), 1.0, patch_sidelength=p,
                                                           reference_strategy=strategy)
    ref, rc = oracle.estimate_motion_cross_correlation_patches(st, 1.0, patch_sidelength=p,
                                                               reference_strategy=strategy)
    assert torch.equal(gc.cpu(), rc) and got.shape == ref.shape
    assert float((got.cpu() - ref).abs().max()) <= 1e-4


# ------------------------------------------------------------------ round 2: parity holes, untested configs


@pytest.fixture(scope="module")
def refb():
    """tests/golden/reference_bodies_with_standins.npz: the reference's own function bodies executed
    with the absent third-party names bound to oracle.thirdparty_semantics (pins the control flow,
    not the third-party semantics; oracle/make_goldens.py:reference_body_vectors)."""
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden",
                        "reference_bodies_with_standins.npz")
    return {k: torch.from_numpy(v) for k, v in np.load(path).items()}


def test_even_effective_smoothing_window(mc, dev, refb):
    """t = 4 with the default window 5 -> min(5, 4) = 4, an EVEN Savitzky-Golay window, which scipy
    accepts (estimate_motion_xc.py:506-529); the product used to raise here."""
    st, _, _ = drift_stack(8, 256, 256)
    got, _ = mc.estimate_motion_cross_correlation_patches(st[:4].to(dev), 1.0, patch_sidelength=64)
    assert float((got.cpu() - refb["drift_t4"]).abs().max()) <= REL
    got, _ = mc.estimate_motion_cross_correlation_patches(st[:6].to(dev), 1.0, patch_sidelength=64,
                                                          smoothing_window_size=7)
    assert float((got.cpu() - refb["drift_t6_w7"]).abs().max()) <= REL


def test_field_smooth_any_window_against_scipy(dev):
    """mc_field_smooth_center for every window 3..t, odd and even, against scipy itself."""
    from scipy.signal import savgol_filter
    from torch_motion_correction_amd import _lib

    lib = _lib.load()
    g = torch.Generator().manual_seed(4)
    for t in (4, 6, 9, 10):
        fld = torch.randn(2, t, 3, 2, generator=g)
        for window in range(3, t + 1):
            out = torch.empty_like(fld, device=dev)
            d = fld.to(dev)
            _lib.check(lib.mc_field_smooth_center(_lib.ptr(d), _lib.ptr(out), t, 6, window, 0,
                                                  _lib.stream_ptr(dev)), "mc_field_smooth_center")
            ref = torch.from_numpy(savgol_filter(fld.numpy(), window, 1, axis=1))
            assert float((out.cpu() - ref).abs().max()) <= 2e-6, (t, window)


def test_reference_frame_follows_python_indexing(mc, dev, refb):
    """reference_frame is a Python index in the reference (xc.py:101,306): negative values select
    from the end and are never 'the current frame' (nothing is skipped); outside [-t, t) raises
    IndexError.  No raw value reaches a device-side table."""
    mov = blob_stack(True).to(dev)
    assert torch.equal(mc.estimate_global_motion(mov, 1.0, reference_frame=-1).cpu(), refb["blob_global_refm1"])
    assert torch.equal(mc.estimate_global_motion(mov, 1.0, reference_frame=0).cpu(), refb["blob_global_ref0"])
    for bad in (5, 17, -6):
        with pytest.raises(IndexError):
            mc.estimate_global_motion(mov, 1.0, reference_frame=bad)
        with pytest.raises(IndexError):
            mc.estimate_motion_cross_correlation_patches(mov, 1.0, patch_sidelength=32,
                                                         reference_strategy="middle_frame", reference_frame=bad)
    # mean_except_current never reads reference_frame (xc.py:310-328): anything goes, as in the reference
    mc.estimate_motion_cross_correlation_patches(mov, 1.0, patch_sidelength=32, reference_frame=99)
    st, _, _ = drift_stack(8, 256, 256)
    for i, kw in ((1, {"reference_frame": 1}), (2, {"reference_frame": -1})):
        got, _ = mc.estimate_motion_cross_correlation_patches(st.to(dev), 1.3, patch_sidelength=64,
                                                              reference_strategy="middle_frame", **kw)
        assert float((got.cpu() - refb[f"drift_opt{i}"]).abs().max()) <= REL, kw
    with pytest.raises(IndexError):
        mc.MoviePipeline(dev, reference_frame=40).run([st.to(dev)])


def test_pixel_shifts_honour_the_pixel_grid(mc, dev):
    """get_pixel_shifts evaluates at `pixel_grid` (correct_motion.py:167-168): a fractional,
    partly out-of-frame sub-grid against the reference's own output."""
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "reference_helpers.npz")
    ref = {k: torch.from_numpy(v) for k, v in np.load(path, allow_pickle=False).items() if v.ndim}
    got = mc.get_pixel_shifts(torch.zeros(37, 53, device=dev), 1.3, ref["gps_lattice"].to(dev),
                              ref["gps_sub_grid"].to(dev)).cpu()
    assert got.shape == ref["gps_sub_out"].shape
    assert float((got - ref["gps_sub_out"]).abs().max()) <= 1e-5 * float(ref["gps_sub_out"].abs().max())
    with pytest.raises(ValueError):
        mc.get_pixel_shifts(torch.zeros(37, 53, device=dev), 1.3, ref["gps_lattice"].to(dev),
                            torch.zeros(4, 3, device=dev))


def local_motion_stack(mc, dev, t, h, w, gh, gw, amp, seed, noise=0.5, dtype=torch.float32):
    """A texture seen through a smooth, small (|shift| <= amp px) local deformation that varies in
    time -- the input the patch estimator is made for.  Built on the GPU with the
    product's own warp; what is compared afterwards is product vs oracle on the SAME input."""
    g = torch.Generator(device=dev).manual_seed(seed)
    base = torch.randn(h, w, generator=g, device=dev)
    base = (base + torch.roll(base, 1, 0) + torch.roll(base, 1, 1) + torch.roll(base, (1, 1), (0, 1))) / 2
    tt = torch.linspace(-1, 1, t)[:, None, None]
    yy = torch.linspace(-1, 1, gh)[None, :, None]
    xx = torch.linspace(-1, 1, gw)[None, None, :]
    true = torch.stack([amp * tt * torch.sin(2.0 * yy + 1.0 * xx), amp * tt * torch.cos(1.5 * xx - yy)])
    frames = torch.empty((t, h, w), dtype=dtype, device=dev)
    for f in range(t):
        one = mc.correct_motion(base[None], -true[:, f:f + 1].to(dev), 1.0, grid_type="bspline")[0]
        frames[f] = (one + noise * torch.randn(h, w, generator=g, device=dev)).to(dtype)
    return frames, true


def test_c5_flow_small(mc, dev):
    """BASELINE C5 as one flow at reduced frame size: an fp16 stack of t = 60 frames (memo eviction,
    Q3) -> estimate_motion_cross_correlation_patches -> motion_correct_sum(bspline, dose-weighted),
    against the oracle on the fp32 up-cast of the same fp16 data (the reference cannot
    run Half on the CPU at all)."""
    t, h, w, p = 60, 264, 372, 64
    st16, _ = local_motion_stack(mc, dev, t, h, w, 3, 4, 1.5, seed=5, dtype=torch.float16)
    up = st16.float().cpu()
    field, pos = mc.estimate_motion_cross_correlation_patches(st16, 0.9, patch_sidelength=p)
    ofield, opos = oracle.estimate_motion_cross_correlation_patches(up, 0.9, patch_sidelength=p)
    assert torch.equal(pos.cpu(), opos) and field.dtype == torch.float32
    assert float((field.cpu() - ofield).abs().max()) <= REL
    total = mc.motion_correct_sum(st16, field, 0.9, grid_type="bspline", dose_per_frame=0.8,
                                  pre_exposure=0.2).cpu()
    oframes = oracle.correct_motion(up, ofield, 0.9, grid_type="bspline")
    ref = oracle.dose_weighted_sum(oframes, 0.9, 0.8, pre_exposure=0.2)
    knife = knife_edge_mask(up, ofield, 0.9, "bspline").any(0)
    # the exposure filter spreads every pixel over the frame: compare where no knife-edge pixel
    # contributes at full weight, i.e. everything but a small share of the border pixels
    assert float(knife.float().mean()) <= 0.02
    d = (total - ref).abs()
    assert float(d[~knife].max()) <= 3 * REL * float(ref.abs().max())
    # and the plain fused sum of the fp16 stack
    plain = mc.motion_correct_sum(st16, field, 0.9, grid_type="bspline").cpu()
    d = (plain - oframes.sum(0)).abs()
    d[knife] = 0
    assert float(d.max()) <= 2 * REL * float(oframes.sum(0).abs().max())


def test_c3_full_frame_size_against_oracle(mc, dev):
    """BASELINE C3 at its real frame size 4092 x 5760 (6 x 10 patches of 1024 px) with 6 frames:
    patch estimate (wave kernels, near-window search) and B-spline warp + sum against the oracle."""
    t, h, w = 6, 4092, 5760
    st, true = local_motion_stack(mc, dev, t, h, w, 6, 10, 2.0, seed=3)
    cpu = st.cpu()
    field, pos = mc.estimate_motion_cross_correlation_patches(st, 1.0, patch_sidelength=1024)
    ofield, opos = oracle.estimate_motion_cross_correlation_patches(cpu, 1.0, patch_sidelength=1024)
    assert tuple(field.shape) == (2, t, 6, 10) and torch.equal(pos.cpu(), opos)
    assert float((field.cpu() - ofield).abs().max()) <= REL
    total, frames = mc.motion_correct_sum(st, field, 1.0, grid_type="bspline", return_frames=True)
    oframes = oracle.correct_motion(cpu, ofield, 1.0, grid_type="bspline")
    knife = knife_edge_mask(cpu, ofield, 1.0, "bspline")
    assert_frames_close_large(frames, oframes, knife)
    osum = oframes.sum(0)
    d = (total.cpu() - osum).abs()
    d[knife.any(0)] = 0
    # the sum adds t frames' coordinate-rounding flips: all but a small share within 2 REL
    assert float((d > 2 * REL * float(osum.abs().max())).float().mean()) <= 2e-2
    assert float(d.max()) <= 20 * REL * float(osum.abs().max())
    # aligned frames add coherently: the sum's spread is close to t x the texture's
    assert float(total.std()) > 0.8 * t * float(st[0].std()) * 0.7


@pytest.mark.parametrize("t,h,w", [(6, 4092, 5760), (3, 8184, 11520)])
def test_k3_formats_global_estimate_known_drift(mc, dev, t, h, w):
    """estimate_global_motion on the K3 detector's two frame formats (BASELINE C3 / C5 frame sizes): the
    shifts equal the drift the stack was built with, with the direct mixed-radix lines (rows of
    2880 / 5760 points: radix 8 8 9 5 / 8 8 9 10; columns of 4092 / 8184 points: radix 31 11 12 / 31 11 24)
    and with the chirp-z lines they replace; zero-dose exposure weighting of the same frames is the
    plain sum / sqrt(t) (row-major mixed-radix full spectrum against a direct sum)."""
    import bench
    from torch_motion_correction_amd import plan

    st, dy, dx = bench.synth_stack(t, h, w, 77, dev)
    want_y = [float(d - dy[t // 2]) for d in dy]
    want_x = [float(d - dx[t // 2]) for d in dx]
    f = mc.estimate_global_motion(st, 1.0).cpu()
    assert f[0, :, 0, 0].tolist() == want_y and f[1, :, 0, 0].tolist() == want_x
    try:
        plan.USE_DIRECT_LINES = False
        plan._LINES.clear()
        fc = mc.estimate_global_motion(st, 1.0).cpu()
    finally:
        plan.USE_DIRECT_LINES = True
        plan._LINES.clear()
    assert torch.equal(fc, f)
    z = mc.dose_weighted_sum(st, 1.0, 0.0)
    plain = st.sum(0) / t**0.5
    assert float((z - plain).abs().max()) <= 1e-4 * float(plain.abs().max())


def torch_gpu_correct_frame(frame, lattice, pixel_spacing):
    """The reference's _correct_frame op sequence (correct_motion.py:81-185) executed by torch's OWN
    ROCm operators on the GPU: an independent fp32 implementation for sizes the CPU oracle does not
    finish in seconds.  frame (h,w) cuda, lattice (2,GH,GW) cuda -> (h,w)."""
    import torch.nn.functional as F

    h, w = frame.shape
    _, GH, GW = lattice.shape
    dev_ = frame.device
    yy, xx = torch.meshgrid(torch.arange(h, dtype=torch.float32, device=dev_),
                            torch.arange(w, dtype=torch.float32, device=dev_), indexing="ij")
    grid = torch.stack([yy, xx], dim=-1)
    interp = (grid / torch.tensor([h - 1, w - 1], dtype=torch.float32, device=dev_)) * torch.tensor(
        [GH - 1, GW - 1], dtype=torch.float32, device=dev_)
    shape = torch.tensor([GH, GW], dtype=torch.float32, device=dev_)
    gs = torch.flip(interp / (0.5 * shape - 0.5) - 1, dims=(-1,))
    shifts = F.grid_sample(lattice[None], gs[None], mode="bicubic", padding_mode="reflection",
                           align_corners=True)[0].permute(1, 2, 0) / pixel_spacing
    del interp, gs
    coords = grid + shifts
    shape = torch.tensor([h, w], dtype=torch.float32, device=dev_)
    gs = torch.flip(coords / (0.5 * shape - 0.5) - 1, dims=(-1,))
    out = F.grid_sample(frame[None, None], gs[None], mode="bicubic", padding_mode="border",
                        align_corners=True)[0, 0]
    hi = torch.tensor([h - 1, w - 1], dtype=torch.float32, device=dev_)
    inside = ((coords >= 0) & (coords <= hi)).all(dim=-1)
    near = ((coords.abs() < 1e-3) | ((coords - hi).abs() < 1e-3)).any(dim=-1)
    return torch.where(inside, out, torch.zeros_like(out)), near


def test_c5_full_frame_size_properties(mc, dev):
    """BASELINE C5's frame size 8184 x 11520 (14 x 21 patches of 1024 px), fp16 storage, 3 frames.
    (1) the patch field against the oracle (three frames are what it finishes in tens of seconds);
    the warp is too large for the CPU oracle in seconds, so size-independent properties --
    (2) corrected frames equal torch's own ROCm grid_sample composition of the same op sequence;
    (3) the fused sum equals the sum of the frames; (4) dose weighting with zero dose is the
    plain sum / sqrt(t); (5) linearity of the warp in the frames."""
    t, h, w = 3, 8184, 11520
    st16, true = local_motion_stack(mc, dev, t, h, w, 14, 21, 2.0, seed=13, dtype=torch.float16)
    field, pos = mc.estimate_motion_cross_correlation_patches(st16, 1.0, patch_sidelength=1024,
                                                              temporal_smoothing=False)
    assert tuple(field.shape) == (2, t, 14, 21) and tuple(pos.shape) == (t, 14, 21, 3)
    ofield, opos = oracle.estimate_motion_cross_correlation_patches(st16.float().cpu(), 1.0, patch_sidelength=1024,
                                                                    temporal_smoothing=False)
    assert torch.equal(pos.cpu(), opos)
    assert float((field.cpu() - ofield).abs().max()) <= REL
    del ofield, opos
    total, frames = mc.motion_correct_sum(st16, field, 1.0, grid_type="bspline", return_frames=True)
    assert float((total - frames.sum(0)).abs().max()) <= 1e-5 * float(total.abs().max())
    from torch_motion_correction_amd import engine

    lat = engine.frame_lattices(field.contiguous(), t, "bspline")
    ref0, near = torch_gpu_correct_frame(st16[0].float(), lat[0], 1.0)
    assert_frames_close_large(frames[0], ref0, near.cpu())
    del ref0, near
    z = mc.motion_correct_sum(st16, field, 1.0, grid_type="bspline", dose_per_frame=0.0)
    assert float((z - total / t**0.5).abs().max()) <= 2e-4 * float(z.abs().max())
    del z
    twice = mc.correct_motion(2.0 * st16.float() + 1.0, field, 1.0, grid_type="bspline")
    inside = frames != 0  # zero-outside pixels stay zero instead of picking up the offset
    lin = (twice - (2.0 * frames + 1.0)).abs()
    assert float(lin[inside].max()) <= 1e-5 * float(frames.abs().max())


# ------------------------------------------------------------------ fp16 frames read natively (N2 / C5)


def test_fp16_statistics_and_patch_rows_read_natively(mc, dev):
    """mc_central_box_stats_t and the 1024-px patch row kernel on fp16 bytes == the same on the fp32
    up-cast of those bytes (bit for bit: the widening is exact and happens in the load)."""
    from torch_motion_correction_amd import engine

    st, _, _ = drift_stack(3, 1100, 1600, seed=5)
    h16 = st.half().to(dev)
    up = h16.float()
    assert torch.equal(engine.central_box_stats(h16), engine.central_box_stats(up))
    a, pa = mc.estimate_motion_cross_correlation_patches(h16, 1.0, patch_sidelength=1024)
    b, pb = mc.estimate_motion_cross_correlation_patches(up, 1.0, patch_sidelength=1024)
    assert torch.equal(pa, pb) and torch.equal(a, b)
    # a patch size the native kernel does not cover: widened internally, same answer
    a, _ = mc.estimate_motion_cross_correlation_patches(h16, 1.0, patch_sidelength=256)
    b, _ = mc.estimate_motion_cross_correlation_patches(up, 1.0, patch_sidelength=256)
    assert torch.equal(a, b)


@pytest.mark.parametrize("shape,grid,ps", [((4, 300, 520), (3, 4), 1.0), ((3, 200, 264), (2, 2), 0.83),
                                           ((3, 130, 96), (2, 3), 1.0), ((2, 100, 101), (2, 2), 1.0)])
def test_fp16_frames_through_the_field_warp(mc, dev, shape, grid, ps):
    """The deformation-field warp on fp16 frames (window DMA'd as fp16, widened in LDS; borders,
    edge tiles, non-unit spacing; a row length that is not a multiple of 8 falls back to a widened
    copy) == the warp of the fp32 up-cast, frames and sum, bit for bit."""
    t, h, w = shape
    g = torch.Generator().manual_seed(h + w)
    st16 = (torch.randn(t, h, w, generator=g) * 2 + 1).half().to(dev)
    field = (torch.randn(2, t, *grid, generator=g) * 1.5).to(dev)
    for gt in ("bspline", "catmull_rom"):
        sa, fa = mc.motion_correct_sum(st16, field, ps, grid_type=gt, return_frames=True)
        sb, fb = mc.motion_correct_sum(st16.float(), field, ps, grid_type=gt, return_frames=True)
        assert fa.dtype == torch.float32 and torch.equal(fa, fb) and torch.equal(sa, sb)
    # rough field: some tile-frames fail the regularity test and take the generic kernel
    rough = (torch.randn(2, t, 6, 6, generator=g) * 6).to(dev)
    fa = mc.correct_motion(st16, rough, ps, grid_type="bspline")
    fb = mc.correct_motion(st16.float(), rough, ps, grid_type="bspline")
    assert torch.equal(fa, fb)
    # and against the oracle on the up-cast
    ref = oracle.correct_motion(st16.float().cpu(), field.cpu(), ps, grid_type="bspline")
    got = mc.correct_motion(st16, field, ps, grid_type="bspline")
    knife = knife_edge_mask(st16.float().cpu(), field.cpu(), ps, "bspline")
    assert_frames_close(got, ref, knife, max_excluded=0.05)


def test_field_warp_with_more_frames_than_one_plan_block(mc, dev):
    """130 frames: the tile kernel keeps its per-frame plan entries in LDS 128 frames at a time
    (warp.hip, warp_field3) -- frames 128, 129 come from the second block.  fp32 and fp16 frames,
    frames and sum, against the oracle (correct_motion.py:81-185)."""
    t, h, w = 130, 416, 512
    g = torch.Generator().manual_seed(130)
    st = torch.randn(t, h, w, generator=g)
    field = torch.randn(2, 5, 2, 2, generator=g) * 2
    ref = oracle.correct_motion(st, field, 1.0, grid_type="bspline")
    total, frames = mc.motion_correct_sum(st.to(dev), field.to(dev), 1.0, grid_type="bspline", return_frames=True)
    knife = knife_edge_mask(st, field, 1.0, "bspline")
    assert_frames_close(frames, ref, knife, max_excluded=0.05)
    assert rel_err(total, frames.sum(0)) <= 1e-5
    only_sum = mc.motion_correct_sum(st.to(dev), field.to(dev), 1.0, grid_type="bspline")
    assert torch.equal(only_sum, total)
    s16, f16 = mc.motion_correct_sum(st.half().to(dev), field.to(dev), 1.0, grid_type="bspline", return_frames=True)
    s32, f32 = mc.motion_correct_sum(st.half().float().to(dev), field.to(dev), 1.0, grid_type="bspline",
                                     return_frames=True)
    assert torch.equal(f16, f32) and torch.equal(s16, s32)


# ------------------------------------------------------------------ N2: hot pixels, scattered spline points


def _hot_pixel_reference(x, thr):
    """numpy restatement of the example's remove_hot_pixels DETECTION (examples/ttMotion.py:145-153)
    and of this package's deterministic replacement rule; x (t,h,w) float64 = raw * gain."""
    out = x.copy()
    counts = []
    t, h, w = x.shape
    for f in range(t):
        fr = x[f]
        m, sd = fr.mean(), fr.std()
        hot = (fr > m + thr * sd) | (fr < m - thr * sd)
        counts.append(int(hot.sum()))
        for y, xx in zip(*np.where(hot)):
            vals = [fr[yy, xc] for yy in range(max(0, y - 1), min(h - 1, y + 1) + 1)
                    for xc in range(max(0, xx - 1), min(w - 1, xx + 1) + 1)
                    if (yy != y or xc != xx) and not hot[yy, xc]]
            out[f, y, xx] = np.mean(vals) if vals else m
    return out, counts


@pytest.mark.parametrize("dtype", [torch.uint8, torch.int16, torch.float16, torch.float32])
def test_condition_movie_hot_pixels(mc, dev, dtype):
    """gain -> hot pixels -> mean-zero (examples/ttMotion.py:90-199): the example's detection rule
    exactly (count and positions), our deterministic replacement, the mean taken after it."""
    g = torch.Generator().manual_seed(12)
    t, h, w = 3, 96, 120
    raw = (torch.rand(t, h, w, generator=g) * 20 + 20)
    low = 250.0 if dtype == torch.uint8 else -200.0  # unsigned storage has no low outliers
    for f, (y, x, v) in enumerate([(0, 0, 250.0), (50, 60, 240.0), (95, 119, low)]):
        raw[f, y, x] = v          # corners and interior, above and below
    raw[1, 50, 61] = 245.0         # two adjacent hot pixels: neither is the other's replacement
    raw = raw.to(dtype)
    gain = torch.rand(h, w, generator=g) * 0.2 + 0.9
    got, counts = mc.condition_movie(raw.to(dev), gain.to(dev), hot_pixel_threshold=10.0, return_hot_counts=True)
    x = raw.float().numpy().astype(np.float64) * gain.numpy().astype(np.float64)
    fixed, ref_counts = _hot_pixel_reference(x, 10.0)
    assert counts.cpu().tolist() == ref_counts and sum(ref_counts) == 4
    ref = torch.from_numpy((fixed - fixed.mean(axis=(1, 2), keepdims=True)).astype(np.float32))
    assert float((got.cpu() - ref).abs().max()) <= 2e-5 * float(ref.abs().max())
    assert float(got.mean(dim=(1, 2)).abs().max()) < 1e-4
    # no hot pixel at a huge threshold: identical to the plain conditioning
    a = mc.condition_movie(raw.to(dev), gain.to(dev), hot_pixel_threshold=1e6)
    b = mc.condition_movie(raw.to(dev), gain.to(dev))
    assert float((a - b).abs().max()) <= 2e-5 * float(b.abs().max())
    # without mean-zero the untouched pixels are raw * gain exactly
    c = mc.condition_movie(raw.to(dev), gain.to(dev), mean_zero=False, hot_pixel_threshold=10.0).cpu()
    keep = torch.from_numpy(fixed == x)
    assert torch.equal(c[keep], (raw.float() * gain)[keep])


def test_evaluate_deformation_field_on_scattered_points(mc, dev):
    """one launch for any set of points (the round-1 version launched once per distinct (t, y)):
    both bases, points on the domain's edges, a leading batch shape, size-1 axes."""
    g = torch.Generator().manual_seed(6)
    for shape in ((2, 5, 4, 6), (2, 1, 3, 1), (3, 4, 1, 1)):
        field = torch.randn(*shape, generator=g)
        tyx = torch.rand(7, 11, 3, generator=g)
        tyx[0, 0] = 0.0
        tyx[0, 1] = 1.0
        tyx[0, 2] = torch.tensor([0.5, 0.0, 1.0])
        for gt in ("catmull_rom", "bspline"):
            got = mc.evaluate_deformation_field(field.to(dev), tyx.to(dev), gt).cpu()
            ref = oracle.evaluate_deformation_field(field, tyx, gt)
            assert got.shape == (7, 11, shape[0])
            assert float((got - ref).abs().max()) <= 2e-5 * max(1.0, float(ref.abs().max()))


# ------------------------------------------------------------------ row-major full-spectrum kernels (full_fft.hip)


@pytest.mark.parametrize("shape", [(3, 256, 64), (2, 256, 256), (2, 512, 1024), (3, 1024, 512), (2, 2048, 256),
                                   (2, 4096, 128), (1, 256, 8192), (2, 4096, 4096),
                                   # K3 formats: 5760 / 11520 columns (2^a 3^2 5), 4092 / 8184 rows (2^a 3 11 31:
                                   # radix-31 and radix-11 passes), alone and together
                                   (2, 256, 5760), (1, 512, 11520), (3, 4092, 64), (2, 8184, 128),
                                   (2, 4092, 5760), (1, 8184, 11520)])
def test_row_major_fourier_shift(mc, dev, shape):
    """correct_motion_fast on power-of-two frames (rows forward, one in-place column kernel for
    forward + phase ramp + inverse, rows inverse; spectrum row-major) against the oracle, against
    the pruned engine's transposed layout, and -- for integer shifts -- against an exact roll."""
    from torch_motion_correction_amd import engine

    t, h, w = shape
    g = torch.Generator().manual_seed(h + w)
    img = torch.randn(t, h, w, generator=g)
    sh = torch.randn(2, t, 1, 1, generator=g) * 3
    got = mc.correct_motion_fast(img.to(dev), sh.clone().to(dev)).cpu()
    if h * w <= 1024 * 1024:
        assert rel_err(got, oracle.correct_motion_fast(img, sh.clone())) <= 2e-5
    try:
        engine.FULL_ROW_MAJOR = False
        old = mc.correct_motion_fast(img.to(dev), sh.clone().to(dev)).cpu()
    finally:
        engine.FULL_ROW_MAJOR = True
    assert rel_err(got, old) <= 2e-5
    ish = torch.tensor([[3.0, -7.0]] * t).t()[:, :, None, None].contiguous()  # field +3 / -7 -> shift by (-3, +7)
    rolled = mc.correct_motion_fast(img.to(dev), ish.clone().to(dev)).cpu()
    assert rel_err(rolled, torch.roll(img, shifts=(-3, 7), dims=(1, 2))) <= 1e-5


@pytest.mark.parametrize("shape,ps,dose,pre,kv", [((6, 256, 256), 1.0, 1.5, 0.0, 300.0),
                                                  ((5, 512, 256), 1.3, 0.8, 2.0, 200.0),
                                                  ((3, 256, 1024), 0.83, 2.5, 0.5, 100.0),
                                                  ((5, 4092, 64), 1.0, 1.2, 0.0, 300.0),
                                                  ((5, 4096, 128), 0.9, 1.1, 0.5, 300.0),
                                                  ((4, 8184, 128), 0.5, 0.9, 1.0, 300.0),
                                                  ((3, 256, 5760), 1.1, 1.0, 0.0, 200.0)])
def test_row_major_dose_weighted_sum(mc, dev, shape, ps, dose, pre, kv):
    """The exposure-filtered sum with the weighted accumulation inside the forward column pass
    (frame loop in registers, chunks of frames carried through A) against the oracle and against
    the separate accumulate kernel on the transposed layout."""
    from torch_motion_correction_amd import engine

    g = torch.Generator().manual_seed(sum(shape))
    m = torch.randn(*shape, generator=g) * 2.0 + 5.0
    got = mc.dose_weighted_sum(m.to(dev), ps, dose, pre_exposure=pre, voltage=kv).cpu()
    ref = oracle.dose_weighted_sum(m, ps, dose, pre_exposure=pre, voltage=kv)
    assert float((got - ref).abs().max()) <= 1e-4 * float(ref.abs().max())
    try:  # two frames per chunk: the accumulator goes through A between chunks
        ws, engine.WORKSPACE_BYTES = engine.WORKSPACE_BYTES, 2 * shape[1] * (shape[2] // 2 + 16) * 8
        chunked = mc.dose_weighted_sum(m.to(dev), ps, dose, pre_exposure=pre, voltage=kv).cpu()
    finally:
        engine.WORKSPACE_BYTES = ws
    assert float((chunked - ref).abs().max()) <= 1e-4 * float(ref.abs().max())
    try:
        engine.FULL_ROW_MAJOR = False
        old = mc.dose_weighted_sum(m.to(dev), ps, dose, pre_exposure=pre, voltage=kv).cpu()
    finally:
        engine.FULL_ROW_MAJOR = True
    assert float((got - old).abs().max()) <= 2e-5 * float(ref.abs().max())
    try:  # 4096 / 4092 rows: the column pass fed from the column-major copy (default) == fed from the rows
        engine.DOSE_COLUMN_MAJOR = False
        rowfed = mc.dose_weighted_sum(m.to(dev), ps, dose, pre_exposure=pre, voltage=kv).cpu()
    finally:
        engine.DOSE_COLUMN_MAJOR = True
    assert float((got - rowfed).abs().max()) <= 2e-6 * float(ref.abs().max())


@pytest.mark.skipif(torch.cuda.device_count() < 2, reason="needs two GPUs (the test box has one: never run on hardware so far)")
def test_entry_points_run_on_a_gpu_that_is_not_the_current_one(mc):
    """libmcorr launches on the CURRENT HIP device; every API entry point switches to the device its
    tensors live on (_lib.device_scope).  With cuda:0 current, the estimators and the corrector must work
    on cuda:1 and agree with the same calls on cuda:0 (this contract was only exercised on a
    one-GPU box)."""
    stack, dy, dx = drift_stack(6, 256, 256, seed=3)
    torch.cuda.set_device(0)
    res = []
    for d in ("cuda:0", "cuda:1"):
        s = stack.to(d)
        f = mc.estimate_global_motion(s, 1.0)
        c = mc.correct_motion(s, f, 1.0)
        pf, _ = mc.estimate_motion_cross_correlation_patches(s, 1.0, patch_sidelength=128)
        lf = mc.estimate_local_motion(s, 1.0, (128, 128), (6, 2, 2), n_iterations=3)
        assert f.device == s.device and c.device == s.device and pf.device == s.device and lf.device == s.device
        res.append((f.cpu(), c.cpu(), pf.cpu(), lf.cpu()))
    assert torch.cuda.current_device() == 0
    for a, b in zip(*res):
        assert torch.equal(a, b)


# ------------------------------------------------------------------ N2: conditioning fused into the first read


def _raw_drift_movie(t, h, w, dtype, seed, amp, pad=64):
    """raw detector-like counts of one texture at integer drift offsets + noise, and a gain reference"""
    g = torch.Generator().manual_seed(seed)
    base = torch.rand(h + 2 * pad, w + 2 * pad, generator=g) * 40 + 10
    dy = torch.round(torch.linspace(-amp, amp + 2, t)).long().tolist()
    dx = torch.round(torch.linspace(amp - 1, -amp, t)).long().tolist()
    raw = torch.empty((t, h, w), dtype=dtype)
    for f in range(t):
        v = base[pad - dy[f]: pad - dy[f] + h, pad - dx[f]: pad - dx[f] + w] + 6 * torch.randn(h, w, generator=g)
        if dtype == torch.int16:
            raw[f] = (v * 8 - 100).round().clamp(-32768, 32767).to(dtype)
        else:
            raw[f] = v.round().clamp(0, 255).to(dtype)
    gain = (1.0 + 0.1 * torch.randn(h, w, generator=g)).clamp(0.5, 1.5)
    return raw, gain, dy, dx


def _numpy_condition(raw, gain):
    """examples/ttMotion.py:90-121 (movie * gain_map) and :180-199 (minus the frame mean), in float64"""
    x = raw.numpy().astype(np.float64) * gain.numpy().astype(np.float64)
    return torch.from_numpy((x - x.mean(axis=(1, 2), keepdims=True)).astype(np.float32))


@pytest.mark.parametrize("dtype,amp", [(torch.uint8, 5), (torch.int16, 5), (torch.uint8, 30)])
def test_fused_raw_path_equals_conditioning_then_the_fp32_path_and_the_oracle(mc, dev, dtype, amp):
    """N2: raw u8 / i16 movie + gain -> shifts + sum (+ frames) with the conditioning done by the kernels
    that read the raw bytes (mc_raw_movie_stats, mc_xc_rows_forward_raw, mc_warp_rigid_raw; no fp32 movie).
    Must equal (a) condition_movie -> estimate_global_motion -> motion_correct_sum on the device: shifts
    exactly, images to 1e-5 of their range (fp32 roundings of  raw * gain - mu  differ in the last bit), and
    (b) the ORACLE on the example's numpy conditioning: shifts exactly (and equal to the known drift),
    frames / sum to the north star's 1e-4.  amp 30: the drift leaves the LDS gain cache several times
    (re-centring) and pushes windows over the frame border (clipped columns of raw and gain)."""
    t, h, w = 6, 512, 4096
    raw, gain, dy, dx = _raw_drift_movie(t, h, w, dtype, 11, amp)
    rd, gd = raw.to(dev), gain.to(dev)
    field, total, frames = mc.motion_correct_raw(rd, gd, 1.0, return_frames=True)
    img = mc.condition_movie(rd, gd)
    fa = mc.estimate_global_motion(img, 1.0)
    sa, fra = mc.motion_correct_sum(img, fa, 1.0, return_frames=True)
    assert torch.equal(field, fa)
    assert rel_err(frames, fra) <= 1e-5 and rel_err(total, sa) <= 1e-5
    expect = torch.tensor([[dy[f] - dy[t // 2], dx[f] - dx[t // 2]] for f in range(t)], dtype=torch.float32)
    assert torch.equal(field[:, :, 0, 0].T.cpu(), expect)
    cond = _numpy_condition(raw, gain)
    of = oracle.estimate_global_motion(cond, 1.0)
    assert torch.equal(field.cpu(), of)
    oc = oracle.correct_motion(cond, of, 1.0)
    knife = knife_edge_mask(cond, of, 1.0, "catmull_rom")
    assert_frames_close(frames, oc, knife, max_excluded=0.05)
    keep = ~knife.any(0)
    err = ((total.cpu() - oc.sum(0)).abs() * keep).max() / oc.sum(0).abs().max()
    assert float(err) <= REL


@pytest.mark.parametrize("shape,dtype", [((5, 512, 1024), torch.uint8), ((5, 1024, 512), torch.int16),
                                         ((3, 4092, 5760), torch.uint8), ((3, 4092, 5760), torch.int16)])
def test_fused_raw_path_on_other_engines(mc, dev, shape, dtype):
    """The raw K1 also exists on the workgroup-per-row engine (power-of-two widths other than 4096) and on the
    mixed-radix rows of the K3 formats (mc_xcg_rows_forward_raw, 5760 columns); the raw warp takes any row of
    whole quads.  Same checks against condition_movie + the fp32 path; the fused route must really have been
    taken (no fp32 movie: the engine function raises instead of falling back)."""
    from torch_motion_correction_amd import engine

    t, h, w = shape
    raw, gain, dy, dx = _raw_drift_movie(t, h, w, dtype, 31, 4)
    rd, gd = raw.to(dev), gain.to(dev)
    rm = engine.RawMovie(rd, gd)
    sh = engine.global_shifts_raw(rm, t // 2, 1.0, 500.0, (300, 10))  # McorrUnsupported if no fused kernel
    field, total, frames = mc.motion_correct_raw(rd, gd, 1.0, return_frames=True)
    assert torch.equal(field[:, :, 0, 0].T, sh)
    img = mc.condition_movie(rd, gd)
    fa = mc.estimate_global_motion(img, 1.0)
    sa, fra = mc.motion_correct_sum(img, fa, 1.0, return_frames=True)
    assert torch.equal(field, fa)
    assert rel_err(frames, fra) <= 1e-5 and rel_err(total, sa) <= 1e-5
    expect = torch.tensor([[dy[f] - dy[t // 2], dx[f] - dx[t // 2]] for f in range(t)], dtype=torch.float32)
    assert torch.equal(field[:, :, 0, 0].T.cpu(), expect)


def test_raw_movie_statistics_match_a_float64_reference(mc, dev):
    from torch_motion_correction_amd import engine

    raw, gain, _, _ = _raw_drift_movie(5, 256, 4096, torch.uint8, 2, 3)
    rm = engine.RawMovie(raw.to(dev), gain.to(dev))
    x = raw.double() * gain.double()
    mu = x.mean(dim=(1, 2))
    assert float((rm.mu.cpu().double() - mu).abs().max()) <= 1e-5 * float(mu.abs().max())
    c = x - mu[:, None, None]
    box = c[:, 64:192, 1024:3072]
    std, mean = torch.std_mean(box)  # unbiased, all frames jointly: normalize_image (utils.py:76-84)
    mr = rm.mean_rstd.cpu().double()
    assert abs(float(mr[0]) - float(mean)) <= 1e-5 * float(std) and abs(float(mr[1]) * float(std) - 1.0) <= 1e-5
    assert float((rm.sub.cpu().double() - (mu + mean)).abs().max()) <= 1e-5 * float(mu.abs().max())
    # no gain, no mean-zero: the frames as they are
    rm0 = engine.RawMovie(raw.to(dev), None, mean_zero=False)
    assert float(rm0.mu.abs().max()) == 0.0


def test_fused_raw_path_falls_back_for_other_shapes_and_pipelines_movies(mc, dev):
    """Frame shapes without a fused kernel take condition_movie + the fp32 path inside motion_correct_raw
    (same results by construction); RawMoviePipeline over several movies equals one call per movie."""
    raw, gain, _, _ = _raw_drift_movie(5, 200, 260, torch.uint8, 5, 3)
    f1, s1 = mc.motion_correct_raw(raw.to(dev), gain.to(dev), 1.0)
    img = mc.condition_movie(raw.to(dev), gain.to(dev))
    f2 = mc.estimate_global_motion(img, 1.0)
    assert torch.equal(f1, f2) and torch.equal(s1, mc.motion_correct_sum(img, f2, 1.0))
    movies = [_raw_drift_movie(4, 256, 4096, torch.uint8, 20 + i, 4)[0].to(dev) for i in range(3)]
    g = _raw_drift_movie(4, 256, 4096, torch.uint8, 20, 4)[1].to(dev)
    pipe = mc.RawMoviePipeline(g, dev, 1.0, return_frames=True)
    res = pipe.run(movies)
    torch.cuda.synchronize()
    for m, r in zip(movies, res):
        f, s, fr = mc.motion_correct_raw(m, g, 1.0, return_frames=True)
        assert torch.equal(r.field, f) and torch.equal(r.total, s) and torch.equal(r.frames, fr)
    with pytest.raises(TypeError):
        mc.RawMoviePipeline(g, dev, 1.0).run([movies[0].float()])
