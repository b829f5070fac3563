"""Developer probe: per-component HIP-vs-oracle error report on the GPU box.
(Development aid; the judged parity tests are tests/test_gpu_*.py.)"""

import sys
import time
import traceback

import numpy as np
import torch

sys.path.insert(0, ".")
import oracle  # noqa: E402
from oracle import thirdparty_semantics as tp  # noqa: E402
from oracle.make_goldens import blob_stack, drift_stack  # noqa: E402

import torch_motion_correction_amd as mc  # noqa: E402
from torch_motion_correction_amd import engine, plan as planmod  # noqa: E402

dev = torch.device("cuda:0")


def rel(a, b):
    a, b = a.detach().cpu().double(), b.detach().cpu().double()
    return float((a - b).abs().max() / max(b.abs().max(), 1e-30))


def step(name, fn):
    t0 = time.time()
    try:
        out = fn()
        print(f"[ok ] {name}: {out}  ({time.time() - t0:.2f}s)", flush=True)
    except Exception:
        print(f"[ERR] {name}:\n{traceback.format_exc()}", flush=True)


def p_mask():
    res = []
    for n in (32, 64, 256, 1024):
        m = planmod.circle_mask(n, n, n / 4, n / 8, dev).cpu()
        o = tp.circle(n / 4, (n, n), smoothing_radius=n / 8)
        res.append((n, float((m - o).abs().max())))
    m = planmod.circle_mask(64, 128, 16.0, 8.0, dev).cpu()
    o = tp.circle(16.0, (64, 128), smoothing_radius=8.0)
    res.append(("64x128", float((m - o).abs().max())))
    return res


def p_filter():
    res = []
    for n, ps in ((64, 1.0), (256, 1.0), (512, 0.83)):
        pl = planmod.get_xc_plan(n, n, ps, 500.0, (300, 10), dev)
        g = pl.geom
        band = oracle.prepare_bandpass_filter((300, 10), (n, n), ps)
        benv = tp.b_envelope(500, (n, n), ps)
        full = band * benv
        rows = list(range(g.kyp)) + list(range(n - g.kyn, n))
        sub = full[rows][:, : g.nkx].T  # (nkx, nky)
        outside = full.clone()
        outside[rows, : g.nkx] = 0
        res.append((n, dict(W=g.W, nkx=g.nkx, kyp=g.kyp, kyn=g.kyn, y0=g.y0, ny=g.ny, x0=g.x0, x1=g.x1,
                            RG=g.RG), float((pl.filt.cpu() - sub).abs().max()),
                    "pruned-away max", float(outside.abs().max())))
    return res


def p_stats():
    g = torch.Generator().manual_seed(0)
    img = torch.randn(5, 64, 96, generator=g) * 3 + 7
    s = engine.central_box_stats(img.to(dev)).cpu()
    box = img[:, 16:48, 24:72]
    std, mean = torch.std_mean(box)
    return float(s[0] - mean), float(s[2] - std), float(s[1] - 1 / std)


def p_spectrum():
    res = []
    for (t, n) in ((3, 64), (2, 256)):
        g = torch.Generator().manual_seed(1)
        img = torch.randn(t, n, n, generator=g)
        pl = planmod.get_xc_plan(n, n, 1.0, 500.0, (300, 10), dev)
        gm = pl.geom
        d = img.to(dev)
        stats = engine.central_box_stats(d)
        off = torch.arange(t, device=dev, dtype=torch.int64) * (n * n)
        S = engine._forward_spectra(d, off, n, None, pl, stats)
        S = torch.view_as_complex(S.cpu())  # (t, nkx, nky)
        norm = oracle.normalize_image(img)
        mask = tp.circle(n / 4, (n, n), smoothing_radius=n / 8)
        spec = torch.fft.rfftn(norm * mask, dim=(-2, -1)) * oracle.prepare_bandpass_filter(
            (300, 10), (n, n), 1.0) * tp.b_envelope(500, (n, n), 1.0)
        rows = list(range(gm.kyp)) + list(range(n - gm.kyn, n))
        sub = spec[:, rows][:, :, : gm.nkx].transpose(1, 2)
        res.append((n, float((S - sub).abs().max() / sub.abs().max())))
    return res


def p_global():
    res = []
    mov = blob_stack(True)
    f = mc.estimate_global_motion(mov.to(dev), 1.0)
    o = oracle.estimate_global_motion(mov, 1.0)
    res.append(("blob", f[:, :, 0, 0].cpu().tolist(), o[:, :, 0, 0].tolist()))
    for (t, n) in ((8, 256), (8, 512)):
        st, dy, dx = drift_stack(t, n, n)
        f = mc.estimate_global_motion(st.to(dev), 1.0).cpu()
        o = oracle.estimate_global_motion(st, 1.0)
        res.append((n, bool((f == o).all()), f[0, :, 0, 0].tolist()))
    return res


def p_correct():
    res = []
    stat = blob_stack(False)
    z = torch.zeros(2, 5, 2, 2)
    res.append(("zero", rel(mc.correct_motion(stat.to(dev), z.to(dev), 1.0), oracle.correct_motion(stat, z, 1.0))))
    f22 = torch.zeros(2, 5, 2, 2)
    for f in range(5):
        f22[0, f], f22[1, f] = 0.1 * f, 0.05 * f
    for gt in ("catmull_rom", "bspline"):
        res.append((gt, rel(mc.correct_motion(stat.to(dev), f22.to(dev), 1.0, grid_type=gt),
                            oracle.correct_motion(stat, f22, 1.0, grid_type=gt))))
    g = torch.Generator().manual_seed(3)
    img = torch.randn(6, 96, 160, generator=g)
    fld = torch.randn(2, 4, 3, 5, generator=g) * 3
    for gt in ("catmull_rom", "bspline"):
        a = mc.correct_motion(img.to(dev), fld.to(dev), 1.3, grid_type=gt)
        b = oracle.correct_motion(img, fld, 1.3, grid_type=gt)
        res.append(("rand96x160 " + gt, rel(a, b)))
    st, dy, dx = drift_stack(8, 256, 256)
    o = oracle.estimate_global_motion(st, 1.0)
    a = mc.correct_motion(st.to(dev), o.to(dev), 1.0)
    b = oracle.correct_motion(st, o, 1.0)
    res.append(("drift256 frames", rel(a, b), "sum", rel(a.sum(0), b.sum(0))))
    s = mc.motion_correct_sum(st.to(dev), o.to(dev), 1.0)
    res.append(("fused sum", rel(s, b.sum(0))))
    return res


def p_lattice():
    g = torch.Generator().manual_seed(5)
    fld = torch.randn(2, 4, 3, 5, generator=g)
    res = []
    for gt in ("catmull_rom", "bspline"):
        a = mc.evaluate_deformation_field_at_t(fld.to(dev), 0.37, (30, 50), gt)
        b = oracle.evaluate_deformation_field_at_t(fld, 0.37, (30, 50), gt)
        res.append((gt, rel(a, b)))
    one = torch.randn(2, 5, 1, 1, generator=g)
    a = mc.evaluate_deformation_field_at_t(one.to(dev), 0.25, (10, 10))
    b = oracle.evaluate_deformation_field_at_t(one, 0.25, (10, 10))
    res.append(("1x1", rel(a, b)))
    a = mc.resample_deformation_field(fld.to(dev), (6, 4, 7))
    b = oracle.resample_deformation_field(fld, (6, 4, 7))
    res.append(("resample", rel(a, b)))
    pts = torch.rand(17, 3, generator=g)
    a = mc.evaluate_deformation_field(fld.to(dev), pts)
    b = oracle.evaluate_deformation_field(fld, pts)
    res.append(("points", rel(a, b)))
    lat = torch.randn(2, 20, 30, generator=g)
    a = mc.get_pixel_shifts(torch.zeros(64, 96, device=dev), 1.7, lat.to(dev))
    b = oracle.get_pixel_shifts(torch.zeros(64, 96), 1.7, lat, tp.coordinate_grid((64, 96)))
    res.append(("pixel_shifts", rel(a, b)))
    return res


def p_fast():
    stat = blob_stack(False)
    f11 = torch.zeros(2, 5, 1, 1)
    for f in range(5):
        f11[0, f], f11[1, f] = 0.1 * f, 0.05 * f
    a = mc.correct_motion_fast(stat.to(dev), f11.clone().to(dev))
    b = oracle.correct_motion_fast(stat, f11.clone())
    g = torch.Generator().manual_seed(3)
    img = torch.randn(3, 64, 128, generator=g)
    sh = torch.randn(2, 3, 1, 1, generator=g) * 4
    a2 = mc.correct_motion_fast(img.to(dev), sh.clone().to(dev))
    b2 = oracle.correct_motion_fast(img, sh.clone())
    return rel(a, b), rel(a2, b2)


def p_patches():
    res = []
    mov = blob_stack(True)
    for s in ("mean_except_current", "middle_frame"):
        a, pos = mc.estimate_motion_cross_correlation_patches(mov.to(dev), 1.0, patch_sidelength=32,
                                                              reference_strategy=s)
        b, posb = oracle.estimate_motion_cross_correlation_patches(mov, 1.0, patch_sidelength=32,
                                                                   reference_strategy=s)
        res.append((s, rel(a, b), bool((pos.cpu() == posb).all())))
    st, dy, dx = drift_stack(8, 256, 256)
    a, _ = mc.estimate_motion_cross_correlation_patches(st.to(dev), 1.0, patch_sidelength=64)
    b, _ = oracle.estimate_motion_cross_correlation_patches(st, 1.0, patch_sidelength=64)
    res.append(("drift256 p64", rel(a, b), float((a.cpu() - b).abs().max())))
    a, _ = mc.estimate_motion_cross_correlation_patches(st.to(dev), 1.0, patch_sidelength=64,
                                                        sub_pixel_refinement=False, temporal_smoothing=False,
                                                        outlier_rejection=False)
    b, _ = oracle.estimate_motion_cross_correlation_patches(st, 1.0, patch_sidelength=64,
                                                            sub_pixel_refinement=False,
                                                            temporal_smoothing=False, outlier_rejection=False)
    res.append(("drift256 p64 raw", rel(a, b)))
    return res


if __name__ == "__main__":
    print(torch.cuda.get_device_name(0), flush=True)
    for name, fn in (("mask", p_mask), ("filter", p_filter), ("stats", p_stats), ("spectrum", p_spectrum),
                     ("global", p_global), ("lattice", p_lattice), ("correct", p_correct),
                     ("fast", p_fast), ("patches", p_patches)):
        step(name, fn)
    torch.cuda.synchronize()
    print("done", flush=True)
