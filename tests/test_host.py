"""CPU tests of the product's host logic (no GPU, no compute calls into libmcorr):
lattice + memo replay pinned by the reference's goldens, pruning geometry proven
conservative against the oracle's filters/mask, spline tap tables against the
oracle's spline, C-ABI symbol export, and world_size-2 gloo sharding."""

import ctypes
import os
import re

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

import oracle
from oracle import thirdparty_semantics as tp
from torch_motion_correction_amd import _lib, lattice, multi_gpu, plan, spline

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


# ------------------------------------------------------------------ lattice / memo replay


def test_patch_centres_match_reference(golden):
    g = golden("patch_grid_reference.npz")
    for key in g.files:
        if key.startswith("centers_"):
            _, dim, p = key.split("_")
            assert np.array_equal(lattice.patch_centers_1d(int(dim), int(p), int(p) // 2), g[key]), key


def test_baseline_config_lattices():
    """SURVEY.md section 8 header: p=1024 -> C2 6x6, C3 6x10, C5 14x21."""
    assert [len(a) for a in lattice.patch_grid_centers(40, 4096, 4096, 1024)] == [6, 6]
    assert [len(a) for a in lattice.patch_grid_centers(40, 4092, 5760, 1024)] == [6, 10]
    assert [len(a) for a in lattice.patch_grid_centers(60, 8184, 11520, 1024)] == [14, 21]
    cy, _ = lattice.patch_grid_centers(40, 4096, 4096, 1024)
    assert cy.tolist() == [512, 1126, 1740, 2355, 2969, 3583]


@pytest.mark.parametrize("t", [5, 8, 40, 51, 60])
@pytest.mark.parametrize("strategy", ["middle_frame", "mean_except_current"])
def test_mask_schedule_matches_reference(golden, t, strategy):
    g = golden("patch_grid_reference.npz")
    ref_expo, cur_expo, processed = lattice.mask_schedule(t, strategy, t // 2)
    assert np.array_equal(ref_expo, g[f"exp_{strategy}_{t}"])
    assert np.array_equal(cur_expo, g[f"cur_{strategy}_{t}"])
    expect = [f for f in range(t) if not (strategy == "middle_frame" and f == t // 2)]
    assert processed == expect


def test_mask_schedule_unknown_strategy():
    with pytest.raises(ValueError, match="Unknown reference_strategy"):
        lattice.mask_schedule(5, "nope", 2)


def test_centers_tensor_layout():
    cy, cx = lattice.patch_grid_centers(3, 64, 96, 32)
    c = lattice.centers_tensor(3, cy, cx)
    assert c.shape == (3, len(cy), len(cx), 3) and c.dtype == torch.int64
    assert c[2, 1, 0].tolist() == [2, int(cy[1]), int(cx[0])]


# ------------------------------------------------------------------ pruning geometry


@pytest.mark.parametrize("n,ps,fr", [(64, 1.0, (300, 10)), (256, 1.0, (300, 10)), (512, 0.83, (300, 10)),
                                     (128, 1.5, (200, 20)), (64, 6.0, (300, 10))])
def test_geometry_is_a_superset_of_the_filter_support(n, ps, fr):
    low, high = plan.band_limits(fr, ps)
    g = plan.xc_geometry(n, n, high, n / 4, n / 8)
    filt = oracle.prepare_bandpass_filter(fr, (n, n), ps) * tp.b_envelope(500, (n, n), ps)
    rows = list(range(g.kyp)) + list(range(n - g.kyn, n))
    outside = filt.clone()
    outside[rows, : g.nkx] = 0
    assert float(outside.abs().max()) == 0.0, "a non-zero filter bin would be pruned away"
    assert g.kyp + g.kyn <= n and 1 <= g.nkx <= n // 2 + 1
    mask = tp.circle(n / 4, (n, n), smoothing_radius=n / 8)
    outside = mask.clone()
    outside[g.y0 : g.y0 + g.ny, g.x0 : g.x1] = 0
    assert float(outside.abs().max()) == 0.0, "a non-zero mask pixel would never be read"
    assert g.ny % g.RG == 0 and n % g.RG == 0 and g.x0 % 2 == 0 and g.x1 % 2 == 0


def test_headline_geometry_prunes():
    """40x4096^2 at 1 A/px, (300,10) A: ~10 % of the columns, ~20 % of the rows kept."""
    low, high = plan.band_limits((300, 10), 1.0)
    g = plan.xc_geometry(4096, 4096, high, 1024, 512)
    assert (g.nkx, g.kyp, g.kyn) == (410, 410, 409)
    assert g.ny <= 3104 and g.x1 - g.x0 <= 3104


def test_full_geometry_keeps_everything():
    g = plan.full_geometry(64, 128)
    assert (g.nkx, g.kyp, g.kyn, g.y0, g.ny, g.x0, g.x1) == (65, 64, 0, 0, 64, 0, 128)


def test_unsupported_sizes_fail_loudly():
    with pytest.raises(NotImplementedError, match="even widths"):
        plan.xc_geometry(4092, 8193, 0.1, 1000, 500)  # odd widths: one sample per line point, at most 8191
    assert plan.xc_geometry(959, 927, 0.1, 927 / 4, 927 / 8).nkx == 93  # the reference's example movie size
    with pytest.raises(NotImplementedError, match="even widths"):
        plan.xc_geometry(8200, 11520, 0.1, 2000, 1000)  # beyond the 8192-row limit
    with pytest.raises(NotImplementedError, match="even widths"):
        plan.xc_geometry(4096, 16400, 0.1, 1000, 500)  # beyond the 16384-column limit
    with pytest.raises(NotImplementedError, match="M="):
        plan.line_plan(8200, -1, "cpu")
    with pytest.raises(NotImplementedError, match="160 KB"):
        plan.full_geometry(8184, 11520)  # the full spectrum of a super-resolution frame (correct_motion_fast)


def test_super_resolution_frames_fit_the_band_limited_transforms():
    """8184 x 11520 (BASELINE config 5 frames): the rows (5760 = 2^7 3^2 5 complex points) and the
    columns (8184 = 2^3 3 11 31: radix-31 and radix-11 passes) are transformed directly by the
    mixed-radix engine (round 2; chirp-z lines of 8192 / 16384 points before)."""
    low, high = plan.band_limits((300, 10), 1.0)
    g = plan.xc_geometry(8184, 11520, high, 8184 / 4, 8184 / 8)
    assert (g.nkx, g.kyp + g.kyn) == (1153, 1637) and g.RG >= 1 and 8184 % g.RG == 0
    fwd, _ = plan.line_plan(5760, -1, "cpu", keep=g.nkx + 1)
    inv, _ = plan.line_plan(5760, +1, "cpu")
    col, _ = plan.line_plan(8184, -1, "cpu")
    assert (fwd.M, fwd.keep, inv.M, col.M) == (5760, 0, 5760, 8184)
    col4, _ = plan.line_plan(4092, +1, "cpu")
    assert (col4.M, col4.keep) == (4092, 0)
    try:  # the chirp-z plans are still there (other lengths, and as the cross-check of the direct lines)
        plan.USE_DIRECT_LINES = False
        plan._LINES.clear()
        fwd, _ = plan.line_plan(5760, -1, "cpu", keep=g.nkx + 1)
        inv, _ = plan.line_plan(5760, +1, "cpu")
        col, _ = plan.line_plan(8184, -1, "cpu")
        col4, _ = plan.line_plan(4092, +1, "cpu")
        assert (fwd.M, fwd.keep, inv.M, col.M, col4.M) == (8192, g.nkx + 1, 16384, 16384, 8192)
    finally:
        plan.USE_DIRECT_LINES = True
        plan._LINES.clear()
    assert 8 * (plan.bluestein_size(5760) * 17 // 16 + 1 + g.nkx * (g.RG + 1)) <= 160 * 1024


def test_k3_geometry_and_chirp_tables():
    """BASELINE config 3 frames (4092 x 5760) go through the chirp-z path."""
    low, high = plan.band_limits((300, 10), 1.0)
    g = plan.xc_geometry(4092, 5760, high, 1023.0, 511.5)
    assert (g.nkx, g.kyp, g.kyn) == (577, 410, 409) and 4092 % g.RG == 0 and g.ny % g.RG == 0
    line, (tw, chirp, bspec) = plan.line_plan(12, -1, "cpu")
    assert line.M == 32 and chirp.shape == (12, 2) and bspec.shape == (32, 2)
    # chirp-z identity on the host: DFT_12(x) == c * ifft(fft(x*c, 32) * B)[:12] * 32 (B holds 1/32)
    rng = np.random.default_rng(0)
    x = rng.standard_normal(12) + 1j * rng.standard_normal(12)
    c = chirp.numpy()[:, 0] + 1j * chirp.numpy()[:, 1]
    B = bspec.numpy()[:, 0] + 1j * bspec.numpy()[:, 1]
    a = np.zeros(32, complex)
    a[:12] = x * c
    X = c * (np.fft.ifft(np.fft.fft(a) * B) * 32)[:12]
    assert np.allclose(X, np.fft.fft(x), atol=1e-5)


def test_band_limits_follow_reference_ops():
    low, high = plan.band_limits((300, 10), 1.0)
    assert low == pytest.approx(1 / 300, rel=1e-6) and high == pytest.approx(0.1, rel=1e-6)
    assert np.float32(high) == np.float32(torch.as_tensor(1 / torch.tensor(10.0)) * 1.0)


def test_twiddles():
    tw = plan.twiddles(8, "cpu").numpy()
    k = np.arange(8)
    assert np.allclose(tw[:, 0], np.cos(2 * np.pi * k / 8), atol=1e-7)
    assert np.allclose(tw[:, 1], -np.sin(2 * np.pi * k / 8), atol=1e-7)


# ------------------------------------------------------------------ spline taps


@pytest.mark.parametrize("kind", ["catmull_rom", "bspline"])
@pytest.mark.parametrize("shape", [(4, 3, 5), (5, 1, 1), (2, 2, 2), (1, 1, 1), (6, 1, 4)])
def test_axis_taps_reproduce_oracle_spline(kind, shape):
    g = torch.Generator().manual_seed(3)
    data = torch.randn(2, *shape, generator=g)
    ut, uy, ux = torch.linspace(0, 1, 7), torch.rand(5, generator=g), torch.linspace(0, 1, 6)
    taps = [spline.axis_taps(n, u, kind) for n, u in zip(shape, (ut, uy, ux))]
    out = torch.zeros(2, 7, 5, 6)
    for it in range(7):
        for iy in range(5):
            for ix in range(6):
                acc = torch.zeros(2)
                for kt in range(4):
                    for ky in range(4):
                        for kx in range(4):
                            w = taps[0][1][it, kt] * taps[1][1][iy, ky] * taps[2][1][ix, kx]
                            acc += w * data[:, taps[0][0][it, kt], taps[1][0][iy, ky], taps[2][0][ix, kx]]
                out[:, it, iy, ix] = acc
    tyx = torch.stack(torch.meshgrid(ut, uy, ux, indexing="ij"), -1)
    ref = tp.cubic_spline_grid_3d(data, tyx, kind).permute(3, 0, 1, 2)
    assert torch.allclose(out, ref, atol=2e-5)


def test_axis_taps_bad_kind():
    with pytest.raises(ValueError, match="grid_type"):
        spline.axis_taps(4, torch.linspace(0, 1, 3), "linear")


# ------------------------------------------------------------------ C ABI


def test_library_exports_every_declared_symbol():
    header = open(os.path.join(ROOT, "include", "mcorr.h")).read()
    declared = set(re.findall(r"^int\s+(mc_\w+)\s*\(", header, flags=re.M))
    assert len(declared) >= 20
    assert declared == set(_lib.SIGNATURES), declared ^ set(_lib.SIGNATURES)
    lib = _lib.load()
    for name in declared:
        assert hasattr(lib, name), name
    assert lib.mc_abi_version() == 1


def test_abi_rejects_bad_arguments_without_touching_the_gpu():
    lib = _lib.load()
    g = plan.xc_geometry(64, 64, 0.1, 16, 8)
    assert lib.mc_xc_rows_lds_bytes(g) > 0
    bad = _lib.XcGeom(W=100, H=64, nkx=5, kyp=5, kyn=4, y0=0, ny=64, x0=0, x1=100, RG=16)
    assert lib.mc_xc_rows_lds_bytes(bad) == -2  # MC_ERR_UNSUPPORTED: not a power of two
    n = ctypes.c_int64(0)
    assert lib.mc_warp_scratch_bytes(4, 64, 64, 10, 10, ctypes.byref(n)) == 0 and n.value > 0
    assert lib.mc_warp_rigid_scratch_bytes(4, 64, 64, ctypes.byref(n)) == 0 and n.value > 0
    assert lib.mc_circle_mask(None, None, 64, 64, 16.0, 8.0, None) == -1  # MC_ERR_ARG
    assert lib.mc_warp_frames(None, 1, 64, 64, None, 10, 10, 1.0, None, None, None, None) == -1


def test_full_spectrum_and_storage_entry_points_validate_on_the_host():
    """The round-2 entry points (row-major full-spectrum transforms, fp16 storage tags) check sizes,
    pointers and storage tags before any launch: pitch rule, supported row / column lengths (powers of
    two and the K3 formats), MC_ERR_ARG for null pointers, MC_ERR_UNSUPPORTED for sizes the library has
    no kernel for and for storage types that are not read natively."""
    lib = _lib.load()
    one = ctypes.c_void_p(16)  # a non-null, 16-byte aligned address: argument checks never dereference it
    assert [lib.mc_full_spectrum_pitch(w) for w in (64, 4096, 5760, 11520)] == [48, 2064, 2896, 5776]
    for (h, w) in ((4096, 4096), (256, 64), (4092, 5760), (8184, 11520), (4092, 4096), (256, 11520)):
        pitch = lib.mc_full_spectrum_pitch(w)
        assert lib.mc_full_rows_forward(None, one, w, one, one, 1, h, w, pitch, None) == -1  # null src
        assert lib.mc_full_cols_shift(None, one, one, 1.0, 1, h, w, pitch, None) == -1
        assert lib.mc_full_rows_inverse(one, None, one, w, one, 1, h, w, pitch, None) == -1
        assert lib.mc_full_cols_dose(one, 0, 0, 1, one, one, h, w, pitch, 1.0, 0.0, 1.0, 300.0, 1, 1, 1.0, None) == -1
    for (h, w) in ((4000, 4096), (4096, 4000), (128, 64), (8192, 64), (4092, 32)):  # no kernel for these
        pitch = lib.mc_full_spectrum_pitch(w)
        assert lib.mc_full_rows_forward(one, one, w, one, one, 1, h, w, pitch, None) == -2
        assert lib.mc_full_cols_shift(one, one, one, 1.0, 1, h, w, pitch, None) == -2
    assert lib.mc_full_rows_forward(one, one, 4096, one, one, 1, 4096, 4096, 2049, None) == -2  # pitch rule
    # storage tags: 2 = fp16, 3 = fp32 are read natively; raw detector types go through mc_condition_movie
    n = ctypes.c_int64(0)
    assert lib.mc_warp_rigid_scratch_bytes(2, 64, 64, ctypes.byref(n)) == 0
    assert lib.mc_warp_rigid_phase_t(one, 0, 2, 64, 64, one, one, one, one, 0, None) == -2   # u8 frames
    assert lib.mc_warp_rigid_phase_t(one, 2, 2, 64, 60, one, one, one, one, 0, None) == -2   # fp16, w % 8 != 0
    assert lib.mc_warp_rigid_phase_t(one, 3, 2, 64, 64, None, one, one, one, 0, None) == -1  # null shifts
    assert lib.mc_xc_provisional_mean_t(one, 1, 16, one, None) == -2 and lib.mc_xc_provisional_mean_t(None, 2, 16, one, None) == -1
    g = plan.xc_geometry(4096, 4096, 0.1, 1024, 512)
    assert lib.mc_xc_rows_forward_stats_t(one, 1, one, 4096, one, one, one, one, 1, g, 1024, 3072, 1024, 3072, one,
                                          one, one, None, None) == -2  # int16 frames are not read natively


def test_fp16_stacks_never_reach_an_fp32_only_kernel():
    """fp16 storage is read natively by the LDS-staged kernels only; every other shape must come back as
    MC_ERR_UNSUPPORTED *before* any launch (the callers then widen once), never fall through to a kernel
    that would read the 16-bit buffer as fp32 -- twice its bytes.  Checked on the host with
    pointers that are never dereferenced: rows that are not whole 8-sample units, an unaligned stack, a
    dense (per-pixel) lattice; and the rigid warp the same way."""
    lib = _lib.load()
    fake = lambda a: ctypes.c_void_p(a)  # noqa: E731  (never dereferenced: the shape checks come first)
    F16, F32 = 2, 3
    ok_ptr, odd_ptr = 0x10000, 0x10004
    cases = [
        (ok_ptr, 4, 64, 68, 10, 10),    # w % 8 != 0 (but w % 4 == 0: the fp32 tile kernels would take it)
        (ok_ptr, 4, 64, 66, 10, 10),    # w % 4 != 0: the untiled fp32 kernel's shape
        (odd_ptr, 4, 64, 64, 10, 10),   # stack not 16-byte aligned
        (ok_ptr, 4, 64, 64, 64, 64),    # dense lattice: warp_field2's case, fp32 only
    ]
    for base, t, h, w, GH, GW in cases:
        rc = lib.mc_warp_frames_t(fake(base), F16, t, h, w, fake(0x20000), GH, GW, 1.0, fake(0x30000), fake(0x40000),
                                  None, None)
        assert rc == -2, (base, t, h, w, GH, GW, rc)
    # storage tags that no warp reads natively
    for tag in (0, 1, 7):
        assert lib.mc_warp_frames_t(fake(ok_ptr), tag, 4, 64, 64, fake(0x20000), 10, 10, 1.0, fake(0x30000),
                                    fake(0x40000), None, None) == -2
    for base, w in ((ok_ptr, 68), (odd_ptr, 64)):
        assert lib.mc_warp_rigid_phase_t(fake(base), F16, 4, 64, w, fake(0x20000), fake(0x30000), fake(0x40000),
                                         None, 0, None) == -2
    header = open(os.path.join(ROOT, "include", "mcorr.h")).read()
    assert "#define MC_STORE_F16 2" in header and "#define MC_STORE_F32 3" in header and F32 == 3


def test_raw_entry_points_validate_on_the_host():
    """N2 entry points (mc_raw_movie_stats, mc_xc_rows_forward_raw, mc_warp_rigid_raw) check storage tags,
    shapes and pointers before any launch."""
    lib = _lib.load()
    fake = lambda a: ctypes.c_void_p(a)  # noqa: E731
    U8, I16, F16, F32 = 0, 1, 2, 3
    p = [fake(0x10000 * (i + 1)) for i in range(8)]
    # statistics: box must lie inside the frame; storage tag 0..3
    assert lib.mc_raw_movie_stats(p[0], U8, p[1], 4, 64, 64, 16, 48, 16, 80, 1, p[2], p[3], p[4], p[5], None) == -1
    assert lib.mc_raw_movie_stats(p[0], 9, p[1], 4, 64, 64, 16, 48, 16, 48, 1, p[2], p[3], p[4], p[5], None) == -2
    assert lib.mc_raw_movie_stats(None, U8, p[1], 4, 64, 64, 16, 48, 16, 48, 1, p[2], p[3], p[4], p[5], None) == -1
    # K1 from raw bytes: u8 / i16 only, power-of-two widths only (the K3 formats have mc_xcg_rows_forward_raw)
    g4096 = plan.xc_geometry(4096, 4096, 0.1, 16, 8)
    gk3 = plan.xc_geometry(4092, 5760, 0.1, 16, 8)
    args = lambda st, g: (p[0], st, p[1], p[2], 4096, p[3], p[4], p[5], p[6], p[7], 2, g, None, None)  # noqa: E731
    assert lib.mc_xc_rows_forward_raw(*args(F32, g4096)) == -2
    assert lib.mc_xc_rows_forward_raw(*args(F16, g4096)) == -2
    assert lib.mc_xc_rows_forward_raw(*args(U8, gk3)) in (-1, -2)
    assert lib.mc_xc_rows_forward_raw(None, U8, p[1], p[2], 4096, p[3], p[4], p[5], p[6], p[7], 2, g4096, None, None) == -1
    # rigid warp from raw bytes: rows of whole quads, aligned buffers, at most 256 frames
    w = lambda st, nf, ww, raw=p[0]: lib.mc_warp_rigid_raw(raw, st, p[1], p[2], nf, 64, ww, p[3], p[4], p[5], None,  # noqa: E731
                                                           0, None)
    assert w(F32, 4, 64) == -2 and w(F16, 4, 64) == -2
    assert w(U8, 4, 66) == -2 and w(U8, 300, 64) == -2 and w(I16, 4, 64, fake(0x10004)) == -2
    assert lib.mc_warp_rigid_raw(p[0], U8, None, p[2], 4, 64, 64, p[3], p[4], p[5], None, 0, None) == -1


def test_product_never_imports_the_oracle():
    pkg = os.path.join(ROOT, "torch_motion_correction_amd")
    for fn in os.listdir(pkg):
        if fn.endswith(".py"):
            src = open(os.path.join(pkg, fn)).read()
            assert not re.search(r"^\s*(from|import)\s+oracle\b", src, flags=re.M), fn


def test_no_gpu_means_loud_failure():
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    import torch_motion_correction_amd as mc

    with pytest.raises(mc.McorrError, match="no CPU fallback"):
        mc.estimate_global_motion(torch.zeros(3, 64, 64), 1.0)


# ------------------------------------------------------------------ sharding (gloo, 2 ranks)


def test_round_robin_assignment():
    a = multi_gpu.assignment(64, 8)
    assert all(len(x) == 8 for x in a)
    assert sorted(i for x in a for i in x) == list(range(64))
    assert multi_gpu.movies_for_rank(5, 1, 2) == [1, 3]
    assert multi_gpu.movies_for_rank(1, 1, 2) == []  # ragged: more ranks than movies
    with pytest.raises(ValueError):
        multi_gpu.movies_for_rank(4, 2, 2)


def _rank_main(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        n_movies = 5
        mine = multi_gpu.movies_for_rank(n_movies, rank, world)
        # a CPU stand-in for the per-movie work: each movie's known integer drift is
        # "estimated" from its id alone, so the gathered table can be checked exactly
        local = multi_gpu.process_shard(mine, load=lambda i: torch.full((2,), float(i)),
                                        work=lambda x: (x * 2).tolist())
        dist.barrier()
        merged = multi_gpu.gather_results(local, world)
        elapsed = torch.tensor([0.1 * (rank + 1)], dtype=torch.float64)
        dist.all_reduce(elapsed, op=dist.ReduceOp.MAX)  # the bench's max-over-ranks
        q.put((rank, mine, merged, float(elapsed)))
    finally:
        dist.destroy_process_group()


def test_two_rank_gloo_sharding():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + (os.getpid() % 2000)
    procs = [ctx.Process(target=_rank_main, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    results = [q.get(timeout=120) for _ in procs]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    results.sort()
    assert results[0][1] == [0, 2, 4] and results[1][1] == [1, 3]
    for _, _, merged, elapsed in results:
        assert merged == {i: [2.0 * i, 2.0 * i] for i in range(5)}
        assert elapsed == pytest.approx(0.2)


@pytest.mark.parametrize("t", [5, 40, 51, 60])
def test_leave_one_out_schedule_replays_the_table(t):
    """the incremental add/rebuild schedule must reproduce S_f exactly, the way the kernel
    applies it (rebuild: replace; otherwise: union)"""
    ref_expo, _, _ = lattice.mask_schedule(t, "mean_except_current", t // 2)
    ptr, idx, rebuild = lattice.leave_one_out_schedule(ref_expo)
    running: set = set()
    for f in range(t):
        lst = set(idx[ptr[f]:ptr[f + 1]].tolist()) if ptr[f + 1] > ptr[f] else set()
        running = lst if rebuild[f] else running | lst
        assert running == {o for o in range(t) if o != f and ref_expo[f, o] == 1}
    if t <= 50:
        assert int(rebuild.sum()) == 1  # only frame 0; afterwards one frame joins per step


def test_movie_pipeline_needs_a_gpu_and_is_exported():
    """No CPU fallback for the batch entry point either."""
    import torch_motion_correction_amd as m

    assert {"motion_correct_movies", "MoviePipeline", "MovieResult"} <= set(m.__all__)
    if not torch.cuda.is_available():
        with pytest.raises(m.McorrError):
            m.motion_correct_movies([torch.zeros(2, 64, 64)], 1.0)


def test_deformation_field_csv_round_trip_and_wire_format(tmp_path):
    """N4: CSV interchange identical to the reference's data_io.py (header, row order,
    float64 images of the float32 values; sorted-unique index axes on read)."""
    import torch_motion_correction_amd as m

    g = torch.Generator().manual_seed(4)
    field = torch.randn(2, 3, 2, 4, generator=g)
    p = tmp_path / "sub" / "field.csv"
    m.write_deformation_field_to_csv(field, p)
    lines = p.read_text().strip().split("\n")
    assert lines[0] == "t,h,w,y_shift,x_shift" and len(lines) == 1 + 3 * 2 * 4
    # the reference's own writer, element by element (data_io.py:39-62)
    t_, h_, w_, y_, x_ = lines[1 + (1 * 2 + 1) * 4 + 2].split(",")
    assert (int(t_), int(h_), int(w_)) == (1, 1, 2)
    assert float(y_) == field[0, 1, 1, 2].item() and float(x_) == field[1, 1, 1, 2].item()
    assert y_ == repr(field[0, 1, 1, 2].item())
    back = m.read_deformation_field_from_csv(p)
    assert back.dtype == torch.float32 and back.device.type == "cpu" and torch.equal(back, field)
    # non-contiguous index values: axes are the sorted sets of values (data_io.py:103-120)
    q = tmp_path / "gaps.csv"
    q.write_text("t,h,w,y_shift,x_shift\n5,0,7,1.5,-2.0\n2,0,3,0.25,4.0\n")
    gaps = m.read_deformation_field_from_csv(q)
    assert gaps.shape == (2, 2, 1, 2)
    assert gaps[0, 0, 0, 0] == 0.25 and gaps[1, 1, 0, 1] == -2.0 and gaps[0, 1, 0, 0] == 0.0
    with pytest.raises(ValueError):
        m.write_deformation_field_to_csv(torch.zeros(3, 2, 2), p)


@pytest.mark.parametrize("src", ["host_wave_fft.cpp", "host_wave_fft512.cpp", "host_wave_fft1024.cpp"])
def test_wave_fft_index_algebra_on_the_host(tmp_path, src):
    """csrc/mc_wave_fft.h is written __host__ __device__: the lane ownership, slab addressing,
    pruned butterflies and in-lane / lane-permute real-FFT unpack of both wave transforms
    (2048- and 512-point rows, 1024-point columns) are executed lane by lane on the CPU and compared with a
    double-precision DFT.  Needs clang++ (ext_vector_type); the ROCm one is used."""
    import shutil
    import subprocess

    cxx = "/opt/rocm/lib/llvm/bin/clang++"
    if not os.path.exists(cxx):
        cxx = shutil.which("clang++")
    if not cxx:
        pytest.skip("no clang++ on this machine")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = tmp_path / "a.out"
    subprocess.run([cxx, "-O1", "-std=c++17", "-I", os.path.join(root, "torch_motion_correction_amd", "csrc"),
                    os.path.join(root, "tests", src), "-o", str(exe), "-lm"], check=True)
    out = subprocess.run([str(exe)], check=True, capture_output=True, text=True).stdout
    assert out.strip().endswith("OK"), out


def test_every_name_the_reference_exports_is_exported_here():
    """src/torch_motion_correction/__init__.py:31-43 (__all__ of the reference)."""
    import torch_motion_correction_amd as mc

    reference_all = ["estimate_local_motion", "correct_motion", "correct_motion_two_grids", "correct_motion_fast",
                     "correct_motion_slow", "get_pixel_shifts", "evaluate_deformation_field",
                     "estimate_global_motion", "estimate_motion_cross_correlation_patches",
                     "write_deformation_field_to_csv", "read_deformation_field_from_csv"]
    for name in reference_all:
        assert name in mc.__all__ and callable(getattr(mc, name)), name
    import inspect

    sig = inspect.signature(mc.estimate_local_motion)
    assert list(sig.parameters) == ["image", "pixel_spacing", "patch_shape", "deformation_field_resolution",
                                    "initial_deformation_field", "device", "n_iterations", "b_factor",
                                    "frequency_range", "optimizer_type", "grid_type", "loss_type",
                                    "optimizer_kwargs", "return_trajectory", "trajectory_kwargs"]  # :28-44
    assert sig.parameters["n_iterations"].default == 100 and sig.parameters["loss_type"].default == "mse"
    assert list(inspect.signature(mc.correct_motion_two_grids).parameters) == [
        "image", "new_deformation_grid", "base_deformation_grid", "pixel_spacing", "grad", "device"]
    assert list(inspect.signature(mc.correct_motion_slow).parameters) == ["image", "deformation_grid", "grad", "device"]


def test_three_operation_division_is_the_correctly_rounded_quotient():
    """warp.hip grid_chain divides by the invariant d = 0.5 n - 0.5 as q = c r; e = fma(-q, d, c);
    q' = fma(e, r, q).  Checked here against exact rational arithmetic (the reference divides)."""
    from fractions import Fraction
    import math

    def rn32(fr):
        if fr == 0:
            return np.float32(0)
        sgn = 1 if fr > 0 else -1
        fr = abs(fr)
        e = math.floor(math.log2(fr))
        while Fraction(2) ** e > fr:
            e -= 1
        while Fraction(2) ** (e + 1) <= fr:
            e += 1
        scaled = fr / Fraction(2) ** e * (1 << 23)
        fl = scaled.numerator // scaled.denominator
        rem = scaled - fl
        if rem > Fraction(1, 2) or (rem == Fraction(1, 2) and fl % 2 == 1):
            fl += 1
        return np.float32(sgn * float(fl) * 2.0 ** (e - 23))

    rng = np.random.default_rng(0)
    for n in (64, 959, 4092, 4096, 5760, 8184, 11520):
        d = np.float32(0.5) * np.float32(n) - np.float32(0.5)
        r = np.float32(1.0) / d
        a = (rng.random(300, dtype=np.float32) * np.float32(n + 40) - np.float32(20)).astype(np.float32)
        a[:100] = np.round(a[:100]) + rng.choice([0, 1e-4, -1e-4, 0.5], 100).astype(np.float32)
        for x in a:
            q = np.float32(x * r)
            e = rn32(Fraction(float(x)) - Fraction(float(q)) * Fraction(float(d)))
            q1 = rn32(Fraction(float(q)) + Fraction(float(e)) * Fraction(float(r)))
            assert q1 == rn32(Fraction(float(x)) / Fraction(float(d))), (n, x)


def test_row_chords_bracket_every_nonzero_mask_value():
    """plan.row_chords: the per-row clamp range K1 uses must contain every non-zero mask sample
    (a missed one would be replaced by a clamped neighbour) and is tight to 4-sample quads."""
    mask = tp.circle(40.0, (128, 160), smoothing_radius=20.0)
    ch = plan.row_chords(mask)
    assert ch.shape == (128, 2) and ch.dtype == torch.int32
    for y in range(128):
        nz = torch.nonzero(mask[y]).flatten()
        lo, hi = int(ch[y, 0]), int(ch[y, 1])
        assert lo % 4 == 0 and hi % 4 == 0 and 0 <= lo <= hi <= 156
        if len(nz):
            assert lo <= int(nz[0]) < lo + 4 and hi <= int(nz[-1]) < hi + 4
        else:
            assert lo == hi == 80
    # the reference's central statistics box lies inside the chords of the reference's mask
    h = w = 256
    m = tp.circle(min(h, w) / 4, (h, w), smoothing_radius=min(h, w) / 8)
    c = plan.row_chords(m)[int(0.25 * h):int(0.75 * h)]
    assert bool((c[:, 0] <= int(0.25 * w)).all()) and bool((c[:, 1] + 4 >= int(0.75 * w)).all())


def test_launches_refuse_a_device_that_is_not_current(monkeypatch):
    """libmcorr launches on the CURRENT HIP device; buffers elsewhere must fail loudly, not launch
    on the wrong GPU (the API enters _lib.device_scope(device) around every call)."""
    import torch
    from torch_motion_correction_amd import _lib

    monkeypatch.setattr(torch.cuda, "current_device", lambda: 1)
    with pytest.raises(_lib.McorrError, match="current device"):
        _lib.stream_ptr(torch.device("cuda:0"))
    with pytest.raises(_lib.McorrError, match="ROCm device"):
        _lib.stream_ptr(torch.device("cpu"))


def test_reference_frame_normalisation():
    from torch_motion_correction_amd import _lib

    assert _lib.normalize_frame_index(-1, 5) == 4 and _lib.normalize_frame_index(3, 5) == 3
    assert _lib.normalize_frame_index(-5, 5) == 0
    for bad in (5, -6, 100):
        with pytest.raises(IndexError):
            _lib.normalize_frame_index(bad, 5)
    with pytest.raises(TypeError):
        _lib.normalize_frame_index(1.5, 5)


def test_mask_schedule_with_a_negative_reference_key():
    """-1 and t-1 are two memo entries of the reference's LazyPatchGrid (the key is the raw int):
    no frame is skipped and the reference entry collects one mask factor per processed frame."""
    ref_expo, cur_expo, processed, ref_read = lattice.mask_schedule(5, "middle_frame", -1, with_ref_reads=True)
    assert processed == [0, 1, 2, 3, 4]
    assert ref_read.tolist() == [0, 1, 2, 3, 4] and cur_expo.tolist() == [0, 0, 0, 0, 0]
    ref_expo, cur_expo, processed, ref_read = lattice.mask_schedule(5, "middle_frame", 4, with_ref_reads=True)
    assert processed == [0, 1, 2, 3] and ref_read.tolist() == [0, 1, 2, 3, -1]


class _FakeResult:
    def __init__(self, field, total):
        self.field, self.total = field, total


class _FakePipeline:
    """CPU stand-in with MoviePipeline's iterate() contract: the 'field' of a movie is its
    per-frame mean (so the gathered table can be checked exactly), the 'sum' its frame sum."""

    def iterate(self, movies):
        for m in movies:
            f = torch.stack([m.mean(dim=(1, 2)), -m.mean(dim=(1, 2))])[:, :, None, None]
            yield _FakeResult(f, m.sum(0))


def _sharded_rank_main(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        loaded = []

        def load(i):
            loaded.append(i)
            return torch.full((3, 4, 5), float(i)) + torch.arange(3.0)[:, None, None]

        fields, sums = multi_gpu.motion_correct_movies_sharded(
            list(range(7)), rank, world, 1.0, load=load, pipeline_factory=_FakePipeline)
        q.put((rank, loaded, {k: v.tolist() for k, v in fields.items()}, sorted(sums)))
    finally:
        dist.destroy_process_group()


def test_sharded_driver_two_ranks_gloo():
    """motion_correct_movies_sharded: every rank touches only its round-robin shard, the fields of
    ALL movies come back on every rank, the sums stay with their rank (7 movies over 2 ranks)."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 31500 + (os.getpid() % 2000)
    procs = [ctx.Process(target=_sharded_rank_main, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    results = sorted(q.get(timeout=120) for _ in procs)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    (r0, l0, f0, s0), (r1, l1, f1, s1) = results
    assert l0 == [0, 2, 4, 6] and l1 == [1, 3, 5] and s0 == l0 and s1 == l1
    assert f0 == f1 and sorted(f0) == list(range(7))
    for i in range(7):
        assert f0[i][0] == [[[float(i)]], [[float(i) + 1]], [[float(i) + 2]]]


def test_sharded_driver_single_rank_needs_no_process_group():
    fields, sums = multi_gpu.motion_correct_movies_sharded(
        [torch.ones(2, 3, 3) * k for k in range(3)], 0, 1, 1.0, pipeline_factory=_FakePipeline)
    assert sorted(fields) == [0, 1, 2] and sorted(sums) == [0, 1, 2]
    assert float(sums[2].sum()) == 2 * 2 * 9
