"""Headline benchmark: frames/s motion-corrected on synthetic 40 x 4096 x 4096 fp32
stacks (BASELINE.json configs[1]): estimate_global_motion -> correct_motion (+ fused
frame sum), inputs and outputs resident in HBM.

    python bench.py --gpus 1 --steps 10 --warmup 3
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

One process per GPU; movies are independent, so ranks never exchange data (weak
scaling: one stack per rank per step).  torch.distributed (RCCL) is used only for the
start/stop barrier and the max-over-ranks of the elapsed time.

Prints ONE JSON line (rank 0).  `roofline` is for the dominant kernel (warp_rigid_dma):
algorithmic bytes per launch = 8 B/pixel/frame x 40 frames (each frame read once and
written once) over its mean duration, measured with HIP events on
the launch stream around each launch inside the timed region.  `cpu_baseline` is the
CPU oracle (a port of the reference's torch-CPU op sequence; the reference itself
cannot be imported) on a bounded sample of the same workload.
"""

from __future__ import annotations

import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

try:
    METRIC = json.load(open(os.path.join(ROOT, "BASELINE.json")))["metric"]
except Exception:
    METRIC = "frames/sec motion-corrected, 40x4kx4k fp32 stack, 1/2/4/8 MI355X"
HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak (MI355X_MICROARCH.md)


def synth_stack(t, h, w, seed, device, noise=1.0, pad=64):
    """SURVEY.md section 8d recipe on the device: white-noise texture cropped at integer
    drift offsets + per-frame noise.  Returns (stack, dy, dx)."""
    g = torch.Generator(device=device).manual_seed(seed)
    base = torch.randn(h + 2 * pad, w + 2 * pad, generator=g, device=device)
    dy = torch.round(torch.linspace(-6, 8, t)).long().tolist()
    dx = torch.round(torch.linspace(5, -4, t)).long().tolist()
    stack = torch.empty((t, h, w), dtype=torch.float32, device=device)
    for f in range(t):
        stack[f] = base[pad - dy[f] : pad - dy[f] + h, pad - dx[f] : pad - dx[f] + w]
        stack[f] += noise * torch.randn(h, w, generator=g, device=device)
    return stack, dy, dx


def synth_local_motion_stack(mc, t, h, w, gh, gw, seed, device, amp=2.0, noise=0.5):
    """Input for the local-motion workload: one texture seen through a smooth, small
    deformation (|shift| <= amp px) that grows with time -- what the patch estimator is made for
    (the rigid-drift stack above is not: its frames are several pixels apart, the correlation with
    the mean of the other frames has one peak per frame and the patch field is meaningless).
    The frames are generated with the product's own warp; this is a timing workload, parity of
    the same construction is tested against the oracle in tests/test_gpu_parity.py."""
    g = torch.Generator(device=device).manual_seed(seed)
    base = torch.randn(h, w, generator=g, device=device)
    base = (base + torch.roll(base, 1, 0) + torch.roll(base, 1, 1) + torch.roll(base, (1, 1), (0, 1))) / 2
    tt = torch.linspace(-1, 1, t)[:, None, None]
    yy = torch.linspace(-1, 1, gh)[None, :, None]
    xx = torch.linspace(-1, 1, gw)[None, None, :]
    true = torch.stack([amp * tt * torch.sin(2.0 * yy + xx), amp * tt * torch.cos(1.5 * xx - yy)]).to(device)
    stack = torch.empty((t, h, w), dtype=torch.float32, device=device)
    for f in range(t):
        stack[f] = mc.correct_motion(base[None], -true[:, f:f + 1], 1.0, grid_type="bspline")[0]
        stack[f] += noise * torch.randn(h, w, generator=g, device=device)
    return stack, float(base.std())


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--frames", type=int, default=40)
    ap.add_argument("--size", type=int, default=4096)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-secondary", action="store_true", help="skip the local-motion (patches) workload")
    ap.add_argument("--no-overlap", action="store_true",
                    help="one stream: every step's estimator waits for the previous step's warp")
    ap.add_argument("--cpu-frames", type=int, default=4)
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    dist = None
    # MC_BENCH_BACKEND / MC_BENCH_ONE_DEVICE exist only to rehearse the multi-rank control
    # path on a single-GPU box (gloo, every rank on cuda:0); the driver uses the defaults.
    backend = os.environ.get("MC_BENCH_BACKEND", "nccl")
    one_device = os.environ.get("MC_BENCH_ONE_DEVICE", "0") == "1"
    dev = torch.device("cuda", 0 if (world == 1 or one_device) else local)
    torch.cuda.set_device(dev)
    if world > 1:
        import torch.distributed as dist

        if backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group(backend)

    import torch_motion_correction_amd as mc
    from torch_motion_correction_amd import engine, pipeline  # noqa: F401

    t, h, w = args.frames, args.size, args.size
    stack, dy, dx = synth_stack(t, h, w, 1234 + rank, dev)
    # consecutive steps alternate between two different stacks with the same drift (same recipe,
    # another seed): no step finds its own input still in a cache
    stack_b, _, _ = synth_stack(t, h, w, 4321 + rank, dev)
    ref = t // 2
    expect = torch.tensor([[dy[f] - dy[ref], dx[f] - dx[ref]] for f in range(t)], dtype=torch.float32)

    warp_events = []

    # One step = one movie through estimate_global_motion -> correct_motion (+ fused sum).
    # With --overlap (default) consecutive steps go through the two-stream movie pipeline
    # (torch_motion_correction_amd/pipeline.py): the estimator of step k+1 is enqueued on a
    # second HIP stream under the warp of step k, as when a list of movies is processed.
    pipe = pipeline.MoviePipeline(dev, 1.0, ref, 500.0, (300, 10), "catmull_rom", return_frames=True,
                                  overlap=not args.no_overlap)

    def timed_warp(fn):  # HIP events on the stream the warp kernel is launched on
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        r = fn()
        e1.record()
        warp_events.append((e0, e1))
        return r

    def run_steps(n, record):
        last = None
        # events around warp_rigid_dma alone (MC_BENCH_NO_EVENTS=1, an A/B knob, times the region without them)
        engine.RIGID_KERNEL_HOOK = timed_warp if record and os.environ.get("MC_BENCH_NO_EVENTS") != "1" else None
        try:
            for res in pipe.iterate([stack, stack_b][i % 2] for i in range(n)):
                last = res  # earlier results are dropped: their memory is reused by the next step
        finally:
            engine.RIGID_KERNEL_HOOK = None
        return last

    def barrier():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    run_steps(args.warmup, False)
    barrier()
    t0 = time.perf_counter()
    out = run_steps(args.steps, True)
    barrier()
    elapsed = time.perf_counter() - t0
    if dist is not None:
        tt = torch.tensor([elapsed], dtype=torch.float64, device=dev if backend == "nccl" else "cpu")
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt.item())

    # the same launch with the chip to itself (outside the timed region): under --overlap the
    # timed launches share HBM with the next step's estimator, which stretches them
    solo_ms = None
    if not args.no_overlap:
        lat = engine.frame_lattices(out.field.contiguous(), t, "catmull_rom")
        ev = []

        def solo_hook(fn):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            fn()
            e1.record()
            ev.append((e0, e1))

        engine.RIGID_KERNEL_HOOK = solo_hook
        try:
            for _ in range(3):
                engine.warp(stack, lat, 1.0, want_frames=True, want_sum=True, rigid=True)
        finally:
            engine.RIGID_KERNEL_HOOK = None
        torch.cuda.synchronize()
        solo_ms = sum(a.elapsed_time(b) for a, b in ev) / len(ev)
    # what this box sustains on a plain copy of buffers larger than the Infinity Cache (read + write
    # bytes per second): the practical ceiling next to the 8 TB/s spec the fractions are quoted against
    copy_rate = None
    if rank == 0:
        try:
            a0 = torch.empty(1 << 28, dtype=torch.float32, device=dev)  # 1 GiB
            b0 = torch.empty_like(a0)
            b0.copy_(a0)
            torch.cuda.synchronize()
            c0, c1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            c0.record()
            for _ in range(5):
                b0.copy_(a0)
            c1.record()
            torch.cuda.synchronize()
            copy_rate = 5 * 2 * a0.numel() * 4 / (c0.elapsed_time(c1) * 1e-3) / 1e9
            del a0, b0
        except Exception:
            copy_rate = None
    shifts = (out.field[:, :, 0, 0].transpose(0, 1) / 1.0).cpu()
    shifts_ok = bool(torch.equal(shifts, expect))
    warp_ms = sum(a.elapsed_time(b) for a, b in warp_events) / max(len(warp_events), 1) if warp_events else float("nan")
    alg_bytes = 8.0 * h * w * t  # read + write of every frame, once
    achieved = alg_bytes / (warp_ms * 1e-3) / 1e9 if warp_events else float("nan")

    cpu = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:  # N = 1 only (contract)
        import oracle

        n = min(args.cpu_frames, t)
        sample = stack[:n].cpu()
        torch.set_num_threads(os.cpu_count() or 1)
        c0 = time.perf_counter()
        ofield = oracle.estimate_global_motion(sample, 1.0)
        osum = oracle.correct_motion(sample, ofield, 1.0).sum(0)
        cpu_s = time.perf_counter() - c0
        # parity at the headline frame size (the oracle as the CHECKER, outside every timed region): the HIP
        # path on the same n frames against what the CPU leg just computed.  Shifts must be equal; the sum
        # is compared away from the 16-px border ring (samples whose coordinate sits exactly on the frame
        # edge are a knife-edge of the reference's zero-outside rule)
        gfield = mc.estimate_global_motion(stack[:n], 1.0)
        gsum = mc.motion_correct_sum(stack[:n], gfield, 1.0)
        torch.cuda.synchronize()
        inner = (slice(16, -16), slice(16, -16))
        cpu = {
            "value": n / cpu_s, "unit": "frames/s", "cores": torch.get_num_threads(), "kind": "port",
            "sample": f"first {n} frames of the same {h}x{w} stack: oracle estimate_global_motion + "
                      f"correct_motion + sum, {cpu_s:.1f} s",
            "parity_shifts_equal": bool(torch.equal(gfield.cpu(), ofield)),
            "parity_rel_err": float((gsum.cpu()[inner] - osum[inner]).abs().max() / osum[inner].abs().max()),
            "parity_tolerance": 1e-4,
        }
        del gfield, gsum

    # The same steps on the same stacks STORED as fp16 (reported, never the headline value): K1 and the
    # rigid warp read the 16-bit samples as they are (8 instead of 12 compulsory B/pixel/frame)
    half_storage = None
    if rank == 0 and world == 1 and not args.no_secondary:
        try:
            sa, sb = stack.half(), stack_b.half()
            n16 = max(10, args.steps // 4)
            for r16 in pipe.iterate([sa, sb][i % 2] for i in range(3)):
                pass
            torch.cuda.synchronize()
            c0 = time.perf_counter()
            for r16 in pipe.iterate([sa, sb][i % 2] for i in range(n16)):
                last16 = r16
            torch.cuda.synchronize()
            el16 = time.perf_counter() - c0
            sh16 = (last16.field[:, :, 0, 0].transpose(0, 1) / 1.0).cpu()
            half_storage = {
                "workload": f"the headline steps on {t}-frame {h}x{w} stacks stored as fp16 (fp32 arithmetic and outputs)",
                "frames_per_s": t * n16 / el16, "ms_per_step": 1e3 * el16 / n16, "steps": n16,
                "whole_step_frac_of_8_bytes_per_px": 8.0 * h * w * t / (el16 / n16) / 1e9 / HBM_PEAK_GBS,
                "shifts_match_ground_truth": bool(torch.equal(sh16, expect)),
            }
            del sa, sb, r16, last16
        except Exception as e:
            half_storage = {"error": repr(e)}

    # N2 (reported, never the headline value): the same movies as RAW uint8 detector counts + a gain reference,
    # through the fused raw pipeline (statistics pass + K1 and the rigid warp conditioning on the fly: no fp32
    # movie), next to conditioning into an fp32 movie first (mc.condition_movie) and the headline pipeline
    raw_u8 = None
    if rank == 0 and world == 1 and not args.no_secondary:
        try:
            gq = torch.Generator(device=dev).manual_seed(99)
            gain = (1.0 + 0.05 * torch.randn(h, w, generator=gq, device=dev)).clamp(0.7, 1.3)
            ra = (stack * 16 + 128).round().clamp(0, 255).to(torch.uint8)
            rb = (stack_b * 16 + 128).round().clamp(0, 255).to(torch.uint8)
            rpipe = pipeline.RawMoviePipeline(gain, dev, 1.0, ref, 500.0, (300, 10), "catmull_rom", return_frames=True,
                                              overlap=not args.no_overlap)
            nraw = max(10, args.steps // 4)
            for rr in rpipe.iterate([ra, rb][i % 2] for i in range(3)):
                pass
            torch.cuda.synchronize()
            torch.cuda.reset_peak_memory_stats()
            m0 = torch.cuda.memory_allocated()
            c0 = time.perf_counter()
            for rr in rpipe.iterate([ra, rb][i % 2] for i in range(nraw)):
                lastr = rr
            torch.cuda.synchronize()
            el_raw = time.perf_counter() - c0
            peak_raw = torch.cuda.max_memory_allocated() - m0
            shr = (lastr.field[:, :, 0, 0].transpose(0, 1) / 1.0).cpu()
            del rr, lastr

            ncond = min(nraw, 10)  # (more steps only measure the caching allocator: every step takes a fresh 2.7 GB movie)

            def conditioned():
                for i in range(ncond):
                    yield mc.condition_movie([ra, rb][i % 2], gain)

            # warm-up as long as the timed loop: every step takes a fresh 2.7 GB movie + outputs, and the caching
            # allocator needs that many rounds before it stops calling hipMalloc (10-30 ms each)
            for _ in pipe.iterate(conditioned()):
                pass
            torch.cuda.synchronize()
            torch.cuda.reset_peak_memory_stats()
            c0 = time.perf_counter()
            for rr in pipe.iterate(conditioned()):
                pass
            torch.cuda.synchronize()
            el_cond = time.perf_counter() - c0
            peak_cond = torch.cuda.max_memory_allocated() - m0
            raw_u8 = {
                "workload": f"the headline steps on {t}-frame {h}x{w} RAW uint8 movies + (h,w) gain reference: "
                            "gain multiply and per-frame mean-zero fused into the kernels that read the raw bytes",
                "frames_per_s": t * nraw / el_raw, "ms_per_step": 1e3 * el_raw / nraw, "steps": nraw,
                "peak_extra_hbm_GB": peak_raw / 1e9,
                "shifts_match_ground_truth": bool(torch.equal(shr, expect)),
                "via_fp32_movie_ms_per_step": 1e3 * el_cond / ncond, "via_fp32_movie_peak_extra_hbm_GB": peak_cond / 1e9,
                # compulsory bytes of the raw flow: the u8 frame read by the statistics, the estimator and the
                # corrector, the corrected fp32 frame written once
                "whole_step_frac_of_7_bytes_per_px": 7.0 * h * w * t / (el_raw / nraw) / 1e9 / HBM_PEAK_GBS,
            }
            del ra, rb, gain, rr
        except Exception as e:
            raw_u8 = {"error": repr(e)}

    # Secondary workload (reported, never the headline value): BASELINE.json configs[2], local
    # motion on a K3-sized stack -- 1024-px patches (6 x 10), B-spline warp, frame sum.
    secondary = None
    if rank == 0 and world == 1 and not args.no_secondary:
        try:
            del out
            torch.cuda.empty_cache()
            t3, h3, w3 = 40, 4092, 5760
            st3, tex_std = synth_local_motion_stack(mc, t3, h3, w3, 6, 10, 7, dev)
            times = []
            for _ in range(3):
                torch.cuda.synchronize()
                c0 = time.perf_counter()
                f3, _ = mc.estimate_motion_cross_correlation_patches(st3, 1.0, patch_sidelength=1024)
                torch.cuda.synchronize()
                c1 = time.perf_counter()
                s3 = mc.motion_correct_sum(st3, f3, 1.0, grid_type="bspline")
                torch.cuda.synchronize()
                times.append((c1 - c0, time.perf_counter() - c1))
            est, cor = min(x[0] for x in times[1:]), min(x[1] for x in times[1:])
            secondary = {
                "workload": f"{t3}-frame {h3}x{w3} fp32 movie, 1024-px patch estimate "
                            f"({f3.shape[2]}x{f3.shape[3]} patches) + B-spline warp + frame sum",
                "estimate_ms": 1e3 * est, "correct_sum_ms": 1e3 * cor, "frames_per_s": t3 / (est + cor),
                # fraction of the 8 TB/s roofline on the flow's algorithmic bytes (sum-only
                # output: every frame read by the estimator and by the corrector, the sum written once)
                "frac": (2.0 * 4 + 4.0 / t3) * h3 * w3 * t3 / (est + cor) / 1e9 / HBM_PEAK_GBS,
                "correct_sum_frac": (4.0 + 4.0 / t3) * h3 * w3 * t3 / cor / 1e9 / HBM_PEAK_GBS,
                "sum_finite": bool(torch.isfinite(s3).all()),
                # aligned frames add coherently: 1.0 = the sum's spread is t x the texture's
                "sum_coherence": float(s3[64:-64, 64:-64].std()) / (t3 * tex_std),
                "input": "one texture through a smooth time-dependent deformation (|shift| <= 2 px) + noise",
            }
            # the optimiser-based estimator on the same movie: 100 Adam iterations refining the
            # patch field's lattice (estimate_local_motion; second call = plans and torch.optim warm)
            lm = []
            for _ in range(2):
                torch.cuda.synchronize()
                c0 = time.perf_counter()
                fl = mc.estimate_local_motion(st3, 1.0, (1024, 1024), (t3, f3.shape[2], f3.shape[3]), f3,
                                              n_iterations=100)
                torch.cuda.synchronize()
                lm.append(time.perf_counter() - c0)
            secondary["local_motion_100_adam_iterations_ms"] = 1e3 * lm[-1]
            secondary["local_motion_field_finite"] = bool(torch.isfinite(fl).all())
            del st3, f3, s3, fl
        except Exception as e:  # never let the secondary workload break the headline line
            secondary = {"error": repr(e)}
        # BASELINE.json configs[4]: 60 x 8184 x 11520 stored as fp16, 1024-px patches (14 x 21), B-spline
        # warp, plain and dose-weighted sum -- one flow, reported next to the C3 numbers
        try:
            torch.cuda.empty_cache()
            t5, h5, w5 = 60, 8184, 11520
            # the same kind of input as C3: one texture seen through a smooth deformation on the 14 x 21
            # patch lattice, |shift| <= 2 px (a rigid several-pixel drift is not what the patch estimator is
            # for: its field on such a stack is meaningless and so is the "aligned" sum)
            st32, tex5 = synth_local_motion_stack(mc, t5, h5, w5, 14, 21, 5, dev)
            st5 = st32.half()
            del st32
            torch.cuda.empty_cache()
            c5t = []
            for _ in range(2):
                torch.cuda.synchronize()
                torch.cuda.reset_peak_memory_stats()
                c0 = time.perf_counter()
                f5, _ = mc.estimate_motion_cross_correlation_patches(st5, 1.0, patch_sidelength=1024)
                torch.cuda.synchronize()
                c1 = time.perf_counter()
                s5 = mc.motion_correct_sum(st5, f5, 1.0, grid_type="bspline")
                torch.cuda.synchronize()
                c2 = time.perf_counter()
                d5 = mc.motion_correct_sum(st5, f5, 1.0, grid_type="bspline", dose_per_frame=1.0)
                torch.cuda.synchronize()
                c5t.append((c1 - c0, c2 - c1, time.perf_counter() - c2))
            est5, cor5, dose5 = c5t[-1]
            secondary["c5"] = {
                "workload": f"{t5}-frame {h5}x{w5} fp16-stored movie, 1024-px patch estimate "
                            f"({f5.shape[2]}x{f5.shape[3]} patches) + B-spline warp + frame sum / dose-weighted sum",
                "estimate_ms": 1e3 * est5, "correct_sum_ms": 1e3 * cor5, "correct_dose_weighted_sum_ms": 1e3 * dose5,
                "frames_per_s_plain_sum": t5 / (est5 + cor5), "frames_per_s_dose_weighted": t5 / (est5 + dose5),
                # 2 h w 2 + h w 4 / 60 = 383.4 MB of algorithmic bytes per frame
                "frac_plain_sum": (2.0 * 2 + 4.0 / t5) * h5 * w5 * t5 / (est5 + cor5) / 1e9 / HBM_PEAK_GBS,
                "frac_dose_weighted": (2.0 * 2 + 4.0 / t5) * h5 * w5 * t5 / (est5 + dose5) / 1e9 / HBM_PEAK_GBS,
                "peak_hbm_GB": torch.cuda.max_memory_allocated() / 1e9,
                "sums_finite": bool(torch.isfinite(s5).all() and torch.isfinite(d5).all()),
                "sum_coherence": float(s5[64:-64, 64:-64].std()) / (t5 * tex5),
                "input": "one texture through a smooth time-dependent deformation (|shift| <= 2 px) + noise, stored as fp16",
                "max_abs_field_px": float(f5.abs().max()),
            }
            del st5, f5, s5, d5
            torch.cuda.empty_cache()
        except Exception as e:
            if isinstance(secondary, dict):
                secondary["c5"] = {"error": repr(e)}

    traffic = None
    tp = os.path.join(ROOT, "profiles", "warp_traffic.json")
    if os.path.exists(tp):
        try:
            traffic = json.load(open(tp)).get("hbm_bytes_per_launch")
        except Exception:
            traffic = None

    if rank == 0:
        total_frames = world * t * args.steps
        line = {
            "metric": METRIC,
            "value": total_frames / elapsed,
            "unit": "frames/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": 1e3 * elapsed / args.steps,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f32",
            "data": "synthetic",
            "config": {
                "workload": f"{t}-frame {h}x{w} fp32 movie, global rigid shift estimate+correct "
                            f"(estimate_global_motion -> correct_motion, frames + sum out), "
                            f"1 stack per GPU per step"
                            + ("" if args.no_overlap else ", consecutive steps overlapped on two HIP streams"),
                "pixel_spacing": 1.0, "b_factor": 500, "frequency_range": [300, 10],
                "shifts_match_ground_truth": shifts_ok,
            },
            "roofline": {
                "bound": "hbm", "kernel": "warp_rigid_dma", "achieved": achieved, "peak": HBM_PEAK_GBS,
                "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS, "traffic": traffic,
                # `traffic` = HBM bytes per launch from the rocprofv3 PMC passes committed under
                # profiles/ (FETCH_SIZE x 2 + WRITE_SIZE, separate passes): NOT measured in this run
                "traffic_source": "profiles/warp_traffic.json (rocprofv3 --pmc, committed)" if traffic else None,
                "ms_per_launch": warp_ms, "algorithmic_bytes_per_launch": alg_bytes,
                # extras: the kernel alone on the chip, and the whole step against SURVEY 8d's
                # compulsory 12 B/pixel/frame (frame read by the estimator, read + written by the warp)
                "ms_per_launch_unshared": solo_ms,
                "frac_unshared": (alg_bytes / (solo_ms * 1e-3) / 1e9 / HBM_PEAK_GBS) if solo_ms else None,
                "whole_step_frac": 12.0 * h * w * t / (elapsed / args.steps) / 1e9 / HBM_PEAK_GBS,
                # context, measured in this run: a torch copy of 1 GiB buffers (read + write GB/s); the
                # kernel's ACTUAL traffic (`traffic`) over its unshared launch time sits on that line
                "box_copy_rate": copy_rate,
                "actual_traffic_rate_unshared": (traffic / (solo_ms * 1e-3) / 1e9) if (traffic and solo_ms) else None,
            },
            "cpu_baseline": cpu,
            "secondary": secondary,
            "fp16_storage": half_storage,
            "raw_u8": raw_u8,
        }
        print(json.dumps(line), flush=True)
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
