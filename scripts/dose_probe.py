"""Dose-weighted sum against the plain fused sum at the C2 size, and the C5 estimate / correct peak memory."""
import sys, time, torch
sys.path.insert(0, ".")
import bench
import torch_motion_correction_amd as mc
dev = torch.device("cuda:0")
def timed(fn, n=3):
    for _ in range(2): r = fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): r = fn()
    torch.cuda.synchronize()
    return 1e3 * (time.perf_counter() - t0) / n, r
st, dy, dx = bench.synth_stack(40, 4096, 4096, 3, dev)
field = mc.estimate_global_motion(st, 1.0)
ms_plain, s0 = timed(lambda: mc.motion_correct_sum(st, field, 1.0))
ms_dose, s1 = timed(lambda: mc.motion_correct_sum(st, field, 1.0, dose_per_frame=1.0))
ms_dose_only, _ = timed(lambda: mc.dose_weighted_sum(st, 1.0, 1.0))
print(f"C2 40x4096^2: plain fused sum {ms_plain:.2f} ms, motion_correct_sum(dose) {ms_dose:.2f} ms, dose_weighted_sum of resident frames {ms_dose_only:.2f} ms", flush=True)
del st, s0, s1
torch.cuda.empty_cache()
if len(sys.argv) > 1:
    t, h, w = 60, 8184, 11520
    g = torch.Generator(device=dev).manual_seed(5)
    stack = torch.empty((t, h, w), dtype=torch.float16, device=dev)
    for f in range(t):
        stack[f] = torch.randn(h, w, generator=g, device=dev).half()
    for name, fn in (("estimate", lambda: mc.estimate_motion_cross_correlation_patches(stack, 1.0, patch_sidelength=1024)),):
        torch.cuda.reset_peak_memory_stats(); base = torch.cuda.memory_allocated()
        ms, r = timed(fn, 2)
        print(f"C5 {name}: {ms:.1f} ms, peak {torch.cuda.max_memory_allocated()/1e9:.1f} GB (stack {base/1e9:.1f} GB)", flush=True)
    field = r[0]
    torch.cuda.reset_peak_memory_stats()
    ms, r = timed(lambda: mc.motion_correct_sum(stack, field, 1.0, grid_type="bspline"), 2)
    print(f"C5 correct+sum: {ms:.1f} ms, peak {torch.cuda.max_memory_allocated()/1e9:.1f} GB", flush=True)
