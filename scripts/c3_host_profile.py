"""Where the wall time of the C3 patch estimate goes on the host (cProfile) against its GPU time (events)."""
import cProfile, pstats, sys, time, torch
sys.path.insert(0, ".")
import bench
import torch_motion_correction_amd as mc
dev = torch.device("cuda:0")
t, h, w = 40, 4092, 5760
st, _ = bench.synth_local_motion_stack(mc, t, h, w, 6, 10, 7, dev)
for _ in range(2):
    f, _ = mc.estimate_motion_cross_correlation_patches(st, 1.0, patch_sidelength=1024)
torch.cuda.synchronize()
for name, fn in (("estimate", lambda: mc.estimate_motion_cross_correlation_patches(st, 1.0, patch_sidelength=1024)),
                 ("correct_sum", lambda: mc.motion_correct_sum(st, f, 1.0, grid_type="bspline"))):
    walls, gpus, hosts = [], [], []
    for _ in range(5):
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        c0 = time.perf_counter(); e0.record()
        fn()
        e1.record(); c1 = time.perf_counter()
        torch.cuda.synchronize()
        walls.append(time.perf_counter() - c0); gpus.append(e0.elapsed_time(e1)); hosts.append(c1 - c0)
    print(f"{name}: wall {1e3*min(walls):.2f} ms, events {min(gpus):.2f} ms, host enqueue {1e3*min(hosts):.2f} ms", flush=True)
    pr = cProfile.Profile(); pr.enable(); fn(); torch.cuda.synchronize(); pr.disable()
    pstats.Stats(pr).sort_stats("cumulative").print_stats(18)
