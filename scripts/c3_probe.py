"""C3-like run: 40 x 4092 x 5760 patches (p=1024 -> 6x10) + B-spline warp, timing + sanity."""
import sys, time, torch
sys.path.insert(0, ".")
import bench
import torch_motion_correction_amd as mc
dev = torch.device("cuda:0")
t, h, w = 40, int(sys.argv[1]) if len(sys.argv) > 1 else 4092, int(sys.argv[2]) if len(sys.argv) > 2 else 5760
stack, dy, dx = bench.synth_stack(t, h, w, 7, dev)
torch.cuda.synchronize()
for it in range(2):
    t0 = time.perf_counter()
    field, pos = mc.estimate_motion_cross_correlation_patches(stack, 1.0, patch_sidelength=1024)
    torch.cuda.synchronize(); t1 = time.perf_counter()
    total = mc.motion_correct_sum(stack, field, 1.0, grid_type="bspline")
    torch.cuda.synchronize(); t2 = time.perf_counter()
    print(f"iter {it}: estimate {1e3*(t1-t0):.1f} ms, correct+sum {1e3*(t2-t1):.1f} ms, field {tuple(field.shape)}", flush=True)
exp_y = torch.tensor([float(d - sum(dy) / t) for d in dy]); exp_x = torch.tensor([float(d - sum(dx) / t) for d in dx])
fy = field[0].mean(dim=(1, 2)).cpu(); fx = field[1].mean(dim=(1, 2)).cpu()
print("patch-mean field y (first 6):", [round(float(v), 2) for v in fy[:6]], " drift-relative:", [round(float(v), 2) for v in (exp_y - exp_y.mean())[:6]])
print("spread across patches (max std over frames):", float(field.std(dim=(2, 3)).max()))
print("sum std (aligned ~ 40):", float(total[64:-64, 64:-64].std()))
print("mem GB", torch.cuda.max_memory_allocated() / 1e9)
