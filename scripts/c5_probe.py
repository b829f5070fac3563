"""C5-like run: 60 x 8184 x 11520 fp16 storage, p=1024 patches (14 x 21) + B-spline warp + sums."""
import sys, time, torch
sys.path.insert(0, ".")
import torch_motion_correction_amd as mc
dev = torch.device("cuda:0")
t, h, w = 60, 8184, 11520
g = torch.Generator(device=dev).manual_seed(5)
pad = 64
base = torch.randn(h + 2 * pad, w + 2 * pad, generator=g, device=dev)
dy = torch.round(torch.linspace(-6, 8, t)).long().tolist()
dx = torch.round(torch.linspace(5, -4, t)).long().tolist()
stack = torch.empty((t, h, w), dtype=torch.float16, device=dev)
for f in range(t):
    fr = base[pad - dy[f] : pad - dy[f] + h, pad - dx[f] : pad - dx[f] + w] + torch.randn(h, w, generator=g, device=dev)
    stack[f] = fr.half()
del base, fr
torch.cuda.synchronize()
print("stack", tuple(stack.shape), stack.dtype, f"{stack.numel() * 2 / 1e9:.1f} GB", flush=True)
for it in range(2):
    torch.cuda.reset_peak_memory_stats()
    t0 = time.perf_counter()
    field, pos = mc.estimate_motion_cross_correlation_patches(stack, 1.0, patch_sidelength=1024)
    torch.cuda.synchronize(); t1 = time.perf_counter()
    total = mc.motion_correct_sum(stack, field, 1.0, grid_type="bspline")
    torch.cuda.synchronize(); t2 = time.perf_counter()
    dsum = mc.motion_correct_sum(stack, field, 1.0, grid_type="bspline", dose_per_frame=1.0)
    torch.cuda.synchronize(); t3 = time.perf_counter()
    print(f"iter {it}: estimate {1e3*(t1-t0):.1f} ms, correct+sum {1e3*(t2-t1):.1f} ms, correct+dose-weighted sum "
          f"{1e3*(t3-t2):.1f} ms, field {tuple(field.shape)}, peak mem {torch.cuda.max_memory_allocated() / 1e9:.1f} GB", flush=True)
    del dsum
fy = field[0].mean(dim=(1, 2)).cpu()
exp_y = torch.tensor([float(d) for d in dy]); exp_y -= exp_y.mean()
print("patch-mean field y (first 6):", [round(float(v), 2) for v in fy[:6]], "expected about", [round(float(v), 2) for v in exp_y[:6]])
print("sum std inside (aligned ~ sqrt(60^2+60) ~ 60):", float(total[64:-64, 64:-64].std()))
