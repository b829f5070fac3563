# sum-only general warp at the C3 size, kernel-level timing (ablation runs)
import os, sys, torch
sys.path.insert(0, ".")
import bench
from torch_motion_correction_amd import engine
dev = torch.device("cuda:0")
t, h, w = 40, 4092, 5760
st, _, _ = bench.synth_stack(t, h, w, 7, dev)
tt = torch.linspace(-1, 1, t)[:, None, None]; yy = torch.linspace(-1, 1, 6)[None, :, None]; xx = torch.linspace(-1, 1, 10)[None, None, :]
field = torch.stack([2.0 * tt * torch.sin(2 * yy + xx), 2.0 * tt * torch.cos(1.5 * xx - yy)]).to(dev)
lat = engine.frame_lattices(field.contiguous(), t, "bspline")
for wf, ws in ((False, True),):
    for _ in range(2): r = engine.warp(st, lat, 1.0, want_frames=wf, want_sum=ws)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(4): r = engine.warp(st, lat, 1.0, want_frames=wf, want_sum=ws)
    e1.record(); torch.cuda.synchronize()
    print(f"frames={int(wf)} sum={int(ws)}: {e0.elapsed_time(e1)/4:.2f} ms", flush=True)
