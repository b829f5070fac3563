# kernel-level timing of the general warp at the C3 frame size (smooth field, unit spacing)
import os, sys, time, torch
sys.path.insert(0, ".")
import bench
import torch_motion_correction_amd as mc
from torch_motion_correction_amd import engine
dev = torch.device("cuda:0")
t, h, w = 40, 4092, 5760
st, _, _ = bench.synth_stack(t, h, w, 7, dev)
tt = torch.linspace(-1, 1, t)[:, None, None]; yy = torch.linspace(-1, 1, 6)[None, :, None]; xx = torch.linspace(-1, 1, 10)[None, None, :]
field = torch.stack([2.0 * tt * torch.sin(2 * yy + xx), 2.0 * tt * torch.cos(1.5 * xx - yy)]).to(dev)
lat = engine.frame_lattices(field.contiguous(), t, "bspline")
res = []
for ps in (1.0, 0.83):
    for wf, ws in ((False, True), (True, True), (True, False)):
        for _ in range(2): r = engine.warp(st, lat, ps, want_frames=wf, want_sum=ws)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(4): r = engine.warp(st, lat, ps, want_frames=wf, want_sum=ws)
        e1.record(); torch.cuda.synchronize()
        x = r[1] if ws else r[0]
        res.append(f"ps={ps} frames={int(wf)} sum={int(ws)}: {e0.elapsed_time(e1)/4:.2f} ms sig={float(x.double().sum()):.4f}")
        del r, x
print("\n".join(res), flush=True)
