import os, sys, time, torch
sys.path.insert(0, ".")
from torch_motion_correction_amd import engine
import torch_motion_correction_amd as mc
dev = torch.device("cuda:0")
t, h, w = 40, int(os.environ.get("HH","4096")), int(os.environ.get("WW","4096"))
g = torch.Generator(device=dev).manual_seed(0)
stack = torch.randn(t, h, w, generator=g, device=dev)
if os.environ.get("HALF", "0") == "1":  # fp16 storage: the rigid kernel reads the 16-bit samples (6 B/px/frame)
    stack = stack.half()
sx = float(os.environ.get("SX", "nan"))
sh = torch.stack([torch.round(torch.linspace(-6, 8, t)), torch.round(torch.linspace(5, -4, t)) if sx != sx else torch.full((t,), sx)], 1)
field = mc.image_shifts_to_deformation_field(sh.to(dev), 1.0).contiguous()
lat = engine.frame_lattices(field, t, "catmull_rom")
for mode in ((True, True), (True, False), (False, True)):
    for _ in range(2):
        engine.warp(stack, lat, 1.0, want_frames=mode[0], want_sum=mode[1], rigid=True)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5):
        engine.warp(stack, lat, 1.0, want_frames=mode[0], want_sum=mode[1], rigid=True)
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 5
    print(os.environ.get("MC_RIGID_VARIANT", "0"), "frames,sum=", mode, f"{ms:.3f} ms", f"{8*h*w*t/ms/1e6:.0f} GB/s (8B/px)", flush=True)
