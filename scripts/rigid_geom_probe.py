"""A/B of the fused-sum rigid warp's tile geometry (MC_RIGID_GEOM=WXWY, one process per variant
because the library reads the variable once).  usage: python scripts/rigid_geom_probe.py [geoms...]"""
import os, subprocess, sys
geoms = sys.argv[1:] or ["14", "22", "24", "41", "42"]
child = r'''
import os, sys, torch, hashlib
sys.path.insert(0, ".")
from torch_motion_correction_amd import engine
import torch_motion_correction_amd as mc
dev = torch.device("cuda:0")
t, h, w = 40, int(os.environ.get("HH", "4096")), int(os.environ.get("WW", "4096"))
g = torch.Generator(device=dev).manual_seed(0)
stack = torch.randn(t, h, w, generator=g, device=dev)
sh = torch.stack([torch.round(torch.linspace(-6, 8, t)) + 0.25, torch.round(torch.linspace(5, -4, t)) - 0.4], 1)
field = mc.image_shifts_to_deformation_field(sh.to(dev), 1.0).contiguous()
lat = engine.frame_lattices(field, t, "catmull_rom")
out = []
for mode in ((True, True), (True, False), (False, True)):
    for _ in range(2):
        fr, sm = engine.warp(stack, lat, 1.0, want_frames=mode[0], want_sum=mode[1], rigid=True)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5):
        engine.warp(stack, lat, 1.0, want_frames=mode[0], want_sum=mode[1], rigid=True)
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 5
    sig = [float(x.double().sum()) if x is not None else None for x in (fr, sm)]
    out.append(f"frames,sum={mode} {ms:.3f} ms {8*h*w*t/ms/1e6:.0f} GB/s sig={sig}")
print(os.environ.get("MC_RIGID_GEOM"), os.environ.get("MC_RIGID_NBUF", "1"), " | ".join(out), flush=True)
'''
for gm in geoms:
    env = dict(os.environ, MC_RIGID_GEOM=gm.split(":")[0])
    if ":" in gm:
        env["MC_RIGID_NBUF"] = gm.split(":")[1]
    subprocess.run([sys.executable, "-c", child], env=env, check=False)
