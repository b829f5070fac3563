"""General (deformation-field) warp at the C3 frame size: time per mode and field roughness, one
process per kernel version (MC_WARP_FIELD is read once).  usage: field_warp_probe.py [versions...]"""
import os, subprocess, sys
child = r'''
import os, sys, time, torch
sys.path.insert(0, ".")
import bench
import torch_motion_correction_amd as mc
dev = torch.device("cuda:0")
t, h, w = 40, 4092, 5760
st, _, _ = bench.synth_stack(t, h, w, 7, dev)
g = torch.Generator().manual_seed(3)
tt = torch.linspace(-1, 1, t)[:, None, None]; yy = torch.linspace(-1, 1, 6)[None, :, None]; xx = torch.linspace(-1, 1, 10)[None, None, :]
smooth = torch.stack([2.0 * tt * torch.sin(2 * yy + xx), 2.0 * tt * torch.cos(1.5 * xx - yy)]).to(dev)
rough = (torch.randn(2, t, 6, 10, generator=g) * 2.0).to(dev)
out = []
for name, field in (("smooth", smooth), ("rough", rough)):
    for ps in (1.0, 0.83):
        for mode in ("sum", "frames+sum", "frames"):
            def run():
                if mode == "frames":
                    return mc.correct_motion(st, field, ps, grid_type="bspline")
                return mc.motion_correct_sum(st, field, ps, grid_type="bspline", return_frames=(mode != "sum"))
            for _ in range(2): r = run()
            torch.cuda.synchronize(); t0 = time.perf_counter()
            for _ in range(4): r = run()
            torch.cuda.synchronize(); ms = 1e3 * (time.perf_counter() - t0) / 4
            x = r[0] if isinstance(r, tuple) else r
            out.append(f"{name} ps={ps} {mode}: {ms:.2f} ms sig={float(x.double().sum()):.6f}")
            del r, x
print("version", os.environ.get("MC_WARP_FIELD", "default"), "\n  " + "\n  ".join(out), flush=True)
'''
for v in (sys.argv[1:] or ["1", "2"]):
    subprocess.run([sys.executable, "-c", child], env=dict(os.environ, MC_WARP_FIELD=v), check=False)
