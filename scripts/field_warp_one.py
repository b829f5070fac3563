# one mode of the general warp at the C3 frame size, a few launches (for rocprofv3 passes)
import os, sys, torch
sys.path.insert(0, ".")
import bench
import torch_motion_correction_amd as mc
dev = torch.device("cuda:0")
t, h, w = 40, 4092, 5760
st, _, _ = bench.synth_stack(t, h, w, 7, dev)
tt = torch.linspace(-1, 1, t)[:, None, None]; yy = torch.linspace(-1, 1, 6)[None, :, None]; xx = torch.linspace(-1, 1, 10)[None, None, :]
field = torch.stack([2.0 * tt * torch.sin(2 * yy + xx), 2.0 * tt * torch.cos(1.5 * xx - yy)]).to(dev)
mode = os.environ.get("MODE", "sum")
for _ in range(int(os.environ.get("REPS", "4"))):
    if mode == "frames":
        mc.correct_motion(st, field, 1.0, grid_type="bspline")
    else:
        mc.motion_correct_sum(st, field, 1.0, grid_type="bspline", return_frames=(mode != "sum"))
torch.cuda.synchronize()
