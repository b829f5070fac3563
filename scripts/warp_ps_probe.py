# general (patch-field) warp at the C3 frame size for a non-unit pixel spacing
import sys, time, torch
sys.path.insert(0, ".")
import bench
import torch_motion_correction_amd as mc
dev = torch.device("cuda:0")
t, h, w = 40, 4092, 5760
st, _, _ = bench.synth_stack(t, h, w, 7, dev)
g = torch.Generator().manual_seed(3)
field = (torch.randn(2, t, 6, 10, generator=g) * 2.0).to(dev)
for ps in (1.0, 0.83):
    for _ in range(2):
        mc.motion_correct_sum(st, field, ps, grid_type="bspline")
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(5):
        mc.motion_correct_sum(st, field, ps, grid_type="bspline")
    torch.cuda.synchronize()
    print(f"pixel_spacing {ps}: correct+sum {1e3*(time.perf_counter()-t0)/5:.2f} ms", flush=True)
