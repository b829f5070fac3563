# BASELINE C3 as one flow (for rocprofv3 passes): 40 x 4092 x 5760 local-motion stack, 1024-px patch
# estimate (6 x 10) + B-spline warp + frame sum, a few repetitions
import os, sys, torch
sys.path.insert(0, ".")
import bench
import torch_motion_correction_amd as mc
dev = torch.device("cuda:0")
st, _ = bench.synth_local_motion_stack(mc, 40, 4092, 5760, 6, 10, 7, dev)
for _ in range(int(os.environ.get("REPS", "3"))):
    f, _ = mc.estimate_motion_cross_correlation_patches(st, 1.0, patch_sidelength=1024)
    s = mc.motion_correct_sum(st, f, 1.0, grid_type="bspline")
torch.cuda.synchronize()
print("sum coherence", float(s[64:-64, 64:-64].std()))
