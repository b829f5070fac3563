import sys, torch
sys.path.insert(0, ".")
import bench
import torch_motion_correction_amd as mc
dev = torch.device("cuda:0")
st, dy, dx = bench.synth_stack(40, 4092, 5760, 3, dev)
for _ in range(3):
    field = mc.estimate_global_motion(st, 1.0)
torch.cuda.synchronize()
if len(sys.argv) > 1:
    for _ in range(2):
        mc.correct_motion_fast(st, field.clone())
    torch.cuda.synchronize()
