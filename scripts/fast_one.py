import sys, torch
sys.path.insert(0, ".")
import bench
import torch_motion_correction_amd as mc
dev = torch.device("cuda:0")
st, dy, dx = bench.synth_stack(40, 4096, 4096, 3, dev)
field = mc.estimate_global_motion(st, 1.0)
for _ in range(3):
    mc.correct_motion_fast(st, field.clone())
torch.cuda.synchronize()
