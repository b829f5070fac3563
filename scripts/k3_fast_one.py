import sys, torch
sys.path.insert(0, ".")
import bench
import torch_motion_correction_amd as mc
dev = torch.device("cuda:0")
t, h, w = (int(a) for a in (sys.argv[1:4] if len(sys.argv) > 3 else (40, 4092, 5760)))
st, dy, dx = bench.synth_stack(t, h, w, 3, dev)
for _ in range(2):
    field = mc.estimate_global_motion(st, 1.0)
    mc.correct_motion_fast(st, field.clone())
    mc.dose_weighted_sum(st, 1.0, 1.0)
torch.cuda.synchronize()
