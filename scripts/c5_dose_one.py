"""The exposure-filtered sum at BASELINE C5's frame size (8184 x 11520), 12 fp32 frames."""
import sys, torch
sys.path.insert(0, ".")
import torch_motion_correction_amd as mc
dev = torch.device("cuda:0")
t, h, w = 12, 8184, 11520
g = torch.Generator(device=dev).manual_seed(5)
st = torch.randn(t, h, w, generator=g, device=dev)
for _ in range(2):
    out = mc.dose_weighted_sum(st, 1.0, 1.0)
torch.cuda.synchronize()
print(float(out.std()))
