import sys, torch
sys.path.insert(0, ".")
import bench
import torch_motion_correction_amd as mc
dev = torch.device("cuda:0")
t, h, w = 60, 8184, 11520
g = torch.Generator(device=dev).manual_seed(5)
base = torch.randn(h + 128, w + 128, generator=g, device=dev)
dy = torch.round(torch.linspace(-6, 8, t)).long().tolist(); dx = torch.round(torch.linspace(5, -4, t)).long().tolist()
stack = torch.empty((t, h, w), dtype=torch.float16, device=dev)
for f in range(t):
    stack[f] = (base[64 - dy[f]: 64 - dy[f] + h, 64 - dx[f]: 64 - dx[f] + w] + torch.randn(h, w, generator=g, device=dev)).half()
del base
for _ in range(2):
    field, pos = mc.estimate_motion_cross_correlation_patches(stack, 1.0, patch_sidelength=1024)
    total = mc.motion_correct_sum(stack, field, 1.0, grid_type="bspline")
torch.cuda.synchronize()
