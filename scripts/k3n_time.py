"""Time the global estimator's kernels on 40 x 4096^2 (run under rocprofv3 for per-kernel numbers)."""
import sys, torch
sys.path.insert(0, ".")
from torch_motion_correction_amd import engine
dev = torch.device("cuda:0")
g = torch.Generator(device=dev).manual_seed(0)
stack = torch.randn(40, 4096, 4096, generator=g, device=dev)
for _ in range(6):
    sh = engine.global_shifts(stack, 20, 1.0, 500.0, (300, 10))
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(10):
    engine.global_shifts(stack, 20, 1.0, 500.0, (300, 10))
e1.record(); torch.cuda.synchronize()
print("global_shifts ms", e0.elapsed_time(e1) / 10, "sum|shift|", float(sh.abs().sum()))
