"""Row-major full-spectrum path: correct_motion_fast and the exposure-filtered sum at 40 x 4096^2
against the size of the spectrum chunk (frames per chunk: does a chunk that fits the Infinity Cache pay?)."""
import sys, torch
sys.path.insert(0, ".")
import bench
import torch_motion_correction_amd as mc
from torch_motion_correction_amd import engine

dev = torch.device("cuda:0")
st, dy, dx = bench.synth_stack(40, 4096, 4096, 3, dev)
field = mc.estimate_global_motion(st, 1.0)
per_frame = 4096 * 2064 * 8


def timeit(fn, n=3):
    fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / n


ws = engine.WORKSPACE_BYTES
for chunk in (1, 2, 3, 4, 8, 20, 40):
    engine.WORKSPACE_BYTES = chunk * per_frame
    tf = timeit(lambda: mc.correct_motion_fast(st, field.clone()))
    td = timeit(lambda: mc.dose_weighted_sum(st, 1.0, 1.0))
    print(f"chunk {chunk:3d}: correct_motion_fast {tf:7.2f} ms   dose_weighted_sum {td:7.2f} ms", flush=True)
engine.WORKSPACE_BYTES = ws
