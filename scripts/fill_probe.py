"""Fixed cost of a timed region of K pipelined steps (what bench.py --steps K sees): total(K) = a K + b."""
import sys, time, torch
sys.path.insert(0, ".")
import bench
from torch_motion_correction_amd import pipeline
dev = torch.device("cuda:0")
t, h, w = 40, 4096, 4096
stack, dy, dx = bench.synth_stack(t, h, w, 1234, dev)
stack_b, _, _ = bench.synth_stack(t, h, w, 4321, dev)
pipe = pipeline.MoviePipeline(dev, 1.0, t // 2, 500.0, (300, 10), "catmull_rom", return_frames=True, overlap=True)
def run(n):
    last = None
    for res in pipe.iterate([stack, stack_b][i % 2] for i in range(n)):
        last = res
    return last
run(5); torch.cuda.synchronize()
for K in (1, 2, 3, 5, 10, 20, 40, 20, 10, 1):
    best = 1e9
    for _ in range(3):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        run(K)
        t1 = time.perf_counter()
        torch.cuda.synchronize()
        t2 = time.perf_counter()
        best = min(best, t2 - t0)
    print(f"K={K:3d}: total {1e3*best:7.3f} ms  per step {1e3*best/K:6.3f}  host enqueue {1e3*(t1-t0):7.3f} ms", flush=True)
