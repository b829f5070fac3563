import sys, time, torch
sys.path.insert(0, ".")
import bench
import torch_motion_correction_amd as mc
from torch_motion_correction_amd import pipeline
dev = torch.device("cuda:0")
t, h, w = 40, 4096, 4096
stack, dy, dx = bench.synth_stack(t, h, w, 1234, dev)
gq = torch.Generator(device=dev).manual_seed(99)
gain = (1.0 + 0.05 * torch.randn(h, w, generator=gq, device=dev)).clamp(0.7, 1.3)
ra = (stack * 16 + 128).round().clamp(0, 255).to(torch.uint8)
del stack
for _ in range(3): c = mc.condition_movie(ra, gain)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(10): c = mc.condition_movie(ra, gain)
torch.cuda.synchronize()
print("condition_movie alone ms", 1e2 * (time.perf_counter() - t0))
pipe = pipeline.MoviePipeline(dev, 1.0, t // 2, 500.0, (300, 10), "catmull_rom", return_frames=True, overlap=True)
def gen(n):
    for i in range(n):
        yield mc.condition_movie(ra, gain)
for _ in pipe.iterate(gen(3)): pass
torch.cuda.synchronize()
for n in (10, 10):
    t0 = time.perf_counter()
    for r in pipe.iterate(gen(n)): pass
    torch.cuda.synchronize()
    print("pipeline over conditioned movies ms/step", 1e3 * (time.perf_counter() - t0) / n, "mem GB", torch.cuda.memory_reserved() / 1e9)
held = [mc.condition_movie(ra, gain) for _ in range(2)]
t0 = time.perf_counter()
for r in pipe.iterate(held[i % 2] for i in range(10)): pass
torch.cuda.synchronize()
print("pipeline over two held fp32 movies ms/step", 1e3 * (time.perf_counter() - t0) / 10)
