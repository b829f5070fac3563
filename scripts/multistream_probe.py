"""Whole movies round-robin over N streams (each movie: estimate then warp on ITS stream)."""
import sys, time, torch
sys.path.insert(0, ".")
import bench
from torch_motion_correction_amd import pipeline
dev = torch.device("cuda:0")
t, h, w = 40, 4096, 4096
stack, dy, dx = bench.synth_stack(t, h, w, 1234, dev)
pipe = pipeline.MoviePipeline(dev, 1.0, return_frames=True, overlap=False)
def run(nstreams, n):
    streams = [torch.cuda.Stream() for _ in range(nstreams)]
    for s in streams: s.wait_stream(torch.cuda.current_stream())
    out = None
    for i in range(n):
        with torch.cuda.stream(streams[i % nstreams]):
            f = pipe._estimate(stack)
            out = pipe._correct(stack, f)
    for s in streams: torch.cuda.current_stream().wait_stream(s)
    return out
for ns in (1, 2, 3, 4, 2, 3):
    run(ns, 4); torch.cuda.synchronize()
    t0 = time.perf_counter(); out = run(ns, 24); torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 24
    print(f"{ns} streams: {dt*1e3:.3f} ms/movie  {t/dt:.0f} frames/s", flush=True)
two = pipeline.MoviePipeline(dev, 1.0, return_frames=True, overlap=True)
for _ in range(2):
    list(two.iterate([stack] * 4)); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for r in two.iterate([stack] * 24): pass
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 24
    print(f"est/warp pipeline: {dt*1e3:.3f} ms/movie  {t/dt:.0f} frames/s", flush=True)
