"""Does a hipGraph replay of one whole step (estimate -> correct + sum) pay for small movies?"""
import sys, time, torch
sys.path.insert(0, ".")
import bench
import torch_motion_correction_amd as mc
from torch_motion_correction_amd import pipeline
dev = torch.device("cuda:0")
for (t, n) in ((8, 512), (16, 1024), (40, 4096)):
    stack, dy, dx = bench.synth_stack(t, n, n, 3, dev)
    pipe = pipeline.MoviePipeline(dev, 1.0, return_frames=True, overlap=False)
    def step():
        f = pipe._estimate(stack)
        return f, pipe._correct(stack, f)
    for _ in range(3): out = step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(50): out = step()
    torch.cuda.synchronize()
    eager = (time.perf_counter() - t0) / 50
    g = torch.cuda.CUDAGraph()
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        for _ in range(2): step()
    torch.cuda.current_stream().wait_stream(s)
    torch.cuda.synchronize()
    with torch.cuda.graph(g):
        gout = step()
    torch.cuda.synchronize()
    for _ in range(3): g.replay()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(50): g.replay()
    torch.cuda.synchronize()
    graphed = (time.perf_counter() - t0) / 50
    same = torch.equal(gout[0], out[0]) and torch.equal(gout[1][1], out[1][1])
    print(f"{t}x{n}x{n}: eager {eager*1e3:.3f} ms, graph replay {graphed*1e3:.3f} ms, identical {same}", flush=True)
