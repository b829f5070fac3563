# PCIe-inclusive rate: stacks start in pinned host memory, every step uploads its own stack
# (a copy stream runs one movie ahead of the two compute streams).  Never the bench's `value`.
import sys, time, torch
sys.path.insert(0, ".")
import bench
from torch_motion_correction_amd import pipeline
dev = torch.device("cuda:0")
t, h, w = 40, 4096, 4096
stack, dy, dx = bench.synth_stack(t, h, w, 1234, dev)
host = torch.empty((t, h, w), dtype=torch.float32, pin_memory=True)
host.copy_(stack)
torch.cuda.synchronize()
# raw H2D bandwidth
bufs = [torch.empty_like(stack) for _ in range(2)]
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for i in range(4):
    bufs[i & 1].copy_(host, non_blocking=True)
e1.record(); torch.cuda.synchronize()
gbps = 4 * host.numel() * 4 / (e0.elapsed_time(e1) * 1e-3) / 1e9
print(f"H2D pinned: {gbps:.1f} GB/s -> {gbps * 1e9 / (h * w * 4):.0f} frames/s upload ceiling", flush=True)
# streamed pipeline: upload of movie k+1 under estimate/correct of movie k
pipe = pipeline.MoviePipeline(dev, 1.0, t // 2, 500.0, (300, 10), "catmull_rom", return_frames=True)
copy_stream = torch.cuda.Stream(dev)
def movies(n):
    ring = [torch.empty_like(stack) for _ in range(3)]
    evs = [None] * 3
    for k in range(n):
        b = ring[k % 3]
        with torch.cuda.stream(copy_stream):
            b.copy_(host, non_blocking=True)
            ev = torch.cuda.Event(); ev.record(copy_stream)
        torch.cuda.current_stream(dev).wait_event(ev)
        pipe._s_est.wait_event(ev); pipe._s_warp.wait_event(ev)
        yield b
for n in (3, 12):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    last = None
    for r in pipe.iterate(movies(n)):
        last = r
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    if n > 3:
        print(f"streamed from pinned host memory: {n * t / dt:.0f} frames/s ({1e3 * dt / n:.1f} ms per movie); shifts ok: "
              f"{bool((last.field[0, :, 0, 0].cpu() == torch.tensor([float(d - dy[t // 2]) for d in dy])).all())}")
