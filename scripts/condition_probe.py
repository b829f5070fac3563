# condition_movie (gain x raw - frame mean) at the C2 size for the storage types it takes
import sys, time, torch
sys.path.insert(0, ".")
import torch_motion_correction_amd as mc
dev = torch.device("cuda:0")
t, h, w = 40, 4096, 4096
gain = (1.0 + 0.05 * torch.randn(h, w, device=dev)).contiguous()
for dt in (torch.uint8, torch.int16, torch.float16, torch.float32):
    raw = (torch.rand(t, h, w, device=dev) * 20).to(dt)
    for _ in range(2):
        out = mc.condition_movie(raw, gain)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(5):
        out = mc.condition_movie(raw, gain)
    torch.cuda.synchronize(); ms = 1e3 * (time.perf_counter() - t0) / 5
    b = raw.element_size()
    print(f"{str(dt):14s}: {ms:.3f} ms per stack  ({(2 * b + 4) * t * h * w / ms / 1e6:.0f} GB/s of 2 reads + 1 write)", flush=True)
    del raw, out
