"""Does data that fits the 256 MiB Infinity Cache move faster than HBM-resident data?  copy / read-only
sweeps of buffers of growing size, repeated back to back."""
import torch
dev = torch.device("cuda:0")
def timeit(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n
for mb in (8, 16, 32, 64, 96, 128, 192, 256, 512, 1024, 2048):
    n = mb * (1 << 20) // 4
    a = torch.randn(n, device=dev); b = torch.empty_like(a)
    t_copy = timeit(lambda: b.copy_(a))
    t_read = timeit(lambda: torch.max(a))  # read-only reduction
    t_fill = timeit(lambda: b.fill_(1.0))
    print(f"{mb:5d} MiB  copy {2*mb/1024/t_copy*1e3*1.048576:7.2f} GB/s (r+w)   max {mb/1024/t_read*1e3*1.048576:7.2f} GB/s (r)   fill {mb/1024/t_fill*1e3*1.048576:7.2f} GB/s (w)", flush=True)
