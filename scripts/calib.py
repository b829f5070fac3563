import torch, time
dev = torch.device("cuda:0")
t, h, w = 40, 4096, 4096
x = torch.randn(t, h, w, device=dev)
y = torch.empty_like(x)
def tm(fn, n=5):
    for _ in range(2): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n
gb = x.numel() * 4 / 1e9
ms = tm(lambda: y.copy_(x)); print(f"copy_  {ms:.3f} ms  {2*gb/ms*1e3:.0f} GB/s (r+w)")
ms = tm(lambda: x.sum(0)); print(f"sum(0) {ms:.3f} ms  {gb/ms*1e3:.0f} GB/s (read)")
ms = tm(lambda: torch.mul(x, 2.0, out=y)); print(f"mul    {ms:.3f} ms  {2*gb/ms*1e3:.0f} GB/s (r+w)")
ms = tm(lambda: y.zero_()); print(f"zero   {ms:.3f} ms  {gb/ms*1e3:.0f} GB/s (write)")
ms = tm(lambda: x.sum()); print(f"sum()  {ms:.3f} ms  {gb/ms*1e3:.0f} GB/s (read)")
