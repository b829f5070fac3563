import os, sys, torch
sys.path.insert(0, ".")
from torch_motion_correction_amd import engine, plan as planmod, _lib
from torch_motion_correction_amd._lib import ptr, stream_ptr, check
dev = torch.device("cuda:0")
t, h, w = 40, 4096, 4096
g = torch.Generator(device=dev).manual_seed(0)
stack = torch.randn(t, h, w, generator=g, device=dev)
pl = planmod.get_xc_plan(h, w, 1.0, 500.0, (300, 10), dev)
gm = pl.geom
lib = _lib.load()
stats = engine.central_box_stats(stack)
off = torch.arange(t, device=dev, dtype=torch.int64) * (h * w)
T1 = torch.empty((t, gm.nkx, gm.ny, 2), device=dev)
S = torch.empty((t, gm.nkx, gm.nky, 2), device=dev)
npairs = t - 1
cur = torch.tensor([f for f in range(t) if f != 20], device=dev, dtype=torch.int32)
ref = torch.full_like(cur, 20)
T2 = torch.empty((npairs, gm.nkx, gm.H, 2), device=dev)
ngrp = gm.H // gm.RG
pv = torch.empty(npairs * ngrp + npairs * gm.H, device=dev); pi = torch.empty(npairs * ngrp + npairs, device=dev, dtype=torch.int32)
peaks = torch.empty(npairs, device=dev, dtype=torch.int32); sh = torch.empty((npairs, 2), device=dev)
st = stream_ptr(dev)
def k0(): engine.central_box_stats(stack)
def k1(): check(lib.mc_xc_rows_forward(ptr(stack), ptr(off), w, None, ptr(pl.mask), ptr(stats), ptr(T1), ptr(pl.tw_row), t, gm, st), "k1")
def k2(): check(lib.mc_xc_cols_forward(ptr(T1), ptr(pl.filt), ptr(S), ptr(pl.tw_col), t, gm, st), "k2")
def k3(): check(lib.mc_xc_cols_inverse(ptr(S), ptr(cur), ptr(S), ptr(ref), ptr(T2), ptr(pl.tw_col), 1.0 / (h * w), npairs, gm, st), "k3")
def k4(): check(lib.mc_xc_rows_inverse_argmax(ptr(T2), ptr(pv), ptr(pi), ptr(peaks), ptr(sh), ptr(pl.tw_row), npairs, gm, st), "k4")
def tm(fn, n=5):
    for _ in range(2): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n
print("geom RG", gm.RG, "ny", gm.ny, "nkx", gm.nkx, "lds", lib.mc_xc_rows_lds_bytes(gm))
for name, fn in (("stats", k0), ("K1 rows_fwd", k1), ("K2 cols_fwd", k2), ("K3 cols_inv", k3), ("K4 rows_inv", k4)):
    print(f"{name:12s} {tm(fn):.3f} ms", flush=True)
def whole():
    engine.global_shifts(stack, 20, 1.0, 500.0, (300, 10))
print(f"global_shifts {tm(whole):.3f} ms")
