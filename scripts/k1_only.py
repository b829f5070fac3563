import os
"""K1 (rows forward, 40 x 4096 x 4096) alone, a few launches: for rocprofv3 passes."""
import sys, torch
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
off = torch.arange(t, device=dev, dtype=torch.int64) * (h * w)
T1 = torch.empty((t, gm.nkx, gm.ny, 2), device=dev)
hl, hu, wl, wu = 1024, 3072, 1024, 3072
m0 = torch.tensor([0.0, 1.0, 1.0], device=dev)
acc = torch.empty(128, dtype=torch.float64, device=dev); fix = torch.empty(2, device=dev); out3 = torch.empty(3, device=dev)
st = stream_ptr(dev)
def k1():
    check(lib.mc_xc_rows_forward_stats(ptr(stack), ptr(off), w, ptr(pl.mask), ptr(m0), ptr(T1), ptr(pl.tw_row), t, gm,
                                       hl, hu, wl, wu, ptr(acc), ptr(fix), ptr(out3),
                                       ptr(pl.chord) if os.environ.get("MC_ROW_CHORDS", "1") != "0" else None, st), "k1")
modes = [int(a) for a in sys.argv[1:]] or [0, 1]  # 0 = wave-per-row engine, 1 = workgroup-per-row
for rep in range(2):
    for mode in modes:
        check(lib.mc_xc_row_engine(mode), "engine")
        for _ in range(3): k1()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10): k1()
        e1.record(); torch.cuda.synchronize()
        print(f"engine mode {mode}: K1 {e0.elapsed_time(e1) / 10:.3f} ms", flush=True)
