"""Three-stage schedule probe: stream A: K1(k+1), warp(k);  stream B: K2..K5(k+1) under warp(k)."""
import sys, time, torch
sys.path.insert(0, ".")
import bench
import torch_motion_correction_amd as mc
from torch_motion_correction_amd import engine, plan as planmod, _lib
from torch_motion_correction_amd._lib import check, ptr, stream_ptr
dev = torch.device("cuda:0")
t, h, w = 40, 4096, 4096
stack, dy, dx = bench.synth_stack(t, h, w, 1234, dev)
ref = t // 2
lib = _lib.load()
pl = planmod.get_xc_plan(h, w, 1.0, 500.0, (300, 10), dev)
g = pl.geom
mhat = engine.mask_spectrum(pl, dev)
hl, hu, wl, wu = h // 4, 3 * h // 4, w // 4, 3 * w // 4
job_off = torch.arange(t, device=dev, dtype=torch.int64) * (h * w)
cur = [f for f in range(t) if f != ref]
cur_idx = torch.tensor(cur, dtype=torch.int32, device=dev)
ref_idx = torch.full_like(cur_idx, ref)
scatter = torch.tensor([cur.index(f) if f != ref else len(cur) for f in range(t)], device=dev)
torch.cuda.synchronize()

def stage1():  # K1 (+ provisional mean)
    st = stream_ptr(dev)
    acc = torch.empty(128, dtype=torch.float64, device=dev)
    m0 = torch.ones(3, dtype=torch.float32, device=dev)
    check(lib.mc_central_box_stats(ptr(stack), 1, h, w, hl, hl + 1, wl, wu, ptr(acc), ptr(m0), st), "m0")
    m0[1:].fill_(1.0)
    fix = torch.empty(2, device=dev); out3 = torch.empty(3, device=dev)
    T1 = torch.empty((t, g.nkx, g.ny, 2), device=dev)
    check(lib.mc_xc_rows_forward_stats(ptr(stack), ptr(job_off), w, ptr(pl.mask), ptr(m0), ptr(T1), ptr(pl.tw_row), t, g,
                                       hl, hu, wl, wu, ptr(acc), ptr(fix), ptr(out3), ptr(pl.chord), st), "k1")
    return T1, fix

def stage2(T1, fix):  # K2 .. K5
    st = stream_ptr(dev)
    S = torch.empty((t, g.nkx, g.nky, 2), device=dev)
    check(lib.mc_xc_cols_forward_fix(ptr(T1), ptr(pl.filt), ptr(S), ptr(pl.tw_col), t, g, ptr(fix), ptr(mhat), st), "k2")
    _, shifts, _ = engine._peaks(S, cur_idx, S, ref_idx, pl, want_nbhd=False)
    padded = torch.cat([shifts, shifts.new_zeros((1, 2))], dim=0)
    return padded.index_select(0, scatter)

def stage3(shifts):
    field = mc.image_shifts_to_deformation_field(shifts, 1.0)
    lat = engine.frame_lattices(field.contiguous(), t, "catmull_rom")
    return engine.warp(stack, lat, 1.0, want_frames=True, want_sum=True, rigid=True)

def serial(n):
    for _ in range(n):
        T1, fix = stage1(); sh = stage2(T1, fix); out = stage3(sh)
    return out

pri = int(sys.argv[1]) if len(sys.argv) > 1 else 0
sA, sB = torch.cuda.Stream(priority=0), torch.cuda.Stream(priority=-1 if pri else 0)
def piped(n):
    out = None
    prev = None  # (shifts, event) of the movie whose warp is pending
    for i in range(n + 1):
        if i < n:
            with torch.cuda.stream(sA):
                T1, fix = stage1()
                e1 = torch.cuda.Event(); e1.record(sA)
        if prev is not None:
            with torch.cuda.stream(sA):
                sA.wait_event(prev[1])
                out = stage3(prev[0])
        if i < n:
            with torch.cuda.stream(sB):
                sB.wait_event(e1)
                T1.record_stream(sB); fix.record_stream(sB)
                sh = stage2(T1, fix)
                e2 = torch.cuda.Event(); e2.record(sB)
                sh.record_stream(sA)
            prev = (sh, e2)
    return out
for name, fn in (("serial", serial), ("3-stage", piped), ("serial", serial), ("3-stage", piped)):
    fn(3); torch.cuda.synchronize()
    t0 = time.perf_counter(); out = fn(20); torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 20
    print(f"{name:12s} {dt*1e3:.3f} ms/movie  {t/dt:.0f} frames/s  sum std {float(out[1].std()):.3f}", flush=True)
