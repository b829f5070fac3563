"""Does overlapping the estimator of movie k+1 with the warp of movie k (two streams) pay?"""
import sys, time, torch
sys.path.insert(0, ".")
import bench
import torch_motion_correction_amd as mc
from torch_motion_correction_amd import engine
dev = torch.device("cuda:0")
t, h, w = 40, 4096, 4096
stack, dy, dx = bench.synth_stack(t, h, w, 1234, dev)
ref = t // 2
def est():
    return engine.global_shifts(stack, ref, 1.0, 500.0, (300, 10))
def cor(shifts):
    field = mc.image_shifts_to_deformation_field(shifts, 1.0)
    lat = engine.frame_lattices(field.contiguous(), t, "catmull_rom")
    return engine.warp(stack, lat, 1.0, want_frames=True, want_sum=True, rigid=True)
def serial(n):
    for _ in range(n):
        out = cor(est())
    return out
pri = int(sys.argv[1]) if len(sys.argv) > 1 else 0
s_est, s_warp = torch.cuda.Stream(priority=-1 if pri == 1 else 0), torch.cuda.Stream(priority=-1 if pri == 2 else 0)
def piped(n):
    out = None
    for _ in range(n):
        with torch.cuda.stream(s_est):
            sh = est()
            ev = torch.cuda.Event(); ev.record(s_est)
        with torch.cuda.stream(s_warp):
            s_warp.wait_event(ev)
            sh.record_stream(s_warp)
            out = cor(sh)
    return out
for name, fn in (("serial", serial), ("two streams", piped), ("serial", serial), ("two streams", piped)):
    fn(3); torch.cuda.synchronize()
    t0 = time.perf_counter(); out = fn(20); torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 20
    print(f"{name:12s} {dt*1e3:.3f} ms/movie  {t/dt:.0f} frames/s  sum std {float(out[1].std()):.3f}", flush=True)
