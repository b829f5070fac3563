# EXPERIMENT: partition the CUs between the estimator's and the warp's stream
# (hipExtStreamCreateWithCUMask) instead of letting both compete for every CU.
import ctypes, sys, time, os, torch
sys.path.insert(0, ".")
import bench
from torch_motion_correction_amd import pipeline
dev = torch.device("cuda:0")
torch.cuda.set_device(dev)
hip = ctypes.CDLL("libamdhip64.so")
def masked_stream(pred):
    words = (ctypes.c_uint32 * 8)()
    n = 0
    for i in range(256):
        if pred(i):
            words[i // 32] |= (1 << (i % 32)); n += 1
    s = ctypes.c_void_p()
    rc = hip.hipExtStreamCreateWithCUMask(ctypes.byref(s), 8, words)
    assert rc == 0, rc
    return torch.cuda.ExternalStream(s.value, dev), n
t, h, w = 40, 4096, 4096
stack, dy, dx = bench.synth_stack(t, h, w, 1234, dev)
def run(pipe, n=20):
    for _ in pipe.iterate([stack] * 3): pass
    torch.cuda.synchronize(); t0 = time.perf_counter()
    last = None
    for r in pipe.iterate([stack] * n): last = r
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    ok = bool((last.field[0, :, 0, 0].cpu() == torch.tensor([float(d - dy[t // 2]) for d in dy])).all())
    return n * t / dt, ok
base = pipeline.MoviePipeline(dev, 1.0, t // 2, 500.0, (300, 10), "catmull_rom", return_frames=True)
print("no masks:", run(base), flush=True)
for name, est_pred in (("est 1/4 interleaved", lambda i: i % 4 == 0), ("est 3/8 interleaved", lambda i: i % 8 in (0, 3, 6)),
                       ("est 1/2 interleaved", lambda i: i % 2 == 0), ("est 1/8 interleaved", lambda i: i % 8 == 0),
                       ("est all, warp 3/4", None)):
    pipe = pipeline.MoviePipeline(dev, 1.0, t // 2, 500.0, (300, 10), "catmull_rom", return_frames=True)
    if est_pred is None:
        pipe._s_warp, nw = masked_stream(lambda i: i % 4 != 0); ne = 256
    else:
        pipe._s_est, ne = masked_stream(est_pred)
        pipe._s_warp, nw = masked_stream(lambda i: not est_pred(i))
    print(f"{name}: est {ne} CUs, warp {nw} CUs:", run(pipe), flush=True)
print("no masks again:", run(base))
