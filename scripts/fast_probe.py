"""correct_motion_fast (a19) and the global estimate at the C2 and K3 frame sizes."""
import sys, time, torch
sys.path.insert(0, ".")
import bench
import torch_motion_correction_amd as mc
dev = torch.device("cuda:0")
def timed(fn, n=3):
    for _ in range(2): r = fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): r = fn()
    torch.cuda.synchronize()
    return 1e3 * (time.perf_counter() - t0) / n
for (t, h, w) in ((40, 4096, 4096), (40, 4092, 5760)):
    st, dy, dx = bench.synth_stack(t, h, w, 3, dev)
    field = mc.estimate_global_motion(st, 1.0)
    ok = field[0, :, 0, 0].cpu().tolist() == [float(d - dy[t // 2]) for d in dy]
    e = timed(lambda: mc.estimate_global_motion(st, 1.0))
    f = timed(lambda: mc.correct_motion_fast(st, field.clone()))
    c = timed(lambda: mc.correct_motion(st, field, 1.0))
    print(f"{t}x{h}x{w}: estimate_global_motion {e:.2f} ms (drift recovered: {ok}), correct_motion_fast {f:.2f} ms, correct_motion (rigid) {c:.2f} ms", flush=True)
    del st
