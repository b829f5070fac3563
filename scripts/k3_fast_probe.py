"""K3-format whole-frame functions: estimate_global_motion, correct_motion_fast, dose_weighted_sum at
40 x 4092 x 5760 and (12 frames) 8184 x 11520."""
import sys, torch
sys.path.insert(0, ".")
import bench
import torch_motion_correction_amd as mc

dev = torch.device("cuda:0")


def timeit(fn, n=3):
    fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / n


for (t, h, w) in ((40, 4092, 5760), (12, 8184, 11520)):
    st, dy, dx = bench.synth_stack(t, h, w, 3, dev)
    field = mc.estimate_global_motion(st, 1.0)
    ok = bool((field[0, :, 0, 0].cpu() - field[0, 0, 0, 0].cpu()).abs().max() > 0)
    tg = timeit(lambda: mc.estimate_global_motion(st, 1.0))
    tf = timeit(lambda: mc.correct_motion_fast(st, field.clone()))
    td = timeit(lambda: mc.dose_weighted_sum(st, 1.0, 1.0))
    print(f"{t} x {h} x {w}: estimate_global_motion {tg:7.2f} ms  correct_motion_fast {tf:7.2f} ms  "
          f"dose_weighted_sum {td:7.2f} ms  (shifts vary: {ok})", flush=True)
    del st
