"""Exposure-weighted sum: column pass fed from the row-major spectra vs from a column-major copy."""
import sys, torch
sys.path.insert(0, ".")
import bench
import torch_motion_correction_amd as mc
from torch_motion_correction_amd import engine
dev = torch.device("cuda:0")


def timeit(fn, n=3):
    fn(); torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n):
        r = fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n, r


for (t, h, w) in ((40, 4096, 4096), (40, 4092, 5760), (12, 8184, 11520)):
    st, _, _ = bench.synth_stack(t, h, w, 3, dev)
    res = {}
    for cm in (False, True):
        engine.DOSE_COLUMN_MAJOR = cm
        ms, out = timeit(lambda: mc.dose_weighted_sum(st, 1.0, 1.0))
        res[cm] = (ms, out)
    d = float((res[True][1] - res[False][1]).abs().max() / res[False][1].abs().max())
    print(f"{t} x {h} x {w}: row-major {res[False][0]:.2f} ms, column-major copy {res[True][0]:.2f} ms, rel diff {d:.1e}", flush=True)
    del st, res
