"""How far the branch-and-bound of the shift search is from opening its fall-back on the bench movie: for every pair
max over the far rows of bound(y) against the maximum found in the near window (mc_xc_correlate_argmax)."""
import sys, numpy as np, torch
sys.path.insert(0, ".")
import bench
from torch_motion_correction_amd import engine, plan as planmod, _lib
from torch_motion_correction_amd._lib import ptr, stream_ptr, check
dev = torch.device("cuda:0")
t, h, w = 40, 4096, 4096
noise = float(sys.argv[1]) if len(sys.argv) > 1 else 1.0
stack, dy, dx = bench.synth_stack(t, h, w, 1234, dev, noise=noise)
pl = planmod.get_xc_plan(h, w, 1.0, 500.0, (300, 10), dev)
g = pl.geom
lib = _lib.load()
S = engine._global_spectra(stack, pl)
ref = t // 2
cur = torch.tensor([f for f in range(t) if f != ref], device=dev, dtype=torch.int32)
rf = torch.full_like(cur, ref)
npairs = t - 1
ngrp = g.H // g.RG
T2 = torch.empty((npairs, g.nkx, g.H, 2), device=dev)
T2n = torch.empty((npairs, g.nkx, 2 * lib.mc_xc_near_rows(g), 2), device=dev)
pv = torch.empty(npairs * ngrp + npairs * g.H, device=dev)
pi = torch.empty(npairs * ngrp + npairs + 1, device=dev, dtype=torch.int32)
peaks = torch.empty(npairs, device=dev, dtype=torch.int32)
sh = torch.empty((npairs, 2), device=dev)
check(lib.mc_xc_correlate_argmax(ptr(S), ptr(cur), ptr(S), ptr(rf), ptr(T2), ptr(T2n), ptr(pv), ptr(pi), ptr(peaks), ptr(sh),
                                 None, 0, None, ptr(pl.tw_col), ptr(pl.tw_row), 1.0 / (h * w), npairs, g, stream_ptr(dev)), "xc")
torch.cuda.synchronize()
bounds = pv[npairs * ngrp:].view(npairs, g.H).cpu().numpy()
order = pi[npairs * ngrp: npairs * ngrp + npairs].cpu().numpy().astype(np.int64)
bits = np.where(order >= 0, order, order ^ 0x7fffffff).astype(np.int32)
best = bits.view(np.float32)
near = lib.mc_xc_near_rows(g) - 8
far = bounds[:, near: g.H - near].max(axis=1)
print(f"noise {noise}: gate {int(pi[npairs * ngrp + npairs])}, near rows {near}, max far bound / best: "
      f"min {float((far / best).min()):.3f} median {float(np.median(far / best)):.3f} max {float((far / best).max()):.3f}")
print("shifts ok:", sh[:3].tolist())
