# random patch shapes / grid resolutions / losses: HIP loss + gradient of estimate_local_motion against
# the oracle's autograd
import sys, random, numpy as np, torch
sys.path.insert(0, ".")
import oracle
from oracle.make_goldens import drift_stack
from torch_motion_correction_amd import local_motion
dev = torch.device("cuda:0")
random.seed(int(sys.argv[1]) if len(sys.argv) > 1 else 0)
bad = 0
for it in range(int(sys.argv[2]) if len(sys.argv) > 2 else 12):
    t = random.randint(3, 6)
    h, w = random.randint(90, 160), random.randint(90, 160)
    patch = (random.choice([32, 40, 48, 33, 64]), random.choice([32, 40, 48, 47, 64]))
    # nt >= 2: a time-constant grid moves all frames of a patch alike, the loss does not depend on it and
    # both gradients are rounding noise around an exact zero
    res = (random.choice([2, 3, t]), random.randint(1, 3), random.randint(1, 3))
    gt = random.choice(["catmull_rom", "bspline"])
    lt = random.choice(["mse", "cc", "ncc"])
    ps = random.choice([1.0, 0.83, 1.7])
    try:
        st, _, _ = drift_stack(t, h, w, seed=200 + it)
        g = torch.Generator().manual_seed(it)
        new = (torch.randn(2, *res, generator=g) * 1.2).requires_grad_(True)
        init = torch.randn(2, *res, generator=g) * 0.5
        op = oracle.LocalMotionProblem(st, ps, patch)
        tot = None
        for a in range(0, op.npatch, 8):
            l = op.batch_loss(new, init, gt, list(range(a, min(a + 8, op.npatch))), lt)
            tot = l if tot is None else tot + l
        tot.backward()
        pp = local_motion.LocalMotionProblem(st.to(dev), ps, patch, res, gt)
        sizes = np.minimum(8, pp.npatch - (np.arange(pp.npatch) // 8) * 8)
        wb = torch.from_numpy(1.0 / sizes.astype(np.float64)).to(dev)
        nd = new.detach().to(dev).requires_grad_(True)
        loss = local_motion._Loss.apply(pp.shifts_px(nd, init.to(dev)), pp, wb, lt)
        loss.backward()
        e0 = abs(loss.item() - tot.item()) / max(abs(tot.item()), 1e-30)
        # a grid with a single node per axis moves every frame alike: the true gradient is zero and
        # both sides hold rounding noise, so errors are measured against the loss scale as well
        e1 = float((nd.grad.cpu() - new.grad).abs().max() / max(float(new.grad.abs().max()), 1e-4 * abs(tot.item())))
        if e0 > 3e-4 or e1 > 3e-4:
            bad += 1
            print("MISMATCH", (t, h, w), patch, res, gt, lt, ps, e0, e1, flush=True)
    except Exception as e:
        bad += 1
        print("EXCEPTION", (t, h, w), patch, res, gt, lt, ps, type(e).__name__, str(e)[:160], flush=True)
print("done, bad =", bad)
