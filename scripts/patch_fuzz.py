# random patch sizes / strategies / options through the patch estimator against the oracle
import sys, random, torch
sys.path.insert(0, ".")
import oracle
from oracle.make_goldens import drift_stack
import torch_motion_correction_amd as mc
dev = torch.device("cuda:0")
random.seed(int(sys.argv[1]) if len(sys.argv) > 1 else 0)
bad = 0
for it in range(int(sys.argv[2]) if len(sys.argv) > 2 else 20):
    t = random.randint(3, 7)
    h, w = random.randint(100, 220), random.randint(100, 220)
    p = random.choice([32, 64, 48, 40, 63, 55, 80, 96])
    kw = dict(patch_sidelength=p, reference_strategy=random.choice(["mean_except_current", "middle_frame"]),
              temporal_smoothing=random.choice([True, False]), outlier_rejection=random.choice([True, False]),
              smoothing_window_size=random.choice([3, 5]))
    try:
        st, _, _ = drift_stack(t, h, w, seed=100 + it)
        got, gc = mc.estimate_motion_cross_correlation_patches(st.to(dev), 1.1, **kw)
        ref, rc = oracle.estimate_motion_cross_correlation_patches(st, 1.1, **kw)
        err = float((got.cpu() - ref).abs().max())
        if not torch.equal(gc.cpu(), rc) or err > 2e-4:
            bad += 1
            print("MISMATCH", (t, h, w), kw, err, flush=True)
    except Exception as e:
        bad += 1
        print("EXCEPTION", (t, h, w), kw, type(e).__name__, str(e)[:150], flush=True)
print("done, bad =", bad)
