# random frame sizes (odd / even / powers of two) through the whole-frame functions against the oracle
import sys, random, torch
sys.path.insert(0, ".")
import oracle
from oracle.make_goldens import drift_stack
import torch_motion_correction_amd as mc
dev = torch.device("cuda:0")
random.seed(int(sys.argv[1]) if len(sys.argv) > 1 else 0)
bad = 0
special = [8, 16, 32, 64, 128, 256, 12, 33, 63, 65, 127, 129, 255, 257, 100, 250]
for it in range(int(sys.argv[2]) if len(sys.argv) > 2 else 30):
    h = random.choice(special + [random.randint(10, 300)])
    w = random.choice(special + [random.randint(10, 300)])
    t = random.randint(2, 5)
    try:
        st, _, _ = drift_stack(t, h, w, seed=it)
        got = mc.estimate_global_motion(st.to(dev), 1.0).cpu()
        ref, ccs = oracle.estimate_global_motion(st, 1.0, return_cc=True)
        ok = True
        for f, cc in ccs.items():
            top = torch.topk(cc.flatten(), 2).values
            if float(top[0] - top[1]) > 1e-5 * float(top[0].abs()) and not torch.equal(got[:, f], ref[:, f]):
                ok = False
        sh = torch.randn(2, t, 1, 1) * 3
        a = mc.correct_motion_fast(st.to(dev), sh.clone().to(dev)).cpu()
        b = oracle.correct_motion_fast(st, sh.clone())
        e1 = float((a - b).abs().max() / b.abs().max())
        fld = torch.randn(2, t, 2, 2) * 2
        c = mc.correct_motion(st.to(dev), fld.to(dev), 1.0).cpu()
        d = oracle.correct_motion(st, fld, 1.0)
        e2 = float((c - d).abs().median() / d.abs().max())
        if not ok or e1 > 3e-5 or e2 > 1e-5:
            bad += 1
            print("MISMATCH", (t, h, w), ok, e1, e2, flush=True)
    except Exception as e:
        bad += 1
        print("EXCEPTION", (t, h, w), type(e).__name__, str(e)[:150], flush=True)
print("done, bad =", bad)
