import sys; sys.path.insert(0, ".")
import torch, numpy as np
import oracle
from oracle.make_goldens import drift_stack
import torch_motion_correction_amd as mc
dev = torch.device("cuda:0")
st, dy, dx = drift_stack(8, 256, 256)
o = oracle.estimate_global_motion(st, 1.0)
a = mc.correct_motion(st.to(dev), o.to(dev), 1.0).cpu()
b = oracle.correct_motion(st, o, 1.0)
for f in range(8):
    d = (a[f] - b[f]).abs()
    bad = d > 1e-3
    ys, xs = torch.nonzero(bad, as_tuple=True)
    print(f, 'shift', o[:, f, 0, 0].tolist(), 'max', float(d.max()), 'nbad', int(bad.sum()),
          'rows', (int(ys.min()), int(ys.max())) if len(ys) else None,
          'cols', (int(xs.min()), int(xs.max())) if len(xs) else None)
    if len(ys):
        y, x = int(ys[0]), int(xs[0])
        print('   first bad', y, x, float(a[f, y, x]), float(b[f, y, x]))
        # which source pixel does a equal?
        v = a[f, y, x]
        m = (st[f] - v).abs() < 1e-6
        print('   a equals src at', torch.nonzero(m)[:4].tolist(), ' b equals src at', torch.nonzero((st[f]-b[f,y,x]).abs()<1e-6)[:4].tolist())
