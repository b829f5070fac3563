import sys; sys.path.insert(0, ".")
import torch
import oracle
from oracle.make_goldens import drift_stack
import torch_motion_correction_amd as mc
dev = torch.device("cuda:0")
st, dy, dx = drift_stack(4, 256, 256)
for shifts in ([[0.5, 0.25]] * 4, [[-3.7, 2.2]] * 4, [[3.0, 0.0], [-3.0, 0.0], [0.0, 3.0], [0.0, -3.0]], [[1.0, 1.0], [2.0, 2.0], [7.3, -9.1], [-12.6, 4.4]]):
    fld = torch.tensor(shifts).T.contiguous()[:, :, None, None]
    a = mc.correct_motion(st.to(dev), fld.to(dev), 1.0).cpu()
    b = oracle.correct_motion(st, fld, 1.0)
    for f in range(4):
        d = (a[f] - b[f]).abs()
        bad = d > 1e-3
        ys, xs = torch.nonzero(bad, as_tuple=True)
        print(shifts[f], 'max', float(d.max()), 'nbad', int(bad.sum()),
              'rows', (int(ys.min()), int(ys.max())) if len(ys) else None,
              'cols', (int(xs.min()), int(xs.max())) if len(xs) else None, 'zero frac got', float((a[f]==0).float().mean()), 'ref', float((b[f]==0).float().mean()))
