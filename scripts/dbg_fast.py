import sys; sys.path.insert(0, ".")
import torch, oracle
import torch_motion_correction_amd as mc
dev = torch.device("cuda:0")
for shape in [(2, 64, 64), (2, 100, 64), (2, 64, 90), (2, 96, 120)]:
    g = torch.Generator().manual_seed(sum(shape))
    img = torch.randn(*shape, generator=g)
    for sh in ([[0., 0.]] * 2, [[1., 0.]] * 2, [[0., 1.]] * 2, [[0.5, 0.25]] * 2):
        fld = -torch.tensor(sh).T.contiguous()[:, :, None, None]
        a = mc.correct_motion_fast(img.to(dev), fld.clone().to(dev)).cpu()
        b = oracle.correct_motion_fast(img, fld.clone())
        print(shape, sh[0], float((a - b).abs().max() / b.abs().max()))
