import sys; sys.path.insert(0, ".")
import ctypes as C, torch
from torch_motion_correction_amd import _lib
from torch_motion_correction_amd._lib import ptr, stream_ptr, check
dev = torch.device("cuda:0")
lib = _lib.load()
t, h, w = 1, 64, 256
img = torch.randn(t, h, w, device=dev)
for sh in ([0.5, 0.0], [0.0, 0.5], [0.5, 0.5], [1.0, 0.0], [1.5, 0.0], [0.001, 0.0], [0.0, 1.25]):
    shifts = torch.tensor([sh], device=dev)
    nb = C.c_int64(0); lib.mc_warp_rigid_scratch_bytes(t, h, w, C.byref(nb))
    scratch = torch.zeros((nb.value + 3) // 4, dtype=torch.float32, device=dev)
    out = torch.full_like(img, 7.0)
    check(lib.mc_warp_rigid(ptr(img), t, h, w, ptr(shifts), ptr(scratch), ptr(out), None, stream_ptr(dev)), "rigid")
    torch.cuda.synchronize()
    S = scratch[t * 5 * (h + w) : t * 5 * (h + w) + 2 * t].view(torch.int32).cpu()
    print(sh, "S", S.tolist(), "zero frac", float((out == 0).float().mean()), "seven frac", float((out == 7).float().mean()), "sample", out[0, 20, 20:23].tolist())
