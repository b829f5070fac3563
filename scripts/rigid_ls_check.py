"""A/B of the rigid fused warp kernels: run once per MC_RIGID_LS value; 'save' writes outputs, 'cmp' compares
with the saved ones bit for bit.  usage: MC_RIGID_LS=0 python scripts/rigid_ls_check.py save; MC_RIGID_LS=1 ... cmp"""
import os, sys, torch
sys.path.insert(0, ".")
from torch_motion_correction_amd import engine
import torch_motion_correction_amd as mc
dev = torch.device("cuda:0")
mode = sys.argv[1]
cases = [(6, 256, 256, 3.0), (5, 300, 260, 9.0), (4, 1000, 1028, 6.5), (3, 64, 64, 2.0), (3, 70, 516, 40.0),
         (7, 1024, 1024, 150.0), (8, 2048, 4096, 8.0), (5, 4092, 5760, 7.0)]
ok = True
for ci, (t, h, w, amp) in enumerate(cases):
    g = torch.Generator(device=dev).manual_seed(100 + ci)
    stack = torch.randn(t, h, w, generator=g, device=dev)
    sh = (torch.rand(t, 2, generator=g, device=dev) * 2 - 1) * amp
    sh[0] = torch.round(sh[0])          # an integer shift (knife-edge weights)
    sh[-1, 0] = 0.0
    field = mc.image_shifts_to_deformation_field(sh, 1.0).contiguous()
    lat = engine.frame_lattices(field, t, "catmull_rom")
    frames, total = engine.warp(stack, lat, 1.0, want_frames=True, want_sum=True, rigid=True)
    torch.cuda.synchronize()
    path = f"/tmp/rigid_ab_{ci}.pt"
    if mode == "save":
        torch.save((frames.cpu(), total.cpu()), path)
        print("saved", (t, h, w), float(frames.abs().max()), float(total.abs().max()))
    else:
        f0, s0 = torch.load(path)
        ef = (frames.cpu() - f0).abs().max().item()
        es = (total.cpu() - s0).abs().max().item()
        same = torch.equal(frames.cpu(), f0) and torch.equal(total.cpu(), s0)
        ok &= same
        print((t, h, w), "bit-equal" if same else f"DIFF frames {ef:.3e} sum {es:.3e}")
if mode == "cmp":
    print("ALL EQUAL" if ok else "MISMATCH")
    sys.exit(0 if ok else 1)
