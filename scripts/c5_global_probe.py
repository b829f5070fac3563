# estimate_global_motion on super-resolution frames (8184 x 11520): chirp-z lines of 16384 points
import sys, time, torch
sys.path.insert(0, ".")
import bench
import torch_motion_correction_amd as mc
dev = torch.device("cuda:0")
t, h, w = int(sys.argv[1]) if len(sys.argv) > 1 else 8, 8184, 11520
stack, dy, dx = bench.synth_stack(t, h, w, 5, dev)
for it in range(2):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    field = mc.estimate_global_motion(stack, 1.0)
    torch.cuda.synchronize(); t1 = time.perf_counter()
    print(f"iter {it}: estimate_global_motion {t} x {h} x {w}: {1e3*(t1-t0):.1f} ms", flush=True)
ref = t // 2
ok = field[0, :, 0, 0].cpu().tolist() == [float(d - dy[ref]) for d in dy] and field[1, :, 0, 0].cpu().tolist() == [float(d - dx[ref]) for d in dx]
print("shifts match known drift:", ok, " peak mem GB", torch.cuda.max_memory_allocated() / 1e9)
# correct_motion_fast on the same frames: x-polyphase Fourier shift; integer shifts = circular roll
sub = stack[:4]
fld = torch.zeros(2, 4, 1, 1, device=dev)
fld[0, :, 0, 0] = torch.tensor([3.0, -2.0, 0.0, 5.0]); fld[1, :, 0, 0] = torch.tensor([-4.0, 1.0, 7.0, 0.0])
for it in range(2):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    out = mc.correct_motion_fast(sub, fld.clone())
    torch.cuda.synchronize(); t1 = time.perf_counter()
    print(f"iter {it}: correct_motion_fast 4 x {h} x {w}: {1e3*(t1-t0):.1f} ms", flush=True)
err = 0.0
for f in range(4):  # the function shifts by -field (Q1)
    want = torch.roll(sub[f], shifts=(-int(fld[0, f, 0, 0]), -int(fld[1, f, 0, 0])), dims=(0, 1))
    err = max(err, float((out[f] - want).abs().max() / want.abs().max()))
print("max rel. deviation from the circular roll:", err)
f2, _ = mc.estimate_motion_cross_correlation_patches(stack, 1.0, patch_sidelength=1024, deformation_field=field)
print("patch estimate with the global prior:", tuple(f2.shape), bool(torch.isfinite(f2).all()))
