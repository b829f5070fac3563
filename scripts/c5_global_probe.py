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
