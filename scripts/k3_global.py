import sys, time, torch
sys.path.insert(0, ".")
import bench
import torch_motion_correction_amd as mc
dev = torch.device("cuda:0")
t, h, w = 40, 4092, 5760
stack, dy, dx = bench.synth_stack(t, h, w, 5, dev)
for it in range(3):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    field = mc.estimate_global_motion(stack, 1.0)
    torch.cuda.synchronize(); t1 = time.perf_counter()
    total = mc.motion_correct_sum(stack, field, 1.0)
    torch.cuda.synchronize(); t2 = time.perf_counter()
    print(f"iter {it}: estimate {1e3*(t1-t0):.1f} ms, correct+sum {1e3*(t2-t1):.1f} ms", flush=True)
ok = field[0, :, 0, 0].cpu().tolist() == [float(d - dy[20]) for d in dy] and field[1, :, 0, 0].cpu().tolist() == [float(d - dx[20]) for d in dx]
print("shifts match known drift:", ok, " sum std:", float(total[64:-64, 64:-64].std()))
