# estimate_local_motion at the C3 frame size: one-time spectra + per-iteration cost
import sys, time, torch
sys.path.insert(0, ".")
import bench
import torch_motion_correction_amd as mc
from torch_motion_correction_amd import local_motion
dev = torch.device("cuda:0")
t, h, w = 40, 4092, 5760
stack, dy, dx = bench.synth_stack(t, h, w, 5, dev)
for rep in range(2):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    prob = local_motion.LocalMotionProblem(stack, 1.0, (1024, 1024), (t, 6, 10), "catmull_rom")
    torch.cuda.synchronize(); t1 = time.perf_counter()
    print(f"setup (60 x 40 patch spectra, basis): {1e3*(t1-t0):.1f} ms; spectra {prob.spectra.numel()*4/1e9:.2f} GB, bins {prob.nkx}x{prob.nky}", flush=True)
new = torch.zeros(2, t, 6, 10, device=dev, requires_grad=True)
init = torch.zeros(2, t, 6, 10, device=dev)
wb = torch.full((prob.npatch,), 1 / 8, dtype=torch.float64, device=dev)
for lt in ("mse", "cc", "ncc"):
    for _ in range(3):
        local_motion._Loss.apply(prob.shifts_px(new, init), prob, wb, lt).backward()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(20):
        local_motion._Loss.apply(prob.shifts_px(new, init), prob, wb, lt).backward()
    torch.cuda.synchronize(); t1 = time.perf_counter()
    print(f"{lt}: loss + gradient {1e3*(t1-t0)/20:.3f} ms/iteration", flush=True)
torch.cuda.synchronize(); t0 = time.perf_counter()
f = mc.estimate_local_motion(stack, 1.0, (1024, 1024), (t, 6, 10), None, n_iterations=100)
torch.cuda.synchronize(); t1 = time.perf_counter()
print(f"estimate_local_motion, 100 adam iterations: {1e3*(t1-t0):.1f} ms total; field range {float(f.min()):.3f}..{float(f.max()):.3f}")
opt = local_motion.setup_optimizer("adam", [new])
def timeit(fn, n=30):
    for _ in range(3): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return 1e3 * (time.perf_counter() - t0) / n
wb8 = wb
print("shifts_px only", timeit(lambda: prob.shifts_px(new, init)))
print("loss fwd only", timeit(lambda: prob.loss_and_grad(prob.shifts_px(new, init).detach(), wb8, "mse")))
def full():
    l = local_motion._Loss.apply(prob.shifts_px(new, init), prob, wb8, "mse"); l.backward(); opt.step(); opt.zero_grad()
print("full iteration", timeit(full))
def nostep():
    l = local_motion._Loss.apply(prob.shifts_px(new, init), prob, wb8, "mse"); l.backward(); new.grad = None
print("no optimiser step", timeit(nostep))
for rep in range(2):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    f = mc.estimate_local_motion(stack, 1.0, (1024, 1024), (t, 6, 10), None, n_iterations=100)
    torch.cuda.synchronize(); t1 = time.perf_counter()
    print(f"estimate_local_motion again, 100 adam iterations: {1e3*(t1-t0):.1f} ms total")
