"""C3 / C5-style patch estimate timing (1024-px patches), fp32 and fp16 storage."""
import sys, time, torch
sys.path.insert(0, ".")
import bench
import torch_motion_correction_amd as mc
dev = torch.device("cuda:0")
st, tex = bench.synth_local_motion_stack(mc, 40, 4092, 5760, 6, 10, 7, dev)
for s, name in ((st, "fp32"), (st.half(), "fp16")):
    ts = []
    for _ in range(4):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        f, _ = mc.estimate_motion_cross_correlation_patches(s, 1.0, patch_sidelength=1024)
        torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
    print(f"40 x 4092 x 5760 {name}: patch estimate {1e3 * min(ts[1:]):.2f} ms", flush=True)
