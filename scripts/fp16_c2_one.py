"""C2 steps on fp16-stored stacks (K1 and the rigid warp read the 16-bit samples), one stream."""
import sys, torch
sys.path.insert(0, ".")
import bench
import torch_motion_correction_amd as mc
dev = torch.device("cuda:0")
st, dy, dx = bench.synth_stack(40, 4096, 4096, 3, dev)
st16 = st.half()
for s in (st16, st):
    for _ in range(3):
        f = mc.estimate_global_motion(s, 1.0)
        mc.motion_correct_sum(s, f, 1.0, return_frames=True)
torch.cuda.synchronize()
