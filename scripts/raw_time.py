"""fused raw path alone, a few repetitions (for rocprofv3 passes)"""
import sys, torch
sys.path.insert(0, ".")
sys.path.insert(0, "scripts")
import torch_motion_correction_amd as mc
from raw_check import raw_movie  # noqa
import os
dev = torch.device("cuda:0")
kind = torch.int16 if os.environ.get("RAW_I16") == "1" else torch.uint8
raw, gain, _, _ = raw_movie(40, 4096, 4096, kind, 3, 7)
for _ in range(int(os.environ.get("REPS", "4"))):
    f, s, fr = mc.motion_correct_raw(raw, gain, 1.0, return_frames=True)
torch.cuda.synchronize()
print("done", float(s.abs().max()))
