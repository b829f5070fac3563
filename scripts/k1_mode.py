"""K1 alone with a fixed engine mode (argv[1]), few launches: for rocprofv3 PMC passes."""
import sys, torch
sys.argv = [sys.argv[0]] + sys.argv[1:]
exec(open("scripts/k1_only.py").read().split("modes = ")[0])
check(lib.mc_xc_row_engine(int(sys.argv[1]) if len(sys.argv) > 1 else 0), "engine")
for _ in range(4): k1()
torch.cuda.synchronize()
