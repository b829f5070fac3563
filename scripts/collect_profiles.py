"""Copy the round's evidence from gpurun_out/ (scratch) into profiles/ (tracked): run after
scripts/gpu_round_profiles.sh and scripts/gpu_c3_profiles.sh.  usage: collect_profiles.py r02"""
import csv, glob, json, os, shutil, sys
tag = sys.argv[1] if len(sys.argv) > 1 else "r02"
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src, dst = os.path.join(root, "gpurun_out"), os.path.join(root, "profiles")
def newest(pattern):
    files = glob.glob(os.path.join(src, pattern), recursive=True)
    return max(files, key=os.path.getmtime) if files else None
def copy(pattern, name):
    f = newest(pattern)
    if f:
        shutil.copy(f, os.path.join(dst, name)); print("copied", os.path.relpath(f, root), "->", name)
    else:
        print("missing", pattern)
copy("round/bench_line.json", f"{tag}_bench_line.json")
copy("round/bench_line_no_overlap.json", f"{tag}_bench_line_no_overlap.json")
copy("round/stats/**/*kernel_stats.csv", f"{tag}_bench_kernel_stats.csv")
copy("round/stats_seq/**/*kernel_stats.csv", f"{tag}_bench_kernel_stats_no_overlap.csv")
copy("round/traffic_summary.json", f"{tag}_hbm_traffic_per_kernel.json")
copy("c3prof/stats/**/*kernel_stats.csv", f"{tag}_c3_kernel_stats.csv")
copy("c3prof/c3_kernels.json", f"{tag}_c3_kernels_pmc.json")
t = os.path.join(dst, f"{tag}_hbm_traffic_per_kernel.json")
if os.path.exists(t):
    rows = json.load(open(t))
    w = [r for r in rows if r["kernel"].startswith("void warp_rigid_dma<true, true")]
    if w:
        r = w[0]
        json.dump({"kernel": r["kernel"], "source": f"profiles/{tag}_hbm_traffic_per_kernel.json (rocprofv3 --pmc FETCH_SIZE / "
                   "WRITE_SIZE, separate passes, bench.py --no-overlap)", "read_GB_fetch_size_x2": r["read_GB"],
                   "write_GB": r["write_GB"], "gfx950_fetch_correction": 2.0,
                   "hbm_bytes_per_launch": (r["read_GB"] + r["write_GB"]) * 1e9},
                  open(os.path.join(dst, "warp_traffic.json"), "w"), indent=1)
        print("updated warp_traffic.json")
