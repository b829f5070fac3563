"""Build libmcorr.so (hand-written HIP kernels + C ABI) for gfx950, in-tree.

``python -m torch_motion_correction_amd._build`` or ``build_library()``.
hipcc cross-compiles without a GPU; the .so is git-ignored but travels with the
source tree to the GPU box.
"""

from __future__ import annotations

import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

PKG_DIR = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(PKG_DIR, "csrc")
INCLUDE = os.path.join(os.path.dirname(PKG_DIR), "include")
LIB_PATH = os.path.join(PKG_DIR, "libmcorr.so")
# (source, object stem, extra flags), slowest first.  xcg_fft.hip is compiled as four objects, one kernel
# family each (XCG_PART): the 97 generic-length kernels took five minutes as one translation unit
SOURCES = [("xcg_fft.hip", "xcg_fft_p3", ["-DXCG_PART=3"]), ("xcg_fft.hip", "xcg_fft_p2", ["-DXCG_PART=2"]),
           ("xcg_fft.hip", "xcg_fft_p0", ["-DXCG_PART=0"]), ("xcg_fft.hip", "xcg_fft_p1", ["-DXCG_PART=1"]),
           ("xc_fft.hip", "xc_fft", []), ("full_fft.hip", "full_fft", []),
           # warp.hip: the SLP vectoriser turns the per-pixel coordinate chain into v_pk_* instructions fed by
           # ~1300 v_mov_b32 per kernel and 90 more VGPRs (warp_field 215 -> 160); packed fp32 issues at half
           # the scalar rate on gfx950, so nothing is gained for it
           ("warp.hip", "warp", ["-fno-slp-vectorize"]), ("plan_stats.hip", "plan_stats", []),
           ("field_post.hip", "field_post", []), ("local_motion.hip", "local_motion", []),
           ("polyphase.hip", "polyphase", [])]
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wno-unused-value"]


def _stale(target: str, deps: list[str]) -> bool:
    if not os.path.exists(target):
        return True
    mt = os.path.getmtime(target)
    return any(os.path.getmtime(d) > mt for d in deps)


def build_library(force: bool = False, verbose: bool = False) -> str:
    headers = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".h")]
    headers.append(os.path.join(INCLUDE, "mcorr.h"))
    objdir = os.path.join(PKG_DIR, "build")
    os.makedirs(objdir, exist_ok=True)

    def compile_one(item) -> str:
        src, stem, extra = item
        obj = os.path.join(objdir, stem + ".o")
        srcp = os.path.join(CSRC, src)
        if force or _stale(obj, [srcp] + headers):
            cmd = [HIPCC, *FLAGS, *extra, f"-I{INCLUDE}", f"-I{CSRC}", "-c", srcp, "-o", obj]
            if verbose:
                print(" ".join(cmd), flush=True)
            subprocess.run(cmd, check=True)
        return obj

    with ThreadPoolExecutor(max_workers=min(8, os.cpu_count() or 4)) as ex:
        objs = list(ex.map(compile_one, SOURCES))
    if force or _stale(LIB_PATH, objs):
        cmd = [HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", *objs, "-o", LIB_PATH]
        if verbose:
            print(" ".join(cmd), flush=True)
        subprocess.run(cmd, check=True)
    return LIB_PATH


if __name__ == "__main__":
    print(build_library(force="--force" in sys.argv, verbose=True))
