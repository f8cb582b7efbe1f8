#!/usr/bin/env python3
"""Builds the register-capped variants of the persistent LM kernel that tools/ab_align_regs.sh swaps in: gs_icp_slam_amd/libgsicp_hip_wN.so is the
product library with `gicp_align_kernel` compiled under `__attribute__((amdgpu_waves_per_eu(N, N)))` (N = 2 / 3 / 4: 256 / 168 / 128 registers per
lane, the compiler spills the rest).  Run `python -m gs_icp_slam_amd.build` first (the other objects are linked as they are); the variants are
measurement aids, git-ignored, and travel to the GPU box like the product library.  Result of round 3: DESIGN.md section 8 (all three lose)."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from gs_icp_slam_amd import build as B  # noqa: E402

CSRC = os.path.join(ROOT, "gs_icp_slam_amd", "csrc")
DECL = "__global__ __launch_bounds__(AL_T) void gicp_align_kernel"


def main():
    src = open(os.path.join(CSRC, "gicp.hip")).read()
    assert src.count(DECL) == 1, "gicp_align_kernel's declaration moved"
    extra = dict(B.SOURCES)["gicp.hip"]
    objs = [os.path.join(CSRC, s.replace(".hip", ".o")) for s, _ in B.SOURCES if s != "gicp.hip"]
    missing = [o for o in objs if not os.path.exists(o)]
    if missing:
        raise SystemExit(f"build the product first (python -m gs_icp_slam_amd.build): {missing}")
    hipcc = B._hipcc()
    tl = B._torch_lib_dir()
    link = [f"-L{tl}", f"-Wl,-rpath,{tl}", "-Wl,--disable-new-dtags"] if tl and os.path.exists(os.path.join(tl, "libamdhip64.so")) else []
    for n in (2, 3, 4):
        tmp = os.path.join(CSRC, f"_variant_gicp_w{n}.hip")      # next to gicp.hip: its relative includes must resolve
        obj = os.path.join(CSRC, f"_variant_gicp_w{n}.o")
        open(tmp, "w").write(src.replace(DECL, f"__global__ __launch_bounds__(AL_T) __attribute__((amdgpu_waves_per_eu({n}, {n}))) void gicp_align_kernel"))
        try:
            subprocess.check_call([hipcc] + B.COMMON + extra + ["-c", tmp, "-o", obj])
            out = os.path.join(ROOT, "gs_icp_slam_amd", f"libgsicp_hip_w{n}.so")
            subprocess.check_call([hipcc, "-shared", "-fPIC", f"--offload-arch={B.ARCH}", "-o", out] + objs + [obj] + link)
            print("built", out)
        finally:
            for f in (tmp, obj):
                if os.path.exists(f):
                    os.remove(f)


if __name__ == "__main__":
    main()
