"""ORACLE / TEST INFRASTRUCTURE — recipe that byte-compiles the reference's own host-side Python where it lies.

The untouched two-process system (gs_icp_slam.py -> mp_Tracker.py + mp_Mapper.py, scene/, gaussian_renderer/, utils/,
arguments/) is what the drop-in packages of this repo must serve.  /root/reference does not exist on the GPU box, and its
sources must not be copied into this repository, so — exactly like a C reference compiled into oracle/_ref/*.so — the reference's
Python files are COMPILED from /root/reference into sourceless byte-code under oracle/_ref/refpy/ (git-ignored, not
gpurun-ignored: it travels to the GPU box next to the built .so files, and never enters history).  Nothing is edited: the
byte-code is what `python -m compileall -b` produces for CPython 3.10, the interpreter on both machines.

    python oracle/make_refpy.py            (also run by `make -f Makefile.ref` and by __graft_entry__.build())

Used by tests/test_reference_slam_gpu.py and tools/run_reference_slam.py only (the checker, never the product).
"""
import os
import py_compile
import sys

REF = os.environ.get("GSICP_REFERENCE", "/root/reference")
HERE = os.path.dirname(os.path.abspath(__file__))
OUT = os.path.join(HERE, "_ref", "refpy")
# the host files of the live path (everything the two processes import), nothing from submodules/ or SIBR_viewers/
FILES = ["gs_icp_slam.py", "gs_icp_slam_unlimit.py", "mp_Tracker.py", "mp_Tracker_unlimit.py", "mp_Mapper.py"]
PACKAGES = ["arguments", "scene", "gaussian_renderer", "utils"]


def main():
    if not os.path.isdir(REF):
        print(f"make_refpy: {REF} not present (GPU box) — keeping whatever is under {OUT}")
        return 0
    srcs = [f for f in FILES if os.path.exists(os.path.join(REF, f))]
    for pkg in PACKAGES:
        for dirpath, _dirs, files in os.walk(os.path.join(REF, pkg)):
            for f in files:
                if f.endswith(".py"):
                    srcs.append(os.path.relpath(os.path.join(dirpath, f), REF))
    n = 0
    for rel in srcs:
        src = os.path.join(REF, rel)
        dst = os.path.join(OUT, rel + "c")          # legacy (sourceless) layout: foo.pyc where foo.py would be
        os.makedirs(os.path.dirname(dst), exist_ok=True)
        if not os.path.exists(dst) or os.path.getmtime(dst) < os.path.getmtime(src):
            py_compile.compile(src, cfile=dst, dfile=os.path.join("<reference>", rel), doraise=True)
            n += 1
    import shutil
    shutil.copytree(os.path.join(REF, "configs"), os.path.join(OUT, "configs"), dirs_exist_ok=True)   # three-line camera configs (data)
    print(f"make_refpy: {len(srcs)} reference modules under {OUT} ({n} compiled now)")
    return 0


if __name__ == "__main__":
    sys.exit(main())
