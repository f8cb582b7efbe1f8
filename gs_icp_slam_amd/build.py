"""Builds libgsicp_hip.so (gfx950) in-tree with hipcc.  Cross-compiles without a GPU.

    python -m gs_icp_slam_amd.build [--force]
"""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
OUT = os.path.join(HERE, "libgsicp_hip.so")
ARCH = "gfx950"

# (source, extra flags).  raster_preprocess is built without FMA contraction so that the floats feeding integer
# outputs (sort keys, tile rectangles) are reproducible against the CPU oracle; see the file header.
SOURCES = [
    ("raster_preprocess.hip", ["-ffp-contract=off"]),
    ("raster.hip", []),
    ("mapper_ops.hip", []),
    ("frontend.hip", ["-ffp-contract=off"]),
    ("gicp.hip", ["-ffp-contract=off"]),
]
COMMON = ["-O3", "-std=c++17", "-fPIC", f"--offload-arch={ARCH}", "-munsafe-fp-atomics", "-Wall", "-Wno-unused-function",
          "-Wno-unused-result"]


def _hipcc():
    for c in (os.environ.get("HIPCC"), "/opt/rocm/bin/hipcc", "hipcc"):
        if c and (os.path.isabs(c) and os.path.exists(c) or not os.path.isabs(c)):
            return c
    raise RuntimeError("hipcc not found")


def _torch_lib_dir():
    try:
        import importlib.util
        spec = importlib.util.find_spec("torch")
        if spec and spec.origin:
            return os.path.join(os.path.dirname(spec.origin), "lib")
    except Exception:
        pass
    return None


def _newer(target, deps):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


def build(force=False, verbose=True):
    hipcc = _hipcc()
    headers = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".hpp")]
    headers.append(os.path.join(HERE, "..", "include", "gsicp_hip.h"))
    objs = []
    for src, extra in SOURCES:
        sp = os.path.join(CSRC, src)
        if not os.path.exists(sp):
            continue
        obj = os.path.join(CSRC, src.replace(".hip", ".o"))
        if force or _newer(obj, [sp] + headers):
            cmd = [hipcc] + COMMON + extra + ["-c", sp, "-o", obj]
            if verbose:
                print(" ".join(cmd), flush=True)
            subprocess.check_call(cmd)
        objs.append(obj)
    if force or _newer(OUT, objs):
        # Link against the HIP runtime PyTorch already carries (torch/lib/libamdhip64.so, no versioned SONAME) so that
        # the process holds ONE runtime: device pointers and streams handed over by torch must belong to the runtime
        # our kernels are launched through.  Falls back to /opt/rocm's runtime when torch is not installed.
        link = []
        tl = _torch_lib_dir()
        if tl and os.path.exists(os.path.join(tl, "libamdhip64.so")):
            link = [f"-L{tl}", f"-Wl,-rpath,{tl}", "-Wl,--disable-new-dtags"]
        cmd = [hipcc, "-shared", "-fPIC", f"--offload-arch={ARCH}", "-o", OUT] + objs + link
        if verbose:
            print(" ".join(cmd), flush=True)
        subprocess.check_call(cmd)
    build_experiments(force=force, verbose=verbose)
    build_pybind_module(force=force, verbose=verbose)
    build_torch_ext_module(force=force, verbose=verbose)
    return OUT


def build_pybind_module(force=False, verbose=True):
    """integration/pygicp_pybind.cpp -> integration/pygicp.<abi>.so: a COMPILED `pygicp` module (PyInit_pygicp) over the C ABI
    (INTEGRATION.md 3.2); it opens libgsicp_hip.so at import (after torch, like the ctypes mirror) instead of linking it.
    Plain g++ + pybind11: no HIP code in the binding."""
    import sysconfig
    try:
        import pybind11
    except ImportError:
        return None
    root = os.path.dirname(HERE)
    src = os.path.join(root, "integration", "pygicp_pybind.cpp")
    if not os.path.exists(src):
        return None
    out = os.path.join(root, "integration", "pygicp" + (sysconfig.get_config_var("EXT_SUFFIX") or ".so"))
    if force or _newer(out, [src, os.path.join(root, "include", "gsicp_hip.h"), OUT]):
        cmd = ["g++", "-O2", "-std=c++17", "-shared", "-fPIC", "-fvisibility=hidden", f"-I{pybind11.get_include()}", f"-I{sysconfig.get_paths()['include']}",
               src, "-o", out, "-ldl"]
        if verbose:
            print(" ".join(cmd), flush=True)
        subprocess.check_call(cmd)
    return out


def build_torch_ext_module(force=False, verbose=True):
    """integration/torch_ext_pybind.cpp -> integration/torch_ext/{diff_gaussian_rasterization,simple_knn}/_C.<abi>.so: the COMPILED torch extension
    `_C` (rasterize_gaussians, rasterize_gaussians_backward, mark_visible, distCUDA2) over the C ABI.  Plain g++ against
    libtorch's headers (torch tensors and the current stream only; no HIP code in the binding); libgsicp_hip.so is opened at first use, not linked."""
    import shutil
    import sysconfig
    try:
        import pybind11  # noqa: F401
        import torch
        from torch.utils import cpp_extension
    except ImportError:
        return None
    root = os.path.dirname(HERE)
    src = os.path.join(root, "integration", "torch_ext_pybind.cpp")
    if not os.path.exists(src):
        return None
    suffix = sysconfig.get_config_var("EXT_SUFFIX") or ".so"
    outs = [os.path.join(root, "integration", "torch_ext", pkg, "_C" + suffix) for pkg in ("diff_gaussian_rasterization", "simple_knn")]
    if force or any(_newer(o, [src, os.path.join(root, "include", "gsicp_hip.h")]) for o in outs):
        tl = _torch_lib_dir()
        rocm_inc = os.path.join(os.environ.get("ROCM_PATH", "/opt/rocm"), "include")
        cmd = (["g++", "-O2", "-std=c++17", "-shared", "-fPIC", "-fvisibility=hidden", "-Wno-deprecated-declarations", "-D__HIP_PLATFORM_AMD__=1", "-DUSE_ROCM=1",
                "-DTORCH_API_INCLUDE_EXTENSION_H", "-DTORCH_EXTENSION_NAME=_C", f"-D_GLIBCXX_USE_CXX11_ABI={int(torch._C._GLIBCXX_USE_CXX11_ABI)}"] +
               [f"-I{p}" for p in cpp_extension.include_paths()] + [f"-I{rocm_inc}", f"-I{sysconfig.get_paths()['include']}", src, "-o", outs[0],
                f"-L{tl}", f"-Wl,-rpath,{tl}", "-lc10", "-lc10_hip", "-ltorch", "-ltorch_cpu", "-ltorch_python", "-ldl"])
        if verbose:
            print(" ".join(cmd), flush=True)
        subprocess.check_call(cmd)
        shutil.copyfile(outs[0], outs[1])
    return outs


def build_experiments(force=False, verbose=True):
    """Side library with recorded experiments (not loaded by the product): csrc/experiments/*.hip -> libgsicp_experiments.so."""
    hipcc = _hipcc()
    exp_dir = os.path.join(CSRC, "experiments")
    out = os.path.join(HERE, "libgsicp_experiments.so")
    srcs = sorted(os.path.join(exp_dir, f) for f in os.listdir(exp_dir) if f.endswith(".hip")) if os.path.isdir(exp_dir) else []
    if not srcs:
        return None
    if force or _newer(out, srcs):
        link = []
        tl = _torch_lib_dir()
        if tl and os.path.exists(os.path.join(tl, "libamdhip64.so")):
            link = [f"-L{tl}", f"-Wl,-rpath,{tl}", "-Wl,--disable-new-dtags"]
        cmd = [hipcc] + COMMON + ["-shared", "-o", out] + srcs + link
        if verbose:
            print(" ".join(cmd), flush=True)
        subprocess.check_call(cmd)
    return out


if __name__ == "__main__":
    build(force="--force" in sys.argv)
