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


# ---------------------------------------------------------------------------------------------------------------- --fused
# INTEGRATION.md §6-8's "few-line edits" applied as an AST transform AT BUILD TIME into oracle/_ref/refpy_fused (git-ignored): the
# reference's files are parsed where they lie, edited in memory and byte-compiled; no reference source is written anywhere.  What the
# edits call lives in the product (gs_icp_slam_amd/refglue.py).
import ast


def _append_patch(tree, func, cls_name):
    """`from gs_icp_slam_amd.refglue import <func>` + `<func>(<cls_name>)` at the end of the module."""
    tree.body.append(ast.ImportFrom(module="gs_icp_slam_amd.refglue", names=[ast.alias(name=func)], level=0))
    tree.body.append(ast.Expr(ast.Call(func=ast.Name(id=func, ctx=ast.Load()), args=[ast.Name(id=cls_name, ctx=ast.Load())], keywords=[])))
    return 2


def _is_self_training_assign(node, value):
    return (isinstance(node, ast.Assign) and len(node.targets) == 1 and isinstance(node.targets[0], ast.Attribute) and
            node.targets[0].attr == "training" and isinstance(node.targets[0].value, ast.Name) and node.targets[0].value.id == "self" and
            isinstance(node.value, ast.Constant) and node.value.value is value)


def _fuse_mapper(tree):
    """Mapper.mapping: the statements between `self.training=True` and `self.training = False` (render_3 ... optimizer.zero_grad, the viewer
    log) become ONE call [REF mp_Mapper.py:219-262]."""
    edits = 0
    for node in ast.walk(tree):
        body = getattr(node, "body", None)
        if not isinstance(body, list):
            continue
        i0 = next((i for i, st in enumerate(body) if _is_self_training_assign(st, True)), None)
        i1 = next((i for i, st in enumerate(body) if _is_self_training_assign(st, False)), None)
        if i0 is None or i1 is None or i1 <= i0:
            continue
        call = ast.Expr(ast.Call(func=ast.Name(id="fused_mapping_iteration", ctx=ast.Load()),
                                 args=[ast.Name(id=n, ctx=ast.Load()) for n in ("self", "viewpoint_cam", "gt_image", "gt_depth_image")], keywords=[]))
        edits += i1 - i0 - 1
        body[i0 + 1:i1] = [call]
    if not edits:
        raise RuntimeError("make_refpy --fused: Mapper.mapping's training block not found")
    tree.body.insert(0, ast.ImportFrom(module="gs_icp_slam_amd.refglue", names=[ast.alias(name="fused_mapping_iteration")], level=0))
    return edits


def _fuse_tracker(tree):
    """Tracker.tracking: the new target arrives as device tensors (`get_values_np()` -> `get_values_tensor()` [REF mp_Tracker.py:286]); the front-end
    method is replaced after the class."""
    edits = 0
    for node in ast.walk(tree):
        if isinstance(node, ast.Attribute) and node.attr == "get_values_np" and isinstance(node.value, ast.Attribute) and node.value.attr == "shared_target_gaussians":
            node.attr = "get_values_tensor"
            edits += 1
    if not edits:
        raise RuntimeError("make_refpy --fused: the tracker's get_values_np() call was not found")
    return edits + _append_patch(tree, "patch_tracker", "Tracker")


def _lift_mapping_block(tree, src, out_dir):
    """The reference's OWN training statements [REF mp_Mapper.py:219-262] as a callable: the slice between `self.training=True` and
    `self.training = False` of Mapper.mapping, wrapped as `reference_training_block(self, viewpoint_cam, gt_image, gt_depth_image, new_keyframe=False)`
    under the module's own imports, byte-compiled into the PLAIN tree (`_lifted_mapping_block.pyc`).  Test infrastructure: lets a GPU test run the
    reference's iteration and the fused one side by side on the box where only byte-code exists (tests/refglue_iteration_probe.py)."""
    block = None
    for node in ast.walk(tree):
        body = getattr(node, "body", None)
        if not isinstance(body, list):
            continue
        i0 = next((i for i, st in enumerate(body) if _is_self_training_assign(st, True)), None)
        i1 = next((i for i, st in enumerate(body) if _is_self_training_assign(st, False)), None)
        if i0 is not None and i1 is not None and i1 > i0:
            block = body[i0 + 1:i1]
    if block is None:
        raise RuntimeError("make_refpy: Mapper.mapping's training block not found")
    imports = [n for n in tree.body if isinstance(n, (ast.Import, ast.ImportFrom))]
    ret = ast.Return(ast.Tuple(elts=[ast.Name(id=n, ctx=ast.Load()) for n in ("loss", "image", "depth_image")], ctx=ast.Load()))
    fn = ast.FunctionDef(name="reference_training_block",
                         args=ast.arguments(posonlyargs=[], args=[ast.arg(arg=a) for a in ("self", "viewpoint_cam", "gt_image", "gt_depth_image", "new_keyframe")],
                                            kwonlyargs=[], kw_defaults=[], defaults=[ast.Constant(False)]),
                         body=list(block) + [ret], decorator_list=[])
    mod = ast.Module(body=imports + [fn], type_ignores=[])
    _compile_tree(mod, src, os.path.join(out_dir, "_lifted_mapping_block.pyc"), os.path.join("<reference, lifted>", "mp_Mapper.py"))
    return len(block)


FUSED_EDITS = {
    "mp_Mapper.py": _fuse_mapper,
    "mp_Tracker.py": _fuse_tracker,
    "mp_Tracker_unlimit.py": _fuse_tracker,
    os.path.join("scene", "gaussian_model.py"): lambda t: _append_patch(t, "patch_gaussian_model", "GaussianModel"),
    os.path.join("scene", "shared_objs.py"): lambda t: _append_patch(t, "patch_shared_targets", "SharedTargetPoints"),
}


def _compile_tree(tree, src, dst, dfile):
    import importlib._bootstrap_external as be
    ast.fix_missing_locations(tree)
    code = compile(tree, dfile, "exec", dont_inherit=True, optimize=-1)
    st = os.stat(src)
    data = be._code_to_timestamp_pyc(code, int(st.st_mtime), st.st_size & 0xFFFFFFFF)
    with open(dst, "wb") as fh:
        fh.write(data)


def main_fused():
    out = os.path.join(HERE, "_ref", "refpy_fused")
    if not os.path.isdir(REF):
        print(f"make_refpy --fused: {REF} not present (GPU box) — keeping whatever is under {out}")
        return 0
    srcs = [f for f in FILES if os.path.exists(os.path.join(REF, f))]
    for pkg in PACKAGES:
        for dirpath, _dirs, files in os.walk(os.path.join(REF, pkg)):
            srcs += [os.path.relpath(os.path.join(dirpath, f), REF) for f in files if f.endswith(".py")]
    report = {}
    for rel in srcs:
        src, dst = os.path.join(REF, rel), os.path.join(out, rel + "c")
        os.makedirs(os.path.dirname(dst), exist_ok=True)
        dfile = os.path.join("<reference, fused>" if rel in FUSED_EDITS else "<reference>", rel)
        if rel in FUSED_EDITS:
            with open(src, "rb") as fh:
                tree = ast.parse(fh.read(), filename=dfile)
            report[rel] = FUSED_EDITS[rel](tree)
            _compile_tree(tree, src, dst, dfile)
        else:
            py_compile.compile(src, cfile=dst, dfile=dfile, doraise=True)
    with open(os.path.join(REF, "mp_Mapper.py"), "rb") as fh:       # the untouched training statements as a callable, into the PLAIN tree
        report["_lifted_mapping_block (plain tree)"] = _lift_mapping_block(ast.parse(fh.read()), os.path.join(REF, "mp_Mapper.py"), OUT)
    missing = [r for r in FUSED_EDITS if r not in report]
    if missing:
        raise RuntimeError(f"make_refpy --fused: files to edit not found: {missing}")
    import shutil
    shutil.copytree(os.path.join(REF, "configs"), os.path.join(out, "configs"), dirs_exist_ok=True)
    print(f"make_refpy --fused: {len(srcs)} reference modules under {out}; statements replaced / added per edited file: {report}")
    return 0


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
    rc = main()
    sys.exit(rc or main_fused())     # both trees, always: the plain one is what the parity / system tests run, the fused one what --fused runs
