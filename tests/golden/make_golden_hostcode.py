#!/usr/bin/env python
"""Golden vectors from the reference's OWN host code for the two "next" rows whose source files cannot be imported whole
(mp_Tracker.py needs cv2 / open3d / rerun, scene/gaussian_model.py needs simple_knn and a GPU): the relevant METHODS are lifted out
of the reference files with `ast` — unmodified — and executed here against a stand-in `self`.

  * Tracker.set_downsample_filter, Tracker.downsample_and_make_pointcloud2                    [REF mp_Tracker.py:394-431]
  * GaussianModel.cat_tensors_to_optimizer, densification_postfix, _prune_optimizer, prune_points   [REF scene/gaussian_model.py:409-492]
    (their hard-coded device="cuda" keyword is dropped by a shim around torch.zeros — there is no GPU in this container)

They pin oracle/frontend_oracle.py and the RefModel restatement in tests/test_store_gpu.py, against which the HIP paths are tested.

    python tests/golden/make_golden_hostcode.py      -> tests/golden/ref_hostcode.npz
"""
import ast
import os
import sys
from types import SimpleNamespace

import numpy as np
import torch
import torch.nn as nn

HERE = os.path.dirname(os.path.abspath(__file__))
REF = "/root/reference"


def lift(path, cls_name, names):
    tree = ast.parse(open(path).read())
    cls = next(n for n in tree.body if isinstance(n, ast.ClassDef) and n.name == cls_name)
    fns = [n for n in cls.body if isinstance(n, ast.FunctionDef) and n.name in names]
    assert len(fns) == len(names), [f.name for f in fns]
    ns = {"torch": torch, "np": np, "nn": nn}
    exec(compile(ast.Module(body=fns, type_ignores=[]), path, "exec"), ns)
    return ns


out = {}
rng = np.random.default_rng(5)

# ------------------------------------------------------------------------------------------------ tracker front-end
tr = lift(os.path.join(REF, "mp_Tracker.py"), "Tracker", ["set_downsample_filter", "downsample_and_make_pointcloud2", "quaternion_multiply",
                                                         "eliminate_overlapped2"])
# quaternion composition and overlap statistics [REF mp_Tracker.py:385-392, 433-439, 235]
from scipy.spatial.transform import Rotation  # noqa: E402  (the reference's own dependency for R -> quaternion)
Rm = Rotation.from_euler("xyz", [20.0, -35.0, 110.0], degrees=True).as_matrix()
q_cam = Rotation.from_matrix(Rm).as_quat()
Q2 = rng.normal(size=(50, 4)); Q2 /= np.linalg.norm(Q2, axis=1, keepdims=True)
out["qm_R"], out["qm_q1"], out["qm_Q2"] = Rm, q_cam, Q2
out["qm_out"] = tr["quaternion_multiply"](None, q_cam, Q2)
dist = np.abs(rng.normal(0, 4e-4, 500)).astype(np.float32)
out["ov_d"] = dist
out["ov_new"] = tr["eliminate_overlapped2"](None, dist, 5e-5)[0]
out["ov_len_corres"] = np.array(len(np.where(dist < 5e-4)[0]))
for tag, (H, W, fx, fy, cx, cy, stride, dscale, trunc) in {"a": (48, 64, 50.0, 52.0, 31.5, 23.5, 5, 1000.0, 3.0),
                                                            "b": (57, 83, 61.3, 60.2, 40.1, 28.7, 7, 6553.5, 2.5)}.items():
    me = SimpleNamespace(H=H, W=W, fx=fx, fy=fy, cx=cx, cy=cy, depth_scale=dscale, depth_trunc=trunc)
    me.downsample_idxs, me.x_pre, me.y_pre = tr["set_downsample_filter"](me, stride)
    depth = rng.uniform(0.3, 4.0, (H, W))
    depth[rng.uniform(size=(H, W)) < 0.2] = 0.0
    depth_raw = np.clip(np.round(depth * dscale), 0, 65535).astype(np.uint16)
    rgb = rng.integers(0, 256, (H, W, 3), dtype=np.uint8)
    pts, col, z, filt = tr["downsample_and_make_pointcloud2"](me, depth_raw, rgb)
    out[f"fe_{tag}_cfg"] = np.array([H, W, fx, fy, cx, cy, stride, dscale, trunc], np.float64)
    out[f"fe_{tag}_depth"], out[f"fe_{tag}_rgb"] = depth_raw, rgb
    out[f"fe_{tag}_pick"] = me.downsample_idxs[0].numpy()
    out[f"fe_{tag}_xpre"], out[f"fe_{tag}_ypre"] = me.x_pre.numpy(), me.y_pre.numpy()
    out[f"fe_{tag}_points"], out[f"fe_{tag}_colors"], out[f"fe_{tag}_z"], out[f"fe_{tag}_filter"] = pts, col, z, filt

# ------------------------------------------------------------------------------------------------ map growth / pruning
_zeros, _zeros_like = torch.zeros, torch.zeros_like
torch.zeros = lambda *a, **k: _zeros(*a, **{kk: v for kk, v in k.items() if kk != "device"})
gm = lift(os.path.join(REF, "scene", "gaussian_model.py"), "GaussianModel",
          ["cat_tensors_to_optimizer", "densification_postfix", "_prune_optimizer", "prune_points"])
NAMES = ("xyz", "f_dc", "f_rest", "opacity", "scaling", "rotation")
ATTR = {"xyz": "_xyz", "f_dc": "_features_dc", "f_rest": "_features_rest", "opacity": "_opacity", "scaling": "_scaling", "rotation": "_rotation"}
SHAPES = {"xyz": (3,), "f_dc": (1, 3), "f_rest": (3, 3), "opacity": (1,), "scaling": (3,), "rotation": (4,)}


def rows(k):
    return {n: torch.from_numpy(rng.normal(size=(k,) + SHAPES[n]).astype(np.float32)) for n in NAMES}


class Me(SimpleNamespace):
    @property
    def get_xyz(self):
        return self._xyz


first = rows(40)
me = Me(**{ATTR[n]: nn.Parameter(first[n].clone().requires_grad_(True)) for n in NAMES})
me.optimizer = torch.optim.Adam([{"params": [getattr(me, ATTR[n])], "lr": 1e-3, "name": n} for n in NAMES], lr=0.0, eps=1e-15)
me.trackable_mask = torch.from_numpy(rng.uniform(size=40) < 0.6)
me.keyframe_idx = torch.arange(40)
for m in ("cat_tensors_to_optimizer", "densification_postfix", "_prune_optimizer", "prune_points"):
    setattr(me, m, (lambda f: (lambda *a, **k: f(me, *a, **k)))(gm[m]))
# give the moments recognisable content (as if steps had been taken)
for n in NAMES:
    p = getattr(me, ATTR[n])
    me.optimizer.state[p] = {"step": torch.tensor(3.0), "exp_avg": torch.from_numpy(rng.normal(size=p.shape).astype(np.float32)),
                             "exp_avg_sq": torch.from_numpy(rng.uniform(size=p.shape).astype(np.float32))}
for n in NAMES:
    out[f"st_first_{n}"] = first[n].numpy()
    out[f"st_first_m_{n}"] = me.optimizer.state[getattr(me, ATTR[n])]["exp_avg"].numpy().copy()
    out[f"st_first_v_{n}"] = me.optimizer.state[getattr(me, ATTR[n])]["exp_avg_sq"].numpy().copy()
out["st_first_trackable"] = me.trackable_mask.numpy().copy()
ops = []
for step, (kind, arg) in enumerate([("cat", 15), ("prune", 0.3), ("cat", 7), ("prune", 0.5)]):
    if kind == "cat":
        new = rows(arg)
        tm = torch.from_numpy(rng.uniform(size=arg) < 0.5)
        me.keyframe_idx = torch.cat([me.keyframe_idx, torch.full((arg,), 100 + step)])
        me.densification_postfix(new["xyz"], new["f_dc"], new["f_rest"], new["opacity"], new["scaling"], new["rotation"], tm)
        for n in NAMES:
            out[f"st_op{step}_new_{n}"] = new[n].numpy()
        out[f"st_op{step}_new_trackable"] = tm.numpy()
    else:
        mask = torch.from_numpy(rng.uniform(size=me._xyz.shape[0]) < arg)
        me.xyz_gradient_accum = torch.arange(me._xyz.shape[0], dtype=torch.float32)[:, None].clone()
        me.denom = me.xyz_gradient_accum + 0.5
        me.max_radii2D = me.xyz_gradient_accum[:, 0] * 2
        me.prune_points(mask)
        out[f"st_op{step}_mask"] = mask.numpy()
        out[f"st_op{step}_accum"] = me.xyz_gradient_accum.numpy().copy()
    ops.append(kind)
    for n in NAMES:
        p = getattr(me, ATTR[n])
        out[f"st_op{step}_{n}"] = p.detach().numpy().copy()
        out[f"st_op{step}_m_{n}"] = me.optimizer.state[p]["exp_avg"].numpy().copy()
        out[f"st_op{step}_v_{n}"] = me.optimizer.state[p]["exp_avg_sq"].numpy().copy()
    out[f"st_op{step}_trackable"] = me.trackable_mask.numpy().copy()
out["st_ops"] = np.array(ops)

# ------------------------------------------------------------------------------------------------ Gaussian initialisation from GICP outputs
# GaussianModel.create_from_pcd2_tensor [REF scene/gaussian_model.py:134-164] with the reference's own RGB2SH / inverse_sigmoid;
# `.cuda()` is a no-op here (no GPU in this container)
sys.path.insert(0, REF)
import utils.general_utils as gu   # noqa: E402
import utils.sh_utils as su        # noqa: E402
_cuda, _ones = torch.Tensor.cuda, torch.ones
torch.Tensor.cuda = lambda self, *a, **k: self
torch.ones = lambda *a, **k: _ones(*a, **{kk: v for kk, v in k.items() if kk != "device"})
tree = ast.parse(open(os.path.join(REF, "scene", "gaussian_model.py")).read())
cls = next(n for n in tree.body if isinstance(n, ast.ClassDef) and n.name == "GaussianModel")
fn = [n for n in cls.body if isinstance(n, ast.FunctionDef) and n.name == "create_from_pcd2_tensor"]
ns = {"torch": torch, "np": np, "nn": nn, "RGB2SH": su.RGB2SH, "inverse_sigmoid": gu.inverse_sigmoid}
exec(compile(ast.Module(body=fn, type_ignores=[]), "gaussian_model.py", "exec"), ns)
for deg in (0, 3):
    n0 = 60
    init_in = dict(points=rng.normal(size=(n0, 3)).astype(np.float32), colors=rng.uniform(0, 1, (n0, 3)).astype(np.float32),
                   rots=rng.normal(size=(n0, 4)).astype(np.float32), scales=rng.uniform(0.005, 0.08, (n0, 3)).astype(np.float32),
                   z=rng.uniform(0.3, 5.0, n0).astype(np.float32), trk=np.sort(rng.choice(n0, 35, replace=False)))
    gm_me = Me(max_sh_degree=deg)
    ns["create_from_pcd2_tensor"](gm_me, torch.from_numpy(init_in["points"]).clone(), torch.from_numpy(init_in["colors"]), torch.from_numpy(init_in["rots"]).clone(),
                                  torch.from_numpy(init_in["scales"]), torch.from_numpy(init_in["z"]), torch.from_numpy(init_in["trk"]))
    for k, v in init_in.items():
        out[f"init{deg}_in_{k}"] = v
    for k, attr in (("xyz", "_xyz"), ("f_dc", "_features_dc"), ("f_rest", "_features_rest"), ("opacity", "_opacity"), ("scaling", "_scaling"),
                    ("rotation", "_rotation")):
        out[f"init{deg}_{k}"] = getattr(gm_me, attr).detach().numpy().copy()
    out[f"init{deg}_trackable"] = gm_me.trackable_mask.numpy().copy()
torch.Tensor.cuda, torch.ones = _cuda, _ones
torch.zeros = _zeros

np.savez_compressed(os.path.join(HERE, "ref_hostcode.npz"), **out)
print("wrote ref_hostcode.npz:", len(out), "arrays; front-end points", out["fe_a_points"].shape, out["fe_b_points"].shape,
      "store final n", out["st_op3_xyz"].shape[0])
