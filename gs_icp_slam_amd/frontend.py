"""Tracker front-end on the device: the per-frame depth -> point-cloud step of the reference's Tracker
[REF mp_Tracker.py:393-431] producing DEVICE tensors that go straight into the tracker's device-pointer overloads.

    fe = DepthFrontEnd(H, W, fx, fy, cx, cy, downsample_scale, depth_scale, depth_trunc)     # once  (set_downsample_filter)
    pc = fe.make_pointcloud(depth_dev, rgb_dev)                                               # per frame, one launch
    reg.set_input_source(pc.points); reg.set_source_trackable(pc.trackable_idx)               # no host round trip

The pick table (row/column subsampling and the pre-divided pixel coordinates) is built exactly as the reference builds it, with
the same torch expressions, so the back-projected points agree bit for bit with the reference's CPU path.
"""
import ctypes
from typing import NamedTuple

import torch

from . import _lib


class PointCloud(NamedTuple):
    points: torch.Tensor         # (n,3) float32, camera frame, picks with non-zero depth in pick order
    colors: torch.Tensor         # (n,3) float32 in [0,1]  (None when no rgb image was given)
    z_values: torch.Tensor       # (n,)  float32
    trackable_idx: torch.Tensor  # (m,)  int32 indices into the arrays above with z <= depth_trunc, ascending


class DepthFrontEnd:
    def __init__(self, H, W, fx, fy, cx, cy, downsample_scale, depth_scale, depth_trunc, device="cuda"):
        self.H, self.W = int(H), int(W)
        self.depth_scale, self.depth_trunc = float(depth_scale), float(depth_trunc)
        self.device = torch.device(device)
        s = downsample_scale
        # rows 0, s-1, 2s-1, ...; every s-th column [REF mp_Tracker.py:393-403]
        rows = s * torch.arange(0, int(self.H / s) + 1) - 1
        rows[0] = 0
        cols = torch.arange(0, self.W, s)
        pick = (rows[:, None] * self.W + cols[None, :]).reshape(-1)
        u = (pick % self.W)
        v = torch.div(pick, self.W, rounding_mode="floor")
        # pre-divided pixel coordinates, float32 like the reference's `(u - cx) / fx` on an int64 tensor [REF mp_Tracker.py:404-411]
        self.pick_idx_cpu, self.x_pre_cpu, self.y_pre_cpu = pick, (u - cx) / fx, (v - cy) / fy
        self.pick_idx = pick.to(self.device)
        self.x_pre = self.x_pre_cpu.to(self.device)
        self.y_pre = self.y_pre_cpu.to(self.device)
        self.n_pick = int(pick.numel())

    def make_pointcloud(self, depth, rgb=None):
        """depth: (H,W) uint16 / int16-as-uint16 or float32 DEVICE tensor (raw sensor units); rgb: (H,W,3) uint8 DEVICE tensor or None.
        Synchronises once (to learn the two counts)."""
        lib = _lib.load()
        if not depth.is_cuda:
            raise RuntimeError("DepthFrontEnd (gfx950): the depth image must live on the HIP device; there is no CPU path")
        dev = depth.device
        if depth.numel() != self.H * self.W:
            raise RuntimeError("DepthFrontEnd: depth image size does not match H x W")
        if depth.dtype in (torch.uint16, torch.int16):
            dtype_code = 0
        elif depth.dtype == torch.float32:
            dtype_code = 1
        else:
            raise RuntimeError("DepthFrontEnd: depth must be uint16 or float32")
        depth = depth.contiguous()
        if rgb is not None:
            if rgb.dtype != torch.uint8 or rgb.numel() != 3 * self.H * self.W or not rgb.is_cuda:
                raise RuntimeError("DepthFrontEnd: rgb must be a (H,W,3) uint8 device tensor")
            rgb = rgb.contiguous()
        n = self.n_pick
        points = torch.empty((n, 3), dtype=torch.float32, device=dev)
        colors = torch.empty((n, 3), dtype=torch.float32, device=dev) if rgb is not None else None
        z = torch.empty((n,), dtype=torch.float32, device=dev)
        trk = torch.empty((n,), dtype=torch.int32, device=dev)
        counts = torch.empty((2,), dtype=torch.int32, device=dev)
        p = lambda t: None if t is None else ctypes.c_void_p(t.data_ptr())
        with torch.cuda.device(dev):
            stream = ctypes.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
            _lib.check(lib.gsicp_frontend_make_pointcloud(n, p(self.pick_idx), p(self.x_pre), p(self.y_pre), p(depth), dtype_code, p(rgb),
                                                          self.depth_scale, self.depth_trunc, p(points), p(colors), p(z), p(trk), p(counts),
                                                          stream), "gsicp_frontend_make_pointcloud")
        n_pts, n_trk = counts.tolist()
        return PointCloud(points[:n_pts], None if colors is None else colors[:n_pts], z[:n_pts], trk[:n_trk])

    @staticmethod
    def to_world(points, pose_c2w):
        """points (n,3) camera frame -> world frame with the 4x4 camera-to-world pose (the `R points - R T` of [REF mp_Tracker.py:224-229]
        written with the inverse pose's rotation/translation is the same rigid map)."""
        pose = torch.as_tensor(pose_c2w, dtype=torch.float32, device=points.device)
        return points @ pose[:3, :3].T + pose[:3, 3]


# ---- the remaining per-frame front-end arithmetic of the tracker, as device-agnostic torch (no host round trip when the inputs are
# ---- device tensors; SURVEY.md §8f rank 3: "world transform, quaternion composition, overlap statistics")

def quaternion_multiply(q1, Q2):
    """q1 * Q2 for one (x,y,z,w) quaternion q1 and an (n,4) batch Q2 — the Hamilton product exactly as
    Tracker.quaternion_multiply writes it [REF mp_Tracker.py:385-392] (used to rotate the GICP covariances' quaternions into the
    world frame at keyframes [REF mp_Tracker.py:259-261, 304-306])."""
    Q2 = torch.as_tensor(Q2)
    q1 = torch.as_tensor(q1, dtype=Q2.dtype, device=Q2.device)
    x0, y0, z0, w0 = q1[0], q1[1], q1[2], q1[3]
    return torch.stack([w0 * Q2[:, 0] + x0 * Q2[:, 3] + y0 * Q2[:, 2] - z0 * Q2[:, 1],
                        w0 * Q2[:, 1] + y0 * Q2[:, 3] + z0 * Q2[:, 0] - x0 * Q2[:, 2],
                        w0 * Q2[:, 2] + z0 * Q2[:, 3] + x0 * Q2[:, 1] - y0 * Q2[:, 0],
                        w0 * Q2[:, 3] - x0 * Q2[:, 0] - y0 * Q2[:, 1] - z0 * Q2[:, 2]], dim=1)


def rotation_to_quaternion_xyzw(R):
    """(x,y,z,w) quaternion of a 3x3 rotation matrix (what scipy's Rotation.from_matrix(R).as_quat() yields, up to the overall sign,
    which is immaterial: q and -q are the same rotation and the rasteriser normalises)."""
    R = torch.as_tensor(R, dtype=torch.float64)
    t = R[0, 0] + R[1, 1] + R[2, 2]
    cand = torch.stack([1.0 + 2.0 * R[0, 0] - t, 1.0 + 2.0 * R[1, 1] - t, 1.0 + 2.0 * R[2, 2] - t, 1.0 + t])   # 4x^2, 4y^2, 4z^2, 4w^2
    k = int(torch.argmax(cand))
    q = torch.empty(4, dtype=torch.float64)
    if k == 3:
        q[3] = cand[3]; q[0] = R[2, 1] - R[1, 2]; q[1] = R[0, 2] - R[2, 0]; q[2] = R[1, 0] - R[0, 1]
    else:
        i, j, l = k, (k + 1) % 3, (k + 2) % 3
        q[i] = cand[k]; q[j] = R[j, i] + R[i, j]; q[l] = R[l, i] + R[i, l]; q[3] = R[l, j] - R[j, l]
    return q / q.norm()


def overlap_statistics(sq_distances, overlapped_th, new_point_th):
    """Keyframe statistics from get_source_correspondence()'s squared distances [REF mp_Tracker.py:231-239, 266-269, 433-439]:
    -> (ratio of trackable points with d^2 < overlapped_th, indices of points with d^2 > new_point_th — the not-yet-mapped ones)."""
    d = torch.as_tensor(sq_distances)
    ratio = float((d < overlapped_th).sum()) / max(int(d.shape[0]), 1)
    return ratio, torch.where(d > new_point_th)[0]
