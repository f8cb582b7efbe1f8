"""TEST INFRASTRUCTURE ONLY (imported by tests/ only).  CPU restatement of the reference tracker's per-frame front-end
[REF mp_Tracker.py:393-431] with plain torch-CPU / numpy ops, used to check gs_icp_slam_amd/frontend.py + csrc/frontend.hip.
The reference file itself cannot be imported here (it needs cv2 / open3d / rerun), so parity is pinned to this restatement of
its fifteen lines of tensor arithmetic, operation by operation."""
import numpy as np
import torch


def downsample_filter(H, W, fx, fy, cx, cy, scale):
    """pick indices and pre-divided pixel coordinates [REF mp_Tracker.py:393-413]"""
    row_starts = scale * torch.arange(0, int(H / scale) + 1)
    row_starts = row_starts - 1
    row_starts[0] = 0
    row_starts = row_starts * W
    grid_r, grid_c = torch.meshgrid(row_starts, torch.arange(0, W, scale), indexing="ij")
    pick = (grid_r + grid_c).flatten()
    vv, uu = torch.meshgrid(torch.arange(0, H), torch.arange(0, W), indexing="ij")
    u = uu.flatten()[pick]
    v = vv.flatten()[pick]
    return pick, (u - cx) / fx, (v - cy) / fy


def make_pointcloud(depth_img, rgb_img, pick, x_pre, y_pre, depth_scale, depth_trunc):
    """[REF mp_Tracker.py:415-431] -> points (n,3), colors (n,3), z (n,), trackable indices (m,)"""
    col = torch.from_numpy(rgb_img).reshape(-1, 3).float()[pick] / 255
    z = torch.from_numpy(depth_img.astype(np.float32)).flatten()[pick] / depth_scale
    nz = torch.where(z != 0)
    near = torch.where(z[nz] <= depth_trunc)
    z = z[nz]
    pts = torch.stack([x_pre[nz] * z, y_pre[nz] * z, z], dim=-1)
    return pts.numpy(), col[nz].numpy(), z.numpy(), near[0].numpy()
