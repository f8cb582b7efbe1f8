"""Deterministic synthetic inputs shaped like the reference's datasets (S-pair, S-tum, S-map).

No dataset ships with the reference tree and there is no network, so every test and benchmark in this repo
runs on these analytic scenes.  Camera conventions restate the reference's helpers:
  * world->view and projection matrices, row-vector (pre-transposed) form: scene/shared_objs.py:8-43, 157-172
  * depth -> point back-projection with a strided pixel pick: mp_Tracker.py:394-431
Pure numpy; nothing here touches the GPU or the oracle.
"""
import math

import numpy as np

REPLICA = dict(W=1200, H=680, fx=600.0, fy=600.0, cx=599.5, cy=339.5, depth_scale=6553.5, depth_trunc=12.0,
               stride=10, max_corr=0.02)  # configs/Replica/caminfo.txt:3 ; replica.sh:135-142
TUM = dict(W=640, H=480, fx=517.3, fy=516.5, cx=318.6, cy=255.3, depth_scale=5000.0, depth_trunc=3.0,
           stride=5, max_corr=0.03)       # configs/TUM/rgbd_dataset_freiburg1_desk.txt:3 ; tum.sh:135-142

ROOM_LO = np.array([-3.0, -1.5, -2.0])
ROOM_HI = np.array([3.0, 1.5, 2.0])
CUBOIDS = [  # (lo, hi) interior boxes
    (np.array([-2.2, 0.3, 0.9]), np.array([-1.0, 1.5, 1.7])),
    (np.array([0.6, 0.7, 1.1]), np.array([1.9, 1.5, 1.9])),
    (np.array([-0.5, -0.4, 1.2]), np.array([0.3, 0.2, 1.6])),
]


# ------------------------------------------------------------------------------------------ cameras
def focal2fov(focal, pixels):
    return 2.0 * math.atan(pixels / (2.0 * focal))


def world2view(Rc2w, t_w2c):
    """4x4 world->camera matrix from a camera-to-world rotation and world->camera translation
    (the (R, t) pair the tracker hands to SharedCam.setup_cam, mp_Tracker.py:224-226, 277)."""
    Rt = np.zeros((4, 4))
    Rt[:3, :3] = Rc2w.T
    Rt[:3, 3] = t_w2c
    Rt[3, 3] = 1.0
    return Rt


def projection_matrix(znear, zfar, fovX, fovY):
    tY, tX = math.tan(fovY / 2), math.tan(fovX / 2)
    top, right = tY * znear, tX * znear
    P = np.zeros((4, 4))
    P[0, 0] = 2.0 * znear / (2 * right)
    P[1, 1] = 2.0 * znear / (2 * top)
    P[3, 2] = 1.0
    P[2, 2] = zfar / (zfar - znear)
    P[2, 3] = -(zfar * znear) / (zfar - znear)
    return P


def make_camera(W, H, fx, fy, pose_c2w=None, znear=0.01, zfar=100.0):
    """Returns the raster-settings camera block: viewmatrix/projmatrix in row-vector form (float32),
    campos, tanfovx, tanfovy."""
    if pose_c2w is None:
        pose_c2w = np.eye(4)
    w2c = np.linalg.inv(pose_c2w)
    fovx, fovy = focal2fov(fx, W), focal2fov(fy, H)
    view = np.float32(w2c).T.copy()                     # world_view_transform = (W2C)^T
    proj = np.float32(projection_matrix(znear, zfar, fovx, fovy)).T
    full = (view @ proj).astype(np.float32)             # full_proj_transform = view^T-form @ proj^T-form
    campos = np.linalg.inv(view.astype(np.float64))[3, :3].astype(np.float32)
    return dict(viewmatrix=view, projmatrix=full, campos=campos, tanfovx=math.tan(fovx * 0.5), tanfovy=math.tan(fovy * 0.5),
                W=W, H=H)


def se3(rot_axis_angle_deg=(0.0, 0.0, 0.0), trans=(0.0, 0.0, 0.0)):
    rx, ry, rz = [math.radians(a) for a in rot_axis_angle_deg]
    cx, sx, cy, sy, cz, sz = math.cos(rx), math.sin(rx), math.cos(ry), math.sin(ry), math.cos(rz), math.sin(rz)
    Rx = np.array([[1, 0, 0], [0, cx, -sx], [0, sx, cx]])
    Ry = np.array([[cy, 0, sy], [0, 1, 0], [-sy, 0, cy]])
    Rz = np.array([[cz, -sz, 0], [sz, cz, 0], [0, 0, 1]])
    T = np.eye(4)
    T[:3, :3] = Rz @ Ry @ Rx
    T[:3, 3] = trans
    return T


DEFAULT_POSE_A = None  # set below, after se3()


# ------------------------------------------------------------------------------------------ ray-cast room
def _slab(o, d, lo, hi):
    with np.errstate(divide="ignore", invalid="ignore"):
        t0 = (lo - o) / d
        t1 = (hi - o) / d
    tmin = np.minimum(t0, t1)
    tmax = np.maximum(t0, t1)
    return tmin.max(axis=-1), tmax.min(axis=-1)


def raycast_depth(cfg, pose_c2w):
    """Exact z-depth image (float64 metres) of the analytic room from a camera-to-world pose."""
    W, H = cfg["W"], cfg["H"]
    u, v = np.meshgrid(np.arange(W, dtype=np.float64), np.arange(H, dtype=np.float64))
    dc = np.stack([(u - cfg["cx"]) / cfg["fx"], (v - cfg["cy"]) / cfg["fy"], np.ones_like(u)], -1)
    d = dc @ pose_c2w[:3, :3].T
    o = pose_c2w[:3, 3]
    _, t_exit = _slab(o, d, ROOM_LO, ROOM_HI)
    depth = t_exit
    for lo, hi in CUBOIDS:
        te, tx = _slab(o, d, lo, hi)
        hit = (te < tx) & (te > 0)
        depth = np.where(hit & (te < depth), te, depth)
    return depth


DEFAULT_POSE_A = se3((12.0, 28.0, 0.0), (-0.9, -0.2, -1.1))


def checker_colors(points_w):
    """25 cm 3-D checker + a smooth tint.  The lattice is offset by 0.31 cells: every wall and cuboid face of the analytic room lies on a
    multiple of 0.1 m, i.e. ON a lattice plane of an un-offset checker (x = 3.0 -> floor(12.0 +- 1e-15)), where the cell parity of a
    ray-cast hit point is decided by its last rounding bit — that was salt-and-pepper noise of +-0.3 on half the pixels of rounds 1-2's
    synthetic images and capped every PSNR measured on them at ~18 dB (profiles/r03_map_quality.md)."""
    c = (np.floor(points_w * 4.0 + 0.31).astype(np.int64).sum(-1) & 1).astype(np.float32)
    base = 0.25 + 0.5 * c[:, None]
    tint = 0.5 + 0.5 * np.sin(points_w * np.array([1.3, 2.1, 0.7]))
    return np.clip(base * 0.6 + 0.4 * tint, 0, 1).astype(np.float32)


def downsample_indices(W, H, stride):
    """Pixel pick of mp_Tracker.py:394-413: rows stride*k-1 (row 0 for k=0), every stride-th column."""
    h_val = stride * np.arange(0, int(H / stride) + 1) - 1
    h_val[0] = 0
    cols = np.arange(0, W, stride)
    idx = (h_val[:, None] * W + cols[None, :]).flatten()
    return idx[idx < W * H]


def frame_points(cfg, pose_c2w, noise_seed=None, holes=0.0):
    """Depth image -> quantised uint16 -> strided camera-frame points, like
    Tracker.downsample_and_make_pointcloud2 (mp_Tracker.py:415-431).
    Returns points (M,3) f32, z (M,), trackable indices, and the metric depth image."""
    W, H = cfg["W"], cfg["H"]
    depth = raycast_depth(cfg, pose_c2w)
    if noise_seed is not None:
        rng = np.random.default_rng(noise_seed)
        sigma = 0.0012 + 0.0019 * (depth - 0.4) ** 2
        depth = depth + rng.normal(size=depth.shape) * sigma
        if holes > 0:
            depth = np.where(rng.random(depth.shape) < holes, 0.0, depth)
    d16 = np.clip(np.round(depth * cfg["depth_scale"]), 0, 65535).astype(np.uint16)
    idx = downsample_indices(W, H, cfg["stride"])
    u = (idx % W).astype(np.float32)
    v = (idx // W).astype(np.float32)
    x_pre = (u - np.float32(cfg["cx"])) / np.float32(cfg["fx"])
    y_pre = (v - np.float32(cfg["cy"])) / np.float32(cfg["fy"])
    z = d16.flatten()[idx].astype(np.float32) / np.float32(cfg["depth_scale"])
    nz = z != 0
    z = z[nz]
    pts = np.stack([x_pre[nz] * z, y_pre[nz] * z, z], -1).astype(np.float32)
    trackable = np.where(z <= cfg["depth_trunc"])[0]
    depth_m = d16.astype(np.float32) / np.float32(cfg["depth_scale"])
    return pts, z, trackable, depth_m


def s_pair(cfg=REPLICA, noise=False, motion=None):
    """Two frames with known relative motion: frame B = frame A o motion.  TUM-shaped default: 1 deg about y and
    2 cm along x.  Replica-shaped default: (0.2 deg x, 0.3 deg y, 8 mm x, 3 mm z) — with Replica's 2 cm
    correspondence gate and stride-10 sampling (3-7 cm point spacing) the 1 deg / 2 cm motion of SURVEY.md §8d lies
    outside GICP's convergence basin (the CPU oracle slides by 69 mm); real Replica inter-frame motion is ~1 cm.
    Frame A looks into a room corner (floor + two walls + cuboids in view) so that all six degrees of freedom are
    observable; from the room centre looking +z only one flat wall is visible and GICP slides along it."""
    pose_a = DEFAULT_POSE_A.copy()
    if motion is None:
        motion = se3((0.2, 0.3, 0.0), (0.008, 0.0, 0.003)) if cfg["max_corr"] < 0.025 else se3((0.0, 1.0, 0.0), (0.02, 0.0, 0.0))
    pose_b = pose_a @ motion
    kw = dict(noise_seed=1, holes=0.15) if noise else {}
    pa = frame_points(cfg, pose_a, **({"noise_seed": 11, "holes": 0.15} if noise else {}))
    pb = frame_points(cfg, pose_b, **kw)
    return dict(cfg=cfg, pose_a=pose_a, pose_b=pose_b, points_a=pa[0], z_a=pa[1], trackable_a=pa[2],
                points_b=pb[0], z_b=pb[1], trackable_b=pb[2], depth_a=pa[3], depth_b=pb[3])


def s_pair_survey(cfg=REPLICA, noise=False):
    """The S-pair exactly as SURVEY.md 8(d) words it: frame A at the room centre looking +z, frame B = A o (1 deg about y,
    then 2 cm along x).  From there a single wall and the three cuboids are in view.  With Replica's 2 cm gate the motion sits
    at the edge of GICP's convergence basin (the 1 deg rotation alone moves points at 2 m depth by 3.5 cm): the pair is used to
    pin HIP == oracle on a hard case and to time the long-iteration regime, not as a tracking-accuracy claim."""
    pose_a = np.eye(4)
    motion = se3((0.0, 1.0, 0.0), (0.02, 0.0, 0.0))
    pose_b = pose_a @ motion
    pa = frame_points(cfg, pose_a, **({"noise_seed": 11, "holes": 0.15} if noise else {}))
    pb = frame_points(cfg, pose_b, **({"noise_seed": 1, "holes": 0.15} if noise else {}))
    return dict(cfg=cfg, pose_a=pose_a, pose_b=pose_b, points_a=pa[0], z_a=pa[1], trackable_a=pa[2],
                points_b=pb[0], z_b=pb[1], trackable_b=pb[2], depth_a=pa[3], depth_b=pb[3])


def render_frame(cfg, pose_c2w, noise_seed=None, holes=0.0):
    """One synthetic RGB-D frame as a sensor would deliver it: (rgb uint8 (H,W,3), depth uint16 (H,W)) — the ray-cast room
    coloured by `checker_colors` at the hit points, depth quantised with the dataset's depth_scale."""
    W, H = cfg["W"], cfg["H"]
    depth = raycast_depth(cfg, pose_c2w)
    u, v = np.meshgrid(np.arange(W, dtype=np.float64), np.arange(H, dtype=np.float64))
    pc = np.stack([(u - cfg["cx"]) / cfg["fx"] * depth, (v - cfg["cy"]) / cfg["fy"] * depth, depth], -1).reshape(-1, 3)
    rgb = (checker_colors(pc @ pose_c2w[:3, :3].T + pose_c2w[:3, 3]).reshape(H, W, 3) * 255.0).astype(np.uint8)
    if noise_seed is not None:
        rng = np.random.default_rng(noise_seed)
        depth = depth + rng.normal(size=depth.shape) * (0.0012 + 0.0019 * (depth - 0.4) ** 2)
        if holes > 0:
            depth = np.where(rng.random(depth.shape) < holes, 0.0, depth)
    d16 = np.clip(np.round(depth * cfg["depth_scale"]), 0, 65535).astype(np.uint16)
    return rgb, d16


def trajectory(n_frames, step=None, start=None, speed=1.0, jitter=0.0, jitter_seed=0):
    """Camera-to-world poses of a smooth synthetic sequence of ANY length that stays inside the analytic room: the camera rides a tilted
    ellipse in the cuboid-free half of the room (|x - c_x| <= 0.75 m, |z - c_z| <= 0.55 m) while its yaw and pitch sway slowly, so that floor,
    walls and the interior cuboids stay in view.  Per frame at most ~7 mm and ~0.25 deg — the order of Replica's inter-frame motion.
    (With `step` given: the older `start o step^k` path, which leaves the room after ~230 frames at the default step.)"""
    if step is not None:
        poses = [DEFAULT_POSE_A.copy() if start is None else start.copy()]
        for _ in range(n_frames - 1):
            poses.append(poses[-1] @ step)
        return poses
    # `speed` scales the motion per frame (2 = 14 mm / 0.5 deg: a fast hand-held sweep); `jitter` adds a zero-mean hand tremor per frame
    # (metres of translation; the same number x 10 in degrees of rotation about every axis), deterministic in `jitter_seed`
    rng = np.random.default_rng(jitter_seed) if jitter > 0 else None
    poses = []
    for k in range(n_frames):
        th = 0.009 * k * speed
        pos = np.array([-0.1 + 0.75 * math.cos(th), -0.1 + 0.12 * math.sin(2.0 * th), -0.9 + 0.55 * math.sin(th)])
        ang = np.array([10.0 + 6.0 * math.sin(1.1 * th), 30.0 + 35.0 * math.sin(0.7 * th), 0.0])
        if rng is not None and k > 0:
            pos = pos + rng.normal(0.0, jitter, 3)
            ang = ang + rng.normal(0.0, 10.0 * jitter, 3)
        poses.append(se3(tuple(ang), tuple(pos)))
    return poses


def raycast_pixels(cfg, pose_c2w, u, v):
    """`raycast_depth` at the given (float) pixel coordinates only — same slab arithmetic, no full image."""
    u, v = np.asarray(u, np.float64), np.asarray(v, np.float64)
    dc = np.stack([(u - cfg["cx"]) / cfg["fx"], (v - cfg["cy"]) / cfg["fy"], np.ones_like(u)], -1)
    d = dc @ pose_c2w[:3, :3].T
    o = pose_c2w[:3, 3]
    _, depth = _slab(o, d, ROOM_LO, ROOM_HI)
    for lo, hi in CUBOIDS:
        te, tx = _slab(o, d, lo, hi)
        hit = (te < tx) & (te > 0)
        depth = np.where(hit & (te < depth), te, depth)
    return depth


def tracker_map_cloud(n_points, n_keyframes=32, seed=5, cfg=REPLICA, keyframe_every=10, first_frame=0, noise=False):
    """World-frame points of a MAP as the tracker sees it after many keyframes [REF mp_Tracker.py:282-288; scene/gaussian_model.py:207-215]:
    the union of `n_keyframes` keyframes of `trajectory` (every `keyframe_every`-th frame), each contributing n_points / n_keyframes
    back-projected depth samples at randomly picked integer pixels (sensor-quantised depth, the front-end's float32 arithmetic), so the
    keyframes overlap on the same surfaces the way a real map does.  Returns dict(points (K,3) f32 world, keyframe (K,) int — which keyframe
    a row came from, z (K,) f32 camera depth, poses: the keyframe poses, frame_ids).  Rows are in keyframe order; callers shuffle."""
    rng = np.random.default_rng(seed)
    poses_all = trajectory(first_frame + keyframe_every * n_keyframes + keyframe_every)
    W, H = cfg["W"], cfg["H"]
    per = [n_points // n_keyframes + (1 if k < n_points % n_keyframes else 0) for k in range(n_keyframes)]
    pts, kf, zs, poses, ids = [], [], [], [], []
    for k in range(n_keyframes):
        fid = first_frame + keyframe_every * k
        pose = poses_all[fid]
        pix = rng.choice(W * H, size=per[k], replace=False) if per[k] <= W * H else rng.integers(0, W * H, per[k])
        u, v = (pix % W).astype(np.float32), (pix // W).astype(np.float32)
        depth = raycast_pixels(cfg, pose, u, v)
        if noise:      # the sensor model of frame_points (SURVEY 8d S-tum): the map's keyframes carry the noise of the frames they came from
            depth = depth + rng.normal(size=depth.shape) * (0.0012 + 0.0019 * (depth - 0.4) ** 2)
        d16 = np.clip(np.round(depth * cfg["depth_scale"]), 0, 65535).astype(np.uint16)
        z = d16.astype(np.float32) / np.float32(cfg["depth_scale"])
        x_pre = (u - np.float32(cfg["cx"])) / np.float32(cfg["fx"])
        y_pre = (v - np.float32(cfg["cy"])) / np.float32(cfg["fy"])
        pc = np.stack([x_pre * z, y_pre * z, z], -1).astype(np.float32)
        keep = (z != 0) & (z <= cfg["depth_trunc"])
        pc, z = pc[keep], z[keep]
        pw = (pc.astype(np.float64) @ pose[:3, :3].T + pose[:3, 3]).astype(np.float32)
        pts.append(pw); kf.append(np.full(len(pw), k, np.int32)); zs.append(z); poses.append(pose); ids.append(fid)
    return dict(points=np.concatenate(pts), keyframe=np.concatenate(kf), z=np.concatenate(zs), poses=poses, frame_ids=ids)


def tracker_map(n_target, cov_fn, n_keyframes=32, seed=5, cfg=REPLICA, pass_fraction=0.5, opacity_th=0.05, noise=False):
    """A synthetic MAP of Gaussians sized so that ~n_target of them pass the tracker hand-off's selection
    `opacity > opacity_th and trackable` [REF scene/gaussian_model.py:207-215]: n_target / pass_fraction rows in random order, positions from
    `tracker_map_cloud`, rotations (xyzw) / scales from `cov_fn(world_points_of_one_keyframe) -> (rots (n,4), scales (n,3))` — the k-NN
    covariance export of whichever registration object the caller hands in (the oracle in tests, the HIP tracker in bench.py) — with the
    scales shrunk as the mapper's initialisation shrinks them (scales / clamp_min(2 z^1.5, 1)) [REF scene/gaussian_model.py:143-145].
    Rows failing the selection are split between non-trackable ones and low-opacity ones."""
    total = int(round(n_target / pass_fraction))
    cloud = tracker_map_cloud(total, n_keyframes=n_keyframes, seed=seed, cfg=cfg, noise=noise)
    pts, kf, z = cloud["points"], cloud["keyframe"], cloud["z"]
    rots = np.zeros((len(pts), 4), np.float32)
    scales = np.zeros((len(pts), 3), np.float32)
    for k in range(n_keyframes):
        sel = np.where(kf == k)[0]
        r, s = cov_fn(pts[sel])
        rots[sel] = np.reshape(r, (-1, 4))
        scales[sel] = np.reshape(s, (-1, 3))
    scales = (scales / np.maximum(2.0 * z[:, None] ** 1.5, 1.0)).astype(np.float32)
    rng = np.random.default_rng(seed + 1)
    perm = rng.permutation(len(pts))
    pts, rots, scales = pts[perm], rots[perm], scales[perm]
    u = rng.random(len(pts))
    fail = u >= pass_fraction
    trackable = ~(fail & (u < pass_fraction + 0.5 * (1.0 - pass_fraction)))          # first half of the failing rows: not trackable
    opacity = np.where(fail & trackable, rng.uniform(0.0, opacity_th, len(pts)),      # second half: opacity at or below the threshold
                       rng.uniform(opacity_th + 1e-3, 1.0, len(pts))).astype(np.float32)
    return dict(points=pts, rotations=rots, scales=scales, opacity=opacity, trackable=trackable, opacity_th=float(opacity_th),
                poses=cloud["poses"], frame_ids=cloud["frame_ids"])


# ------------------------------------------------------------------------------------------ surfel map
def _faces():
    """(origin, edge_u, edge_v, inward/outward normal) rectangles of the room (facing in) and cuboids (facing out)."""
    faces = []

    def box(lo, hi, inward):
        sgn = -1.0 if inward else 1.0
        for ax in range(3):
            a1, a2 = (ax + 1) % 3, (ax + 2) % 3
            for side, val in ((0, lo[ax]), (1, hi[ax])):
                o = lo.copy()
                o[ax] = val
                eu = np.zeros(3); eu[a1] = hi[a1] - lo[a1]
                ev = np.zeros(3); ev[a2] = hi[a2] - lo[a2]
                n = np.zeros(3); n[ax] = (1.0 if side else -1.0) * sgn
                faces.append((o, eu, ev, n))
    box(ROOM_LO, ROOM_HI, True)
    for lo, hi in CUBOIDS:
        box(lo, hi, False)
    return faces


def quat_from_R_batch(Rm):
    """Vectorised (N,3,3) -> (N,4) xyzw, w >= 0 branch-free variant (Shepperd via max component)."""
    m = Rm
    t = np.stack([1 + m[:, 0, 0] - m[:, 1, 1] - m[:, 2, 2], 1 - m[:, 0, 0] + m[:, 1, 1] - m[:, 2, 2],
                  1 - m[:, 0, 0] - m[:, 1, 1] + m[:, 2, 2], 1 + m[:, 0, 0] + m[:, 1, 1] + m[:, 2, 2]], -1)
    k = t.argmax(-1)
    q = np.zeros((m.shape[0], 4))
    for c in range(4):
        sel = k == c
        if not sel.any():
            continue
        ms, s = m[sel], 2 * np.sqrt(t[sel, c])
        if c == 0:
            q[sel] = np.stack([0.25 * s, (ms[:, 0, 1] + ms[:, 1, 0]) / s, (ms[:, 0, 2] + ms[:, 2, 0]) / s, (ms[:, 2, 1] - ms[:, 1, 2]) / s], -1)
        elif c == 1:
            q[sel] = np.stack([(ms[:, 0, 1] + ms[:, 1, 0]) / s, 0.25 * s, (ms[:, 1, 2] + ms[:, 2, 1]) / s, (ms[:, 0, 2] - ms[:, 2, 0]) / s], -1)
        elif c == 2:
            q[sel] = np.stack([(ms[:, 0, 2] + ms[:, 2, 0]) / s, (ms[:, 1, 2] + ms[:, 2, 1]) / s, 0.25 * s, (ms[:, 1, 0] - ms[:, 0, 1]) / s], -1)
        else:
            q[sel] = np.stack([(ms[:, 2, 1] - ms[:, 1, 2]) / s, (ms[:, 0, 2] - ms[:, 2, 0]) / s, (ms[:, 1, 0] - ms[:, 0, 1]) / s, 0.25 * s], -1)
    return q


def s_map(P=300_000, seed=2, perturb_seed=None):
    """P surfels on the room surfaces: activated parameters as the mapper would hand them to the rasteriser
    (means, scales = std-devs, normalised xyzw quaternions, opacity in (0,1), SH DC)."""
    rng = np.random.default_rng(seed)
    faces = _faces()
    areas = np.array([np.linalg.norm(np.cross(f[1], f[2])) for f in faces])
    counts = rng.multinomial(P, areas / areas.sum())
    means, Rs = [], []
    for (o, eu, ev, n), c in zip(faces, counts):
        a, b = rng.random(c), rng.random(c)
        means.append(o[None] + a[:, None] * eu[None] + b[:, None] * ev[None])
        u = eu / np.linalg.norm(eu)
        v = np.cross(n, u)
        th = rng.random(c) * 2 * np.pi
        ex = np.cos(th)[:, None] * u[None] + np.sin(th)[:, None] * v[None]
        ey = np.cross(np.broadcast_to(n, ex.shape), ex)
        Rs.append(np.stack([ex, ey, np.broadcast_to(n, ex.shape)], -1))  # columns = local axes, z = normal
    means = np.concatenate(means).astype(np.float32)
    Rm = np.concatenate(Rs)
    quats = quat_from_R_batch(Rm).astype(np.float32)
    tang = np.exp(rng.normal(math.log(0.01), 0.5, size=(P, 2)))
    scales = np.concatenate([tang, 0.1 * tang.min(-1, keepdims=True)], -1).astype(np.float32)
    opac = (1.0 / (1.0 + np.exp(-rng.normal(1.0, 2.0, size=P)))).astype(np.float32)
    rgb = rng.random((P, 3)).astype(np.float32)
    sh_dc = ((rgb - 0.5) / 0.28209479177387814).astype(np.float32)
    out = dict(means3D=means, scales=scales, rotations=quats, opacities=opac[:, None], shs=sh_dc[:, None, :], rgb=rgb)
    if perturb_seed is not None:
        pr = np.random.default_rng(perturb_seed)
        out["means3D"] = (means + pr.normal(0, 0.002, means.shape)).astype(np.float32)
        out["shs"] = (out["shs"] + pr.normal(0, 0.2, out["shs"].shape)).astype(np.float32)
        out["opacities"] = np.clip(out["opacities"] * np.exp(pr.normal(0, 0.2, (P, 1))), 1e-3, 0.999).astype(np.float32)
        out["scales"] = (out["scales"] * np.exp(pr.normal(0, 0.1, (P, 3)))).astype(np.float32)
    return out


def random_gaussians(P, seed=0, sh_degree=0, spread=1.0, zmin=1.0, zmax=6.0):
    """Small random cloud in front of an identity camera — unit-test input."""
    rng = np.random.default_rng(seed)
    means = np.stack([rng.uniform(-spread, spread, P), rng.uniform(-spread * 0.6, spread * 0.6, P), rng.uniform(zmin, zmax, P)], -1)
    q = rng.normal(size=(P, 4)); q /= np.linalg.norm(q, axis=1, keepdims=True)
    scales = np.exp(rng.normal(math.log(0.05), 0.6, size=(P, 3)))
    opac = rng.uniform(0.05, 0.95, size=(P, 1))
    M = (sh_degree + 1) ** 2
    shs = rng.normal(0, 0.5, size=(P, M, 3))
    shs[:, 0] += 0.5
    f32 = np.float32
    return dict(means3D=means.astype(f32), rotations=q.astype(f32), scales=scales.astype(f32), opacities=opac.astype(f32),
                shs=shs.astype(f32))
