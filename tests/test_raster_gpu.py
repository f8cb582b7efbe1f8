"""GPU parity: HIP rasteriser (through the drop-in diff_gaussian_rasterization API / C ABI) vs the CPU oracle.

Bars (BASELINE.json north_star): bit-exact tile / Gaussian indices; RGB and depth within 1e-5 ON >= 99.98 % OF THE PIXELS (>= 99.96 % at the
training-stage sizes).  exp() differs in the last ulps between libm and the GPU, so a pixel whose alpha / transmittance test sits within 1e-5
(relative) of its threshold may legitimately flip; the oracle reports that margin per pixel and such "fragile" pixels (<= 2e-4 of the image,
asserted) are excluded from the 1e-5 bar but bounded in number.  The adversarial long-list scenes (thousands of threshold decisions per pixel)
and the UHD scene run at stated looser tolerances (5e-5 / 2e-4 / 1e-3) — see each test.
"""
import numpy as np
import pytest

from gs_icp_slam_amd import synth
from tests import util

pytestmark = pytest.mark.gpu
TOL = 1e-5
FRAGILE = 1e-5


def run_product(g, cam, bg, sh_degree=0, grads=None, tile_mod=1, tile_rem=0, depth_mode=0):
    import torch
    from diff_gaussian_rasterization import GaussianRasterizer
    t = util.torch_inputs(g, requires_grad=grads is not None)
    rs = util.make_settings(cam, bg, sh_degree, tile_mod=tile_mod, tile_rem=tile_rem, depth_mode=depth_mode)
    means2D = torch.zeros_like(t["means3D"], requires_grad=True)
    rast = GaussianRasterizer(raster_settings=rs)
    depth, color, radii, is_used = rast(means3D=t["means3D"], means2D=means2D, shs=t.get("shs"), colors_precomp=t.get("colors_precomp"),
                                        opacities=t["opacities"], scales=t.get("scales"), rotations=t.get("rotations"),
                                        cov3D_precomp=t.get("cov3D_precomp"))
    fn = depth.grad_fn
    out = dict(depth=depth.detach().cpu().numpy()[0], color=color.detach().cpu().numpy(), radii=radii.cpu().numpy(),
               is_used=is_used.cpu().numpy())
    if fn is not None:
        saved = fn.saved_tensors
        out["scratch"] = (saved[7], saved[8], saved[9])
        out["num_rendered"] = fn.num_rendered
    if grads is not None:
        gc, gd = grads
        loss = (color * torch.from_numpy(gc).cuda()).sum() + (depth[0] * torch.from_numpy(gd).cuda()).sum()
        loss.backward()
        out["grads"] = {k: (v.grad.detach().cpu().numpy() if v.grad is not None else None) for k, v in t.items()}
        out["grads"]["means2D"] = means2D.grad.detach().cpu().numpy()
    return out


def check_forward(g, cam, bg, sh_degree, hip_lib, label, tol=TOL, max_fragile=2e-4, fp32_noise_cap=None):
    """`fp32_noise_cap` (trained maps, lists of 300-400 entries per pixel): the image bar becomes min(max(tol, 1e-7 n_contrib), cap) per pixel —
    two valid fp32 evaluation orders of a 400-term transmittance product drift apart by up to ~N ulps (measured 2.1e-5 at N = 402, where the
    fp32 oracle is as far from its own fp64 instance) — and the number of pixels beyond `tol` is returned (at most 1e-4 of the image)."""
    o = util.oracle_forward(g, cam, bg, sh_degree)
    p = run_product(g, cam, bg, sh_degree)
    P = g["means3D"].shape[0]
    W, H = cam["W"], cam["H"]
    # ---- integer outputs: bit-exact
    assert np.array_equal(p["radii"], o["radii"]), f"{label}: radii differ at {np.flatnonzero(p['radii'] != o['radii'])[:10]}"
    assert p["num_rendered"] == o["num_rendered"], f"{label}: num_rendered {p['num_rendered']} vs {o['num_rendered']}"
    s = util.read_scratch(hip_lib, p["scratch"], P, p["num_rendered"], W, H)
    assert np.array_equal(s["point_list"], o["point_list"]), f"{label}: sorted Gaussian list differs"
    assert np.array_equal(s["tile_keys"], (o["keys"] >> np.uint64(32)).astype(np.uint32)), f"{label}: sorted tile keys differ"
    assert np.array_equal(s["ranges"], o["ranges"]), f"{label}: tile ranges differ"
    vis = o["radii"] > 0
    # per-Gaussian geometry (float, same op order, no FMA): bit-exact centre/depth, conic to 1 ulp-ish
    assert np.array_equal(s["rec"][vis, 0:3], o["geom"][vis, 0:3]), f"{label}: pixel centre / depth not bit-exact"
    np.testing.assert_allclose(s["rec"][vis, 4:8], o["geom"][vis, 3:7], rtol=1e-6, atol=1e-30)
    np.testing.assert_allclose(s["rec"][vis, 8:11], o["geom"][vis, 7:10], rtol=1e-6, atol=1e-7)
    # ---- images
    ok = o["margin"] > FRAGILE
    frac_fragile = 1.0 - ok.mean()
    assert frac_fragile < max_fragile, f"{label}: {frac_fragile:.2e} fragile pixels"
    dc = np.abs(p["color"] - o["color"]).max(0)
    dd = np.abs(p["depth"] - o["depth"]) / np.maximum(1.0, np.abs(o["depth"]))
    extra = {}
    if fp32_noise_cap is None:
        assert dc[ok].max() <= tol, f"{label}: colour err {dc[ok].max():.3e}"
        assert dd[ok].max() <= tol, f"{label}: depth err {dd[ok].max():.3e}"
    else:
        # per-pixel bar: tol, or fp32's drift over the pixel's own blend list where that is larger (1e-7 = 1.7 ulp per blended entry)
        tol_px = np.minimum(np.maximum(tol, 1e-7 * o["n_contrib"].astype(np.float64)), fp32_noise_cap)
        over = ok & ((dc > tol) | (dd > tol))
        extra = dict(pixels_over_1e5=int(over.sum()), max_n_contrib=int(o["n_contrib"].max()),
                     min_n_contrib_of_a_pixel_over_1e5=int(o["n_contrib"][over].min()) if over.any() else None,
                     worst_ratio_to_bound=float(max((dc / tol_px)[ok].max(), (dd / tol_px)[ok].max())))
        assert over.sum() <= 1e-4 * W * H, f"{label}: {int(over.sum())} robust pixels beyond {tol}"
        assert (dc <= tol_px)[ok].all(), f"{label}: colour err {dc[ok].max():.3e} beyond max(1e-5, 1e-7 n_contrib) at {int((dc > tol_px)[ok].sum())} pixels"
        assert (dd <= tol_px)[ok].all(), f"{label}: depth err {dd[ok].max():.3e} beyond max(1e-5, 1e-7 n_contrib)"
    assert np.array_equal(s["n_contrib"][ok], o["n_contrib"][ok]), \
        f"{label}: n_contrib differs on {int((s['n_contrib'] != o['n_contrib'])[ok].sum())} robust pixels (margins {np.sort(o['margin'][ok & (s['n_contrib'] != o['n_contrib'])])[:5]})"
    np.testing.assert_allclose(s["final_T"][ok], o["final_T"][ok], rtol=max(1e-4, 20 * tol), atol=1e-6)
    # is_used: robust subset relation (flips only through fragile pixels)
    diff = np.flatnonzero(p["is_used"] != o["is_used"])
    assert len(diff) <= max(2, int(1e-4 * P)), f"{label}: is_used differs for {len(diff)} Gaussians"
    return o, p, dict(fragile=float(frac_fragile), max_color_err=float(dc[ok].max()), max_depth_err=float(dd[ok].max()), **extra)


@pytest.mark.parametrize("deg", [0, 1, 2, 3])
def test_forward_small_random(hip_lib, deg):
    cam = synth.make_camera(160, 96, 120.0, 120.0)
    g = synth.random_gaussians(500, seed=10 + deg, sh_degree=deg)
    check_forward(g, cam, [0.1, 0.2, 0.3], deg, hip_lib, f"random deg{deg}")


def test_forward_ragged_sizes(hip_lib):
    # image not a multiple of 16, Gaussians behind the camera / off-screen / huge
    cam = synth.make_camera(203, 77, 150.0, 140.0)
    g = synth.random_gaussians(800, seed=3, spread=3.0, zmin=-2.0, zmax=5.0)
    g["scales"][:20] *= 30.0
    check_forward(g, cam, [0.0, 0.0, 0.0], 0, hip_lib, "ragged")


def test_forward_single_surfel_orientation(hip_lib):
    """F7: quaternions are (x,y,z,w).  A thin disc whose local z is rotated 90 deg about x must render as a
    horizontal streak, not a filled disc."""
    cam = synth.make_camera(128, 128, 100.0, 100.0)
    c = np.float32(np.sqrt(0.5))
    g = dict(means3D=np.array([[0, 0, 2.0]], np.float32), scales=np.array([[0.3, 0.3, 0.003]], np.float32),
             rotations=np.array([[c, 0, 0, c]], np.float32), opacities=np.array([[0.9]], np.float32),
             shs=np.array([[[1.0, 1.0, 1.0]]], np.float32))
    o, p, _ = check_forward(g, cam, [0, 0, 0], 0, hip_lib, "surfel")
    img = p["color"][0]
    ys, xs = np.nonzero(img > 0.05)
    assert (xs.max() - xs.min()) > 4 * (ys.max() - ys.min())


def test_forward_empty_and_all_culled(hip_lib):
    import torch
    from diff_gaussian_rasterization import GaussianRasterizer
    cam = synth.make_camera(64, 48, 60.0, 60.0)
    rs = util.make_settings(cam, [0.2, 0.3, 0.4])
    for P in (0, 5):
        g = synth.random_gaussians(max(P, 1), seed=1, zmin=-5.0, zmax=-1.0)  # all behind the camera
        t = util.torch_inputs({k: v[:P] for k, v in g.items()})
        depth, color, radii, used = GaussianRasterizer(rs)(means3D=t["means3D"], means2D=torch.zeros_like(t["means3D"]), shs=t["shs"],
                                                           opacities=t["opacities"], scales=t["scales"], rotations=t["rotations"])
        assert torch.allclose(color[:, 0, 0], torch.tensor([0.2, 0.3, 0.4], device="cuda"))
        assert float(depth.abs().max()) == 0.0 and int(radii.sum()) == 0 and int(used.sum()) == 0


def test_argument_validation():
    import torch
    from diff_gaussian_rasterization import GaussianRasterizer
    cam = synth.make_camera(64, 48, 60.0, 60.0)
    rs = util.make_settings(cam, [0, 0, 0])
    t = util.torch_inputs(synth.random_gaussians(4))
    with pytest.raises(Exception):
        GaussianRasterizer(rs)(means3D=t["means3D"], means2D=None, opacities=t["opacities"], scales=t["scales"], rotations=t["rotations"])
    with pytest.raises(Exception):
        GaussianRasterizer(rs)(means3D=t["means3D"], means2D=None, shs=t["shs"], opacities=t["opacities"], scales=t["scales"])


@pytest.mark.parametrize("deg", [0, 1, 2, 3])
def test_backward_small_random(hip_lib, deg):
    cam = synth.make_camera(160, 96, 120.0, 120.0)
    g = synth.random_gaussians(400, seed=20 + deg, sh_degree=deg)
    rng = np.random.default_rng(1)
    gc = rng.normal(size=(3, 96, 160)).astype(np.float32)
    gd = rng.normal(size=(96, 160)).astype(np.float32)
    bg = [0.3, 0.1, 0.2]
    o = util.oracle_backward(g, cam, bg, gc, gd, deg)
    p = run_product(g, cam, bg, deg, grads=(gc, gd))
    pairs = [("means3D", "dL_dmeans3D"), ("opacities", "dL_dopacity"), ("scales", "dL_dscales"), ("rotations", "dL_drots"),
             ("shs", "dL_dsh"), ("means2D", "dL_dmeans2D")]
    for name, key in pairs:
        a, b = p["grads"][name].reshape(-1), o[key].reshape(-1)
        scale = np.abs(b).max() + 1e-12
        err = np.abs(a - b).max() / scale
        assert err < 2e-4, f"deg{deg} grad {name}: rel-to-max err {err:.3e}"


def test_backward_gradcheck_against_fp64_oracle(hip_lib):
    """The f32 HIP gradients must agree with the oracle's fp64 analytic gradients (themselves finite-difference
    checked in tests/test_oracle_raster.py)."""
    cam = synth.make_camera(96, 64, 80.0, 80.0)
    g = synth.random_gaussians(60, seed=33, spread=0.7, zmin=1.5, zmax=4.0)
    rng = np.random.default_rng(2)
    gc = rng.normal(size=(3, 64, 96)).astype(np.float32)
    gd = rng.normal(size=(64, 96)).astype(np.float32)
    o = util.oracle_backward({k: v.astype(np.float64) for k, v in g.items()}, cam, [0, 0, 0], gc, gd, 0, dtype=np.float64)
    p = run_product(g, cam, [0, 0, 0], 0, grads=(gc, gd))
    for name, key in [("means3D", "dL_dmeans3D"), ("opacities", "dL_dopacity"), ("scales", "dL_dscales"), ("rotations", "dL_drots"), ("shs", "dL_dsh")]:
        a, b = p["grads"][name].reshape(-1).astype(np.float64), o[key].reshape(-1)
        err = np.abs(a - b).max() / (np.abs(b).max() + 1e-12)
        assert err < 5e-4, f"grad {name}: {err:.3e}"


def test_colors_precomp_and_cov_precomp_paths(hip_lib):
    cam = synth.make_camera(128, 80, 100.0, 100.0)
    g = synth.random_gaussians(300, seed=5)
    o0 = util.oracle_forward(g, cam, [0, 0, 0], 0)
    g2 = dict(means3D=g["means3D"], opacities=g["opacities"], colors_precomp=o0["geom"][:, 7:10].copy(), scales=g["scales"], rotations=g["rotations"])
    check_forward(g2, cam, [0, 0, 0], 0, hip_lib, "colors_precomp")


def _cov3d_from(scales, quats_xyzw):
    """6 unique entries (xx, xy, xz, yy, yz, zz) of R diag(s^2) R^T — what GaussianModel.get_covariance hands over as cov3D_precomp
    [REF scene/gaussian_model.py:28-33; utils/general_utils.py:73-123]."""
    x, y, z, r = [quats_xyzw[:, i].astype(np.float64) for i in range(4)]
    R = np.stack([1 - 2 * (y * y + z * z), 2 * (x * y - r * z), 2 * (x * z + r * y), 2 * (x * y + r * z), 1 - 2 * (x * x + z * z),
                  2 * (y * z - r * x), 2 * (x * z - r * y), 2 * (y * z + r * x), 1 - 2 * (x * x + y * y)], -1).reshape(-1, 3, 3)
    M = R * scales[:, None, :].astype(np.float64)
    S = M @ np.swapaxes(M, 1, 2)
    return np.stack([S[:, 0, 0], S[:, 0, 1], S[:, 0, 2], S[:, 1, 1], S[:, 1, 2], S[:, 2, 2]], -1).astype(np.float32)


def test_precomputed_covariance_and_colour_paths_forward_and_backward(hip_lib):
    """The two optional input forms of the API [REF gaussian_renderer/__init__.py:268-292: pipe.compute_cov3D_python / convert_SHs_python]:
    cov3D_precomp instead of (scales, rotations) and colors_precomp instead of SHs — forward lists / images and the gradients that reach
    the precomputed tensors themselves (dL/dcov3D, dL/dcolors), against the oracle."""
    cam = synth.make_camera(160, 96, 120.0, 120.0)
    g = synth.random_gaussians(400, seed=23)
    rng = np.random.default_rng(4)
    gp = dict(means3D=g["means3D"], opacities=g["opacities"], colors_precomp=rng.uniform(0, 1, (400, 3)).astype(np.float32),
              cov3D_precomp=_cov3d_from(g["scales"], g["rotations"]))
    bg = [0.2, 0.0, 0.1]
    check_forward(gp, cam, bg, 0, hip_lib, "cov3D_precomp + colors_precomp")
    gc = rng.normal(size=(3, 96, 160)).astype(np.float32)
    gd = rng.normal(size=(96, 160)).astype(np.float32)
    o = util.oracle_backward(gp, cam, bg, gc, gd, 0)
    o64 = util.oracle_backward({k: v.astype(np.float64) for k, v in gp.items()}, cam, bg, gc, gd, 0, dtype=np.float64)
    p = run_product(gp, cam, bg, 0, grads=(gc, gd))
    assert p["grads"].get("scales") is None and p["grads"].get("rotations") is None
    for name, key in [("means3D", "dL_dmeans3D"), ("opacities", "dL_dopacity"), ("cov3D_precomp", "dL_dcov3D"), ("colors_precomp", "dL_dcolors"),
                      ("means2D", "dL_dmeans2D")]:
        a, b, b64 = p["grads"][name].reshape(-1).astype(np.float64), o[key].reshape(-1).astype(np.float64), o64[key].reshape(-1)
        mx = np.abs(b).max()
        assert (np.abs(a - b) <= 2e-4 * mx + 1e-4 * np.abs(b) + 2 * np.abs(b - b64)).all(), f"{name}: {np.abs(a - b).max() / mx:.3e} of max"
    # the same scene through (scales, rotations) renders the same image (the covariance is the same matrix, up to its float32 rounding)
    gs = dict(means3D=g["means3D"], opacities=g["opacities"], colors_precomp=gp["colors_precomp"], scales=g["scales"], rotations=g["rotations"])
    q = run_product(gs, cam, bg, 0)
    assert np.abs(q["color"] - p["color"]).max() < 2e-3


def test_tile_sharding_composes_to_full_image(hip_lib):
    cam = synth.make_camera(208, 112, 150.0, 150.0)
    g = synth.random_gaussians(600, seed=8)
    g["scales"][:12] *= 25.0           # a few splats wider than 48 tiles: emitted by a whole wave, slot ranks counted per rank
    full = run_product(g, cam, [0.1, 0.1, 0.1], 0)
    acc_c = np.zeros_like(full["color"]); acc_d = np.zeros_like(full["depth"])
    tot = 0
    for r in range(3):
        part = run_product(g, cam, [0.1, 0.1, 0.1], 0, tile_mod=3, tile_rem=r)
        acc_c += part["color"]; acc_d += part["depth"]; tot += part["num_rendered"]
    assert tot == full["num_rendered"]
    assert np.array_equal(acc_c, full["color"]) and np.array_equal(acc_d, full["depth"])


@pytest.mark.parametrize("res", ["replica", "tum"])
def test_full_size_smap(hip_lib, res):
    """BASELINE sizes: P = 300 k surfels, 1200x680 and 640x480.  Full oracle comparison (the C oracle needs a few
    seconds) plus size-independent properties."""
    cfg = synth.REPLICA if res == "replica" else synth.TUM
    cam = synth.make_camera(cfg["W"], cfg["H"], cfg["fx"], cfg["fy"], synth.DEFAULT_POSE_A)
    g = synth.s_map(300_000, seed=2)
    o, p, stats = check_forward(g, cam, [0, 0, 0], 0, hip_lib, f"S-map {res}")
    print("S-map", res, "num_rendered", p["num_rendered"], "visible", int((p["radii"] > 0).sum()), stats)
    s = util.read_scratch(hip_lib, p["scratch"], 300_000, p["num_rendered"], cfg["W"], cfg["H"])
    assert np.all(np.diff(s["tile_keys"].astype(np.int64)) >= 0)                       # sortedness by tile
    rec_depth = s["rec"][:, 2][s["point_list"]]
    same_tile = np.diff(s["tile_keys"].astype(np.int64)) == 0
    assert np.all(np.diff(rec_depth)[same_tile] >= 0)                                   # front-to-back inside a tile
    assert int((s["ranges"][:, 1] - s["ranges"][:, 0]).sum()) == p["num_rendered"]      # ranges partition the list
    assert float(p["color"].min()) >= 0.0 and np.all(s["final_T"] <= 1.0)


STAGE_CAMS = {
    # name: (W, H, fx, fy) — BASELINE sizes and the coarse-to-fine training_stage sizes render_3 derives from them
    # (image_width / (2 * stage), same tan(fov/2)) [REF gaussian_renderer/__init__.py:238-242; mp_Mapper.py:207-216]
    "replica": (1200, 680, 600.0, 600.0), "tum": (640, 480, 517.3, 516.5),
    "replica_stage1": (600, 340, 300.0, 300.0), "replica_stage2": (300, 170, 150.0, 150.0),
}


def _stage_cam(name):
    W, H, fx, fy = STAGE_CAMS[name]
    return synth.make_camera(W, H, fx, fy, synth.DEFAULT_POSE_A)


@pytest.mark.parametrize("res", ["replica_stage1", "replica_stage2"])
def test_training_stage_sizes_forward(hip_lib, res):
    """render_3's training_stage 1 / 2 resolutions (600x340, 300x170) with the full S-map: lists bit-exact, images within 1e-5."""
    cam = _stage_cam(res)
    g = synth.s_map(300_000, seed=2)
    o, p, stats = check_forward(g, cam, [0, 0, 0], 0, hip_lib, f"S-map {res}", max_fragile=4e-4)
    print("S-map", res, "num_rendered", p["num_rendered"], stats)


def _write_parity_report(tag, report):
    """One JSON per (resolution, depth rule) under gpurun_out/parity_report/ (tools/collect_profiles.py folds them into
    profiles/rNN_parity_report.json)."""
    import json
    import os
    d = os.environ.get("GSICP_PARITY_REPORT_DIR") or os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out", "parity_report")
    os.makedirs(d, exist_ok=True)
    with open(os.path.join(d, tag + ".json"), "w") as fh:
        json.dump(report, fh, indent=1)


def compare_backward(g, cam, depth_mode, label, report_tag, gaussians_label, seed=7, max_fragile=4e-4, cond_scale=False):
    """All six gradients of (g, cam) under random dL/dcolour, dL/ddepth against the fp32 oracle, ELEMENT-WISE, every visible Gaussian to
        |hip - oracle| <= f * (1e-5 * max|oracle| + 1e-4 * |oracle|) + min(2 |oracle_f32 - oracle_f64|, 1e-4 * max|oracle|)
    (fragile pixels get zero upstream gradients on both sides; see test_full_size_smap_backward).  f = 1, except with `cond_scale` (trained
    maps): f = max(1, (chi / 100)^2), capped so that the term stays <= 2e-2 max|oracle|, where chi is the eigenvalue ratio of the
    Gaussian's screen-space covariance (from the conic the forward reports).  Training stretches some Gaussians into needles a hundred pixels
    long and a fraction of a pixel thin (scales like 1.4 mm x 76 mm x 2 um measured); the conic -> covariance derivative divides by det^2, so
    fp32 carries a relative error ~eps chi^2 there WHATEVER the evaluation order — measured on the trained map: no Gaussian with chi < 100
    (95.6 % of the map) misses the f = 1 bound, the ones that do have chi 150..11 000 (median 1 500), and on them the fp32 ORACLE is as far
    from its own fp64 instance as the HIP result is (RMS ratio 0.75-1.00).  The report counts the Gaussians with f > 1 and the elements that
    need it, and the RMS distances to the fp64 oracle on those Gaussians are asserted (HIP no farther than 2 x the fp32 oracle: the RMS over a few thousand such
    Gaussians is carried by a handful of outliers — measured ratios 0.59-1.35).
    Writes the parity report, returns it."""
    import oracle
    W, H = cam["W"], cam["H"]
    P = g["means3D"].shape[0]
    rng = np.random.default_rng(seed)
    gc = rng.normal(size=(3, H, W)).astype(np.float32)
    gd = rng.normal(size=(H, W)).astype(np.float32)
    bg = [0.0, 0.0, 0.0]
    oracle.raster_set_depth_mode(depth_mode)
    try:
        of = util.oracle_forward(g, cam, bg, 0)
        fragile = of["margin"] <= FRAGILE
        assert fragile.sum() <= max_fragile * W * H, f"{int(fragile.sum())} fragile pixels"
        gc[:, fragile] = 0.0
        gd[fragile] = 0.0
        o = util.oracle_backward(g, cam, bg, gc, gd, 0)
        o64 = util.oracle_backward({k: v.astype(np.float64) for k, v in g.items()}, cam, bg, gc, gd, 0, dtype=np.float64)
    finally:
        oracle.raster_set_depth_mode(0)
    p = run_product(g, cam, bg, 0, grads=(gc, gd), depth_mode=depth_mode)
    import os
    if os.environ.get("GSICP_PARITY_DUMP"):     # the product's gradients for off-line analysis against the oracle (which needs no GPU)
        os.makedirs(os.environ["GSICP_PARITY_DUMP"], exist_ok=True)
        np.savez(os.path.join(os.environ["GSICP_PARITY_DUMP"], report_tag + "_hip_grads.npz"), **{k: v for k, v in p["grads"].items() if v is not None})
    assert np.array_equal(p["radii"], of["radii"])
    n_vis = int((of["radii"] > 0).sum())
    tile_len = (of["ranges"][:, 1].astype(np.int64) - of["ranges"][:, 0])
    gx = (W + 15) // 16

    def list_depth(wg):   # longest tile list the Gaussian sits in
        x, y, r = of["geom"][wg, 0], of["geom"][wg, 1], of["radii"][wg]
        tx0, tx1 = int(max(0, (x - r) // 16)), int(min(gx - 1, (x + r) // 16))
        ty0, ty1 = int(max(0, (y - r) // 16)), int(min((H + 15) // 16 - 1, (y + r) // 16))
        return int(max([tile_len[ty * gx + tx] for ty in range(ty0, ty1 + 1) for tx in range(tx0, tx1 + 1)] or [0]))

    pairs = [("means3D", "dL_dmeans3D"), ("opacities", "dL_dopacity"), ("scales", "dL_dscales"), ("rotations", "dL_drots"),
             ("shs", "dL_dsh"), ("means2D", "dL_dmeans2D")]
    report = dict(scene=gaussians_label, resolution=f"{W}x{H}", depth_mode=depth_mode, gaussians=P, visible=n_vis, duplicates=int(p["num_rendered"]),
                  longest_tile_list=int(tile_len.max()), mean_tile_list=float(tile_len.mean()),
                  fragile_pixels_zeroed=int(fragile.sum()), fragile_fraction=float(fragile.mean()),
                  bound="|hip - oracle32| <= 1e-5 max|g| + 1e-4 |g| + min(2 |oracle32 - oracle64|, 1e-4 max|g|), every visible Gaussian", gradients={})
    failures = []
    fscale = np.ones((P, 1))
    if cond_scale:
        ca, cb, cc = (of["geom"][:, 3 + i].astype(np.float64) for i in range(3))
        tr, det = ca + cc, ca * cc - cb * cb
        disc = np.sqrt(np.maximum(tr * tr - 4.0 * det, 0.0))
        chi = np.where(of["radii"] > 0, (tr + disc) / np.maximum(tr - disc, 1e-300), 1.0)
        fscale = np.maximum(1.0, (chi / 100.0) ** 2)[:, None]
        report.update(conditioning="base bound x max(1, (chi / 100)^2), term capped at 2e-2 max|g|; chi = eigenvalue ratio of the screen-space covariance",
                      gaussians_with_chi_over_100=int((chi > 100.0).sum()), chi_quantiles_50_99_max=[float(v) for v in np.quantile(chi[of["radii"] > 0], [0.5, 0.99, 1.0])])
    for name, key in pairs:
        a = p["grads"][name].reshape(P, -1).astype(np.float64)
        b = o[key].reshape(P, -1).astype(np.float64)
        b64 = o64[key].reshape(P, -1)
        if name == "means2D":
            a, b, b64 = a[:, :2], b[:, :2], b64[:, :2]   # the third column is never written by the rasteriser (upstream leaves it zero)
        mx = np.abs(b).max()
        base = 1e-5 * mx + 1e-4 * np.abs(b)
        bound = np.minimum(base * fscale, np.maximum(base, 2e-2 * mx)) + np.minimum(2.0 * np.abs(b - b64), 1e-4 * mx)
        err = np.abs(a - b)
        ratio = (err / bound).max(1)
        worst = int(np.argmax(ratio))
        extra = {}
        if cond_scale:
            ill = fscale[:, 0] > 1.0
            rms_h, rms_o = (float(np.sqrt(((x_ - b64)[ill] ** 2).mean())) if ill.any() else 0.0 for x_ in (a, b))
            extra = dict(elements_needing_chi_scaling=int((err > base + np.minimum(2.0 * np.abs(b - b64), 1e-4 * mx)).sum()),
                         rms_to_fp64_on_chi_over_100={"hip": rms_h, "oracle32": rms_o})
            if rms_h > 2.0 * rms_o + 1e-6 * mx:
                failures.append(f"{label} grad {name}: on the ill-conditioned Gaussians HIP is farther from the fp64 oracle (rms {rms_h:.3e}) than the fp32 oracle is ({rms_o:.3e})")
        report["gradients"][name] = dict(**extra, 
            max_ratio_to_bound=float(ratio.max()), max_ratio_to_base_bound=float((err / base).max()), worst_gaussian=worst,
            worst_abs_err=float(err[worst].max()), grad_max=float(mx), max_err_over_grad_max=float(err.max() / mx),
            worst_gaussian_longest_list=list_depth(worst), gaussians_needing_conditioning_term=int((err > base).any(1).sum()),
            elements_needing_conditioning_term=int((err > base).sum()), elements=int(err.size),
            oracle32_vs_oracle64_max_over_grad_max=float(np.abs(b - b64).max() / mx))
        if ratio.max() > 1.0:
            failures.append(f"{label} depth_mode {depth_mode} grad {name}: worst Gaussian {worst} err {err[worst]} vs oracle {b[worst]} (max|grad| {mx:.3e})")
        # culled Gaussians get exactly zero
        assert not a[of["radii"] == 0].any()
    report["passed"] = not failures
    _write_parity_report(report_tag, report)
    print(f"backward {label} depth_mode {depth_mode}: {report}")
    assert not failures, failures
    return report


@pytest.mark.parametrize("depth_mode", [0, 1])
@pytest.mark.parametrize("res", ["replica", "tum", "replica_stage1", "replica_stage2"])
def test_full_size_smap_backward(hip_lib, res, depth_mode):
    """R-bwd at BASELINE sizes: S-map P = 300 k at 1200x680 and 640x480 (and the training_stage sizes), random dL/dcolour and
    dL/ddepth, both depth rules, all six gradients compared with the fp32 oracle ELEMENT-WISE, EVERY visible Gaussian to the same bound:
        |hip - oracle| <= 1e-5 * max|oracle| + 1e-4 * |oracle| + min(2 |oracle_f32 - oracle_f64|, 1e-4 * max|oracle|).
    The last term is the fp32 oracle's OWN rounding noise on that element (conic -> covariance -> quaternion is a sum of products
    of dL/dSigma ~ 1e5 with derivatives ~ 1e-4 that cancel to O(10): where fp32 cannot do better the bar is what fp32 delivers);
    the number of elements that need it is reported.
    Fragile pixels (forward decision margin < 1e-5: exp() differs in the last ulps between libm and the GPU, so an alpha or transmittance
    test there may legitimately flip; a few per 10^4 pixels) get dL/dcolour = dL/ddepth = 0 on BOTH sides: every term a pixel contributes
    to any gradient is linear in its dL/dpixel, so such a pixel contributes exactly nothing whichever way its decisions fall, and no
    Gaussian sees a flipped decision — there is no looser class of Gaussians (round 2 held up to 8 % of the visible set to 5e-3)."""
    compare_backward(synth.s_map(300_000, seed=2), _stage_cam(res), depth_mode, f"S-map {res}", f"{res}_depth{depth_mode}", "S-map (synthetic surfels, seed 2)")


# ---------------------------------------------------------------------------------------------- a TRAINED map, several keyframe poses
_TRAINED = {}


def _trained_map():
    """The map the fused loop (tools/slam_demo.py) builds on the 240-frame synthetic sequence: >= 1 600 mapper iterations (6 per frame + 200 after
    the last), keyframe growth, pruning.  Thousands of Adam steps have stretched and overlapped the Gaussians the way a real run does —
    the mapper's actual workload [REF mp_Mapper.py:200-223], unlike the untrained S-map surfels.  Built once per session in a child process."""
    if not _TRAINED:
        import os
        import subprocess
        import sys
        import tempfile
        root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
        path = os.path.join(tempfile.mkdtemp(prefix="gsicp_trained_"), "map.npz")
        subprocess.run([sys.executable, os.path.join(root, "tools", "slam_demo.py"), "240", "--iters", "6", "--post-iters", "200", "--no-asserts",
                        "--save-map", path], check=True, cwd=root, stdout=subprocess.DEVNULL, timeout=900)
        z = np.load(path)
        assert int(z["mapper_iterations"]) >= 1500
        _TRAINED.update(g={k: np.ascontiguousarray(z[k]) for k in ("means3D", "shs", "opacities", "scales", "rotations")}, poses=z["keyframe_poses"],
                        iterations=int(z["mapper_iterations"]))
    return _TRAINED


def _trained_poses(hip_lib):
    """First keyframe, a middle one, the last one, and the keyframe whose view has the LONGEST tile list (found with the product's own ranges)."""
    t = _trained_map()
    if "pick" not in t:
        cfg = synth.REPLICA
        longest = []
        for k, pose in enumerate(t["poses"]):
            cam = synth.make_camera(cfg["W"], cfg["H"], cfg["fx"], cfg["fy"], pose)
            p = run_product(t["g"], cam, [0, 0, 0], 0)
            s = util.read_scratch(hip_lib, p["scratch"], t["g"]["means3D"].shape[0], p["num_rendered"], cfg["W"], cfg["H"])
            longest.append(int((s["ranges"][:, 1].astype(np.int64) - s["ranges"][:, 0]).max()))
        n = len(t["poses"])
        pick = []
        # first, middle, last, then the keyframes with the longest tile lists (longest first) until four DISTINCT views are picked
        for k in [0, n // 2, n - 1] + [int(j) for j in np.argsort(-np.asarray(longest), kind="stable")]:
            if k not in pick and len(pick) < 4:
                pick.append(k)
        t["pick"], t["longest"] = pick, longest
    return t["pick"]


@pytest.mark.parametrize("which", [0, 1, 2, 3])
def test_trained_map_forward_and_backward(hip_lib, which):
    """Rasteriser parity on a TRAINED map from several keyframe poses at 1200x680: lists / ranges / radii / n_contrib
    bit-exact; images within 1e-5 outside the fragile pixels (their fraction is reported, bound 2e-3 here: a trained map's lists are 250 entries long on
    average, up to 835, so a pixel takes far more threshold decisions, and the oracle's margin counts a decision as fragile inside the evaluation noise of ITS operands: 8 eps x the
    magnitude of power's cancelling terms for the alpha tests, 2e-7 x entries blended for the transmittance test — oracle/raster_oracle.cpp) EXCEPT where a pixel blends 300-420 entries: there two fp32 evaluation orders of the transmittance
    product differ by up to 2e-5 (measured: 19 of 816 000 pixels of keyframe 0, all with n_contrib >= 320; the fp32 oracle is as far from
    its own fp64 instance on them), so the bar per pixel is max(1e-5, 1e-7 x its n_contrib) (1.7 ulp per blended entry, capped at 1e-4), at most 1e-4
    of the pixels may exceed 1e-5, and their number is reported; and all six gradients to the bound of test_full_size_smap_backward, its base term scaled by the conditioning of the
    Gaussian's screen-space covariance where that exceeds 100 (compare_backward: 4.4 % of the trained map, needles that fp32 itself cannot resolve)."""
    t = _trained_map()
    pick = _trained_poses(hip_lib)
    if which >= len(pick):
        pytest.skip("the trained map holds fewer than four keyframe poses")
    k = pick[which]
    cfg = synth.REPLICA
    cam = synth.make_camera(cfg["W"], cfg["H"], cfg["fx"], cfg["fy"], t["poses"][k])
    label = f"trained map ({t['iterations']} iterations, {t['g']['means3D'].shape[0]} Gaussians), keyframe {k} of {len(t['poses'])}"
    o, p, stats = check_forward(t["g"], cam, [0, 0, 0], 0, hip_lib, label, max_fragile=2e-3, fp32_noise_cap=1e-4)
    print(label, "num_rendered", p["num_rendered"], "longest tile list", t["longest"][k], stats)
    rep = compare_backward(t["g"], cam, 0, label, f"trained_kf{k}_depth0", label, max_fragile=2e-3, cond_scale=True)
    rep.update(forward=stats, keyframe=int(k), longest_list_is_max_over_keyframes=bool(t["longest"][k] == max(t["longest"])))
    _write_parity_report(f"trained_kf{k}_depth0", rep)
    # The bars are FROZEN: the conditioning allowances above must not absorb a regression.  Measured in round 4:
    # 3-19 pixels of 816 000 beyond 1e-5 per trained view (all blending >= 320 entries), 4.0-4.7 % of the Gaussians with chi > 100.
    assert stats["pixels_over_1e5"] <= 25, f"{label}: {stats['pixels_over_1e5']} pixels beyond 1e-5 (frozen bar: 25)"
    assert rep["gaussians_with_chi_over_100"] <= 0.05 * t["g"]["means3D"].shape[0], \
        f"{label}: {rep['gaussians_with_chi_over_100']} Gaussians take the conditioning-scaled bound (frozen bar: 5 %)"


def _backward_both_variants(hip_lib, g, cam, seed, **kw):
    """Gradients of (g, cam) from the legacy per-Gaussian walk and from round 5's run summation + compacted algebra (twice)."""
    rng = np.random.default_rng(seed)
    gc = rng.normal(size=(3, cam["H"], cam["W"])).astype(np.float32)
    gd = rng.normal(size=(cam["H"], cam["W"])).astype(np.float32)
    out = []
    prev = hip_lib.gsicp_raster_set_legacy_backward(1)
    try:
        out.append(run_product(g, cam, [0, 0, 0], grads=(gc, gd), **kw)["grads"])
        hip_lib.gsicp_raster_set_legacy_backward(0)
        out.append(run_product(g, cam, [0, 0, 0], grads=(gc, gd), **kw)["grads"])
        out.append(run_product(g, cam, [0, 0, 0], grads=(gc, gd), **kw)["grads"])
    finally:
        hip_lib.gsicp_raster_set_legacy_backward(prev)
    return out


@pytest.mark.parametrize("scene", ["small", "smap", "smap_sharded", "huge_splats", "trained"])
def test_run_summation_backward_against_the_legacy_walk(hip_lib, scene):
    """Round 5's per-Gaussian pass (entry_run_sum_kernel + the compacted preprocess_backward_kernel) against the kernel of rounds 3-4
    (GSICP_PREBWD_LEGACY / gsicp_raster_set_legacy_backward), same forward, same upstream gradients: ALL SIX GRADIENTS BIT-IDENTICAL — the run
    summation is the same left fold in slot order (staged through LDS instead of walked in global memory), the algebra the same code over a compacted
    list — and two runs of the new pass bit-identical to each other (the slot allocator places the runs differently from launch to launch; nothing
    may depend on it).  Scenes: 400 random Gaussians on 160x96; the S-map at 1200x680 (82 % culled: the compacted list matters); the same with 2-way
    tile sharding (runs hold only this rank's tiles; visible Gaussians with NO slot exist); 64 splats that cover hundreds of tiles each (runs of
    several 64-record trips: the spill path); the trained map (80-97 % visible, runs up to 340 records)."""
    cfg = synth.REPLICA
    kw = {}
    if scene == "small":
        g, cam = synth.random_gaussians(400, seed=3), synth.make_camera(160, 96, 120.0, 120.0)
    elif scene in ("smap", "smap_sharded"):
        g, cam = synth.s_map(300_000, seed=2), synth.make_camera(cfg["W"], cfg["H"], cfg["fx"], cfg["fy"], synth.DEFAULT_POSE_A)
        if scene == "smap_sharded":
            kw = dict(tile_mod=2, tile_rem=1)
    elif scene == "huge_splats":
        g = synth.random_gaussians(64, seed=11)
        g["scales"] = (g["scales"] * 0 + np.float32(1.0)) * np.random.default_rng(1).uniform(0.3, 1.2, g["scales"].shape).astype(np.float32)
        g["opacities"] = np.full_like(g["opacities"], 0.05)
        cam = synth.make_camera(640, 480, 500.0, 500.0)
    else:
        t = _trained_map()
        g, cam = t["g"], synth.make_camera(cfg["W"], cfg["H"], cfg["fx"], cfg["fy"], t["poses"][len(t["poses"]) // 2])
    legacy, new1, new2 = _backward_both_variants(hip_lib, g, cam, seed=5, **kw)
    worst = {}
    for name in legacy:
        if legacy[name] is None:
            assert new1[name] is None
            continue
        assert np.array_equal(new1[name], new2[name]), f"{scene} {name}: two runs of the new pass differ"
        assert np.isfinite(new1[name]).all()
        a, b = legacy[name].astype(np.float64), new1[name].astype(np.float64)
        mx = np.abs(a).max()
        err = np.abs(a - b)
        worst[name] = float(err.max() / mx) if mx > 0 else 0.0
        assert np.array_equal(legacy[name], new1[name]), \
            f"{scene} {name}: new pass differs from the legacy walk on {int((legacy[name] != new1[name]).sum())} elements (max |diff| / max|g| = {worst[name]:.3e})"
    print(f"run-summation backward vs legacy walk, {scene}: bit-identical ({list(worst)})")


def test_alpha_normalised_depth_mode_forward_and_backward(hip_lib):
    """depth_mode = 1 (SURVEY 8a unknown, exposed as an option): D = sum z alpha T / (1 - T_final).  Forward image and all six
    gradients against the oracle's instance of the same rule; colours are untouched by the option."""
    import oracle
    cam = synth.make_camera(160, 96, 120.0, 120.0)
    g = synth.random_gaussians(400, seed=41)
    rng = np.random.default_rng(3)
    gc = rng.normal(size=(3, 96, 160)).astype(np.float32)
    gd = rng.normal(size=(96, 160)).astype(np.float32)
    bg = [0.2, 0.1, 0.3]
    base = run_product(g, cam, bg, 0, grads=(gc, gd))
    oracle.raster_set_depth_mode(1)
    try:
        of = util.oracle_forward(g, cam, bg, 0)
        o = util.oracle_backward(g, cam, bg, gc, gd, 0)
        o64 = util.oracle_backward({k: v.astype(np.float64) for k, v in g.items()}, cam, bg, gc, gd, 0, dtype=np.float64)
    finally:
        oracle.raster_set_depth_mode(0)
    p = run_product(g, cam, bg, 0, grads=(gc, gd), depth_mode=1)
    ok = of["margin"] > FRAGILE
    assert np.array_equal(p["color"], base["color"])
    assert np.abs(p["depth"] - of["depth"])[ok].max() <= 1e-5 * max(1.0, float(of["depth"].max()))
    assert np.abs(p["depth"] - base["depth"]).max() > 0.1          # the option changes the image
    covered = of["final_T"] < 0.5
    assert np.all(p["depth"][covered] >= base["depth"][covered] - 1e-6)   # dividing by accumulated alpha <= 1 can only raise it
    for name, key in [("means3D", "dL_dmeans3D"), ("opacities", "dL_dopacity"), ("scales", "dL_dscales"), ("rotations", "dL_drots"),
                      ("shs", "dL_dsh"), ("means2D", "dL_dmeans2D")]:
        a, b, b64 = p["grads"][name].reshape(-1).astype(np.float64), o[key].reshape(-1).astype(np.float64), o64[key].reshape(-1)
        mx = np.abs(b).max()
        err = np.abs(a - b)
        assert (err <= 2e-4 * mx + 1e-4 * np.abs(b) + 2 * np.abs(b - b64)).all(), f"{name}: {err.max() / mx:.3e} of max"
        assert np.abs(a - base["grads"][name].reshape(-1)).max() > 1e-3 * mx or name == "shs"   # depth gradients differ from mode 0


def test_long_tile_lists_use_the_fallback_sort(hip_lib):
    """> 1024 entries in one tile (here > 4096): the list is sorted in 1024-entry chunks in LDS and the chunks are merged by rank through
    global memory; lists must still be exact."""
    cam = synth.make_camera(48, 32, 40.0, 40.0)
    g = synth.random_gaussians(5000, seed=12, spread=0.2, zmin=2.0, zmax=6.0)
    g["scales"] = (g["scales"] * 6.0).astype(np.float32)          # every Gaussian covers the whole 3x2-tile image
    g["opacities"] = (g["opacities"] * 0.02).astype(np.float32)   # keep transmittance alive so lists are traversed deep
    # thousands of threshold decisions per pixel: many more pixels sit near a threshold, so only the robust ones are compared
    o, p, _ = check_forward(g, cam, [0, 0, 0], 0, hip_lib, "long lists", tol=5e-5, max_fragile=0.05)
    assert (o["ranges"][:, 1].astype(np.int64) - o["ranges"][:, 0]).max() > 4096


def test_very_long_tile_lists_merge_several_sorted_chunks(hip_lib):
    """20 000 entries in every tile: twenty 1024-entry chunks sorted in LDS and merged by rank.  The lists (bit-exact against the oracle's
    stable 64-bit sort) are the point here; images only on robust pixels."""
    cam = synth.make_camera(32, 32, 30.0, 30.0)
    g = synth.random_gaussians(20000, seed=14, spread=0.15, zmin=2.0, zmax=6.0)
    g["scales"] = (g["scales"] * 8.0).astype(np.float32)
    g["opacities"] = (g["opacities"] * 0.004).astype(np.float32)
    g["means3D"][:4000, 2] = g["means3D"][4000:8000, 2]          # exact depth ties across chunks: the Gaussian id must break them
    o, p, _ = check_forward(g, cam, [0, 0, 0], 0, hip_lib, "very long lists", tol=2e-4, max_fragile=0.2)
    assert (o["ranges"][:, 1].astype(np.int64) - o["ranges"][:, 0]).min() > 4 * 4096


def test_many_tiles_uhd(hip_lib):
    """3840x2160 = 32 400 tiles: the tile multi-split needs 127 KB of dynamic LDS per workgroup."""
    cam = synth.make_camera(3840, 2160, 2800.0, 2800.0)
    g = synth.random_gaussians(3000, seed=13, spread=1.5, zmin=2.0, zmax=6.0)
    # Integer outputs stay bit-exact.  Colour tolerance is relaxed: with ~100 px splats the three terms of `power` reach 1e3 and
    # cancel to O(1), so fp32 rounding (FMA-contracted on the GPU, plain mul/add in the oracle) shows up at ~1e-4 in alpha.
    check_forward(g, cam, [0.1, 0.1, 0.1], 0, hip_lib, "uhd", tol=1e-3)


def test_mark_visible():
    import torch
    from diff_gaussian_rasterization import GaussianRasterizer
    cam = synth.make_camera(64, 48, 60.0, 60.0)
    rs = util.make_settings(cam, [0, 0, 0])
    pos = torch.tensor([[0, 0, 1.0], [0, 0, 0.1], [0, 0, -3.0], [5, 5, 0.21]], device="cuda")
    assert GaussianRasterizer(rs).markVisible(pos).cpu().tolist() == [True, False, False, True]


def test_sharded_wrapper_collectives_on_rccl_world1(hip_lib):
    """The N-GPU path's collectives and autograd plumbing on real device tensors: backend "nccl" (= RCCL) with a single rank
    (gpurun exposes one GPU), forced through the all-reduce branch.  Must equal the plain rasteriser bit-for-bit."""
    import os
    import torch
    import torch.distributed as dist
    from diff_gaussian_rasterization import GaussianRasterizer
    from gs_icp_slam_amd.sharded import ShardedGaussianRasterizer
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29581")
    if not dist.is_initialized():
        dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    try:
        cam = synth.make_camera(176, 96, 130.0, 130.0)
        g = synth.random_gaussians(500, seed=21)
        rs = util.make_settings(cam, [0.1, 0.2, 0.3])
        outs = []
        for mk in (lambda: GaussianRasterizer(rs), lambda: ShardedGaussianRasterizer(rs, force_collectives=True)):
            t = util.torch_inputs(g, requires_grad=True)
            m2 = torch.zeros_like(t["means3D"], requires_grad=True)
            d, c, r, u = mk()(means3D=t["means3D"], means2D=m2, shs=t["shs"], opacities=t["opacities"], scales=t["scales"], rotations=t["rotations"])
            (c.square().sum() + d.sum()).backward()
            outs.append((c.detach(), d.detach(), {k: v.grad.clone() for k, v in t.items()}, m2.grad.clone(), u))
        assert torch.equal(outs[0][0], outs[1][0]) and torch.equal(outs[0][1], outs[1][1]) and torch.equal(outs[0][4], outs[1][4])
        for k in outs[0][2]:
            assert torch.equal(outs[0][2][k], outs[1][2][k]), k
        assert torch.equal(outs[0][3], outs[1][3])
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("n", [3, 40, 64, 65, 100, 128, 129, 200, 256, 257, 400, 512, 513, 900, 1024, 1025, 2500])
def test_register_tile_sort_equals_the_lds_network_and_the_oracle(hip_lib, n):
    """Round 6: the per-tile sort keeps every compare-exchange with partner distance < 128 in REGISTERS (lane exchanges; LDS only for the cross-block steps
    of lists above 128 entries).  One list length per size class of the network — 64, 128, 256, 512, 1024 padded entries, their edges, and the chunked path
    beyond 1024 — with exact depth TIES (the Gaussian id must break them): block bits, Gaussian ids, tile keys and images are the same
    bits as the all-LDS network of rounds 2-5 (gsicp_raster_set_tile_sort_lds), and the lists equal the oracle's stable 64-bit sort."""
    cam = synth.make_camera(48, 32, 40.0, 40.0)                    # 3 x 2 tiles
    g = synth.random_gaussians(n, seed=100 + n, spread=0.2, zmin=2.0, zmax=6.0)
    g["scales"] = (g["scales"] * 6.0).astype(np.float32)          # every Gaussian covers the whole image: every tile's list has n entries
    g["opacities"] = (g["opacities"] * 0.02 + 0.01).astype(np.float32)
    if n >= 8:
        g["means3D"][: n // 4, 2] = g["means3D"][n // 4: 2 * (n // 4), 2]      # a quarter of the depths tied pairwise
    out = {}
    prev = hip_lib.gsicp_raster_set_tile_sort_lds(0)
    try:
        for lds in (0, 1):
            hip_lib.gsicp_raster_set_tile_sort_lds(lds)
            p = run_product(g, cam, [0, 0, 0])
            geom, binning, img = p["scratch"]
            lay = (__import__("ctypes").c_size_t * 12)()
            hip_lib.gsicp_raster_layout(n, p["num_rendered"], 48, 32, lay)
            b8 = binning.cpu().numpy()
            R = p["num_rendered"]
            raw = b8[lay[4]: lay[4] + 4 * R].view(np.uint32).copy()
            s = util.read_scratch(hip_lib, p["scratch"], n, R, 48, 32)
            # (the raw words carry EMISSION SLOTS, whose allocation order across the preprocess workgroups differs from run to run above 256 Gaussians: the block
            # bits on top of them and the Gaussian ids behind them are what must agree)
            out[lds] = (raw >> np.uint32(28), s["point_list"].copy(), s["tile_keys"].copy(), s["ranges"].copy(), p["color"].copy())
    finally:
        hip_lib.gsicp_raster_set_tile_sort_lds(prev)
    for a, b in zip(out[0], out[1]):
        assert np.array_equal(a, b)
    longest = int((out[0][3][:, 1].astype(np.int64) - out[0][3][:, 0]).max())
    assert longest >= (n + 1) // 2, f"the scene must put most of its {n} Gaussians into one tile list (longest: {longest})"
    o = util.oracle_forward(g, cam, [0, 0, 0], 0)
    assert np.array_equal(out[0][1], o["point_list"]) and np.array_equal(out[0][3], o["ranges"])
