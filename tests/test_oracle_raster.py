"""CPU: the rasteriser oracle against finite differences, against itself in fp64, against the committed golden vectors,
and against properties that need no oracle.  (No reference golden vectors exist — parity unpinned)"""
import os

import numpy as np
import pytest

import oracle
from gs_icp_slam_amd import synth
from tests import util

GOLD = os.path.join(os.path.dirname(__file__), "golden")


def _loss(g, cam, bg, wc, wd, deg, **over):
    a = dict(g)
    a.update(over)
    o = util.oracle_forward(a, cam, bg, deg, dtype=np.float64)
    return float((o["color"] * wc).sum() + (o["depth"] * wd).sum())


@pytest.mark.parametrize("deg", [0, 2])
def test_backward_matches_finite_differences_fp64(deg):
    W, H = 48, 32
    cam = synth.make_camera(W, H, 45.0, 45.0)
    g = {k: v.astype(np.float64) for k, v in synth.random_gaussians(25, seed=3 + deg, sh_degree=deg, spread=0.8, zmin=1.5, zmax=4).items()}
    rng = np.random.default_rng(5)
    wc, wd = rng.normal(size=(3, H, W)), rng.normal(size=(H, W))
    bg = [0.1, 0.2, 0.3]
    gr = util.oracle_backward(g, cam, bg, wc, wd, deg, dtype=np.float64)
    for name, key in (("means3D", "dL_dmeans3D"), ("opacities", "dL_dopacity"), ("scales", "dL_dscales"), ("rotations", "dL_drots"),
                      ("shs", "dL_dsh")):
        base, an = g[name], gr[key].reshape(g[name].shape)
        for idx in list(np.ndindex(*base.shape))[:12]:
            bp, bm = base.copy(), base.copy()
            bp[idx] += 1e-6
            bm[idx] -= 1e-6
            num = (_loss(g, cam, bg, wc, wd, deg, **{name: bp}) - _loss(g, cam, bg, wc, wd, deg, **{name: bm})) / 2e-6
            assert abs(num - an[idx]) <= 1e-5 * max(1.0, np.abs(an).max()), (name, idx, num, an[idx])


def test_f32_and_f64_instantiations_agree():
    cam = synth.make_camera(96, 64, 80.0, 80.0)
    g = synth.random_gaussians(150, seed=9)
    a = util.oracle_forward(g, cam, [0, 0, 0], 0)
    b = util.oracle_forward({k: v.astype(np.float64) for k, v in g.items()}, cam, [0, 0, 0], 0, dtype=np.float64)
    ok = a["margin"] > 1e-4
    assert ok.mean() > 0.99
    assert np.abs(a["color"] - b["color"]).max(0)[ok].max() < 2e-5
    assert np.abs(a["radii"] - b["radii"]).max() <= 1


@pytest.mark.parametrize("name", ["raster_deg0", "raster_deg3"])
def test_oracle_reproduces_golden(name):
    z = np.load(os.path.join(GOLD, name + ".npz"))
    g = {k[3:]: z[k] for k in z.files if k.startswith("in_")}
    cam = dict(viewmatrix=z["viewmatrix"], projmatrix=z["projmatrix"], campos=z["campos"], tanfovx=float(z["tanfovx"]),
               tanfovy=float(z["tanfovy"]), W=int(z["W"]), H=int(z["H"]))
    deg = int(z["deg"])
    f = util.oracle_forward(g, cam, z["bg"], deg)
    assert np.array_equal(f["radii"], z["fwd_radii"]) and np.array_equal(f["point_list"], z["fwd_point_list"])
    assert np.array_equal(f["ranges"], z["fwd_ranges"]) and np.array_equal(f["n_contrib"], z["fwd_n_contrib"])
    np.testing.assert_allclose(f["color"], z["fwd_color"], atol=2e-6)
    np.testing.assert_allclose(f["depth"], z["fwd_depth"], atol=1e-5)
    b = util.oracle_backward(g, cam, z["bg"], z["grad_color"], z["grad_depth"], deg)
    for k in ("dL_dmeans3D", "dL_dopacity", "dL_dscales", "dL_drots", "dL_dsh", "dL_dmeans2D"):
        np.testing.assert_allclose(b[k], z["bwd_" + k], rtol=1e-4, atol=1e-5 * np.abs(z["bwd_" + k]).max())


def test_binning_properties_and_edge_cases():
    cam = synth.make_camera(203, 77, 150.0, 140.0)
    g = synth.random_gaussians(600, seed=3, spread=3.0, zmin=-2.0, zmax=5.0)
    o = util.oracle_forward(g, cam, [0, 0, 0], 0)
    keys = o["keys"]
    assert np.all(np.diff(keys.astype(np.int64)) >= 0)                                   # (tile, depth) sorted
    assert int((o["ranges"][:, 1].astype(np.int64) - o["ranges"][:, 0]).sum()) == o["num_rendered"]
    assert np.all(o["radii"][g["means3D"][:, 2] <= 0.2] == 0)                               # near-plane cull
    assert o["final_T"].max() <= 1.0 and o["color"].min() >= 0.0
    # empty input
    e = util.oracle_forward({k: v[:0] for k, v in g.items()}, cam, [0.2, 0.3, 0.4], 0)
    assert e["num_rendered"] == 0 and np.allclose(e["color"][:, 0, 0], [0.2, 0.3, 0.4]) and e["depth"].max() == 0


def test_quaternion_order_is_xyzw():
    """reference utils/general_utils.py:89-99 reads q = (x, y, z, w): a 90 deg rotation about x of a thin disc must turn
    its footprint from a filled blob into a horizontal streak."""
    cam = synth.make_camera(128, 128, 100.0, 100.0)
    c = np.float32(np.sqrt(0.5))
    base = dict(means3D=np.array([[0, 0, 2.0]], np.float32), scales=np.array([[0.3, 0.3, 0.003]], np.float32),
                opacities=np.array([[0.9]], np.float32), shs=np.array([[[1.0, 1.0, 1.0]]], np.float32))
    flat = util.oracle_forward(dict(base, rotations=np.array([[0, 0, 0, 1]], np.float32)), cam, [0, 0, 0], 0)["color"][0]
    edge = util.oracle_forward(dict(base, rotations=np.array([[c, 0, 0, c]], np.float32)), cam, [0, 0, 0], 0)["color"][0]
    ys, xs = np.nonzero(flat > 0.05)
    assert abs((xs.max() - xs.min()) - (ys.max() - ys.min())) <= 2
    ys, xs = np.nonzero(edge > 0.05)
    assert (xs.max() - xs.min()) > 4 * (ys.max() - ys.min())


def test_alpha_normalised_depth_rule_backward_matches_finite_differences():
    """The oracle's instance of the optional depth rule D = sum z alpha T / (1 - T_final) (depth_mode 1): analytic gradients in fp64
    against central differences."""
    import oracle
    from gs_icp_slam_amd import synth
    from tests import util
    cam = synth.make_camera(48, 32, 40.0, 40.0)
    g = {k: v.astype(np.float64) for k, v in synth.random_gaussians(12, seed=3, spread=0.5, zmin=1.5, zmax=3.0).items()}
    rng = np.random.default_rng(0)
    gc, gd = rng.normal(size=(3, 32, 48)), rng.normal(size=(32, 48))
    bg = [0.1, 0.2, 0.3]
    oracle.raster_set_depth_mode(1)
    try:
        def loss(gg):
            o = util.oracle_forward(gg, cam, bg, 0, dtype=np.float64)
            return (o["color"] * gc).sum() + (o["depth"] * gd).sum()
        b = util.oracle_backward(g, cam, bg, gc, gd, 0, dtype=np.float64)
        plain = util.oracle_forward(g, cam, bg, 0, dtype=np.float64)
        for name, key, picks in (("opacities", "dL_dopacity", [(0, 0), (3, 0), (7, 0)]), ("means3D", "dL_dmeans3D", [(0, 0), (3, 2), (7, 1)]),
                                 ("scales", "dL_dscales", [(1, 0), (5, 2)]), ("rotations", "dL_drots", [(2, 1), (9, 3)])):
            for idx in picks:
                h = 1e-6
                gp, gm = {k: v.copy() for k, v in g.items()}, {k: v.copy() for k, v in g.items()}
                gp[name][idx] += h
                gm[name][idx] -= h
                fd = (loss(gp) - loss(gm)) / (2 * h)
                an = b[key].reshape(g[name].shape)[idx]
                assert abs(fd - an) <= 1e-5 * (abs(an) + 1e-3), (name, idx, fd, an)
    finally:
        oracle.raster_set_depth_mode(0)
    zero = util.oracle_forward(g, cam, bg, 0, dtype=np.float64)
    cover = zero["final_T"] < 1.0
    np.testing.assert_allclose(plain["depth"][cover], (zero["depth"] / (1.0 - zero["final_T"] + (~cover)))[cover], rtol=1e-12)
