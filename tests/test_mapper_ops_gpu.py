"""GPU parity for the §8f "next" operators: fused mapping loss (vs golden vectors produced by the REFERENCE'S OWN
utils/loss_utils.py, and vs the pinned oracle at full size) and multi-tensor Adam (vs torch.optim.Adam, the reference's
optimiser, on CPU)."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(__file__), "golden", "mapper_loss.npz")


def _run(c):
    from gs_icp_slam_amd.loss import mapper_loss_parts
    image = torch.tensor(c["image"], device="cuda", requires_grad=True)
    depth = torch.tensor(c["depth"], device="cuda", requires_grad=True)
    loss, parts = mapper_loss_parts(image, depth, torch.tensor(c["gt_image"], device="cuda"), torch.tensor(c["gt_depth"], device="cuda"))
    loss.backward()
    return loss.item(), parts.cpu().numpy(), image.grad.cpu().numpy(), depth.grad.cpu().numpy()


@pytest.mark.parametrize("name", ["a", "b"])
def test_loss_matches_reference_golden(name):
    z = np.load(GOLD)
    c = {k[2:]: z[k] for k in z.files if k.startswith(name + "_")}
    loss, parts, gi, gd = _run(c)
    assert abs(loss - c["loss"]) < 2e-6 and abs(parts[1] - c["l1"]) < 2e-6 and abs(parts[2] - c["ssim"]) < 5e-6 and abs(parts[3] - c["l1_d"]) < 1e-7
    assert np.abs(gi - c["grad_image"]).max() <= 2e-4 * np.abs(c["grad_image"]).max()
    assert np.array_equal(gi == 0, c["grad_image"] == 0)            # masked elements carry exactly no gradient
    np.testing.assert_allclose(gd, c["grad_depth"], atol=1e-12, rtol=1e-5)


@pytest.mark.parametrize("res", [(1200, 680), (640, 480)])
def test_loss_full_size_vs_oracle(res):
    from oracle import loss_oracle
    W, H = res
    rng = np.random.default_rng(W)
    yy, xx = np.mgrid[0:H, 0:W]
    gt = np.stack([0.5 + 0.4 * np.sin(xx / (9.0 + c) + yy / 17.0) for c in range(3)]).astype(np.float32)
    img = np.clip(gt + rng.normal(0, 0.05, gt.shape), 0, 1).astype(np.float32)
    gtd = (2.0 + np.sin(xx / 31.0)).astype(np.float32)[None]
    gtd[:, H // 2: H // 2 + 20, W // 3: W // 2] = 0
    dep = (gtd + rng.normal(0, 0.03, gtd.shape)).astype(np.float32)
    c = dict(image=img, depth=dep, gt_image=gt, gt_depth=gtd)
    loss, parts, gi, gd = _run(c)
    # float64 oracle at this size: a float32 mean over 2.4 M elements (what torch does on CPU) is itself only good to ~1e-5
    ti, td = torch.tensor(img, dtype=torch.float64, requires_grad=True), torch.tensor(dep, dtype=torch.float64, requires_grad=True)
    lo, l1, ss, ld = loss_oracle.mapper_loss(ti, td, torch.tensor(gt, dtype=torch.float64), torch.tensor(gtd, dtype=torch.float64))
    lo.backward()
    assert abs(loss - lo.item()) < 2e-6 and abs(parts[2] - ss.item()) < 3e-6 and abs(parts[1] - l1.item()) < 1e-6
    assert np.abs(gi - ti.grad.numpy()).max() <= 2e-4 * np.abs(ti.grad.numpy()).max()
    robust = np.abs(dep / np.float32(10) - gtd / np.float32(10)) > 1e-7   # sign(0) in float32 vs a tiny residue in float64
    np.testing.assert_allclose(gd[robust], td.grad.numpy()[robust], atol=1e-12, rtol=1e-5)


def test_loss_scales_with_upstream_gradient_and_rejects_cpu():
    from gs_icp_slam_amd.loss import mapper_loss
    z = np.load(GOLD)
    c = {k[2:]: z[k] for k in z.files if k.startswith("b_")}
    image = torch.tensor(c["image"], device="cuda", requires_grad=True)
    depth = torch.tensor(c["depth"], device="cuda", requires_grad=True)
    (3.0 * mapper_loss(image, depth, torch.tensor(c["gt_image"], device="cuda"), torch.tensor(c["gt_depth"], device="cuda"))).backward()
    np.testing.assert_allclose(image.grad.cpu().numpy(), 3.0 * c["grad_image"], atol=6e-4 * np.abs(c["grad_image"]).max())
    with pytest.raises(RuntimeError):
        mapper_loss(torch.zeros(3, 8, 8), torch.zeros(1, 8, 8), torch.zeros(3, 8, 8), torch.zeros(1, 8, 8))


def test_fused_adam_matches_torch_adam():
    """Six parameter groups with the reference's learning rates and eps [REF arguments/__init__.py:141-148; scene/gaussian_model.py:231]."""
    from gs_icp_slam_amd.optim import FusedAdam
    rng = np.random.default_rng(0)
    shapes = [(5000, 3), (5000, 1, 3), (5000, 0, 3), (5000, 1), (5000, 3), (5000, 4)]
    lrs = [1.6e-6 * 2.5, 2.5e-3, 1.25e-4, 0.05, 5e-3, 1e-3]
    init = [rng.normal(size=s).astype(np.float32) for s in shapes]
    ref_p = [torch.tensor(a, requires_grad=True) for a in init]
    gpu_p = [torch.tensor(a, device="cuda", requires_grad=True) for a in init]
    ref = torch.optim.Adam([{"params": [p], "lr": lr} for p, lr in zip(ref_p, lrs)], lr=0.0, eps=1e-15)
    fused = FusedAdam([{"params": [p], "lr": lr} for p, lr in zip(gpu_p, lrs)], lr=0.0, eps=1e-15)
    for it in range(25):
        for p, q in zip(ref_p, gpu_p):
            g = (rng.normal(size=tuple(p.shape)) * (10.0 ** rng.integers(-6, 1))).astype(np.float32)
            p.grad = torch.tensor(g)
            q.grad = torch.tensor(g, device="cuda")
        ref.step()
        fused.step()
    for p, q in zip(ref_p, gpu_p):
        if p.numel():
            np.testing.assert_allclose(q.detach().cpu().numpy(), p.detach().numpy(), rtol=2e-5, atol=1e-7)
    st = fused.state[gpu_p[0]]
    assert set(st) >= {"step", "exp_avg", "exp_avg_sq"} and int(st["step"]) == 25
    np.testing.assert_allclose(st["exp_avg"].cpu().numpy(), ref.state[ref_p[0]]["exp_avg"].numpy(), rtol=1e-5, atol=1e-7)


def test_fused_activations_match_torch_ops():
    from gs_icp_slam_amd.activations import activate
    torch.manual_seed(3)
    P = 5000
    raw = [torch.randn(P, 1, device="cuda") * 3, torch.randn(P, 3, device="cuda") - 4, torch.randn(P, 4, device="cuda")]
    raw[2][7] = 0.0                                    # zero quaternion: normalize clamps the norm at 1e-12
    a = [r.clone().requires_grad_(True) for r in raw]
    b = [r.clone().requires_grad_(True) for r in raw]
    oa, sa, qa = activate(*a)
    ob, sb, qb = torch.sigmoid(b[0]), torch.exp(b[1]), torch.nn.functional.normalize(b[2])
    torch.testing.assert_close(oa, ob, rtol=2e-6, atol=1e-7)
    torch.testing.assert_close(sa, sb, rtol=2e-6, atol=0)
    torch.testing.assert_close(qa, qb, rtol=2e-6, atol=1e-7)
    w = [torch.randn_like(oa), torch.randn_like(sa), torch.randn_like(qa)]
    (oa * w[0]).sum().add((sa * w[1]).sum()).add((qa * w[2]).sum()).backward()
    (ob * w[0]).sum().add((sb * w[1]).sum()).add((qb * w[2]).sum()).backward()
    for x, y in zip(a, b):
        torch.testing.assert_close(x.grad, y.grad, rtol=2e-5, atol=1e-6)
    # partial gradients: only the opacity branch is used downstream
    c = [r.clone().requires_grad_(True) for r in raw]
    oc, _, _ = activate(*c)
    oc.sum().backward()
    torch.testing.assert_close(c[0].grad, (ob * (1 - ob)).detach(), rtol=2e-5, atol=1e-7)
    assert torch.all(c[1].grad == 0) and torch.all(c[2].grad == 0)
    with pytest.raises(RuntimeError):
        activate(*[r.cpu() for r in raw])


def test_raw_parameter_rasteriser_equals_activations_plus_rasteriser():
    """GaussianRasterizationSettings(raw_params=True): sigmoid / exp / normalize and their chain rule inside the preprocess kernels must give
    what the separate activation operator followed by the rasteriser gives — same lists, images to rounding, gradients w.r.t. the RAW
    parameters to 1e-5 of their maximum."""
    from diff_gaussian_rasterization import GaussianRasterizer
    from gs_icp_slam_amd import synth
    from gs_icp_slam_amd.activations import activate
    from tests.util import make_settings
    P, W, H = 20000, 320, 200
    cfg = synth.REPLICA
    cam = synth.make_camera(W, H, cfg["fx"] * W / cfg["W"], cfg["fy"] * H / cfg["H"], synth.DEFAULT_POSE_A)
    g = synth.s_map(P, seed=9)
    raw = {"means3D": torch.from_numpy(g["means3D"]), "scales": torch.log(torch.from_numpy(g["scales"])),
           "rotations": torch.from_numpy(g["rotations"]) * 1.7,          # un-normalised on purpose
           "opacities": torch.logit(torch.from_numpy(g["opacities"]).clamp(1e-4, 1 - 1e-4)), "shs": torch.from_numpy(g["shs"])}
    gen = torch.Generator(device="cuda").manual_seed(0)
    wc, wd = torch.rand((3, H, W), device="cuda", generator=gen), torch.rand((1, H, W), device="cuda", generator=gen)
    outs = []
    for fused in (False, True):
        t = {k: v.clone().cuda().requires_grad_(True) for k, v in raw.items()}
        rs = make_settings(cam, [0.0, 0.0, 0.0])._replace(capacity=2_000_000, raw_params=fused)
        m2 = torch.zeros_like(t["means3D"], requires_grad=True)
        if fused:
            o, s_, q = t["opacities"], t["scales"], t["rotations"]
        else:
            o, s_, q = activate(t["opacities"], t["scales"], t["rotations"])
        depth, color, radii, used = GaussianRasterizer(rs)(means3D=t["means3D"], means2D=m2, shs=t["shs"], opacities=o, scales=s_, rotations=q)
        ((color * wc).sum() + (depth * wd).sum()).backward()
        outs.append((depth.detach(), color.detach(), radii, {k: v.grad.clone() for k, v in t.items()}, m2.grad.clone()))
    (d0, c0, r0, g0, m0), (d1, c1, r1, g1, m1) = outs
    assert (r0 != r1).sum() <= 2                      # a 1-ulp difference in a scale may move a 3-sigma radius across an integer
    assert float((c0 - c1).abs().max()) < 2e-5 and float((d0 - d1).abs().max()) < 1e-4
    for k in g0:
        mx = float(g0[k].abs().max())
        assert float((g0[k] - g1[k]).abs().max()) <= 2e-5 * mx + 1e-12, (k, float((g0[k] - g1[k]).abs().max()), mx)
    assert float((m0 - m1).abs().max()) <= 2e-5 * float(m0.abs().max())


@pytest.mark.parametrize("res", [(1200, 680), (640, 480), (75, 53)])
def test_loss_pass2_with_hoisted_pixel_loads_equals_the_late_loads_bit_for_bit(res):
    """Round 6: loss pass 2 requests the mask / target / image values of its own pixels together with its staging loads instead of after
    its convolutions (gsicp_mapper_loss_set_hoist): the four loss values and both gradient images are the SAME BITS as the late-load kernel's, with and
    without gradients being asked for, at both benchmark resolutions and on a ragged image (partial tiles on both edges, masked holes)."""
    from gs_icp_slam_amd import _lib
    from gs_icp_slam_amd.loss import mapper_loss_and_grads, mapper_loss_parts
    lib = _lib.load()
    W, H = res
    rng = np.random.default_rng(W + H)
    yy, xx = np.mgrid[0:H, 0:W]
    gt = np.stack([0.5 + 0.4 * np.sin(xx / (9.0 + c) + yy / 17.0) for c in range(3)]).astype(np.float32)
    img = np.clip(gt + rng.normal(0, 0.05, gt.shape), 0, 1).astype(np.float32)
    gtd = (2.0 + np.sin(xx / 31.0)).astype(np.float32)[None]
    gtd[:, H // 2: H // 2 + 9, W // 3: W // 2] = 0
    dep = (gtd + rng.normal(0, 0.03, gtd.shape)).astype(np.float32)
    t = [torch.tensor(a, device="cuda") for a in (img, dep, gt, gtd)]
    out = {}
    prev = lib.gsicp_mapper_loss_set_hoist(1)
    try:
        for form in (1, 0):
            lib.gsicp_mapper_loss_set_hoist(form)
            parts, g_img, g_dep = mapper_loss_and_grads(*t)
            with torch.no_grad():
                value_only = mapper_loss_parts(*t)[1]
            out[form] = (parts.clone(), g_img.clone(), g_dep.clone(), value_only.clone())
    finally:
        lib.gsicp_mapper_loss_set_hoist(prev)
    for a, b in zip(out[1], out[0]):
        assert torch.equal(a, b)
    assert torch.equal(out[1][0], out[1][3])          # the value-only call (reduce kernel) gives the same four values
    assert float(out[1][0][0]) > 0 and bool(out[1][1].abs().sum() > 0) and bool(out[1][2].abs().sum() > 0)
