"""CPU: the mapping-loss oracle is PINNED — against the reference's own utils/loss_utils.py (imported when /root/reference is
present) and against golden vectors produced by running the reference's code (tests/golden/make_golden_loss.py)."""
import os
import sys

import numpy as np
import pytest
import torch

from oracle import loss_oracle

GOLD = os.path.join(os.path.dirname(__file__), "golden", "mapper_loss.npz")


def _case(z, n):
    return {k[len(n) + 1:]: z[k] for k in z.files if k.startswith(n + "_")}


@pytest.mark.parametrize("name", ["a", "b"])
def test_oracle_matches_reference_golden(name):
    c = _case(np.load(GOLD), name)
    image = torch.tensor(c["image"], requires_grad=True)
    depth = torch.tensor(c["depth"], requires_grad=True)
    loss, l1, ss, ld = loss_oracle.mapper_loss(image, depth, torch.tensor(c["gt_image"]), torch.tensor(c["gt_depth"]))
    loss.backward()
    assert abs(loss.item() - c["loss"]) < 1e-6 and abs(l1.item() - c["l1"]) < 1e-6 and abs(ss.item() - c["ssim"]) < 1e-6
    assert abs(ld.item() - c["l1_d"]) < 1e-7
    np.testing.assert_allclose(image.grad.numpy(), c["grad_image"], atol=1e-9, rtol=1e-5)
    np.testing.assert_allclose(depth.grad.numpy(), c["grad_depth"], atol=1e-12, rtol=1e-6)


@pytest.mark.skipif(not os.path.isdir("/root/reference/utils"), reason="reference checkout not present (GPU box)")
def test_oracle_matches_reference_import():
    sys.path.insert(0, "/root/reference")
    try:
        from utils.loss_utils import l1_loss as ref_l1, ssim as ref_ssim
    finally:
        sys.path.pop(0)
    rng = np.random.default_rng(3)
    img = torch.tensor(rng.random((3, 40, 56)).astype(np.float32))
    gt = torch.tensor(rng.random((3, 40, 56)).astype(np.float32))
    gt[:, 5:9] = 0
    assert torch.allclose(ref_l1(img, gt)[1], loss_oracle.l1_loss(img, gt), atol=1e-7)
    assert torch.allclose(ref_ssim(img, gt)[1], loss_oracle.ssim(img, gt), atol=1e-6)


def test_ssim_gradient_formula_in_float64():
    """The closed-form SSIM gradient the HIP kernel implements against autograd of the oracle in float64."""
    rng = np.random.default_rng(5)
    H, W = 24, 30
    x = torch.tensor(rng.random((3, H, W)), dtype=torch.float64, requires_grad=True)
    y = torch.tensor(rng.random((3, H, W)), dtype=torch.float64)
    loss_oracle.ssim(x, y).backward()
    w = loss_oracle._window(3, torch.float64)
    conv = lambda t: torch.nn.functional.conv2d(t, w, padding=5, groups=3)
    xd = x.detach()
    mu1, mu2, e11, e22, e12 = conv(xd), conv(y), conv(xd * xd), conv(y * y), conv(xd * y)
    C1, C2 = 0.01 ** 2, 0.03 ** 2
    a, b = 2 * mu1 * mu2 + C1, 2 * (e12 - mu1 * mu2) + C2
    c, d = mu1 ** 2 + mu2 ** 2 + C1, (e11 - mu1 ** 2) + (e22 - mu2 ** 2) + C2
    S = a * b / (c * d)
    N = S.numel()
    A = (2 * mu2 * (b - a) / (c * d) - 2 * mu1 * S * (d - c) / (c * d)) / N
    B = (-S / d) / N
    C = (2 * a / (c * d)) / N
    g = conv(A) + 2 * xd * conv(B) + y * conv(C)
    np.testing.assert_allclose(g.numpy(), x.grad.numpy(), atol=1e-12, rtol=1e-9)


def test_window_construction_used_by_the_hip_host_code_is_bit_identical():
    """gsicp_mapper_loss builds the 2-D window on the host as: float32 exp, exact (double) sum rounded to float32, float32 divide,
    float32 outer product.  That must reproduce create_window [REF utils/loss_utils.py:27-35] bit-for-bit."""
    import math
    g = np.array([np.float32(math.exp(-((i - 5) ** 2) / (2.0 * 1.5 * 1.5))) for i in range(11)], np.float32)
    s = np.float32(g.astype(np.float64).sum())
    g = (g / s).astype(np.float32)
    w2 = (g[:, None] * g[None, :]).astype(np.float32)
    assert np.array_equal(w2, loss_oracle._window(1, torch.float32)[0, 0].numpy())
    if os.path.isdir("/root/reference/utils"):
        sys.path.insert(0, "/root/reference")
        try:
            from utils.loss_utils import create_window
        finally:
            sys.path.pop(0)
        assert np.array_equal(w2, create_window(11, 1)[0, 0].numpy())
