import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("P", [1, 3, 4, 257, 5000])
def test_dist2_matches_oracle(P):
    import torch
    import oracle
    from simple_knn._C import distCUDA2
    pts = np.random.default_rng(P).normal(size=(P, 3)).astype(np.float32)
    got = distCUDA2(torch.from_numpy(pts).cuda()).cpu().numpy()
    want = oracle.knn_dist2(pts)
    np.testing.assert_allclose(got, want, rtol=2e-6, atol=1e-12)
    perm = np.random.default_rng(1).permutation(P)
    got_p = distCUDA2(torch.from_numpy(pts[perm]).cuda()).cpu().numpy()
    np.testing.assert_allclose(got_p, got[perm], rtol=2e-6, atol=1e-12)   # permutation invariance


def test_dist2_rejects_cpu_tensor():
    import torch
    from simple_knn._C import distCUDA2
    with pytest.raises(RuntimeError):
        distCUDA2(torch.zeros(4, 3))
