"""Pins the two host-code restatements to the reference's OWN methods, lifted unmodified out of mp_Tracker.py and
scene/gaussian_model.py with `ast` and executed by tests/golden/make_golden_hostcode.py:
  * oracle/frontend_oracle.py (and DepthFrontEnd's pick table)  vs  Tracker.set_downsample_filter / downsample_and_make_pointcloud2;
  * tests/test_store_gpu.py::RefModel                            vs  GaussianModel.cat_tensors_to_optimizer / densification_postfix /
                                                                     _prune_optimizer / prune_points.
The HIP paths are tested bit-for-bit against these restatements on the GPU (tests/test_frontend.py, tests/test_store_gpu.py)."""
import os

import numpy as np
import pytest
import torch

GOLD = os.path.join(os.path.dirname(__file__), "golden", "ref_hostcode.npz")
NAMES = ("xyz", "f_dc", "f_rest", "opacity", "scaling", "rotation")


@pytest.fixture(scope="module")
def gold():
    return np.load(GOLD)


@pytest.mark.parametrize("tag", ["a", "b"])
def test_frontend_restatement_equals_reference_methods(gold, tag):
    from oracle import frontend_oracle as fo
    from gs_icp_slam_amd.frontend import DepthFrontEnd
    H, W, fx, fy, cx, cy, stride, dscale, trunc = gold[f"fe_{tag}_cfg"]
    H, W, stride = int(H), int(W), int(stride)
    pick, x_pre, y_pre = fo.downsample_filter(H, W, fx, fy, cx, cy, stride)
    assert np.array_equal(pick.numpy(), gold[f"fe_{tag}_pick"])
    assert np.array_equal(x_pre.numpy(), gold[f"fe_{tag}_xpre"]) and np.array_equal(y_pre.numpy(), gold[f"fe_{tag}_ypre"])
    pts, col, z, trk = fo.make_pointcloud(gold[f"fe_{tag}_depth"], gold[f"fe_{tag}_rgb"], pick, x_pre, y_pre, float(dscale), float(trunc))
    assert np.array_equal(pts, gold[f"fe_{tag}_points"]) and np.array_equal(col, gold[f"fe_{tag}_colors"])
    assert np.array_equal(z, gold[f"fe_{tag}_z"]) and np.array_equal(trk, gold[f"fe_{tag}_filter"])
    fe = DepthFrontEnd(H, W, fx, fy, cx, cy, stride, dscale, trunc, device="cpu")     # the product mirror's table (index arithmetic)
    assert np.array_equal(fe.pick_idx_cpu.numpy(), gold[f"fe_{tag}_pick"])
    assert np.array_equal(fe.x_pre_cpu.numpy(), gold[f"fe_{tag}_xpre"]) and np.array_equal(fe.y_pre_cpu.numpy(), gold[f"fe_{tag}_ypre"])


def test_store_reference_restatement_equals_reference_methods(gold):
    from tests.test_store_gpu import RefModel
    first = {n: torch.from_numpy(gold[f"st_first_{n}"]) for n in NAMES}
    ref = RefModel(first, torch.from_numpy(gold["st_first_trackable"]), device="cpu")
    for n in NAMES:
        ref.opt.state[ref.p[n]] = {"step": torch.tensor(3.0), "exp_avg": torch.from_numpy(gold[f"st_first_m_{n}"].copy()),
                                   "exp_avg_sq": torch.from_numpy(gold[f"st_first_v_{n}"].copy())}
    for step, kind in enumerate(gold["st_ops"]):
        if kind == "cat":
            new = {n: torch.from_numpy(gold[f"st_op{step}_new_{n}"]) for n in NAMES}
            ref.cat(new, torch.from_numpy(gold[f"st_op{step}_new_trackable"]))
        else:
            ref.accum = torch.arange(ref.p["xyz"].shape[0], dtype=torch.float32)[:, None].clone()
            ref.prune(torch.from_numpy(gold[f"st_op{step}_mask"]))
            assert np.array_equal(ref.accum.numpy(), gold[f"st_op{step}_accum"])
        for n in NAMES:
            st = ref.opt.state[ref.p[n]]
            assert np.array_equal(ref.p[n].detach().numpy(), gold[f"st_op{step}_{n}"]), (step, n)
            assert np.array_equal(st["exp_avg"].numpy(), gold[f"st_op{step}_m_{n}"]), (step, n)
            assert np.array_equal(st["exp_avg_sq"].numpy(), gold[f"st_op{step}_v_{n}"]), (step, n)
        assert np.array_equal(ref.trackable.numpy(), gold[f"st_op{step}_trackable"])
    assert ref.p["xyz"].shape[0] == gold[f"st_op{len(gold['st_ops']) - 1}_xyz"].shape[0] > 0


def test_quaternion_composition_and_overlap_statistics_equal_reference_methods(gold):
    from gs_icp_slam_amd.frontend import overlap_statistics, quaternion_multiply, rotation_to_quaternion_xyzw
    got = quaternion_multiply(torch.from_numpy(gold["qm_q1"]), torch.from_numpy(gold["qm_Q2"])).numpy()
    np.testing.assert_allclose(got, gold["qm_out"], rtol=0, atol=1e-15)
    q = rotation_to_quaternion_xyzw(gold["qm_R"]).numpy()
    ref = gold["qm_q1"]
    assert min(np.abs(q - ref).max(), np.abs(q + ref).max()) < 1e-12          # same rotation (sign of a quaternion is free)
    for R_test in (np.eye(3), np.diag([1.0, -1.0, -1.0]), np.diag([-1.0, 1.0, -1.0]), np.diag([-1.0, -1.0, 1.0])):   # w = 0 branches
        qq = rotation_to_quaternion_xyzw(R_test).numpy()
        x, y, z, w = qq
        Rb = np.array([[1 - 2 * (y * y + z * z), 2 * (x * y - w * z), 2 * (x * z + w * y)], [2 * (x * y + w * z), 1 - 2 * (x * x + z * z), 2 * (y * z - w * x)],
                       [2 * (x * z - w * y), 2 * (y * z + w * x), 1 - 2 * (x * x + y * y)]])
        np.testing.assert_allclose(Rb, R_test, atol=1e-12)
    ratio, new_idx = overlap_statistics(torch.from_numpy(gold["ov_d"]), 5e-4, 5e-5)
    assert abs(ratio - int(gold["ov_len_corres"]) / len(gold["ov_d"])) < 1e-12
    assert np.array_equal(new_idx.numpy(), gold["ov_new"])


@pytest.mark.parametrize("deg", [0, 3])
def test_gaussian_initialisation_equals_reference_create_from_pcd2_tensor(gold, deg):
    from gs_icp_slam_amd.gaussian_store import rows_from_gicp
    t = lambda k: torch.from_numpy(gold[f"init{deg}_in_{k}"])
    rows, mask = rows_from_gicp(t("points"), t("colors"), t("rots"), t("scales"), t("z"), t("trk"), max_sh_degree=deg)
    for k in NAMES:
        assert rows[k].shape == gold[f"init{deg}_{k}"].shape, k
        assert np.array_equal(rows[k].numpy(), gold[f"init{deg}_{k}"]), k
    assert np.array_equal(mask.numpy(), gold[f"init{deg}_trackable"])
