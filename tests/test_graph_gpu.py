"""GPU tests of the sync-free forward (capacity mode), the capturable Adam and the captured mapper iteration: each must
reproduce the synchronous / eager path, which the other test files pin against the oracle."""
import numpy as np
import pytest
import torch

from gs_icp_slam_amd import synth
from tests.util import make_settings, torch_inputs

pytestmark = pytest.mark.gpu


def _scene(P=20000, W=320, H=200, seed=5):
    cfg = dict(synth.REPLICA)
    cam = synth.make_camera(W, H, cfg["fx"] * W / cfg["W"], cfg["fy"] * H / cfg["H"], synth.DEFAULT_POSE_A)
    cam["fx"], cam["fy"] = cfg["fx"] * W / cfg["W"], cfg["fy"] * H / cfg["H"]
    g = synth.s_map(P, seed=seed)
    return g, cam


def _render(rs, t, seed=0):
    from diff_gaussian_rasterization import GaussianRasterizer
    rast = GaussianRasterizer(rs)
    means2D = torch.zeros_like(t["means3D"], requires_grad=True)
    depth, color, radii, used = rast(means3D=t["means3D"], means2D=means2D, shs=t["shs"], opacities=t["opacities"], scales=t["scales"],
                                     rotations=t["rotations"])
    gen = torch.Generator(device="cuda").manual_seed(seed)
    wc = torch.rand(color.shape, device="cuda", generator=gen)
    wd = torch.rand(depth.shape, device="cuda", generator=gen)
    ((color * wc).sum() + (depth * wd).sum()).backward()
    grads = {k: v.grad.clone() for k, v in t.items() if v.grad is not None}
    grads["means2D"] = means2D.grad.clone()
    for v in t.values():
        v.grad = None
    return depth.detach(), color.detach(), radii, used, grads, rast


def test_async_forward_backward_equals_synchronous_path():
    g, cam = _scene()
    t = torch_inputs(g, requires_grad=True)
    rs = make_settings(cam, [0.1, 0.2, 0.3])
    d0, c0, r0, u0, g0, rast0 = _render(rs, t)
    node_R = int((r0 > 0).sum())
    assert node_R > 0
    for capacity in (4_000_000, 1_000_003):    # different capacities -> different split-block counts, same result
        d1, c1, r1, u1, g1, rast1 = _render(rs._replace(capacity=capacity), t)
        R = int(rast1.num_rendered.item())
        assert 0 < R <= capacity
        assert torch.equal(d0, d1) and torch.equal(c0, c1) and torch.equal(r0, r1) and torch.equal(u0, u1)
        for k in g0:
            assert torch.equal(g0[k], g1[k]), k


def test_async_overflow_renders_nothing_and_reports_the_count():
    g, cam = _scene()
    t = torch_inputs(g, requires_grad=True)
    rs = make_settings(cam, [0.1, 0.2, 0.3])
    *_, rast_big = _render(rs._replace(capacity=4_000_000), t)
    R = int(rast_big.num_rendered.item())
    d, c, r, u, grads, rast = _render(rs._replace(capacity=R - 1), t)
    assert int(rast.num_rendered.item()) == R                  # the caller sees the true count and can retry
    assert torch.all(d == 0) and torch.all(u == 0)
    assert torch.allclose(c, torch.tensor([0.1, 0.2, 0.3], device="cuda").view(3, 1, 1).expand_as(c))
    for k, v in grads.items():
        assert torch.all(v == 0), k
    # exactly at capacity is fine
    d2, c2, *_ = _render(rs._replace(capacity=R), t)
    d0, c0, *_ = _render(rs, t)
    assert torch.equal(d0, d2) and torch.equal(c0, c2)


def test_capturable_adam_equals_host_step_adam():
    from gs_icp_slam_amd.optim import FusedAdam
    torch.manual_seed(0)
    shapes = [(1000, 3), (1000, 1, 3), (1000, 1), (1000, 4)]
    lrs = [1e-3, 2.5e-3, 0.05, 1e-3]
    p_a = [torch.randn(s, device="cuda", requires_grad=True) for s in shapes]
    p_b = [p.detach().clone().requires_grad_(True) for p in p_a]
    oa = FusedAdam([{"params": [p], "lr": lr} for p, lr in zip(p_a, lrs)], lr=0.0, eps=1e-15)
    ob = FusedAdam([{"params": [p], "lr": lr} for p, lr in zip(p_b, lrs)], lr=0.0, eps=1e-15, capturable=True)
    for it in range(12):
        if it == 6:                       # learning-rate change is picked up by both
            for o in (oa, ob):
                o.param_groups[0]["lr"] = 5e-3
            ob.sync_lr()
        for pa, pb in zip(p_a, p_b):
            gr = torch.randn_like(pa)
            pa.grad, pb.grad = gr.clone(), gr.clone()
        oa.step()
        ob.step()
    assert int(ob.state[p_b[0]]["step"].item()) == 12
    for pa, pb in zip(p_a, p_b):
        # both variants round like torch.optim.Adam up to the last bits of step_size and 1/sqrt(1 - beta2^t) (host pow vs device pow):
        # a few parts per million of one update (lr <= 0.05), which is the scale that matters for elements that happen to sit near zero
        torch.testing.assert_close(pa, pb, rtol=2e-6, atol=2e-7)


def test_row_freeze_and_in_kernel_step_bump():
    """Round 5: (a) FusedAdam.set_row_freeze — the named groups leave the rows with a non-zero mask word bit for bit alone (parameter and both
    moments) while every other row, and every row of the other groups, steps exactly as an optimiser without the mask; (b) the step counter of the
    capturable path is advanced by the LAST WORKGROUP of the Adam launch (no bump kernel): it counts every step once, also across a captured and
    replayed graph, and a tripped guard counts a skipped step instead."""
    from gs_icp_slam_amd.optim import FusedAdam
    torch.manual_seed(1)
    names = ["xyz", "f_dc", "opacity", "scaling", "rotation"]
    shapes = [(3001, 3), (3001, 1, 3), (3001, 1), (3001, 3), (3001, 4)]
    lrs = [4e-6, 2.5e-3, 0.05, 5e-3, 1e-3]
    p_a = [torch.randn(s, device="cuda", requires_grad=True) for s in shapes]
    p_b = [p.detach().clone().requires_grad_(True) for p in p_a]
    start = [p.detach().clone() for p in p_a]
    oa = FusedAdam([{"params": [p], "lr": lr, "name": n} for p, lr, n in zip(p_a, lrs, names)], lr=0.0, eps=1e-15, capturable=True)
    ob = FusedAdam([{"params": [p], "lr": lr, "name": n} for p, lr, n in zip(p_b, lrs, names)], lr=0.0, eps=1e-15, capturable=True)
    mask = (torch.rand(3001, device="cuda") < 0.4).to(torch.int32)
    ob.set_row_freeze(mask, ("xyz", "scaling", "rotation"))
    guard = torch.zeros(1, dtype=torch.int32, device="cuda")
    ob.set_overflow_guard(guard, 10)
    grads = [[torch.randn_like(p) for p in p_a] for _ in range(9)]

    def feed(k):
        for pa, pb, gr in zip(p_a, p_b, grads[k]):
            pa.grad, pb.grad = gr.clone(), gr.clone()
    for k in range(3):                      # eager steps (the first allocates the device lr arrays and the done word)
        feed(k)
        oa.step(); ob.step()
    assert int(ob.state[p_b[0]]["step"].item()) == 3 and int(oa.state[p_a[0]]["step"].item()) == 3
    # captured + replayed: static gradient tensors
    for pb in p_b:
        pb.grad = torch.zeros_like(pb)
    gph = torch.cuda.CUDAGraph()
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        for pb, gr in zip(p_b, grads[3]):
            pb.grad.copy_(gr)
    torch.cuda.current_stream().wait_stream(side)
    torch.cuda.synchronize()
    with torch.cuda.graph(gph):
        ob.step()
    # the capture itself executes nothing: steps 3..7 are replays (one with the guard tripped)
    for k in range(3, 8):
        for pb, gr in zip(p_b, grads[k]):
            pb.grad.copy_(gr)
        if k == 5:
            guard.fill_(11)                 # tripped: nothing moves, the step is not counted, the skipped counter is
        gph.replay()
        guard.zero_()
        if k != 5:
            for pa, gr in zip(p_a, grads[k]):
                pa.grad = gr.clone()
            oa.step()
    torch.cuda.synchronize()
    assert int(ob.state[p_b[0]]["step"].item()) == 7 and int(ob.skipped_steps.item()) == 1
    assert int(oa.state[p_a[0]]["step"].item()) == 7
    frozen = mask.bool()
    for n, pa, pb, s0 in zip(names, p_a, p_b, start):
        if n in ("xyz", "scaling", "rotation"):
            assert torch.equal(pb.detach()[frozen], s0[frozen]), f"{n}: frozen rows moved"
            assert torch.equal(ob.state[pb]["exp_avg"][frozen], torch.zeros_like(s0[frozen])), f"{n}: frozen rows' moments moved"
            assert torch.equal(pb.detach()[~frozen], pa.detach()[~frozen]), f"{n}: free rows differ from the unmasked optimiser"
        else:
            assert torch.equal(pb.detach(), pa.detach()), f"{n}: a group outside the freeze differs"


def _mapper_setup(P, W, H, capturable):
    from gs_icp_slam_amd.optim import FusedAdam
    g, cam = _scene(P, W, H)
    raw = {"means3D": torch.from_numpy(g["means3D"]), "scales": torch.log(torch.from_numpy(g["scales"])),
           "rotations": torch.from_numpy(g["rotations"]), "opacities": torch.logit(torch.from_numpy(g["opacities"]).clamp(1e-4, 1 - 1e-4)),
           "shs": torch.from_numpy(g["shs"])}
    params = {k: v.cuda().contiguous().requires_grad_(True) for k, v in raw.items()}
    lrs = {"means3D": 4e-6, "shs": 2.5e-3, "opacities": 0.05, "scales": 5e-3, "rotations": 1e-3}
    opt = FusedAdam([{"params": [params[k]], "lr": lr} for k, lr in lrs.items()], lr=0.0, eps=1e-15, capturable=capturable)
    return g, cam, params, opt


def test_captured_iteration_equals_eager_iterations():
    from diff_gaussian_rasterization import GaussianRasterizer
    from gs_icp_slam_amd.graph import MapperIterationGraph, default_activations
    from gs_icp_slam_amd.loss import mapper_loss_parts
    P, W, H = 20000, 320, 200
    g, cam, params_e, opt_e = _mapper_setup(P, W, H, capturable=False)
    _, _, params_g, opt_g = _mapper_setup(P, W, H, capturable=True)
    rs = make_settings(cam, [0.0, 0.0, 0.0])
    # two keyframes (views + targets): targets rendered from a perturbed map
    views = []
    for pose_seed in (0, 1):
        pose = synth.DEFAULT_POSE_A if pose_seed == 0 else synth.se3((12.5, 27.0, 0.5), (-0.88, -0.22, -1.08))
        cam_k = synth.make_camera(W, H, cam["fx"], cam["fy"], pose)
        rs_k = make_settings(cam_k, [0.0, 0.0, 0.0])
        g2 = synth.s_map(P, seed=5, perturb_seed=7)
        t2 = torch_inputs(g2)
        with torch.no_grad():
            d, c, _, _ = GaussianRasterizer(rs_k)(means3D=t2["means3D"], means2D=torch.zeros_like(t2["means3D"]), shs=t2["shs"],
                                                  opacities=t2["opacities"], scales=t2["scales"], rotations=t2["rotations"])
        views.append((rs_k, c.clone(), d.clone()))
    schedule = [0, 1, 0, 1, 1, 0]    # capture() applies NO update (its warm-up is rolled back): the graph side replays all six

    losses_e = []
    for k in schedule:
        rs_k, gt_c, gt_d = views[k]
        a = default_activations(params_e)
        m2 = torch.zeros_like(a["means3D"], requires_grad=True)
        depth, color, _, _ = GaussianRasterizer(rs_k)(means3D=a["means3D"], means2D=m2, shs=a["shs"], opacities=a["opacities"],
                                                      scales=a["scales"], rotations=a["rotations"])
        loss, parts = mapper_loss_parts(color, depth, gt_c, gt_d)
        loss.backward()
        opt_e.step()
        opt_e.zero_grad(set_to_none=True)
        losses_e.append(float(loss))

    mg = MapperIterationGraph(params_g, opt_g, H, W, cam["tanfovx"], cam["tanfovy"], sh_degree=0, capacity=2_000_000, warmup=2)
    rs0, c0, d0 = views[0]
    mg.set_view(rs0.viewmatrix, rs0.projmatrix, rs0.campos, c0, d0)
    before = {k: v.detach().clone() for k, v in params_g.items()}
    mg.capture()                      # 2 warm-up iterations on view 0, rolled back, then capture (capture itself executes nothing)
    for k in before:
        assert torch.equal(before[k], params_g[k]), f"capture() moved {k}"
    assert int(opt_g.state[params_g["means3D"]]["step"].item()) == 0
    assert all(not bool(opt_g.state[p]["exp_avg"].any()) and not bool(opt_g.state[p]["exp_avg_sq"].any()) for p in params_g.values())
    losses_g = []
    for k in schedule:
        rs_k, gt_c, gt_d = views[k]
        mg.set_view(rs_k.viewmatrix, rs_k.projmatrix, rs_k.campos, gt_c, gt_d)
        losses_g.append(float(mg.step()))
        assert not mg.overflowed()
    np.testing.assert_allclose(losses_g, losses_e, rtol=1e-5)
    assert int(opt_g.state[params_g["means3D"]]["step"].item()) == len(schedule)
    for k in params_e:
        torch.testing.assert_close(params_g[k], params_e[k], rtol=1e-4, atol=1e-6, msg=lambda m, k=k: f"{k}: {m}")
    # regression: replays after a full device synchronise (memset nodes in the graph once came back with garbage here)
    torch.cuda.synchronize()
    for _ in range(8):
        loss = mg.step()
    torch.cuda.synchronize()
    assert not mg.overflowed() and 0 < int(mg.num_rendered.item()) < 2_000_000
    assert float(loss) < losses_g[-1]


def test_overflowing_replay_skips_the_optimiser_step_on_the_device():
    """a replay whose duplicate count exceeds the capacity renders nothing and yields all-zero gradients; Adam must not move
    the parameters on stale momentum, decay exp_avg_sq or advance the step count.  The guard lives in the Adam kernels (no host sync)."""
    from gs_icp_slam_amd.graph import MapperIterationGraph
    P, W, H = 20000, 320, 200
    g, cam, params, opt = _mapper_setup(P, W, H, capturable=True)
    rs = make_settings(cam, [0.0, 0.0, 0.0])
    gt_c = torch.rand((3, H, W), device="cuda")
    gt_d = torch.rand((1, H, W), device="cuda") + 1.0
    # capacity fits view A but not view B (a camera pulled back: many more tiles per Gaussian... here simply a far smaller capacity)
    probe = MapperIterationGraph(params, opt, H, W, cam["tanfovx"], cam["tanfovy"], sh_degree=0, capacity=4_000_000, warmup=1)
    probe.set_view(rs.viewmatrix, rs.projmatrix, rs.campos, gt_c, gt_d)
    probe.capture()
    probe.step()
    R = int(probe.num_rendered.item())
    steps0 = int(opt.state[params["means3D"]]["step"].item())
    assert steps0 == 1 and probe.skipped_steps() == 0
    del probe
    small = MapperIterationGraph(params, opt, H, W, cam["tanfovx"], cam["tanfovy"], sh_degree=0, capacity=R // 2, warmup=1)
    small.set_view(rs.viewmatrix, rs.projmatrix, rs.campos, gt_c, gt_d)
    small.capture()
    snap = {k: v.detach().clone() for k, v in params.items()}
    snap_m = {k: opt.state[v]["exp_avg"].clone() for k, v in params.items()}
    snap_v = {k: opt.state[v]["exp_avg_sq"].clone() for k, v in params.items()}
    for _ in range(3):
        small.step()
    assert small.overflowed() and small.skipped_steps() == 3
    assert int(opt.state[params["means3D"]]["step"].item()) == steps0
    for k, v in params.items():
        assert torch.equal(v, snap[k]), f"{k} moved on an overflowed replay"
        assert torch.equal(opt.state[v]["exp_avg"], snap_m[k]) and torch.equal(opt.state[v]["exp_avg_sq"], snap_v[k])
    # capacity auto-grow: the host notices the skipped steps, the lists are enlarged, the iteration is re-captured over the same parameters
    # and optimiser state (capturing applies no update), and the lost steps can be repeated
    lost = small.ensure_capacity()
    assert lost == 3 and small.regrowths == 1 and small.capacity >= R
    for k, v in params.items():
        assert torch.equal(v, snap[k]), f"{k} moved while re-capturing"
    for _ in range(lost):
        small.step()
    assert not small.overflowed() and small.skipped_steps() == 3 and small.ensure_capacity() == 0 and small.regrowths == 1
    assert int(opt.state[params["means3D"]]["step"].item()) == steps0 + 3
    assert not torch.equal(params["means3D"], snap["means3D"])


def test_eager_steps_after_a_capture_keep_lr_arrays_and_are_not_gated_by_the_graphs_guard():
    """(i) an eager capturable step() that sees fewer gradients must not free device lr arrays a captured graph still reads;
    (ii) the overflow guard / live-row count a graph binds are scoped to ITS launches — an eager step on the same optimiser afterwards is
    not skipped because the graph's last replay overflowed."""
    from gs_icp_slam_amd.graph import MapperIterationGraph
    P, W, H = 6000, 160, 96
    g, cam, params, opt = _mapper_setup(P, W, H, capturable=True)
    rs = make_settings(cam, [0.0, 0.0, 0.0])
    gt_c = torch.rand((3, H, W), device="cuda")
    gt_d = torch.rand((1, H, W), device="cuda") + 1.0
    mg = MapperIterationGraph(params, opt, H, W, cam["tanfovx"], cam["tanfovy"], sh_degree=0, capacity=2_000_000, warmup=1)
    mg.set_view(rs.viewmatrix, rs.projmatrix, rs.campos, gt_c, gt_d)
    mg.capture()
    assert opt._guard is None and opt._live_rows is None        # nothing stays bound after capture()
    lr_ptrs = {k: v[0].data_ptr() for k, v in opt._lr_dev.items()}
    assert lr_ptrs
    # an eager step where only ONE parameter has a gradient (different bucket) ...
    for p_ in params.values():
        p_.grad = None
    params["opacities"].grad = torch.zeros_like(params["opacities"])
    opt.step()
    # ... must leave the graph's lr arrays alive and in place
    assert all(k in opt._lr_dev and opt._lr_dev[k][0].data_ptr() == ptr for k, ptr in lr_ptrs.items())
    before = params["means3D"].detach().clone()
    mg.step()
    torch.cuda.synchronize()
    assert not torch.equal(params["means3D"], before) and torch.isfinite(params["means3D"]).all()
    # (ii) a tiny-capacity graph overflows on replay; an eager step afterwards still updates
    tiny = MapperIterationGraph(params, opt, H, W, cam["tanfovx"], cam["tanfovy"], sh_degree=0, capacity=16, warmup=1)
    tiny.set_view(rs.viewmatrix, rs.projmatrix, rs.campos, gt_c, gt_d)
    tiny.capture()
    tiny.step()
    assert tiny.overflowed() and tiny.skipped_steps() >= 1
    before = params["opacities"].detach().clone()
    for p_ in params.values():
        p_.grad = None
    params["opacities"].grad = torch.ones_like(params["opacities"])
    opt.step()
    torch.cuda.synchronize()
    assert not torch.equal(params["opacities"], before), "an eager step was gated by a graph's stale overflow count"


def test_one_graph_per_training_stage_resolution_shares_parameters_and_optimiser():
    """render_3's coarse-to-fine `training_stage` renders at W/2 x H/2 or W/4 x H/4 with the same tan(fov/2) [REF gaussian_renderer/__init__.py:238-242;
    mp_Mapper.py:207-216].  A captured iteration is tied to one resolution, so the mapper keeps one MapperIterationGraph per stage over the SAME
    parameters and the SAME capturable optimiser; alternating them must equal the eager iterations at the respective resolutions (each graph's
    captured Adam launches carry that graph's own overflow guard)."""
    from diff_gaussian_rasterization import GaussianRasterizer
    from gs_icp_slam_amd.graph import MapperIterationGraph, default_activations
    from gs_icp_slam_amd.loss import mapper_loss_parts
    P, W, H = 15000, 320, 192
    g, cam, params_e, opt_e = _mapper_setup(P, W, H, capturable=False)
    _, _, params_g, opt_g = _mapper_setup(P, W, H, capturable=True)
    stages = {0: (W, H), 1: (W // 2, H // 2)}
    views, graphs = {}, {}
    for st, (w, h) in stages.items():
        cam_s = synth.make_camera(w, h, cam["fx"] * w / W, cam["fy"] * h / H, synth.DEFAULT_POSE_A)
        rs_s = make_settings(cam_s, [0.0, 0.0, 0.0])
        t2 = torch_inputs(synth.s_map(P, seed=5, perturb_seed=7))
        with torch.no_grad():
            d, c, _, _ = GaussianRasterizer(rs_s)(means3D=t2["means3D"], means2D=torch.zeros_like(t2["means3D"]), shs=t2["shs"],
                                                  opacities=t2["opacities"], scales=t2["scales"], rotations=t2["rotations"])
        views[st] = (rs_s, c.clone(), d.clone())
        assert abs(cam_s["tanfovx"] - cam["tanfovx"]) < 1e-12
    for st, (w, h) in stages.items():        # both graphs are BUILT before either is captured: the guard must not leak from one to the other
        graphs[st] = MapperIterationGraph(params_g, opt_g, h, w, cam["tanfovx"], cam["tanfovy"], sh_degree=0, capacity=(2_000_000 if st == 0 else 700_000), warmup=1)
    for st in stages:
        rs_s, c, d = views[st]
        graphs[st].set_view(rs_s.viewmatrix, rs_s.projmatrix, rs_s.campos, c, d)
        graphs[st].capture()
    schedule = [0, 1, 1, 0, 1, 0]
    for st in schedule:
        rs_s, gt_c, gt_d = views[st]
        a = default_activations(params_e)
        m2 = torch.zeros_like(a["means3D"], requires_grad=True)
        depth, color, _, _ = GaussianRasterizer(rs_s)(means3D=a["means3D"], means2D=m2, shs=a["shs"], opacities=a["opacities"], scales=a["scales"],
                                                      rotations=a["rotations"])
        loss, _ = mapper_loss_parts(color, depth, gt_c, gt_d)
        loss.backward()
        opt_e.step()
        opt_e.zero_grad(set_to_none=True)
        lg = float(graphs[st].step())
        assert not graphs[st].overflowed()
        np.testing.assert_allclose(lg, float(loss.detach()), rtol=2e-5)
    assert int(opt_g.state[params_g["means3D"]]["step"].item()) == len(schedule) and graphs[0].skipped_steps() == 0
    for k in params_e:
        torch.testing.assert_close(params_g[k], params_e[k], rtol=1e-4, atol=1e-6, msg=lambda m, k=k: f"{k}: {m}")


def test_graph_rejects_host_step_optimizer():
    from gs_icp_slam_amd.graph import MapperIterationGraph
    g, cam, params, opt = _mapper_setup(1000, 64, 48, capturable=False)
    with pytest.raises(RuntimeError):
        MapperIterationGraph(params, opt, 48, 64, cam["tanfovx"], cam["tanfovy"], 0, capacity=1000)


def test_one_capture_survives_map_growth_and_pruning():
    """the map's parameters and Adam moments live in a GaussianStore(stable=True) (full-capacity buffers, live count on
    the device).  ONE captured MapperIterationGraph then keeps replaying across append (keyframes) and prune with zero re-captures, and every
    replay equals the eager iteration on the live rows of a reference-style (re-created tensors) map."""
    from diff_gaussian_rasterization import GaussianRasterizer
    from gs_icp_slam_amd.gaussian_store import GaussianStore, PARAM_NAMES
    from gs_icp_slam_amd.graph import MapperIterationGraph, default_activations
    from gs_icp_slam_amd.loss import mapper_loss_parts
    from gs_icp_slam_amd.optim import FusedAdam
    P0, W, H = 12000, 320, 200
    g, cam = _scene(P0 + 8000, W, H)
    lrs = {"xyz": 4e-6, "f_dc": 2.5e-3, "f_rest": 1.25e-4, "opacity": 0.05, "scaling": 5e-3, "rotation": 1e-3}

    def rows(lo, hi):
        return dict(xyz=torch.from_numpy(g["means3D"][lo:hi]).cuda(), f_dc=torch.from_numpy(g["shs"][lo:hi]).cuda(),
                    f_rest=torch.zeros((hi - lo, 0, 3), device="cuda"),
                    opacity=torch.logit(torch.from_numpy(g["opacities"][lo:hi]).clamp(1e-4, 1 - 1e-4)).cuda(),
                    scaling=torch.log(torch.from_numpy(g["scales"][lo:hi])).cuda(), rotation=torch.from_numpy(g["rotations"][lo:hi]).cuda())

    # graph side: stable store, ONE capture
    store = GaussianStore(40000, n_rest=0, stable=True)
    store.append(rows(0, P0))
    opt_g = store.attach(FusedAdam, lrs, lr=0.0, eps=1e-15, capturable=True)
    pg = {"means3D": store.params["xyz"], "shs": store.params["f_dc"], "opacities": store.params["opacity"], "scales": store.params["scaling"],
          "rotations": store.params["rotation"]}
    ptrs = {k: v.data_ptr() for k, v in pg.items()}
    # eager side: plain store (tensors re-created on every append / prune, as the reference does), host-step FusedAdam
    ref = GaussianStore(40000, n_rest=0)
    ref.append(rows(0, P0))
    opt_e = ref.attach(FusedAdam, lrs, lr=0.0, eps=1e-15)

    rs = make_settings(cam, [0.0, 0.0, 0.0])
    g2 = synth.s_map(P0 + 8000, seed=5, perturb_seed=7)
    t2 = torch_inputs(g2)
    with torch.no_grad():
        gt_d, gt_c, _, _ = GaussianRasterizer(rs)(means3D=t2["means3D"], means2D=torch.zeros_like(t2["means3D"]), shs=t2["shs"],
                                                  opacities=t2["opacities"], scales=t2["scales"], rotations=t2["rotations"])
    mg = MapperIterationGraph(pg, opt_g, H, W, cam["tanfovx"], cam["tanfovy"], sh_degree=0, capacity=2_000_000, warmup=1,
                              live_count=store.live_count)
    mg.set_view(rs.viewmatrix, rs.projmatrix, rs.campos, gt_c, gt_d)
    mg.capture()
    graph_obj = mg.graph

    def eager_step():
        pe = {"means3D": ref.params["xyz"], "shs": ref.params["f_dc"], "opacities": ref.params["opacity"], "scales": ref.params["scaling"],
              "rotations": ref.params["rotation"]}
        a = default_activations(pe)
        m2 = torch.zeros_like(a["means3D"], requires_grad=True)
        depth, color, _, _ = GaussianRasterizer(rs)(means3D=a["means3D"], means2D=m2, shs=a["shs"], opacities=a["opacities"],
                                                    scales=a["scales"], rotations=a["rotations"])
        loss, _ = mapper_loss_parts(color, depth, gt_c, gt_d)
        loss.backward()
        opt_e.step()
        opt_e.zero_grad(set_to_none=True)
        return float(loss.detach())

    def both(n_it):
        for _ in range(n_it):
            le = eager_step()
            lg = float(mg.step())
            assert not mg.overflowed()
            np.testing.assert_allclose(lg, le, rtol=2e-5)

    both(3)
    for lo, hi in ((P0, P0 + 3000), (P0 + 3000, P0 + 8000)):           # two keyframes: new rows in place, live count bumped on the device
        store.append(rows(lo, hi))
        ref.append(rows(lo, hi))
        both(3)
    remove = torch.rand(store.n, device="cuda", generator=torch.Generator(device="cuda").manual_seed(3)) < 0.25
    store.prune(remove)                                                 # survivors compacted and copied back: same addresses
    ref.prune(remove)
    assert store.n == ref.n and int(store.live_count.item()) == store.n
    both(3)
    store.append(rows(0, 1000))                                         # growth after a prune: claimed rows start with zero moments
    ref.append(rows(0, 1000))
    both(2)
    assert mg.graph is graph_obj, "the graph was re-captured"
    assert all(pg[k].data_ptr() == ptrs[k] for k in pg), "a parameter buffer moved"
    for name in PARAM_NAMES:
        if name == "f_rest":
            continue
        torch.testing.assert_close(store.live(name), ref.params[name].detach(), rtol=1e-4, atol=1e-6, msg=lambda m, n=name: f"{n}: {m}")
        torch.testing.assert_close(store._sets[0][("m", name)][: store.n], ref.view("m", name), rtol=1e-3, atol=1e-9)


def test_sparse_gradient_rows_and_prebumped_step_equal_the_zero_filled_step():
    """Round 6.  (a) FusedAdam.set_grad_row_mask(radii): rows with radii <= 0 take g = 0 WITHOUT their gradient being read — the
    gradient tensors here hold NaN in those rows — and every tensor ends bit for bit where the same optimiser ends on zero-filled gradients (xyz rows of 3
    floats straddle the 16-byte units, opacity rows are single floats, quaternions whole units: all three row widths of the kernel).  (b) step(
    step_already_bumped=True) after the counter was advanced ahead of it, in stream order, equals the step that bumps behind itself; a tripped guard
    skips both alike."""
    from gs_icp_slam_amd.optim import FusedAdam
    torch.manual_seed(3)
    P = 4099
    shapes = [(P, 3), (P, 1, 3), (P, 1), (P, 3), (P, 4)]
    lrs = [4e-6, 2.5e-3, 0.05, 5e-3, 1e-3]
    p_a = [torch.randn(s, device="cuda", requires_grad=True) for s in shapes]
    p_b = [p.detach().clone().requires_grad_(True) for p in p_a]
    oa = FusedAdam([{"params": [p], "lr": lr} for p, lr in zip(p_a, lrs)], lr=0.0, eps=1e-15, capturable=True)
    ob = FusedAdam([{"params": [p], "lr": lr} for p, lr in zip(p_b, lrs)], lr=0.0, eps=1e-15, capturable=True)
    guard = torch.zeros(1, dtype=torch.int32, device="cuda")
    for o in (oa, ob):
        o.set_overflow_guard(guard, 10)
    assert ob.shared_step_tensor() is None                      # no state before the first step
    for it in range(7):
        radii = (torch.rand(P, device="cuda") < (0.2 if it % 2 else 0.7)).to(torch.int32) * 5
        if it == 3:
            radii.zero_()                                       # nothing visible at all: no gradient element is read
        vis = radii > 0
        for pa, pb in zip(p_a, p_b):
            gr = torch.randn_like(pa)
            dense = gr.clone()
            dense[~vis] = 0.0
            sparse = gr.clone()
            sparse[~vis] = float("nan")
            pa.grad, pb.grad = dense, sparse
        if it == 5:
            guard.fill_(11)                                     # tripped: both skip, neither counts the step
        oa.step()
        ob.set_grad_row_mask(radii)
        st = ob.shared_step_tensor()
        if st is None:                                          # first step: the counter does not exist yet
            ob.step()
        else:
            if it != 5:
                st += 1                                         # what the loss kernel's finishing thread does under an untripped guard
            ob.step(step_already_bumped=True)
        ob.set_grad_row_mask(None)
        guard.zero_()
    assert int(oa.state[p_a[0]]["step"].item()) == 6 and int(ob.state[p_b[0]]["step"].item()) == 6
    assert int(oa.skipped_steps.item()) == 1
    for pa, pb in zip(p_a, p_b):
        assert torch.isfinite(pb).all()
        assert torch.equal(pa, pb)
        for name in ("exp_avg", "exp_avg_sq"):
            assert torch.equal(oa.state[pa][name], ob.state[pb][name]), name
    with pytest.raises(RuntimeError):
        FusedAdam([p_a[0]], lr=1e-3).step(step_already_bumped=True)
    with pytest.raises(RuntimeError):
        ob.set_grad_row_mask(torch.zeros(P, device="cuda"))     # not int32


def test_sparse_backward_leaves_the_visible_rows_bit_identical():
    """GaussianRasterizationSettings.sparse_grads: the backward skips the zero rows of culled Gaussians; every visible row of every gradient is the same
    bits as without the flag (and without it every culled row is written with zeros, as the reference's backward does)."""
    g, cam = _scene()
    rs = make_settings(cam, [0.1, 0.2, 0.3])
    raw = {"means3D": torch.from_numpy(g["means3D"]), "scales": torch.log(torch.from_numpy(g["scales"])), "rotations": torch.from_numpy(g["rotations"]),
           "opacities": torch.logit(torch.from_numpy(g["opacities"]).clamp(1e-4, 1 - 1e-4)), "shs": torch.from_numpy(g["shs"])}
    out = {}
    for sparse in (False, True):
        t = {k: v.cuda().contiguous().requires_grad_(True) for k, v in raw.items()}
        d, c, radii, u, grads, rast = _render(rs._replace(capacity=2_000_000, raw_params=True, sparse_grads=sparse), t)
        out[sparse] = (radii.clone(), grads)
    vis = out[False][0] > 0
    assert torch.equal(out[False][0], out[True][0]) and 0 < int(vis.sum()) < vis.numel()
    for k, v in out[False][1].items():
        assert torch.all(v[~vis] == 0), f"{k}: culled rows of the dense backward must be zero"
        assert torch.equal(v[vis], out[True][1][k][vis]), f"{k}: visible rows differ under sparse_grads"


def test_captured_iteration_is_bit_identical_with_and_without_sparse_gradients_and_the_bump_in_the_loss(monkeypatch):
    """The captured mapper iteration of round 6 (sparse gradient rows, the step bump inside the loss kernel, the forward's counters cleared by the keyframe-selection
    launch: 14 kernel nodes) against the same iteration with all three switched off (GSICP_SPARSE_GRADS=0, GSICP_STEP_BUMP_IN_LOSS=0, GSICP_PREZERO=0: zero-filled
    gradients, the one-thread bump launch, the forward's own zero fill): identical parameters, moments, step counts, losses, radii, is_used and duplicate counts after
    replays over two views — one of them without a set_view() in front."""
    from diff_gaussian_rasterization import GaussianRasterizer
    from gs_icp_slam_amd.graph import MapperIterationGraph
    P, W, H = 20000, 320, 200
    views = None
    res = {}
    for mode in ("new", "old"):
        for k in ("GSICP_SPARSE_GRADS", "GSICP_STEP_BUMP_IN_LOSS", "GSICP_PREZERO"):
            if mode == "old":
                monkeypatch.setenv(k, "0")
            else:
                monkeypatch.delenv(k, raising=False)
        g, cam, params, opt = _mapper_setup(P, W, H, capturable=True)
        if views is None:
            views = []
            for pose in (synth.DEFAULT_POSE_A, synth.se3((12.5, 27.0, 0.5), (-0.88, -0.22, -1.08))):
                cam_k = synth.make_camera(W, H, cam["fx"], cam["fy"], pose)
                rs_k = make_settings(cam_k, [0.0, 0.0, 0.0])
                t2 = torch_inputs(synth.s_map(P, seed=5, perturb_seed=7))
                with torch.no_grad():
                    d, c, _, _ = GaussianRasterizer(rs_k)(means3D=t2["means3D"], means2D=torch.zeros_like(t2["means3D"]), shs=t2["shs"],
                                                          opacities=t2["opacities"], scales=t2["scales"], rotations=t2["rotations"])
                views.append((rs_k, c.clone(), d.clone()))
        mg = MapperIterationGraph(params, opt, H, W, cam["tanfovx"], cam["tanfovy"], sh_degree=0, capacity=2_000_000, warmup=2)
        assert mg._sparse == (mode == "new") and mg._bump_in_loss == (mode == "new") and mg._prezero == (mode == "new")
        mg.set_view(views[0][0].viewmatrix, views[0][0].projmatrix, views[0][0].campos, views[0][1], views[0][2])
        mg.capture()
        assert (mg._zero_region is not None) == (mode == "new")      # the captured forward's counter region, cleared by the selection launch instead of a launch of its own
        losses = []
        for k in (0, 1, 1, 0, 1):
            rs_k, gt_c, gt_d = views[k]
            mg.set_view(rs_k.viewmatrix, rs_k.projmatrix, rs_k.campos, gt_c, gt_d)
            losses.append(float(mg.step()))
        losses.append(float(mg.step()))          # a replay WITHOUT a set_view() in front: the same keyframe again (the selection launch is re-issued)
        torch.cuda.synchronize()
        res[mode] = (losses, {k: v.detach().clone() for k, v in params.items()},
                     {k: (opt.state[v]["exp_avg"].clone(), opt.state[v]["exp_avg_sq"].clone()) for k, v in params.items()},
                     int(opt.state[params["means3D"]]["step"].item()), mg.skipped_steps(), mg.is_used.clone(), mg.radii.clone(), int(mg.num_rendered.item()))
        mg.release()
    assert res["new"][0] == res["old"][0] and res["new"][3] == res["old"][3] == 6 and res["new"][4] == res["old"][4] == 0
    assert torch.equal(res["new"][5], res["old"][5]) and torch.equal(res["new"][6], res["old"][6]) and res["new"][7] == res["old"][7] > 0    # is_used, radii, duplicate count
    for k in res["new"][1]:
        assert torch.equal(res["new"][1][k], res["old"][1][k]), k
        assert torch.equal(res["new"][2][k][0], res["old"][2][k][0]) and torch.equal(res["new"][2][k][1], res["old"][2][k][1]), k


@pytest.mark.parametrize("which", ["default_activations", "torch_activations"])
def test_graph_with_separate_activation_operators_follows_the_fused_graph(which):
    """MapperIterationGraph(activations=...): the activation getters as operators of their own (one fused launch, or the reference's torch ops) instead of inside
    the rasteriser's preprocess kernels — the dense-gradient form of the captured iteration (no sparse rows: the activation backward reads every row), with the
    step bump in the loss and the pre-zeroed forward still on.  Four replays over two views follow the fused-activation graph: same losses to 1e-5, parameters
    within Adam's last-bits tolerance."""
    import gs_icp_slam_amd.graph as graph_mod
    from diff_gaussian_rasterization import GaussianRasterizer
    P, W, H = 20000, 320, 200
    res = {}
    views = None
    for mode in ("fused", which):
        g, cam, params, opt = _mapper_setup(P, W, H, capturable=True)
        if views is None:
            views = []
            for pose in (synth.DEFAULT_POSE_A, synth.se3((12.5, 27.0, 0.5), (-0.88, -0.22, -1.08))):
                cam_k = synth.make_camera(W, H, cam["fx"], cam["fy"], pose)
                rs_k = make_settings(cam_k, [0.0, 0.0, 0.0])
                t2 = torch_inputs(synth.s_map(P, seed=5, perturb_seed=7))
                with torch.no_grad():
                    d, c, _, _ = GaussianRasterizer(rs_k)(means3D=t2["means3D"], means2D=torch.zeros_like(t2["means3D"]), shs=t2["shs"],
                                                          opacities=t2["opacities"], scales=t2["scales"], rotations=t2["rotations"])
                views.append((rs_k, c.clone(), d.clone()))
        act = None if mode == "fused" else getattr(graph_mod, mode)
        mg = graph_mod.MapperIterationGraph(params, opt, H, W, cam["tanfovx"], cam["tanfovy"], sh_degree=0, capacity=2_000_000, warmup=2, activations=act)
        assert mg._sparse == (mode == "fused") and mg._bump_in_loss and mg._prezero
        mg.set_view(views[0][0].viewmatrix, views[0][0].projmatrix, views[0][0].campos, views[0][1], views[0][2])
        mg.capture()
        losses = []
        for k in (0, 1, 1, 0):
            rs_k, gt_c, gt_d = views[k]
            mg.set_view(rs_k.viewmatrix, rs_k.projmatrix, rs_k.campos, gt_c, gt_d)
            losses.append(float(mg.step()))
        torch.cuda.synchronize()
        assert not mg.overflowed() and mg.skipped_steps() == 0 and int(opt.state[params["means3D"]]["step"].item()) == 4
        res[mode] = (losses, {k: v.detach().clone() for k, v in params.items()})
        mg.release()
    np.testing.assert_allclose(res[which][0], res["fused"][0], rtol=1e-5)
    for k in res["fused"][1]:
        torch.testing.assert_close(res[which][1][k], res["fused"][1][k], rtol=1e-4, atol=1e-6, msg=lambda m, k=k: f"{k}: {m}")
