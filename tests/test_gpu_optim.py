"""§8-f rank 1 on the GPU: the one-launch Adam step and the densification statistics against the golden vectors of
the reference's own optimizer, the CPU oracle, and torch.optim.Adam itself (the optimizer the reference constructs)."""
import os
from types import SimpleNamespace

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

G = np.load(os.path.join(os.path.dirname(__file__), "golden", "optim_adam_n67.npz"))
NAMES = [str(n) for n in G["names"]]
ARGS = SimpleNamespace(percent_dense=0.01, position_lr_init=0.00016, position_lr_final=0.0000016,
                       position_lr_delay_mult=0.01, position_lr_max_steps=30000, feature_lr=0.0025, opacity_lr=0.05,
                       scaling_lr=0.001, rotation_lr=0.001, skeleton_gs_position_lr=0.00001)


def _close(got, want, what):
    want = np.asarray(want)
    np.testing.assert_allclose(got.detach().cpu().numpy(), want, rtol=2e-6, atol=2e-7 * float(np.abs(want).max()), err_msg=what)


def _model():
    from riggs_amd.gaussian_model import GaussianModel
    t = lambda n: torch.from_numpy(G["p0_" + n])  # noqa: E731
    gm = GaussianModel.from_tensors(t("xyz"), t("f_dc"), t("f_rest"), t("scaling"), t("rotation"), t("opacity"))
    gm.training_setup(ARGS)
    return gm


def test_training_setup_step_and_schedule_match_reference_golden():
    gm = _model()
    assert [g["name"] for g in gm.optimizer.param_groups] == NAMES
    for it in range(int(G["steps"])):
        np.testing.assert_allclose([g["lr"] for g in gm.optimizer.param_groups], G["lr%d" % it], rtol=1e-12)
        for grp in gm.optimizer.param_groups:
            grp["params"][0].grad = torch.from_numpy(G["g%d_%s" % (it, grp["name"])]).cuda()
        gm.optimizer.step()
        gm.update_learning_rate(1000 * (it + 1))
        gm.optimizer.zero_grad(set_to_none=True)
        for grp in gm.optimizer.param_groups:
            n, st = grp["name"], gm.optimizer.state[grp["params"][0]]
            _close(grp["params"][0], G["p%d_%s" % (it + 1, n)], "param " + n)
            _close(st["exp_avg"], G["m%d_%s" % (it + 1, n)], "exp_avg " + n)
            _close(st["exp_avg_sq"], G["v%d_%s" % (it + 1, n)], "exp_avg_sq " + n)
            assert float(st["step"]) == it + 1


def test_matches_torch_adam_ragged_sizes_many_groups_and_surgery():
    from riggs_amd.optim import FusedAdam
    g = torch.Generator().manual_seed(5)
    shapes = [(1001, 3), (1001, 1, 3), (1001, 15, 3), (1001, 1), (1001, 3), (1001, 4), (7,), (1, 1), (513, 5), (2, 3)]
    shapes += [(k + 3, 2) for k in range(27)]  # 37 tensors > 32 per launch: two launches
    lrs = [8e-4, 2.5e-3, 1.25e-4, 5e-2, 5e-3, 1e-3, 1e-2, 1e-1, 3e-3, 1e-3] + [1e-3 * (1 + k % 5) for k in range(27)]
    mk = lambda: [torch.nn.Parameter(torch.randn(s, generator=torch.Generator().manual_seed(i)).cuda()) for i, s in enumerate(shapes)]  # noqa: E731
    pa, pb = mk(), mk()
    oa = FusedAdam([{"params": [p], "lr": lr, "name": str(i)} for i, (p, lr) in enumerate(zip(pa, lrs))], lr=0.0, eps=1e-15)
    ob = torch.optim.Adam([{"params": [p], "lr": lr, "name": str(i)} for i, (p, lr) in enumerate(zip(pb, lrs))], lr=0.0, eps=1e-15)

    def check():
        for ga, gb in zip(oa.param_groups, ob.param_groups):
            a, b = ga["params"][0], gb["params"][0]
            tol = dict(rtol=3e-6, atol=3e-7 * float(b.detach().abs().max()))
            torch.testing.assert_close(a, b, **tol)
            torch.testing.assert_close(oa.state[a]["exp_avg"], ob.state[b]["exp_avg"], rtol=3e-6, atol=3e-7 * float(ob.state[b]["exp_avg"].abs().max()) + 1e-30)
            torch.testing.assert_close(oa.state[a]["exp_avg_sq"], ob.state[b]["exp_avg_sq"], rtol=3e-6, atol=1e-30)

    def step(k):
        for ga, gb in zip(oa.param_groups, ob.param_groups):
            a, b = ga["params"][0], gb["params"][0]
            gr = torch.randn(a.shape, generator=g).cuda() * (10.0 ** float(torch.randint(-6, 3, (1,), generator=g)))
            if k % 2 and a.dim() > 1:
                gr[::3] = 0  # invisible Gaussians: exact zeros
            a.grad, b.grad = gr.clone(), gr.clone()
        oa.step(), ob.step()
        oa.zero_grad(set_to_none=True), ob.zero_grad(set_to_none=True)

    for k in range(3):
        step(k)
        check()
    # the reference's pruning surgery (scene/gaussian_model.py:356-371) applied to both optimizers, then more steps
    for opt in (oa, ob):
        for group in opt.param_groups[:6]:
            old = group["params"][0]
            mask = torch.arange(old.shape[0], device=old.device) % 5 != 0
            st = opt.state.get(old)
            st["exp_avg"], st["exp_avg_sq"] = st["exp_avg"][mask], st["exp_avg_sq"][mask]
            del opt.state[old]
            group["params"][0] = torch.nn.Parameter(old[mask].requires_grad_(True))
            opt.state[group["params"][0]] = st
    for k in range(2):
        step(k)
        check()
    sd = oa.state_dict()  # same layout as torch.optim.Adam's
    assert set(sd["state"][0]) == {"step", "exp_avg", "exp_avg_sq"} and sd["param_groups"][0]["eps"] == 1e-15


def test_rejects_configurations_the_reference_does_not_use():
    from riggs_amd.optim import FusedAdam
    p = [torch.nn.Parameter(torch.zeros(4).cuda())]
    with pytest.raises(NotImplementedError):
        FusedAdam(p, weight_decay=0.1)
    with pytest.raises(NotImplementedError):
        FusedAdam(p, amsgrad=True)
    o = FusedAdam([torch.nn.Parameter(torch.zeros(4))], lr=1e-3)
    o.param_groups[0]["params"][0].grad = torch.ones(4)
    with pytest.raises(RuntimeError):  # CPU tensors: the product path is GPU-only, no silent fallback
        o.step()


def test_densification_stats_match_reference_golden_and_oracle():
    from oracle import optim_ref as O
    gm = _model()
    N = int(G["N"])
    for it in range(2):
        vt = SimpleNamespace(grad=torch.from_numpy(G["ds_grad%d" % it]).cuda())
        gm.add_densification_stats(vt, torch.from_numpy(G["ds_filter%d" % it]).cuda())
        np.testing.assert_allclose(gm.xyz_gradient_accum.cpu().numpy(), G["ds_accum%d" % it], rtol=1e-6)
        np.testing.assert_array_equal(gm.denom.cpu().numpy(), G["ds_denom%d" % it])
    # with the max_radii2D update of train_rig.py:333-335, against the oracle, ragged N
    g = torch.Generator().manual_seed(9)
    N = 100003
    vg = (torch.randn(N, 3, generator=g) * 1e-3).cuda()
    filt = (torch.rand(N, generator=g) > 0.5).cuda()
    radii = torch.randint(0, 60, (N,), generator=g, dtype=torch.int32).cuda()
    acc, den = torch.rand(N, 1, generator=g).cuda(), torch.randint(0, 9, (N, 1), generator=g).float().cuda()
    mr = (torch.rand(N, generator=g) * 50).cuda()
    want = O.densification_stats(vg.cpu().numpy(), filt.cpu().numpy(), acc.cpu().numpy(), den.cpu().numpy(), radii.cpu().numpy(), mr.cpu().numpy())
    from riggs_amd.optim import densify_stats
    densify_stats(vg, filt, acc, den, radii, mr)
    np.testing.assert_allclose(acc.cpu().numpy(), want[0], rtol=1e-6)
    np.testing.assert_array_equal(den.cpu().numpy(), want[1])
    np.testing.assert_array_equal(mr.cpu().numpy(), want[2])


def test_capturable_adam_replays_in_a_hipgraph_like_eager_torch_adam():
    """capturable=True: step counts (and a scheduled lr) in device memory, so ONE captured step replays correctly."""
    from riggs_amd.optim import FusedAdam
    g = torch.Generator().manual_seed(11)
    shapes = [(1001, 3), (1001, 15, 3), (513,)]
    mk = lambda: [torch.nn.Parameter(torch.randn(s, generator=torch.Generator().manual_seed(i)).cuda()) for i, s in enumerate(shapes)]  # noqa: E731
    pa, pb = mk(), mk()
    lr_t = torch.tensor(8e-4, device="cuda")
    oa = FusedAdam([{"params": [pa[0]], "lr": lr_t}, {"params": [pa[1]], "lr": 2e-3}, {"params": [pa[2]], "lr": 1e-2}],
                   lr=0.0, eps=1e-15, capturable=True)
    ob = torch.optim.Adam([{"params": [pb[0]], "lr": 8e-4}, {"params": [pb[1]], "lr": 2e-3}, {"params": [pb[2]], "lr": 1e-2}],
                          lr=0.0, eps=1e-15)
    grads = [torch.zeros_like(p) for p in pa]  # static gradient buffers, refilled before every replay
    for p, gr in zip(pa, grads):
        p.grad = gr

    def fill(k):
        out = []
        for gr in grads:
            v = torch.randn(gr.shape, generator=g).cuda() * (0.1 + k)
            gr.copy_(v)
            out.append(v)
        return out
    s = torch.cuda.Stream()
    vals = fill(0)
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        oa.step()                                   # warm-up (allocates state), also step 1
    torch.cuda.current_stream().wait_stream(s)
    for p, v in zip(pb, vals):
        p.grad = v.clone()
    ob.step()
    graph = torch.cuda.CUDAGraph()
    vals = fill(1)
    with torch.cuda.graph(graph, stream=s):
        oa.step()                                   # captured (capture records, it does not execute) ...
    graph.replay()                                  # ... step 2
    for p, v in zip(pb, vals):
        p.grad = v.clone()
    ob.step()
    for k in range(2, 5):                           # replays: steps 3..5, with a learning-rate schedule on group 0
        lr = 8e-4 * (0.5 ** k)
        lr_t.fill_(lr)
        ob.param_groups[0]["lr"] = lr
        vals = fill(k)
        graph.replay()
        for p, v in zip(pb, vals):
            p.grad = v.clone()
        ob.step()
    torch.cuda.synchronize()
    for a, b in zip(pa, pb):
        torch.testing.assert_close(a.detach(), b.detach(), rtol=5e-6, atol=5e-7 * float(b.detach().abs().max()))
        assert float(oa.state[a]["step"]) == 5.0 and oa.state[a]["step"].is_cuda


@pytest.mark.parametrize("capturable", [False, True])
def test_step_many_equals_stepping_each_optimizer(capturable):
    """``optim.step_many`` (one step-count increment and one Adam launch per (betas, eps) across the optimizers — what a
    captured train step uses for the Gaussians' and the skeleton's optimizers) gives bit for bit what ``o.step()`` per
    optimizer gives, lazy state initialisation and a different configuration (its own launch) included."""
    from riggs_amd.optim import FusedAdam, step_many
    g = torch.Generator().manual_seed(7)
    shapes = [(301, 3), (17,), (64, 8), (5, 5, 5)]

    def make():
        ps = [torch.randn(*s, generator=torch.Generator().manual_seed(11 + i)).cuda().requires_grad_(True) for i, s in enumerate(shapes)]
        a = FusedAdam([{"params": [ps[0]], "lr": 1e-3}, {"params": [ps[1]], "lr": 5e-3}], lr=0.0, eps=1e-15, capturable=capturable)
        b = FusedAdam([{"params": [ps[2]], "lr": 2e-3}], lr=0.0, eps=1e-15, capturable=capturable)
        c = FusedAdam([{"params": [ps[3]], "lr": 1e-2}], lr=0.0, eps=1e-8, betas=(0.8, 0.99), capturable=capturable)
        return ps, [a, b, c]
    (p1, o1), (p2, o2) = make(), make()
    for it in range(3):
        grads = [torch.randn(*s, generator=g).cuda() for s in shapes]
        for p, q, gr in zip(p1, p2, grads):
            p.grad, q.grad = gr.clone(), gr.clone()
        for o in o1:
            o.step()
        step_many(o2)
        for p, q in zip(p1, p2):
            assert torch.equal(p, q), it
    for oa, ob in zip(o1, o2):
        for sa, sb in zip(oa.state.values(), ob.state.values()):
            assert float(sa["step"]) == float(sb["step"]) == 3.0
            assert torch.equal(sa["exp_avg"], sb["exp_avg"]) and torch.equal(sa["exp_avg_sq"], sb["exp_avg_sq"])


def test_step_many_honours_step_hooks_and_lr_schedulers():
    """``step_many`` merges the optimizers' launches only while nothing hangs on their ``step()``: with a registered step
    hook or an lr_scheduler wrapped around one of them it steps each optimizer through its own ``step()`` (hooks fire, the
    scheduler's call counter moves, no "scheduler.step() before optimizer.step()" warning) — same parameters either way."""
    import warnings
    from riggs_amd.optim import FusedAdam, step_many

    def make():
        ps = [torch.randn(33, 3, generator=torch.Generator().manual_seed(3 + i)).cuda().requires_grad_(True) for i in range(2)]
        return ps, [FusedAdam([{"params": [p], "lr": 1e-2}], lr=0.0, eps=1e-15) for p in ps]
    (p1, o1), (p2, o2) = make(), make()
    fired = []
    o2[0].register_step_post_hook(lambda opt, args, kwargs: fired.append("post"))
    sched = torch.optim.lr_scheduler.StepLR(o2[1], step_size=1, gamma=0.5)
    g = torch.Generator().manual_seed(5)
    with warnings.catch_warnings():
        warnings.filterwarnings("error", message=".*lr_scheduler.step.*")  # the scheduler's ordering warning would fail the test
        for it in range(2):
            grads = [torch.randn(33, 3, generator=g).cuda() for _ in range(2)]
            for p, q, gr in zip(p1, p2, grads):
                p.grad, q.grad = gr.clone(), gr.clone()
            if it == 1:
                o1[1].param_groups[0]["lr"] = 5e-3  # what the scheduler did to its twin
            step_many(o1)
            step_many(o2)
            sched.step()
    assert fired == ["post", "post"]
    assert torch.equal(p1[0], p2[0]) and torch.equal(p1[1], p2[1])
    assert o2[1].param_groups[0]["lr"] == pytest.approx(2.5e-3)


@pytest.mark.parametrize("capturable", [False, True])
def test_cached_launch_plan_follows_everything_that_can_change_between_steps(capturable):
    """FusedAdam keeps its marshalled launch arguments across steps (riggs_amd.optim._Plan).  Whatever changes between two steps —
    new gradient tensors, a gradient missing, a non-contiguous gradient, a learning rate, ``p.data`` re-pointed, a moment tensor
    replaced, ``load_state_dict``, the reference's optimizer surgery (new parameter objects), an optimizer stepped alone after a
    merged step — the update equals torch.optim.Adam's on a twin, step for step."""
    from riggs_amd.optim import FusedAdam, step_many
    gen = torch.Generator().manual_seed(5)
    shapes = [(301, 3), (301, 1, 3), (301, 15, 3), (301, 1), (7,), (64, 5)]

    def make(cls, **kw):
        ps = [torch.nn.Parameter(torch.randn(s, generator=torch.Generator().manual_seed(i)).cuda()) for i, s in enumerate(shapes)]
        return ps, cls([{"params": [p], "lr": 1e-3 * (i + 1), "name": str(i)} for i, p in enumerate(ps[:4])], lr=0.0, eps=1e-15, **kw), \
            cls([{"params": ps[4:], "lr": 5e-4, "name": "rest"}], lr=0.0, eps=1e-15, **kw)
    pa, a1, a2 = make(FusedAdam, capturable=capturable)
    pb, b1, b2 = make(torch.optim.Adam)

    def grads(skip=(), strided=()):
        for i, (p, q) in enumerate(zip(pa, pb)):
            if i in skip:
                p.grad = q.grad = None
                continue
            g = torch.randn(p.shape, generator=gen).cuda()
            if i in strided:  # a non-contiguous view of the same values
                g2 = torch.empty(p.shape + (2,), device="cuda")[..., 0]
                g2.copy_(g)
                p.grad, q.grad = g2, g.clone()
            else:
                p.grad, q.grad = g, g.clone()

    def both(merged=True):
        if merged:
            step_many([a1, a2])
        else:
            a1.step()
            a2.step()
        b1.step()
        b2.step()
        for i, (p, q) in enumerate(zip(pa, pb)):
            assert float((p.detach() - q.detach()).abs().max()) <= 2e-6 * float(q.detach().abs().max()) + 1e-9, i
    for _ in range(3):
        grads()
        both()                                             # same plan, new gradient objects
    assert a1._hip_plan is not None
    plan = a1._hip_plan
    grads()
    both()
    assert a1._hip_plan is plan                            # ... reused, not rebuilt
    grads(skip=(2, 5))
    both()                                                 # two parameters without a gradient this step
    grads(strided=(1,))
    both()                                                 # a gradient that has to be made contiguous
    grads()
    for o in (a1, b1):
        o.param_groups[1]["lr"] = 7e-3                      # update_learning_rate between steps
    both()
    with torch.no_grad():                                  # p.data re-pointed (same object, new storage)
        pa[0].data = pa[0].data.clone()
    grads()
    both()
    st = a1.state[pa[3]]
    st["exp_avg"] = st["exp_avg"].clone()                  # a moment tensor replaced behind the same parameter
    grads()
    both()
    sd_a, sd_b = a1.state_dict(), b1.state_dict()
    a1.load_state_dict(sd_a)
    b1.load_state_dict(sd_b)
    grads()
    both()
    both(merged=False)                                     # the members stepped alone after merged steps (same gradients again)
    grads()
    both()
    # the reference's replace_tensor_to_optimizer (scene/gaussian_model.py:338-351): a new, longer parameter object in the group
    for o, ps in ((a1, pa), (b1, pb)):
        grp = o.param_groups[0]
        old = grp["params"][0]
        stored = o.state.pop(old)
        new = torch.nn.Parameter(torch.cat([old.detach(), old.detach()[:10]]).requires_grad_(True))
        stored["exp_avg"] = torch.cat([stored["exp_avg"], torch.zeros_like(stored["exp_avg"][:10])])
        stored["exp_avg_sq"] = torch.cat([stored["exp_avg_sq"], torch.zeros_like(stored["exp_avg_sq"][:10])])
        grp["params"][0] = new
        o.state[new] = stored
        ps[0] = new
    grads()
    both()
    grads()
    both()
