"""A frame that went wrong inside a hipGraph — NaN pose after a lost PoseMLP hand-off, truncated lists, an exchange that
unpacked nothing — must become a SKIPPED step on the device (include/riggs_hip.h: riggs_gate): the consumers of its gradients
(riggs_adam_step_gated, riggs_grad_rows_pack_gated) read the status words themselves and leave parameters, moments and step
counts bit for bit; GraphedTrainStep.check() then repairs a lost hand-off through the layered PoseMLP kernels."""
import os
import sys

import pytest
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

pytestmark = pytest.mark.gpu


def _params(seed_base=0):
    shapes = [(1001, 3), (1001, 15, 3), (1001, 1), (7,), (513, 5)] + [(k + 3, 2) for k in range(31)]  # 36 tensors: two Adam launches
    return [torch.nn.Parameter(torch.randn(s, generator=torch.Generator().manual_seed(seed_base + i)).cuda()) for i, s in enumerate(shapes)]


def _opt(params, gate=None):
    from riggs_amd.optim import FusedAdam
    o = FusedAdam([{"params": [p], "lr": 1e-3 * (1 + i % 4), "name": str(i)} for i, p in enumerate(params)], lr=0.0, eps=1e-15,
                  capturable=True)
    o.gate = gate
    return o


def _state(opt):
    out = []
    for g in opt.param_groups:
        p = g["params"][0]
        st = opt.state[p]
        out += [p.detach().clone(), st["exp_avg"].clone(), st["exp_avg_sq"].clone(), st["step"].clone()]
    return out


def test_gated_adam_equals_the_ungated_step_and_is_a_noop_when_a_word_is_raised():
    from riggs_amd import _lib as L
    words = torch.zeros(8, dtype=torch.int32, device="cuda")
    gate = L.FrameGate([lambda: (words, 2, 0xFFFFFFFF), lambda: (words, 5, 3), lambda: None])
    pa, pb = _params(), _params()
    oa, ob = _opt(pa, gate), _opt(pb, None)
    grads = [torch.randn(p.shape, generator=torch.Generator().manual_seed(100 + i)).cuda() for i, p in enumerate(pa)]
    for it in range(3):
        for p, q, g in zip(pa, pb, grads):
            p.grad, q.grad = g * (it + 1), g * (it + 1)
        oa.step()
        ob.step()
    torch.cuda.synchronize()
    for a, b in zip(_state(oa), _state(ob)):
        assert torch.equal(a, b)                      # an open gate changes nothing, bit for bit
    assert gate.read_skipped() == 0
    before = _state(oa)
    words[5] = 4                                      # outside the mask (the rasterizer's flags are bits 0 and 1): still open
    oa.step()
    torch.cuda.synchronize()
    assert gate.read_skipped() == 0 and float(oa.state[pa[0]]["step"]) == 4.0
    assert not torch.equal(pa[0].detach(), before[0])
    for raised in ((2, 1), (5, 2), (5, 1)):           # the PoseMLP's sticky word; the sort barrier's bit; the arena overflow's bit
        before = _state(oa)
        words.zero_()
        words[raised[0]] = raised[1]
        for p in pa:
            p.grad = torch.full_like(p, float("nan"))  # what a poisoned frame hands to the optimizer
        oa.step()
        oa.step()
        torch.cuda.synchronize()
        for a, b in zip(_state(oa), before):
            assert torch.equal(a, b)                  # parameters, both moments, step counts: untouched
        assert gate.read_skipped() == 2
    words.zero_()
    for p, g in zip(pa, grads):
        p.grad = g
    oa.step()
    torch.cuda.synchronize()
    assert float(oa.state[pa[0]]["step"]) == 5.0 and all(torch.isfinite(p).all() for p in pa)


def _train_step(N=20000, side=128):
    import bench
    from riggs_amd.graph import GraphedTrainStep
    from riggs_amd.optim import FusedAdam
    old = dict(bench.WORKLOAD)
    bench.WORKLOAD.update(N=N, J=24, H=side, W=side)
    try:
        sc, cam, gm, sw = bench.build_workload(0, "cuda:0")
    finally:
        bench.WORKLOAD.clear()
        bench.WORKLOAD.update(old)
    gm.training_setup(bench._train_args(), capturable=True)
    sk = FusedAdam([{"params": g["params"], "lr": 5e-4, "name": g["name"]} for g in sw.trainable_parameters()], lr=0.0, eps=1e-15,
                   capturable=True)
    gt = torch.rand(3, side, side, generator=torch.Generator().manual_seed(3)).cuda()
    gts = GraphedTrainStep(gm, sw, cam, torch.zeros(3, device="cuda"), gt, [gm.optimizer, sk], lambda_dssim=0.2)
    gts.capture()
    return gts, gm, sw, sk


def _full_state(gts):
    out = []
    for o in gts.optimizers:
        out += _state(o)
    return out


def test_a_poisoned_replay_of_the_captured_iteration_is_a_skipped_step_and_check_repairs_it():
    from riggs_amd import _lib as L
    gts, gm, sw, sk = _train_step()
    for _ in range(3):
        gts.run()
    torch.cuda.synchronize()
    gts.check()
    assert gts.skipped_steps == 0 and gts.recovered_steps == 0
    pn = sw.pose_net
    word = int(L.lib().riggs_pose_mlp_status_word(len(pn.net), pn.net[0].out_features))
    steps_before = float(gm.optimizer.state[gm._xyz]["step"])
    for bit in (1, 2):                                # the library's test hook: a hand-off lost in the forward / in the backward launch
        before = _full_state(gts)
        pn._hip_sync[word + 1] = bit
        gts.run()                                     # the poisoned replay (NaN pose, NaN gradients) ...
        pn._hip_sync[word + 1] = 0
        gts.run()                                     # ... and one more before the host looks: the word is sticky
        torch.cuda.synchronize()
        assert int(pn._hip_sync[word]) != 0
        for a, b in zip(_full_state(gts), before):
            assert torch.equal(a, b)                  # nothing reached parameters, moments or step counts
        gts.check()                                   # clears the word, re-runs the iteration through the layered kernels
        assert gts.skipped_steps == 2 and gts.recovered_steps == (1 if bit == 1 else 2)
        after = _full_state(gts)
        assert all(torch.isfinite(t).all() for t in after)
        assert not torch.equal(after[0], before[0])   # the repaired iteration did step
        assert int(pn._hip_sync[word]) == 0
    assert float(gm.optimizer.state[gm._xyz]["step"]) == steps_before + 2
    # and the graph goes on as before: same step as an uninterrupted twin would take is not checkable bit for bit (float atomics),
    # but the loss keeps falling and every replay is applied
    l0 = float(gts.run()["loss"])
    for _ in range(20):
        out = gts.run()
    torch.cuda.synchronize()
    gts.check()
    assert gts.skipped_steps == 0 and float(out["loss"]) < l0
    assert float(gm.optimizer.state[gm._xyz]["step"]) == steps_before + 2 + 21


def test_layered_pose_mlp_option_gives_the_one_launch_results():
    """riggs_set_option("pose_mlp_layered", 1): what the repair path runs — same pose, same gradients."""
    from riggs_amd import _lib as L
    from riggs_amd import synth
    from riggs_amd.skeleton import SkeletonWarp
    sc = synth.make_scene(500, 24, 5)
    torch.manual_seed(2)
    sw = SkeletonWarp(joints=sc["joints"], parent_indices=sc["parents"], K=-1, hyper_dim=8, use_skinning_weight_mlp=False,
                      use_template_offsets=False).cuda()
    x = sc["xyz"].cuda()
    t = sw.expand_time(torch.tensor([0.4], device="cuda"))

    def run():
        for p in sw.parameters():
            p.grad = None
        out = sw(x, t, motion_mask=None)
        (out["d_xyz"].square().sum() + out["d_rotation"].sum() + out["d_nodes"].sum()).backward()
        return [out["d_xyz"].detach().clone(), out["local_rotation"].detach().clone()] + [p.grad.clone() for p in sw.pose_net.parameters()]
    a = run()
    L.set_option("pose_mlp_layered", 1)
    try:
        b = run()
    finally:
        L.set_option("pose_mlp_layered", 0)
    for u, v in zip(a, b):
        assert float((u - v).abs().max()) <= 1e-5 * max(1e-12, float(u.abs().max()))


def test_a_gated_pack_marks_the_segment_and_every_rank_skips_the_unpack():
    import bench
    from riggs_amd import _lib as L
    from riggs_amd.dist import FlatGradAllReduce, SparseRowExchange, row_exchange_order
    from riggs_amd.rasterizer import RasterArena
    N, side = 6001, 96
    old = dict(bench.WORKLOAD)
    bench.WORKLOAD.update(N=N, J=8, H=side, W=side)
    try:
        sc, cam, gm, sw = bench.build_workload(0, "cuda:0")
    finally:
        bench.WORKLOAD.clear()
        bench.WORKLOAD.update(old)
    ordered, n_rows = row_exchange_order(gm, sw)
    gimg = torch.rand(3, side, side, generator=torch.Generator().manual_seed(1)).cuda()
    bench.make_step(cam, gm, sw, gimg, RasterArena(), 1, None)()
    grads = [p.grad.detach().clone().reshape(N, -1) for p in ordered[:n_rows]]
    words = torch.zeros(4, dtype=torch.int32, device="cuda")
    good = SparseRowExchange(grads, capacity=N, world=2)
    good.gate = L.FrameGate([lambda: (words, 0, 0xFFFFFFFF)])
    good.pack()                                       # gate open: the plain segment
    plain = SparseRowExchange(grads, capacity=N, world=2)
    plain.pack()
    torch.cuda.synchronize()
    assert torch.equal(good.segment, plain.segment) and 0 < int(good.segment[1]) < N
    bad = SparseRowExchange(grads, capacity=N, world=2)
    bad.gate = good.gate
    words[0] = 1
    bad.pack()                                        # this rank's frame was invalid
    torch.cuda.synchronize()
    assert int(bad.segment[1]) == -1 and int(bad.segment[0]) == 0 and int(bad.segment[2]) == N
    # the other rank's unpack sees {its own good segment, the invalid one}: nothing is unpacked, the status says why
    local = [g.clone() for g in grads]
    ex = SparseRowExchange(local, capacity=N, world=2)
    ex.gathered.copy_(torch.cat([good.segment, bad.segment]))
    ex._unpack(ex)
    torch.cuda.synchronize()
    for a, b in zip(local, grads):
        assert torch.equal(a, b)
    src = ex.status_source()
    assert int(src[0][src[1]]) & 2                    # the word every rank's optimizer is gated on
    # ... for THIS step only: a following good step lowers it again (its optimizers run), the sticky report stays for the host
    ex.gathered.copy_(torch.cat([good.segment, good.segment]))
    ex._unpack(ex)
    torch.cuda.synchronize()
    assert int(src[0][src[1]]) == 0 and int(ex.status[1]) & 2
    for a, b in zip(local, grads):
        a.copy_(b)
    assert not ex.check() and ex.invalid_frame and ex.need == int(good.segment[1])   # (the marker does not count as a need)
    assert ex.check() and not ex.invalid_frame        # reading cleared it
    # the dense path: the validity slot of the bucket travels with the all-reduce
    bucket = FlatGradAllReduce([torch.nn.Parameter(torch.zeros(10, 3, device="cuda")), torch.nn.Parameter(torch.zeros(5, device="cuda"))],
                               register=False)
    assert bucket.flat.numel() == bucket.numel + 4 and bucket.tail.numel() == 4
    bucket.frame_gate = good.gate
    bucket.publish_validity()
    t, i, m = bucket.validity_source()
    assert float(bucket.tail[0]) == 1.0 and (int(t.view(torch.int32)[i]) & m) != 0
    words[0] = 0
    bucket.publish_validity()
    assert float(bucket.tail[0]) == 0.0 and (int(t.view(torch.int32)[i]) & m) == 0


def test_an_eager_trainer_survives_a_poisoned_frame_and_is_told():
    """An unmodified train_rig.py issues everything eagerly and never looks at a status word.  A frame poisoned by a lost PoseMLP
    hand-off (the library's fault hook) must not reach its parameters: the eager FusedAdam leaves every element whose gradient is
    not finite untouched (riggs_adam_step_guarded) and says so; the skeleton's non-blocking watcher (PoseMLP.watch) reports the
    time-out, clears the sticky word and — from the third one on — switches to the layered PoseMLP kernels."""
    import warnings

    import bench
    from riggs_amd import _lib as L
    from riggs_amd.optim import FusedAdam
    from riggs_amd.render import render
    old = dict(bench.WORKLOAD)
    bench.WORKLOAD.update(N=5000, J=8, H=64, W=64)
    try:
        sc, cam, gm, sw = bench.build_workload(0, "cuda:0")
    finally:
        bench.WORKLOAD.clear()
        bench.WORKLOAD.update(old)
    gm.training_setup(bench._train_args())                      # the reference's call: an eager (non-capturable) FusedAdam
    sk = FusedAdam([{"params": g["params"], "lr": 5e-4, "name": g["name"]} for g in sw.trainable_parameters()], lr=0.0, eps=1e-15)
    assert gm.optimizer.skip_nonfinite and not gm.optimizer.hip_capturable
    pn = sw.pose_net
    word = int(L.lib().riggs_pose_mlp_status_word(len(pn.net), pn.net[0].out_features))
    bg = torch.zeros(3, device="cuda")
    target = torch.rand(3, 64, 64, generator=torch.Generator().manual_seed(1)).cuda()

    def iteration():
        for o in (gm.optimizer, sk):
            o.zero_grad(set_to_none=True)
        d = sw(gm.get_xyz, sw.expand_time(cam.fid), motion_mask=gm.motion_mask)
        out = render(cam, gm, bench.Pipe, bg, d["d_xyz"], d["d_rotation"], d["d_scaling"])
        ((out["render"] - target).abs().mean() + 1e-3 * d["d_nodes"].square().sum()).backward()
        gm.optimizer.step()
        sk.step()
    params = [p for o in (gm.optimizer, sk) for g in o.param_groups for p in g["params"]]
    with warnings.catch_warnings(record=True) as seen:
        warnings.simplefilter("always")
        for _ in range(3):
            iteration()
        torch.cuda.synchronize()
        ours = lambda: [str(w.message) for w in seen if "hand-off" in str(w.message) or "NaN or Inf" in str(w.message)]  # noqa: E731
        assert all(torch.isfinite(p).all() for p in params) and not ours()
        try:
            for k in range(3):                                      # three poisoned frames, healthy ones in between
                pn._hip_sync[word + 1] = 1                          # the fault hook: this launch loses a hand-off
                iteration()
                pn._hip_sync[word + 1] = 0
                torch.cuda.synchronize()
                assert int(pn._hip_sync[word]) != 0 or pn.handoff_timeouts > k   # the sticky word is up (or already reported)
                assert all(torch.isfinite(p).all() for p in params), "a poisoned frame reached the parameters"
                for o in (gm.optimizer, sk):
                    for st in o.state.values():
                        assert torch.isfinite(st["exp_avg"]).all() and torch.isfinite(st["exp_avg_sq"]).all()
                for _ in range(40):                                 # the watcher looks every 16th call, without blocking
                    iteration()
                torch.cuda.synchronize()
                assert pn.handoff_timeouts == k + 1 and int(pn._hip_sync[word]) == 0
            texts = ours()
            assert sum("hand-off" in t for t in texts) == 3 and any("one-launch-per-layer" in t for t in texts)
            assert any("NaN or Inf" in t for t in texts) and sk.nonfinite_seen > 0
        finally:
            L.set_option("pose_mlp_layered", 0)
    # torch.optim.Adam's behaviour on request
    p = torch.nn.Parameter(torch.ones(8, device="cuda"))
    o = FusedAdam([p], lr=1e-2, skip_nonfinite=False)
    p.grad = torch.full_like(p, float("nan"))
    o.step()
    assert torch.isnan(p).all()
    q = torch.nn.Parameter(torch.ones(8, device="cuda"))
    o = FusedAdam([q], lr=1e-2)
    q.grad = torch.tensor([float("nan"), 1.0, float("inf"), 1.0, 1.0, 1.0, 1.0, -float("inf")], device="cuda")
    o.step()
    torch.cuda.synchronize()
    want = torch.ones(8) - 1e-2
    want[[0, 2, 7]] = 1.0
    assert torch.allclose(q.detach().cpu(), want, atol=1e-6) and int(o._nonfinite_count) == 3


def test_eager_guard_raises_when_the_gradients_stay_nonfinite():
    """The eager FusedAdam's guard rides out single poisoned frames; gradients that KEEP coming back NaN are a diverged run,
    which torch.optim.Adam would have shown as NaN parameters: after a few consecutive watch windows it raises."""
    import warnings

    from riggs_amd import optim as O
    q = torch.nn.Parameter(torch.ones(64, device="cuda"))
    o = O.FusedAdam([q], lr=1e-2)
    raised = None
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        for it in range(40 * O.NONFINITE_WINDOWS_LIMIT):
            q.grad = torch.full_like(q, float("nan"))
            try:
                o.step()
            except RuntimeError as e:
                raised = (it, str(e))
                break
            torch.cuda.synchronize()
    assert raised is not None and "diverged" in raised[1] and raised[0] <= 16 * (O.NONFINITE_WINDOWS_LIMIT + 2)
    assert torch.equal(q.detach().cpu(), torch.ones(64))
    # a transient fault does not: one poisoned step, then healthy ones
    p = torch.nn.Parameter(torch.ones(64, device="cuda"))
    o = O.FusedAdam([p], lr=1e-3)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        for it in range(200):
            p.grad = torch.full_like(p, float("nan") if it % 50 == 3 else 1.0)
            o.step()
            torch.cuda.synchronize()
    assert torch.isfinite(p).all()
