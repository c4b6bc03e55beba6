"""GPU: the gradient-row exchange kernels (csrc/exchange.hip through riggs_grad_rows_pack / _unpack) against the numpy
restatement of the segment format (tests/rows_ref.py), on gradients of real frames: the row list comes from the
rasterizer's backward itself."""
import os
import sys

import numpy as np
import pytest
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import rows_ref  # noqa: E402
from riggs_amd import synth  # noqa: E402

pytestmark = pytest.mark.gpu

N, J, H, W = 6001, 8, 96, 112   # 24 blocks of 256, the last one ragged
WORLD = 3


def _frames():
    """WORLD frames of one scene from different cameras: per frame the six per-Gaussian gradients (dL/d _features_dc,
    _features_rest, _opacity, _scaling, _xyz, _rotation) and the packed segment the HIP kernel made right behind the backward."""
    import bench
    from riggs_amd.dist import SparseRowExchange, row_exchange_order
    from riggs_amd.rasterizer import RasterArena
    old = dict(bench.WORKLOAD)
    bench.WORKLOAD.update(N=N, J=J, H=H, W=W)
    try:
        sc, cam0, gm, sw = bench.build_workload(0, "cuda:0")
    finally:
        bench.WORKLOAD.clear()
        bench.WORKLOAD.update(old)
    ordered, n_rows = row_exchange_order(gm, sw)
    gimg = torch.rand(3, H, W, generator=torch.Generator().manual_seed(1)).cuda()
    frames = []
    for r in range(WORLD):
        cam = synth.look_at_camera(H, W, azimuth_deg=20.0 + 110.0 * r, fid=0.2 + 0.3 * r).to("cuda:0")
        step = bench.make_step(cam, gm, sw, gimg, RasterArena(), 1, None)
        step()
        grads = [p.grad.detach().clone().reshape(N, -1) for p in ordered[:n_rows]]
        ex = SparseRowExchange(grads, capacity=N, world=WORLD)
        ex.pack()                                                # reads the row list the backward just left in its workspace
        torch.cuda.synchronize()
        frames.append((grads, ex.segment.clone(), ex))
    return frames


@pytest.fixture(scope="module")
def frames():
    return _frames()


def test_pack_lists_exactly_the_rows_with_a_gradient_in_order(frames):
    for grads, seg, ex in frames:
        seg = seg.cpu().numpy()
        nz = np.zeros(N, bool)
        for g in grads:
            nz |= (g.cpu().numpy() != 0).any(1)
        need = int(seg[1])
        assert seg[0] == need and seg[2] == N and seg[3] == 60 and 0 < need < N
        ro = rows_ref.rows_offset(N)
        rows = seg[ro:ro + need * 60].view(np.float32).reshape(need, 60)
        idx = rows[:, 0].copy().view(np.int32)
        assert np.all(np.diff(idx) > 0) and idx.min() >= 0 and idx.max() < N          # ascending, unique
        listed = np.zeros(N, bool)
        listed[idx] = True
        assert not (nz & ~listed).any()                          # every row with a non-zero gradient is listed
        assert (listed & ~nz).sum() <= 0.02 * need               # (a listed row may have rounded to zero everywhere)
        # the whole segment, bit for bit, is what the numpy restatement builds from the same list
        want = rows_ref.pack([g.cpu().numpy() for g in grads], listed, 1.0 / WORLD, N, 60)
        assert np.array_equal(seg, want)
        nb = (N + 255) // 256
        assert seg[4 + nb] == need and np.array_equal(seg[4:4 + nb + 1], np.concatenate([[0], np.cumsum(np.bincount(idx // 256, minlength=nb))]))


def test_unpack_sums_in_rank_order_bit_exactly(frames):
    from riggs_amd.dist import SparseRowExchange
    local = [g.clone() for g in frames[1][0]]                    # "this rank" is rank 1
    ex = SparseRowExchange(local, capacity=N, world=WORLD)
    ex.gathered.copy_(torch.cat([f[1] for f in frames]))
    ex._unpack(ex)
    torch.cuda.synchronize()
    assert ex.check() and ex.need == max(int(f[1][1]) for f in frames)
    want = [g.cpu().numpy().copy() for g in frames[1][0]]
    rows_ref.unpack(want, torch.stack([f[1] for f in frames]).cpu().numpy(), N, 60)
    for a, b in zip(local, want):
        assert np.array_equal(a.cpu().numpy(), b)               # same additions in the same order: the same bits
    # and it is the mean of the three frames' gradients
    for k, a in enumerate(local):
        mean = sum(f[0][k].double() for f in frames) / WORLD
        torch.testing.assert_close(a.double(), mean, rtol=1e-6, atol=1e-7 * float(mean.abs().max()))


def test_unpack_skips_everything_when_a_segment_overflowed(frames):
    from riggs_amd.dist import SparseRowExchange
    need = [int(f[1][1]) for f in frames]
    cap = max(need) - 7                                          # the fullest rank does not fit
    local = [g.clone() for g in frames[0][0]]
    ex = SparseRowExchange(local, capacity=cap, world=WORLD)
    segs = []
    for grads, _, _ in frames:
        listed = np.zeros(N, bool)
        for g in grads:
            listed |= (g.cpu().numpy() != 0).any(1)
        segs.append(torch.from_numpy(rows_ref.pack([g.cpu().numpy() for g in grads], listed, 1.0 / WORLD, cap, 60)))
    ex.gathered.copy_(torch.cat(segs))
    ex._unpack(ex)
    torch.cuda.synchronize()
    assert not ex.check() and ex.need >= max(need) - 120        # (need counts listed rows; the numpy list is the non-zero ones)
    for a, b in zip(local, frames[0][0]):
        assert torch.equal(a, b)
    # the status is STICKY: an overflow on step k is still reported when the caller polls on step k + n, together with the step
    # that failed first; reading clears it
    assert ex.check() and ex.calls == 0
    ex._unpack(ex)                                               # step 1: overflows (the same segments)
    ok = SparseRowExchange([g.clone() for g in frames[0][0]], capacity=N, world=WORLD)
    ok.status = ex.status                                        # (same status words: the steps of one exchange)
    ok.gathered.copy_(torch.cat([f[1] for f in frames]))
    ok._unpack(ok)                                               # steps 2, 3: fit
    ok._unpack(ok)
    torch.cuda.synchronize()
    assert not ex.check() and ex.calls == 3 and ex.overflow_call == 1 and ex.need >= max(need) - 120
    assert ex.check() and ex.calls == 0 and ex.overflow_call == 0


def test_pack_into_a_small_capacity_reports_the_need_and_keeps_the_first_rows(frames):
    from riggs_amd.dist import SparseRowExchange
    grads, seg_full, ex_full = frames[2]
    # the workspace still describes the LAST backward (frame 2)
    ex = SparseRowExchange(grads, capacity=100, world=WORLD)
    ex.pack()
    torch.cuda.synchronize()
    seg, full = ex.segment.cpu().numpy(), seg_full.cpu().numpy()
    assert seg[0] == 100 and seg[1] == full[1]
    ro = rows_ref.rows_offset(N)
    assert np.array_equal(seg[ro:ro + 100 * 60], full[ro:ro + 100 * 60])


def test_exchange_rejects_cpu_tensors_and_foreign_workspaces(frames):
    from riggs_amd.dist import SparseRowExchange
    with pytest.raises(RuntimeError):
        SparseRowExchange([torch.zeros(10, 3)], capacity=4)
    ex = SparseRowExchange([torch.zeros(N + 5, 3, device="cuda")], capacity=4, world=2)
    with pytest.raises(RuntimeError):
        ex.pack()                                                # the last backward was over N Gaussians, not N + 5


def test_row_exchange_around_a_split_captured_frame_is_the_identity_at_world_one():
    """The whole host path on one GPU: a frame captured as two graphs with the pack INSIDE graph (a)
    (GraphedFrame.after_raster_backward), gather (a copy at world 1), small all-reduce (nothing at world 1), ordered unpack —
    the gradients come back bit for bit (scale 1, every row occurs once), replay after replay, and rows without a gradient
    stay exactly zero."""
    import bench
    from riggs_amd.dist import FlatGradAllReduce, SparseRowExchange, row_exchange_order
    from riggs_amd.graph import GraphedFrame
    old = dict(bench.WORKLOAD)
    bench.WORKLOAD.update(N=N, J=J, H=H, W=W)
    try:
        sc, cam, gm, sw = bench.build_workload(0, "cuda:0")
    finally:
        bench.WORKLOAD.clear()
        bench.WORKLOAD.update(old)
    ordered, n_rows = row_exchange_order(gm, sw)
    bucket = FlatGradAllReduce(ordered)
    try:
        rows = SparseRowExchange([v.view(N, -1) for v in bucket.views[:n_rows]], rest=bucket.flat[bucket.offsets[n_rows]:], capacity=N)
        gf = GraphedFrame(gm, sw, cam, torch.zeros(3, device="cuda"), bench.params_of(gm, sw), split_backward=True)
        gf.after_raster_backward = rows.pack
        gf.capture()
        with pytest.raises(RuntimeError, match="captured"):      # the captured pack has the old segment baked in
            rows.resize(100)
        gf.set_inputs(gimg=torch.rand(3, H, W, generator=torch.Generator().manual_seed(3)).cuda())
        for cam_az in (45.0, 160.0):
            gf.run_a(cam=synth.look_at_camera(H, W, azimuth_deg=cam_az).to("cuda:0"))
            rows.launch()
            gf.run_b()
            torch.cuda.synchronize()
            before = [g.clone() for g in rows.rows]
            rest_before = rows.rest.clone()
            rows.launch_rest()
            rows.wait()
            torch.cuda.synchronize()
            assert rows.check() and 0 < rows.need < N
            for a, b in zip(rows.rows, before):
                assert torch.equal(a, b)
            assert torch.equal(rows.rest, rest_before)
            touched = torch.zeros(N, dtype=torch.bool, device="cuda")
            for g in before:
                touched |= (g != 0).any(1)
            assert int(touched.sum()) <= rows.need <= int(touched.sum()) + 0.02 * N
            assert all(p.grad.data_ptr() == v.data_ptr() for p, v in zip(bucket.params, bucket.views))
    finally:
        bucket.unregister()


def test_sharded_adam_hip_step_equals_torch_adam_on_one_rank():
    """riggs_amd.dist.ShardedAdam on the GPU (its shard update runs through riggs_adam_step on slices of the flat buffers;
    world size 1: the collectives drop out) against torch.optim.Adam over three steps, with per-group learning rates and
    tensor sizes that are not multiples of four."""
    from riggs_amd.dist import ShardedAdam
    g = torch.Generator().manual_seed(4)
    shapes = [(1001, 1, 3), (1001, 15, 3), (1001, 1), (1001, 3), (1001, 4), (7,), (9, 17)]
    mine = [torch.nn.Parameter(torch.randn(*s, generator=g).cuda()) for s in shapes]
    ref = [torch.nn.Parameter(p.detach().clone()) for p in mine]
    lrs = [2.5e-3, 1.25e-4, 5e-2, 1e-3, 1e-3, 5e-4, 5e-4]
    opt = ShardedAdam([{"params": [p], "lr": lr} for p, lr in zip(mine, lrs)], eps=1e-15)
    topt = torch.optim.Adam([{"params": [p], "lr": lr} for p, lr in zip(ref, lrs)], lr=0.0, eps=1e-15)
    assert all(p.data_ptr() >= opt.flat_p.data_ptr() for p in mine)
    for step in range(3):
        grads = [torch.randn(*s, generator=g).cuda() for s in shapes]
        for v, gr, r in zip(opt.bucket.views, grads, ref):
            v.copy_(gr)
            r.grad = gr.clone()
        opt.step()
        topt.step()
    for a, b in zip(mine, ref):
        torch.testing.assert_close(a.detach(), b.detach(), rtol=3e-6, atol=3e-7)
