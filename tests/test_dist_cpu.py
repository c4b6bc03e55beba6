"""CPU, world_size 2, gloo: the frame-sharded gradient all-reduce (riggs_amd/dist.py).
Each rank computes the gradient of ITS frame with the CPU oracle pipeline (the HIP path needs a
GPU); after the all-reduce every rank must hold the average of the per-frame gradients computed
serially, and the flat buffer views must alias p.grad."""
import math
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from oracle import deform_ref as O
from oracle import raster_ref as RR
from riggs_amd import synth
from riggs_amd.dist import FlatGradAllReduce, frame_for_rank

N, J, H, W = 600, 6, 48, 48


def _frame_grads(rank_frame):
    sc = synth.make_scene(N, J, 9, scale=0.05)
    cam = synth.look_at_camera(H, W, azimuth_deg=45.0 * rank_frame)
    leaf = lambda t: t.clone().requires_grad_(True)  # noqa: E731
    names = ("xyz", "features_dc", "features_rest", "scaling", "rotation", "opacity", "local_rotation", "global_trans",
             "node_radius")
    P = {k: leaf(sc[k]) for k in names}
    dv = O.deform_by_pose(P["xyz"].detach(), sc["joints"], sc["parents"], P["node_radius"], P["local_rotation"],
                          P["global_trans"], sc["motion_mask"], -1)
    m3, op, scl, rot, shs = O.render_glue(P["xyz"], P["features_dc"], P["features_rest"], P["scaling"], P["rotation"],
                                          P["opacity"], dv["d_xyz"], dv["d_rotation"], dv["d_scaling"])
    out, saved = RR.forward(m3.detach().numpy(), op.detach().numpy(), cam.world_view_transform.numpy(),
                            cam.full_proj_transform.numpy(), cam.camera_center.numpy(), math.tan(cam.FoVx / 2),
                            math.tan(cam.FoVy / 2), H, W, np.zeros(3, np.float32), shs=shs.detach().numpy(),
                            scales=scl.detach().numpy(), rotations=rot.detach().numpy())
    g = RR.backward(saved, np.full((3, H, W), 1.0 / (3 * H * W), np.float32), None, None)
    T = torch.from_numpy
    torch.autograd.backward([m3, op, scl, rot, shs], [T(g["means3D"]), T(g["opacities"]), T(g["scales"]),
                                                      T(g["rotations"]), T(g["shs"])])
    return [P[k] for k in names]


def _worker(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.set_num_threads(1)
    params = _frame_grads(frame_for_rank(list(range(8)), 0, rank, world))
    ar = FlatGradAllReduce(params)
    flat = ar()
    assert all(p.grad.data_ptr() >= flat.data_ptr() for p in params)  # views into the flat buffer
    q.put((rank, [p.grad.clone().numpy() for p in params]))
    dist.barrier()
    dist.destroy_process_group()


def test_frame_sharded_all_reduce_matches_serial_average():
    world = 2
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    got = dict(q.get(timeout=240) for _ in range(world))
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    serial = [[p.grad.numpy() for p in _frame_grads(f)] for f in range(world)]
    mean = [sum(g) / world for g in zip(*serial)]
    for r in range(world):
        for a, b in zip(got[r], mean):
            np.testing.assert_allclose(a, b, rtol=1e-5, atol=1e-9)
    assert not np.allclose(serial[0][0], serial[1][0])  # the two frames really differ


def test_frame_assignment_is_one_frame_per_rank():
    frames = list(range(8))
    assert [frame_for_rank(frames, 0, r, 8) for r in range(8)] == frames
    assert frame_for_rank(frames, 1, 3, 4) == 7


def test_bucket_slices_are_not_handed_out_twice_and_die_with_their_parameter():
    """riggs_amd.dist.grad_out: a parameter whose .grad already aliases its slice gets a fresh buffer for the next
    backward (autograd then accumulates: no lost / doubled gradient), and an entry does not outlive its parameter."""
    import gc
    from riggs_amd import dist as D
    ps = [torch.nn.Parameter(torch.randn(8, 3)), torch.nn.Parameter(torch.randn(12))]
    bucket = FlatGradAllReduce(ps, register=True)
    try:
        first = D.grad_out(ps[0])
        assert first.data_ptr() == bucket.views[0].data_ptr()
        first.fill_(1.0)
        ps[0].grad = first                      # what AccumulateGrad does with the first backward's output
        second = D.grad_out(ps[0])              # second backward of the same step
        assert second.data_ptr() != first.data_ptr()
        second.fill_(2.0)
        ps[0].grad += second
        assert float(ps[0].grad.sum()) == 3.0 * 24 and float(bucket.views[0].sum()) == 3.0 * 24
        ps[0].grad = None
        assert D.grad_out(ps[0]).data_ptr() == first.data_ptr()
        flat = D.grad_out_flat(ps)
        assert flat.data_ptr() == bucket.flat.data_ptr() and flat.numel() == 36
        ps[1].grad = bucket.views[1]
        assert D.grad_out_flat(ps).data_ptr() != bucket.flat.data_ptr()
        # an entry whose parameter died is dropped instead of capturing the next tensor at that address
        key = ps[1].data_ptr()
        bucket.params, bucket.views = bucket.params[:1], bucket.views[:1]
        del ps[1]
        gc.collect()
        probe = torch.empty(12)
        probe.data_ptr = lambda: key  # a new tensor the allocator placed at the dead parameter's address
        assert D._entry(probe) is None and key not in D._SLICES
    finally:
        bucket.unregister()


# ---- the overlapped / sharded / compacted exchange paths (world size 2 over gloo) ------------------------------------
def _spawn(target, world=2, timeout=240):
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_entry_point, args=(target, r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    got = dict(q.get(timeout=timeout) for _ in range(world))
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    return got


def _entry_point(target, rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.set_num_threads(1)
    q.put((rank, target(rank, world)))
    dist.barrier()
    dist.destroy_process_group()


def _toy_params(seed=0):
    g = torch.Generator().manual_seed(seed)
    shapes = [(37, 1, 3), (37, 15, 3), (37, 1), (37, 3), (37, 3), (37, 4), (6,), (9, 17), (9,)]
    return [torch.nn.Parameter(torch.randn(*s, generator=g)) for s in shapes]


def _rank_grads(params, rank):
    g = torch.Generator().manual_seed(100 + rank)
    return [torch.randn(p.shape, generator=g) for p in params]


def _w_overlapped(rank, world):
    from riggs_amd.dist import OverlappedExchange
    params = _toy_params()
    bucket = FlatGradAllReduce(params, register=False)
    for v, gr in zip(bucket.views, _rank_grads(params, rank)):
        v.copy_(gr)
    split = bucket.offsets[4]  # the first four tensors form phase 1
    ex = OverlappedExchange(bucket, split, chunk_bytes=256)  # several collectives per phase
    ex.launch(1)
    assert len(ex.pending) > 1
    ex.launch(2)
    ex.wait()
    return [v.clone().numpy() for v in bucket.views]


def test_overlapped_two_phase_exchange_equals_the_mean():
    got = _spawn(_w_overlapped)
    params = _toy_params()
    mean = [sum(gs) / 2 for gs in zip(_rank_grads(params, 0), _rank_grads(params, 1))]
    for r in range(2):
        for a, b in zip(got[r], mean):
            np.testing.assert_allclose(a, b.numpy(), rtol=1e-6, atol=1e-7)


def _w_sharded(rank, world):
    from riggs_amd.dist import ShardedAdam
    params = _toy_params()
    groups = [{"params": params[:2], "lr": 2.5e-3}, {"params": params[2:6], "lr": 1e-2}, {"params": params[6:], "lr": 5e-4}]
    opt = ShardedAdam(groups, eps=1e-15)
    assert opt.shard % 4 == 0 and opt.shard * world >= opt.bucket.numel
    for step in range(3):
        for v, gr in zip(opt.bucket.views, _rank_grads(params, rank + 10 * step)):
            v.copy_(gr)
        opt.step()
    assert all(p.data_ptr() >= opt.flat_p.data_ptr() for p in params)  # parameters are views of the gathered flat buffer
    return [p.detach().clone().numpy() for p in params], (opt.m.numel(), opt.bucket.numel)


def test_sharded_adam_equals_adam_on_the_averaged_gradients():
    got = _spawn(_w_sharded)
    params = _toy_params()
    ref = torch.optim.Adam([{"params": params[:2], "lr": 2.5e-3}, {"params": params[2:6], "lr": 1e-2},
                            {"params": params[6:], "lr": 5e-4}], lr=0.0, eps=1e-15)
    for step in range(3):
        for p, g0, g1 in zip(params, _rank_grads(params, 10 * step), _rank_grads(params, 1 + 10 * step)):
            p.grad = (g0 + g1) / 2
        ref.step()
    for r in range(2):
        vals, (m_numel, n) = got[r]
        assert m_numel < 0.51 * n + 8  # each rank holds half of the moments
        for a, p in zip(vals, params):
            np.testing.assert_allclose(a, p.detach().numpy(), rtol=2e-5, atol=2e-7)
    for a, b in zip(got[0][0], got[1][0]):
        assert np.array_equal(a, b)  # replicas stay bit-identical


def _w_sparse(rank, world):
    from riggs_amd.dist import sparse_rows_all_reduce
    N = 200
    g = torch.Generator().manual_seed(7 + rank)
    rows = torch.randperm(N, generator=g)[:25]          # different Gaussians are touched on different ranks
    grads = [torch.zeros(N, 3), torch.zeros(N, 15, 3), torch.zeros(N, 1)]
    for t in grads:
        t[rows] = torch.randn(25, *t.shape[1:], generator=g)
    dense = [t.clone() for t in grads]
    need = sparse_rows_all_reduce(grads, capacity=40)
    for t in dense:
        dist.all_reduce(t)
        t /= world
    small = [t.clone() for t in dense]  # (values irrelevant) a capacity that is too small is reported, not silently wrong
    mine = torch.ones(N, 2) * (rank + 1)
    over = sparse_rows_all_reduce([mine], capacity=16)
    kept = bool((mine == rank + 1).all())               # ... and on overflow the LOCAL gradients survive (a dense redo is possible)
    return [a.numpy() for a in grads], [b.numpy() for b in dense], int(need), int(over), len(small), kept


def test_compacted_row_exchange_equals_the_dense_all_reduce_and_reports_overflow():
    got = _spawn(_w_sparse)
    for r in range(2):
        sparse, dense, need, over, _, kept = got[r]
        assert need == 25 and over == 200   # rows the fullest rank needed; 200 > capacity 16: the caller must go dense
        assert kept                          # (nothing was written back: no row was zeroed by the truncated compaction)
        for a, b in zip(sparse, dense):
            np.testing.assert_allclose(a, b, rtol=1e-6, atol=1e-7)


# ---- the gradient-row exchange: protocol over gloo with the numpy restatement of the segment format standing in for
# the HIP pack / unpack kernels (tests/rows_ref.py; the kernels themselves are checked against it in test_gpu_api.py) ----
def _w_rows(rank, world):
    import rows_ref
    from riggs_amd.dist import SparseRowExchange
    N = 700                                                 # three blocks of 256, the last one ragged
    g = torch.Generator().manual_seed(31 + rank)
    touched = torch.zeros(N, dtype=torch.bool)
    touched[torch.randperm(N, generator=g)[:60]] = True    # different Gaussians on different ranks, some shared
    touched[5] = True
    grads = [torch.zeros(N, 3), torch.zeros(N, 45), torch.zeros(N, 1), torch.zeros(N, 4)]
    for t in grads:
        t[touched] = torch.randn(int(touched.sum()), t.shape[1], generator=g)
    rest = torch.randn(50, generator=g)
    dense = [t.clone() for t in grads] + [rest.clone()]
    for t in dense:
        dist.all_reduce(t)
        t /= world

    def pack(ex):
        seg = rows_ref.pack([t.numpy() for t in ex.rows], touched.numpy(), 1.0 / ex.world, ex.capacity, ex.row_floats)
        ex.segment.copy_(torch.from_numpy(seg))

    def unpack(ex):
        need, bad = rows_ref.unpack([t.numpy() for t in ex.rows], ex.gathered.view(ex.world, -1).numpy(), ex.capacity, ex.row_floats)
        ex.status[0], ex.status[1] = need, int(bad)

    out = {}
    for name, cap in (("fits", 80), ("overflows", 40)):
        mine = [t.clone() for t in grads]
        r = rest.clone()
        ex = SparseRowExchange(mine, rest=r, capacity=cap, pack=pack, unpack=unpack)
        ex.pack()
        ex.launch()
        ex.launch_rest()
        ex.wait()
        ok = ex.check()
        untouched = all(torch.equal(a, b) for a, b in zip(mine, grads))
        if not ok:
            ex.dense_fallback()
        out[name] = (ok, ex.need, untouched, [t.numpy() for t in mine] + [r.numpy()], ex.wins)
    return out, [t.numpy() for t in dense]


def test_gradient_row_exchange_equals_the_dense_mean_and_falls_back_on_overflow():
    import sys
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    got = _spawn(_w_rows)
    for r in range(2):
        out, dense = got[r]
        ok, need, untouched, vals, wins = out["fits"]
        assert ok and need == 61 and not untouched and wins
        for a, b in zip(vals, dense):
            np.testing.assert_allclose(a, b, rtol=1e-6, atol=1e-7)
        ok, need, untouched, vals, _ = out["overflows"]
        assert not ok and need == 61 and untouched          # nothing was unpacked: the rows still hold the local gradients
        for a, b in zip(vals, dense):                       # ... and the dense fallback of that step gives the mean
            np.testing.assert_allclose(a, b, rtol=1e-6, atol=1e-7)
    for a, b in zip(got[0][0]["fits"][3], got[1][0]["fits"][3]):
        assert np.array_equal(a, b)                         # replicas are bit-identical after the ordered unpack


def test_segment_size_matches_the_library():
    from riggs_amd import _lib as L
    from riggs_amd.dist import segment_words
    import ctypes as C
    lib = L.lib()
    w = (C.c_int32 * 6)(3, 45, 1, 3, 3, 4)
    assert lib.riggs_grad_rows_row_floats(6, w) == 60
    for N, cap in ((1, 0), (700, 80), (300_000, 30_000), (2_000_000, 1)):
        assert lib.riggs_grad_rows_segment_bytes(N, 60, cap) == 4 * segment_words(N, 60, cap)


def test_sharded_adam_reads_the_learning_rate_of_its_groups_at_every_step():
    """The reference trainer rewrites the xyz learning rate every iteration (update_learning_rate): the sharded optimizer keeps
    the param groups and follows them (world 1 on the CPU; the schedule hook ``set_lr`` by group name)."""
    from riggs_amd.dist import ShardedAdam
    a = [torch.nn.Parameter(torch.randn(7, 3)), torch.nn.Parameter(torch.randn(5))]
    b = [torch.nn.Parameter(p.detach().clone()) for p in a]
    opt = ShardedAdam([{"params": [a[0]], "lr": 1e-2, "name": "xyz"}, {"params": [a[1]], "lr": 1e-3, "name": "f_dc"}])
    ref = torch.optim.Adam([{"params": [b[0]], "lr": 1e-2}, {"params": [b[1]], "lr": 1e-3}], eps=1e-15)
    g = torch.Generator().manual_seed(3)
    for step, lr in enumerate((1e-2, 5e-3, 1e-4)):
        opt.set_lr("xyz", lr)
        ref.param_groups[0]["lr"] = lr
        for pa, pb, v in zip(a, b, opt.bucket.views):
            gr = torch.randn(pa.shape, generator=g)
            v.copy_(gr)
            pb.grad = gr.clone()
        opt.step()
        ref.step()
    for pa, pb in zip(a, b):
        np.testing.assert_allclose(pa.detach().numpy(), pb.detach().numpy(), rtol=2e-5, atol=2e-7)
    with pytest.raises(KeyError):
        opt.set_lr("nope", 1.0)


def test_direct_exchange_refuses_gradients_that_do_not_alias_the_bucket():
    from riggs_amd.dist import FlatGradAllReduce, OverlappedExchange
    ps = [torch.nn.Parameter(torch.randn(4, 3)), torch.nn.Parameter(torch.randn(6))]
    bucket = FlatGradAllReduce(ps, register=False)
    ps[0].grad = bucket.views[0]
    ps[1].grad = torch.zeros(6)               # a stray buffer: the direct reduction of bucket.flat would never see it
    ex = OverlappedExchange(bucket, bucket.offsets[1])
    with pytest.raises(RuntimeError, match="does not alias"):
        ex.launch(1)
    ps[1].grad = bucket.views[1]
    ex.launch(1); ex.launch(2); ex.wait()
