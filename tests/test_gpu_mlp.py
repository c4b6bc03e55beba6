"""-m gpu: the fused MFMA MLP heads (csrc/mlp.hip) against the fp32 host mirrors.

The reference computes WeightMLP / DeformMLP in fp32 (skeleton_utils/network_utils.py:6-112); the fused path rounds the
operands to 16 bits (fp32 accumulation).  With fp16 operands (the default: 11 significand bits, the incoming gradient scaled
by a power of two on the device) the outputs are within 4e-4 .. 1.5e-3 of the fp32 mirror and the parameter gradients within
5 % over all parameters (1-6.5 % per tensor, first layers worst); with bf16 (8 bits) it is 3e-3 .. 1e-2 and 4-19 %.  The sharper test of the kernels
themselves: agreement with a torch emulation that applies the SAME roundings (identical ReLU masks up to a handful of
elements in millions)."""
import pytest
import torch

pytestmark = pytest.mark.gpu

from riggs_amd import mlp as M  # noqa: E402
from riggs_amd.skeleton import DeformMLP, WeightMLP, _embed  # noqa: E402


def _rel(a, b):
    return float((a - b).abs().max() / b.abs().max().clamp_min(1e-20))


def _mrel(a, b):
    return float((a - b).abs().mean() / b.abs().mean().clamp_min(1e-20))


def test_mfma_fragment_layout_selftest():
    assert torch.equal(M.layout_probe().cpu(), M.expected_probe())


def _nets(N):
    torch.manual_seed(3)
    x = torch.randn(N, 3, device="cuda") * 0.5
    wm = WeightMLP(3, 23).cuda()
    dn = DeformMLP(xyz_input_ch=3, time_input_ch=96).cuda()
    with torch.no_grad():
        dn.gaussian_warp.weight.mul_(2000.0)
    pose = torch.randn(96, device="cuda")[None].expand(N, -1)
    return [("WeightMLP", wm, wm.weight_predict, _embed(x, wm.multires).contiguous()),
            ("DeformMLP", dn, dn.gaussian_warp, torch.cat([_embed(x, dn.multires), pose], -1).contiguous())]


def _hidden(net, xe, rnd=lambda t: t):
    hs, h = [], rnd(xe)
    x0 = h
    for i, layer in enumerate(net.linear):
        h = rnd(torch.relu(torch.nn.functional.linear(h, rnd(layer.weight), layer.bias)))
        hs.append(h)
        if i in net.skips:
            h = torch.cat([x0, h], -1)
    return h, hs


@pytest.mark.parametrize("fmt,gscale", [("fp16", 1.0), ("fp16", 3e-8), ("bf16", 1.0)])
@pytest.mark.parametrize("N", [1, 63, 20_011])
def test_forward_and_gradients_vs_fp32_mirror_and_bf16_emulation(N, fmt, gscale):
    """``gscale`` = 3e-8: gradients the size a per-pixel-averaged image loss produces — below half precision's normal range
    unless they are scaled on the way in (riggs_amd.mlp.grad_scale)."""
    half = M.FORMATS[fmt]
    out_tol = {"fp16": (1.5e-3, 6e-3), "bf16": (6e-3, 4e-2)}[fmt]
    grad_tol = {"fp16": 0.08, "bf16": 0.2}[fmt]   # every tensor; fp16 additionally: 5 % over all parameters (below)
    for name, net, head, xe in _nets(N):
        g = torch.randn(N, head.weight.shape[0], device="cuda") * gscale
        # fp32 mirror
        out32 = head(_hidden(net, xe)[0])
        (out32 * g).sum().backward()
        g32 = {n: q.grad.clone() for n, q in net.named_parameters()}
        for q in net.parameters():
            q.grad = None
        # fused
        fh = M.FusedHead(net.linear, head, xe.shape[1], net.skips[0], fmt)
        out = fh(xe)
        (out * g).sum().backward()
        assert _rel(out, out32) < (out_tol[0] if name == "WeightMLP" else out_tol[1]), (name, _rel(out, out32))
        if N > 1000:  # gradient statistics need rows; operand rounding + the ReLU-mask flips it causes bound this comparison
            for n, q in net.named_parameters():
                assert _mrel(q.grad, g32[n]) < grad_tol, (name, fmt, n, _mrel(q.grad, g32[n]))
            if fmt == "fp16":  # measured: WeightMLP 1-4 % per tensor, DeformMLP (head scaled x2000) 2-6.5 %; bf16: 4-11 % / 6-19 %
                num = sum(float((q.grad - g32[n]).abs().sum()) for n, q in net.named_parameters())
                den = sum(float(g32[n].abs().sum()) for n, q in net.named_parameters())
                assert num / den < 0.05, (name, num / den)
        # bf16 emulation: same roundings in torch
        class RoundBF(torch.autograd.Function):
            @staticmethod
            def forward(ctx, t):
                return t.to(half).float()

            @staticmethod
            def backward(ctx, gg):
                return gg
        h, hs = _hidden(net, xe, RoundBF.apply)
        for t in hs:
            t.retain_grad()
        out_e = torch.nn.functional.linear(h, RoundBF.apply(head.weight), head.bias)
        sc = M.grad_scale(g) if fmt == "fp16" else None
        gs = g if sc is None else g * sc
        (out_e * gs.to(half).float()).sum().backward()
        assert _rel(out, out_e) < (2e-3 if fmt == "fp16" else 1.5e-2), (name, _rel(out, out_e))  # accumulation order differs from the library GEMM
        p = fh._packed()
        o2, (acts, masks) = M.forward(p, xe, True)
        dpre, _db = M.backward_data(p, g, masks, sc)
        if sc is not None:
            assert float(torch.log2(sc).frac()) == 0.0 and 256.0 <= float((g * sc).abs().max()) <= 1024.0
        flips = sum(int(((acts[l].float() > 0) != (hs[l] > 0)).sum()) for l in range(p.depth))
        assert flips <= max(4, int(2e-5 * acts.numel())), flips
        if N > 1000:
            for l in range(p.depth):
                dref = hs[l].grad * (hs[l] > 0)
                assert _mrel(dpre[l].float(), dref) < (4e-3 if fmt == "fp16" else 2e-2), (name, l, _mrel(dpre[l].float(), dref))
        for q in net.parameters():
            q.grad = None


@pytest.mark.parametrize("fmt", ["fp16", "bf16"])
@pytest.mark.parametrize("N", [1, 31, 33, 4097, 50_003])
def test_param_grads_kernel_equals_the_products_of_its_operands(N, fmt):
    """riggs_mlp_wgrad (csrc/mlp_wgrad.hip) against float64 products of the SAME 16-bit operands — what is left is the fp32
    accumulation order (1e-5 of a tensor's largest entry) — for both heads (embedding widths 64 and 128 after padding), ragged
    N (the last 32-Gaussian stage reads a zero page), with and without the gradient scale; and against the library path."""
    for name, net, head, xe in _nets(N):
        fh = M.FusedHead(net.linear, head, xe.shape[1], net.skips[0], fmt)
        pk = fh._packed()
        xb = M.embed_bf16(pk, xe)
        _out, (acts, masks) = M.forward(pk, xe, True, xb)
        g = torch.randn(N, pk.out_ch, device="cuda") * (3e-8 if fmt == "fp16" else 1.0)
        sc = M.grad_scale(g) if fmt == "fp16" else None
        dpre, db = M.backward_data(pk, g, masks, sc)
        got = M.param_grads(pk, xb, acts, dpre, g, sc)
        lib = M.library_param_grads(pk, xb, acts, dpre, db, g, sc)
        inv = 1.0 if sc is None else 1.0 / float(sc)
        gob = (g if sc is None else g * sc).to(pk.dtype).double()
        x64 = xb[:N, :pk.in_ch].double()
        want = []
        for l in range(pk.depth):
            d64 = dpre[l].double()
            a64 = x64 if l == 0 else acts[l - 1].double()
            if l == pk.skip + 1:
                a64 = torch.cat([x64, a64], 1)
            want += [d64.t() @ a64 * inv, d64.sum(0) * inv]
        want += [gob.t() @ acts[pk.depth - 1].double() * inv, gob.sum(0) * inv]
        assert len(got) == len(want) == len(lib) == 2 * pk.depth + 2
        for i, (a, b, c) in enumerate(zip(got, want, lib)):
            assert a.shape == b.shape == c.shape and a.dtype == torch.float32, (name, i, a.shape, b.shape)
            tol = 2e-5 * float(b.abs().max()) + 1e-30
            assert float((a.double() - b).abs().max()) <= tol, (name, fmt, N, i, float((a.double() - b).abs().max()), tol)
            assert float((c.double() - b).abs().max()) <= 50 * tol, (name, "library", i)


@pytest.mark.parametrize("fmt", ["fp16", "bf16"])
@pytest.mark.parametrize("N", [1, 129, 4097, 50_003])
def test_stored_activations_are_the_layers_outputs(N, fmt):
    """The activations the training forward leaves in HBM — sent one 16-byte piece per thread and K-step under the NEXT layer's
    product, behind counted waits (mlp_gemm_hidden_stb) — are every layer's relu(W a + b) of the STORED layer below (same 16-bit operands, fp32 accumulation:
    what is left is the rounding of the result), finite everywhere, for ragged row counts and for rows past 2^15 (byte offsets
    past 2^24); three launches each: a missed wait state or a wrong vmcnt count shows as sporadic garbage."""
    for name, net, head, xe in _nets(N):
        fh = M.FusedHead(net.linear, head, xe.shape[1], net.skips[0], fmt)
        pk = fh._packed()
        xb = M.embed_bf16(pk, xe)
        out0 = M.forward(pk, xe, False, xb)[0]
        x16 = xb[:N, :pk.in_ch].float()
        eps = 2.0 ** -10 if fmt == "fp16" else 2.0 ** -7
        for _rep in range(3):
            out, (acts, _masks) = M.forward(pk, xe, True, xb)
            assert torch.equal(out, out0), (name, fmt, N)
            assert bool(torch.isfinite(acts.float()).all()), (name, fmt, N)
            for l in range(pk.depth):
                W = net.linear[l].weight.detach().to(pk.dtype).float()
                b = net.linear[l].bias.detach().float()
                a_in = x16 if l == 0 else acts[l - 1].float()
                if l == pk.skip + 1:
                    a_in = torch.cat([x16, a_in], 1)
                ref = torch.relu(a_in @ W.t() + b)
                d = (acts[l].float() - ref).abs()
                assert bool((d <= 2 * eps * (ref.abs() + a_in.abs().max() * 1e-2 + 1e-6)).all()), (name, fmt, N, l, float(d.max()))


@pytest.mark.parametrize("fmt", ["fp16", "bf16"])
@pytest.mark.parametrize("N", [1, 63, 20_011])
def test_constant_input_tail_through_the_biases(N, fmt):
    """DeformMLP's pose is ONE vector for all rows (skeleton_warp.py:152): with ``tail_ch`` the head packs the positional
    embedding alone, the pose enters through the two biases in fp32 (riggs_mlp_tail_bias) and its weight columns' gradient is
    (bias gradient) x pose (riggs_mlp_wgrad_tail).  Against the fp32 mirror (the bars of the generic fused path) and against the
    generic fused path itself (same kernels with the pose as 96 more operand columns, rounded to 16 bits)."""
    name, net, head, xe = _nets(N)[1]
    assert name == "DeformMLP"
    torch.manual_seed(11)
    x = torch.randn(N, 3, device="cuda") * 0.5
    pose = torch.randn(96, device="cuda")
    xe = torch.cat([_embed(x, net.multires), pose[None].expand(N, -1)], -1).contiguous()
    g = torch.randn(N, head.weight.shape[0], device="cuda") * (3e-8 if fmt == "fp16" else 1.0)
    out32 = head(_hidden(net, xe)[0])
    (out32 * g).sum().backward()
    g32 = {n: q.grad.clone() for n, q in net.named_parameters()}
    for q in net.parameters():
        q.grad = None
    gen = M.FusedHead(net.linear, head, xe.shape[1], net.skips[0], fmt)
    out_g = gen(xe)
    (out_g * g).sum().backward()
    ggen = {n: q.grad.clone() for n, q in net.named_parameters()}
    for q in net.parameters():
        q.grad = None
    fh = M.FusedHead(net.linear, head, xe.shape[1] - 96, net.skips[0], fmt, tail_ch=96)
    with pytest.raises(ValueError):
        fh(M.embed_positions_bf16(x, net.multires, fmt=fmt), n_rows=N)
    out = fh(M.embed_positions_bf16(x, net.multires, fmt=fmt), n_rows=N, tail=pose)
    (out * g).sum().backward()
    tol = {"fp16": 6e-3, "bf16": 4e-2}[fmt]
    assert _rel(out, out32) < tol, _rel(out, out32)
    assert _rel(out, out_g) < tol, _rel(out, out_g)
    skip_w = "linear.%d.weight" % (net.skips[0] + 1)
    for n, q in net.named_parameters():
        assert q.grad is not None and q.grad.shape == g32[n].shape, n
        assert bool(torch.isfinite(q.grad).all()), n
        if N > 1000:
            assert _mrel(q.grad, g32[n]) < {"fp16": 0.08, "bf16": 0.2}[fmt], (n, _mrel(q.grad, g32[n]))
            assert _mrel(q.grad, ggen[n]) < {"fp16": 0.08, "bf16": 0.2}[fmt], (n, _mrel(q.grad, ggen[n]))
        if n in ("linear.0.weight", skip_w):  # the pose's columns: exactly (bias gradient) x pose
            b = net.linear[0].bias.grad if n == "linear.0.weight" else net.linear[net.skips[0] + 1].bias.grad
            want = b[:, None] * pose[None, :]
            got = q.grad[:, 27:27 + 96]
            assert float((got - want).abs().max()) <= 1e-6 * float(want.abs().max()) + 1e-30, n
    for q in net.parameters():
        q.grad = None


def test_skeleton_warp_with_fused_heads_tracks_the_fp32_heads():
    from riggs_amd import synth
    from riggs_amd.skeleton import SkeletonWarp
    sc = synth.make_scene(5_003, 24, 9)
    sw = SkeletonWarp(joints=sc["joints"], parent_indices=sc["parents"], K=-1, hyper_dim=8).cuda()
    x = sc["xyz"].cuda()
    q = torch.nn.functional.normalize(torch.randn(24, 4, device="cuda"), dim=-1)
    gt = torch.zeros(3, device="cuda")
    outs = {}
    for fused in (False, True):
        sw.use_fused_heads(fused)
        o = sw.deform_by_pose(x, {"local_rotation": q.clone().requires_grad_(True), "global_trans": gt.clone().requires_grad_(True)}, None)
        outs[fused] = (o["d_xyz"].detach(), o["d_rotation"].detach())
    assert _rel(outs[True][0], outs[False][0]) < 2e-2 and _rel(outs[True][1], outs[False][1]) < 2e-2


def test_fused_head_follows_the_fused_optimizer():
    """FusedAdam updates parameters through raw pointers; the bf16 copies of the fused head must follow."""
    from riggs_amd.optim import FusedAdam
    name, net, head, xe = _nets(4_096)[0]
    fh = M.FusedHead(net.linear, head, xe.shape[1], net.skips[0])
    opt = FusedAdam(list(net.parameters()), lr=1e-2, eps=1e-15)
    out0 = fh(xe)
    (out0 ** 2).sum().backward()
    opt.step()
    out1 = fh(xe).detach()
    ref1 = head(_hidden(net, xe)[0]).detach()
    assert _rel(out1, out0.detach()) > 1e-2          # the step changed the function
    assert _rel(out1, ref1) < 6e-3                   # and the fused head sees the new weights


def test_embedding_kernel_matches_the_reference_embedder():
    torch.manual_seed(1)
    N = 1_001
    x = torch.randn(N, 3, device="cuda")
    pose = torch.randn(96, device="cuda")
    for multires, tail, fmt in ((10, None, "bf16"), (4, pose, "bf16"), (10, None, "fp16"), (4, pose, "fp16")):
        xb = M.embed_positions_bf16(x, multires, tail, fmt=fmt)
        ref = _embed(x, multires) if tail is None else torch.cat([_embed(x, multires), tail[None].expand(N, -1)], -1)
        assert xb.shape == ((N + 127) // 128 * 128, (ref.shape[1] + 31) // 32 * 32) and xb.dtype == M.FORMATS[fmt]
        assert float((xb[:N, :ref.shape[1]].float() - ref).abs().max()) < (8e-3 if fmt == "bf16" else 1e-3) * float(ref.abs().max())  # operand rounding
        assert float(xb[N:].float().abs().max()) == 0.0 and float(xb[:, ref.shape[1]:].float().abs().max()) == 0.0


def test_empty_input():
    name, net, head, xe = _nets(8)[0]
    fh = M.FusedHead(net.linear, head, xe.shape[1], net.skips[0])
    out = fh(xe[:0].contiguous())
    assert out.shape == (0, head.weight.shape[0])
    out.sum().backward()
    assert all(q.grad is not None and float(q.grad.abs().max()) == 0.0 for q in net.parameters())


def test_fused_heads_inside_a_captured_training_iteration():
    """The bf16 copies of the head weights are rebuilt INSIDE the hipGraph (the captured optimizer step changes the fp32
    masters without any Python running): every replay must see the weights of the step before it."""
    import bench
    from riggs_amd import synth
    from riggs_amd.gaussian_model import GaussianModel
    from riggs_amd.graph import GraphedTrainStep
    from riggs_amd.optim import FusedAdam
    from riggs_amd.skeleton import SkeletonWarp
    sc = synth.make_scene(4_000, 8, 4)
    cam = synth.look_at_camera(64, 80, fid=0.3).to("cuda")
    gm = GaussianModel.from_tensors(sc["xyz"], sc["features_dc"], sc["features_rest"], sc["scaling"], sc["rotation"],
                                    sc["opacity"], device="cuda")
    torch.manual_seed(0)
    sw = SkeletonWarp(joints=sc["joints"], parent_indices=sc["parents"], K=-1, hyper_dim=8).cuda().use_fused_heads(True)
    gm.training_setup(bench._train_args(), capturable=True)
    opt = FusedAdam([{"params": g["params"], "lr": 5e-4, "name": g["name"]} for g in sw.trainable_parameters()],
                    lr=0.0, eps=1e-15, capturable=True)
    target = torch.rand(3, 64, 80, device="cuda")
    gts = GraphedTrainStep(gm, sw, cam, torch.zeros(3, device="cuda"), target, [gm.optimizer, opt], lambda_dssim=0.2)
    gts.capture(warmup=1)
    w0 = sw.skinning_weight_mlp.linear[2].weight.detach().clone()
    losses = []
    for _ in range(3):
        gts.run()
        losses.append(float(gts.out["loss"]))
    w1 = sw.skinning_weight_mlp.linear[2].weight.detach().clone()
    assert all(v == v and v < 10 for v in losses)
    assert float((w1 - w0).abs().max()) > 1e-4                       # the captured optimizer moved the masters
    pk = sw._fh_w._pk                                                 # the bf16 copy the LAST replay used = masters before its own step
    assert float((pk.rows(2)[:, :256].float() - w1).abs().max()) < 2e-3   # within one Adam step (5e-4) + bf16 rounding of the masters


def test_grad_scale_kernels_match_the_formula():
    """riggs_mlp_grad_scale: 2^floor(log2(1024 / max|g|)) from two launches — against the same formula in torch, for ragged sizes
    (the vector loop's tail), a single element, all zeros (clamped at 1e-30), values around the 16-bit subnormal range, and twice in
    a row (the scratch word is left zero)."""
    torch.manual_seed(0)
    for n, mag in ((1, 1.0), (3, 1e-8), (4, 3.0), (1025, 1e-6), (300_000 * 23, 3e-8), (7, 0.0)):
        g = torch.randn(n, device="cuda") * mag
        for _ in range(2):
            sc = M.grad_scale(g)
            amax = g.abs().amax().clamp_min(1e-30)
            want = torch.exp2(torch.floor(torch.log2(1024.0 / amax)))
            assert sc.shape == (1,) and float(torch.log2(sc).frac()) == 0.0
            ratio = float(sc) / float(want)
            assert ratio in (0.5, 1.0, 2.0), (n, mag, float(sc), float(want))   # (log2 of an exact power of two may round either way)
            if mag > 0:
                assert 256.0 <= float(amax) * float(sc) <= 2048.0
    assert all(int(w) == 0 for w in M._ZERO_WORD.values())


def _sparse_cotangent(N, out_ch, frac, seed=5, mag=3e-8):
    g = torch.Generator(device="cuda").manual_seed(seed)
    v = torch.randn(N, out_ch, device="cuda", generator=g) * mag
    if frac <= 0.0:
        return torch.zeros_like(v)
    if frac >= 1.0:
        return v
    live = torch.rand(N, device="cuda", generator=g) < frac
    # (a live row may hold zeros in some columns, and one row is live through a single column only)
    v = v * live[:, None] * (torch.rand(N, out_ch, device="cuda", generator=g) < 0.7)
    if N > 2:
        v[N // 2] = 0.0
        v[N // 2, out_ch - 1] = mag
    return v


@pytest.mark.parametrize("N", [1, 255, 256, 257, 4_097, 70_001])
@pytest.mark.parametrize("frac", [0.0, 0.1, 1.0])
def test_live_rows_compaction(N, frac):
    """riggs_mlp_live_rows: ascending indices of the rows with a non-zero, their count, the gathered operand and cotangent rows,
    the zero fill up to the next multiple of 128 — against torch.nonzero."""
    name, net, head, xe = _nets(N)[0]
    pk = M.FusedHead(net.linear, head, xe.shape[1], net.skips[0])._packed()
    xb = M.embed_bf16(pk, xe)
    g = _sparse_cotangent(N, pk.out_ch, frac)
    idx, count, xl, gl = M.live_rows(pk, g, xb)
    want = torch.nonzero((g != 0).any(1)).flatten()
    m = int(count)
    assert m == want.numel()
    assert torch.equal(idx[:m].long(), want)
    assert torch.equal(xl[:m], xb[want]) and torch.equal(gl[:m], g[want])
    pad = (m + 127) // 128 * 128
    assert float(xl[m:pad].float().abs().max()) == 0.0 if pad > m else True


@pytest.mark.parametrize("fmt", ["fp16", "bf16"])
@pytest.mark.parametrize("N,frac", [(1, 1.0), (300, 0.1), (20_011, 0.0), (20_011, 0.1), (20_011, 0.3), (20_011, 1.0), (70_001, 0.05)])
def test_row_sparse_backward_equals_the_dense_backward(N, frac, fmt):
    """FusedHead(sparse_rows=True): no activations stored by the forward; the backward repeats the forward for the rows whose
    cotangent is non-zero and runs the data-gradient / parameter-gradient launches on those rows (their number stays on the
    device).  Every parameter gradient equals the dense pass to 1e-5 of the tensor's largest entry (same 16-bit operands, same
    products; the fp32 accumulation order over the rows differs), the dense pass's data gradients are EXACT zeros in the rows the
    sparse pass skips, and with an all-zero cotangent every gradient is exactly zero in both."""
    for name, net, head, xe in _nets(N):
        g = _sparse_cotangent(N, head.weight.shape[0], frac, mag=3e-8 if fmt == "fp16" else 1.0)
        grads = {}
        outs = {}
        for sparse in (False, True):
            fh = M.FusedHead(net.linear, head, xe.shape[1], net.skips[0], fmt, sparse_rows=sparse)
            for q in net.parameters():
                q.grad = None
            out = fh(xe)
            torch.autograd.backward([out], [g])
            grads[sparse] = [q.grad.clone() for q in net.parameters()]
            outs[sparse] = out.detach()
            if sparse:
                assert int(fh.last_live_count) == int((g != 0).any(1).sum())
        assert torch.equal(outs[True], outs[False])  # (the same forward kernel arithmetic, with and without the stores)
        for a, b, (n_, _q) in zip(grads[True], grads[False], net.named_parameters()):
            tol = 1e-5 * float(b.abs().max())
            assert float((a - b).abs().max()) <= tol, (name, fmt, N, frac, n_, float((a - b).abs().max()), tol)
            if frac == 0.0:
                assert float(a.abs().max()) == 0.0 and float(b.abs().max()) == 0.0
        # the dense pass's data gradients of the dead rows: exact zeros (what makes skipping them exact)
        pk = M.FusedHead(net.linear, head, xe.shape[1], net.skips[0], fmt)._packed()
        _o, (acts, masks) = M.forward(pk, xe, True)
        sc = M.grad_scale(g) if fmt == "fp16" else None
        dpre, _ = M.backward_data(pk, g, masks, sc, bias_sums=False)
        dead = ~(g != 0).any(1)
        if bool(dead.any()):
            assert float(dpre[:, dead].float().abs().max()) == 0.0
        for q in net.parameters():
            q.grad = None


def test_row_sparse_backward_takes_an_unaligned_cotangent_view():
    """A contiguous gradient VIEW at an odd float offset (a narrow of a torch.cat backward) is accepted."""
    name, net, head, xe = _nets(257)[0]
    fh = M.FusedHead(net.linear, head, xe.shape[1], net.skips[0], sparse_rows=True)
    out = fh(xe)
    flat = torch.randn(1 + out.numel(), device="cuda") * 1e-6
    g = flat[1:].view_as(out)
    assert g.data_ptr() % 16 != 0 and g.is_contiguous()
    torch.autograd.backward([out], [g])
    assert all(q.grad is not None and bool(torch.isfinite(q.grad).all()) for q in net.parameters())


@pytest.mark.parametrize("N", [1, 127, 4_097, 70_001])
def test_forward_epilogue_sigmoid_and_residual(N):
    """struct riggs_mlp_epilogue: the head's value through a sigmoid (network_utils.py:107) and joined with a residual,
    res_out = res_base + out * res_mask (skeleton_warp.py:152-161), inside the forward launch — against torch on the plain output
    of the same kernel (same products: the epilogue is the only difference)."""
    for name, net, head, xe in _nets(N):
        pk = M.FusedHead(net.linear, head, xe.shape[1], net.skips[0])._packed()
        xb = M.embed_bf16(pk, xe)
        plain, _ = M.forward(pk, xe, False, xb)
        g = torch.Generator(device="cuda").manual_seed(N)
        base = torch.randn(N, pk.out_ch, device="cuda", generator=g)
        mask = torch.rand(N, 1, device="cuda", generator=g)
        sig, _ = M.forward(pk, xe, False, xb, sigmoid=True)
        assert float((sig - torch.sigmoid(plain)).abs().max()) <= 2e-7
        for m in (None, mask):
            out, _, joined = M.forward(pk, xe, False, xb, res_base=base, res_mask=m)
            assert torch.equal(out, plain)
            want = base + (plain if m is None else plain * m)
            assert float((joined - want).abs().max()) <= 1e-6 * max(1.0, float(want.abs().max())), (name, N)
        out, _, joined = M.forward(pk, xe, False, xb, sigmoid=True, res_base=base, res_mask=mask)
        assert torch.equal(out, sig) and float((joined - (base + sig * mask)).abs().max()) <= 1e-6 * max(1.0, float(base.abs().max()))


@pytest.mark.parametrize("N,out_ch", [(1, 3), (5, 3), (1023, 23), (70_001, 3), (20_011, 23)])
def test_cotangent_kernel_matches_the_formula(N, out_ch):
    """riggs_mlp_cotangent: g_eff = (g + g_rows * row_mask[row]) * s (1 - s) + coef * out with every piece optional, the fp16 scale
    of g_eff and mean(out^2) — against torch (ragged sizes: the vector loop's tail; twice in a row: the scratch word is left zero)."""
    gen = torch.Generator(device="cuda").manual_seed(N + out_ch)
    r = lambda *s: torch.randn(*s, device="cuda", generator=gen)  # noqa: E731
    g, gr, out = r(N, out_ch) * 3e-7, r(N, out_ch) * 3e-7, r(N, out_ch) * 0.1
    mask = torch.rand(N, 1, device="cuda", generator=gen)
    sig = torch.sigmoid(r(N, out_ch) * 3)
    coef = torch.full((1,), 2e-6, device="cuda")
    msq = torch.zeros(1, device="cuda")
    cases = [(g, None, None, None, None), (g, None, None, sig, None), (None, gr, mask, None, out), (g, gr, mask, None, out),
             (g, gr, None, sig, out), (None, gr, None, None, None)]
    for a, b, m, s_, o in cases:
        for _ in range(2):
            ge, sc = M.cotangent(a, b, m, s_, o, coef if o is not None else None, msq if o is not None else None)
            want = torch.zeros(N, out_ch, device="cuda")
            if a is not None:
                want = want + a
            if b is not None:
                want = want + (b if m is None else b * m)
            if s_ is not None:
                want = want * (s_ * (1 - s_))
            if o is not None:
                want = want + coef * o
            assert float((ge - want).abs().max()) <= 1e-6 * float(want.abs().max()) + 1e-30
            amax = ge.abs().amax().clamp_min(1e-30)
            assert float(sc) / float(torch.exp2(torch.floor(torch.log2(1024.0 / amax)))) in (0.5, 1.0, 2.0)
            if o is not None:
                assert abs(float(msq) - float((o * o).mean())) <= 1e-5 * float((o * o).mean())
    assert all(int(w[0]) == 0 for w in M._L2_SCRATCH.values())


@pytest.mark.parametrize("N,frac", [(1, 1.0), (300, 0.1), (4_097, 0.0), (70_001, 0.1)])
def test_live_rows_with_the_sigmoid_factor_and_the_scale(N, frac):
    """riggs_mlp_live_rows(sigmoid_out, scale): the rows are tested and gathered on g * s (1 - s) (a saturated output — s exactly 1
    — kills its entry like torch's sigmoid_backward does), and the fp16 gradient scale of that cotangent comes out of the same two
    launches: equal to riggs_mlp_grad_scale on the materialised product."""
    name, net, head, xe = _nets(N)[0]
    pk = M.FusedHead(net.linear, head, xe.shape[1], net.skips[0])._packed()
    xb = M.embed_bf16(pk, xe)
    g = _sparse_cotangent(N, pk.out_ch, frac)
    gen = torch.Generator(device="cuda").manual_seed(3)
    sig = torch.sigmoid(torch.randn(N, pk.out_ch, device="cuda", generator=gen) * 4)
    if N > 4:
        sig[N // 3] = 1.0   # a saturated row: its cotangent is exactly zero whatever g holds
    ge = g * (sig * (1 - sig))
    idx, count, xl, gl, sc = M.live_rows(pk, g, xb, sig, want_scale=True)
    want = torch.nonzero((ge != 0).any(1)).flatten()
    m = int(count)
    assert m == want.numel() and torch.equal(idx[:m].long(), want)
    assert torch.equal(xl[:m], xb[want])
    assert float((gl[:m] - ge[want]).abs().max()) <= 1e-6 * float(ge.abs().max()) + 1e-38 if m else True
    assert float(sc) == float(M.grad_scale(ge))


@pytest.mark.parametrize("sparse", [False, True])
def test_fused_head_with_the_folds_equals_the_composition(sparse):
    """FusedHead(out_sigmoid=True) and FusedHead(res=(base, mask)): outputs and every parameter gradient equal the same head
    followed by the sigmoid / the torch residual join (the folds move elementwise work into the launches, nothing else) —
    to 1e-5 of each tensor's largest entry (2e-4 against torch.sigmoid's own arithmetic); the residual's cotangent reaches `base`
    untouched."""
    N = 20_011
    for name, net, head, xe in _nets(N):
        out_ch = head.weight.shape[0]
        gen = torch.Generator(device="cuda").manual_seed(11)
        g = _sparse_cotangent(N, out_ch, 0.2, mag=3e-7)
        base0 = torch.randn(N, out_ch, device="cuda", generator=gen)
        mask = torch.rand(N, 1, device="cuda", generator=gen)
        res = {}
        for fold in (False, True):
            for q in net.parameters():
                q.grad = None
            base = base0.clone().requires_grad_(True)
            if name == "WeightMLP":
                fh = M.FusedHead(net.linear, head, xe.shape[1], net.skips[0], sparse_rows=sparse, out_sigmoid=fold)
                if fold:
                    out = fh(xe)
                    torch.autograd.backward([out], [g])
                else:
                    # (the same sigmoid arithmetic as the kernel's, 1 / (1 + exp(-x)): torch.sigmoid's own form differs by an ulp
                    # of s, which the factor (1 - s) magnifies to ~2e-5 of the gradients — checked looser below)
                    o = fh(xe)
                    out = 1.0 / (1.0 + torch.exp(-o.detach()))
                    torch.autograd.backward([o], [g * (out * (1 - out))])
                res[fold] = (out.detach(), None, [q.grad.clone() for q in net.parameters()])
                if fold:
                    for q in net.parameters():
                        q.grad = None
                    torch.autograd.backward([torch.sigmoid(M.FusedHead(net.linear, head, xe.shape[1], net.skips[0], sparse_rows=sparse)(xe))], [g])
                    for a, (n_, q) in zip(res[True][2], net.named_parameters()):
                        assert float((a - q.grad).abs().max()) <= 2e-4 * float(q.grad.abs().max()) + 1e-30, (name, sparse, n_, "torch.sigmoid")
            else:
                fh = M.FusedHead(net.linear, head, xe.shape[1], net.skips[0], sparse_rows=sparse)
                if fold:
                    off, joined = fh(xe, res=(base, mask))
                else:
                    off = fh(xe)
                    joined = base + off * mask
                # (a direct cotangent on the offsets AND one through the join, as a caller with its own loss on the offsets has)
                torch.autograd.backward([off, joined], [0.5 * g, g])
                res[fold] = (joined.detach(), base.grad.clone(), [q.grad.clone() for q in net.parameters()])
        assert float((res[True][0] - res[False][0]).abs().max()) <= 1e-6 * max(1.0, float(res[False][0].abs().max()))
        if res[False][1] is not None:
            assert torch.equal(res[True][1], res[False][1])
        for a, b, (n_, _q) in zip(res[True][2], res[False][2], net.named_parameters()):
            assert float((a - b).abs().max()) <= 1e-5 * float(b.abs().max()) + 1e-30, (name, sparse, n_)
        for q in net.parameters():
            q.grad = None
