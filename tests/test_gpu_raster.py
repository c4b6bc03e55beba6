"""-m gpu: HIP rasterizer (through the C ABI) vs the CPU oracle on identical seeded inputs.

Bar (BASELINE.json north_star): bit-exact tile/depth ordering and indexing; rendered values and
dL/dparam within 1e-4 relative.  Full-size configs are checked through size-independent properties.
"""
import math

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import raster_ref as RR  # noqa: E402
from riggs_amd import synth  # noqa: E402
from riggs_amd.rasterizer import (GaussianRasterizer, rasterize_backward, rasterize_forward,  # noqa: E402
                                  saved_views, RasterArena)
from tests import gpu_util as U  # noqa: E402


def _grads_close(hip, ref, what, frac=1e-4):
    U.assert_close(hip.cpu().numpy().reshape(ref.shape), ref, what, U.REL_TOL, frac)


CASES = [
    # N, J, seed, H, W, scale, cam kwargs
    (2000, 8, 1235, 128, 128, 0.03, dict()),                       # small chain-like scene
    (10000, 8, 1235, 256, 256, 0.012, dict()),                     # BASELINE config C1 size
    (5000, 24, 7, 200, 333, 0.02, dict(azimuth_deg=90.0)),         # ragged image (not multiples of 16)
    (3001, 24, 9, 160, 160, 0.25, dict(radius=1.2)),               # camera inside the cloud: near culls, big splats; odd N
    (3000, 8, 17, 160, 160, 0.1, dict()),                          # rectangles of <= 16 and > 16 tiles side by side
    (30000, 24, 11, 96, 96, 0.05, dict()),                         # deep occlusion: pixels saturate, most Gaussians get no gradient
    (4000, 8, 13, 1168, 2064, 0.02, dict()),                       # 9 417 tiles (> 8 192: the per-thread tile arrays of bin_offsets overflow to their fallback)
]


@pytest.mark.parametrize("N,J,seed,H,W,scale,camkw", CASES)
def test_forward_backward_parity_vs_oracle(N, J, seed, H, W, scale, camkw):
    sc, act, cam = U.activated_scene(N, J, seed, H, W, scale=scale, **camkw)
    bg = [0.1, 0.3, 0.7]
    out_o, so = U.oracle_forward(act, cam, bg)
    color, radii, depth, alpha, s = U.hip_forward(act, cam, bg)
    v = saved_views(s)
    assert so.R > 0
    U.compare_forward_state(so, v, out_o, color, depth, alpha, radii)
    # backward with an L1-like image gradient plus depth/alpha cotangents
    g = torch.Generator().manual_seed(seed)
    gc = torch.sign(torch.rand(3, H, W, generator=g) - 0.5) / (3 * H * W)
    gd = torch.randn(1, H, W, generator=g) / (H * W)
    ga = torch.randn(1, H, W, generator=g) / (H * W)
    go = RR.backward(so, gc.numpy(), gd.numpy()[0], ga.numpy()[0])
    d = lambda t: t.cuda().contiguous()  # noqa: E731
    gh = rasterize_backward(s, d(act["means3D"]), d(act["shs"]), None, d(act["opacities"]), d(act["scales"]),
                            d(act["rotations"]), None, None, None, d(gc), d(gd), d(ga))
    g_means3D, g_means2D, g_sh, _, g_opac, g_scales, g_rots, _, _ = gh
    _grads_close(g_means2D, go["means2D"], "dL/dmeans2D")
    _grads_close(g_means3D, go["means3D"], "dL/dmeans3D")
    _grads_close(g_opac, go["opacities"], "dL/dopacity")
    _grads_close(g_scales, go["scales"], "dL/dscales")
    _grads_close(g_rots, go["rotations"], "dL/drotations")
    _grads_close(g_sh, go["shs"], "dL/dsh")
    if scale == 0.1:  # the case is there for the binning walk: lane groups that mix rectangles of <= 16 and > 16 tiles
        tt = v["tiles_touched"].cpu().numpy()
        assert (tt > 16).sum() > 100 and ((tt > 0) & (tt <= 16)).sum() > 100
    if N >= 30000:  # the case is there for the sparse-gradient paths: make sure it exercises them
        untouched = float((g_opac.reshape(-1) == 0).float().mean())
        assert 0.5 < untouched < 1.0, untouched


@pytest.mark.parametrize("deg", [0, 1, 2])
def test_lower_sh_degrees(deg):
    sc, act, cam = U.activated_scene(3000, 8, 21, 96, 96, scale=0.03)
    out_o, so = U.oracle_forward(act, cam, [0, 0, 0], sh_degree=deg)
    color, radii, depth, alpha, s = U.hip_forward(act, cam, [0, 0, 0], sh_degree=deg)
    U.compare_forward_state(so, saved_views(s), out_o, color, depth, alpha, radii)
    gc = torch.ones(3, 96, 96) / (96 * 96)
    go = RR.backward(so, gc.numpy(), None, None)
    d = lambda t: t.cuda().contiguous()  # noqa: E731
    gh = rasterize_backward(s, d(act["means3D"]), d(act["shs"]), None, d(act["opacities"]), d(act["scales"]),
                            d(act["rotations"]), None, None, None, d(gc), None, None)
    _grads_close(gh[2], go["shs"], "dL/dsh deg %d" % deg)
    _grads_close(gh[0], go["means3D"], "dL/dmeans3D deg %d" % deg)


def test_colors_precomp_and_cov3d_precomp():
    sc, act, cam = U.activated_scene(3000, 8, 33, 112, 80, scale=0.03)
    g = torch.Generator().manual_seed(1)
    colors = torch.rand(3000, 3, generator=g)
    cov6 = torch.from_numpy(U.oracle_forward(act, cam, [0, 0, 0])[1].cov3D.copy())
    out_o, so = U.oracle_forward(act, cam, [1, 1, 1], colors=colors, cov6=cov6, mod=1.0)
    color, radii, depth, alpha, s = U.hip_forward(act, cam, [1, 1, 1], colors=colors, cov6=cov6)
    U.compare_forward_state(so, saved_views(s), out_o, color, depth, alpha, radii)
    gc = torch.randn(3, 112, 80, generator=g) / (112 * 80)
    go = RR.backward(so, gc.numpy(), None, None)
    d = lambda t: t.cuda().contiguous()  # noqa: E731
    gh = rasterize_backward(s, d(act["means3D"]), None, d(colors), d(act["opacities"]), None, None, d(cov6), None,
                            None, d(gc), None, None)
    _grads_close(gh[3], go["colors_precomp"], "dL/dcolors_precomp")
    _grads_close(gh[7], go["cov3D_precomp"], "dL/dcov3D_precomp")
    _grads_close(gh[0], go["means3D"], "dL/dmeans3D")


def test_scale_modifier():
    sc, act, cam = U.activated_scene(2000, 8, 5, 64, 64, scale=0.03)
    out_o, so = U.oracle_forward(act, cam, [0, 0, 0], mod=0.5)
    color, radii, depth, alpha, s = U.hip_forward(act, cam, [0, 0, 0], mod=0.5)
    U.compare_forward_state(so, saved_views(s), out_o, color, depth, alpha, radii)


def test_empty_and_all_culled_inputs():
    cam = synth.look_at_camera(48, 48)
    st = U.settings_for(cam, [0.2, 0.4, 0.6])
    z = lambda *s: torch.zeros(*s, device="cuda")  # noqa: E731
    color, radii, depth, alpha, s = rasterize_forward(st, z(0, 3), z(0, 16, 3), None, z(0, 1), z(0, 3), z(0, 4), None)
    assert radii.numel() == 0 and torch.allclose(color[2], torch.full_like(color[2], 0.6)) and float(alpha.abs().max()) == 0
    # everything behind the camera
    sc, act, cam = U.activated_scene(500, 8, 3, 48, 48)
    act["means3D"] = act["means3D"] + torch.tensor([0.0, 0.0, 0.0])
    cam_far = synth.look_at_camera(48, 48, radius=4.0)
    act["means3D"] = act["means3D"] * 0 + cam_far.camera_center + torch.tensor([0.0, 0.0, 0.0])
    color, radii, depth, alpha, s = U.hip_forward(act, cam_far, [0.2, 0.4, 0.6])
    assert int(radii.max()) == 0 and saved_views(s)["R"] == 0
    assert torch.allclose(color[0], torch.full_like(color[0], 0.2))


def test_rasterizer_module_errors_and_autograd_surface():
    sc, act, cam = U.activated_scene(1500, 8, 11, 64, 64, scale=0.03)
    st = U.settings_for(cam, [0, 0, 0])
    r = GaussianRasterizer(raster_settings=st)
    d = lambda t: t.cuda().contiguous().requires_grad_(True)  # noqa: E731
    m3, op, sc_, ro, sh = d(act["means3D"]), d(act["opacities"]), d(act["scales"]), d(act["rotations"]), d(act["shs"])
    m2 = torch.zeros_like(m3, requires_grad=True)
    with pytest.raises(Exception, match="SHs or precomputed colors"):
        r(means3D=m3, means2D=m2, opacities=op, shs=None, colors_precomp=None, scales=sc_, rotations=ro)
    with pytest.raises(Exception, match="scale/rotation pair or precomputed 3D covariance"):
        r(means3D=m3, means2D=m2, opacities=op, shs=sh, scales=sc_, rotations=None)
    color, radii, depth, alpha = r(means3D=m3, means2D=m2, opacities=op, shs=sh, colors_precomp=None, scales=sc_,
                                   rotations=ro, cov3D_precomp=None)
    assert color.shape == (3, 64, 64) and depth.shape == (1, 64, 64) and alpha.shape == (1, 64, 64)
    assert radii.dtype == torch.int32 and radii.shape == (1500,)
    (color.sum() + 0.1 * depth.sum()).backward()
    for t in (m3, m2, op, sc_, ro, sh):
        assert t.grad is not None and torch.isfinite(t.grad).all()
    assert float(m2.grad[:, 2].abs().max()) == 0.0 and float(m2.grad[:, :2].abs().max()) > 0
    with pytest.raises(Exception):
        r(means3D=act["means3D"], means2D=None, opacities=act["opacities"], shs=act["shs"], scales=act["scales"],
          rotations=act["rotations"])  # CPU tensors: the product path is GPU-only and says so


def test_arena_mode_matches_sync_mode_and_recovers_from_overflow():
    sc, act, cam = U.activated_scene(4000, 8, 2, 128, 128, scale=0.03)
    d = lambda t: t.cuda().contiguous()  # noqa: E731
    st = U.settings_for(cam, [0, 0, 0])
    args = (d(act["means3D"]), d(act["shs"]), None, d(act["opacities"]), d(act["scales"]), d(act["rotations"]), None)
    ref = rasterize_forward(st, *args)
    arena = RasterArena(min_capacity=16)
    a1 = rasterize_forward(st, *args, arena=arena)   # first call synchronises and sizes the arena
    a2 = rasterize_forward(st, *args, arena=arena)   # second call: no host sync, padded sort
    # (bitwise equality holds in the reproducible mode only: by default a long list may be composited by several workgroups
    # whose partial results are combined — which ones depends on timing, and the combination rounds differently)
    same = lambda x, y: float((x - y).abs().max()) <= 2e-6 * max(1.0, float(x.abs().max()))  # noqa: E731
    assert same(ref[0], a1[0]) and same(ref[0], a2[0])
    assert torch.equal(ref[1], a2[1])
    assert arena.resolve() and arena.last_R == saved_views(ref[4])["R"]
    # force a too-small arena: the frame is flagged when its counters are consumed, then the arena regrows
    arena.capacity, arena.binning, arena.last_R, arena.min_capacity = 0, None, 10, 16
    a3 = rasterize_forward(st, *args, arena=arena)
    assert a3[0].shape == ref[0].shape  # memory-safe, image undefined
    from riggs_amd._lib import RiggsHipError
    with pytest.raises(RiggsHipError, match="overflowed"):
        rasterize_forward(st, *args, arena=arena)
    a4 = rasterize_forward(st, *args, arena=arena)
    assert arena.resolve() and same(ref[0], a4[0])


def test_arena_follows_a_growing_scene_and_image():
    """The arena also holds tables sized by the number of Gaussians and of tiles: at an unchanged instance capacity a scene with
    more Gaussians or a larger image must get a larger arena (it used to be sized by the capacity alone)."""
    d = lambda t: t.cuda().contiguous()  # noqa: E731
    arena = RasterArena(min_capacity=1 << 21)  # (plenty for both frames: the capacity never forces a regrow)
    same = lambda x, y: float((x - y).abs().max()) <= 2e-6 * max(1.0, float(x.abs().max()))  # noqa: E731
    sizes = []
    for n, hw in ((2000, 64), (60_000, 512), (2000, 64)):
        sc, act, cam = U.activated_scene(n, 8, 5, hw, hw, scale=0.01)
        st = U.settings_for(cam, [0, 0, 0])
        args = (d(act["means3D"]), d(act["shs"]), None, d(act["opacities"]), d(act["scales"]), d(act["rotations"]), None)
        ref = rasterize_forward(st, *args)
        for _ in range(2):  # (the second call is the one without the host synchronisation)
            out = rasterize_forward(st, *args, arena=arena)
            assert same(ref[0], out[0]) and torch.equal(ref[1], out[1])
        assert arena.resolve()
        sizes.append(arena.binning.numel())
    assert sizes[1] > sizes[0] and sizes[2] == sizes[1] and arena.capacity == 1 << 21


@pytest.mark.parametrize("N,J,H,W", [(150_000, 24, 800, 800), (300_000, 32, 800, 800)])
def test_full_size_properties(N, J, H, W):
    """BASELINE configs C2 / C3 at full size: size-independent properties instead of the oracle."""
    sc, act, cam = U.activated_scene(N, J, 1234 + 2, H, W)
    U.check_full_size_properties(act, cam)


@pytest.mark.parametrize("H,W", [(2560, 2560), (2160, 3840)])
def test_large_tile_grids(H, W):
    """Grids beyond 4096 tiles take the two-level (grouped) binning: 25 600 tiles (2560 x 2560 px, the most the one-level
    sort's per-workgroup tile table ever fitted) and 4K UHD (3840 x 2160: 32 400 tiles — upstream passes any H, W:
    gaussian_renderer/__init__.py:57-70).  A handful of Gaussians must match the oracle bit for bit in ordering, with tile ids
    far above 16 000, and the gradients agree."""
    sc, act, cam = U.activated_scene(300, 8, 41, H, W, scale=0.05)
    out_o, so = U.oracle_forward(act, cam, [0.0, 0.1, 0.2])
    color, radii, depth, alpha, s = U.hip_forward(act, cam, [0.0, 0.1, 0.2])
    U.compare_forward_state(so, saved_views(s), out_o, color, depth, alpha, radii)
    assert int(saved_views(s)["tile_keys"].max()) > 16_000  # instances land on tiles all over the grid
    gc = torch.ones(3, H, W) / (3 * H * W)
    go = RR.backward(so, gc.numpy(), None, None)
    d = lambda t: t.cuda().contiguous()  # noqa: E731
    gh = rasterize_backward(s, d(act["means3D"]), d(act["shs"]), None, d(act["opacities"]), d(act["scales"]),
                            d(act["rotations"]), None, None, None, d(gc), None, None)
    # (300 Gaussians: a handful of elements that are differences of large cancelling terms carry the atomics' reordering noise
    # at their neighbours' magnitude — the max-norm bar is the check here)
    U.assert_close(gh[0].cpu().numpy(), go["means3D"], "dL/dmeans3D (%d tiles)" % (((H + 15) // 16) * ((W + 15) // 16)), U.REL_TOL, 1e-4, 0.05)
    U.assert_close(gh[4].cpu().numpy().reshape(go["opacities"].shape), go["opacities"], "dL/dopacity (large grid)", U.REL_TOL, 1e-4, 0.05)


def test_tile_grids_beyond_the_work_list_packing_are_rejected():
    """65 535 tiles is the limit: more is an error, not a mis-render."""
    from riggs_amd._lib import RiggsHipError
    H, W = 4112, 4096  # 257 x 256 = 65 792 tiles
    sc, act, cam = U.activated_scene(50, 8, 41, H, W, scale=0.05)
    with pytest.raises(RiggsHipError, match="image too large"):
        U.hip_forward(act, cam, [0, 0, 0])


def test_loss_on_depth_or_alpha_only_backpropagates():
    """set_materialize_grads(False): a loss that ignores the colour image hands grad_color = None to the backward."""
    sc, act, cam = U.activated_scene(1500, 8, 11, 64, 64, scale=0.03)
    r = GaussianRasterizer(raster_settings=U.settings_for(cam, [0, 0, 0]))
    d = lambda t: t.cuda().contiguous().requires_grad_(True)  # noqa: E731
    m3, op, sc_, ro, sh = d(act["means3D"]), d(act["opacities"]), d(act["scales"]), d(act["rotations"]), d(act["shs"])
    color, radii, depth, alpha = r(means3D=m3, means2D=torch.zeros_like(m3, requires_grad=True), opacities=op, shs=sh,
                                   scales=sc_, rotations=ro)
    (0.3 * depth.sum() + alpha.sum()).backward()
    out_o, so = U.oracle_forward(act, cam, [0, 0, 0])
    go = RR.backward(so, np.zeros((3, 64, 64), np.float32), np.full((64, 64), 0.3, np.float32), np.ones((64, 64), np.float32))
    _grads_close(m3.grad, go["means3D"], "depth/alpha-only dL/dmeans3D")
    _grads_close(op.grad, go["opacities"], "depth/alpha-only dL/dopacity")
    assert float(sh.grad.abs().max()) == 0.0


def test_overflowed_frame_backpropagates_exact_zeros():
    """Device-side guard: when the instance arena overflowed, the frame's gradients are exact zeros — an
    optimizer step queued behind the backward (captured graph) never consumes gradients of a truncated image."""
    sc, act, cam = U.activated_scene(4000, 8, 2, 128, 128, scale=0.03)
    d = lambda t: t.cuda().contiguous()  # noqa: E731
    st = U.settings_for(cam, [0, 0, 0])
    args = (d(act["means3D"]), d(act["shs"]), None, d(act["opacities"]), d(act["scales"]), d(act["rotations"]), None)
    arena = RasterArena(min_capacity=16)
    rasterize_forward(st, *args, arena=arena)
    arena.capacity, arena.binning, arena.last_R, arena.min_capacity = 0, None, 10, 16  # next frame: far too small an arena
    out = rasterize_forward(st, *args, arena=arena)
    gc = torch.ones(3, 128, 128, device="cuda")
    g = rasterize_backward(out[4], *args, None, None, gc, None, None)
    assert int(out[4].counters[1]) == 1
    for t in g:
        if t is not None:
            assert float(t.abs().max()) == 0.0
    from riggs_amd._lib import RiggsHipError
    with pytest.raises(RiggsHipError, match="overflowed"):
        arena.resolve()


def test_depth_sort_takes_two_passes_when_the_depths_share_their_top_byte_and_three_otherwise():
    """The depth sort skips its third 12-bit pass when every visible depth key has the same top byte (all depths inside
    [2, 8), [0.5, 2), ...): counters[2] records which path ran.  Both orders are checked bit for bit against the oracle."""
    for radius, want in ((4.0, 0), (1.6, 1)):   # z in [3, 5] -> two passes; z in (0.2, 2.6] straddles 0.5 and 2 -> three
        sc, act, cam = U.activated_scene(6000, 24, 19, 144, 176, scale=0.03, radius=radius)
        out_o, so = U.oracle_forward(act, cam, [0, 0, 0])
        color, radii, depth, alpha, s = U.hip_forward(act, cam, [0, 0, 0])
        assert int(s.counters[2]) == want, (radius, int(s.counters[2]))
        z = so.depths[so.radii > 0]
        assert (z.view(np.uint32) >> 24).min() != (z.view(np.uint32) >> 24).max() if want else True
        U.compare_forward_state(so, saved_views(s), out_o, color, depth, alpha, radii)


def test_depth_sort_of_a_million_keys_uses_the_large_chunks():
    """From a million Gaussians on the depth sort works on chunks of 8192 keys instead of 2048 (launch_depth_sort): all three
    passes of that variant, bit for bit against the oracle (tests/test_gpu_configs.py: C5 covers its two-pass path)."""
    sc, act, cam = U.activated_scene(1_050_000, 8, 23, 96, 128, scale=0.004, radius=1.6)
    out_o, so = U.oracle_forward(act, cam, [0, 0, 0])
    color, radii, depth, alpha, s = U.hip_forward(act, cam, [0, 0, 0])
    assert int(s.counters[2]) == 1
    U.compare_forward_state(so, saved_views(s), out_o, color, depth, alpha, radii)


def test_ordered_reduction_mode_gives_bitwise_reproducible_gradients():
    """riggs_raster_cfg.deterministic: every tile instance writes its gradient row, a second kernel sums each
    Gaussian's rows in ascending tile order — no float atomics, so two runs agree BIT FOR BIT (the default path only to
    rounding), and the result equals the atomics path / the oracle within the usual tolerance."""
    from riggs_amd import rasterizer as RZ
    sc, act, cam = U.activated_scene(20000, 24, 11, 160, 176, scale=0.05)   # deep: hundreds of instances per Gaussian sum
    bg = [0.1, 0.2, 0.3]
    g = torch.Generator().manual_seed(4)
    gc = (torch.sign(torch.rand(3, 160, 176, generator=g) - 0.5) / (3 * 160 * 176)).cuda()
    gd = (torch.randn(1, 160, 176, generator=g) / (160 * 176)).cuda()
    d = lambda t: t.cuda().contiguous()  # noqa: E731
    args = (d(act["means3D"]), d(act["shs"]), None, d(act["opacities"]), d(act["scales"]), d(act["rotations"]), None)

    def grads():
        out = rasterize_forward(U.settings_for(cam, bg), *args)
        return rasterize_backward(out[4], *args, None, None, gc, gd, None)
    plain = [grads(), grads()]
    RZ.set_ordered_backward(True)
    try:
        ordered = [grads(), grads()]
    finally:
        RZ.set_ordered_backward(False)
    names = "means3D means2D sh colors opac scales rots cov dscaling".split()
    differs = 0
    for nm, a, b, p, q in zip(names, ordered[0], ordered[1], plain[0], plain[1]):
        if a is None:
            continue
        assert torch.equal(a, b), "ordered mode not reproducible: " + nm
        differs += int(not torch.equal(p, q))
        U.assert_close(a.cpu().numpy(), p.cpu().numpy(), "ordered vs atomics dL/d" + nm, 2e-5, 1e-4)
    assert differs > 0, "the atomics path happened to be reproducible here: the scene does not exercise the point of the mode"
    out_o, so = U.oracle_forward(act, cam, bg)
    go = RR.backward(so, gc.cpu().numpy(), gd.cpu().numpy()[0], None)
    _grads_close(ordered[0][0], go["means3D"], "ordered dL/dmeans3D vs oracle")
    _grads_close(ordered[0][4], go["opacities"], "ordered dL/dopacity vs oracle")
