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


def _grads_close(hip, ref, what, frac=1e-5):
    """Every gradient element within 1e-4 of the tensor's largest entry, up to ``frac`` of the elements (round 3: 1e-4; the
    observed share over the whole suite is ZERO since the threshold flips are gone — profiles/*_parity_stats.json — so the
    allowance is a tenth of what it was; the per-element bound stays: sums of cancelling terms sit at 1.4e-3 of it)."""
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


@pytest.mark.parametrize("which", ["precomp", "deg0", "deg1", "deg2", "case3", "case5"])
def test_one_wave_per_block_backward_on_the_optional_inputs(which):
    """riggs_set_option("preprocess_bwd_lean", 1): the one-wave-per-block form of the per-Gaussian backward (the default only
    with sparse gradient rows) forced onto the paths the random sweep does not draw — precomputed colours and covariances, the
    lower SH degrees with their shorter rows, a camera inside the cloud, the deep-occlusion scene whose blocks list nothing —
    through the very assertions of the tests above."""
    from riggs_amd import _lib as L
    try:
        L.set_option("preprocess_bwd_lean", 1)
        if which == "precomp":
            test_colors_precomp_and_cov3d_precomp()
        elif which.startswith("deg"):
            test_lower_sh_degrees(int(which[3:]))
        else:
            test_forward_backward_parity_vs_oracle(*CASES[int(which[4:])])
    finally:
        L.set_option("preprocess_bwd_lean", -1)


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
    # (bitwise equality frame after frame holds with cfg.deterministic or while no tile is composited wide — see
    # test_wide_forward_blocks_against_the_oracle; this small scene never qualifies, the bound is kept loose on purpose)
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


def test_wide_forward_blocks_against_the_oracle():
    """The forward composites the tiles whose walk went "fwd_wide_min" instances deep in the PREVIOUS frame of the same arena
    with 32 lanes per pixel on 4 x 2 pixel blocks (render.hip: fw_block<32>: hand-written DPP scans and folds, checkpoints from
    lanes LEAD + k, ticket == 31) — a path no fresh-arena comparison reaches.  Here: a translucent scene (no pixel saturates, so
    a walk is as deep as its list), the gate lowered to 256 instances, two frames through one arena; the SECOND frame must have
    wide tiles, and its image / n_contrib / final_T, the checkpoints the backward consumes (through every gradient) and the
    bit-exact ordering state are compared with the oracle, and the image with the history-free frame (same values to ~1e-6:
    another fold order).  With cfg.deterministic the history is ignored: no wide tile, bitwise equal frames."""
    from riggs_amd import _lib as L
    from riggs_amd.rasterizer import set_ordered_backward
    N, H, W = 30_000, 160, 208
    sc, act, cam = U.activated_scene(N, 8, 41, H, W, scale=0.03)
    act["opacities"] = act["opacities"] * 0.02  # translucent: T stays > 1e-4 through thousands of instances
    bg = [0.1, 0.3, 0.7]
    out_o, so = U.oracle_forward(act, cam, bg)
    assert int(so.n_contrib.max()) >= 1024, "the scene must walk deep for this test"
    d = lambda t: t.cuda().contiguous()  # noqa: E731
    st = U.settings_for(cam, bg, debug=True)
    args = (d(act["means3D"]), d(act["shs"]), None, d(act["opacities"]), d(act["scales"]), d(act["rotations"]), None)
    assert L.get_option("fwd_wide_min") == 4096 and L.get_option("fwd_wide_tiles") == 256
    L.set_option("fwd_wide_min", 256)
    try:
        arena = RasterArena(min_capacity=16)
        f1 = rasterize_forward(st, *args, arena=arena)
        v1 = saved_views(f1[4])
        assert int(v1["fwd_ctr"][1]) == 0, "a fresh arena has no history: no wide tile"
        img1 = f1[0].clone()
        f2 = rasterize_forward(st, *args, arena=arena)
        color, radii, depth, alpha, s = f2
        v2 = saved_views(s)
        n_wide, n_tiles = int(v2["fwd_ctr"][1]), int(v2["fwd_ctr"][0])
        assert n_wide >= 8 and n_wide < n_tiles, (n_wide, n_tiles)  # both block shapes in one launch
        U.compare_forward_state(so, v2, out_o, color, depth, alpha, radii)
        assert float((color - img1).abs().max()) <= 2e-6 * max(1.0, float(img1.abs().max()))
        # the backward reads the wide blocks' checkpoints and n_contrib
        g = torch.Generator().manual_seed(3)
        gc = torch.sign(torch.rand(3, H, W, generator=g) - 0.5) / (3 * H * W)
        gd = torch.randn(1, H, W, generator=g) / (H * W)
        ga = torch.randn(1, H, W, generator=g) / (H * W)
        go = RR.backward(so, gc.numpy(), gd.numpy()[0], ga.numpy()[0])
        gh = rasterize_backward(s, *args[:2], None, *args[3:6], None, None, None, d(gc), d(gd), d(ga))
        for got, name in ((gh[1], "means2D"), (gh[0], "means3D"), (gh[4], "opacities"), (gh[5], "scales"), (gh[6], "rotations"), (gh[2], "shs")):
            _grads_close(got, go[name], "wide tiles: dL/d" + name)
        # a third frame: the history now comes from a frame that had wide tiles (their depth is written by the 32nd block)
        f3 = rasterize_forward(st, *args, arena=arena)
        assert int(saved_views(f3[4])["fwd_ctr"][1]) == n_wide and torch.equal(f3[0], color)
        # the deterministic configuration ignores the history: frame after frame bitwise equal, never wide
        set_ordered_backward(True)
        try:
            arena_d = RasterArena(min_capacity=16)
            frames = [rasterize_forward(st, *args, arena=arena_d) for _ in range(3)]
            for f in frames:
                assert int(saved_views(f[4])["fwd_ctr"][1]) == 0
            assert torch.equal(frames[0][0], frames[2][0]) and torch.equal(frames[1][0], frames[2][0])
            assert torch.equal(frames[0][0], img1)  # ... and equal to the history-free frame of the default configuration
        finally:
            set_ordered_backward(False)
        # "fwd_wide_tiles" = 0 turns the wide form off in the default configuration as well
        L.set_option("fwd_wide_tiles", 0)
        f4 = rasterize_forward(st, *args, arena=arena)
        assert int(saved_views(f4[4])["fwd_ctr"][1]) == 0 and torch.equal(f4[0], img1)
    finally:
        L.set_option("fwd_wide_min", -1)
        L.set_option("fwd_wide_tiles", -1)
    assert L.get_option("fwd_wide_min") == 4096 and L.get_option("fwd_wide_tiles") == 256


def test_arena_history_does_not_survive_a_new_scene_size_or_a_recycled_block():
    """The walk history is trusted when a stamp follows it; the stamp is a function of the tile AND Gaussian counts and
    RasterArena zero-fills the words when it allocates or when (capacity, N, H, W) change — so a frame never composites wide
    because of ANOTHER scene's history (same image, other N, through one arena; and a new arena on a recycled allocator block)."""
    from riggs_amd import _lib as L
    H, W = 160, 208
    d = lambda t: t.cuda().contiguous()  # noqa: E731
    L.set_option("fwd_wide_min", 256)
    try:
        def frame(n, arena, seed=41):
            sc, act, cam = U.activated_scene(n, 8, seed, H, W, scale=0.03)
            st = U.settings_for(cam, [0, 0, 0])
            a = (d(act["means3D"]), d(act["shs"]), None, d(act["opacities"] * 0.02), d(act["scales"]), d(act["rotations"]), None)
            return rasterize_forward(st, *a, arena=arena)
        arena = RasterArena(min_capacity=1 << 21)
        frame(30_000, arena)
        assert int(saved_views(frame(30_000, arena)[4])["fwd_ctr"][1]) > 0    # history in use
        assert int(saved_views(frame(29_000, arena)[4])["fwd_ctr"][1]) == 0   # another N in the same arena: no history
        assert int(saved_views(frame(29_000, arena)[4])["fwd_ctr"][1]) > 0    # ... until this size has its own
        del arena
        arena1 = RasterArena(min_capacity=1 << 21)
        frame(29_000, arena1)
        assert int(saved_views(frame(29_000, arena1)[4])["fwd_ctr"][1]) > 0
        nbytes, where = arena1.binning.numel(), arena1.binning.data_ptr()
        del arena1
        arena2 = RasterArena(min_capacity=1 << 21)   # same size: the caching allocator hands the block back, stamp and all
        out = frame(29_000, arena2)
        assert arena2.binning.numel() == nbytes
        assert int(saved_views(out[4])["fwd_ctr"][1]) == 0, "recycled block at %s" % ("the same address" if arena2.binning.data_ptr() == where else "another address")
    finally:
        L.set_option("fwd_wide_min", -1)


@pytest.mark.parametrize("case", [1, 4, 6])
def test_two_level_tile_sort_forced_where_both_sorts_fit(case):
    """riggs_set_option("bin_grouped", 1): the two-level tile sort on scenes the direct counting sort would take — the same
    list, ranges and images bit for bit against the oracle's key sort — and back."""
    from riggs_amd import _lib as L
    N, J, seed, H, W, scale, camkw = CASES[case]
    sc, act, cam = U.activated_scene(N, J, seed, H, W, scale=scale, **camkw)
    out_o, so = U.oracle_forward(act, cam, [0, 0, 0])
    assert L.get_option("bin_grouped") == -1
    imgs = []
    try:
        for forced in (1, 0):
            L.set_option("bin_grouped", forced)
            color, radii, depth, alpha, s = U.hip_forward(act, cam, [0, 0, 0])
            U.compare_forward_state(so, saved_views(s), out_o, color, depth, alpha, radii)
            imgs.append(color)
    finally:
        L.set_option("bin_grouped", -1)
    assert torch.equal(imgs[0], imgs[1])
    from riggs_amd._lib import RiggsHipError
    with pytest.raises(RiggsHipError, match="bin_grouped"):
        L.set_option("bin_grouped", 2)
    with pytest.raises(RiggsHipError, match="unknown option"):
        L.set_option("no_such_option", 1)


def test_render_rejects_a_binning_arena_that_is_too_small():
    """riggs_raster_render is told the arena's size and refuses one smaller than riggs_raster_binning_bytes for its arguments
    (a C caller that kept an arena across a change of N / H / W would otherwise write out of bounds)."""
    import ctypes as C
    from riggs_amd import _lib as L
    sc, act, cam = U.activated_scene(2000, 8, 5, 64, 64, scale=0.03)
    color, radii, depth, alpha, s = U.hip_forward(act, cam, [0, 0, 0])
    lib = L.lib()
    need = lib.riggs_raster_binning_bytes(s.cap, s.N, s.H, s.W)
    assert s.binning.numel() >= need
    rc = lib.riggs_raster_render(C.byref(s.cfg), s.geom.data_ptr(), s.binning.data_ptr(), s.cap, need - 1, s.img.data_ptr(),
                                 color.data_ptr(), depth.data_ptr(), alpha.data_ptr(), s.counters.data_ptr(), L.stream_ptr())
    assert rc != 0 and b"smaller than riggs_raster_binning_bytes" in lib.riggs_last_error()
    # the size is the same whichever tile sort runs (the larger layout is reserved): forcing one never invalidates an arena
    sizes = set()
    for forced in (-1, 0, 1):
        L.set_option("bin_grouped", forced)
        sizes.add(lib.riggs_raster_binning_bytes(1 << 20, 499_999, 2560, 2560))
        sizes.add(lib.riggs_raster_binning_bytes(1 << 20, 500_000, 2560, 2560) - (500_000 - 499_999) * 8)
    L.set_option("bin_grouped", -1)
    assert len(sizes) <= 2 and max(sizes) - min(sizes) <= 4096, sizes  # (monotonic up to alignment in N across the switch)


@pytest.mark.parametrize("case,deg,split,glue", [(1, 3, True, True), (3, 3, True, False), (2, 2, False, False), (0, 0, True, True),
                                                 (4, 1, False, True), (6, 3, True, False)])
def test_colour_job_of_the_tile_sort_equals_the_colours_of_preprocess(case, deg, split, glue):
    """riggs_set_option("color_side_jobs"): the SH colours evaluated by extra workgroups of the tile sort's scatter launch
    (csrc/color_job.h; the default wherever the direct tile sort runs) against preprocess_fwd evaluating them itself: colours,
    clamp bits, images bit for bit — and both against the oracle.  Split (_features_dc / _features_rest: direct-to-LDS rows)
    and single-tensor layouts, every degree, with and without the fused deformation residual (the view direction is taken
    from the DEFORMED mean), ragged last blocks (odd N), 9 417 tiles (one wave per workgroup of the scatter launch)."""
    from riggs_amd import _lib as L
    N, J, seed, H, W, scale, camkw = CASES[case]
    sc, act, cam = U.activated_scene(N, J, seed, H, W, scale=scale, **camkw)
    M = (deg + 1) ** 2
    d = lambda t: t.cuda().contiguous()  # noqa: E731
    gen = torch.Generator().manual_seed(seed)
    dx = 0.02 * torch.randn(N, 3, generator=gen)
    means = act["means3D"] + dx if glue else act["means3D"]
    shs = act["shs"][:, :M].contiguous()
    act_o = dict(act, means3D=means, shs=shs)
    out_o, so = U.oracle_forward(act_o, cam, [0.1, 0.2, 0.3], sh_degree=deg)
    st = U.settings_for(cam, [0.1, 0.2, 0.3], deg, 1.0, True)
    assert L.get_option("color_side_jobs") == 1
    res = []
    try:
        for on in (1, 0):
            L.set_option("color_side_jobs", on)
            a_shs = d(shs[:, :1]) if split else d(shs)
            a_rest = d(shs[:, 1:]) if split else None
            if glue:  # raw parameters in, activations in the kernel
                color, radii, depth, alpha, s = rasterize_forward(
                    st, d(sc["xyz"]), a_shs, None, d(sc["opacity"]), d(sc["scaling"]), d(sc["rotation"]), None, d_xyz=d(dx),
                    glue=True, shs_rest=a_rest)
            else:
                color, radii, depth, alpha, s = rasterize_forward(st, d(act["means3D"]), a_shs, None, d(act["opacities"]),
                                                                  d(act["scales"]), d(act["rotations"]), None, shs_rest=a_rest)
            v = saved_views(s)
            res.append((color, v["rgb"].clone(), v["clamped"].clone(), radii))
    finally:
        L.set_option("color_side_jobs", 1)
    vis = (res[0][3] > 0)
    assert int(vis.sum()) > 100
    assert torch.equal(res[0][0], res[1][0])
    assert torch.equal(res[0][1][vis], res[1][1][vis]) and torch.equal(res[0][2][vis], res[1][2][vis])
    rgb_o = so.rgb[vis.cpu().numpy()]
    U.assert_close(res[0][1].cpu().numpy()[vis.cpu().numpy(), :3], rgb_o, "rgb (colour job)", 1e-5 if not glue else 1e-4)
    if not glue:
        U.assert_close(res[0][0].cpu().numpy(), out_o["color"], "image (colour job)", U.REL_TOL, 1e-4)


@pytest.mark.parametrize("case,opacity_scale", [(1, 1.0), (2, 0.05), (5, 1.0), (4, 0.3)])
def test_tight_lists_are_the_canonical_lists_minus_the_instances_that_cannot_reach_a_pixel(case, opacity_scale):
    """cfg.tight_lists (RasterArena(tight_lists=True) / rasterize_forward(tight_lists=True)): the tile rectangle of a Gaussian is cut down by its alpha >= 1/255 box.
    Contract: radii unchanged; the instance list = the CANONICAL list (the oracle's key sort) with the instances outside the box
    removed, order kept — bit for bit, the predicate evaluated here in float32 from the extents the kernel stored; no instance
    that reaches alpha >= 1/255 at a pixel centre of its tile (float64 truth) is ever dropped; image, depth, alpha and every
    gradient equal the canonical ones (the dropped instances fail the alpha test everywhere in their tile)."""
    from riggs_amd import rasterizer as RZ
    N, J, seed, H, W, scale, camkw = CASES[case]
    sc, act, cam = U.activated_scene(N, J, seed, H, W, scale=scale, **camkw)
    act["opacities"] = act["opacities"] * opacity_scale
    bg = [0.1, 0.3, 0.7]
    out_o, so = U.oracle_forward(act, cam, bg)
    c_color, c_radii, c_depth, c_alpha, c_s = U.hip_forward(act, cam, bg)
    assert c_s.cfg.tight_lists == 0
    try:
        color, radii, depth, alpha, s = U.hip_forward(act, cam, bg, tight_lists=True)
        assert s.cfg.tight_lists == 1
        v = saved_views(s)
        g = torch.Generator().manual_seed(seed)
        gc = torch.sign(torch.rand(3, H, W, generator=g) - 0.5) / (3 * H * W)
        gd = torch.randn(1, H, W, generator=g) / (H * W)
        ga = torch.randn(1, H, W, generator=g) / (H * W)
        d = lambda t: t.cuda().contiguous()  # noqa: E731
        gh = rasterize_backward(s, d(act["means3D"]), d(act["shs"]), None, d(act["opacities"]), d(act["scales"]), d(act["rotations"]),
                                None, None, None, d(gc), d(gd), d(ga))
    finally:
        pass
    assert torch.equal(radii, c_radii) and np.array_equal(radii.cpu().numpy(), so.radii)
    gx = (W + 15) // 16
    xyd, rgbh = v["xyd"].cpu().numpy(), v["rgb"].cpu().numpy()
    px, py, hx, hy = xyd[:, 0], xyd[:, 1], xyd[:, 3], rgbh[:, 3]
    pl = so.point_list.astype(np.int64)
    tile = (so.keys >> np.uint64(32)).astype(np.int64)
    tx, ty = tile % gx, tile // gx
    f32, inv16 = np.float32, np.float32(1.0 / 16.0)
    with np.errstate(invalid="ignore", over="ignore"):
        x_lo, x_hi = np.floor((px - hx) * inv16), np.floor((px + hx) * inv16)
        y_lo, y_hi = np.floor((py - hy) * inv16), np.floor((py + hy) * inv16)
    keep = (hx[pl] >= 0) & (tx >= x_lo[pl]) & (tx <= x_hi[pl]) & (ty >= y_lo[pl]) & (ty <= y_hi[pl])
    # a rectangle that the box empties in ONE direction is emptied altogether
    want_pl, want_tile = pl[keep], tile[keep]
    assert v["R"] == int(keep.sum()), (v["R"], int(keep.sum()), so.R)
    assert np.array_equal(v["point_list"].cpu().numpy().astype(np.int64), want_pl), "tight list != filtered canonical list"
    assert np.array_equal(v["tile_keys"].cpu().numpy().astype(np.int64), want_tile)
    assert np.array_equal(v["tiles_touched"].cpu().numpy().astype(np.int64), np.bincount(want_pl, minlength=N))
    T = gx * ((H + 15) // 16)
    cnt = np.bincount(want_tile, minlength=T)
    rg = v["ranges"].cpu().numpy().astype(np.int64)
    assert np.array_equal((rg[:, 1] - rg[:, 0]), cnt)
    removed = 1.0 - keep.mean()
    assert removed > (0.02 if opacity_scale == 1.0 else 0.10), removed
    # nothing alive is dropped: float64 truth over the pixel centres of every canonical instance's tile
    co = so.conic_o[pl].astype(np.float64)
    cx, cy = so.xy[pl, 0].astype(np.float64), so.xy[pl, 1].astype(np.float64)
    alive = np.zeros(pl.shape[0], bool)
    for a in range(0, pl.shape[0], 100_000):
        b = min(pl.shape[0], a + 100_000)
        qx = (tx[a:b, None] * 16 + np.arange(16)[None]).astype(np.float64)
        qy = (ty[a:b, None] * 16 + np.arange(16)[None]).astype(np.float64)
        dx, dy = (cx[a:b, None] - qx)[:, None, :], (cy[a:b, None] - qy)[:, :, None]
        power = -0.5 * (co[a:b, 0, None, None] * dx * dx + co[a:b, 2, None, None] * dy * dy) - co[a:b, 1, None, None] * dx * dy
        ok = (power <= 0) & (co[a:b, 3, None, None] * np.exp(np.minimum(power, 0)) >= 1.0 / 255.0) & (qx[:, None, :] < W) & (qy[:, :, None] < H)
        alive[a:b] = ok.any(axis=(1, 2))
    assert not (alive & ~keep).any(), "a contributing instance was dropped"
    # same picture, same gradients
    for got, ref, name in ((color, c_color, "color"), (depth, c_depth, "depth"), (alpha, c_alpha, "alpha")):
        assert float((got - ref).abs().max()) <= 2e-6 * max(1.0, float(ref.abs().max())), name
    U.assert_close(color.cpu().numpy(), out_o["color"], "tight color vs oracle", U.REL_TOL, 2e-5)
    go = RR.backward(so, gc.numpy(), gd.numpy()[0], ga.numpy()[0])
    for got, name in ((gh[1], "means2D"), (gh[0], "means3D"), (gh[4], "opacities"), (gh[5], "scales"), (gh[6], "rotations"), (gh[2], "shs")):
        _grads_close(got, go[name], "tight lists: dL/d" + name)


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


def test_walk_histories_are_kept_per_view():
    """The forward's walk history (which tiles the next frame composites wide, the order of its work list) belongs to a VIEW: the
    binning arena keeps 128 of them keyed by the view matrix (include/riggs_hip.h: RIGGS_BIN_WALK_HIST).  Three cameras in turn:
    each gets a slot of its own once, comes back to it ever after (the cursor stops), a static camera stays in its slot, and the
    image does not depend on any of it; "fwd_hist_view_tol" = 0 is one history whatever the view."""
    import ctypes as C

    import bench
    from riggs_amd import _lib as L
    from riggs_amd import synth
    from riggs_amd.rasterizer import RasterArena
    from riggs_amd.render import render
    old = dict(bench.WORKLOAD)
    bench.WORKLOAD.update(N=20000, J=8, H=160, W=176)
    try:
        sc, cam0, gm, sw = bench.build_workload(0, "cuda:0")
    finally:
        bench.WORKLOAD.clear()
        bench.WORKLOAD.update(old)
    H, W, N = 160, 176, 20000
    cams = [synth.look_at_camera(H, W, azimuth_deg=a, fid=0.3).to("cuda") for a in (0.0, 120.0, 240.0)]
    near = synth.look_at_camera(H, W, azimuth_deg=2.0, fid=0.3).to("cuda")   # an orbit step: the first camera's history
    bg = torch.zeros(3, device="cuda")
    arena = RasterArena(min_capacity=1 << 20)
    gm._frame_arena = arena

    def frame(cam):
        with torch.no_grad():
            d = sw(gm.get_xyz, sw.expand_time(cam.fid), motion_mask=gm.motion_mask)
            img = render(cam, gm, bench.Pipe, bg, d["d_xyz"], d["d_rotation"], d["d_scaling"], arena=arena)["render"].clone()
        torch.cuda.synchronize()
        off = (C.c_size_t * L.BIN_NFIELDS)()
        L.lib().riggs_raster_binning_layout(arena.capacity, N, H, W, off)
        hdr = arena.binning[off[L.BIN_WALK_HIST]:off[L.BIN_WALK_HIST] + 8].view(torch.int32)
        return img, int(hdr[0]), int(hdr[1])
    try:
        seen = []
        for k in range(9):
            _, cursor, slot = frame(cams[k % 3])
            seen.append(slot)
        assert len(set(seen[:3])) == 3 and seen[3:6] == seen[:3] and seen[6:] == seen[:3] and cursor == 3, (seen, cursor)
        img_a, cursor, slot = frame(near)
        assert slot == seen[0] and cursor == 3                    # within the tolerance: the first view's history, no new slot
        img_b, _, _ = frame(near)
        assert float((img_a - img_b).abs().max()) <= 1e-6         # (the form a tile is composited in does not change the image)
        L.set_option("fwd_hist_view_tol", 0)
        slots = [frame(c)[2] for c in cams]
        assert slots == [0, 0, 0]
    finally:
        L.set_option("fwd_hist_view_tol", -1)
