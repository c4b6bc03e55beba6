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


def hip_forward(act, cam, bg, sh_degree=3, colors=None, cov6=None, mod=1.0, debug=True):
    d = lambda t: None if t is None else t.cuda().contiguous()  # noqa: E731
    st = settings_for(cam, bg, sh_degree, mod, debug)
    return rasterize_forward(st, d(act["means3D"]), None if colors is not None else d(act["shs"]), d(colors),
                             d(act["opacities"]), None if cov6 is not None else d(act["scales"]),
                             None if cov6 is not None else d(act["rotations"]), d(cov6))


def frac_bad(a, b, rel=REL_TOL):
    """Fraction of elements whose error exceeds rel * max|b| (scale-relative, as north_star states)."""
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    scale = max(np.abs(b).max(), 1e-12)
    return float((np.abs(a - b) > rel * scale).mean()), float(np.abs(a - b).max() / scale)


def assert_close(a, b, what, rel=REL_TOL, allow_frac=0.0):
    fb, mx = frac_bad(a, b, rel)
    assert fb <= allow_frac, "%s: %.3g of elements beyond %.1e rel (max rel err %.3g)" % (what, fb, rel, mx)


def compare_forward_state(saved_oracle, v, out_oracle, color, depth, alpha, radii, px_outlier_frac=2e-5):
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
