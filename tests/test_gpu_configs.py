"""-m gpu: BASELINE.json configs C2 .. C5 at full size.

C4 = ZJU-MoCap-like: ~500k Gaussians, 24 joints, 1024x1024, the camera built from an intrinsic matrix K with an
off-centre principal point (+13, -7) px — the reference's ``getProjectionMatrix_from_K`` path
(/root/reference/utils/graphics_utils.py:79-100, scene/cameras.py:61-72).
C5 = 2M Gaussians, 64 joints, 1920x1080.
At sizes the CPU oracle finishes in seconds the HIP path is compared with it; at full size through properties.
"""
import math
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import deform_ref as O  # noqa: E402
from oracle import raster_ref as RR  # noqa: E402
from riggs_amd import synth  # noqa: E402
from riggs_amd.gaussian_model import GaussianModel  # noqa: E402
from riggs_amd.rasterizer import rasterize_backward, rasterize_forward, saved_views  # noqa: E402
from riggs_amd.render import render  # noqa: E402
from riggs_amd.skeleton import SkeletonWarp  # noqa: E402
from tests import gpu_util as U  # noqa: E402

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def zju_K(H, W, fovx=0.6911112, dx=13.0, dy=-7.0):
    """Intrinsics with the principal point moved off the image centre."""
    fx = W / (2.0 * math.tan(fovx / 2))
    fy = H / (2.0 * math.tan(fovx / 2))
    return np.array([[fx, 0.0, W / 2.0 + dx], [0.0, fy, H / 2.0 + dy], [0.0, 0.0, 1.0]])


def test_c4_from_K_camera_parity_vs_oracle():
    """1024x1024, projection from K with a (+13, -7) px principal point, 20k Gaussians: HIP vs the CPU oracle."""
    N, J, H, W = 20_000, 24, 1024, 1024
    sc, act, cam = U.activated_scene(N, J, 1238, H, W, scale=0.02, K=zju_K(H, W))
    cam_sym = synth.look_at_camera(H, W)
    assert not torch.allclose(cam.full_proj_transform, cam_sym.full_proj_transform, atol=1e-3)  # really off-centre
    bg = [0.2, 0.1, 0.0]
    out_o, so = U.oracle_forward(act, cam, bg)
    color, radii, depth, alpha, s = U.hip_forward(act, cam, bg)
    U.compare_forward_state(so, saved_views(s), out_o, color, depth, alpha, radii)
    # ... and it moves the picture by (13, -7) px: the pixel centres differ from the symmetric projection's by that much
    so_sym = U.oracle_forward(act, cam_sym, bg)[1]
    both = (so.radii > 0) & (so_sym.radii > 0)
    shift = (so.xy[both] - so_sym.xy[both]).mean(0)
    assert abs(shift[0] - 13.0) < 0.05 and abs(shift[1] + 7.0) < 0.05, shift
    g = torch.Generator().manual_seed(5)
    gc = torch.sign(torch.rand(3, H, W, generator=g) - 0.5) / (3 * H * W)
    go = RR.backward(so, gc.numpy(), None, None)
    d = lambda t: t.cuda().contiguous()  # noqa: E731
    gh = rasterize_backward(s, d(act["means3D"]), d(act["shs"]), None, d(act["opacities"]), d(act["scales"]),
                            d(act["rotations"]), None, None, None, d(gc), None, None)
    for got, name in ((gh[0], "means3D"), (gh[1], "means2D"), (gh[2], "shs"), (gh[4], "opacities"), (gh[5], "scales"),
                      (gh[6], "rotations")):
        U.assert_close(got.cpu().numpy().reshape(go[name].shape), go[name], "C4 dL/d" + name, U.REL_TOL, 1e-4)


def test_c4_full_size_properties():
    """~500k Gaussians, 24 joints, 1024x1024, from-K camera: size-independent properties."""
    N, J, H, W = 500_000, 24, 1024, 1024
    sc, act, cam = U.activated_scene(N, J, 1238, H, W, K=zju_K(H, W))
    R, vis = U.check_full_size_properties(act, cam)
    assert R > 2_000_000 and vis > 0.9 * N


PIPE_FRAC = 5e-5     # per tensor: share of the elements allowed beyond 1e-4 of max|oracle| (observed worst: 3.3e-6, one element of C3)
DEFORM_TOL = 2e-5    # HIP d_xyz / d_rotation vs the oracle's deformation at full size, of max|oracle| (observed ~1e-6; bench.py's bar: 1e-4)
RADIUS_TOL = 1e-4    # dL/d node_radius (J sums over all Gaussians), relative to its largest entry


def _pipeline_parity_vs_oracle(N, J, H, W, cam_cpu, tag, surface=False):
    """The whole hot path (PoseMLP -> FK -> skinning -> fused glue -> rasterizer forward + backward) on the HIP side against
    the CPU oracle AT FULL SIZE.  The oracle's rasterizer is fed the HIP side's deformed means / rotations (values; the
    derivatives stay the oracle's own), so both sides take the same threshold decisions and the bars are the sharp ones:
      * radii, tiles_touched, depth bits, pixel centres, R, the sorted point list, the (tile | depth) keys and the tile ranges
        BIT-EXACT, n_contrib / image / depth / alpha / final_T within tolerance (U.compare_forward_state: the HIP rasterizer on
        the oracle's activated inputs);
      * the fused pipeline's image and every parameter gradient: per tensor <= PIPE_FRAC of the elements beyond 1e-4 of
        max|oracle|, dL/d node_radius and the PoseMLP parameter gradients within RADIUS_TOL of their largest entry.
    Returns R."""
    import bench
    from riggs_amd.rasterizer import RasterArena
    old = dict(bench.WORKLOAD)
    bench.WORKLOAD.update(N=N, J=J, H=H, W=W)
    try:
        sc, _, gm, sw = bench.build_workload(0, "cuda:0")
        if surface:  # the dense-gradient scene of bench.py: a thin, mostly opaque skin around the same skeleton
            sc = synth.make_surface_scene(N, J, bench.WORKLOAD["seed"])
            gm = GaussianModel.from_tensors(sc["xyz"], sc["features_dc"], sc["features_rest"], sc["scaling"], sc["rotation"],
                                            sc["opacity"], device="cuda:0")
    finally:
        bench.WORKLOAD.clear()
        bench.WORKLOAD.update(old)
    cam = cam_cpu.to("cuda:0")
    gimg = torch.sign(torch.rand(3, H, W, generator=torch.Generator().manual_seed(9)) - 0.5) / (3 * H * W)
    step = bench.make_step(cam, gm, sw, gimg.cuda(), RasterArena(), 1, None)
    pkg = step()
    torch.cuda.synchronize()
    names = ("xyz", "features_dc", "features_rest", "opacity", "scaling", "rotation", "node_radius")
    all_params = bench.params_of(gm, sw)
    hip_grads = {k: p.grad.detach().cpu().numpy() for k, p in zip(names, all_params)}
    hip_grads["means2D"] = pkg["viewspace_points"].grad.detach().cpu().numpy()
    pose_params = list(sw.pose_net.parameters())
    assert len(all_params) == len(names) + len(pose_params)
    hip_pose_grads = [p.grad.detach().clone() for p in pose_params]
    hip_image = pkg["render"].detach().cpu().numpy()
    del pkg
    with torch.no_grad():
        t_in = sw.expand_time(cam.fid)
        na = sw.get_pose_info(t_in)
        dv = sw(gm.get_xyz.detach(), t_in, motion_mask=gm.motion_mask)
    pose = (na["local_rotation"].detach().cpu(), na["global_trans"].detach().cpu())
    deformed = (dv["d_xyz"].detach().cpu(), dv["d_rotation"].detach().cpu())
    bench._set_threads(16)
    bench.DEFORM_PARITY.clear()
    out_o, ora_grads, so, act = bench._oracle_iteration(sc, cam_cpu, gimg, pose, deformed=deformed, want_saved=True)
    # ---- (0) the HIP deformation's forward values against the oracle's own (oracle/deform_ref.py restating
    # skeleton_warp.py:130-172), checked inside _oracle_iteration BEFORE they replace them: without this the oracle below
    # rasterizes whatever the HIP skinning produced and agrees with it (at C5 this is the two-Gaussians-per-lane instantiation)
    assert set(bench.DEFORM_PARITY) == {"d_xyz", "d_rotation"}
    for k, e in bench.DEFORM_PARITY.items():
        U.STATS.append((tag + " full size deform forward " + k, int(N), 0.0, float(e), 0.0))
        assert e <= DEFORM_TOL, (tag, k, e)
    # ---- (1) the rasterizer's forward state on identical activated inputs: bit-exact ordering and indexing at FULL size
    color, radii, depth, alpha, s = U.hip_forward(act, cam_cpu, [0.0, 0.0, 0.0])
    U.compare_forward_state(so, saved_views(s), out_o, color, depth, alpha, radii)
    del color, radii, depth, alpha, s
    torch.cuda.empty_cache()
    # ---- (2) the fused pipeline: image and gradients
    worst = 0.0
    pairs = [("image", hip_image, out_o["color"])] + [("dL/d_" + k, hip_grads[k], ora_grads[k]) for k in hip_grads]
    assert len(pairs) == 9
    for name, a, b in pairs:
        a, b = np.asarray(a, np.float64).reshape(-1), np.asarray(b, np.float64).reshape(-1)
        scale = float(np.abs(b).max())
        err = np.abs(a - b)
        assert scale > 0
        if b.size < 1000:
            # dL/d node_radius: J sums over all the Gaussians of terms that cancel (random-sign cotangent); with the threshold
            # decisions shared between the two sides what is left is the summation order
            U.STATS.append((tag + " full size " + name, int(b.size), float((err > RADIUS_TOL * scale).mean()), float(err.max() / scale), 0.0))
            assert err.max() <= RADIUS_TOL * scale, (tag, name, err.max() / scale)
            continue
        frac = float((err > 1e-4 * scale).mean())
        worst = max(worst, frac)
        assert frac <= PIPE_FRAC, (tag, name, frac, err.max() / scale)
        U.STATS.append((tag + " full size " + name, int(b.size), frac, float(err.max() / scale), float((err > 1e-4 * np.abs(b) + 1e-6 * scale).mean())))
    assert worst <= PIPE_FRAC
    # ---- (3) the PoseMLP parameter gradients: the oracle's dL/d(local_rotation, global_trans) pushed through a float64 torch
    # copy of the network on the CPU (plain torch ops: the reference's arithmetic, not the HIP kernels)
    import copy
    sw64 = copy.deepcopy(sw).cpu().double()
    for p in sw64.pose_net.parameters():
        p.grad = None
    na = sw64.get_pose_info(sw64.expand_time(cam_cpu.fid.double()))
    torch.autograd.backward([na["local_rotation"], na["global_trans"]],
                            [torch.from_numpy(ora_grads["local_rotation"]).double(), torch.from_numpy(ora_grads["global_trans"]).double()])
    ref_pose = [p.grad for p in sw64.pose_net.parameters()]
    tot = max(float(g.abs().max()) for g in ref_pose)
    for k, (g, h) in enumerate(zip(ref_pose, hip_pose_grads)):
        e = float((g - h.double().cpu()).abs().max())
        U.STATS.append((tag + " full size dL/d_pose_net[%d]" % k, int(h.numel()), 0.0, e / tot, 0.0))
        assert e <= RADIUS_TOL * tot, (tag, "pose_net", k, e / tot)
    return so.R


def test_c4_full_size_pipeline_parity_vs_oracle():
    """C4 at FULL size — 500k Gaussians, 24 joints, 1024x1024, projection from K with the off-centre principal point —
    against the CPU oracle (a few seconds on the host)."""
    H = W = 1024
    assert _pipeline_parity_vs_oracle(500_000, 24, H, W, synth.look_at_camera(H, W, K=zju_K(H, W), fid=0.37), "C4") > 2_000_000


@pytest.mark.parametrize("N,J,tag", [(150_000, 24, "C2"), (300_000, 32, "C3")])
def test_c2_c3_full_size_pipeline_parity_vs_oracle(N, J, tag):
    """C2 (D-NeRF-like: 150k Gaussians, 24 joints) and C3 (300k, 32 joints) at 800x800, full size, against the CPU oracle."""
    assert _pipeline_parity_vs_oracle(N, J, 800, 800, synth.look_at_camera(800, 800, fid=0.37), tag) > 500_000


def test_dense_gradient_scene_full_size_pipeline_parity_vs_oracle():
    """The dense-gradient scene bench.py reports (thin opaque skin: pixels saturate after a few layers, the T < 1e-4 stop and
    the 0.99 cap carry most pixels) at the headline size, against the CPU oracle."""
    assert _pipeline_parity_vs_oracle(300_000, 24, 800, 800, synth.look_at_camera(800, 800, fid=0.37), "dense scene", surface=True) > 500_000


def test_c5_full_size_pipeline_parity_vs_oracle():
    """C5 at FULL size — 2M Gaussians, 64 joints (259 PoseMLP head rows: one launch with an extra workgroup), 1920x1080, R = 38.8 M tile
    instances, the binning walking nine batches per wave — against the CPU oracle (~half a minute on the host)."""
    H, W = 1080, 1920
    assert _pipeline_parity_vs_oracle(2_000_000, 64, H, W, synth.look_at_camera(H, W, fid=0.37), "C5") > 30_000_000


def test_c4_reference_glue_fixture_through_the_hip_glue_path():
    """tests/golden/glue_iso_K.npz — captured from the reference's own render() glue + Camera (isotropic Gaussians,
    projection from K): the RAW parameters go through the fused HIP glue (cfg.glue, isotropic) and must give the
    image / radii / gradients of the oracle rasterizer fed with the reference's captured kwargs."""
    g = np.load(os.path.join(GOLD, "glue_iso_K.npz"))
    T = lambda k: torch.from_numpy(np.ascontiguousarray(g[k])).float()  # noqa: E731
    H, W, N = int(g["H"]), int(g["W"]), g["xyz"].shape[0]
    assert bool(g["isotropic"])
    bg = np.array([0.0, 0.3, 0.1], np.float32)
    out_o, so = RR.forward(g["means3D"], g["opacities"], g["viewmatrix"], g["projmatrix"], g["campos"], float(g["tanfovx"]),
                           float(g["tanfovy"]), H, W, bg, shs=g["shs"], scales=g["scales"], rotations=g["rotations"],
                           sh_degree=int(g["sh_degree"]))
    assert so.R > 0
    st = U.GaussianRasterizationSettings(
        image_height=H, image_width=W, tanfovx=float(g["tanfovx"]), tanfovy=float(g["tanfovy"]),
        bg=torch.from_numpy(bg).cuda(), scale_modifier=1.0, viewmatrix=T("viewmatrix").cuda(), projmatrix=T("projmatrix").cuda(),
        sh_degree=int(g["sh_degree"]), campos=T("campos").cuda(), prefiltered=False, debug=True)
    d = lambda k: T(k).cuda().contiguous()  # noqa: E731
    raw = dict(xyz=d("xyz"), dc=d("features_dc"), rest=d("features_rest"), op=d("opacity"), sc=d("scaling"), rot=d("rotation"),
               dx=d("d_xyz"), dr=d("d_rotation"), ds=d("d_scaling"))
    color, radii, depth, alpha, s = rasterize_forward(st, raw["xyz"], raw["dc"], None, raw["op"], raw["sc"], raw["rot"], None,
                                                      d_xyz=raw["dx"], d_rotation=raw["dr"], d_scaling=raw["ds"], glue=True,
                                                      isotropic=True, shs_rest=raw["rest"])
    assert np.array_equal(radii.cpu().numpy(), so.radii)
    U.assert_close(color.cpu().numpy(), out_o["color"], "glue_iso_K color", U.REL_TOL, 1e-4)
    U.assert_close(alpha.cpu().numpy()[0], out_o["alpha"], "glue_iso_K alpha", U.REL_TOL, 1e-4)
    # gradients w.r.t. the RAW parameters: oracle rasterizer backward chained through the oracle glue with autograd
    gen = torch.Generator().manual_seed(3)
    gc = torch.randn(3, H, W, generator=gen) / (H * W)
    go = RR.backward(so, gc.numpy(), None, None)
    P = {k: T(k).clone().requires_grad_(True) for k in ("xyz", "features_dc", "features_rest", "scaling", "rotation", "opacity")}
    m3, op, scl, rot, shs = O.render_glue(P["xyz"], P["features_dc"], P["features_rest"], P["scaling"], P["rotation"],
                                          P["opacity"], T("d_xyz"), T("d_rotation"), T("d_scaling"), True)
    np.testing.assert_allclose(m3.detach().numpy(), g["means3D"], rtol=1e-6, atol=1e-7)  # oracle glue == reference capture
    F = torch.from_numpy
    torch.autograd.backward([m3, op, scl, rot, shs], [F(go["means3D"]), F(go["opacities"]), F(go["scales"]), F(go["rotations"]),
                                                      F(go["shs"])])
    gh = rasterize_backward(s, raw["xyz"], raw["dc"], None, raw["op"], raw["sc"], raw["rot"], None, raw["dx"], raw["dr"], gc.cuda(),
                            None, None, d_scaling=raw["ds"], shs_rest=raw["rest"])
    g_xyz, _, (g_dc, g_rest), _, g_op, g_sc, g_rot, _, _ = gh
    for got, name in ((g_xyz, "xyz"), (g_dc, "features_dc"), (g_rest, "features_rest"), (g_op, "opacity"), (g_sc, "scaling")):
        U.assert_close(got.cpu().numpy().reshape(P[name].grad.shape), P[name].grad.numpy(), "glue_iso_K dL/d_" + name, 2e-4, 1e-3)
    # isotropic: Sigma = s^2 I, the rotation gradient is rounding noise in both implementations
    assert float(g_rot.abs().max()) < 1e-3 * float(g_xyz.abs().max())


def test_c5_full_size_properties_and_memory():
    """2M Gaussians, 64 joints, 1920x1080: rasterizer properties at full size, then one whole frame (PoseMLP -> FK -> LBS over
    63 bones -> fused glue + rasterizer -> backward) with its peak memory and finite, non-trivial gradients."""
    N, J, H, W = 2_000_000, 64, 1080, 1920
    fovx = 0.6911112
    fovy = 2.0 * math.atan(math.tan(fovx / 2) * H / W)  # 16:9
    sc, act, cam = U.activated_scene(N, J, 1239, H, W, fovx=fovx, fovy=fovy)
    torch.cuda.reset_peak_memory_stats()
    R, vis = U.check_full_size_properties(act, cam)
    assert R > 10_000_000 and vis > 0.9 * N
    del act
    torch.cuda.empty_cache()
    gm = GaussianModel.from_tensors(sc["xyz"], sc["features_dc"], sc["features_rest"], sc["scaling"], sc["rotation"], sc["opacity"])
    torch.manual_seed(0)
    sw = SkeletonWarp(joints=sc["joints"], parent_indices=sc["parents"], K=-1, hyper_dim=8, use_skinning_weight_mlp=False,
                      use_template_offsets=False).cuda()
    sw._node_radius.data = sc["node_radius"].cuda()

    class Pipe:
        convert_SHs_python = compute_cov3D_python = debug = False
    camg = cam.to("cuda")
    dv = sw(gm.get_xyz.detach(), sw.expand_time(camg.fid), motion_mask=gm.motion_mask)
    pkg = render(camg, gm, Pipe, torch.zeros(3, device="cuda"), dv["d_xyz"], dv["d_rotation"], dv["d_scaling"])
    g = torch.Generator().manual_seed(1)
    gimg = (torch.sign(torch.rand(3, H, W, generator=g) - 0.5) / (3 * H * W)).cuda()
    pkg["render"].backward(gimg)
    torch.cuda.synchronize()
    for p in gm.parameters() + [sw._node_radius] + list(sw.pose_net.parameters()):
        assert p.grad is not None and torch.isfinite(p.grad).all()
    assert float(gm._xyz.grad.abs().max()) > 0 and float(sw._node_radius.grad.abs().max()) > 0
    # 64 joints: 4 * 64 + 3 = 259 head rows > width 256 — the one-launch PoseMLP kernels give the three rows beyond the width a
    # workgroup of their own; LBS walked 63 bones
    assert dv["d_nodes"].shape == (64, 3)
    peak = torch.cuda.max_memory_allocated() / 2 ** 30
    assert peak < 20.0, "C5 peak memory %.1f GiB" % peak  # (7 % of the 288 GB; two forward states of ~7 GB each are alive at the peak of the property checks)
