"""The PyTorch extension front-end (riggs_amd/csrc_torch/riggs_torch.cpp: torch.ops.riggs.pose_deform / glue_raster) against the
ctypes autograd nodes it stands in for: the same C-ABI calls, so the same outputs bit for bit and the same gradients up to the
compositing backward's float atomics."""
import math
import os
import sys

import pytest
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
pytestmark = pytest.mark.gpu

from riggs_amd import _torch_ext as TX  # noqa: E402
from riggs_amd import synth  # noqa: E402
from riggs_amd.gaussian_model import GaussianModel  # noqa: E402
from riggs_amd.rasterizer import RasterArena  # noqa: E402
from riggs_amd.render import render  # noqa: E402
from riggs_amd.skeleton import SkeletonWarp  # noqa: E402


class Pipe:
    convert_SHs_python = compute_cov3D_python = debug = False


def _scene(N, J, heads, iso=False):
    sc = synth.make_scene(N, J, 31 + J, chain=False, scale=0.03)
    gm = GaussianModel.from_tensors(sc["xyz"], sc["features_dc"], sc["features_rest"], sc["scaling"], sc["rotation"], sc["opacity"])
    torch.manual_seed(5)
    sw = SkeletonWarp(joints=sc["joints"], parent_indices=sc["parents"], K=-1, hyper_dim=8, use_skinning_weight_mlp=heads,
                      use_template_offsets=heads).cuda()
    sw._node_radius.data = sc["node_radius"].cuda()
    return sc, gm, sw


def _frames(gm, sw, cam, n, mask):
    """n eager frames (the reference's two calls + backward); returns the last frame's outputs and gradients."""
    arena = RasterArena()
    bg = torch.zeros(3, device="cuda")
    params = gm.parameters() + [p for p in sw.parameters() if p.requires_grad]
    for it in range(n):
        for p in params:
            p.grad = None
        dv = sw(gm.get_xyz.detach(), sw.expand_time(torch.tensor([0.37], device="cuda")), motion_mask=mask)
        pkg = render(cam, gm, Pipe, bg, dv["d_xyz"], dv["d_rotation"], dv["d_scaling"], arena=arena)
        w = torch.linspace(0.5, 1.5, pkg["render"].numel(), device="cuda").reshape(pkg["render"].shape)
        ((pkg["render"] * w).sum() + 0.1 * pkg["depth"].sum() + 0.05 * (dv["d_nodes"] ** 2).sum()).backward()
    torch.cuda.synchronize()
    outs = [pkg["render"].detach().clone(), pkg["depth"].detach().clone(), pkg["alpha"].detach().clone(), pkg["radii"].clone(),
            dv["d_xyz"].detach().clone(), dv["d_rotation"].detach().clone(), dv["d_nodes"].detach().clone(),
            dv["local_rotation"].detach().clone()]
    grads = [(n_, None if p.grad is None else p.grad.clone()) for n_, p in
             list(zip(["xyz", "f_dc", "f_rest", "opacity", "scaling", "rotation"], gm.parameters()[:6])) + list(sw.named_parameters())]
    grads.append(("viewspace_points", pkg["viewspace_points"].grad.clone()))
    return outs, grads


@pytest.mark.parametrize("N,J,heads,masked", [(4000, 8, False, False), (20_011, 24, False, True), (6000, 12, True, True)])
def test_extension_nodes_equal_the_ctypes_nodes(N, J, heads, masked):
    assert TX.available(), "riggs_amd/lib/libriggs_torch.so is missing: __graft_entry__.build() builds it"
    sc, gm, sw = _scene(N, J, heads)
    cam = synth.look_at_camera(128, 160).to("cuda")
    mask = torch.rand(N, 1, generator=torch.Generator().manual_seed(1)).cuda() if masked else None
    res = {}
    try:
        for on in (False, True):
            TX.enable(on)
            res[on] = _frames(gm, sw, cam, 3, mask)   # (frame 1 sizes the arena from a host read in both; 2 and 3 take the extension)
    finally:
        TX.enable(True)
    for a, b in zip(res[True][0], res[False][0]):
        assert torch.equal(a, b)
    for (na, a), (nb, b) in zip(res[True][1], res[False][1]):
        assert na == nb and (a is None) == (b is None), na
        if a is not None:
            scale = float(b.abs().max())
            assert float((a - b).abs().max()) <= 2e-5 * scale + 1e-30, (na, float((a - b).abs().max()), scale)


def test_extension_is_used_by_the_eager_frame_and_not_inside_a_capture(monkeypatch):
    """The dispatch: an eager frame with an arena history calls torch.ops.riggs.*; a frame without a history, or one issued while a
    gradient bucket is registered, takes the ctypes nodes."""
    assert TX.available()
    sc, gm, sw = _scene(3000, 8, False)
    cam = synth.look_at_camera(96, 96).to("cuda")
    calls = {"pose": 0, "raster": 0}
    real_pose, real_raster = torch.ops.riggs.pose_deform, torch.ops.riggs.glue_raster

    class Spy:
        def __init__(self, f, k):
            self.f, self.k = f, k

        def __call__(self, *a):
            calls[self.k] += 1
            return self.f(*a)
    from riggs_amd import render as RM, skeleton as SM
    monkeypatch.setattr(SM.torch.ops.riggs, "pose_deform", Spy(real_pose, "pose"), raising=False)
    monkeypatch.setattr(RM.torch.ops.riggs, "glue_raster", Spy(real_raster, "raster"), raising=False)
    _frames(gm, sw, cam, 3, None)
    assert calls["pose"] == 3 and calls["raster"] == 2   # (the first render has no arena history: ctypes node + host read)
    from riggs_amd.dist import FlatGradAllReduce
    bucket = FlatGradAllReduce(gm.parameters()[:6] + [p for p in sw.parameters() if p.requires_grad])
    try:
        before = dict(calls)
        _frames(gm, sw, cam, 2, None)
        assert calls == before                           # bucket slices are the ctypes nodes' business
    finally:
        bucket.unregister()
