"""-m gpu: dual-quaternion blending on the HIP kernels (csrc/dq.hip through riggs_amd.dual_quaternion, the mirror of
utils/dual_quaternion.py) against (a) golden vectors of the reference's own functions + autograd and (b) the pinned numpy
oracle (float64) at larger sizes.  Bar: values 1e-5, gradients 1e-4 of the largest entry (north_star: 1e-4 relative)."""
import glob
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import dq_ref as O  # noqa: E402
from riggs_amd import dual_quaternion as DQ  # noqa: E402

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
FILES = sorted(glob.glob(os.path.join(GOLD, "dqb_*.npz")))


def close(a, b, what, tol):
    a = a.detach().cpu().numpy() if isinstance(a, torch.Tensor) else np.asarray(a)
    a, b = a.astype(np.float64), np.asarray(b, np.float64)
    assert a.shape == b.shape, (what, a.shape, b.shape)
    scale = max(1e-30, float(np.abs(b).max()))
    err = float(np.abs(a - b).max())
    assert err <= tol * max(1.0, scale), (what, err, scale)


def dev(a, grad=False):
    return torch.from_numpy(np.ascontiguousarray(a)).float().cuda().requires_grad_(grad)


@pytest.mark.parametrize("path", FILES, ids=[os.path.basename(p)[:-4] for p in FILES])
def test_hip_matches_reference_golden(path):
    g = np.load(path)
    mode, rot_as_q = str(g["mode"]), bool(g["rot_as_q"])
    if mode == "tblend":
        T, w = dev(g["transformations"], True), dev(g["weights"], True)
        out = DQ.transformation_blending(T, w)
        close(out, g["out"], "transformation_blending", 1e-5)
        (out * dev(g["gout"])).sum().backward()
        close(w.grad, g["grad_weights"], "dL/dweights", 1e-4)
        close(T.grad, g["grad_transformations"], "dL/dtransformations", 1e-4)
        return
    if mode == "interp":
        q0, t0, q1, t1, wt = (dev(g[k], True) for k in ("q0", "t0", "q1", "t1", "weight"))
        rot, t_ = DQ.interpolate(q0, t0, q1, t1, wt, rot_as_q=rot_as_q)
        close(rot, g["out_rot"], "interpolate rot", 1e-5)
        close(t_, g["out_t"], "interpolate t", 1e-5)
        ((rot * dev(g["g_rot"])).sum() + (t_ * dev(g["g_t"])).sum()).backward()
        for k, v in (("q0", q0), ("t0", t0), ("q1", q1), ("t1", t1), ("weight", wt)):
            close(v.grad, g["grad_" + k], "dL/d" + k, 1e-4)
        return
    q, t, w = dev(g["q"], True), dev(g["t"], True), dev(g["weights"], True)
    rot, t_ = DQ.DQBlending(q, t, w, rot_as_q=rot_as_q)
    close(rot, g["out_rot"], "rotation", 1e-5)
    close(t_, g["out_t"], "translation", 1e-5)
    ((rot * dev(g["g_rot"])).sum() + (t_ * dev(g["g_t"])).sum()).backward()
    close(q.grad, g["grad_q"], "dL/dq", 1e-4)
    close(t.grad, g["grad_t"], "dL/dt", 1e-4)
    close(w.grad, g["grad_weights"], "dL/dweights", 1e-4)


@pytest.mark.parametrize("N,K,shared,nodes,rot_as_q", [
    (300_001, 23, True, False, True),      # the headline skeleton's bones, ragged last tile
    (100_000, 63, True, True, False),      # 64 joints: K odd and just under a chunk, node-axis normalisation
    (20_000, 64, True, False, True),       # exactly one chunk
    (5_000, 200, True, True, True),        # several chunks of nodes (accumulators for 4)
    (3_000, 700, True, False, False),      # ... for 16
    (77, 1, True, False, True),            # one node: the blend is that node's transform
    (200_003, 3, False, True, True),       # per-row transforms, K nearest-3 style
    (50_000, 8, False, True, False),       # the widest per-row form
    (64, 5, False, False, True),
])
def test_hip_against_the_oracle(N, K, shared, nodes, rot_as_q):
    rng = np.random.default_rng(N + K)
    shape = (K,) if shared else (N, K)
    q = rng.normal(size=shape + (4,)) * (0.5 + rng.random(shape + (1,)))
    t = 0.5 * rng.normal(size=shape + (3,))
    w = rng.random((N, K)) ** 3 + 1e-3
    w /= w.sum(-1, keepdims=True)
    g_t = rng.normal(size=(N, 3))
    q3 = q[None] if shared else q
    rot_o, t_o, cache = O.dq_blending(q3, t[None] if shared else t, w, rot_as_q, norm_over_nodes=nodes)
    g_rot = rng.normal(size=rot_o.shape)
    gq_o, gt_o, gw_o = O.dq_blending_backward(cache, g_rot, g_t)
    qd, td, wd = dev(q, True), dev(t, True), dev(w, True)
    out_mode = 1 if rot_as_q else 0
    rot, t_ = DQ._DQBlend.apply(qd, td, wd, shared, nodes, out_mode)
    close(rot.reshape(N, -1), rot_o, "rotation", 2e-5)
    close(t_, t_o, "translation", 2e-5)
    ((rot.reshape(N, -1) * dev(g_rot)).sum() + (t_ * dev(g_t)).sum()).backward()
    close(qd.grad, gq_o.reshape(qd.shape), "dL/dq", 1e-4)
    close(td.grad, gt_o.reshape(td.shape), "dL/dt", 1e-4)
    close(wd.grad, gw_o, "dL/dweights", 1e-4)
    # deterministic: no atomics anywhere
    qd2, td2, wd2 = dev(q, True), dev(t, True), dev(w, True)
    rot2, t2 = DQ._DQBlend.apply(qd2, td2, wd2, shared, nodes, out_mode)
    ((rot2.reshape(N, -1) * dev(g_rot)).sum() + (t2 * dev(g_t)).sum()).backward()
    assert torch.equal(rot, rot2) and torch.equal(qd.grad, qd2.grad) and torch.equal(wd.grad, wd2.grad)


def test_shapes_the_reference_accepts_and_errors():
    rng = np.random.default_rng(1)
    q, t, w = dev(rng.normal(size=(7, 4))), dev(rng.normal(size=(7, 3))), dev(rng.random((50, 7)))
    r2, t2 = DQ.DQBlending(q, t, w)                    # 2-D q: per-quaternion normalisation
    r3, t3 = DQ.DQBlending(q[None], t[None], w)        # 3-D q: node-axis normalisation — a different result, as in the reference
    assert r2.shape == (50, 4) and t2.shape == (50, 3) and float((r2 - r3).abs().max()) > 1e-3
    R, _ = DQ.DQBlending(q, t, w, rot_as_q=False)
    assert R.shape == (50, 3, 3)
    eye = torch.einsum("nij,nkj->nik", R, R)
    assert float((eye - torch.eye(3, device="cuda")).abs().max()) < 1e-5   # a rotation
    # one node with weight 1: the blend returns that node's (normalised) rotation and its translation
    r1, t1 = DQ.DQBlending(q[:1], t[:1], torch.ones(4, 1, device="cuda"), rot_as_q=False)
    ref = DQ.quaternion_to_matrix(q[:1])
    assert float((r1 - ref).abs().max()) < 1e-5 and float((t1 - t[:1]).abs().max()) < 1e-5
    # shapes the kernels have no form for (more than 8 transforms of its own per row, extra leading dimensions) go through the
    # composition the reference itself is — QT2DQ, weighted sum, DQ2QT — on the device: at K = 8 both routes exist and agree
    q8, t8, w8 = dev(rng.normal(size=(50, 8, 4))), dev(rng.normal(size=(50, 8, 3))), dev(rng.random((50, 8)))
    ka, kb = DQ.DQBlending(q8, t8, w8)
    ga, gb = DQ._blend_general(q8, t8, w8, 1)
    assert float((ka - ga).abs().max()) < 1e-5 and float((kb - gb).abs().max()) < 1e-5 * max(1.0, float(gb.abs().max()))
    r9, t9 = DQ.DQBlending(dev(rng.normal(size=(50, 9, 4))), dev(rng.normal(size=(50, 9, 3))), dev(rng.random((50, 9))))
    assert r9.shape == (50, 4) and t9.shape == (50, 3) and float((r9.norm(dim=-1) - 1).abs().max()) < 1e-5
    rb, tb = DQ.DQBlending(dev(rng.normal(size=(2, 5, 7, 4))), dev(rng.normal(size=(2, 5, 7, 3))), dev(rng.random((2, 5, 7))))
    assert rb.shape == (2, 5, 4) and tb.shape == (2, 5, 3) and bool(torch.isfinite(tb).all())
    with pytest.raises(L_ERR):
        DQ.DQBlending(q.cpu(), t.cpu(), w.cpu())
    # the torch-level helpers agree with the kernel's two halves
    dq = DQ.QT2DQ(q, t)
    Rh, th = DQ.DQ2QT((dq[None] * w[..., None]).sum(-2), rot_as_q=False)
    assert float((Rh - R).abs().max()) < 1e-5


from riggs_amd._lib import RiggsHipError as L_ERR  # noqa: E402


def test_dual_quaternion_skinning_with_the_skeleton_weights():
    """The composition north_star names — per-joint rigid transforms blended as dual quaternions with the skeleton's skinning
    weights: SkeletonWarp's bone weights (HIP, nn_weight) + forward kinematics (HIP) -> transformation_blending (HIP) -> x' = R x
    + t, against the same composition of oracle pieces (oracle/deform_ref.py weights and FK, oracle/dq_ref.py blend)."""
    from oracle import deform_ref as DR
    from riggs_amd import synth
    from riggs_amd.skeleton import SkeletonWarp, fk_forward
    N, J = 20_000, 24
    sc = synth.make_scene(N, J, 5)
    sw = SkeletonWarp(is_blender=True, joints=sc["joints"], parent_indices=sc["parents"], K=-1, hyper_dim=8,
                      use_skinning_weight_mlp=False, use_template_offsets=False).cuda()
    sw._node_radius.data = sc["node_radius"].cuda()
    x = sc["xyz"].cuda()
    pose = {"local_rotation": sc["local_rotation"].cuda(), "global_trans": sc["global_trans"].cuda()}
    out = sw.deform_by_pose(x, pose, None)
    w = out["nn_weight"]                                  # (N, J - 1), rows sum to 1
    par = sw._parents_dev(x.device)
    G, _, _ = fk_forward(pose["local_rotation"], sw._joints(), par, pose["global_trans"])   # (J, 12)
    xp, R, t = DQ.dqb_skinning(x, G[1:].reshape(J - 1, 3, 4), w)
    # oracle composition
    o = DR.deform_by_pose(sc["xyz"], sc["joints"], sc["parents"], sc["node_radius"], sc["local_rotation"], sc["global_trans"],
                          sc["motion_mask"], -1)
    Go = o["transforms"].numpy().astype(np.float64)
    To = O.transformation_blending(Go[1:], o["nn_weight"].numpy().astype(np.float64))
    xo = np.einsum("nij,nj->ni", To[:, :3, :3], sc["xyz"].numpy().astype(np.float64)) + To[:, :3, 3]
    close(xp, xo, "dual-quaternion skinned positions", 2e-5)
    # ... and it is a rigid motion per point where linear blend skinning is not: |det R| = 1
    assert float((torch.linalg.det(R) - 1).abs().max()) < 1e-4
