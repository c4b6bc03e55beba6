"""Host-side mirror of RigGS's skeleton deformation interface, backed by libriggs_hip.so.

Mirrors (same names, argument meaning, return dict) of
  * ``SkeletonWarp``   /root/reference/skeleton_utils/skeleton_warp.py:10-300
  * ``PoseMLP``        /root/reference/skeleton_utils/network_utils.py:115-150
  * ``SkeletonModel``  /root/reference/scene/skeleton_model.py:9-85 (``step`` only)
The per-Gaussian work (bone distances, skinning weights, LBS of means and quaternions,
and the reductions of the backward) and the one-row PoseMLP run in HIP kernels.  The optional
per-Gaussian MLP heads of the stage-2 recipe (``WeightMLP`` / ``DeformMLP``, ``use_skinning_weight_mlp`` /
``use_template_offsets``: network_utils.py:6-112) are batched MLPs whose GEMMs go to hipBLASLt through
torch; the HIP skinning kernels consume the weight head's output (``weight_mod``) and return its gradient
(``use_fused_heads(True)`` switches them to the fused bf16-MFMA kernels of riggs_amd.mlp).
"""
from __future__ import annotations

import math

import torch
import torch.nn as nn
import torch.nn.functional as F

from . import _lib as L


# --------------------------------------------------------------------------- PoseMLP (A1)
def _embed(t: torch.Tensor, multires: int) -> torch.Tensor:
    """get_embedder(multires, 1): [t, sin(2^k t), cos(2^k t)] (utils/time_utils.py:208-256)."""
    out = [t]
    for k in range(multires):
        f = float(2.0 ** k)
        out.append(torch.sin(t * f))
        out.append(torch.cos(t * f))
    return torch.cat(out, -1)


class PoseMLP(nn.Module):
    def __init__(self, input_ch, output_ch, depth=8, hidden_dimensions=256, multires=8):
        super().__init__()
        self.skips = [depth // 2]
        self.multires = multires
        if multires > 0:
            input_ch = input_ch * (1 + 2 * multires)
        self.net = nn.ModuleList(
            [nn.Linear(input_ch, hidden_dimensions)]
            + [nn.Linear(hidden_dimensions, hidden_dimensions) if i not in self.skips
               else nn.Linear(hidden_dimensions + input_ch, hidden_dimensions) for i in range(depth - 1)])
        self.rotation_predictor = nn.Linear(hidden_dimensions, output_ch)
        self.translation_predictor = nn.Linear(hidden_dimensions, 3)
        # hand-off state of the one-launch HIP forward (include/riggs_hip.h: sync_state): zeroed once, here
        # (never inside a hipGraph capture), then owned by the kernel; follows the module across .cuda()/.to()
        # (size: riggs_pose_mlp_sync_bytes — granules, the status words, the granules of the placement check; the formula is
        # restated here because the module is also built where the library is not, and checked in forward())
        self.register_buffer("_hip_sync", torch.zeros((2 * depth * hidden_dimensions + 4 + 63) // 64 * 64 + 128, dtype=torch.int32),
                             persistent=False)

    def check_status(self):
        """Blocking read of the sticky status word of the one-launch HIP kernels (include/riggs_hip.h:
        riggs_pose_mlp_status_word).  Raises — and clears the word — when a hand-off spin timed out since the last
        check: that launch's pose was poisoned with NaN (the kernels need their <= 96 workgroups co-resident; a GPU
        shared with long-running kernels of another stream or process can break that)."""
        sync = self._hip_sync
        if not sync.is_cuda:
            return
        w = int(L.lib().riggs_pose_mlp_status_word(len(self.net), self.net[0].out_features))
        if w < sync.numel() and int(sync[w].item()) != 0:
            sync[w] = 0
            raise L.RiggsHipError("PoseMLP: a workgroup hand-off of the one-launch kernel timed out (GPU shared with "
                                  "another long-running kernel?); the pose of that step was NaN — discard the step")

    def watch(self):
        """What an EAGER caller gets instead of ``check_status`` (an unmodified train_rig.py never calls it, and a blocking read
        per iteration would stall a host-bound loop): a non-blocking look at the sticky status word every few calls
        (riggs_amd._lib.Watch).  A time-out seen this way is reported as a RuntimeWarning and the word cleared; the frame it
        belongs to had a NaN pose — ``FusedAdam`` leaves every element whose gradient is not finite untouched
        (riggs_adam_step_guarded), so the parameters survive it — and from the third one on the process switches to the
        one-launch-per-layer PoseMLP kernels (``riggs_set_option("pose_mlp_layered", 1)``: no hand-offs to lose).  Does nothing
        for a module whose word a captured graph owns (GraphedFrame / GraphedTrainStep gate on it and repair in ``check()``)."""
        sync = self._hip_sync
        if getattr(self, "_status_owned", False) or not sync.is_cuda or torch.cuda.is_current_stream_capturing():
            return
        w = getattr(self, "_status_index", None)
        if w is None:
            w = self._status_index = int(L.lib().riggs_pose_mlp_status_word(len(self.net), self.net[0].out_features))
            self._watch, self.handoff_timeouts = L.Watch(), 0
        if w >= sync.numel():
            return
        if self._watch.poll(sync, w):
            import warnings
            self._watch.value = 0
            sync[w:w + 1].zero_()
            self.handoff_timeouts += 1
            safe = self.handoff_timeouts >= 3
            if safe:
                L.set_option("pose_mlp_layered", 1)
            warnings.warn("PoseMLP: a workgroup hand-off of the one-launch kernel timed out (GPU shared with another long-running "
                          "kernel?) — %d so far; the pose of that frame was NaN and its non-finite gradients were not applied by "
                          "FusedAdam%s" % (self.handoff_timeouts, "; switching to the one-launch-per-layer PoseMLP kernels" if safe else ""),
                          RuntimeWarning, stacklevel=3)

    def _fusable(self, t):
        w = self.net[0].out_features
        return (t.is_cuda and t.dtype == torch.float32 and t.numel() == 1 and self.multires > 0 and w <= 256
                and self.net[0].in_features == 1 + 2 * self.multires and len(self.net) <= 12
                and all(l.out_features == w for l in self.net))

    def forward(self, t, rot_bias=None):
        """``rot_bias`` (4,) is added to every predicted quaternion (skeleton_warp.py:118 does it outside the
        network; folding it into the head kernel saves one elementwise launch per frame)."""
        if self._fusable(t):  # one row: three HIP launches instead of ~60 torch ops
            params = []
            for l in self.net:
                params += [l.weight, l.bias]
            params += [self.rotation_predictor.weight, self.rotation_predictor.bias,
                       self.translation_predictor.weight, self.translation_predictor.bias]
            sync = self._hip_sync
            if sync.device != t.device or sync.numel() * 4 < L.lib().riggs_pose_mlp_sync_bytes(len(self.net),
                                                                                                   self.net[0].out_features):
                sync = None  # the library then clears a private state per call
            else:
                self.watch()
            rot, tr = _PoseMLPFn.apply(t.reshape(1), len(self.net), self.net[0].out_features, self.multires,
                                       self.skips[0], rot_bias, sync, *params)
            return {"rotation": rot, "translation": tr}
        t_emb = _embed(t, self.multires) if self.multires > 0 else t
        h = t_emb + 0.0
        for i, layer in enumerate(self.net):
            h = F.relu(layer(h))
            if i in self.skips:
                h = torch.cat([t_emb, h], -1)
        rot = self.rotation_predictor(h)
        if rot_bias is not None:
            rot = (rot.reshape(-1, 4) + rot_bias).reshape(rot.shape)
        return {"rotation": rot, "translation": self.translation_predictor(h)}


class WeightMLP(nn.Module):
    """Per-Gaussian skinning-weight modulation (skeleton_utils/network_utils.py:73-112): PE(x, 10 frequencies) -> 8 x
    Linear(256) + ReLU with the embedding re-concatenated after layer 4 -> Linear(J-1) -> sigmoid.  A plain batched MLP:
    its GEMMs go to hipBLASLt through torch by default (the reference's fp32 arithmetic; the fused bf16-MFMA version is
    riggs_amd.mlp, opt-in through ``SkeletonWarp.use_fused_heads``); what is HIP
    here is the consumer — the skinning kernels take its output as ``weight_mod`` and return ``dL/dweight_mod``."""

    def __init__(self, input_ch, output_ch, D=8, W=256, multires=10):
        super().__init__()
        self.D, self.W, self.output_ch, self.multires = D, W, output_ch, multires
        self.skips = [D // 2]
        self.input_ch = 3 * (1 + 2 * multires)
        self.linear = nn.ModuleList(
            [nn.Linear(self.input_ch, W)] + [nn.Linear(W, W) if i not in self.skips else nn.Linear(W + self.input_ch, W)
                                             for i in range(D - 1)])
        self.weight_predict = nn.Linear(W, output_ch)

    def trainable_parameters(self):
        return [{"params": list(self.parameters()), "name": "weight_mlp"}]

    def forward(self, x, **kwargs):
        x_emb = _embed(x, self.multires)
        h = x_emb
        for i, layer in enumerate(self.linear):
            h = F.relu(layer(h))
            if i in self.skips:
                h = torch.cat([x_emb, h], -1)
        return torch.sigmoid(self.weight_predict(h))


class DeformMLP(nn.Module):
    """Template offsets ``detail_net`` (skeleton_utils/network_utils.py:6-70): [PE(x, 4 frequencies), pose] -> 8 x
    Linear(256) + ReLU (skip after layer 4) -> Linear(3), with the reference's initialisation.  Library GEMMs, as WeightMLP."""

    def __init__(self, D=8, W=256, xyz_input_ch=3, time_input_ch=1, output_ch=3, t_multires=-1, multires=4):
        super().__init__()
        self.D, self.W, self.output_ch, self.t_multires, self.multires = D, W, output_ch, t_multires, multires
        self.skips = [D // 2]
        if t_multires > 0:
            time_input_ch = time_input_ch * (1 + 2 * t_multires)
        if multires > 0:
            xyz_input_ch = xyz_input_ch * (1 + 2 * multires)
        self.input_ch = xyz_input_ch + time_input_ch
        self.linear = nn.ModuleList(
            [nn.Linear(self.input_ch, W)] + [nn.Linear(W, W) if i not in self.skips else nn.Linear(W + self.input_ch, W)
                                             for i in range(D - 1)])
        self.gaussian_warp = nn.Linear(W, output_ch)
        for layer in self.linear:
            nn.init.kaiming_uniform_(layer.weight, mode="fan_in", nonlinearity="relu")
            nn.init.zeros_(layer.bias)
        nn.init.normal_(self.gaussian_warp.weight, mean=0, std=1e-5)
        nn.init.zeros_(self.gaussian_warp.bias)

    def trainable_parameters(self):
        return [{"params": list(self.parameters()), "name": "offset_mlp"}]

    def forward(self, x, t, **kwargs):
        t_emb = _embed(t, self.t_multires) if self.t_multires > 0 else t
        x_emb = _embed(x, self.multires) if self.multires > 0 else x
        inp = torch.cat([x_emb, t_emb], dim=-1)
        h = inp
        for i, layer in enumerate(self.linear):
            h = F.relu(layer(h))
            if i in self.skips:
                h = torch.cat([inp, h], -1)
        return self.gaussian_warp(h)


class _PoseMLPFn(torch.autograd.Function):
    """PoseMLP forward/backward through riggs_pose_mlp_* (csrc/pose_mlp.hip)."""

    @staticmethod
    def _ptrs(params, depth):
        import ctypes as C
        Wp = (C.c_void_p * depth)(*[params[2 * l].data_ptr() for l in range(depth)])
        bp = (C.c_void_p * depth)(*[params[2 * l + 1].data_ptr() for l in range(depth)])
        return Wp, bp

    @staticmethod
    def forward(ctx, t, depth, width, multires, skip, rot_bias, sync, *params):
        ctx.set_materialize_grads(False)
        params = [p.contiguous() for p in params]
        lib = L.lib()
        dev = t.device
        n_rot = params[2 * depth].shape[0]
        acts = torch.empty(lib.riggs_pose_mlp_acts_floats(depth, width, multires), dtype=torch.float32, device=dev)
        rot = torch.empty(n_rot, dtype=torch.float32, device=dev)
        tr = torch.empty(3, dtype=torch.float32, device=dev)
        Wp, bp = _PoseMLPFn._ptrs(params, depth)
        h = params[2 * depth:]
        L.check(lib.riggs_pose_mlp_forward(depth, width, multires, skip, n_rot, Wp, bp, h[0].data_ptr(),
                                           h[1].data_ptr(), h[2].data_ptr(), h[3].data_ptr(), t.data_ptr(),
                                           L.ptr(rot_bias), L.ptr(sync), acts.data_ptr(), rot.data_ptr(), tr.data_ptr(), L.stream_ptr()),
                "riggs_pose_mlp_forward")
        ctx.save_for_backward(acts, *params)
        ctx.cfg = (depth, width, multires, skip, n_rot)
        ctx.sync = sync
        return rot, tr

    @staticmethod
    def backward(ctx, g_rot, g_tr):
        acts, *params = ctx.saved_tensors
        depth, width, multires, skip, n_rot = ctx.cfg
        lib = L.lib()
        dev = acts.device
        g_rot = torch.zeros(n_rot, device=dev) if g_rot is None else g_rot.contiguous()
        g_tr = torch.zeros(3, device=dev) if g_tr is None else g_tr.contiguous()
        from .dist import grad_out_flat
        flat = grad_out_flat(params)  # the flat gradient bucket's own range when one is registered
        dzs = torch.empty(lib.riggs_pose_mlp_backward_workspace_floats(depth, width, multires), dtype=torch.float32,
                          device=dev)
        Wp, bp = _PoseMLPFn._ptrs(params, depth)
        h = params[2 * depth:]
        L.check(lib.riggs_pose_mlp_backward(depth, width, multires, skip, n_rot, Wp, bp, h[0].data_ptr(),
                                            h[1].data_ptr(), h[2].data_ptr(), h[3].data_ptr(), acts.data_ptr(),
                                            g_rot.data_ptr(), g_tr.data_ptr(), dzs.data_ptr(), flat.data_ptr(),
                                            L.ptr(ctx.sync), L.stream_ptr()), "riggs_pose_mlp_backward")
        grads, o = [], 0
        for p in params:
            n = p.numel()
            grads.append(flat[o:o + n].view_as(p))
            o += n
        return (None, None, None, None, None, None, None, *grads)


# --------------------------------------------------------------------------- HIP ops
def fk_forward(local_rot, joints, parents_i32, global_trans):
    J = joints.shape[0]
    f32 = dict(dtype=torch.float32, device=joints.device)
    transforms = torch.empty(J, 12, **f32)
    node_rot = torch.empty(J, 4, **f32)
    d_nodes = torch.empty(J, 3, **f32)
    L.check(L.lib().riggs_fk_forward(J, local_rot.data_ptr(), joints.data_ptr(), parents_i32.data_ptr(),
                                     global_trans.data_ptr(), transforms.data_ptr(), node_rot.data_ptr(),
                                     d_nodes.data_ptr(), L.stream_ptr()), "riggs_fk_forward")
    return transforms, node_rot, d_nodes


_LBS_TABLE_MIN_N = 1_000_000  # (csrc/deform.hip: LBS_PTS2_MIN_N — below it the library keeps the bone records in LDS anyway)


def _bone_table(N: int, device):
    """Scratch for the skinning forward's bone table (``riggs_lbs_bone_table_bytes``; include/riggs_hip.h: bone_table) — handed
    over for the scenes large enough for the library to use it, or whenever ``riggs_set_option("lbs_scalar", 1)`` asks for the form.
    A fresh tensor per call: stream-ordered like every other buffer of the call, owned by the graph when captured."""
    if N < _LBS_TABLE_MIN_N and L.OPTIONS_SET.get("lbs_scalar", 0) <= 0:
        return None
    return torch.empty(int(L.lib().riggs_lbs_bone_table_bytes()), dtype=torch.uint8, device=device)


def lbs_forward(x, joints, parents_i32, rho, transforms, node_rot, global_trans, mask, K=-1, want_weights=False,
                weight_mod=None):
    N, J = x.shape[0], joints.shape[0]
    f32 = dict(dtype=torch.float32, device=x.device)
    d_xyz = torch.empty(N, 3, **f32)
    d_rot = torch.empty(N, 4, **f32)
    Kp = K if K > 0 else J - 1
    w = torch.empty(N, Kp, **f32) if want_weights else None
    idx = torch.empty(N, Kp, dtype=torch.int64, device=x.device) if want_weights else None
    L.check(L.lib().riggs_lbs_forward(N, J, K, x.data_ptr(), joints.data_ptr(), parents_i32.data_ptr(), rho.data_ptr(),
                                      transforms.data_ptr(), node_rot.data_ptr(), global_trans.data_ptr(), L.ptr(mask),
                                      L.ptr(weight_mod), d_xyz.data_ptr(), d_rot.data_ptr(), L.ptr(w), L.ptr(idx),
                                      L.ptr(_bone_table(N, x.device)), L.stream_ptr()),
            "riggs_lbs_forward")
    return d_xyz, d_rot, w, idx


class _DeformByPose(torch.autograd.Function):
    """deform_by_pose as one autograd node: FK (1 workgroup) + fused skinning/LBS."""

    @staticmethod
    def forward(ctx, local_rot, global_trans, rho, mask, x, joints, parents_i32, K, weight_mod=None):
        ctx.set_materialize_grads(False)
        if weight_mod is not None:
            weight_mod = L.require_cuda_f32("skinning weight offsets", weight_mod, (x.shape[0], joints.shape[0] - 1))
        local_rot = L.require_cuda_f32("local_rotation", local_rot, (joints.shape[0], 4))
        global_trans = L.require_cuda_f32("global_trans", global_trans.reshape(-1), (3,))
        rho = L.require_cuda_f32("_node_radius", rho, (joints.shape[0],))
        mflat = None if mask is None else L.require_cuda_f32("motion_mask", mask.reshape(-1), (x.shape[0],))
        transforms, node_rot, d_nodes = fk_forward(local_rot, joints, parents_i32, global_trans)
        d_xyz, d_rot, _, _ = lbs_forward(x, joints, parents_i32, rho, transforms, node_rot, global_trans, mflat, K,
                                         weight_mod=weight_mod)
        ctx.save_for_backward(local_rot, global_trans, rho, mflat, x, joints, parents_i32, transforms, node_rot, weight_mod)
        ctx.K = K
        ctx.mask_shape = None if mask is None else mask.shape
        ctx.mark_non_differentiable(node_rot)
        return d_xyz, d_rot, d_nodes, transforms, node_rot

    @staticmethod
    def backward(ctx, g_xyz, g_rot, g_nodes, g_transforms, _g_node_rot):
        local_rot, global_trans, rho, mflat, x, joints, parents_i32, transforms, node_rot, weight_mod = ctx.saved_tensors
        N, J = x.shape[0], joints.shape[0]
        f32 = dict(dtype=torch.float32, device=x.device)
        g_xyz = torch.zeros(N, 3, **f32) if g_xyz is None else g_xyz.contiguous()
        g_rot = torch.zeros(N, 4, **f32) if g_rot is None else g_rot.contiguous()
        dG = torch.empty(J, 12, **f32)
        from .dist import grad_out
        drho = grad_out(rho, (J,))
        dgt = torch.empty(3, **f32)
        need_mask = mflat is not None and ctx.needs_input_grad[3]
        dmask = torch.empty(N, **f32) if need_mask else None
        dmod = torch.empty(N, J - 1, **f32) if weight_mod is not None else None
        lib = L.lib()
        st = L.stream_ptr()
        ws = torch.empty(lib.riggs_lbs_backward_workspace_bytes(N, J), dtype=torch.uint8, device=x.device)
        L.check(lib.riggs_lbs_backward(N, J, ctx.K, x.data_ptr(), joints.data_ptr(), parents_i32.data_ptr(),
                                       rho.data_ptr(), transforms.data_ptr(), node_rot.data_ptr(),
                                       global_trans.data_ptr(), L.ptr(mflat), L.ptr(weight_mod), g_xyz.data_ptr(),
                                       g_rot.data_ptr(), dG.data_ptr(), drho.data_ptr(), dgt.data_ptr(), L.ptr(dmask),
                                       L.ptr(dmod), ws.data_ptr(), st),
                "riggs_lbs_backward")
        if g_transforms is not None:
            dG = dG + g_transforms
        dq = torch.empty(J, 4, **f32)
        gn = None if g_nodes is None else g_nodes.contiguous()
        L.check(lib.riggs_fk_backward(J, local_rot.data_ptr(), joints.data_ptr(), parents_i32.data_ptr(), dG.data_ptr(),
                                      L.ptr(gn), dq.data_ptr(), dgt.data_ptr(), st), "riggs_fk_backward")
        gmask = dmask.reshape(ctx.mask_shape) if need_mask else None
        return dq, dgt, drho, gmask, None, None, None, None, dmod


_ACTS_FLOATS = {}  # PoseMLP shape -> riggs_pose_mlp_acts_floats


class _PoseDeform(torch.autograd.Function):
    """SkeletonWarp.forward(x, t, mask) as ONE autograd node over three launches forward and three backward: PoseMLP, then
    forward kinematics + skinning in one launch (riggs_lbs_forward_fk); skinning backward (two launches), then the reverse
    sweep of the kinematic chain + PoseMLP backward in one launch (riggs_pose_mlp_backward_fk).  Same arithmetic as
    _PoseMLPFn + _DeformByPose (get_pose_info + deform_by_pose), two launches fewer per frame (~13 us of a 385 us frame)."""

    @staticmethod
    def forward(ctx, t, rot_bias, sync, rho, mask, x, joints, parents_i32, K, weight_mod, fixed, depth, width, multires, skip, *params):
        ctx.set_materialize_grads(False)
        ctx.fixed = fixed  # None, or (coef, loss_out) device scalars: the template frame's pose regulariser (riggs_pose_mlp_backward_fk)
        params = [p.contiguous() for p in params]
        lib, dev = L.lib(), x.device
        N, J = x.shape[0], joints.shape[0]
        f32 = dict(dtype=torch.float32, device=dev)
        n_rot = params[2 * depth].shape[0]
        if weight_mod is not None:
            weight_mod = L.require_cuda_f32("skinning weight offsets", weight_mod, (N, J - 1))
        rho = L.require_cuda_f32("_node_radius", rho, (J,))
        mflat = None if mask is None else L.require_cuda_f32("motion_mask", mask.reshape(-1), (N,))
        # (two allocations for the eight outputs / saved arrays — every piece on a 256-byte boundary: eight torch.empty calls
        # were 15 us of an eagerly issued frame)
        key = (depth, width, multires)
        n_acts = _ACTS_FLOATS.get(key)
        if n_acts is None:
            n_acts = _ACTS_FLOATS[key] = lib.riggs_pose_mlp_acts_floats(depth, width, multires)
        o_small = (n_acts + 63) & ~63
        sbuf = torch.empty(o_small + J * 23 + 4, **f32)
        acts, small = sbuf[:n_acts], sbuf[o_small:]
        local_rot, transforms = small[:J * 4].view(J, 4), small[J * 4:J * 16].view(J, 12)
        node_rot, d_nodes, global_trans = small[J * 16:J * 20].view(J, 4), small[J * 20:J * 23].view(J, 3), small[J * 23:J * 23 + 3]
        o_rot = (3 * N + 63) & ~63
        dbuf = torch.empty(o_rot + 4 * N, **f32)
        d_xyz, d_rot = dbuf[:3 * N].view(N, 3), dbuf[o_rot:].view(N, 4)
        Wp, bp = _PoseMLPFn._ptrs(params, depth)
        h = params[2 * depth:]
        st = L.stream_ptr()
        L.check(lib.riggs_pose_mlp_forward(depth, width, multires, skip, n_rot, Wp, bp, h[0].data_ptr(), h[1].data_ptr(),
                                           h[2].data_ptr(), h[3].data_ptr(), t.data_ptr(), L.ptr(rot_bias), L.ptr(sync),
                                           acts.data_ptr(), local_rot.data_ptr(), global_trans.data_ptr(), st),
                "riggs_pose_mlp_forward")
        L.check(lib.riggs_lbs_forward_fk(N, J, K, x.data_ptr(), joints.data_ptr(), parents_i32.data_ptr(), rho.data_ptr(),
                                         local_rot.data_ptr(), global_trans.data_ptr(), L.ptr(mflat), L.ptr(weight_mod),
                                         transforms.data_ptr(), node_rot.data_ptr(), d_nodes.data_ptr(), d_xyz.data_ptr(),
                                         d_rot.data_ptr(), L.ptr(_bone_table(N, x.device)), st), "riggs_lbs_forward_fk")
        ctx.save_for_backward(acts, local_rot, global_trans, rho, mflat, x, joints, parents_i32, transforms, node_rot, weight_mod,
                              *params)
        ctx.cfg = (depth, width, multires, skip, n_rot, K)
        ctx.sync = sync
        ctx.mask_shape = None if mask is None else mask.shape
        ctx.mark_non_differentiable(node_rot)
        return d_xyz, d_rot, d_nodes, local_rot, global_trans, transforms, node_rot

    @staticmethod
    def backward(ctx, g_xyz, g_rot, g_nodes, g_local_rot, g_global_trans, g_transforms, _g_node_rot):
        acts, local_rot, global_trans, rho, mflat, x, joints, parents_i32, transforms, node_rot, weight_mod, *params = ctx.saved_tensors
        depth, width, multires, skip, n_rot, K = ctx.cfg
        N, J = x.shape[0], joints.shape[0]
        lib, dev = L.lib(), x.device
        f32 = dict(dtype=torch.float32, device=dev)
        g_xyz = torch.zeros(N, 3, **f32) if g_xyz is None else g_xyz.contiguous()
        g_rot = torch.zeros(N, 4, **f32) if g_rot is None else g_rot.contiguous()
        bsmall = torch.empty(J * 16 + 12, **f32)  # (dG | dq | dgt | dgt_total: one allocation)
        dG, dq = bsmall[:J * 12].view(J, 12), bsmall[J * 12:J * 16].view(J, 4)
        dgt, dgt_total = bsmall[J * 16:J * 16 + 3], bsmall[J * 16 + 4:J * 16 + 7]
        from .dist import grad_out, grad_out_flat
        drho = grad_out(rho, (J,))
        need_mask = mflat is not None and ctx.needs_input_grad[4]
        dmask = torch.empty(N, **f32) if need_mask else None
        dmod = torch.empty(N, J - 1, **f32) if weight_mod is not None else None
        st = L.stream_ptr()
        ws = torch.empty(lib.riggs_lbs_backward_workspace_bytes(N, J), dtype=torch.uint8, device=dev)
        L.check(lib.riggs_lbs_backward(N, J, K, x.data_ptr(), joints.data_ptr(), parents_i32.data_ptr(), rho.data_ptr(),
                                       transforms.data_ptr(), node_rot.data_ptr(), global_trans.data_ptr(), L.ptr(mflat),
                                       L.ptr(weight_mod), g_xyz.data_ptr(), g_rot.data_ptr(), dG.data_ptr(), drho.data_ptr(),
                                       dgt.data_ptr(), L.ptr(dmask), L.ptr(dmod), ws.data_ptr(), st), "riggs_lbs_backward")
        if g_transforms is not None:
            dG = dG + g_transforms
        if g_global_trans is not None:
            dgt = dgt + g_global_trans.reshape(-1)
        gn = None if g_nodes is None else g_nodes.contiguous()
        gq = None if g_local_rot is None else g_local_rot.contiguous()
        flat = grad_out_flat(params)  # the flat gradient bucket's own range when one is registered
        dzs = torch.empty(lib.riggs_pose_mlp_backward_workspace_floats(depth, width, multires), **f32)
        Wp, bp = _PoseMLPFn._ptrs(params, depth)
        h = params[2 * depth:]
        L.check(lib.riggs_pose_mlp_backward_fk(depth, width, multires, skip, n_rot, Wp, bp, h[0].data_ptr(), h[1].data_ptr(),
                                               h[2].data_ptr(), h[3].data_ptr(), acts.data_ptr(), J, local_rot.data_ptr(),
                                               joints.data_ptr(), parents_i32.data_ptr(), transforms.data_ptr(), dG.data_ptr(), L.ptr(gn), L.ptr(gq),
                                               dgt.data_ptr(), dq.data_ptr(), dgt_total.data_ptr(),
                                               L.ptr(ctx.fixed[0]) if ctx.fixed else None, L.ptr(ctx.fixed[1]) if ctx.fixed else None,
                                               dzs.data_ptr(), flat.data_ptr(), L.ptr(ctx.sync), st), "riggs_pose_mlp_backward_fk")
        # (one split + a view per matrix: a slice and a view per parameter were 30 us of an eagerly issued frame)
        grads = [g_ if p.dim() == 1 else g_.view(p.shape) for g_, p in zip(flat.split_with_sizes([p.numel() for p in params]), params)]
        gmask = dmask.reshape(ctx.mask_shape) if need_mask else None
        return (None, None, None, drho, gmask, None, None, None, None, dmod, None, None, None, None, None, *grads)


class _LazyDeformDict(dict):
    """deform_by_pose's return dict; ``nn_idx`` / ``nn_weight`` (needed by render_rig.py:156-158,
    not by training) are materialised by the HIP kernel on first access."""

    _LAZY = ("nn_idx", "nn_weight")

    def __init__(self, *a, producer=None, **k):
        super().__init__(*a, **k)
        self._producer = producer

    def _fill(self):
        if self._producer is not None:
            w, idx = self._producer()
            self._producer = None
            dict.__setitem__(self, "nn_weight", w)
            dict.__setitem__(self, "nn_idx", idx)

    def __getitem__(self, key):
        if key in self._LAZY and self._producer is not None:
            self._fill()
        return dict.__getitem__(self, key)

    def get(self, key, default=None):
        if key in self._LAZY and self._producer is not None:
            self._fill()
        return dict.get(self, key, default)


class _BaseNetworkPlaceholder(nn.Module):
    """Stand-in for the stage-1 network a ``ControlNodeWarp`` owns (utils/time_utils.py:797-804).  ``SkeletonWarp.forward``
    never evaluates it; it exists in checkpoints (``network.*``) and is ticked by ``update`` (a frequency-mask schedule of
    the stage-1 embedding: utils/time_utils.py:455-458) — nothing on this path depends on either."""

    def __init__(self):
        super().__init__()
        self.name = "static"
        self.param = nn.Parameter(torch.zeros(1), requires_grad=False)
        self.reg_loss = 0.0

    def update(self, iteration, *args, **kwargs):
        return


class SkeletonWarp(nn.Module):
    """HIP-backed mirror of skeleton_utils/skeleton_warp.py:SkeletonWarp (:10-300) and of what it inherits from
    ``ControlNodeWarp`` (utils/time_utils.py:770-932, :1238-1260): constructor signature, attributes, state-dict keys,
    ``as_gaussians`` / ``init_gaussians`` / ``update`` — the surface ``scene/skeleton_model.py``, ``train_rig.py``,
    ``render_rig.py`` and ``interactive_GUI.py`` touch (recorded from the reference: tests/golden/skeleton_api_calls.json)."""

    def __init__(self, is_blender=True, joints=None, parent_indices=None, init_pcl=None, K=3, use_hash=False, hash_time=False,
                 enable_densify_prune=False, pred_opacity=False, pred_color=False, with_arap_loss=False, with_node_weight=False,
                 local_frame=False, d_rot_as_res=True, skinning=False, hyper_dim=2, progressive_brand_time=False, max_d_scale=-1,
                 is_scene_static=False, use_skinning_weight_mlp=True, use_template_offsets=True, **kwargs):
        super().__init__()
        if joints is None or parent_indices is None:
            raise ValueError("joints and parent_indices are required")
        if skinning:
            raise ValueError("skinning=True leaves the reference's SkeletonWarp without _node_radius (utils/time_utils.py:809) "
                             "and its deform_by_pose fails on node_radius: not a configuration of this path")
        J = joints.shape[0]
        if J > 64:
            raise ValueError("at most 64 joints are supported by the LDS-staged kernels")
        self.K = K
        self.name = "node"
        # attributes of the base class (utils/time_utils.py:773-819), kept because callers and checkpoints read them
        self.use_hash, self.hash_time, self.enable_dp = use_hash, hash_time, enable_densify_prune
        self.with_node_weight, self.local_frame, self.skinning = with_node_weight, local_frame, False
        self.pred_opacity, self.pred_color, self.max_d_scale = pred_opacity, pred_color, max_d_scale
        self.is_scene_static = is_scene_static
        self.is_blender = is_blender
        self.d_rot_as_res = d_rot_as_res
        self.hyper_dim = hyper_dim
        self.reg_loss = 0.0
        if with_arap_loss and not is_scene_static:  # :789-794 (read by the stage-1 trainer only)
            self.lambda_arap_landmarks, self.lambda_arap_steps = [1e-4, 1e-4, 1e-5, 1e-5, 0], [0, 5000, 10000, 20000, 20001]
        else:
            self.lambda_arap_landmarks, self.lambda_arap_steps = [0], [0]
        nodes = torch.randn(J, 3 + hyper_dim)
        nodes[:, :3] = joints.detach().float().cpu()
        self.nodes = nn.Parameter(nodes, requires_grad=False)  # skeleton_warp.py:14-16
        self._node_radius = nn.Parameter(torch.randn(J))        # utils/time_utils.py:807
        if with_node_weight:
            self._node_weight = nn.Parameter(torch.zeros(J, 1))  # :811 (state only: the skeleton's weights ignore it)
        self.register_buffer("parents", parent_indices.detach().long().cpu().clone(), persistent=False)  # an attribute, not state, in the reference
        self.use_skinning_weight_mlp = use_skinning_weight_mlp
        self.use_template_offsets = use_template_offsets
        self.skinning_weight_offsets = None
        self.control_nodes = nn.Parameter(torch.zeros(512, 3))  # checkpoint compatibility (:31)
        self.template_offsets = None
        # the stage-2 objective's two regularisers as cotangents inside this module's own backward launches (set per call by
        # riggs_amd.graph.GraphedTrainStep; None: a caller that writes the terms in torch — the reference's train_rig.py:446-482
        # — gets them through autograd): (coef, mean_sq) device scalars for the fused DeformMLP (riggs_mlp_l2_grad_scale),
        # (coef, loss) for the PoseMLP's backward (riggs_pose_mlp_backward_fk)
        self.template_l2 = None
        self.template_fixed = None
        self.pose_net = PoseMLP(1, J * 4)
        # (the heads are constructed AFTER pose_net: they draw from the global RNG, and the pose network of a given seed
        # — the benchmark's scene — must not depend on whether they exist)
        if use_skinning_weight_mlp:
            self.skinning_weight_mlp = WeightMLP(input_ch=3, output_ch=J - 1)  # skeleton_warp.py:24-28
        self.detail_net = DeformMLP(xyz_input_ch=3, time_input_ch=J * 4, t_multires=-1)  # :32 (always constructed)
        # checkpoint compatibility (skeleton.pth, scene/skeleton_model.py:43-72): the reference's state dict also holds the
        # `inited` flag and the parameter of its (static) base network — utils/time_utils.py:288-300, :799-805
        self.register_buffer("inited", torch.tensor(True))
        self.network = _BaseNetworkPlaceholder()
        self.register_buffer("_rot_bias", torch.tensor([1.0, 0.0, 0.0, 0.0]), persistent=False)  # skeleton_warp.py:118
        self.nodes_color_visualization = torch.ones(J, 3 + hyper_dim)  # :816
        self.cached_nn_weight = False  # :819-820 (a GUI toggle; the skeleton's weights are never cached)
        self.nn_weight = self.nn_dist = self.nn_idxs = None
        self.gs = None
        self._parents_i32 = None
        self._joints_key, self._joints_cache = None, None

    # -- reference surface ------------------------------------------------------------------
    @property
    def node_radius(self):
        return torch.exp(self._node_radius)

    @property
    def node_num(self):
        return self.nodes.shape[0]

    def expand_time(self, t):
        return t.unsqueeze(0).expand(self.nodes.shape[0], -1)  # utils/time_utils.py:929-932

    def update_control_nodes(self, nodes):
        self.control_nodes.data = nodes

    @property
    def node_weight(self):  # utils/time_utils.py:877-879
        return torch.sigmoid(self._node_weight)

    def update(self, iteration):
        """Per-iteration hook of the base network (utils/time_utils.py:821-822 <- scene/skeleton_model.py:84-85 <-
        train_rig.py:533): nothing the skeleton path evaluates has a schedule."""
        self.network.update(iteration)

    @property
    def param_names(self):  # utils/time_utils.py:834-842 (skinning is never on here)
        return ["nodes", "_node_radius", "_node_weight"] if self.with_node_weight else ["nodes", "_node_radius"]

    # ---- the joints as a small Gaussian model (utils/time_utils.py:1238-1260): what SkeletonModel.train_setting sets an
    # optimizer up for (scene/skeleton_model.py:38-39) and the GUI's skeleton-only view renders (interactive_GUI.py:265,391)
    @staticmethod
    def _gaussian_classes():
        try:  # the trainer's own classes when this module runs inside a RigGS checkout
            from scene.gaussian_model import BasicPointCloud, StandardGaussianModel
        except Exception:
            from .gaussian_model import BasicPointCloud, StandardGaussianModel
        return BasicPointCloud, StandardGaussianModel

    @property
    def as_gaussians(self):
        if getattr(self, "gs", None) is None:
            print("Building Learnable Gaussians for Nodes!")
            BasicPointCloud, StandardGaussianModel = self._gaussian_classes()
            joints = self.nodes[..., :3].detach()
            pcd = BasicPointCloud(points=joints, colors=torch.zeros_like(joints), normals=joints)
            self.gs = StandardGaussianModel(sh_degree=0, all_the_same=True, with_motion_mask=False)
            self.gs.create_from_pcd(pcd=pcd, spatial_lr_scale=0.0, print_info=False)  # distCUDA2 on J points
            self.gs._scaling.data = torch.log(1e-2 * torch.ones_like(self.gs._scaling))
            self.gs._xyz.data = self.nodes[..., :3]
        return self.gs

    def init_gaussians(self, init_pcl, with_motion_mask):
        if getattr(self, "gs", None) is None:
            print("Initialize Learnable Gaussians for Nodes with Point Clouds!")
            BasicPointCloud, StandardGaussianModel = self._gaussian_classes()
            pcd = BasicPointCloud(points=init_pcl.detach(), colors=torch.zeros_like(init_pcl), normals=torch.zeros_like(init_pcl))
            self.gs = StandardGaussianModel(sh_degree=0, all_the_same=True, with_motion_mask=with_motion_mask)
            self.gs.create_from_pcd(pcd=pcd, spatial_lr_scale=0.0, print_info=False)
        return self.gs

    def state_dict(self, *args, **kwargs):
        """The module's entries plus, once the joint Gaussians exist, theirs as ``gs_<name>`` (utils/time_utils.py:867-872)."""
        sd = super().state_dict(*args, **kwargs)
        if getattr(self, "gs", None) is not None:
            prefix = kwargs.get("prefix", args[1] if len(args) > 1 else "")
            for name in self.gs.param_names():
                sd[prefix + "gs_" + name] = getattr(self.gs, name)
        return sd

    def load_state_dict(self, state_dict, strict=True, **kw):
        """Accepts the reference's ``skeleton.pth`` (utils/time_utils.py:844-865): node parameters are assigned (re-created
        when the joint count differs), ``gs_*`` entries go to the joint Gaussians, and entries of the base deformation
        network other than the static placeholder (a stage-1 leftover this path never evaluates) are dropped."""
        state_dict = dict(state_dict)
        for key in self.param_names:
            if key in state_dict:
                v = state_dict.pop(key)
                cur = getattr(self, key)
                if cur.shape != v.shape:
                    print(f"Loading nodes mismatching the original setting: {cur.shape} and {v.shape}")
                    setattr(self, key, nn.Parameter(v.detach().clone().to(cur.device), requires_grad=cur.requires_grad))
                else:
                    cur.data = v.detach().to(cur.device, cur.dtype).clone()
        for key in [k for k in state_dict if k.startswith("gs_")]:
            v, name = state_dict.pop(key), key[3:]
            try:
                getattr(self.as_gaussians, name).data = v
            except Exception:
                print(f"Directly set as values for {key} when loading deform gaussians")
                setattr(self.as_gaussians, name, v)
        mine = super().state_dict()
        sd = {k: v for k, v in state_dict.items() if k in mine or not k.startswith("network.")}
        sd.setdefault("network.param", mine["network.param"])  # absent when the file's base network was not the static one
        for key in self.param_names:
            sd[key] = getattr(self, key).data
        return super().load_state_dict(sd, strict=strict, **kw)

    def trainable_parameters(self):
        params = [{"params": [self._node_radius], "name": "nodes"},
                  {"params": list(self.pose_net.parameters()), "name": "pose"}]
        if self.use_skinning_weight_mlp:
            params.append({"params": list(self.skinning_weight_mlp.parameters()), "name": "skinning_mlp"})
        if self.use_template_offsets:
            params.append({"params": list(self.detail_net.parameters()), "name": "detail_net"})
        return params  # skeleton_warp.py:276-288

    def _parents_dev(self, device):
        if self._parents_i32 is None or self._parents_i32.device != device:
            p = self.parents.detach().cpu().to(torch.int32).clone()
            p[0] = 0
            if bool((p[1:] >= torch.arange(1, p.shape[0], dtype=torch.int32)).any()):
                raise ValueError("parents[i] < i is required (skeleton_warp.py:257-263)")
            self._parents_i32 = p.to(device)
        return self._parents_i32

    def _joints(self):
        """Rest joint positions as a contiguous (J, 3) tensor; re-sliced only when ``nodes`` was modified
        (``nodes`` is a frozen parameter in rig training: skeleton_warp.py:14-16), so no copy per frame."""
        key = (self.nodes.data_ptr(), self.nodes._version)
        if self._joints_key != key:
            self._joints_cache = self.nodes[:, :3].detach().contiguous()
            self._joints_key = key
        return self._joints_cache

    def get_pose_info(self, t):
        if t.dim() == 0:
            t = self.expand_time(t)
        m = self.pose_net(t[0], rot_bias=self._rot_bias)  # identity-quaternion bias of skeleton_warp.py:118
        return {"local_rotation": m["rotation"].reshape(-1, 4), "global_trans": m["translation"], "t": t[0]}

    def forward(self, x, t, motion_mask, **kwargs):
        if t.dim() == 0:
            t = self.expand_time(t)
        if x.is_cuda and self.pose_net._fusable(t[0]) and self.pose_net.rotation_predictor.out_features == 4 * self.nodes.shape[0]:
            return self.deform_by_pose(x, None, motion_mask, _time=t[0])  # PoseMLP + FK + skinning as one autograd node
        return self.deform_by_pose(x, self.get_pose_info(t), motion_mask)

    # ---- the per-Gaussian MLP heads: fp32 GEMMs through torch (the reference's arithmetic, default), or — opt-in —
    # the fused MFMA kernels of riggs_amd.mlp (~7x faster forward + backward at 300k Gaussians;
    # ``fmt``: "fp16" (default; gradients within ~2 % of the fp32 path) or "bf16")
    def use_fused_heads(self, on: bool = True, fmt: str = None, sparse_weight_rows: bool = True):
        """``sparse_weight_rows`` (default on): the WeightMLP's backward runs on the rows whose cotangent is non-zero — the
        Gaussians the render gave a gradient; the skinning backward writes exact zeros for the rest and the reference has no other
        term on this head (its regulariser is commented out, train_rig.py:433-444) — and its forward stores no activations
        (riggs_amd.mlp.FusedHead).  Same parameter gradients as the dense pass; turn it off for a caller that puts a dense loss on
        ``skinning_weight_offsets``."""
        from .mlp import DEFAULT_FORMAT, _fmt_dtype
        self._fused_heads = bool(on)
        self._fused_sparse_w = bool(sparse_weight_rows)
        self._fused_fmt = fmt or DEFAULT_FORMAT
        _fmt_dtype(self._fused_fmt)
        self._fh_w = self._fh_d = None
        return self

    def _fused_embedding(self, x, multires: int):
        """The positional embedding of ``x`` as the fused heads' 16-bit operand, computed ONCE per call of the warp for both heads:
        get_embedder's columns are [x, sin(2^k x), cos(2^k x) for k < multires] (utils/time_utils.py:208-256), so the DeformMLP's
        (multires 4: 27 columns) are the first columns of the WeightMLP's (multires 10: 63) — same values, same rounding — and both
        operands are 64 wide; the DeformMLP's packed weights are zero past its own columns (riggs_mlp_pack) and the partial-sum
        launch writes only those of its gradient."""
        from .mlp import embed_positions_bf16
        widest = max(int(self.skinning_weight_mlp.multires), int(self.detail_net.multires)) if (
            self.use_skinning_weight_mlp and self.use_template_offsets) else multires
        if 3 * (1 + 2 * widest) > 64 or widest < multires:  # (one 64-column operand must hold both)
            widest = multires
        # (the cache lives for ONE call of deform_by_pose, which clears it on entry: nothing can change x in between)
        key = (x.data_ptr(), x.shape[0], self._fused_fmt, widest)
        hit = getattr(self, "_emb_cache", None)
        if hit is None or hit[0] != key:
            hit = self._emb_cache = (key, embed_positions_bf16(x, widest, fmt=self._fused_fmt))
        return hit[1]

    def _head_weight(self, x):
        if not getattr(self, "_fused_heads", False):
            return self.skinning_weight_mlp(x)
        from .mlp import FusedHead
        net = self.skinning_weight_mlp
        if getattr(self, "_fh_w", None) is None:
            # (the sigmoid of network_utils.py:107 inside the forward launch, its derivative inside the backward's first launch)
            self._fh_w = FusedHead(net.linear, net.weight_predict, net.input_ch, net.skips[0], self._fused_fmt,
                                   sparse_rows=getattr(self, "_fused_sparse_w", True), out_sigmoid=True)
        return self._fh_w(self._fused_embedding(x, net.multires), n_rows=x.shape[0])

    def _head_detail(self, x, pose, res=None):
        """The template offsets; with ``res = (d_xyz, mask)`` on the fused path ``(offsets, d_xyz + offsets * mask)`` — joined by
        the forward launch (riggs_mlp_epilogue) instead of two elementwise launches (and one in the backward)."""
        if not getattr(self, "_fused_heads", False):
            return self.detail_net(x, pose)
        from .mlp import FusedHead
        net = self.detail_net
        if net.t_multires <= 0 and net.multires > 0:  # the reference's configuration: PE(x) and the raw pose vector
            # (the pose is ONE vector for all Gaussians — skeleton_warp.py:152 expands it — so it enters through the biases of the two
            # layers that read the input, in fp32, and the kernels' operand is the positional embedding alone: riggs_amd.mlp.Packed)
            n_pose = int(pose.shape[-1])
            if getattr(self, "_fh_d", None) is None:
                self._fh_d = FusedHead(net.linear, net.gaussian_warp, net.input_ch - n_pose, net.skips[0], self._fused_fmt, tail_ch=n_pose)
            return self._fh_d(self._fused_embedding(x, net.multires), n_rows=x.shape[0],
                              l2=getattr(self, "template_l2", None), res=res, tail=pose[0])
        if getattr(self, "_fh_d", None) is None:
            self._fh_d = FusedHead(net.linear, net.gaussian_warp, net.input_ch, net.skips[0], self._fused_fmt)
        t_emb = _embed(pose, net.t_multires) if net.t_multires > 0 else pose
        x_emb = _embed(x, net.multires) if net.multires > 0 else x
        return self._fh_d(torch.cat([x_emb, t_emb], dim=-1), l2=getattr(self, "template_l2", None))

    def deform_by_pose(self, x, node_attrs, motion_mask, _time=None):
        self._emb_cache = None  # (_fused_embedding: shared by the two heads of THIS call)
        x = L.require_cuda_f32("x", x.detach(), (x.shape[0], 3))
        if _time is None:
            local_rot, global_trans = node_attrs["local_rotation"], node_attrs["global_trans"]
        joints = self._joints()
        par = self._parents_dev(x.device)
        mask = motion_mask
        if mask is not None and not isinstance(mask, torch.Tensor):
            mask = None if float(mask) == 1.0 else torch.full((x.shape[0], 1), float(mask), device=x.device)
        weight_mod = None
        if self.use_skinning_weight_mlp:  # skeleton_warp.py:56-61
            if self.K > 0:
                raise NotImplementedError("use_skinning_weight_mlp with K > 0: the reference gathers the MLP output with the "
                                          "1-based bone indices (skeleton_warp.py:59), which runs off its (N, J-1) columns; "
                                          "only K = -1 is well defined")
            weight_mod = self._head_weight(x)
            self.skinning_weight_offsets = weight_mod
        if _time is not None:
            pn = self.pose_net
            params = []
            for l in pn.net:
                params += [l.weight, l.bias]
            params += [pn.rotation_predictor.weight, pn.rotation_predictor.bias, pn.translation_predictor.weight,
                       pn.translation_predictor.bias]
            sync = pn._hip_sync
            if sync.device != x.device or sync.numel() * 4 < L.lib().riggs_pose_mlp_sync_bytes(len(pn.net), pn.net[0].out_features):
                sync = None
            else:
                pn.watch()
            self._fixed_folded = getattr(self, "template_fixed", None) is not None
            from . import _torch_ext as TX
            from .dist import _SLICES, _entry
            if (TX.active() and x.is_cuda and _time.is_cuda
                    and (not _SLICES or (_entry(self._node_radius) is None and all(_entry(p_) is None for p_ in params)))):
                # the same node in C++ (csrc_torch/riggs_torch.cpp: torch.ops.riggs.pose_deform) — an eagerly issued frame is
                # host-bound; with a registered gradient bucket or inside a capture the ctypes node below runs
                fx = getattr(self, "template_fixed", None)
                d_xyz, d_rot, d_nodes, local_rot, global_trans, transforms, node_rot = torch.ops.riggs.pose_deform(
                    _time.reshape(1), self._rot_bias, sync, self._node_radius, mask, x, joints, par, weight_mod,
                    fx[0] if fx else None, fx[1] if fx else None, _bone_table(x.shape[0], x.device), self.K, len(pn.net),
                    pn.net[0].out_features, pn.multires, pn.skips[0], params)
            else:
                d_xyz, d_rot, d_nodes, local_rot, global_trans, transforms, node_rot = _PoseDeform.apply(
                    _time.reshape(1), self._rot_bias, sync, self._node_radius, mask, x, joints, par, self.K, weight_mod,
                    getattr(self, "template_fixed", None), len(pn.net), pn.net[0].out_features, pn.multires, pn.skips[0], *params)
            node_attrs = {"local_rotation": local_rot, "global_trans": global_trans, "t": _time}
        else:
            d_xyz, d_rot, d_nodes, transforms, node_rot = _DeformByPose.apply(
                local_rot, global_trans, self._node_radius, mask, x, joints, par, self.K, weight_mod)
        if self.use_template_offsets:  # skeleton_warp.py:152-158: offsets join the blended position before the mask
            pose = local_rot.detach().reshape(-1)[None].expand(x.shape[0], -1)
            net = self.detail_net
            fold = (getattr(self, "_fused_heads", False) and net.t_multires <= 0 and net.multires > 0
                    and (mask is None or not mask.requires_grad))
            if fold:
                self.template_offsets, d_xyz = self._head_detail(x, pose, res=(d_xyz, mask))
            else:
                self.template_offsets = self._head_detail(x, pose)
                d_xyz = d_xyz + (self.template_offsets if mask is None else self.template_offsets * mask)
        else:
            self.template_offsets = None
        wm = None if weight_mod is None else weight_mod.detach()
        rho = self._node_radius.detach()
        gt = global_trans.detach().reshape(-1).contiguous()
        mflat = None if mask is None else mask.detach().reshape(-1).contiguous()

        def producer():
            _, _, w, idx = lbs_forward(x, joints, par, rho.contiguous(), transforms.detach(), node_rot, gt, mflat,
                                       self.K, want_weights=True, weight_mod=wm)
            return w, idx
        zs = getattr(self, "_zero_scaling", None)
        if zs is None or zs.shape[0] != x.shape[0] or zs.device != x.device:
            zs = self._zero_scaling = torch.zeros(x.shape[0], 3, device=x.device)  # constant (skeleton_warp.py:165)
        return _LazyDeformDict(
            {"d_xyz": d_xyz, "d_rotation": d_rot, "d_scaling": zs,
             "d_nodes": d_nodes, "nn_idx": None, "nn_weight": None, "local_rotation": node_attrs["local_rotation"],
             "global_trans": global_trans, "d_opacity": None, "d_color": None}, producer=producer)

    def node_deformation(self, x, node_attrs):
        x = x.detach()
        joints = self._joints()
        par = self._parents_dev(joints.device)
        local_rot = L.require_cuda_f32("local_rotation", node_attrs["local_rotation"], (joints.shape[0], 4))
        gt = L.require_cuda_f32("global_trans", node_attrs["global_trans"].reshape(-1), (3,))
        _, _, d_nodes = fk_forward(local_rot.detach(), joints, par, gt.detach())
        return {"d_xyz": d_nodes - x, "d_opacity": None, "d_color": None,
                "local_rotation": node_attrs["local_rotation"]}


class SkeletonModel:
    """scene/skeleton_model.py:9-85, method for method: the deformation module, its Adam with one group per entry of
    ``trainable_parameters()``, the exponential learning-rate schedule, checkpoints."""

    def __init__(self, is_blender=False, d_rot_as_res=True, **kwargs):
        self.deform = SkeletonWarp(is_blender=is_blender, d_rot_as_res=d_rot_as_res, **kwargs).cuda()
        self.name = self.deform.name
        self.optimizer = None
        self.spatial_lr_scale = 1
        self.d_rot_as_res = d_rot_as_res

    @property
    def reg_loss(self):
        return self.deform.reg_loss

    def step(self, xyz, time_emb, **kwargs):
        return self.deform(xyz, time_emb, **kwargs)

    def train_setting(self, training_args):  # :24-39
        from .gaussian_model import get_expon_lr_func
        groups = [{"params": g["params"], "lr": training_args.deform_mlp_lr_init, "name": g["name"]}
                  for g in self.deform.trainable_parameters()]
        self.optimizer = torch.optim.Adam(groups, lr=0.0, eps=1e-15)
        self.deform_scheduler_args = get_expon_lr_func(lr_init=training_args.deform_mlp_lr_init,
                                                       lr_final=training_args.deform_mlp_lr_final,
                                                       lr_delay_mult=training_args.deform_mlp_lr_delay_mult,
                                                       max_steps=training_args.deform_mlp_lr_max_steps)
        if self.name == "node":
            self.deform.as_gaussians.training_setup(training_args)

    def update_learning_rate(self, iteration, warmup_stage):  # :74-82
        lr = 5e-4 if warmup_stage else self.deform_scheduler_args(iteration)
        for group in self.optimizer.param_groups:
            group["lr"] = lr

    def update(self, iteration):  # :84-85
        self.deform.update(iteration)

    # ---- checkpoints (scene/skeleton_model.py:43-72) --------------------------------------------------------------
    def save_weights(self, model_path, iteration):
        import os
        out = os.path.join(model_path, "skeleton/iteration_{}".format(iteration))
        os.makedirs(out, exist_ok=True)
        torch.save(self.deform.state_dict(), os.path.join(out, "skeleton.pth"))

    def save_joints(self, model_path, iteration, d_nodes, idx):  # :49-56
        import os
        out = os.path.join(model_path, "skeleton/iteration_{}".format(iteration))
        os.makedirs(out, exist_ok=True)
        parents = self.deform.parents.cpu()
        tag = "/t" + str(idx).zfill(3)
        write_to_obj(self.deform.nodes[:, :3].detach().cpu(), out + "/template_nodes.obj", parents)
        write_to_obj(d_nodes.detach().cpu(), out + tag + "_d_nodes.obj", parents)
        write_to_obj(self.deform.control_nodes.detach().cpu(), out + tag + "_control_nodes.obj", parents)

    def load_weights(self, model_path, iteration=-1):
        import os
        root = os.path.join(model_path, "skeleton")
        if iteration == -1:  # searchForMaxIteration (utils/system_utils.py)
            its = [int(f.split("_")[-1]) for f in os.listdir(root)] if os.path.isdir(root) else []
            if not its:
                return False
            iteration = max(its)
        path = os.path.join(root, "iteration_{}/skeleton.pth".format(iteration))
        if not os.path.exists(path):
            return False
        self.deform.load_state_dict(torch.load(path, map_location=self.deform.nodes.device))
        self.deform.parents = self.deform.parents.int()  # :68
        return True


def write_to_obj(points, path, parents=None):
    """Wavefront OBJ: one ``v x y z`` row per point, then — with a parent array — one 1-based ``l child parent`` row per
    non-root entry (the format of skeleton_utils/visualization.py's writer of the same name)."""
    with open(path, "w") as f:
        for p in points:
            f.write("v %f %f %f\n" % (float(p[0]), float(p[1]), float(p[2])))
        if parents is not None:
            for j in range(1, len(parents)):
                f.write("l %d %d\n" % (j + 1, int(parents[j]) + 1))
