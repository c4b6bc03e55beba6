"""hipGraph capture of one hot-path frame (deform -> render -> backward).

The per-frame step issues ~100+ small launches (PoseMLP layers, the sorts' passes, our kernels);
replaying them from a captured hipGraph removes the host launch cost that otherwise bounds the
iteration rate (the reference loop is host-bound the same way: train_rig.py:535-554 issues every
kernel eagerly and synchronises on ``num_rendered``).  Frame-dependent inputs (camera matrices,
time, dL/dimage) live in static device buffers that ``run()`` refreshes before each replay; the
tile-instance arena has a fixed capacity inside the graph and the overflow flag is read back after
the replay (``check()``), so a frame that outgrows it is detected, never silently wrong.
"""
from __future__ import annotations

import torch

from . import _lib as L
from .rasterizer import RasterArena
from .render import render, RenderPkg
from .synth import Camera


class _Pipe:
    convert_SHs_python = False
    compute_cov3D_python = False
    debug = False


class GraphedFrame:
    def __init__(self, gm, sw, cam: Camera, bg: torch.Tensor, params, headroom: float = 1.5, fused: bool = True,
                 split_backward: bool = False, sparse_grad_rows: bool = False, tight_lists: bool = False):
        """``split_backward``: capture the frame as TWO graphs — (a) forward + rasterizer backward, (b) deformation backward
        (skinning, FK, PoseMLP) — so that a data-parallel caller can put the all-reduce of the gradients that are final after
        (a) on the links while (b) still runs (riggs_amd.dist.OverlappedExchange): ``run_a()``, ``run_b()``."""
        self.split = bool(split_backward)
        # ``sparse_grad_rows`` (opt-in): the per-Gaussian backward rewrites only the rows that have a gradient now or had one
        # in the previous replay (riggs_raster_cfg.sparse_zero).  The caller promises that THIS frame is the only writer of
        # the Gaussians' gradient buffers between its replays: an optimizer only reads them and
        # riggs_amd.dist.SparseRowExchange records what it writes (``record_rows``), but a dense in-place all-reduce, an eager
        # backward or ANOTHER captured frame over the same parameters (it is handed the same bucket slices) does write them —
        # call ``mark_all_rows()`` after such a writer, or leave this off.
        self.sparse_rows = bool(sparse_grad_rows)
        self.gm, self.sw, self.params = gm, sw, list(params)
        dev = bg.device
        self.cam = Camera(cam.image_height, cam.image_width, cam.FoVx, cam.FoVy, cam.world_view_transform.clone(),
                          cam.full_proj_transform.clone(), cam.camera_center.clone(), cam.fid.clone())
        self.bg = bg.clone()
        self.gimg = torch.zeros(3, cam.image_height, cam.image_width, device=dev)
        # ``tight_lists`` (opt-in, riggs_raster_cfg.tight_lists): per-tile lists without the instances that cannot reach
        # alpha >= 1/255 in their tile — same radii, image and gradients, a fifth fewer instances; a trainer that never reads
        # the lists wants it on.  The default keeps upstream's canonical lists.
        self.arena = RasterArena(growth=headroom, tight_lists=tight_lists)
        self.fused = fused
        self.graph = None
        self.out = None
        # optional callable issued right behind the rasterizer's backward of a split frame, INSIDE graph (a): e.g. the pack
        # of riggs_amd.dist.SparseRowExchange (it then bakes that object's segment buffer into the graph)
        self.after_raster_backward = None
        # optional exchange object (pack / launch / launch_rest / wait: riggs_amd.dist.SparseRowExchange) whose whole step —
        # pack behind the rasterizer's backward, the all-gather on the links under the deformation backward, the small dense
        # all-reduce, the ordered unpack — is captured together with the split frame as ONE graph (``capture_exchange``): a
        # rank's step is then a single graph launch instead of two plus five eager host calls
        self.exchange = None
        self.exchange_in_graph = False
        # the PoseMLP's sticky status word belongs to this object from here on (gated consumers, ``check()``): the eager
        # callers' non-blocking watcher (PoseMLP.watch) must not clear it underneath
        pn = getattr(sw, "pose_net", None)
        if pn is not None:
            pn._status_owned = True

    def release(self):
        """Give the PoseMLP's sticky status word back to the eager callers' watcher (PoseMLP.watch reports and clears time-outs
        again) and drop the captured graph: call it when this object is not replayed any more."""
        pn = getattr(self.sw, "pose_net", None)
        if pn is not None:
            pn._status_owned = False
        self.graph = None
        if self.split:
            self.graph_b = None

    def __del__(self):
        try:
            pn = getattr(getattr(self, "sw", None), "pose_net", None)
            if pn is not None:
                pn._status_owned = False
        except Exception:
            pass

    # ---- the frame's "valid" words (include/riggs_hip.h: riggs_gate), for whoever consumes its gradients on the device
    def _pose_status(self):
        pn = getattr(self.sw, "pose_net", None)
        sync = getattr(pn, "_hip_sync", None)
        if sync is None or not sync.is_cuda:
            return None
        w = int(L.lib().riggs_pose_mlp_status_word(len(pn.net), pn.net[0].out_features))
        return (sync, w, 0xFFFFFFFF) if w < sync.numel() else None

    def _arena_flags(self):
        c = self.arena.static_counters  # (this frame's counters; pinned by a capture)
        return None if c is None else (c, 1, 3)  # bit 0: the instance arena overflowed, bit 1: the depth sort's barrier timed out

    def gate_sources(self):
        """Callables for a ``riggs_amd._lib.FrameGate``: the sticky status word of the one-launch PoseMLP kernels and the
        rasterizer's overflow / sort-barrier flags of the frame rendered last through this object's arena."""
        return [self._pose_status, self._arena_flags]

    def _frame(self):
        for p in self.params:
            p.grad = None
        t_in = self.sw.expand_time(self.cam.fid)
        dv = self.sw(self.gm.get_xyz.detach(), t_in, motion_mask=self.gm.motion_mask)
        pkg = render(self.cam, self.gm, _Pipe, self.bg, dv["d_xyz"], dv["d_rotation"], dv["d_scaling"],
                     fused=self.fused, arena=self.arena)
        pkg["render"].backward(self.gimg)
        # only detached outputs are kept: a live autograd graph would pin AccumulateGrad nodes to this stream
        # (visibility_filter is left lazy: RenderPkg derives it from the static ``radii`` buffer on access)
        out = {k: (v.detach() if isinstance(v, torch.Tensor) else v) for k, v in dict.items(pkg)
               if k not in ("viewspace_points", "visibility_filter")}
        out["viewspace_points_grad"] = pkg["viewspace_points"].grad
        return RenderPkg(out, cache=False)

    def _frame_a(self):
        for p in self.params:
            p.grad = None
        t_in = self.sw.expand_time(self.cam.fid)
        dv = self.sw(self.gm.get_xyz.detach(), t_in, motion_mask=self.gm.motion_mask)
        # the residuals enter the render as leaves: the backward of (a) stops at them, (b) continues from their gradients
        self._dx = dv["d_xyz"].detach().requires_grad_(True)
        self._dr = dv["d_rotation"].detach().requires_grad_(True)
        pkg = render(self.cam, self.gm, _Pipe, self.bg, self._dx, self._dr, dv["d_scaling"], fused=self.fused, arena=self.arena)
        pkg["render"].backward(self.gimg)
        if self.after_raster_backward is not None:
            self.after_raster_backward()
        self._dv = dv
        out = {k: (v.detach() if isinstance(v, torch.Tensor) else v) for k, v in dict.items(pkg)
               if k not in ("viewspace_points", "visibility_filter")}
        out["viewspace_points_grad"] = pkg["viewspace_points"].grad
        return RenderPkg(out, cache=False)

    def _frame_b(self):
        torch.autograd.backward([self._dv["d_xyz"], self._dv["d_rotation"]], [self._dx.grad, self._dr.grad])
        self._dv = None

    def _frame_exchanged(self):
        """The split frame with the exchange's calls between its halves — eagerly in the warm-up (which also brings the
        communicators up: a collective cannot be captured before its first eager call), then inside ONE capture."""
        from .rasterizer import last_backward_workspace
        ex = self.exchange
        out = self._frame_a()
        # (pack from — and, with record_rows, record the unpacked rows in — the workspace THIS frame's backward uses)
        ex.workspace = last_backward_workspace()[0]
        ex.pack()
        ex.launch()
        self._frame_b()
        ex.launch_rest()
        ex.wait()
        return out

    def set_inputs(self, cam: Camera = None, gimg: torch.Tensor = None):
        if cam is not None:
            if (cam.image_height, cam.image_width, cam.FoVx, cam.FoVy) != (
                    self.cam.image_height, self.cam.image_width, self.cam.FoVx, self.cam.FoVy):
                raise ValueError("image size / field of view are baked into the captured graph: capture a new one")
            dst = [self.cam.world_view_transform, self.cam.full_proj_transform, self.cam.camera_center, self.cam.fid]
            src = [cam.world_view_transform, cam.full_proj_transform, cam.camera_center, cam.fid]
            if all(t.is_cuda and t.dtype == d.dtype and t.shape == d.shape for t, d in zip(src, dst)):
                torch._foreach_copy_(dst, src)  # ONE launch for the four small tensors (four are ~20 us of every replay)
            else:
                for d, t in zip(dst, src):
                    d.copy_(t, non_blocking=True)
        if gimg is not None:
            self.gimg.copy_(gimg, non_blocking=True)

    def capture(self, warmup: int = 2):
        """Eager warm-up (sizes the arena, initialises library workspaces, creates the persistent buffers) then capture."""
        import gc
        from . import rasterizer as R
        from .dist import FlatGradAllReduce, _entry
        if self.sparse_rows and getattr(self, "_own_bucket", None) is None:
            # sparse gradient rows need the Gaussians' gradient buffers to be persistent memory of their own: the slices of
            # a registered bucket (the caller's — e.g. the data-parallel one — or one made here)
            gp = list(self.gm.parameters())
            have = [_entry(p) is not None for p in gp]
            if not any(have):
                self._own_bucket = FlatGradAllReduce(gp, register=True)
            elif not all(have):
                self.sparse_rows = False
        if not hasattr(self, "stream"):
            self.stream = torch.cuda.Stream()  # (one stream per frame: a re-capture finds its persistent buffers again)
        s = self.stream
        s.wait_stream(torch.cuda.current_stream())
        self.arena.sparse_grad_rows = self.sparse_rows  # (eager frames never skip rows; they create the persistent screen-space buffer)
        with torch.cuda.stream(s):
            for _ in range(warmup):
                if self.split and self.exchange_in_graph:
                    self.out = self._frame_exchanged()
                elif self.split:
                    self.out = self._frame_a()
                    self._frame_b()
                else:
                    self.out = self._frame()
                torch.cuda.current_stream().synchronize()
                self.arena.resolve()
                self.out = None
            self.arena.top_up()  # (the captured frame must find its arena: an allocation + a history reset inside the capture replay for ever)
        torch.cuda.current_stream().wait_stream(s)
        torch.cuda.synchronize()
        for p in self.params:
            p.grad = None
        gc.collect()
        self.graph = torch.cuda.CUDAGraph()
        if self.split and self.exchange_in_graph:
            self.graph_b = None
            # (thread-local capture errors: the process group's watchdog thread polls the events of the warm-up's
            # collectives; in the default global mode such a call from ANOTHER thread, landing inside the capture,
            # invalidates it — seen as "operation not permitted when stream is capturing", now and then)
            with torch.cuda.graph(self.graph, stream=s, capture_error_mode="thread_local"):
                self.out = self._frame_exchanged()
        elif self.split:
            self.graph_b = torch.cuda.CUDAGraph()
            with torch.cuda.graph(self.graph, stream=s):
                self.out = self._frame_a()
            with torch.cuda.graph(self.graph_b, stream=s, pool=self.graph.pool()):
                self._frame_b()
        else:
            with torch.cuda.graph(self.graph, stream=s):  # same stream as the warm-up: AccumulateGrad nodes match
                self.out = self._frame()
        self.grads = [p.grad for p in self.params]  # static gradient buffers refilled by every replay
        self.backward_workspace = R.last_backward_workspace()[0]  # baked into the graph: must live as long as it does
        self.sparse_outputs = []
        wanted = list(R._LAST_SPARSE_OUTPUTS)
        if wanted:
            # the kernels leave untouched rows alone: every buffer they write must be one that LIVES between replays — a
            # parameter's .grad or the kept screen-space gradient — not an intermediate the graph's pool may reuse
            alive = {g.data_ptr(): g for g in self.grads if g is not None}
            vg = dict.get(self.out, "viewspace_points_grad")
            if vg is not None:
                alive[vg.data_ptr()] = vg
            if not all(ptr in alive for ptr in wanted):
                # (autograd kept a copy instead of the kernel's buffer for some gradient: capture again, writing every row)
                self.sparse_rows = False
                return self.capture(warmup=0)
            self.sparse_outputs = [alive[ptr] for ptr in wanted]
            self.reset_sparse_rows()
        return self

    def capture_exchange(self, exchange, warmup: int = 2):
        """Capture the split frame AND ``exchange``'s step as one graph (see ``self.exchange``): ``run()`` is then the whole
        data-parallel step.  The exchange's segment buffers are baked in (``resize`` afterwards raises); its status words stay
        readable (``exchange.check()``)."""
        if not self.split:
            raise ValueError("capture_exchange needs split_backward=True")
        self.exchange, self.exchange_in_graph = exchange, True
        if getattr(exchange, "gate", None) is None and hasattr(exchange, "gate"):
            # an invalid frame of THIS rank (NaN pose, truncated lists) must not travel: its segment is marked instead, every
            # rank skips the unpack and raises its exchange status — the word the ranks' optimizers are gated on
            exchange.gate = L.FrameGate(self.gate_sources(), device=self.bg.device)
        bucket = getattr(exchange, "validity", None)
        if bucket is not None and bucket.frame_gate is None:
            bucket.frame_gate = exchange.gate
        return self.capture(warmup=warmup)

    def recapture(self, params=None, warmup: int = 1):
        """After densification / pruning replaced the Gaussians' parameter tensors (a new N: scene/gaussian_model.py:445-514,
        every ``densification_interval`` iterations in train_rig.py:317-365): everything a captured graph has baked in — the
        graphs themselves, the instance arena and its capacity, the gradient buffers, the private gradient bucket of the
        sparse-row mode and that mode's row list — is dropped, the current parameters are picked up (``params``; a
        ``GraphedTrainStep`` collects them from the models itself) and the frame is captured again (``warmup`` eager frames first:
        they size the arena for the new N and, for a train step, ARE training iterations).  A data-parallel caller that
        published a bucket of its own registers a new one for the new tensors before calling this."""
        self.graph = None
        if self.split:
            self.graph_b = None
        self.out = self.grads = None
        own = getattr(self, "_own_bucket", None)
        if own is not None:
            own.unregister()
            self._own_bucket = None
        self.sparse_outputs = []
        self.backward_workspace = None
        self.arena = RasterArena(growth=self.arena.growth, tight_lists=self.arena.tight_lists)
        self.arena.sparse_grad_rows = self.sparse_rows
        if params is not None:
            self.params = list(params)
        for p in self.params:
            p.grad = None
        return self.capture(warmup=warmup)

    def mark_all_rows(self):
        """After another writer of the captured gradient buffers (a dense all-reduce in place ...): the next replay rewrites
        every row; the values now in the buffers are kept."""
        if self.sparse_outputs:
            from .rasterizer import mark_all_rows
            mark_all_rows(self.backward_workspace, int(self.gm.get_xyz.shape[0]))

    def reset_sparse_rows(self):
        """Zero the captured gradient buffers and the workspace's row list together: the state ``sparse_grad_rows`` starts
        from, and what to call after anything else has written into those buffers (e.g. a dense all-reduce)."""
        for t in self.sparse_outputs:
            t.zero_()
        self.backward_workspace.zero_()

    def run_a(self, cam: Camera = None, gimg: torch.Tensor = None):
        """split_backward: forward + rasterizer backward (every Gaussian gradient is final afterwards; ``_xyz`` / ``_rotation``
        gradients are still read by ``run_b``)."""
        if self.graph is None:
            self.capture()
        self.set_inputs(cam, gimg)
        self.graph.replay()
        return self.out

    def run_b(self):
        """split_backward: the deformation backward (skeleton gradients)."""
        self.graph_b.replay()

    def run(self, cam: Camera = None, gimg: torch.Tensor = None):
        """Replay one frame.  Outputs (``self.out`` dict, ``p.grad`` of every parameter) are static tensors that
        the next replay overwrites."""
        if self.graph is None:
            self.capture()
        self.set_inputs(cam, gimg)
        self.graph.replay()
        if self.split and not self.exchange_in_graph:
            self.graph_b.replay()
        return self.out

    def check(self):
        """Blocking read-back of the instance counters of the last replay; raises on arena overflow and on a timed-out
        workgroup hand-off of the one-launch PoseMLP kernels (their status word is sticky across replays)."""
        pose_net = getattr(self.sw, "pose_net", None)
        if pose_net is not None and hasattr(pose_net, "check_status"):
            pose_net.check_status()
        c = self.arena.static_counters[:2].tolist()
        R, overflow = int(c[0]) & 0xFFFFFFFF, int(c[1])
        if overflow & 2:
            raise L.RiggsHipError("the depth sort's third pass could not synchronise its workgroups inside the captured graph")
        if overflow:
            raise L.RiggsHipError("instance arena overflowed inside the captured graph (R=%d > capacity=%d): "
                                  "re-capture with more headroom" % (R, self.arena.capacity))
        return R


class GraphedTrainStep(GraphedFrame):
    """One WHOLE training iteration as a hipGraph (train_rig.py:535-554 minus logging / densification): deform -> render ->
    image loss (fused L1 + SSIM, riggs_amd.loss) -> backward -> optimizer steps (riggs_amd.optim.FusedAdam with
    ``capturable=True``: step counts and scheduled learning rates live on the device).  The ground-truth image, the camera
    and the time are static device buffers refreshed by ``run()``; ``out["loss"]`` / ``out["l1"]`` are device scalars.

    With ``thinned`` (the frame's (M, 2) silhouette-skeleton pixels) the skeleton projection loss of train_rig.py:459-470 is
    part of the iteration: the gradient is that of ``loss_img + projection_weight * cal_skeleton_loss(d_nodes, camera)``
    (``out["loss"]`` stays the image term, ``out["projection_loss"]`` the unweighted projection term); the weight is a
    device scalar the host refreshes between replays (riggs_amd.loss.ProjectionLossWeights).  The pixel buffer has a fixed
    capacity (``max_pixels``) and a device-side count, so one graph serves frames of any pixel count up to it.

    The two regularisers of the stage-2 objective that the reference adds once the MLP heads are on (85 % of its shipped
    100 000 iterations) are part of the iteration when their weights are given: ``lambda_template_offsets`` —
    ``lambda * mean(template_offsets^2)`` over ALL Gaussians, x1e3 on the template camera (train_rig.py:446-456) — and
    ``lambda_template_fixed`` — ``lambda * mean((local_rotation - (1,0,0,0))^2)`` on the template camera only (:474-482).  Both
    enter as cotangents INSIDE launches the backward makes anyway (the fused DeformMLP's gradient-scale launch,
    ``riggs_mlp_l2_grad_scale``; the PoseMLP's backward, ``riggs_pose_mlp_backward_fk``): their coefficients are device scalars
    that ``run(is_template=...)`` refreshes, their values land in ``out["template_offsets_loss"]`` /
    ``out["template_fixed_loss"]`` (unweighted means, what the reference logs).  With the fp32 (torch) heads the L2 is a second
    autograd root instead."""

    def __init__(self, gm, sw, cam: Camera, bg: torch.Tensor, gt_image: torch.Tensor, optimizers, lambda_dssim: float = 0.2,
                 headroom: float = 1.5, thinned: torch.Tensor = None, projection_weight: float = 1e-3, K=None,
                 max_pixels: int = None, sparse_grad_rows: bool = False, tight_lists: bool = False,
                 lambda_template_offsets: float = None, lambda_template_fixed: float = None, is_template: bool = False):
        params = gm.parameters() + [p for g in sw.trainable_parameters() for p in g["params"]]
        super().__init__(gm, sw, cam, bg, params, headroom=headroom, fused=True, sparse_grad_rows=sparse_grad_rows,
                         tight_lists=tight_lists)
        for o in optimizers:
            if not getattr(o, "hip_capturable", False):
                raise ValueError("GraphedTrainStep needs FusedAdam(capturable=True) optimizers")
        self.optimizers = list(optimizers)
        self.gt = gt_image.clone()
        self.lam = float(lambda_dssim)
        self.thinned = None
        if thinned is not None:
            from .loss import sampling_steps
            m = thinned.shape[0]
            self.thinned = torch.zeros(max(int(max_pixels or m), m), 2, device=bg.device)  # fixed capacity, device-side count
            self.thinned[:m] = thinned
            self.pixel_count = torch.tensor([m], dtype=torch.int32, device=bg.device)
            self.proj_weight = torch.full((), float(projection_weight), device=bg.device)
            self.cam.K = K
            with torch.no_grad():
                d_nodes = sw(gm.get_xyz.detach()[:1], sw.expand_time(self.cam.fid), motion_mask=None)["d_nodes"]
            self.proj_steps = sampling_steps(d_nodes, sw.parents)
            self.proj_parents = sw.parents.to(device=bg.device, dtype=torch.int32).clone()
        self.one = torch.ones((), device=bg.device)
        # A frame that went wrong inside the graph — the pose NaN after a lost PoseMLP hand-off, the lists truncated by an arena
        # overflow, the depth sort's barrier timed out — must not reach the parameters: the optimizers' launches read the
        # frame's status words themselves and turn into no-ops (parameters, moments, step counts bit for bit; check() counts
        # the skipped steps and repairs a lost hand-off by running the iteration once through the layered PoseMLP kernels)
        self.gate = L.FrameGate(self.gate_sources(), device=bg.device)
        for o in self.optimizers:
            o.gate = self.gate
        self.skipped_steps = 0     # replays the gate turned into no-ops, as of the last check()
        self.recovered_steps = 0   # iterations re-run eagerly after a lost hand-off
        dev = bg.device
        self.lam_t = None if lambda_template_offsets is None else float(lambda_template_offsets)
        self.lam_f = None if lambda_template_fixed is None else float(lambda_template_fixed)
        if self.lam_t is not None:
            self.template_weight = torch.zeros((), device=dev)        # lambda (x1e3 on the template frame): the torch heads' root weight
            self.template_coef = torch.zeros(1, device=dev)           # 2 lambda / (3 N): the fused head's cotangent coefficient
            self.template_loss = torch.zeros(1, device=dev)           # mean(template_offsets^2)
        if self.lam_f is not None:
            self.fixed_coef = torch.zeros(1, device=dev)              # 2 lambda / (4 J) on the template frame, else 0
            self.fixed_loss = torch.zeros(1, device=dev)              # mean((local_rotation - unit)^2)
        self.set_template_frame(is_template)

    def set_template_frame(self, is_template: bool):
        """Is the frame of the NEXT replay the template camera's (``viewpoint_cam.uid == template_idx``, train_rig.py:451,475)?
        Refreshes the regularisers' device-side coefficients (two small fills; also after a densification changed N)."""
        self.is_template = bool(is_template)
        n3 = 3.0 * max(1, int(self.gm.get_xyz.shape[0]))
        if self.lam_t is not None:
            lam = self.lam_t * (1e3 if self.is_template else 1.0)
            self.template_weight.fill_(lam)
            self.template_coef.fill_(2.0 * lam / n3)
        if self.lam_f is not None:
            j4 = 4.0 * int(self.sw.nodes.shape[0])
            self.fixed_coef.fill_(2.0 * self.lam_f / j4 if self.is_template else 0.0)

    def _frame(self):
        from .loss import image_loss, cal_skeleton_loss
        for p in self.params:
            p.grad = None
        t_in = self.sw.expand_time(self.cam.fid)
        sw = self.sw
        fused_heads = bool(getattr(sw, "_fused_heads", False))
        want_t = self.lam_t is not None and sw.use_template_offsets
        # the regularisers as cotangents inside the heads' / the PoseMLP's own backward launches (device-side coefficients)
        sw.template_l2 = (self.template_coef, self.template_loss) if (want_t and fused_heads) else None
        sw.template_fixed = (self.fixed_coef, self.fixed_loss) if self.lam_f is not None else None
        sw._fixed_folded = False
        try:
            dv = sw(self.gm.get_xyz.detach(), t_in, motion_mask=self.gm.motion_mask)
        finally:
            fixed_folded = sw._fixed_folded
            sw.template_l2 = sw.template_fixed = None  # (eager callers of the same warp carry these terms through autograd)
        pkg = render(self.cam, self.gm, _Pipe, self.bg, dv["d_xyz"], dv["d_rotation"], dv["d_scaling"],
                     fused=self.fused, arena=self.arena)
        loss, l1 = image_loss(pkg["render"], self.gt, self.lam)
        proj = None
        roots, seeds = [loss], [self.one]  # (explicit seeds: ``loss.backward()`` launches a fill for its ones)
        if self.thinned is not None:
            # several roots, one backward pass: no launches for "loss + weight * projection" and its autograd mirror
            self.cam.thinned = self.thinned
            proj, wproj = cal_skeleton_loss(dv["d_nodes"], self.proj_parents, self.cam, t=self.proj_steps, weight=self.proj_weight,
                                            pixel_count=self.pixel_count)
            roots.append(wproj)
            seeds.append(self.one)
        t_loss = f_loss = None
        if want_t:
            if fused_heads:
                t_loss = self.template_loss
            else:  # the fp32 heads: the reference's own arithmetic (utils/loss_utils.py:29-30), weighted by the device scalar
                t_loss = (sw.template_offsets ** 2).mean()
                roots.append(t_loss)
                seeds.append(self.template_weight)
        if self.lam_f is not None:
            if fixed_folded:
                f_loss = self.fixed_loss
            else:  # (a pose network the one-node path does not take: the same term through autograd)
                unit = torch.tensor([1.0, 0.0, 0.0, 0.0], device=self.bg.device)
                f_loss = ((dv["local_rotation"].reshape(-1, 4) - unit) ** 2).mean()
                roots.append(f_loss)
                seeds.append(self.fixed_coef.reshape(()) * (2.0 * int(sw.nodes.shape[0])))  # coef = 2 lambda / (4 J) -> lambda
        torch.autograd.backward(roots, seeds)
        from .optim import step_many
        step_many(self.optimizers)
        out = {k: (v.detach() if isinstance(v, torch.Tensor) else v) for k, v in dict.items(pkg)
               if k not in ("viewspace_points", "visibility_filter")}
        out["viewspace_points_grad"] = pkg["viewspace_points"].grad
        out["loss"], out["l1"] = loss.detach(), l1.detach()
        if proj is not None:
            out["projection_loss"] = proj.detach()
        if t_loss is not None:
            out["template_offsets_loss"] = t_loss.detach()
        if f_loss is not None:
            out["template_fixed_loss"] = f_loss.detach()
        return RenderPkg(out, cache=False)

    def recapture(self, params=None, warmup: int = 1):
        if params is None:
            params = self.gm.parameters() + [p for g in self.sw.trainable_parameters() for p in g["params"]]
        self.set_template_frame(self.is_template)  # (N may have changed: the L2's coefficient is 2 lambda / (3 N))
        return super().recapture(params, warmup)

    def check(self):
        """As ``GraphedFrame.check`` — but here nothing a bad replay computed has reached the parameters (the optimizer
        launches were gated on the device), so a lost PoseMLP hand-off is REPAIRED instead of raised: the sticky word is
        cleared and the iteration is run once, eagerly, through the one-launch-per-layer PoseMLP kernels (no hand-off inside a
        launch); ``skipped_steps`` / ``recovered_steps`` say what happened.  An arena overflow still raises (the graph has to
        be captured again with a larger arena); the replays since it happened were skipped steps."""
        self.skipped_steps = self.gate.read_skipped()
        st = self._pose_status()
        if st is not None and int(st[0][st[1]].item()) != 0:
            # (cleared on the frame's own stream, in front of the re-run: its gated optimizer launches must see the word down)
            self.stream.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(self.stream):
                st[0][st[1]] = 0
            self._rerun_layered()
        return super().check()

    def _rerun_layered(self):
        before = L.get_option("pose_mlp_layered")  # (a permanent switch made by PoseMLP.watch() or by the user stays as it was)
        L.set_option("pose_mlp_layered", 1)
        try:
            with torch.cuda.stream(self.stream):
                self._frame()  # (the current static inputs: the iteration the last replay skipped)
                torch.cuda.current_stream().synchronize()
                self.arena.resolve()
        finally:
            L.set_option("pose_mlp_layered", before)
        for p, g in zip(self.params, self.grads):  # the graph's own gradient buffers stay the parameters' .grad
            p.grad = g
        self.mark_all_rows()  # (the eager backward went through the same workspace: the next replay rewrites every row)
        self.recovered_steps += 1

    def run(self, cam: Camera = None, gt_image: torch.Tensor = None, thinned: torch.Tensor = None,
            projection_weight=None, is_template: bool = None):
        if is_template is not None and bool(is_template) != self.is_template:
            self.set_template_frame(is_template)
        if gt_image is not None:
            self.gt.copy_(gt_image, non_blocking=True)
        if thinned is not None:
            m = thinned.shape[0]
            if self.thinned is None or not 1 <= m <= self.thinned.shape[0]:
                raise ValueError("the captured graph holds 1..max_pixels thinned pixels: capture a new one with more room")
            self.thinned[:m].copy_(thinned, non_blocking=True)
            self.pixel_count.fill_(m)
        if projection_weight is not None:
            self.proj_weight.fill_(float(projection_weight))
        return super().run(cam, None)
