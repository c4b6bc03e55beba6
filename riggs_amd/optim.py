"""The Gaussian optimizer of the reference — ``torch.optim.Adam(l, lr=0.0, eps=1e-15)`` built by
``GaussianModel.training_setup`` (/root/reference/scene/gaussian_model.py:197-221) and stepped at
/root/reference/train_rig.py:527 — with ``step()`` as ONE HIP launch over all parameter groups
(``riggs_adam_step``, csrc/optim.hip) instead of ~10 elementwise passes per tensor.

``FusedAdam`` *is* a ``torch.optim.Adam``: same constructor, ``param_groups``, per-parameter ``state`` with the keys
``step`` / ``exp_avg`` / ``exp_avg_sq``, ``state_dict`` — so the reference's optimizer surgery (``replace_tensor_to_optimizer``,
``_prune_optimizer``, ``cat_tensors_to_optimizer``: scene/gaussian_model.py:338-420) and ``update_learning_rate`` work on it
unchanged.  Only the update itself is replaced; there is no CPU / eager fallback.
"""
from __future__ import annotations

import ctypes as C

import torch

from . import _lib as L

_MAX = 32  # tensors per launch (ADAM_MAX_GROUPS in csrc/optim.hip)
NONFINITE_WINDOWS_LIMIT = 4  # consecutive watch windows (16 steps each) with non-finite gradients before the eager guard raises


class FusedAdam(torch.optim.Adam):
    """``capturable=True`` keeps every parameter's step count as a 0-dim DEVICE tensor (the layout of
    ``torch.optim.Adam(capturable=True)``) and accepts 0-dim device tensors as a group's ``lr``: the bias corrections are
    then evaluated inside the kernel, so a captured hipGraph of the step stays correct across replays (a learning-rate
    schedule is applied with ``group["lr"].fill_(value)`` between replays)."""

    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=0, amsgrad=False, capturable=False,
                 skip_nonfinite=True, **kw):
        if weight_decay != 0 or amsgrad or kw.get("maximize", False):
            raise NotImplementedError("FusedAdam implements the reference's configuration: plain Adam "
                                      "(weight_decay=0, amsgrad=False, maximize=False)")
        kw.pop("foreach", None), kw.pop("fused", None)
        super().__init__(params, lr=lr, betas=betas, eps=eps, weight_decay=0, amsgrad=False, foreach=False)
        self.hip_capturable = bool(capturable)
        # a frame's "valid" gate (riggs_amd._lib.FrameGate; capturable mode): when any of its device words is raised — the
        # pose was NaN after a lost PoseMLP hand-off, the instance lists were truncated, the exchange unpacked nothing — the
        # update is a no-op ON THE DEVICE: parameters, moments and step counts stay bit for bit (GraphedTrainStep sets it)
        self.gate = None
        self._hip_plan = None
        # the eager mode (host-side step counts: no gate): an element whose gradient is NaN / Inf keeps its parameter and moments
        # (riggs_adam_step_guarded) — a frame poisoned by a lost PoseMLP hand-off must not destroy an unmodified trainer's run —
        # and is counted on the device; the count is looked at without blocking (riggs_amd._lib.Watch) and reported as a
        # RuntimeWarning; non-finite gradients in NONFINITE_WINDOWS_LIMIT consecutive windows raise (a diverged run must not train
        # on unnoticed).  A deviation from torch.optim.Adam, documented in INTEGRATION.md section 4; ``skip_nonfinite=False``:
        # torch's behaviour (NaN propagates into the parameters).
        self.skip_nonfinite = bool(skip_nonfinite)
        self.nonfinite_seen = 0

    @torch.no_grad()
    def step(self, closure=None):
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        _step([self])
        return loss

    # (everything that can replace state tensors behind a parameter that stays the same object drops the cached launch plan; the
    # reference's optimizer surgery — replace_tensor_to_optimizer, _prune_optimizer, cat_tensors_to_optimizer — installs NEW
    # parameter objects, which the plan's signature sees by itself)
    def load_state_dict(self, state_dict):
        self._hip_plan = None
        return super().load_state_dict(state_dict)

    def add_param_group(self, param_group):
        self._hip_plan = None
        return super().add_param_group(param_group)

    def _gather(self, by_cfg):
        """Append this optimizer's (parameter, gradient, moments, group, step) tuples to ``by_cfg`` (keyed by betas / eps), creating
        the state lazily like torch.optim.Adam."""
        cap = getattr(self, "hip_capturable", False)
        for group in self.param_groups:
            if group.get("weight_decay", 0) != 0 or group.get("amsgrad", False) or group.get("maximize", False):
                raise NotImplementedError("FusedAdam: plain Adam only")
            for p in group["params"]:
                g = p.grad
                if g is None:
                    continue
                if not (p.is_cuda and p.dtype is torch.float32 and p.is_contiguous()):
                    L.require_cuda_f32("parameter", p)
                    raise L.RiggsHipError("FusedAdam needs contiguous parameters")
                if g.is_sparse:
                    raise RuntimeError("FusedAdam does not support sparse gradients")
                st = self.state[p]
                if len(st) == 0:  # same lazy initialisation as torch.optim.Adam (device step tensor when capturable)
                    st["step"] = torch.zeros((), dtype=torch.float32, device=p.device) if cap else torch.tensor(0.0, dtype=torch.float32)
                    st["exp_avg"] = torch.zeros_like(p, memory_format=torch.preserve_format)
                    st["exp_avg_sq"] = torch.zeros_like(p, memory_format=torch.preserve_format)
                if cap and not st["step"].is_cuda:
                    st["step"] = st["step"].to(p.device)
                m, v = st["exp_avg"], st["exp_avg_sq"]
                if not (m.is_contiguous() and v.is_contiguous()):
                    st["exp_avg"], st["exp_avg_sq"] = m, v = m.contiguous(), v.contiguous()
                if not (g.is_cuda and g.dtype is torch.float32 and g.is_contiguous() and g.shape == p.shape):
                    g = L.require_cuda_f32("gradient", g, tuple(p.shape))
                key = (group["betas"][0], group["betas"][1], group["eps"])
                by_cfg.setdefault(key, []).append((p, g, m, v, group, st))


class _Chunk:
    """One Adam launch (<= 32 tensors of one (betas, eps) configuration): its marshalled argument arrays, kept across steps."""
    __slots__ = ("n", "b1", "b2", "eps", "params", "groups", "states", "moments", "pptr", "p_arr", "g_arr", "m_arr", "v_arr", "numel",
                 "lr", "lr_dev", "step_arr", "steps_i64", "keep")


class _Plan:
    """What a step of a fixed set of optimizers launches, marshalled once: a training iteration steps the same 33 tensors every time
    and only their GRADIENTS are new objects (zero_grad(set_to_none=True)), yet checking and marshalling every tensor from scratch
    was ~8 us each — 0.27 ms of host time per eagerly issued iteration.  ``sig`` = the parameter objects that had a gradient, in
    order; ``refresh`` re-validates what can change behind an unchanged signature (the parameter's storage, the moment tensors,
    the gradient's layout, the learning rates) and re-reads the gradient pointers — anything unexpected rebuilds the plan."""

    def __init__(self, optimizers, cap):
        self.cap = cap
        by_cfg = {}
        for o in optimizers:
            o._gather(by_cfg)
        self.chunks = []
        for (b1, b2, eps), items in by_cfg.items():
            for i in range(0, len(items), _MAX):
                it = items[i:i + _MAX]
                c = _Chunk()
                n = c.n = len(it)
                c.b1, c.b2, c.eps = float(b1), float(b2), float(eps)
                c.params, c.groups, c.states = [t[0] for t in it], [t[4] for t in it], [t[5] for t in it]
                c.moments = [(t[2], t[3]) for t in it]
                c.pptr = [t[0].data_ptr() for t in it]
                arr = lambda k: (C.c_void_p * n)(*[t[k].data_ptr() for t in it])  # noqa: E731
                c.p_arr, c.g_arr, c.m_arr, c.v_arr = arr(0), arr(1), arr(2), arr(3)
                c.numel = (C.c_int64 * n)(*[t[0].numel() for t in it])
                c.lr = (C.c_double * n)()
                c.lr_dev = (C.c_void_p * n)()
                c.step_arr = (C.c_void_p * n)()
                c.steps_i64 = (C.c_int64 * n)()
                c.keep = [t[1] for t in it]  # (gradients the pointers of this step belong to, incl. converted copies)
                self.chunks.append(c)
        self.sig = _signature(optimizers)[0]

    def refresh(self, grads):
        """False = something changed behind the signature: rebuild."""
        k = 0
        f32 = torch.float32
        for c in self.chunks:
            g_arr, pptr, moments, states, params = c.g_arr, c.pptr, c.moments, c.states, c.params
            for j in range(c.n):
                p, g, st = params[j], grads[k], states[j]
                k += 1
                m, v = moments[j]
                if (st.get("exp_avg") is not m or st.get("exp_avg_sq") is not v or p.data_ptr() != pptr[j] or g.dtype is not f32
                        or g.layout is not torch.strided or not g.is_cuda or not g.is_contiguous() or g.shape != p.shape):
                    return False
                g_arr[j] = g.data_ptr()
            c.keep = None
        return k == len(grads)


def _signature(optimizers):
    sig, grads = [], []
    for o in optimizers:
        for group in o.param_groups:
            key = (group["betas"][0], group["betas"][1], group["eps"])
            for p in group["params"]:
                g = p.grad
                if g is not None:
                    sig.append((id(p), key))
                    grads.append(g)
    return sig, grads


def _ordered_grads(plan, optimizers):
    """The gradients in the plan's chunk order (chunks are grouped by configuration, not by optimizer)."""
    out = []
    for c in plan.chunks:
        out += [p.grad for p in c.params]
    return out


def _step(optimizers):
    """One step of FusedAdam instances of the same mode and gate: the merged, cached launch plan (kept on the first optimizer)."""
    first = optimizers[0]
    cap, gate = bool(getattr(first, "hip_capturable", False)), getattr(first, "gate", None)
    if gate is not None and not cap:
        raise L.RiggsHipError("a gated FusedAdam keeps its step counts on the device: capturable=True")
    plan = getattr(first, "_hip_plan", None)
    others = tuple(id(o) for o in optimizers)
    ok = False
    if isinstance(plan, _Plan) and plan.cap == cap and getattr(first, "_hip_plan_for", None) == others and all(
            getattr(o, "_hip_plan", None) is not None or o is first for o in optimizers):
        sig, _ = _signature(optimizers)
        ok = sig == plan.sig and plan.refresh(_ordered_grads(plan, optimizers))
    if not ok:
        plan = _Plan(optimizers, cap)
        first._hip_plan, first._hip_plan_for = plan, others
        for o in optimizers[1:]:
            o._hip_plan = True  # (marker: load_state_dict / add_param_group of ANY member resets it and forces a rebuild)
    guard = None
    if not cap and all(getattr(o, "skip_nonfinite", False) for o in optimizers) and plan.chunks:
        guard = _nonfinite_counter(first, plan.chunks[0].params[0].device)
    _launch(plan, cap, gate, guard)
    if guard is not None and not torch.cuda.is_current_stream_capturing():
        w = first._nonfinite_watch
        before = w.n // w.period
        n = w.poll(guard, 0)
        if n > first.nonfinite_seen:
            import warnings
            warnings.warn("FusedAdam: %d gradient elements were NaN or Inf since the last report and were NOT applied (their "
                          "parameters and moments are unchanged); skip_nonfinite=False restores torch.optim.Adam's behaviour"
                          % (n - first.nonfinite_seen), RuntimeWarning, stacklevel=3)
            first.nonfinite_seen = n
            first._nonfinite_windows = getattr(first, "_nonfinite_windows", 0) + 1
            first._nonfinite_last_window = w.n // w.period
            # the guard exists for a TRANSIENT fault (one frame poisoned by a lost PoseMLP hand-off); gradients that keep
            # coming back non-finite are a diverged run, which torch.optim.Adam would have shown as NaN parameters: say so loudly
            if first._nonfinite_windows >= NONFINITE_WINDOWS_LIMIT:
                raise RuntimeError("FusedAdam: non-finite gradients in %d consecutive watch windows (%d steps): the run has diverged "
                                   "(the guard only rides out single poisoned frames)" % (first._nonfinite_windows,
                                                                                        first._nonfinite_windows * w.period))
        elif w.n // w.period != before and w.n // w.period > getattr(first, "_nonfinite_last_window", -10) + 2:
            first._nonfinite_windows = 0  # (two clean windows in a row: the fault was transient)


def _nonfinite_counter(opt, device):
    c = getattr(opt, "_nonfinite_count", None)
    if c is None or c.device != device:
        c = opt._nonfinite_count = torch.zeros(1, dtype=torch.int32, device=device)
        opt._nonfinite_watch = L.Watch(period=16)
        opt.nonfinite_seen = 0
    return c


def _launch(plan, cap, gate=None, guard=None):
    lib = L.lib()
    st_ptr = L.stream_ptr()
    gs = gate.struct() if gate is not None else None
    chunks = plan.chunks
    coef = None
    if cap:
        steps = []
        for c in chunks:
            for j in range(c.n):
                st = c.states[j]["step"]
                if not st.is_cuda:
                    st = c.states[j]["step"] = st.to(c.params[j].device)
                steps.append(st)
                c.step_arr[j] = st.data_ptr()
        if steps and gate is None:
            torch._foreach_add_(steps, 1.0)  # one multi-tensor launch; the kernels below read the new counts
        elif steps:  # ... or one launch of the library's that advances them behind the gate (and counts a skipped step)
            sp = (C.c_void_p * len(steps))(*[t.data_ptr() for t in steps])
            if len({(c.b1, c.b2) for c in chunks}) == 1:
                # ... and evaluates the bias corrections of the new counts once, for the update's launches to read (they are
                # two double-precision pow() in the prologue of each of the update's 16 384 workgroups otherwise)
                coef = getattr(plan, "coef", None)
                if coef is None or coef.numel() != 2 * len(steps) or coef.device != steps[0].device:
                    coef = plan.coef = torch.empty(2 * len(steps), device=steps[0].device)
                L.check(lib.riggs_adam_steps_advance_coef(len(steps), sp, C.byref(gs), gate.skipped.data_ptr(), chunks[0].b1, chunks[0].b2,
                                                          coef.data_ptr(), st_ptr), "riggs_adam_steps_advance_coef")
            else:
                coef = None
                L.check(lib.riggs_adam_steps_advance_gated(len(steps), sp, C.byref(gs), gate.skipped.data_ptr(), st_ptr),
                        "riggs_adam_steps_advance_gated")
    at = 0
    for c in chunks:
        n = c.n
        for j in range(n):
            lr = c.groups[j]["lr"]
            if isinstance(lr, torch.Tensor):
                if cap:
                    c.lr[j], c.lr_dev[j] = 0.0, L.require_cuda_f32("lr", lr).data_ptr()
                else:
                    c.lr[j] = float(lr)
            else:
                c.lr[j] = float(lr)
                if cap:
                    c.lr_dev[j] = None
        if cap and gate is not None and coef is not None:
            L.check(lib.riggs_adam_step_gated_coef(n, c.p_arr, c.g_arr, c.m_arr, c.v_arr, c.numel, c.lr, c.step_arr, c.lr_dev, c.b1, c.b2,
                                                   c.eps, C.byref(gs), coef.data_ptr() + 8 * at, st_ptr), "riggs_adam_step_gated_coef")
            at += n
        elif cap and gate is not None:
            L.check(lib.riggs_adam_step_gated(n, c.p_arr, c.g_arr, c.m_arr, c.v_arr, c.numel, c.lr, c.step_arr, c.lr_dev, c.b1, c.b2,
                                              c.eps, C.byref(gs), None, 0, st_ptr), "riggs_adam_step_gated")
        elif cap:
            L.check(lib.riggs_adam_step_capturable(n, c.p_arr, c.g_arr, c.m_arr, c.v_arr, c.numel, c.lr, c.step_arr, c.lr_dev,
                                                   c.b1, c.b2, c.eps, st_ptr), "riggs_adam_step_capturable")
        else:
            # (host-side step counts, as torch.optim.Adam keeps them: ONE increment and one read for the chunk)
            cpu_steps = [st["step"] for st in c.states]
            torch._foreach_add_(cpu_steps, 1)
            for j, v in enumerate(torch.stack(cpu_steps).tolist()):
                c.steps_i64[j] = int(v)
            if guard is not None:
                L.check(lib.riggs_adam_step_guarded(n, c.p_arr, c.g_arr, c.m_arr, c.v_arr, c.numel, c.lr, c.steps_i64, c.b1, c.b2, c.eps,
                                                    guard.data_ptr(), st_ptr), "riggs_adam_step_guarded")
            else:
                L.check(lib.riggs_adam_step(n, c.p_arr, c.g_arr, c.m_arr, c.v_arr, c.numel, c.lr, c.steps_i64, c.b1, c.b2, c.eps, st_ptr),
                        "riggs_adam_step")
        # the kernels write through raw pointers: tell autograd (and anything that caches derived copies of the
        # parameters by version, e.g. the bf16 weights of riggs_amd.mlp) that the tensors changed
        torch.autograd.graph.increment_version(c.params)


@torch.no_grad()
def step_many(optimizers):
    """``for o in optimizers: o.step()`` for FusedAdam instances with ONE step-count increment and one Adam launch per (betas, eps)
    configuration ACROSS the optimizers (train_rig.py:527-554 steps the Gaussians' and the skeleton's optimizers back to back: two
    launches of ~5 us floor each and a second Adam kernel otherwise).  Other optimizers in the list are stepped as they are."""
    fused = [o for o in optimizers if isinstance(o, FusedAdam)]
    caps = {bool(getattr(o, "hip_capturable", False)) for o in fused}
    # an optimizer with registered step hooks, or whose ``step`` an lr_scheduler has wrapped (it counts the calls and warns
    # when ``scheduler.step()`` comes first), must go through ITS ``step()``: the merged launch below bypasses both
    from torch.optim.optimizer import _global_optimizer_post_hooks, _global_optimizer_pre_hooks
    hooked = bool(_global_optimizer_pre_hooks or _global_optimizer_post_hooks) or any(
        o._optimizer_step_pre_hooks or o._optimizer_step_post_hooks or hasattr(o.step, "_wrapped_by_lr_sched") for o in fused)
    if len(fused) < 2 or len(caps) != 1 or hooked:
        for o in optimizers:
            o.step()
        return
    gates = {id(getattr(o, "gate", None)) for o in fused}
    if len(gates) != 1:  # (optimizers behind different gates keep their own launches)
        for o in optimizers:
            o.step()
        return
    _step(fused)
    for o in optimizers:
        if not isinstance(o, FusedAdam):
            o.step()


def densify_stats(viewspace_grad, update_filter, xyz_gradient_accum, denom, radii=None, max_radii2D=None):
    """In place: ``xyz_gradient_accum[f] += ||viewspace_grad[f, :2]||``, ``denom[f] += 1`` (add_densification_stats,
    scene/gaussian_model.py:516-518) and optionally ``max_radii2D[f] = max(max_radii2D[f], radii[f])``
    (train_rig.py:333-335) in one HIP launch."""
    N = viewspace_grad.shape[0]
    vg = L.require_cuda_f32("viewspace gradient", viewspace_grad, (N, 3))
    f = update_filter
    if f.dtype != torch.bool or not f.is_cuda or f.numel() != N:
        raise L.RiggsHipError("update_filter must be a CUDA bool tensor with one entry per Gaussian")
    f = f.reshape(-1).contiguous()
    acc = L.require_cuda_f32("xyz_gradient_accum", xyz_gradient_accum)
    den = L.require_cuda_f32("denom", denom)
    if acc.numel() != N or den.numel() != N or acc.data_ptr() != xyz_gradient_accum.data_ptr() or den.data_ptr() != denom.data_ptr():
        raise L.RiggsHipError("xyz_gradient_accum / denom must be contiguous with one entry per Gaussian")
    rp = mp = None
    if max_radii2D is not None:
        if radii is None or radii.dtype != torch.int32 or not radii.is_contiguous():
            raise L.RiggsHipError("radii must be the contiguous int32 tensor the rasterizer returned")
        mr = L.require_cuda_f32("max_radii2D", max_radii2D, (N,))
        if mr.data_ptr() != max_radii2D.data_ptr():
            raise L.RiggsHipError("max_radii2D must be contiguous")
        rp, mp = radii.data_ptr(), mr.data_ptr()
    L.check(L.lib().riggs_densify_stats(N, vg.data_ptr(), f.data_ptr(), rp, acc.data_ptr(), den.data_ptr(), mp,
                                        L.stream_ptr()), "riggs_densify_stats")
