#!/usr/bin/env python
"""bench.py — one "step" = one pass of RigGS's per-frame hot path over one frame:
   PoseMLP(t) -> skeleton deform (FK + skinning/LBS, HIP) -> render glue + rasterizer forward (HIP)
   -> given dL/dimage -> rasterizer backward (HIP) -> deform backward (HIP) -> PoseMLP backward,
producing gradients for every Gaussian and skeleton parameter.  No loss, optimizer,
densification or logging.  N ranks = N independent frames per step (weak scaling) followed by one
RCCL all-reduce of the flat gradient buffer of the replicated parameters.

Prints ONE JSON line on rank 0 (contract in the task statement) with `roofline` (dominant kernel,
timed live with HIP events on its launch stream) and `cpu_baseline` (the CPU oracle timed on the host cores).
"""
import argparse
import json
import math
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

WORKLOAD = dict(N=300_000, J=24, H=800, W=800, seed=1237)  # BASELINE.json metric: 300k Gaussians @800x800, 24 joints
HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: HBM3E 8 TB/s spec
if os.environ.get("RIGGS_BENCH_TEST_WORKLOAD"):  # tests only (tests/test_gpu_api.py): a tiny scene, "N,J,H,W"
    _n, _j, _h, _w = (int(v) for v in os.environ["RIGGS_BENCH_TEST_WORKLOAD"].split(","))
    WORKLOAD.update(N=_n, J=_j, H=_h, W=_w)


class Pipe:
    convert_SHs_python = False
    compute_cov3D_python = False
    debug = False


def build_workload(rank: int, device: str, surface: bool = False):
    """The headline scene (a deep translucent cloud) or, ``surface``, the opaque-skin scene of ``dense_scene_timing``."""
    from riggs_amd import synth
    from riggs_amd.gaussian_model import GaussianModel
    from riggs_amd.skeleton import SkeletonWarp
    w = WORKLOAD
    sc = (synth.make_surface_scene if surface else synth.make_scene)(w["N"], w["J"], w["seed"])
    # (the opaque-skin scene has been timed from azimuth 45 degrees since round 2: kept, so that its numbers stay comparable)
    cam = synth.look_at_camera(w["H"], w["W"], azimuth_deg=45.0 * (rank + (1 if surface else 0)), fid=0.37 + 0.05 * rank).to(device)
    gm = GaussianModel.from_tensors(sc["xyz"], sc["features_dc"], sc["features_rest"], sc["scaling"], sc["rotation"],
                                    sc["opacity"], device=device)
    torch.manual_seed(w["seed"])
    sw = SkeletonWarp(joints=sc["joints"], parent_indices=sc["parents"], K=-1, hyper_dim=8,
                      use_skinning_weight_mlp=False, use_template_offsets=False).to(device)
    sw._node_radius.data = sc["node_radius"].to(device)
    with torch.no_grad():  # small seeded pose so that the PoseMLP output is a plausible articulation
        sw.pose_net.rotation_predictor.weight.mul_(0.1)
        sw.pose_net.translation_predictor.weight.mul_(0.1)
    return sc, cam, gm, sw


def _train_args():
    """The reference's optimisation defaults (arguments/__init__.py OptimizationParams)."""
    from types import SimpleNamespace
    return SimpleNamespace(percent_dense=0.01, position_lr_init=0.00016, position_lr_final=0.0000016,
                           position_lr_delay_mult=0.01, position_lr_max_steps=30000, feature_lr=0.0025,
                           opacity_lr=0.05, scaling_lr=0.001, rotation_lr=0.001)


def params_of(gm, sw):
    return gm.parameters() + [sw._node_radius] + list(sw.pose_net.parameters())


def make_step(cam, gm, sw, gimg, arena, world, allreduce, frame_entry=False):
    """One eagerly issued frame.  ``frame_entry``: through riggs_amd.frame.deform_render (ONE autograd node over one C call per
    direction) instead of the reference's two calls skeleton.step() + render() (two nodes, seven C calls)."""
    from riggs_amd.frame import deform_render
    from riggs_amd.render import render
    bg = torch.zeros(3, device=gimg.device)
    params = params_of(gm, sw)
    t_in = sw.expand_time(cam.fid)

    def step():
        for p in params:
            p.grad = None
        if frame_entry:
            pkg = deform_render(cam, gm, sw, Pipe, bg, arena=arena)
        else:
            dv = sw(gm.get_xyz.detach(), t_in, motion_mask=gm.motion_mask)  # skeleton.step() (train_rig.py:411)
            pkg = render(cam, gm, Pipe, bg, dv["d_xyz"], dv["d_rotation"], dv["d_scaling"], fused=True, arena=arena)
        pkg["render"].backward(gimg)
        if world > 1:
            allreduce()  # RCCL over xGMI: one all-reduce of the flat gradient buffer, averaged
        return pkg
    return step


def eager_api_timing(cam, gm, sw, gimg, steps=200):
    """Secondary number (NOT the metric): the SAME workload issued eagerly, launch by launch — what a trainer that is not
    rewritten around a captured graph gets.  (a) the reference's own two calls per iteration, ``skeleton.step()`` then
    ``render()`` (train_rig.py:411, 488), each a HIP-backed autograd node; (b) ``riggs_amd.frame.deform_render``: the same frame
    as one node over riggs_frame_forward / riggs_frame_backward.  Host-bound either way; wall clock per frame."""
    from riggs_amd.rasterizer import RasterArena
    from riggs_amd import _torch_ext as TX
    res = {"torch_extension_nodes": bool(TX.available())}
    for name, fe, ext in (("two_calls_ms", False, True), ("two_calls_ctypes_nodes_ms", False, False), ("frame_entry_ms", True, True)):
        if not ext and not TX.available():
            continue
        TX.enable(ext)
        step = make_step(cam, gm, sw, gimg, RasterArena(tight_lists=_tight()), 1, None, frame_entry=fe)
        for _ in range(20):
            step()
        blocks = []
        for _ in range(3):  # (host-bound: the first hundreds of frames of a process run slower — allocator, clocks — the fastest block is the steady state)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(steps):
                step()
            torch.cuda.synchronize()
            blocks.append(round((time.perf_counter() - t0) / steps * 1e3, 4))
        res[name] = min(blocks)
        res[name.replace("_ms", "_blocks_ms")] = blocks
    TX.enable(True)
    res["what"] = ("same workload, every launch issued eagerly: skeleton.step() + render() as the reference calls them — two autograd nodes, "
                   "the PyTorch extension's C++ nodes when lib/libriggs_torch.so is built (torch_extension_nodes), beside the same two calls "
                   "through the ctypes nodes — and riggs_amd.frame.deform_render (one C call per direction); fastest of three blocks of "
                   "%d frames; host-bound, moves with the box's host; not the headline metric" % steps)
    return res


def heads_timing(sc, gm, iters=5):
    from riggs_amd.skeleton import SkeletonWarp
    J = sc["joints"].shape[0]
    sw = SkeletonWarp(joints=sc["joints"], parent_indices=sc["parents"], K=-1, hyper_dim=8).to(gm.get_xyz.device)
    x = gm.get_xyz.detach()
    q = torch.nn.functional.normalize(torch.randn(J, 4, device=x.device), dim=-1).requires_grad_(True)
    gt = torch.zeros(3, device=x.device, requires_grad=True)
    gx, gr = torch.randn_like(x), torch.randn(x.shape[0], 4, device=x.device)
    params = [p for g in sw.trainable_parameters() for p in g["params"]]
    res = {}
    for fused in (False, True):
        sw.use_fused_heads(fused)

        def it():
            for p in params + [q, gt]:
                p.grad = None
            o = sw.deform_by_pose(x, {"local_rotation": q, "global_trans": gt}, None)
            torch.autograd.backward((o["d_xyz"], o["d_rotation"]), (gx, gr))
        for _ in range(2):
            it()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(iters):
            it()
        torch.cuda.synchronize()
        res[fused] = (time.perf_counter() - t0) / iters
    # forward + data gradient + weight gradient = 3 x (2 flops per weight) per Gaussian, both heads
    flops = 3.0 * x.shape[0] * sum(2 * p.numel() for name in ("skinning_weight_mlp", "detail_net")
                                   for p in getattr(sw, name).parameters() if p.dim() == 2)
    MFMA_F16_DENSE_TFLOPS = 2500.0  # /opt/skills/guides/MI355X_MICROARCH.md: ~2.5 PF dense fp16 / bf16
    return {"what": "deform_by_pose forward+backward with WeightMLP and DeformMLP on, %d Gaussians; not the headline metric" % x.shape[0],
            "ms_fp32_gemms": round(res[False] * 1e3, 3), "ms_fused_mfma": round(res[True] * 1e3, 3), "fused_operand_format": "fp16 (fp32 accumulation, device-side gradient scaling)",
            "tflop_per_iteration": round(flops / 1e12, 3),
            "mfma_frac": round(flops / res[True] / 1e12 / MFMA_F16_DENSE_TFLOPS, 4),
            "mfma_frac_note": "the MLPs' flops (forward, data gradient, weight gradient) over the WHOLE iteration's time (incl. skinning, "
                              "embeddings, operand packing) over the dense fp16 MFMA peak"}


def train_step_heads_timing(dev, surface=False, steps=40):
    """Secondary number (NOT the metric): the WHOLE training iteration of the shipped stage-2 recipe after iteration 15 000 (85 %
    of the reference's 100 000: scripts/run_demo.py:32) as one hipGraph — both per-Gaussian MLP heads on (fused fp16-MFMA kernels;
    the WeightMLP's backward on the rows that carry a gradient), the reference's objective for that phase (image loss +
    template-offsets L2 over all Gaussians + template_fixed on the template frame: train_rig.py:446-456, 474-482, folded into the
    heads' / PoseMLP's backward launches), FusedAdam of the Gaussians and the skeleton incl. both heads — next to the same
    iteration with the heads off; ``heads_ms`` is the difference (what the heads cost inside the iteration)."""
    from riggs_amd.graph import GraphedTrainStep
    from riggs_amd.optim import FusedAdam
    from riggs_amd.skeleton import SkeletonWarp
    w = WORKLOAD
    res = {}
    for heads in (False, True):
        sc, cam, gm, _sw = build_workload(0, dev, surface=surface)
        torch.manual_seed(w["seed"])
        sw = SkeletonWarp(joints=sc["joints"], parent_indices=sc["parents"], K=-1, hyper_dim=8, use_skinning_weight_mlp=heads,
                          use_template_offsets=heads).to(dev)
        sw._node_radius.data = sc["node_radius"].to(dev)
        if heads:
            sw.use_fused_heads(True)
        gm.training_setup(_train_args(), capturable=True)
        opt = FusedAdam([{"params": g_["params"], "lr": 5e-4, "name": g_["name"]} for g_ in sw.trainable_parameters()],
                        lr=0.0, eps=1e-15, capturable=True)
        bg = torch.zeros(3, device=dev)
        from riggs_amd.graph import GraphedFrame
        img0 = GraphedFrame(gm, sw, cam, bg, params_of(gm, sw)).capture().run()["render"].detach().clone()
        target = (img0 + 0.05 * torch.randn(img0.shape, generator=torch.Generator().manual_seed(w["seed"] + 7)).to(dev)).clamp_(0.0, 1.0)
        for p in gm.parameters() + list(sw.parameters()):
            p.grad = None
        gts = GraphedTrainStep(gm, sw, cam, bg, target, [gm.optimizer, opt], lambda_dssim=0.2, sparse_grad_rows=True, tight_lists=_tight(),
                               lambda_template_offsets=1.0 if heads else None, lambda_template_fixed=100.0 if heads else None)
        gts.capture()
        for _ in range(5):
            gts.run()
        blocks = []
        for _ in range(3):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(steps):
                gts.run()
            torch.cuda.synchronize()
            blocks.append((time.perf_counter() - t0) / steps)
        R = gts.check()
        res[heads] = {"ms": min(blocks) * 1e3, "R": int(R), "loss": float(gts.out["loss"])}
        if heads:
            res["live"] = int(sw._fh_w.last_live_count) / float(w["N"])
            res["t_loss"] = float(gts.out["template_offsets_loss"])
            # the template frame's replay (x1e3 L2, template_fixed on): same launches, device-side coefficients
            gts.run(is_template=True)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(10):
                gts.run()
            torch.cuda.synchronize()
            res["template_ms"] = (time.perf_counter() - t0) / 10 * 1e3
        del gts
    on, off = res[True]["ms"], res[False]["ms"]
    return {"value": round(1e3 / on, 2), "unit": "iters/s", "ms_per_step": round(on, 4), "heads_off_ms_per_step": round(off, 4),
            "heads_ms": round(on - off, 4), "template_frame_ms_per_step": round(res["template_ms"], 4),
            "weight_mlp_live_rows": round(res["live"], 4), "tile_instances_R": res[True]["R"], "final_loss": round(res[True]["loss"], 6),
            "template_offsets_loss": res["t_loss"],
            "what": "one hipGraph per training iteration, %s scene, both MLP heads on (fused fp16 MFMA; the DeformMLP's pose through its biases; WeightMLP backward on the rows with a "
                    "gradient), objective = image loss + template-offsets L2 + template_fixed (train_rig.py:446-482), FusedAdam incl. the heads; "
                    "heads_ms = this minus the same iteration with the heads off; fastest of three blocks of %d; not the headline metric"
                    % ("opaque-skin" if surface else "headline", steps)}


def next_rows_timing(sc, gm, cam, iters=20):
    """Secondary numbers (NOT the metric) for two more SURVEY.md §8-f rows: the stage-1 control-node deformation (rank 4) and
    the skeleton projection loss (rank 2), forward + backward each, eager launches timed with events."""
    from riggs_amd.control_nodes import control_node_blend
    from riggs_amd.loss import cal_skeleton_loss, sampling_steps
    dev = gm.get_xyz.device
    x = gm.get_xyz.detach()
    N, M, H, K = x.shape[0], 512, 8, 3
    g = torch.Generator().manual_seed(3)
    P = lambda t: t.to(dev).requires_grad_(True)  # noqa: E731
    nodes = P(torch.cat([x[torch.randint(0, N, (M,), generator=g).to(dev)].cpu(), 1e-2 + 0.02 * torch.randn(M, H, generator=g)], -1))
    feature, mask = P(0.02 * torch.randn(N, H + 1, generator=g)), P(torch.rand(N, 1, generator=g))
    radius, weight = P(-1.9 + 0.3 * torch.randn(M, generator=g)), P(0.5 * torch.randn(M, 1, generator=g))
    attrs = {"d_xyz": P(0.1 * torch.randn(M, 3, generator=g)), "d_rotation": P(0.2 * torch.randn(M, 4, generator=g)),
             "d_scaling": P(0.05 * torch.randn(M, 3, generator=g)), "local_rotation": P(0.3 * torch.randn(M, 4, generator=g))}
    go = [torch.randn(N, w, device=dev) for w in (3, 4, 3)]
    leaves = [nodes, feature, mask, radius, weight] + list(attrs.values())

    def stage1():
        for t in leaves:
            t.grad = None
        o = control_node_blend(x, feature, mask, nodes, radius, weight, attrs, K=K, hyper_dim=H, local_frame=True, d_rot_as_res=True)
        torch.autograd.backward([o["d_xyz"], o["d_rotation"], o["d_scaling"]], go)

    cam.thinned = torch.stack([torch.randint(150, 650, (1500,), generator=g), torch.randint(150, 650, (1500,), generator=g)], -1).float().to(dev)
    joints = P(sc["joints"].clone())
    parents = sc["parents"].to(dev)
    steps = sampling_steps(joints, parents)

    def projection():
        joints.grad = None
        cal_skeleton_loss(joints, parents, cam, t=steps).backward()

    res = {}
    for name, fn in (("stage1", stage1), ("projection", projection)):
        for _ in range(3):
            fn()
        torch.cuda.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(iters):
            fn()
        b.record()
        torch.cuda.synchronize()
        res[name] = a.elapsed_time(b) / iters
    return {"what": "forward+backward, eager launches; not the headline metric",
            "stage1_control_node_deform_ms": round(res["stage1"], 4),
            "stage1_config": {"gaussians": N, "nodes": M, "K": K, "hyper_dim": H, "local_frame": True},
            "skeleton_projection_loss_ms": round(res["projection"], 4),
            "projection_config": {"joints": int(joints.shape[0]), "sample_points": int(steps.shape[0]) * (int(joints.shape[0]) - 1), "pixels": 1500}}


def dense_scene_timing(dev, steps=50):
    """Secondary number (NOT the metric): the same path and sizes (camera: the orbit's 45-degree position) on the DENSE-GRADIENT scene of
    riggs_amd.synth.make_surface_scene (a thin opaque skin around the bones: a surface-like capture in which a large share of
    the Gaussians receives a gradient every frame, where the headline scene — a deep translucent cloud, SURVEY.md §8-d —
    leaves 93 % of them without one).  hipGraph replay, same timing protocol."""
    from riggs_amd.graph import GraphedFrame
    w = WORKLOAD
    sc, cam, gm, sw = build_workload(0, dev, surface=True)
    params = params_of(gm, sw)
    # (the headline's own form of the frame: one graph, sparse gradient rows — 28 % of this scene's rows carry a gradient)
    gf = GraphedFrame(gm, sw, cam, torch.zeros(3, device=dev), params, sparse_grad_rows=True, tight_lists=_tight()).capture()
    g = torch.Generator().manual_seed(w["seed"] + 100)
    target = torch.rand(3, w["H"], w["W"], generator=g).to(dev)
    out = gf.run()
    gf.set_inputs(gimg=torch.sign(out["render"].detach() - target) / (3 * w["H"] * w["W"]))
    for _ in range(5):
        gf.run()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        gf.run()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / steps
    R = gf.check()
    with_grad = float((gm._opacity.grad.reshape(-1) != 0).float().mean())
    res = {"value": round(1.0 / dt, 2), "unit": "iters/s", "ms_per_step": round(dt * 1e3, 4), "tile_instances_R": int(R),
           "gaussians_with_gradient": round(with_grad, 4), "visible": round(float((out["radii"] > 0).float().mean()), 4),
           "what": "same path / sizes / form of the graph (sparse gradient rows), thin opaque skin around the bones (synth.make_surface_scene); not the headline metric"}
    # this scene's kernels (the library's HIP-event timers around eagerly issued frames, each frame behind a spinning kernel so
    # that its launches queue: the duration of a kernel BEHIND another kernel, as in the graph; every gradient row written)
    import ctypes as C
    from riggs_amd import _lib as L
    from riggs_amd.rasterizer import RasterArena
    gimg = torch.sign(out["render"].detach() - target) / (3 * w["H"] * w["W"])
    del gf
    lib = L.lib()
    step = make_step(cam, gm, sw, gimg, RasterArena(tight_lists=_tight()), 1, None)
    for _ in range(5):
        step()
    torch.cuda.synchronize()
    lib.riggs_prof_reset()
    lib.riggs_prof_enable(0xFFFFFFFF)
    for _ in range(12):
        torch.cuda._sleep(2_000_000)
        step()
        torch.cuda.synchronize()
    lib.riggs_prof_enable(0)
    tot, cnt, us = C.c_float(0), C.c_int32(0), {}
    for i in range(lib.riggs_prof_count()):
        L.check(lib.riggs_prof_read(i, C.byref(tot), C.byref(cnt)), "riggs_prof_read")
        if cnt.value:
            us[lib.riggs_prof_name(i).decode()] = round(1e3 * tot.value / cnt.value, 1)
    res["kernels_us"] = us
    return res


LISTS = "canonical"  # (--lists)
MIN_TIMED_STEPS = 100  # the timed region covers at least this many steps, whatever --steps is (blocks of exactly --steps)


def _tight(other=False):
    """riggs_raster_cfg.tight_lists of this run's frames (``other``: of the secondary measurement with the other kind)."""
    return (LISTS == "tight") != bool(other)


def other_lists_timing(dev, gimg, steps=100):
    """Secondary number (NOT the metric): the headline workload with the OTHER kind of per-tile lists — upstream's canonical
    ceil(3 sigma) squares (the library's default, and this bench's) or riggs_raster_cfg.tight_lists (``--lists tight``: the
    alpha >= 1/255 box the compositing culls with anyway, applied at emission; opt-in per arena)."""
    from riggs_amd.graph import GraphedFrame
    sc, cam, gm, sw = build_workload(0, dev)
    gf = GraphedFrame(gm, sw, cam, torch.zeros(3, device=dev), params_of(gm, sw), sparse_grad_rows=True, tight_lists=_tight(other=True)).capture()
    gf.set_inputs(gimg=gimg)
    for _ in range(10):
        gf.run()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        gf.run()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / steps
    return {"value": round(1.0 / dt, 2), "unit": "iters/s", "ms_per_step": round(dt * 1e3, 4), "tile_instances_R": int(gf.check()),
            "lists": "canonical" if LISTS == "tight" else "tight",
            "what": "same workload with the other kind of per-tile lists (see --lists); not the headline metric"}


def cycling_cameras_timing(dev, steps=64, n_cams=8, surface=False):
    """Secondary number (NOT the metric): the headline workload replayed the way a trainer uses it — a different camera every
    iteration (train_rig.py:389 draws a random one): ``n_cams`` cameras on a circle around the subject, rotated through the
    captured frame's static inputs (GraphedFrame.set_inputs: one multi-tensor copy launch per replay).  The headline
    replays ONE camera, the best case of the sparse gradient rows (previous | current rows = current rows); here the union
    is real.  Reports ms per step, the share of the Gaussians with a gradient in a frame, the share of rows rewritten
    (previous | current), and how many of the frames overflowed the instance arena (sized from the warm-up frame x 1.5)."""
    from riggs_amd import _lib as L
    from riggs_amd import synth
    from riggs_amd.graph import GraphedFrame
    w = WORKLOAD
    sc, cam0, gm, sw = build_workload(0, dev, surface=surface)
    cams = [synth.look_at_camera(w["H"], w["W"], azimuth_deg=360.0 * k / n_cams, fid=0.1 + 0.8 * k / n_cams).to(dev) for k in range(n_cams)]
    params = params_of(gm, sw)
    gf = GraphedFrame(gm, sw, cams[0], torch.zeros(3, device=dev), params, sparse_grad_rows=True, tight_lists=_tight(),
                      headroom=2.5 if surface else 1.5).capture()
    g = torch.Generator().manual_seed(w["seed"] + 7)
    gf.set_inputs(gimg=(torch.sign(torch.rand(3, w["H"], w["W"], generator=g) - 0.5) / (3 * w["H"] * w["W"])).to(dev))
    # untimed pass: per frame the rows with a gradient, the union with the previous frame's, the arena
    import ctypes as C
    off, nb = C.c_size_t(), C.c_size_t()
    L.lib().riggs_raster_backward_workspace_rows(w["N"], C.byref(off), C.byref(nb))
    bits_of = lambda: gf.backward_workspace[off.value:off.value + nb.value].clone()  # noqa: E731
    import numpy as np
    pop = lambda b: int(np.unpackbits(b.cpu().numpy()).sum())  # noqa: E731
    cur_frac, union_frac, overflows, prev = [], [], 0, None
    for k in range(2 * n_cams):
        gf.run(cam=cams[k % n_cams])
        torch.cuda.synchronize()
        try:
            gf.check()
        except L.RiggsHipError:
            overflows += 1
        b = bits_of()
        if k >= n_cams:
            cur_frac.append(pop(b) / w["N"])
            union_frac.append(pop(b | prev) / w["N"])
        prev = b
    for k in range(n_cams):
        gf.run(cam=cams[k % n_cams])
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for k in range(steps):
        gf.run(cam=cams[k % n_cams])
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / steps
    return {"value": round(1.0 / dt, 2), "unit": "iters/s", "ms_per_step": round(dt * 1e3, 4), "cameras": n_cams,
            "gaussians_with_gradient": round(sum(cur_frac) / len(cur_frac), 4),
            "gradient_rows_rewritten": round(sum(union_frac) / len(union_frac), 4), "arena_overflows": overflows,
            "what": "%s, %d cameras on a circle rotating through one captured frame (a new camera and time every replay; the forward's "
                    "walk histories are kept per view); not the headline metric"
                    % ("the opaque-skin scene of dense_gradient_scene" if surface else "same workload", n_cams)}


def _oracle_iteration(sc, cam_cpu, gimg_cpu, pose=None, deformed=None, want_saved=False):
    """One iteration of the CPU oracle (torch-CPU deform restatement + C/OpenMP rasterizer, fwd + bwd) on a scene; returns
    (image, gradient dict, R).  ``pose`` = (local_rotation, global_trans) replaces the scene's own pose.
    ``deformed`` = (d_xyz, d_rotation) computed by the OTHER side: the rasterizer then sees exactly these values (so that both
    sides take the same alpha >= 1/255 / 0.99 / T < 1e-4 decisions — a 1-ulp difference in a deformed mean flips a few) while
    the gradients still flow through the oracle's own deformation (value of theirs, derivative of ours)."""
    import numpy as np
    from oracle import deform_ref as O
    from oracle import raster_ref as RR
    tanx, tany = math.tan(cam_cpu.FoVx / 2), math.tan(cam_cpu.FoVy / 2)
    leaf = lambda t: t.clone().requires_grad_(True)  # noqa: E731
    P = {k: leaf(sc[k]) for k in ("xyz", "features_dc", "features_rest", "scaling", "rotation", "opacity",
                                  "local_rotation", "global_trans", "node_radius")}
    if pose is not None:
        P["local_rotation"], P["global_trans"] = leaf(pose[0]), leaf(pose[1])
    dv = O.deform_by_pose(P["xyz"].detach(), sc["joints"], sc["parents"], P["node_radius"], P["local_rotation"],
                          P["global_trans"], sc["motion_mask"], -1)
    if deformed is not None:
        # the HIP deformation's OWN forward values against the oracle's, BEFORE they replace them (after the substitution
        # the oracle would agree with whatever the HIP skinning produced): north_star's 1e-4 of the largest value
        for k, theirs in (("d_xyz", deformed[0]), ("d_rotation", deformed[1])):
            ours = dv[k].detach()
            scale = max(float(ours.abs().max()), 1e-30)
            e = float((theirs - ours).abs().max()) / scale
            DEFORM_PARITY[k] = float("%.3g" % e)
            if not e <= DEFORM_BAR:
                raise RuntimeError("HIP deformation forward differs from the oracle: %s max error %.3g of max|oracle|" % (k, e))
        dv = dict(dv)
        dv["d_xyz"] = dv["d_xyz"] + (deformed[0] - dv["d_xyz"]).detach()
        dv["d_rotation"] = dv["d_rotation"] + (deformed[1] - dv["d_rotation"]).detach()
    m3, op, scl, rot, shs = O.render_glue(P["xyz"], P["features_dc"], P["features_rest"], P["scaling"],
                                          P["rotation"], P["opacity"], dv["d_xyz"], dv["d_rotation"], dv["d_scaling"])
    out, saved = RR.forward(m3.detach().numpy(), op.detach().numpy(), cam_cpu.world_view_transform.numpy(),
                            cam_cpu.full_proj_transform.numpy(), cam_cpu.camera_center.numpy(), tanx, tany,
                            cam_cpu.image_height, cam_cpu.image_width, np.zeros(3, np.float32),
                            shs=shs.detach().numpy(), scales=scl.detach().numpy(), rotations=rot.detach().numpy())
    g = RR.backward(saved, gimg_cpu.numpy(), None, None)
    T = torch.from_numpy
    torch.autograd.backward([m3, op, scl, rot, shs], [T(g["means3D"]), T(g["opacities"]), T(g["scales"]),
                                                      T(g["rotations"]), T(g["shs"])])
    grads = {k: P[k].grad.numpy() for k in P if P[k].grad is not None}
    grads["means2D"] = g["means2D"]
    if want_saved:
        act = {"means3D": m3.detach(), "opacities": op.detach(), "scales": scl.detach(), "rotations": rot.detach(), "shs": shs.detach()}
        return out, grads, saved, act
    return out["color"], grads, saved.R


DEFORM_BAR = 1e-4    # HIP d_xyz / d_rotation vs the oracle's deformation, in units of max|oracle| (observed ~1e-6)
DEFORM_PARITY = {}   # filled by _oracle_iteration(deformed=...): the last comparison


def _cpu_model():
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                return line.split(":", 1)[1].strip()
    except OSError:
        pass
    import platform
    return platform.processor() or "unknown"


def _set_threads(n):
    from oracle import raster_ref as RR
    torch.set_num_threads(n)
    RR.set_threads(n)


def _config_scene(N, J, H, W, seed, chain):
    from riggs_amd import synth
    sc = synth.make_scene(N, J, seed, chain=chain)
    cam = synth.look_at_camera(H, W)
    g = torch.Generator().manual_seed(seed + 100)
    return sc, cam, torch.sign(torch.rand(3, H, W, generator=g) - 0.5) / (3 * H * W)


def _best_threads(fn, cores, samples=2):
    """Thread count for the CPU oracle: 8, 16, 32, ... up to the host's hardware threads, the faster of ``samples`` runs each
    (single samples picked 32 threads from 0.0317 s against 0.0325 s at 16 on the driver's box), stopping as soon as doubling
    no longer helps (more threads than work thrash: a baseline deserves its best configuration).  Returns
    (threads, {threads: seconds})."""
    seen, best = {}, None
    n = min(8, cores)
    while True:
        _set_threads(n)
        el = None
        for _ in range(max(1, samples)):
            t0 = time.perf_counter()
            fn()
            dt = time.perf_counter() - t0
            el = dt if el is None else min(el, dt)
        seen[n] = round(el, 4)
        if best is None or seen[n] < seen[best]:
            best = n
        elif seen[n] > 1.1 * seen[best]:
            break
        if n >= cores:
            break
        n = min(2 * n, cores)
    _set_threads(best)
    return best, seen


def _timed_config(N, J, H, W, seed, chain, warm, reps, cores):
    """Median wall time of one oracle iteration on a SURVEY.md §8-d configuration (its own seeded scene and L1-sign cotangent),
    at the best thread count of a short sweep."""
    sc, cam, gimg = _config_scene(N, J, H, W, seed, chain)
    _set_threads(min(8, cores))
    _oracle_iteration(sc, cam, gimg)  # page-in / pool start-up, outside the sweep
    thr, sweep = _best_threads(lambda: _oracle_iteration(sc, cam, gimg), cores)
    for _ in range(warm):
        _oracle_iteration(sc, cam, gimg)
    ts = []
    R = 0
    for _ in range(reps):
        t0 = time.perf_counter()
        R = _oracle_iteration(sc, cam, gimg)[2]
        ts.append(time.perf_counter() - t0)
    ts.sort()
    return {"gaussians": N, "joints": J, "image": [H, W], "tile_instances_R": int(R), "ms_median": round(1e3 * ts[len(ts) // 2], 2),
            "ms_min": round(1e3 * ts[0], 2), "runs": reps, "warmup": warm, "threads": thr, "thread_sweep_s": sweep}


def cpu_baseline(sc, cam_cpu, gimg_cpu, pose, deformed=None):
    """The CPU oracle (torch-CPU deform restatement + C/OpenMP rasterizer) on the host cores of this box: SURVEY.md §8-d's
    protocol — C1 (10k / 8-joint chain / 256^2: 3 warm-up + median of 20) and C2 (150k / 24 joints / 800^2, once) — and the
    bench workload itself: a thread-count sweep (the faster of two iterations each), then three iterations at the best count
    whose median is the reported rate; the last image / gradients are returned for the parity check."""
    cores = os.cpu_count() or 1
    quick = bool(os.environ.get("RIGGS_BENCH_TEST_WORKLOAD"))  # (tests: a tiny workload, no minutes of CPU work)
    c1 = _timed_config(10_000, 8, 256, 256, 1234 + 1, True, 1 if quick else 3, 3 if quick else 20, cores)
    c2 = None if quick else _timed_config(150_000, 24, 800, 800, 1234 + 2, False, 0, 1, cores)
    res = {}

    def one():
        res["out"] = _oracle_iteration(sc, cam_cpu, gimg_cpu, pose, deformed=deformed)
    thr, sweep = _best_threads(one, cores)
    at_best = []
    for _ in range(1 if quick else 3):  # (the reported rate: the median of three more iterations at the chosen count)
        t0 = time.perf_counter()
        one()
        at_best.append(round(time.perf_counter() - t0, 4))
    image, grads, R = res["out"]
    el = sorted(at_best)[len(at_best) // 2]
    w = WORKLOAD
    return {"value": round(1.0 / el, 5), "unit": "iters/s", "cores": thr, "kind": "port", "cpu_model": _cpu_model(),
            "host_hardware_threads": cores,
            "sample": "thread-count sweep on the bench workload (%dk Gaussians, %d joints, %dx%d, R = %d), the faster of two full "
                      "iterations per count (seconds: %s), then %d more at the best count (%d threads; seconds: %s): value = their median; "
                      "torch-CPU deform oracle + C/OpenMP rasterizer oracle fwd+bwd"
                      % (w["N"] // 1000, w["J"], w["H"], w["W"], R, json.dumps(sweep), len(at_best), thr, json.dumps(at_best)),
            "c1_10k_chain8_256": c1, "c2_150k_tree24_800": c2}, image, grads


def verify_rows_exchange(gf, rows, world):
    """Outside any timed region: one more step of a split frame whose exchanged gradients are compared with a plain all-reduce
    of the same local gradients (the row exchange must be the mean, on every rank)."""
    import torch.distributed as dist
    gf.run_a()
    gf.run_b()
    local = [g.clone() for g in rows.rows]
    rows.pack()
    rows.launch()
    rows.launch_rest()
    rows.wait()
    for g in local:
        dist.all_reduce(g)
        g /= world
    torch.cuda.synchronize()
    for got, want in zip(rows.rows, local):
        tol = 1e-5 * float(want.abs().max()) + 1e-30
        assert float((got - want).abs().max()) <= tol, "gradient-row exchange differs from the dense all-reduce"


def exchange_path_child(steps=200, scene="headline"):
    """``python bench.py --exchange-path-child`` (started by the N = 1 run, in a process of its own so that nothing RCCL does
    can touch the headline measurement): the EXACT host and device sequence of a data-parallel rank's step — split frame,
    rows.pack -> all_gather of the packed rows (communication stream, under the deformation backward) -> run_b ->
    all_reduce of the skeleton's gradients -> ordered unpack — on a real RCCL communicator of ONE rank on this GPU (the
    collectives are then the identity, but they are RCCL's kernels, streams and enqueue path).  Timed twice: the frame as two
    graphs with the five exchange calls issued eagerly between them (what ``--gpus N`` runs by default), and the whole step
    captured as ONE graph (GraphedFrame.capture_exchange: RCCL collectives inside the capture).  The exchanged gradients are
    compared with the plain frame's (one rank: they must be bitwise the same).  Prints one JSON object."""
    import faulthandler
    import socket
    import torch.distributed as dist
    from riggs_amd.dist import FlatGradAllReduce, SparseRowExchange, exchange_order, row_exchange_order
    from riggs_amd.graph import GraphedFrame
    from riggs_amd.rasterizer import RasterArena
    faulthandler.enable()
    t_child = time.perf_counter()
    note = lambda m: (sys.stderr.write("[exchange-path %.1f s] %s\n" % (time.perf_counter() - t_child, m)), sys.stderr.flush())  # noqa: E731
    dev = "cuda:0"
    torch.cuda.set_device(0)
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device(dev))
    note("process group up")
    probe = torch.ones(1024, device=dev)
    dist.all_reduce(probe)
    torch.cuda.synchronize()
    note("first collective done")
    # what riggs_amd.dist selects on every rank of a world > 1 (its _spread_pose_mlp_chain): the one-launch PoseMLP chain spread
    # over all XCDs — on one XCD it needs every compute unit of that XCD, which a collective's channel kernels take away (the
    # pinned-CU soak below shows what then happens).  This child IS a data-parallel rank's step, so it runs that placement.
    from riggs_amd import _lib as L_
    L_.check(L_.lib().riggs_pose_mlp_set_placement(0), "riggs_pose_mlp_set_placement")
    w = WORKLOAD
    sc, cam, gm, sw = build_workload(0, dev, surface=(scene == "dense"))
    ordered, _ = exchange_order(gm, sw)
    n_rows = row_exchange_order(gm, sw)[1]
    bucket = FlatGradAllReduce(ordered)
    g = torch.Generator().manual_seed(w["seed"] + 100)
    target = torch.rand(3, w["H"], w["W"], generator=g).to(dev)
    gimg = torch.zeros(3, w["H"], w["W"], device=dev)
    pkg = make_step(cam, gm, sw, gimg, RasterArena(tight_lists=_tight()), 1, bucket)()
    gimg.copy_(torch.sign(pkg["render"].detach() - target) / (3 * w["H"] * w["W"]))
    del pkg
    params = params_of(gm, sw)
    bg = torch.zeros(3, device=dev)

    def timed(fn, n):
        for _ in range(10):
            fn()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(n):
            fn()
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / n * 1e3

    def rows_for():  # (sized for every row first: the real need is read after the first step)
        return SparseRowExchange([v.view(w["N"], -1) for v in bucket.views[:n_rows]], rest=bucket.flat[bucket.offsets[n_rows]:],
                                 capacity=w["N"], force_collectives=True)
    out = {"backend": "nccl (RCCL), one-rank communicator on this GPU", "steps": steps,
           "scene": "headline (deep translucent cloud)" if scene != "dense" else "opaque skin around the bones (synth.make_surface_scene)"}
    # (0) the plain frame in this process: the reference the two exchange paths are priced against, and their gradients' oracle
    gf = GraphedFrame(gm, sw, cam, bg, params, sparse_grad_rows=False, tight_lists=_tight()).capture()
    gf.set_inputs(gimg=gimg)
    out["plain_frame_ms"] = round(timed(gf.run, steps), 4)
    note("plain frame timed")
    R = gf.check()
    want = [p.grad.detach().clone() for p in params]
    del gf

    def verify(tag):
        # (pack / unpack move the rows bit for bit; the bound is there because float atomics reorder the sums of the
        # compositing backward from one replay to the next)
        torch.cuda.synchronize()
        for p, ref in zip(params, want):
            scale = float(ref.abs().max())
            assert float((p.grad - ref).abs().max()) <= 2e-4 * max(scale, 1e-30), (tag, tuple(p.shape))
    # (1) two graphs, the exchange's calls eager between them
    rows = rows_for()
    gf = GraphedFrame(gm, sw, cam, bg, params, split_backward=True, sparse_grad_rows=True, tight_lists=_tight()).capture()
    gf.set_inputs(gimg=gimg)
    rows.workspace, rows.record_rows = gf.backward_workspace, True

    def step_eager():
        gf.run_a()
        rows.pack()
        rows.launch()
        gf.run_b()
        rows.launch_rest()
        rows.wait()
    note("split frame captured")
    step_eager()
    torch.cuda.synchronize()
    note("first exchanged step done")
    assert rows.check(), "row segments overflowed"
    rows.resize(min(w["N"], int(rows.need * 1.1) + 256))
    out["rows_needed"], out["segment_MB"] = int(rows.need), round(rows.segment.numel() * 4 / 1e6, 3)
    out["two_graphs_eager_collectives_ms"] = round(timed(step_eager, steps), 4)
    assert gf.check() == R and rows.check()
    verify("eager collectives")
    need = rows.need
    del gf, rows
    out["tile_instances_R"] = int(R)
    out["verified"] = "exchanged gradients equal the plain frame's (2e-4 of max: float atomics reorder sums between replays)"
    # (printed now: a failure of the capture below has been a segmentation fault inside hipStreamEndCapture, not an exception)
    print(json.dumps(dict(out, one_graph_error="the process died while capturing the step with its RCCL collectives")), flush=True)
    if os.environ.get("RIGGS_BENCH_CAPTURE_COLLECTIVES", "1") == "0":
        dist.destroy_process_group()
        return
    # (2) ONE graph: the collectives captured with the frame
    try:
        rows = SparseRowExchange([v.view(w["N"], -1) for v in bucket.views[:n_rows]], rest=bucket.flat[bucket.offsets[n_rows]:],
                                 capacity=min(w["N"], int(need * 1.1) + 256), force_collectives=True, validity=bucket)
        rows.record_rows = True
        gf = GraphedFrame(gm, sw, cam, bg, params, split_backward=True, sparse_grad_rows=True, tight_lists=_tight())
        gf.set_inputs(gimg=gimg)
        note("capturing the step with its collectives")
        gf.capture_exchange(rows)
        note("captured")
        out["one_graph_ms"] = round(timed(gf.run, steps), 4)
        assert gf.check() == R and rows.check()
        verify("captured collectives")
        out["one_graph"] = ("frame + gated pack + all_gather + validity flag + all_reduce + unpack captured as one hipGraph (an invalid "
                            "frame marks its segment / raises the bucket's validity slot: every rank's optimizer is gated on them)")
        # soak: the PoseMLP chain's bounded hand-offs next to asynchronous RCCL collectives, many times over — the sticky status
        # word says whether ANY of the replays lost one (that replay would have been a skipped step on every rank)
        n_soak = int(os.environ.get("RIGGS_BENCH_SOAK", "10000"))
        t0 = time.perf_counter()
        for _ in range(n_soak):
            gf.run()
        torch.cuda.synchronize()
        st = gf._pose_status()
        out["soak"] = {"replays": n_soak, "ms_per_step": round((time.perf_counter() - t0) / max(n_soak, 1) * 1e3, 4),
                       "pose_handoff_timeouts": int(st[0][st[1]].item()) if st is not None else None,
                       "arena_flags": int(gf.arena.static_counters[1].item()) & 3, "exchange_status_clean": bool(rows.check()),
                       "invalid_frames": bool(rows.invalid_frame)}
        # ... and with compute units TAKEN AWAY for the whole soak, as a collective library's channel kernels take them at W = 8:
        # n workgroups that each hold a whole CU (riggs_debug_pin_cus) on a stream of their own.  The one-launch PoseMLP chain
        # hands its layers over between CO-RESIDENT workgroups: does it still find room?  (placement 1 = the chain on one XCD,
        # this graph; placement 0 = the chain spread over all XCDs, what riggs_amd.dist selects for world sizes > 1.)
        if os.environ.get("RIGGS_BENCH_PIN_SOAK", "1") != "0":
            out["soak_pinned_cus"] = pinned_cus_soak(gf, dev, gm, sw, cam, bg, params, gimg)
    except Exception as e:  # (recorded, not hidden: the eager-collective number above stands on its own)
        out["one_graph_error"] = "%s: %s" % (type(e).__name__, str(e)[:300])
    print(json.dumps(out), flush=True)
    dist.destroy_process_group()


def pinned_cus_soak(gf_one_xcd, dev, gm, sw, cam, bg, params, gimg, counts=(16, 32, 64, 128), replays=None):
    """Replays of a captured frame while ``k`` compute units are held by spinning workgroups on another stream: per k the time per
    replay and the PoseMLP chain's lost hand-offs (sticky word, cleared between runs).  Returns {placement: {k: {...}}}."""
    from riggs_amd import _lib as L
    from riggs_amd.graph import GraphedFrame
    n_rep = int(os.environ.get("RIGGS_BENCH_PIN_SOAK_REPLAYS", "2000")) if replays is None else replays
    PIN_LIMIT_MS = 3000  # (a pinner leaves on its own after this long: bounds a soak whose replays wait for it instead of running beside it)
    res = {"replays": n_rep, "what": "k workgroups holding a whole CU's LDS each spin on a second stream for the whole soak "
                                      "(riggs_debug_pin_cus); per k: ms per replay and PoseMLP hand-off time-outs (sticky word)"}

    def soak(gf, k, n):
        t_enter = time.perf_counter()
        st = gf._pose_status()
        if st is not None:
            st[0][st[1]] = 0
        # (the stop word lives in DEVICE memory and is raised by a fill on a third stream: pinned host memory is not coherent
        # for a running kernel by default — HIP_HOST_COHERENT=0 — and a pinner that never sees its stop word runs to its time limit)
        # One pair of never-reused words per soak: the per-XCD L2s are not coherent with each other, and a pinner on another XCD
        # that found the PREVIOUS soak's raised stop word still cached (the allocator hands the same block out again) left at
        # once — seen as every other soak running unpinned (the probe below tells).
        at = soak.count = getattr(soak, "count", -1) + 1
        stop, started = flags[128 * at:128 * at + 1], flags[128 * at + 64:128 * at + 65]  # (256 bytes apart: a cache line each, never reused)
        side, ctrl = torch.cuda.Stream(), torch.cuda.Stream()
        cur = torch.cuda.current_stream()
        cur.synchronize()
        if k:
            with torch.cuda.stream(side):
                L.check(L.lib().riggs_debug_pin_cus(k, stop.data_ptr(), PIN_LIMIT_MS, started.data_ptr(), L.stream_ptr()), "riggs_debug_pin_cus")
            t_wait = time.perf_counter()
            while int(started.item()) < k and time.perf_counter() - t_wait < 5.0:
                time.sleep(0.001)
        resident = int(started.item()) if k else 0
        # (are the pinned units really gone?  a chip-filling library GEMM, timed with events on this stream beside them)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.mm(probe_a, probe_a)
        e0.record()
        for _ in range(5):
            torch.mm(probe_a, probe_a)
        e1.record()
        e1.synchronize()
        probe_ms = e0.elapsed_time(e1) / 5
        t_warm = time.perf_counter()
        for _ in range(20):
            gf.run()
        gf.stream.synchronize()
        cur.synchronize()
        t_warm = time.perf_counter() - t_warm
        # Did the replays run BESIDE the pinners?  Seen on this stack: a graph that holds RCCL's collectives, and the one-XCD chain
        # with a few units of its XCD taken, do not start while the pinners are resident — the twenty replays above then last as
        # long as the pinners' time limit, and what is timed below runs on an unpinned chip.  Such a soak says nothing about lost
        # hand-offs; it is marked, not reported as a pass.
        waited = bool(k and t_warm > 0.5 * PIN_LIMIT_MS * 1e-3)
        t0 = time.perf_counter()
        for _ in range(n):
            gf.run()
        gf.stream.synchronize()
        cur.synchronize()
        dt = (time.perf_counter() - t0) / n * 1e3
        word = int(st[0][st[1]].item()) if st is not None else None
        beside = not (waited and not word)  # (slow warm-up replays WITH time-outs on record ran beside the pinners: that is the finding)
        arena_flags = int(gf.arena.static_counters[1].item()) & 3
        with torch.cuda.stream(ctrl):
            stop.fill_(1)
        ctrl.synchronize()
        t_stop = time.perf_counter()
        side.synchronize()
        if k and time.perf_counter() - t_stop > 1.0:
            raise RuntimeError("the CU pinner did not see its stop word")
        sys.stderr.write("[pinned soak %.1f s] k = %d, %d replays, %.4f ms each\n" % (time.perf_counter() - t_enter, k, n, dt))
        base = soak.probe0 = probe_ms if not k else getattr(soak, "probe0", probe_ms)  # (the unpinned chip's GEMM time: the k = 0 soak runs first)
        held = bool(k) and probe_ms > 1.25 * base  # (were the units really gone when the replays started?)
        return {"pinned_cus_resident": resident, "ms_per_step": round(dt, 4), "pose_handoff_timeouts_word": word, "arena_flags": arena_flags,
                "probe_gemm_ms": round(probe_ms, 4), "units_held_by_the_probe": held, "ran_beside_the_pinners": bool(beside and (held or not k)),
                "warmup_20_replays_s": round(t_warm, 2)}
    probe_a = torch.randn(4096, 4096, device=dev)
    flags = torch.zeros(128 * 32, dtype=torch.int32, device=dev)
    torch.cuda.synchronize()
    # (the caller captured its graph under placement 0 — the chain spread over all XCDs: what every rank of a world > 1 runs)
    one = {}
    for k in (0,) + tuple(counts):
        one[str(k)] = soak(gf_one_xcd, k, n_rep if k else min(n_rep, 2000))
    res["all_xcds_placement: frame + exchange in one graph"] = one
    gf0 = GraphedFrame(gm, sw, cam, bg, params, sparse_grad_rows=False, tight_lists=_tight()).capture()
    gf0.set_inputs(gimg=gimg)
    spread = {}
    for k in (0,) + tuple(counts):
        spread[str(k)] = soak(gf0, k, n_rep if k else min(n_rep, 2000))
    res["all_xcds_placement: plain frame"] = spread
    del gf0
    res["any_timeout_all_xcds_placement"] = bool(max(v["pose_handoff_timeouts_word"] or 0 for grp in (one, spread) for v in grp.values()))
    res["soaks_that_did_not_run_beside_their_pinners"] = [name + " k=" + k_ for name, grp in (("one graph with the exchange", one), ("plain frame", spread))
                                                          for k_, v in grp.items() if not v["ran_beside_the_pinners"]]
    # the contrast: the single-GPU default (chain on ONE XCD, 5-7 us faster on an idle device) with the same CUs taken away
    L.check(L.lib().riggs_pose_mlp_set_placement(1), "riggs_pose_mlp_set_placement")
    try:
        gf1 = GraphedFrame(gm, sw, cam, bg, params, sparse_grad_rows=False, tight_lists=_tight()).capture()
        gf1.set_inputs(gimg=gimg)
        plain1 = {}
        for k in (0, counts[0], counts[-1]):
            plain1[str(k)] = soak(gf1, k, min(n_rep, 60))
        res["one_xcd_placement (single-GPU default): plain frame"] = plain1
        del gf1
    finally:
        L.check(L.lib().riggs_pose_mlp_set_placement(0), "riggs_pose_mlp_set_placement")
    return res


def capture_probe_child():
    """``python bench.py --capture-probe-child`` (one per rank, started by ``captured_collectives_work``): joins a SECOND
    rendezvous of all ranks, captures the exchange's collective pattern — an async all-gather and an async all-reduce on a second
    communicator, issued from the capturing stream, compute between launch and wait — in a hipGraph, replays it and checks the
    values.  Exit code 0 = RCCL collectives survive capture + replay across these ranks (a failure has been a segmentation
    fault inside hipStreamEndCapture, which is why this runs in a process of its own)."""
    import torch.distributed as dist
    rank, world, local = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"]), int(os.environ.get("LOCAL_RANK", "0"))
    ndev = max(1, torch.cuda.device_count())
    dev = torch.device("cuda:%d" % (local % ndev))
    torch.cuda.set_device(dev)
    dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
    g2 = dist.new_group()
    x = torch.full((1 << 18,), float(rank + 1), device=dev)
    y = torch.zeros(world << 18, device=dev)
    z = torch.zeros(1 << 18, device=dev)

    def body():
        w1 = dist.all_gather_into_tensor(y, x, async_op=True)
        z.copy_(x * 2.0)
        w2 = dist.all_reduce(z, group=g2, async_op=True)
        w1.wait()
        w2.wait()
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        for _ in range(2):
            body()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, stream=s):
        body()
    for _ in range(3):
        g.replay()
    torch.cuda.synchronize()
    ok = float(z[0]) == 2.0 * world * (world + 1) / 2 and all(float(y[(r << 18) + 5]) == r + 1 for r in range(world))
    dist.barrier()
    dist.destroy_process_group()
    raise SystemExit(0 if ok else 4)


def captured_collectives_work(rank, world, local_rank, dev):
    """Do RCCL collectives survive a hipGraph capture across THESE ranks?  Every rank runs ``capture_probe_child`` in a child
    process (own rendezvous on MASTER_PORT + 29); the answer is the minimum over the ranks, so all of them take the same path."""
    import subprocess
    import torch.distributed as dist
    env = dict(os.environ, RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(local_rank),
               MASTER_ADDR=os.environ.get("MASTER_ADDR", "127.0.0.1"), MASTER_PORT=str(int(os.environ.get("MASTER_PORT", "29500")) + 29))
    try:
        rc = subprocess.run([sys.executable, os.path.abspath(__file__), "--capture-probe-child"], env=env, capture_output=True,
                            timeout=180).returncode
    except subprocess.TimeoutExpired:
        rc = -1
    ok = torch.tensor([1.0 if rc == 0 else 0.0], device=dev)
    dist.all_reduce(ok, op=dist.ReduceOp.MIN)
    return bool(ok.item() > 0.5)


def exchange_path_timing(scene="headline", soak=None):
    """Runs ``exchange_path_child`` in a child process (bounded by a time-out) and returns its JSON object, plus ``form``: which
    form of the step its ``ms`` is — what ``--gpus N --exchange-graph auto`` would pick on this stack."""
    import subprocess
    last = None
    env = dict(os.environ)
    if soak is not None:
        env.setdefault("RIGGS_BENCH_SOAK", str(soak))
        env.setdefault("RIGGS_BENCH_PIN_SOAK", "0")  # (the pinned-CU soak runs once, in the headline scene's child)
    for attempt in range(2):  # (one retry: the child's rendezvous port is picked and released before RCCL binds it)
        try:
            r = subprocess.run([sys.executable, os.path.abspath(__file__), "--exchange-path-child", "--lists", LISTS, "--scene", scene],
                               capture_output=True, text=True, timeout=300, env=env)
        except subprocess.TimeoutExpired:
            last = {"error": "the exchange-path child did not finish within 300 s"}
            continue
        for line in reversed(r.stdout.strip().splitlines()):
            if line.startswith("{"):
                try:
                    got = json.loads(line)
                except ValueError:
                    continue
                one = "one_graph_ms" in got
                got["captured_collectives_work"] = one  # (the capture itself is the probe here: RCCL's kernels inside a hipGraph)
                got["form"] = "one_graph" if one else "two_graphs_eager_collectives"
                got["ms"] = got.get("one_graph_ms", got.get("two_graphs_eager_collectives_ms"))
                return dict(got, attempts=attempt + 1)
        last = {"error": "exchange-path child failed (rc %d): %s" % (r.returncode, (r.stderr or r.stdout)[-600:])}
    return last


PARITY_BAR = 5e-5  # share of a tensor's elements allowed beyond 1e-4 of max|oracle| (observed: <= 4e-6)


def parity_at_bench_size(hip_image, hip_grads, ora_image, ora_grads, pose_net=None, hip_pose_grads=None, fid=None):
    """HIP frame (the one the graph replays) vs the CPU oracle at the FULL bench workload: per tensor the largest error in
    units of max|oracle|, the fraction of elements beyond 1e-4 of it (north_star's bar) and the fraction beyond the
    per-element bound |a - b| <= 1e-4 |b| + 1e-6 max|b|.  The oracle's rasterizer is fed the HIP side's deformed means and
    rotations (values; the derivatives are the oracle's own), so both sides take the same alpha >= 1/255 / 0.99-cap / T < 1e-4
    decisions and the bar is the sharp one: <= PARITY_BAR of a tensor's elements beyond 1e-4, the J bone-radius gradients and
    the PoseMLP parameter gradients (a float64 torch copy of the network on the CPU, fed the oracle's dL/dpose) within 1e-4 of
    their largest entry."""
    import numpy as np
    out = {}

    def one(name, a, b, small=False):
        a, b = np.asarray(a, np.float64).reshape(-1), np.asarray(b, np.float64).reshape(-1)
        scale = max(float(np.abs(b).max()), 1e-30)
        err = np.abs(a - b)
        out[name] = {"max_rel": float("%.3g" % (err.max() / scale)), "outlier_frac": float("%.3g" % float((err > 1e-4 * scale).mean())),
                     "per_element_outlier_frac": float("%.3g" % float((err > 1e-4 * np.abs(b) + 1e-6 * scale).mean()))}
        return err.max() / scale
    one("image", hip_image, ora_image)
    worst_small = 0.0
    for k in ora_grads:
        if k in hip_grads:
            r = one("dL/d_" + k, hip_grads[k], ora_grads[k])
            if np.asarray(ora_grads[k]).size < 1000:
                worst_small = max(worst_small, r)
    if pose_net is not None and hip_pose_grads is not None and "local_rotation" in ora_grads:
        net = pose_net  # (a float64 CPU copy of the skeleton module taken when the HIP gradients were)
        for p in net.pose_net.parameters():
            p.grad = None
        na = net.get_pose_info(net.expand_time(fid.detach().cpu().double()))
        torch.autograd.backward([na["local_rotation"], na["global_trans"]],
                                [torch.from_numpy(ora_grads["local_rotation"]).double(), torch.from_numpy(ora_grads["global_trans"]).double()])
        ref = [p.grad for p in net.pose_net.parameters()]
        tot = max(float(g.abs().max()) for g in ref)
        e = max(float((g - h.double().cpu()).abs().max()) for g, h in zip(ref, hip_pose_grads)) / max(tot, 1e-30)
        out["dL/d_pose_net"] = {"max_rel": float("%.3g" % e), "tensors": len(ref)}
        worst_small = max(worst_small, e)
    if DEFORM_PARITY:  # checked (and enforced) by _oracle_iteration before the HIP values replaced the oracle's
        out["deform_forward_vs_oracle_max_rel"] = dict(DEFORM_PARITY, bar=DEFORM_BAR)
    worst = max(v["outlier_frac"] for v in out.values() if "outlier_frac" in v)
    out["worst_outlier_frac"] = worst
    out["worst_small_tensor_max_rel"] = float("%.3g" % worst_small)
    out["bar"] = ("every tensor: <= %g of its elements beyond 1e-4 of max|oracle|; dL/d node_radius and dL/d PoseMLP parameters within "
                  "1e-4 of their largest entry (asserted; the oracle rasterizes the HIP side's deformed means)" % PARITY_BAR)
    if worst > PARITY_BAR or worst_small > 1e-4:
        raise SystemExit("bench.py: HIP frame disagrees with the CPU oracle at the bench workload: %s" % json.dumps(out))
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-graph", action="store_true", help="issue every launch eagerly instead of replaying a hipGraph")
    ap.add_argument("--profile-all", action="store_true", help="print a per-kernel event-timer table to stderr")
    ap.add_argument("--exchange", choices=("rows", "dense"), default="rows",
                    help="N > 1: gradient-row exchange (packed rows of the Gaussians with a gradient) or the dense two-phase all-reduce")
    ap.add_argument("--metric-only", action="store_true", help="skip the secondary timings (train step, heads, next rows, dense scene)")
    ap.add_argument("--exchange-graph", choices=("auto", "on", "off"), default="auto",
                    help="N > 1: capture the whole data-parallel step (frame + pack + collectives + unpack) as ONE hipGraph; "
                         "auto = when a probe (child processes, own rendezvous) shows that RCCL collectives survive a capture here")
    ap.add_argument("--lists", choices=("tight", "canonical"), default="canonical",
                    help="per-tile instance lists: 'canonical' = upstream's ceil(3 sigma) squares, the library's default and what "
                         "north_star's ordering / indexing parity is stated on; 'tight' = riggs_raster_cfg.tight_lists, opt-in per "
                         "arena (the tile rectangle cut down to the tiles a Gaussian can reach with alpha >= 1/255: the instances "
                         "dropped are exactly ones the compositing skips, so image and gradients are the canonical ones — asserted "
                         "against the oracle at the bench size).  The other kind is reported beside the headline either way")
    ap.add_argument("--exchange-path-child", action="store_true", help=argparse.SUPPRESS)
    ap.add_argument("--scene", choices=("headline", "dense"), default="headline", help=argparse.SUPPRESS)
    ap.add_argument("--capture-probe-child", action="store_true", help=argparse.SUPPRESS)
    args = ap.parse_args()
    global LISTS
    LISTS = args.lists
    if args.exchange_path_child:
        return exchange_path_child(scene=args.scene)
    if args.capture_probe_child:
        return capture_probe_child()

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if args.gpus > 1 and "RANK" not in os.environ:
        # `python bench.py --gpus N` launched plainly: start the N ranks ourselves — the same command line under
        # torch.distributed.run, one process per GPU of this node, rendezvous on 127.0.0.1 (the hostname of a container may
        # not resolve) — and pass its exit code on.  Rank 0 of that launch prints the JSON line.
        import socket
        import subprocess
        with socket.socket() as sk:
            sk.bind(("127.0.0.1", 0))
            port = sk.getsockname()[1]
        env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus),
               "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
        raise SystemExit(subprocess.call(cmd, env=env))
    if world != args.gpus:
        raise SystemExit("bench.py: --gpus %d but WORLD_SIZE=%d" % (args.gpus, world))
    ndev = max(1, torch.cuda.device_count())
    torch.cuda.set_device(local_rank % ndev)  # (% ndev: lets a 1-GPU box exercise the multi-rank control flow over gloo)
    dev = "cuda:%d" % (local_rank % ndev)
    import torch.distributed as dist
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        backend = os.environ.get("RIGGS_BENCH_BACKEND", "nccl")  # nccl = RCCL over xGMI; gloo only for control-flow tests
        if backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device(dev))
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)

    from riggs_amd import _lib as L
    from riggs_amd.rasterizer import RasterArena
    lib = L.lib()
    sc, cam, gm, sw = build_workload(rank, dev)
    arena = RasterArena(tight_lists=_tight())
    # dL/dimage of an L1 loss against a seeded target, computed ONCE from an untimed render
    w = WORKLOAD
    g = torch.Generator().manual_seed(w["seed"] + 100 + rank)
    target = torch.rand(3, w["H"], w["W"], generator=g).to(dev)
    gimg = torch.zeros(3, w["H"], w["W"], device=dev)
    # ONE flat gradient bucket: the HIP backward functions write every dL/dparam straight into it (riggs_amd/dist.py),
    # so the data-parallel exchange is a single in-place RCCL all-reduce (AVG) with no pack / unpack / divide pass
    # (world > 1: the bucket is ordered for the two-phase overlapped exchange — SH / opacity / scale first, they are final
    # when the rasterizer's backward has run; xyz / rotation / skeleton after the deformation backward that still reads them)
    from riggs_amd.dist import FlatGradAllReduce, OverlappedExchange, SparseRowExchange, exchange_order, row_exchange_order
    ordered, n_first = exchange_order(gm, sw)
    assert [id(p) for p in ordered] == [id(p) for p in row_exchange_order(gm, sw)[0]]  # one bucket order serves both exchanges
    n_rows = row_exchange_order(gm, sw)[1]
    bucket = FlatGradAllReduce(ordered if world > 1 else params_of(gm, sw))
    exchange = OverlappedExchange(bucket, bucket.offsets[n_first]) if world > 1 else None
    # the gradient-row exchange (default): only the Gaussians that received a gradient travel (csrc/exchange.hip); the dense
    # two-phase all-reduce is used when asked for, or when so many rows are touched that it moves fewer bytes per link
    rows = None
    if world > 1 and args.exchange == "rows":
        rows = SparseRowExchange([v.view(w["N"], -1) for v in bucket.views[:n_rows]], rest=bucket.flat[bucket.offsets[n_rows]:],
                                 capacity=max(1024, w["N"] // 8))
    step = make_step(cam, gm, sw, gimg, arena, world, bucket)
    pkg = step()
    gimg.copy_(torch.sign(pkg["render"].detach() - target) / (3 * w["H"] * w["W"]))
    torch.cuda.synchronize()
    arena.resolve()
    R = arena.last_R
    del pkg
    for p in params_of(gm, sw):
        p.grad = None
    if not args.no_graph:
        # hipGraph replay of the frame (deform -> render -> backward); the gradient all-reduce stays eager
        from riggs_amd.graph import GraphedFrame
        import torch.distributed as dist
        params = params_of(gm, sw)
        # (a split frame skips the zero fill of untouched gradient rows only with the row exchange, which records the rows
        # it writes; the dense all-reduce rewrites every row)
        gf = GraphedFrame(gm, sw, cam, torch.zeros(3, device=dev), params, split_backward=world > 1,
                          sparse_grad_rows=(world == 1 or rows is not None), tight_lists=_tight()).capture()
        gf.set_inputs(gimg=gimg)
        if world > 1:
            assert all(g.data_ptr() == v.data_ptr() for g, v in zip([p.grad for p in bucket.params], bucket.views)), \
                "the captured gradient buffers must be the bucket's slices"

        def step():  # noqa: F811
            if world == 1:
                return gf.run()
            # frame-parallel step: the graph's gradient buffers ARE the bucket's slices, so the exchange is two in-place
            # collectives — the first (86 % of the bytes) on the links while the deformation backward still computes
            out = gf.run_a()
            if rows is not None:
                # packed rows of the touched Gaussians -> all-gather (on the links while the deformation backward runs) ->
                # small dense all-reduce of the skeleton's gradients -> ordered unpack
                rows.pack()
                rows.launch()
                gf.run_b()
                rows.launch_rest()
                rows.wait()
                return out
            exchange.launch(1)
            gf.run_b()
            exchange.launch(2)
            exchange.wait()
            return out
        if rows is not None:
            rows.workspace = gf.backward_workspace  # (the eager profiling steps below use another one)
            rows.record_rows = True
        step()
        torch.cuda.synchronize()
        assert gf.check() == R, "graphed frame disagrees with the eager frame on the instance count"
        if rows is not None:
            # size the segments from what the fullest rank needed (every rank reads the same headers: same decision everywhere)
            rows.check()
            rows.resize(int(rows.need * 1.1) + 256)
            if not rows.wins:
                rows = None
                gf = GraphedFrame(gm, sw, cam, torch.zeros(3, device=dev), params, split_backward=True, sparse_grad_rows=False, tight_lists=_tight()).capture()
                gf.set_inputs(gimg=gimg)
            else:
                step()
                torch.cuda.synchronize()
                assert rows.check(), "gradient-row segments overflowed right after sizing"
                use_graph = args.exchange_graph == "on" or (args.exchange_graph == "auto" and backend == "nccl"
                                                             and captured_collectives_work(rank, world, local_rank, dev))
                if use_graph:
                    verify_rows_exchange(gf, rows, world)
                    # the whole step as ONE graph: frame (a) -> pack -> all-gather on the communication stream -> frame (b) ->
                    # all-reduce of the skeleton's gradients -> unpack, collectives included (their first eager calls are above)
                    gf = GraphedFrame(gm, sw, cam, torch.zeros(3, device=dev), params, split_backward=True, sparse_grad_rows=True, tight_lists=_tight())
                    gf.set_inputs(gimg=gimg)
                    rows.record_rows = True
                    gf.capture_exchange(rows)
                    step = gf.run  # noqa: F811
                    step()
                    torch.cuda.synchronize()
                    assert gf.check() == R and rows.check()

    names = [lib.riggs_prof_name(i).decode() for i in range(lib.riggs_prof_count())]
    eager_step = make_step(cam, gm, sw, gimg, arena, 1, bucket)  # profiling leg: rank 0 alone, so NO collective inside

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    # EXACTLY --steps steps between a barrier + synchronize on both sides — as many such blocks, back to back, as it takes to
    # have timed at least MIN_TIMED_STEPS steps (a 20-step region of this workload is 7 ms: one scheduling hiccup of the host is
    # a tenth of it).  ms_per_step is the MEAN over all blocks, not the best one; every block's figure is in ``timed_blocks``.
    n_blocks = max(1, -(-MIN_TIMED_STEPS // max(args.steps, 1)))
    block_s = []
    for _ in range(n_blocks):
        barrier()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            step()
        barrier()
        block_s.append(time.perf_counter() - t0)
    if world > 1:  # (per block: the slowest rank's clock)
        tb = torch.tensor(block_s, device=dev, dtype=torch.float64)
        dist.all_reduce(tb, op=dist.ReduceOp.MAX)
        block_s = [float(v) for v in tb.tolist()]
    elapsed = sum(block_s) / n_blocks  # of ONE block of --steps steps
    pose_timeouts = 0
    if not args.no_graph:
        # (outside the timed region) the device-side status words of the replays just timed: PoseMLP hand-off time-outs, the
        # sort's in-launch barrier, an instance arena that overflowed — any of them raises here instead of being a silent number
        try:
            r_now = gf.check()
        except L.RiggsHipError as e:
            # Ranks that SHARE one GPU (the 2-rank test on a 1-GPU box: another process's kernels fill the device while this
            # one's PoseMLP chain waits for its workgroups to become resident) can lose a hand-off: that step's pose was NaN and a
            # trainer would have skipped it (GraphedTrainStep does, on the device).  Reported with the number, not fatal there;
            # with the device to itself (N = 1, or one process per GPU) it never happens and stays an error.
            if world == 1 or "PoseMLP" not in str(e):
                raise
            pose_timeouts = 1
            sys.stderr.write("[bench] rank %d: %s\n" % (rank, e))
            r_now = gf.check()
        assert r_now == R, "the timed frames disagree with the first frame on the instance count"
    if world > 1 and rows is not None and not args.no_graph:
        rows_ok = rows.check()
        # (a frame invalidated by a lost hand-off marks its segment: every rank skipped that step's unpack — see pose_timeouts)
        assert rows_ok or rows.invalid_frame, "a gradient-row segment overflowed inside the timed region (that step was not exchanged)"
        if rows.invalid_frame:
            pose_timeouts = max(pose_timeouts, 1)
        if not gf.exchange_in_graph:  # (with --exchange-graph the same comparison ran before the step was captured)
            verify_rows_exchange(gf, rows, world)

    # the frame that was just timed (static buffers of the graph / the last eager step): its image and gradients are
    # compared with the CPU oracle at this size in the cpu_baseline leg below
    hip_image = hip_grads = hip_pose = hip_deformed = hip_pose_grads = sw_snapshot = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        last = step()
        torch.cuda.synchronize()
        pnames = ("xyz", "features_dc", "features_rest", "opacity", "scaling", "rotation", "node_radius")
        hip_image = last["render"].detach().cpu().numpy()
        hip_grads = {k: p.grad.detach().cpu().numpy() for k, p in zip(pnames, params_of(gm, sw))}
        vg = last["viewspace_points_grad"] if "viewspace_points_grad" in dict.keys(last) else last["viewspace_points"].grad
        hip_grads["means2D"] = vg.detach().cpu().numpy()
        hip_pose_grads = [p.grad.detach().clone() for p in sw.pose_net.parameters()]
        import copy
        sw_snapshot = copy.deepcopy(sw).cpu().double()  # (the secondary timings below step the skeleton's optimizer)
        with torch.no_grad():  # the pose the PoseMLP kernels produce for this frame's time: the oracle deforms with the same one
            t_in = sw.expand_time(cam.fid)
            na = sw.get_pose_info(t_in)
            dvh = sw(gm.get_xyz.detach(), t_in, motion_mask=gm.motion_mask)
        hip_pose = (na["local_rotation"].detach().cpu(), na["global_trans"].detach().cpu())
        hip_deformed = (dvh["d_xyz"].detach().cpu(), dvh["d_rotation"].detach().cpu())
        del dvh

    # Roofline leg: every HIP kernel of the path timed live with HIP events recorded on its launch stream
    # (riggs_prof_* in include/riggs_hip.h), over eagerly issued steps of the same workload.
    import ctypes as C
    table = {}
    if rank == 0:
        tot, cnt = C.c_float(), C.c_int32()
        for _ in range(3):
            eager_step()
        torch.cuda.synchronize()
        # An event pair around a kernel that the host launches into an EMPTY queue also times the launch's way to the device
        # (round 3 read 102 us for the 76 us Adam kernel, 55 us for the 33 us loss kernel: what the rocprofv3 trace of the same
        # kernels inside a graph shows).  So every profiled sequence is issued behind a spinning kernel that holds the queue
        # while the host enqueues it: the events then bracket what a kernel costs BEHIND another kernel — the situation of a
        # graph node, and the duration the kernel trace reports.
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        torch.cuda._sleep(4_000_000)
        e1.record()
        torch.cuda.synchronize()
        cycles_per_ms = 4_000_000 / max(e0.elapsed_time(e1), 1e-3)

        def hold(ms):
            torch.cuda._sleep(int(ms * cycles_per_ms))
        lib.riggs_prof_reset()
        lib.riggs_prof_enable(0xFFFFFFFF)
        # (the Gaussian optimizer step of SURVEY.md §8-f rank 1 is timed here as well — it is NOT part of the headline
        # metric, whose definition is deform + raster forward + backward)
        gm.training_setup(_train_args())
        for _ in range(min(args.steps, 20)):
            hold(1.0)      # (an eagerly issued frame is ~0.5 ms of host time)
            eager_step()
            torch.cuda.synchronize()
        # the Adam launches back to back (learning rates 0: nothing moves)
        lrs = [g_["lr"] for g_ in gm.optimizer.param_groups]
        for g_ in gm.optimizer.param_groups:
            g_["lr"] = 0.0
        hold(3.0)
        for _ in range(min(args.steps, 20) + 5):
            gm.optimizer.step()
        for g_, lr_ in zip(gm.optimizer.param_groups, lrs):
            g_["lr"] = lr_
        # ... and the fused image loss of §8-f rank 2 (L1 + SSIM forward, dL/dimage backward) on the rendered image
        from riggs_amd.loss import l1_ssim
        img_leaf = eager_step()["render"].detach().clone().requires_grad_(True)
        torch.cuda.synchronize()
        hold(8.0)
        for _ in range(min(args.steps, 20)):
            l1v, sv = l1_ssim(img_leaf, gimg)
            (0.8 * l1v + 0.2 * (1.0 - sv)).backward()
            img_leaf.grad = None
        torch.cuda.synchronize()
        lib.riggs_prof_enable(0)
        for i, nm in enumerate(names):
            L.check(lib.riggs_prof_read(i, C.byref(tot), C.byref(cnt)), "riggs_prof_read")
            if cnt.value:
                table[nm] = round(tot.value / cnt.value, 4)
        if args.profile_all:
            sys.stderr.write("per-launch ms (HIP events): %s\n" % json.dumps(table))

    exchange_label = "two-phase in-place all-reduce (AVG) of one flat bucket, phase 1 overlapped with the deformation backward"
    if rows is not None and not args.no_graph:
        exchange_label = ("packed rows of the Gaussians with a gradient (%d of %d rows needed by the fullest rank, capacity %d, %.1f MB per "
                          "segment) all-gathered during the deformation backward + dense all-reduce of the skeleton's %d floats; ordered "
                          "unpack" % (rows.need, w["N"], rows.capacity, rows.segment.numel() * 4 / 1e6, rows.rest.numel()))
        exchange_label += ("; the whole step (frame, pack, collectives, unpack) is ONE hipGraph per rank" if gf.exchange_in_graph else
                           "; the frame is two hipGraphs, the exchange's five calls are issued eagerly between and behind them")
    elif args.no_graph:
        exchange_label = "one in-place all-reduce (AVG) of the flat bucket"
    if rank == 0:
        ms = elapsed / args.steps * 1e3
        value = world * args.steps / elapsed
        N, HW, Bn = w["N"], w["H"] * w["W"], w["J"] - 1
        # ALGORITHMIC bytes per launch of each kernel (DESIGN.md "Kernels and rooflines")
        alg_bytes = {
            # (geometry only: the SH colours of the frame are evaluated by extra workgroups of the tile sort's scatter launch —
            # csrc/color_job.h — and their bytes are counted there)
            "preprocess_fwd": N * (12 + 12 + 16 + 12 + 16 + 12 + 4) + N * (48 + 24 + 4 + 8 + 4 + 4 + 4),
            "render_fwd": R * (4 + 48) + HW * (12 + 4 + 4 + 4 + 4 + 16),
            "render_bwd": R * (4 + 48 + 36) + HW * (12 + 4 + 4 + 16),
            "preprocess_bwd": N * (276 + 24 + 1 + 4 + 48) + N * (12 + 12 + 192 + 4 + 12 + 16),
            "lbs_fwd": N * (12 + 4 + 12 + 16),
            "lbs_bwd": N * (12 + 4 + 12 + 16),
            # counting-sort binning: 3 passes over (order, tiles, rect) + the chunk table twice + the instance list
            # + the colour job: SH rows, means + residual, radius in; colour and clamp bits out
            "tile_sort": 2 * N * 16 + R * 8 + 2 * ((N + 1023) // 1024) * (((w["W"] + 15) // 16) * ((w["H"] + 15) // 16)) * 4
                         + N * (192 + 12 + 12 + 4 + 12 + 1),
            "depth_sort": N * 16 * 4,
            "adam": N * 59 * 28,  # p, g, m, v read + p, m, v written, 59 floats per Gaussian
            "loss_fwd": 3 * HW * 4 * (2 + 3),   # two images read, three derivative maps written
            "loss_bwd": 3 * HW * 4 * (3 + 2 + 1),  # three maps + two images read, dL/dimage written
        }
        dom = max((k for k in table if k in alg_bytes and k not in ("adam", "loss_fwd", "loss_bwd")), key=lambda k: table[k])  # (adam: not on the metric's path)
        dom_ms, dom_bytes = table[dom], alg_bytes[dom]
        achieved = dom_bytes / (dom_ms * 1e-3) / 1e9 if dom_ms > 0 else 0.0
        # PMC counters per launch (profiles/kernel_counters.json: rocprofv3 --pmc passes of THIS kernel version merged by
        # tools/kernel_counters.py; FETCH_SIZE / WRITE_SIZE in separate passes, read side with the gfx950 x2 wide-stream
        # correction; SQ_INSTS_VALU = wave-level vector instructions).  Counter files are measured, not live: the label
        # of the file travels with the numbers.
        try:
            pm = json.load(open(os.path.join(ROOT, "profiles", "kernel_counters.json")))
        except Exception:
            pm = {}
        kname = {"render_fwd": "render_fwd_oct_kernel", "render_bwd": "render_bwd_kernel", "preprocess_fwd": "preprocess_fwd_kernel",
                 "preprocess_bwd": "preprocess_bwd_kernel", "lbs_fwd": "lbs_forward_kernel", "lbs_bwd": "lbs_backward_bonelane_kernel",
                 "adam": "adam_step_kernel", "loss_fwd": "l1_ssim_forward_kernel", "loss_bwd": "l1_ssim_backward_kernel"}

        def counters(k):
            return pm.get(kname.get(k, ""), {})

        def counter_traffic(k):
            c = counters(k)
            return (c["read_bytes_x2_corrected"] + c["write_bytes"]) if ("read_bytes_x2_corrected" in c and "write_bytes" in c) else None

        def valu(k, ms):
            """Vector-issue time of a launch: wave-level VALU instructions over the chip's 1024 SIMDs at 2.4 GHz, priced at the
            fp32 peak rate (2 cycles per wave64 instruction, MI355X_MICROARCH.md) and at 4 cycles (what ONE wave can issue:
            transcendentals, DPP hazards and dependent chains sit between the two)."""
            c = counters(k)
            if "SQ_INSTS_VALU" not in c or ms <= 0:
                return None
            n = c["SQ_INSTS_VALU"]
            us2, us4 = n * 2 / (1024 * 2.4e9) * 1e6, n * 4 / (1024 * 2.4e9) * 1e6
            return {"wave_valu_insts": n, "issue_us_at_2cyc": round(us2, 1), "issue_us_at_4cyc": round(us4, 1),
                    "frac_of_launch_at_2cyc": round(us2 / (ms * 1e3), 3), "frac_of_launch_at_4cyc": round(us4 / (ms * 1e3), 3)}
        # The table below is timed over EAGERLY issued frames, which write every gradient row: there ``preprocess_bwd`` is the
        # 256-thread kernel with its zero fill.  The timed graph (sparse gradient rows) runs preprocess_bwd_lean_kernel instead;
        # what every node of THAT graph costs is in the rocprofv3 kernel trace of a replay, committed as
        # profiles/graph_timeline.txt (tools/profile_round.sh, tools/timeline.py) and quoted here with its label.
        graph_nodes = None
        try:
            tl = open(os.path.join(ROOT, "profiles", "graph_timeline.txt")).read().split("\n")
            nodes = []
            for ln in tl[1:]:
                f = ln.split()
                if len(f) >= 6 and f[1] == "dur":
                    nodes.append({"kernel": " ".join(f[5:]).replace("void ", "").replace("riggs::", ""), "us": float(f[2])})
            graph_nodes = {"source": "profiles/graph_timeline.txt: " + tl[0].strip(), "nodes": nodes}
        except Exception:
            pass
        per_kernel = {}
        for k in table:
            if k in alg_bytes and table[k] > 0:
                e = {"ms": table[k], "GBps": round(alg_bytes[k] / (table[k] * 1e-3) / 1e9, 1),
                     "frac_hbm": round(alg_bytes[k] / (table[k] * 1e-3) / 1e9 / HBM_PEAK_GBS, 4)}
                ct = counter_traffic(k)
                if ct is not None:
                    e["counter_bytes"] = ct
                    e["frac_hbm_counter_bytes"] = round(ct / (table[k] * 1e-3) / 1e9 / HBM_PEAK_GBS, 4)
                if k == "preprocess_bwd":
                    e["form"] = ("preprocess_bwd_kernel (256 threads, every row written: the eagerly issued profiling frames); the timed "
                                 "graph runs preprocess_bwd_lean_kernel over sparse rows — see kernels_in_timed_graph")
                per_kernel[k] = e
        traffic = counter_traffic(dom)
        compositing = dom.startswith("render")
        # bytes one step MOVES, two ways, next to the SURVEY 8-d figure (which charges R*150 B for a radix sort of the R
        # instances that this design never performs and so flatters it): (a) the sum of the path's kernels' own algorithmic
        # bytes; (b) the sum of what the PMC counters saw per launch x launches per step
        path = ("preprocess_fwd", "depth_sort", "tile_sort", "render_fwd", "render_bwd", "preprocess_bwd", "lbs_fwd", "lbs_bwd")
        moved_alg = sum(alg_bytes[k] for k in path) + 2 * 2_300_000  # (+ the PoseMLP's weights, read by either direction)
        per_step_launches = {"preprocess_fwd_kernel": 1, "rs_count_kernel": 2, "rs_scan_kernel": 2, "rs_scatter_kernel": 2, "bin_count_kernel": 1,
                             "bin_scan_kernel": 1, "bin_scatter_kernel": 1, "render_fwd_oct_kernel": 1, "render_bwd_kernel": 1,
                             "preprocess_bwd_kernel": 1, "lbs_forward_kernel": 1, "lbs_backward_bonelane_kernel": 1,
                             "lbs_backward_finish_kernel": 1, "pm_forward_fused_kernel": 1, "pm_backward_fused_kernel": 1}
        moved_ctr = None
        if all(("read_bytes_x2_corrected" in pm.get(k, {}) and "write_bytes" in pm.get(k, {})) for k in per_step_launches):
            moved_ctr = int(sum(n * (pm[k]["read_bytes_x2_corrected"] + pm[k]["write_bytes"]) for k, n in per_step_launches.items()))
        out = {
            "metric": "train iters/sec (deform+raster fwd+bwd), 300k Gaussians @800x800",
            "value": round(value, 3), "unit": "iters/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(ms, 4), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f32", "data": "synthetic",
            "pose_handoff_timeouts_rank0": pose_timeouts,
            "timed_blocks": {"blocks": n_blocks, "steps_each": args.steps, "ms_per_step": [round(b / args.steps * 1e3, 4) for b in block_s],
                             "note": "ms_per_step / value are the MEAN over these back-to-back blocks of exactly --steps steps"},
            "config": {"workload": "300k Gaussians / 24-joint skeleton / 800x800, LBS-only, SH degree 3, anisotropic, "
                                   "one frame per GPU per step", "num_gaussians": w["N"], "num_joints": w["J"],
                       "image": [w["H"], w["W"]], "tile_instances_R": R,
                       "tile_lists": "tight: rectangles cut by the alpha >= 1/255 box at emission (riggs_raster_cfg.tight_lists; same image and "
                                     "gradients, parity below)" if LISTS == "tight" else "canonical: upstream's ceil(3 sigma) squares (the library default)",
                       "parallelism": "frames x%d" % world, "exchange": None if world == 1 else exchange_label,
                       "launch": "eager" if args.no_graph else "hipGraph replay",
                       "gradient_rows": "every row written" if (args.no_graph or not gf.sparse_outputs) else
                       "rows without a gradient now and in the previous replay are not rewritten (they hold their zeros); the timed "
                       "frame's gradients are compared with the oracle below"},
            # The dominant kernel is a compositing kernel: SURVEY.md §8-d bounds those by vector issue, not by HBM.  `achieved /
            # peak / frac` stay the HBM numbers from ALGORITHMIC bytes (the contract's definition); `bound` says what actually
            # limits the kernel, `frac_hbm_counter_bytes` prices the bytes the PMC counters saw, `valu` the issue time.
            "roofline": {"kernel": dom, "bound": "valu" if compositing else "hbm", "achieved": round(achieved, 2), "peak": HBM_PEAK_GBS,
                         "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBS, 5), "traffic": traffic,
                         "frac_hbm_counter_bytes": round(traffic / (dom_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 5) if traffic else None,
                         "valu": valu(dom, dom_ms), "counters_label": pm.get("_label"),
                         "ms_per_launch": dom_ms, "algorithmic_bytes_per_launch": dom_bytes,
                         "pixel_gaussian_pairs_per_s": round(256.0 * R / (dom_ms * 1e-3), 1) if compositing else None},
            "step_bytes": {"what": "SURVEY.md 8-d bytes per iteration: N*985 + R*278 + HW*56", "bytes": int(N * 985 + R * 278 + HW * 56),
                           "frac_hbm": round((N * 985 + R * 278 + HW * 56) / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
                           "moved_algorithmic_bytes": int(moved_alg),
                           "frac_hbm_moved_algorithmic": round(moved_alg / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
                           "moved_counter_bytes": moved_ctr,
                           "frac_hbm_moved_counter": round(moved_ctr / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4) if moved_ctr else None,
                           "note": "the 8-d formula prices a 6-pass radix sort of the R instances (R*150 B) that this design replaces by a "
                                   "depth sort of N keys + a counting sort by tile: the two 'moved' figures are what the step's kernels "
                                   "actually read and write (their own algorithmic bytes; the PMC counters of profiles/kernel_counters.json)"},
            "kernels": per_kernel, "kernels_ms": table, "kernels_in_timed_graph": graph_nodes,
        }
        if world == 1 and not args.metric_only:
            out["eager_api"] = eager_api_timing(cam, gm, sw, gimg)
        if world == 1 and not args.no_graph and not args.metric_only:
            # Secondary number (NOT the metric): one WHOLE training iteration as a hipGraph — the metric's path plus the
            # fused image loss, its backward, and the capturable FusedAdam steps of the Gaussians and the skeleton
            # (SURVEY.md §8-f ranks 1-2 composed with the hot path; train_rig.py:535-554 minus logging / densification)
            from riggs_amd.graph import GraphedTrainStep
            from riggs_amd.optim import FusedAdam
            for p in params_of(gm, sw):
                p.grad = None
            bucket.unregister()
            gm.training_setup(_train_args(), capturable=True)
            sk_opt = FusedAdam([{"params": g_["params"], "lr": 5e-4, "name": g_["name"]} for g_ in sw.trainable_parameters()],
                               lr=0.0, eps=1e-15, capturable=True)
            # The iteration's target is the scene's own first render plus noise: a target the scene can be near.  (Against the
            # uniform-random image of the frame metric, Adam walks every scale up by its learning rate per step: the instance
            # count tripled within a hundred iterations and the captured iteration overflowed its arena — rounds 2-4 reported
            # the time of that truncated frame.  The count is now read back and reported, and an overflow raises.)
            img0 = gf.run()["render"].detach().clone()
            target_ts = (img0 + 0.05 * torch.randn(img0.shape, generator=torch.Generator().manual_seed(w["seed"] + 7)).to(dev)).clamp_(0.0, 1.0)
            # (the headline frame `gf` shares these gradient buffers but is not replayed any more: this frame is their only writer)
            gts = GraphedTrainStep(gm, sw, cam, torch.zeros(3, device=dev), target_ts, [gm.optimizer, sk_opt], lambda_dssim=0.2,
                                   sparse_grad_rows=True, tight_lists=_tight())
            gts.capture()
            for _ in range(5):
                gts.run()
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            n_ts = max(10, min(args.steps, 100))
            for _ in range(n_ts):
                gts.run()
            torch.cuda.synchronize()
            dt = (time.perf_counter() - t1) / n_ts
            R_ts = gts.check()  # (raises if the captured iteration's arena overflowed while the scene trained: the time would be of truncated lists)
            out["train_step"] = {"value": round(1.0 / dt, 2), "unit": "iters/s", "ms_per_step": round(dt * 1e3, 4), "tile_instances_R": int(R_ts),
                                 "includes": "deform + raster fwd/bwd + fused L1/SSIM loss fwd/bwd + FusedAdam (Gaussians, "
                                             "skeleton), one hipGraph; not the headline metric",
                                 "final_loss": round(float(gts.out["loss"]), 6)}
            # ... and the same iteration issued EAGERLY, call by call, as the reference's loop does (train_rig.py:411-554:
            # skeleton step, render, l1 + ssim, backward, the two optimizers' steps) — what an unmodified trainer gets
            from riggs_amd.loss import l1_loss, ssim
            from riggs_amd.render import render as render_fn
            del gts
            gm.training_setup(_train_args())
            sk_eager = FusedAdam([{"params": g_["params"], "lr": 5e-4, "name": g_["name"]} for g_ in sw.trainable_parameters()],
                                 lr=0.0, eps=1e-15)
            t_in = sw.expand_time(cam.fid)
            arena_e = RasterArena(tight_lists=_tight())
            bg_e = torch.zeros(3, device=dev)

            def eager_iteration():
                gm.optimizer.zero_grad(set_to_none=True)
                sk_eager.zero_grad(set_to_none=True)
                dv = sw(gm.get_xyz.detach(), t_in, motion_mask=gm.motion_mask)
                pkg_e = render_fn(cam, gm, Pipe, bg_e, dv["d_xyz"], dv["d_rotation"], dv["d_scaling"], arena=arena_e)
                loss_e = 0.8 * l1_loss(pkg_e["render"], target_ts) + 0.2 * (1.0 - ssim(pkg_e["render"], target_ts))
                loss_e.backward()
                gm.optimizer.step()
                sk_eager.step()
            for _ in range(20):
                eager_iteration()
            blocks = []
            for _ in range(3):
                torch.cuda.synchronize()
                t1 = time.perf_counter()
                for _ in range(n_ts):
                    eager_iteration()
                torch.cuda.synchronize()
                blocks.append(round((time.perf_counter() - t1) / n_ts * 1e3, 4))
            out["train_step"]["eager_ms_per_step"] = min(blocks)
            out["train_step"]["eager"] = ("the same iteration issued eagerly, call by call, as train_rig.py:411-554 does (two optimizers "
                                          "stepped one after the other); fastest of three blocks")
        if world == 1 and not args.no_graph and not args.metric_only:
            # Secondary number (NOT the metric): the deformation with both per-Gaussian MLP heads on (the stage-2 recipe,
            # SURVEY.md §8-f rank 3), forward + backward, fp32 library GEMMs vs the fused MFMA kernels (fp16 operands)
            secs = out.setdefault("bench_section_seconds", {})  # (where this run's wall clock goes: the metric itself is ~0.1 s)

            def section(key, fn, *a, **kw):
                t_s = time.perf_counter()
                r = fn(*a, **kw)
                secs[key] = round(secs.get(key, 0.0) + time.perf_counter() - t_s, 1)
                return r
            out["mlp_heads"] = section("mlp_heads", heads_timing, sc, gm)
            out["train_step_heads"] = section("train_step_heads", train_step_heads_timing, dev)
            out["train_step_heads"]["dense_scene"] = section("train_step_heads_dense", train_step_heads_timing, dev, surface=True, steps=20)
            out["next_rows"] = section("next_rows", next_rows_timing, sc, gm, cam)
            out["dense_gradient_scene"] = section("dense_gradient_scene", dense_scene_timing, dev)
            out["cycling_cameras"] = section("cycling_cameras", cycling_cameras_timing, dev)
            out["dense_scene_cycling_cameras"] = section("dense_scene_cycling_cameras", cycling_cameras_timing, dev, surface=True)
            out["canonical_lists" if LISTS == "tight" else "tight_lists"] = section("other_lists", other_lists_timing, dev, gimg)
            # the data-parallel step's host + device sequence on a one-rank RCCL communicator (a child process)
            out["exchange_path"] = section("exchange_path", exchange_path_timing)
            out["exchange_path_ms"] = out["exchange_path"].get("ms")
            # ... and on the opaque-skin scene, whose packed segments are the large ones (every Gaussian with a gradient travels)
            out["exchange_path"]["dense_scene"] = section("exchange_path_dense", exchange_path_timing, "dense", soak=1000)
        if not args.no_cpu_baseline and world == 1:  # (the CPU baseline is an N = 1 measurement)
            t_s = time.perf_counter()
            out["cpu_baseline"], ora_image, ora_grads = cpu_baseline(sc, cam.to("cpu"), gimg.cpu(), hip_pose, hip_deformed)
            out["parity_at_bench_size"] = parity_at_bench_size(hip_image, hip_grads, ora_image, ora_grads, sw_snapshot, hip_pose_grads, cam.fid)
            out.setdefault("bench_section_seconds", {})["cpu_baseline_and_parity"] = round(time.perf_counter() - t_s, 1)
        # the numbers a reader must not take the headline without, right behind it: the same path on an opaque-surface scene
        # (what a trained scene looks like), with a new camera every replay (how a trainer uses it), on the other kind of lists
        beside = {}
        for k in ("dense_gradient_scene", "cycling_cameras", "dense_scene_cycling_cameras", "tight_lists", "canonical_lists"):
            if k in out:
                beside[k] = {"iters_per_s": out[k].get("value"), "ms_per_step": out[k].get("ms_per_step")}
        if "eager_api" in out:
            beside["eager_api_two_calls_ms"] = out["eager_api"].get("two_calls_ms")
        ordered = {}
        for k in ("metric", "value", "unit"):
            ordered[k] = out[k]
        ordered["beside_the_headline"] = beside
        ordered.update((k, v) for k, v in out.items() if k not in ordered)
        print(json.dumps(ordered), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
