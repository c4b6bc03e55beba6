"""The reference's recorded call sequence (tests/golden/skeleton_api_calls.json: TrainRig.train_step, render_rig.render_set /
generate_random_motion, GUI.test_step, SkeletonModel's methods — recorded against the reference's own SkeletonWarp by
tests/golden/record_api.py) replayed IN FULL on the HIP classes on the GPU: every read, write and call, every result compared
by type, shape, dtype and value (1e-4 of the tensor's largest magnitude; the recorded side is the reference on the CPU)."""
import json
import os

import pytest
import torch

from tests import api_replay as A

pytestmark = pytest.mark.gpu

HERE = os.path.dirname(os.path.abspath(__file__))
REC = json.load(open(os.path.join(HERE, "golden", "skeleton_api_calls.json")))
EVENTS = REC["events"]


def _ctor_kwargs():
    return {k: A.rebuild(d, "cpu") for k, d in REC["constructor"].items()}


def test_reference_skeleton_model_over_the_hip_warp_gpu():
    """scene/skeleton_model.py:4 swapped to riggs_amd.skeleton.SkeletonWarp: all events on the
    deformation module and on its joint Gaussians, whoever made them."""
    from riggs_amd.skeleton import SkeletonWarp
    warp = SkeletonWarp(**_ctor_kwargs()).cuda()
    A.seed_module(warp, REC["seed"])
    events = [ev for ev in EVENTS if ev["path"] in ("deform", "gs")]
    n, problems = A.replay(events, {"deform": warp}, "cuda")
    assert not problems, "\n".join(problems)
    assert n == len(events)
    warp.pose_net.check_status()
    kinds = {(ev["op"], ev["name"]) for ev in events}
    assert {("call", "__call__"), ("call", "deform_by_pose"), ("call", "node_deformation"), ("call", "get_pose_info"),
            ("get", "as_gaussians"), ("call", "update")} <= kinds


def test_mirror_skeleton_model_driven_like_the_trainer_gpu(tmp_path):
    """Both classes swapped: what train_rig.py / render_rig.py / interactive_GUI.py send to the SkeletonModel and, through its
    ``.deform``, to the SkeletonWarp."""
    from riggs_amd.skeleton import SkeletonModel
    model = SkeletonModel(**_ctor_kwargs())
    assert model.deform.nodes.is_cuda
    A.seed_module(model.deform, REC["seed"])
    events = []
    for ev in EVENTS:
        if ev["who"].startswith("scene/skeleton_model.py"):  # what the reference's SkeletonModel does inside its own methods
            continue
        if ev["op"] == "call" and ev["name"] in ("save_weights", "load_weights", "save_joints"):
            ev = json.loads(json.dumps(ev))
            ev["args"][0] = {"t": "str", "v": str(tmp_path)}
        events.append(ev)
    n, problems = A.replay(events, {"skeleton": model}, "cuda")
    assert not problems, "\n".join(problems)
    assert n == len(events)
    model.deform.pose_net.check_status()
    # the optimizer the trainer steps (train_rig.py:529) owns the tensors the kernels wrote gradients for
    d = model.step(torch.randn(64, 3, device="cuda"), model.deform.expand_time(torch.tensor([0.4], device="cuda")),
                   motion_mask=torch.ones(64, 1, device="cuda"))
    (d["d_xyz"].sum() + d["d_rotation"].sum() + d["d_nodes"].sum()).backward()
    for g in model.optimizer.param_groups:
        assert all(p.grad is not None and torch.isfinite(p.grad).all() for p in g["params"]), g["name"]
    model.optimizer.step()
