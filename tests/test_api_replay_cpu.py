"""The reference's call sequence on the skeleton objects, replayed on the HIP-backed classes (host side, no GPU).

``tests/golden/skeleton_api_calls.json`` was recorded by running the reference's own ``TrainRig.train_step`` /
``render_rig.render_set`` / ``generate_random_motion`` / ``GUI.test_step`` / ``SkeletonModel`` methods against the reference's
``SkeletonWarp`` behind recording proxies (tests/golden/record_api.py).  Here every recorded attribute read, attribute write
and call that needs no kernel is applied to ``riggs_amd.skeleton`` and every result compared (type, shape, dtype, values of
small tensors); tests/test_gpu_api_replay.py applies ALL of them on the GPU.  The two scenarios of INTEGRATION.md §2:
the reference's ``SkeletonModel`` over the HIP ``SkeletonWarp`` (events on the ``deform`` / ``gs`` paths, whoever made
them), and the mirror ``SkeletonModel`` driven by the trainer (everything the reference's callers do from outside).
"""
import json
import os
import subprocess
import sys

import pytest
import torch

from tests import api_replay as A

HERE = os.path.dirname(os.path.abspath(__file__))
REC = json.load(open(os.path.join(HERE, "golden", "skeleton_api_calls.json")))
EVENTS = REC["events"]
NEEDS_KERNELS = {"__call__", "step", "deform_by_pose", "node_deformation"}


def _ctor_kwargs(device="cpu"):
    return {k: A.rebuild(d, device) for k, d in REC["constructor"].items()}


def _brute_knn(points):
    d = torch.cdist(points.double(), points.double()) ** 2
    d.fill_diagonal_(float("inf"))
    return d.topk(3, dim=1, largest=False).values.mean(1).float()


@pytest.fixture
def host_only(monkeypatch):
    """No GPU here: ``.cuda()`` is the identity and the 3-NN distances of the J joints (csrc/knn.hip on the GPU box) come
    from a brute-force stand-in — test scaffolding; the product raises without the library's kernels."""
    import riggs_amd.knn
    monkeypatch.setattr(torch.nn.Module, "cuda", lambda self, *a, **k: self)
    monkeypatch.setattr(riggs_amd.knn, "distCUDA2", _brute_knn)


def _host_skip(inner_of_model):
    def skip(ev):
        if ev["op"] == "call" and ev["name"] in NEEDS_KERNELS:
            return True
        if ev["op"] == "get" and ev["name"] == "template_offsets":  # (the value a skipped forward leaves behind)
            return True
        if ev["op"] == "call" and ev["name"] == "save_joints":  # its d_nodes argument is fine, but it writes files: kept, see below
            return False
        return inner_of_model(ev)
    return skip


def test_recording_covers_the_callers_the_drop_in_names():
    files = {ev["who"].split(":")[0] for ev in EVENTS}
    assert {"train_rig.py", "render_rig.py", "interactive_GUI.py", "scene/skeleton_model.py"} <= files
    names = {(ev["path"], ev["op"], ev["name"]) for ev in EVENTS}
    for need in (("deform", "get", "as_gaussians"), ("deform", "call", "update"), ("deform", "call", "trainable_parameters"),
                 ("deform", "call", "expand_time"), ("deform", "call", "deform_by_pose"), ("deform", "call", "get_pose_info"),
                 ("deform", "call", "node_deformation"), ("deform", "call", "update_control_nodes"),
                 ("deform", "set", "use_template_offsets"), ("deform", "set", "use_skinning_weight_mlp"),
                 ("deform", "call", "state_dict"), ("deform", "call", "load_state_dict"), ("gs", "call", "training_setup"),
                 ("skeleton", "call", "train_setting"), ("skeleton", "call", "update_learning_rate"), ("skeleton", "call", "update"),
                 ("skeleton", "call", "step"), ("skeleton", "call", "save_weights"), ("skeleton", "call", "load_weights"),
                 ("skeleton", "get", "optimizer"), ("skeleton", "get", "d_rot_as_res")):
        assert need in names, need
    assert REC["constructor"]["K"]["v"] == -1 and "skinning" in REC["constructor"]  # train_rig.py:84's keywords


def test_reference_skeleton_model_over_the_hip_warp(host_only, tmp_path):
    """INTEGRATION.md §2's import swap: whatever the reference's SkeletonModel / trainer / renderers / GUI read, write or call
    on ``.deform`` (and on its joint Gaussians) exists on riggs_amd.skeleton.SkeletonWarp with the same results."""
    from riggs_amd.skeleton import SkeletonWarp
    kw = _ctor_kwargs()
    warp = SkeletonWarp(**kw)
    A.seed_module(warp, REC["seed"])
    events = [ev for ev in EVENTS if ev["path"] in ("deform", "gs")]
    n, problems = A.replay(events, {"deform": warp}, "cpu", skip=_host_skip(lambda ev: False))
    assert not problems, "\n".join(problems)
    assert n >= 40


def test_mirror_skeleton_model_driven_like_the_trainer(host_only, tmp_path, monkeypatch):
    """The fully swapped scenario: riggs_amd.skeleton.SkeletonModel receives what the reference's callers send to
    scene/skeleton_model.py's class (the accesses that class makes internally are its own business)."""
    from riggs_amd.skeleton import SkeletonModel
    model = SkeletonModel(**_ctor_kwargs())
    A.seed_module(model.deform, REC["seed"])
    monkeypatch.chdir(tmp_path)
    inner = lambda ev: ev["who"].startswith("scene/skeleton_model.py")  # noqa: E731
    events = []
    for ev in EVENTS:  # checkpoints go to this test's directory (the recording's temporary directory is gone)
        if ev["op"] == "call" and ev["name"] in ("save_weights", "load_weights", "save_joints"):
            ev = json.loads(json.dumps(ev))
            ev["args"][0] = {"t": "str", "v": str(tmp_path)}
        events.append(ev)
    n, problems = A.replay(events, {"skeleton": model}, "cpu", skip=_host_skip(inner))
    assert not problems, "\n".join(problems)
    assert n >= 60
    assert os.path.exists(os.path.join(str(tmp_path), "skeleton/iteration_15001/skeleton.pth"))
    assert os.path.exists(os.path.join(str(tmp_path), "skeleton/iteration_15001/t003_d_nodes.obj"))


@pytest.mark.skipif(not os.path.isdir("/root/reference/scene"), reason="needs the reference checkout (build container only)")
def test_reference_trainer_classes_run_on_the_hip_warp(tmp_path):
    """The probe of VERDICT round 4: the reference's OWN scene/skeleton_model.py with ``SkeletonWarp`` swapped for the HIP
    one — constructor with train_rig.py:84's keywords, train_setting (as_gaussians.training_setup),
    update_learning_rate, update, save / load."""
    code = r'''
import sys, types
sys.path.insert(0, %r); sys.path.insert(0, %r)
import _ref_shim as S
S.install()
import torch
from argparse import ArgumentParser
with S.quiet():
    import skeleton_utils.skeleton_warp as sw_mod
    from riggs_amd.skeleton import SkeletonWarp
    sw_mod.SkeletonWarp = SkeletonWarp                      # scene/skeleton_model.py:4 resolves to the HIP class
    from scene.skeleton_model import SkeletonModel
    from arguments import OptimizationParams
p = ArgumentParser(); opt = OptimizationParams(p).extract(p.parse_args([]))
joints = torch.rand(8, 3); parents = torch.tensor([-1, 0, 1, 1, 3, 0, 5, 2])
with S.quiet():
    m = SkeletonModel(K=-1, is_blender=True, skinning=False, hyper_dim=8, joints=joints, parent_indices=parents, pred_opacity=False,
                      pred_color=False, use_hash=False, hash_time=False, d_rot_as_res=True, local_frame=False,
                      progressive_brand_time=False, with_arap_loss=True, max_d_scale=-1, enable_densify_prune=False,
                      is_scene_static=False, use_skinning_weight_mlp=True, use_template_offsets=True)
    assert type(m.deform).__module__ == "riggs_amd.skeleton"
    m.train_setting(opt)
assert [g["name"] for g in m.optimizer.param_groups] == ["nodes", "pose", "skinning_mlp", "detail_net"]
gs = m.deform.as_gaussians
assert type(gs).__module__ == "scene.gaussian_model" and gs.get_xyz.shape == (8, 3) and gs.optimizer is not None
m.update_learning_rate(10, True); assert m.optimizer.param_groups[0]["lr"] == 5e-4
m.update_learning_rate(10, False); m.update(3)
m.save_weights(%r, 7)
assert m.load_weights(%r, -1) and m.load_weights(%r, 7) and not m.load_weights(%r, 8)
assert any(k.startswith("gs_") for k in m.deform.state_dict())
print("ok")
''' % (os.path.join(HERE, "golden"), os.path.dirname(HERE), str(tmp_path), str(tmp_path), str(tmp_path), str(tmp_path))
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and r.stdout.strip().endswith("ok"), r.stdout + r.stderr
