"""CPU: the C-ABI library loads and exports every symbol include/riggs_hip.h declares; host-side
logic that needs no GPU."""
import os
import re

import pytest
import torch

from riggs_amd import _lib


def _header_functions():
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    txt = open(os.path.join(root, "include", "riggs_hip.h")).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(riggs_[a-z0-9_]+)\s*\(", txt)))


def test_library_exports_every_declared_symbol():
    declared = _header_functions()
    assert len(declared) >= 18
    L = _lib.lib()  # loads without a GPU; argtypes set for every symbol
    for name in declared:
        assert hasattr(L, name), "symbol %s declared in include/riggs_hip.h but not exported" % name
    assert sorted(_lib.exported_symbols()) == declared
    assert L.riggs_version() >= 100


def test_product_path_refuses_cpu_tensors():
    from riggs_amd.rasterizer import GaussianRasterizer, GaussianRasterizationSettings
    st = GaussianRasterizationSettings(16, 16, 0.5, 0.5, torch.zeros(3), 1.0, torch.eye(4), torch.eye(4), 3,
                                       torch.zeros(3), False, False)
    r = GaussianRasterizer(st)
    with pytest.raises(_lib.RiggsHipError, match="CUDA"):
        r(means3D=torch.zeros(4, 3), means2D=torch.zeros(4, 3), opacities=torch.zeros(4, 1), shs=torch.zeros(4, 16, 3),
          scales=torch.ones(4, 3), rotations=torch.ones(4, 4))
    with pytest.raises(Exception, match="SHs or precomputed colors"):
        r(means3D=torch.zeros(4, 3), means2D=None, opacities=torch.zeros(4, 1), scales=torch.ones(4, 3),
          rotations=torch.ones(4, 4))


def test_missing_library_fails_loudly(monkeypatch):
    monkeypatch.setattr(_lib, "_lib", None)
    monkeypatch.setattr(_lib, "SO_PATH", "/nonexistent/libriggs_hip.so")
    with pytest.raises(_lib.RiggsHipError, match="no fallback"):
        _lib.lib()


def test_skeleton_host_logic_cpu():
    from riggs_amd.skeleton import SkeletonWarp
    joints = torch.randn(5, 3)
    sw = SkeletonWarp(joints=joints, parent_indices=torch.tensor([-1, 0, 1, 1, 3]), K=-1, hyper_dim=8,
                      use_skinning_weight_mlp=False, use_template_offsets=False)
    assert sw.nodes.shape == (5, 11) and not sw.nodes.requires_grad
    assert sw.expand_time(torch.tensor([0.5])).shape == (5, 1)
    info = sw.get_pose_info(sw.expand_time(torch.tensor([0.5])))
    assert info["local_rotation"].shape == (5, 4) and info["global_trans"].shape == (3,)
    names = [g["name"] for g in sw.trainable_parameters()]
    assert names == ["nodes", "pose"]
    keys = set(sw.state_dict().keys())
    assert {"nodes", "_node_radius", "control_nodes", "pose_net.rotation_predictor.weight"} <= keys
    with pytest.raises(ValueError):
        bad = SkeletonWarp(joints=joints, parent_indices=torch.tensor([-1, 2, 1, 1, 3]), K=-1,
                           use_skinning_weight_mlp=False, use_template_offsets=False)
        bad._parents_dev(torch.device("cpu"))


def test_argument_validation_of_the_next_row_entries_needs_no_gpu():
    """Bad sizes / NULL buffers are rejected with a message before any HIP call (no compute, no GPU)."""
    L = _lib.lib()
    N = None
    # control nodes: K > 8, hyper > 11, K > M
    for (n, m, k, hyper) in ((10, 32, 9, 0), (10, 32, 3, 12), (10, 2, 3, 0)):
        rc = L.riggs_cnode_forward(n, m, k, hyper, hyper, 3 + hyper, 0, *([N] * 17))
        assert rc != 0 and L.riggs_last_error()
    assert L.riggs_cnode_forward(10, 32, 3, 0, 0, 3, 1, *([N] * 17)) != 0            # local_frame without local_rotation
    assert b"local_rotation" in L.riggs_last_error()
    assert L.riggs_cnode_backward_workspace_floats(1000, 64, 3, 8) >= 33 * 3000
    assert 1 <= L.riggs_cnode_backward_blocks(10, 64, 8) <= L.riggs_cnode_backward_blocks(10 ** 6, 64, 8) <= 512
    # skeleton projection loss: one joint, empty sample / pixel sets, too many samples per bone
    for (j, s, m) in ((1, 4, 4), (5, 0, 4), (5, 4, 0), (5, 4096, 4)):
        rc = L.riggs_skeleton_projection_forward(j, s, m, N, N, N, N, 1.0, 1.0, 0.0, 0.0, N, N, N, N, N, N)
        assert rc != 0 and L.riggs_last_error()
    assert L.riggs_skeleton_projection_state_floats(24, 41, 1500) == 2 * 41 * 23 + 2 * 1500 + 6 * 23


def test_integration_stub_matches_the_header_struct():
    """INTEGRATION.md's ctypes stub of riggs_raster_cfg, the struct in include/riggs_hip.h and riggs_amd._lib.RasterCfg
    list the same fields in the same order (a binding written from a stale stub makes the library read past the struct)."""
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    doc = open(os.path.join(root, "INTEGRATION.md")).read()
    stub = doc[doc.index("class RasterCfg(C.Structure)"):]
    stub = stub[:stub.index("def rasterize")]
    stub_fields = re.findall(r'\("([a-z_0-9A-Z]+)",\s*C\.(c_[a-z0-9_]+)\)', stub)
    import ctypes as C
    canon = lambda name: getattr(C, name).__name__            # (c_int32 is an alias of c_int on this ABI)
    mine = [(n, t.__name__) for n, t in _lib.RasterCfg._fields_]
    assert [(n, canon(t)) for n, t in stub_fields] == mine
    hdr = open(os.path.join(root, "include", "riggs_hip.h")).read()
    body = hdr[hdr.index("typedef struct riggs_raster_cfg"):hdr.index("} riggs_raster_cfg;")]
    body = re.sub(r"/\*.*?\*/", "", body, flags=re.S)
    hdr_fields = re.findall(r"\b(?:const\s+)?(int32_t|float)\s*(\*?)\s*([a-zA-Z_0-9]+)\s*;", body)
    ctype = {("int32_t", ""): "c_int32", ("float", ""): "c_float", ("float", "*"): "c_void_p"}
    assert [(n, canon(ctype[(t, p)])) for t, p, n in hdr_fields] == mine
    # the stub's constructor call passes one value per field
    call = stub_fields and doc[doc.index("cfg = RasterCfg("):]
    call = call[:call.index("\n    u8 =")]
    depth, n_args = 0, 1
    for ch in call[call.index("(") + 1:call.rindex(")")]:
        depth += ch in "([" 
        depth -= ch in ")]"
        n_args += (ch == "," and depth == 0)
    assert n_args == len(mine)


def test_pose_mlp_sync_buffer_is_as_large_as_the_library_wants():
    """PoseMLP restates riggs_pose_mlp_sync_bytes for its persistent hand-off state (built on CPU-only hosts too); a buffer
    that is too small silently falls back to a per-launch private state (no sticky status word)."""
    from riggs_amd import _lib as L
    from riggs_amd.skeleton import PoseMLP
    for joints, depth, width in ((24, 8, 256), (64, 8, 256), (6, 3, 32), (12, 12, 128)):
        net = PoseMLP(1, joints * 4, hidden_dimensions=width, depth=depth)
        assert net._hip_sync.numel() * 4 >= L.lib().riggs_pose_mlp_sync_bytes(depth, width)
        assert int(L.lib().riggs_pose_mlp_status_word(depth, width)) + 1 < net._hip_sync.numel()


def test_torch_extension_loads_and_registers_its_ops():
    """lib/libriggs_torch.so (riggs_amd/csrc_torch/riggs_torch.cpp; built by __graft_entry__.build()): loads next to libriggs_hip.so
    without a GPU, speaks the library's ABI version and registers the two nodes' ops (no compute call here)."""
    import torch
    from riggs_amd import _lib as L
    from riggs_amd import _torch_ext as TX
    from riggs_amd import build as B
    B.build_torch()
    assert TX.available()
    assert int(torch.ops.riggs.abi_version()) == int(L.lib().riggs_version())
    for name in ("pose_deform", "glue_raster"):
        assert hasattr(torch.ops.riggs, name)
    sch = str(torch.ops.riggs.glue_raster.default._schema)
    assert "Tensor? d_xyz" in sch or "Tensor?" in sch
