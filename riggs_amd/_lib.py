"""ctypes binding of libriggs_hip.so (include/riggs_hip.h).

The product path has NO fallback: if the HIP library is missing or fails to load,
every op raises.  PyTorch is used only for device memory and streams.
"""
from __future__ import annotations

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
SO_PATH = os.path.join(_HERE, "lib", "libriggs_hip.so")

# enum mirrors (include/riggs_hip.h)
GEOM_XYD, GEOM_CONIC_O, GEOM_RGB, GEOM_COV3D, GEOM_CLAMPED, GEOM_TILES, GEOM_RECT, GEOM_DEPTH_ORDER, \
    GEOM_NFIELDS = range(9)
IMG_FINAL_T, IMG_N_CONTRIB, IMG_RANGES, IMG_FWD_CTR, IMG_NFIELDS = range(5)
BIN_POINT_LIST, BIN_TILE_KEYS, BIN_WALK_HIST, BIN_NFIELDS = range(4)


class RasterCfg(C.Structure):
    _fields_ = [
        ("num_points", C.c_int32), ("sh_degree", C.c_int32), ("sh_coeffs", C.c_int32),
        ("image_height", C.c_int32), ("image_width", C.c_int32),
        ("tanfovx", C.c_float), ("tanfovy", C.c_float), ("scale_modifier", C.c_float),
        ("bg", C.c_void_p), ("viewmatrix", C.c_void_p), ("projmatrix", C.c_void_p), ("campos", C.c_void_p),
        ("debug", C.c_int32), ("glue", C.c_int32), ("isotropic", C.c_int32), ("deterministic", C.c_int32),
        ("sparse_zero", C.c_int32), ("tight_lists", C.c_int32),
    ]


class Frame(C.Structure):
    """struct riggs_frame (include/riggs_hip.h), field for field."""
    _fields_ = [
        ("depth", C.c_int32), ("width", C.c_int32), ("multires", C.c_int32), ("skip", C.c_int32), ("n_rot", C.c_int32),
        ("weights", C.c_void_p), ("biases", C.c_void_p),
        ("W_rot", C.c_void_p), ("b_rot", C.c_void_p), ("W_tr", C.c_void_p), ("b_tr", C.c_void_p), ("t", C.c_void_p), ("rot_bias4", C.c_void_p),
        ("sync_state", C.c_void_p), ("acts", C.c_void_p), ("local_rot", C.c_void_p), ("global_trans", C.c_void_p),
        ("num_joints", C.c_int32), ("K", C.c_int32),
        ("joints", C.c_void_p), ("parents", C.c_void_p), ("node_radius_log", C.c_void_p), ("motion_mask", C.c_void_p), ("weight_mod", C.c_void_p),
        ("transforms", C.c_void_p), ("node_rot", C.c_void_p), ("d_nodes", C.c_void_p), ("d_xyz", C.c_void_p), ("d_rotation", C.c_void_p),
        ("cfg", RasterCfg),
        ("xyz", C.c_void_p), ("features_dc", C.c_void_p), ("features_rest", C.c_void_p), ("opacity", C.c_void_p), ("scaling", C.c_void_p),
        ("rotation", C.c_void_p), ("d_scaling", C.c_void_p),
        ("geom", C.c_void_p), ("radii", C.c_void_p), ("counters", C.c_void_p),
        ("binning", C.c_void_p), ("instance_capacity", C.c_int64), ("binning_bytes", C.c_size_t), ("image_state", C.c_void_p),
        ("out_color", C.c_void_p), ("out_depth", C.c_void_p), ("out_alpha", C.c_void_p),
    ]


class FrameGrads(C.Structure):
    """struct riggs_frame_grads."""
    _fields_ = [(n, C.c_void_p) for n in (
        "dL_dcolor", "dL_ddepth", "dL_dalpha", "raster_workspace", "dL_dxyz", "dL_dmeans2D", "dL_dfeatures_dc", "dL_dfeatures_rest",
        "dL_dopacity", "dL_dscaling", "dL_drotation", "dL_dd_scaling", "dL_dtransforms", "dL_dnode_radius_log",
        "dL_dglobal_trans_skinning", "dL_dmotion_mask", "dL_dweight_mod", "lbs_workspace", "dL_dd_nodes", "g_local_rot", "g_global_trans",
        "dL_dlocal_rot", "dL_dglobal_trans", "pose_workspace", "pose_flat_grads")]


class MlpEpilogue(C.Structure):
    """struct riggs_mlp_epilogue (include/riggs_hip.h): the output epilogue of riggs_mlp_forward."""
    _fields_ = [("sigmoid", C.c_int32), ("reserved", C.c_int32), ("res_base", C.c_void_p), ("res_mask", C.c_void_p),
                ("res_out", C.c_void_p)]


class GateStruct(C.Structure):
    """struct riggs_gate."""
    _fields_ = [("n", C.c_int32), ("reserved", C.c_int32), ("word", C.c_void_p * 4), ("mask", C.c_uint32 * 4)]


class Watch:
    """A non-blocking look at one 32-bit device word, for callers that must not synchronise (an eagerly issued training
    iteration is bound by the host: a blocking read per iteration would expose the whole queue's latency).  Every ``period``-th
    ``poll`` enqueues an asynchronous copy of the word into pinned host memory behind the work issued so far; a poll returns the
    most recent value whose copy has landed (0 until then).  Never call it while the current stream is being captured."""

    def __init__(self, period: int = 16):
        self.period, self.n, self.event, self.host, self.value = int(period), 0, None, None, 0

    def poll(self, tensor, index: int) -> int:
        import torch
        if self.event is not None and self.event.query():
            self.value, self.event = int(self.host[0]), None
        self.n += 1
        if self.event is None and self.n % self.period == 0:
            if self.host is None:
                self.host = torch.zeros(1, dtype=tensor.dtype).pin_memory()
            self.host.copy_(tensor[index:index + 1], non_blocking=True)
            self.event = torch.cuda.Event()
            self.event.record()
        return self.value


class FrameGate:
    """A frame's "valid" gate (include/riggs_hip.h: riggs_gate): up to four (device word, mask) pairs — the sticky status word
    of the one-launch PoseMLP kernels, the rasterizer's overflow / sort-barrier flags, the gradient-row exchange's status —
    that the consumers of the frame's gradients (the fused Adam, the exchange's pack) read ON THE DEVICE: a frame that went
    wrong is a skipped step, also inside a hipGraph where the host cannot look first.  ``sources``: callables returning
    ``(int32 / uint32 device tensor, word index, mask)`` or None, resolved at every launch (the rasterizer's counters are a
    new tensor per frame until a capture pins them).  ``skipped`` counts the steps the gate turned into no-ops."""

    def __init__(self, sources=(), device=None):
        import torch
        self.sources = list(sources)
        self.skipped = torch.zeros(1, dtype=torch.int32, device=device or "cuda")
        self._keep = []

    def struct(self):
        g = GateStruct()
        keep, n = [], 0
        for src in self.sources:
            got = src()
            if got is None:
                continue
            t, index, mask = got
            if not (t.is_cuda and t.element_size() == 4 and t.is_contiguous() and 0 <= index < t.numel()):
                raise RiggsHipError("a gate word must be an element of a contiguous 32-bit device tensor")
            if n >= 4:
                raise RiggsHipError("at most four gate words")
            g.word[n] = t.data_ptr() + 4 * int(index)
            g.mask[n] = int(mask) & 0xFFFFFFFF
            keep.append(t)
            n += 1
        g.n = n
        self._keep = keep  # the words outlive the launches that were handed their addresses
        return g

    def read_skipped(self, clear=True) -> int:
        n = int(self.skipped.item())
        if clear and n:
            self.skipped.zero_()
        return n


_lib = None

_P = C.c_void_p
_SIGS = {
    "riggs_version": (C.c_int, []),
    "riggs_last_error": (C.c_char_p, []),
    "riggs_raster_geom_bytes": (C.c_size_t, [C.c_int32]),
    "riggs_raster_image_bytes": (C.c_size_t, [C.c_int32, C.c_int32]),
    "riggs_raster_binning_bytes": (C.c_size_t, [C.c_int64, C.c_int32, C.c_int32, C.c_int32]),
    "riggs_raster_backward_workspace_bytes": (C.c_size_t, [C.c_int32]),
    "riggs_raster_backward_workspace_bytes_ordered": (C.c_size_t, [C.c_int32, C.c_int64]),
    "riggs_raster_backward_workspace_rows": (C.c_int, [C.c_int32, C.POINTER(C.c_size_t), C.POINTER(C.c_size_t)]),
    "riggs_raster_geom_layout": (C.c_int, [C.c_int32, C.POINTER(C.c_size_t)]),
    "riggs_raster_image_layout": (C.c_int, [C.c_int32, C.c_int32, C.POINTER(C.c_size_t)]),
    "riggs_raster_binning_layout": (C.c_int, [C.c_int64, C.c_int32, C.c_int32, C.c_int32, C.POINTER(C.c_size_t)]),
    "riggs_raster_preprocess": (C.c_int, [C.POINTER(RasterCfg)] + [_P] * 11 + [_P, _P, _P, _P]),
    "riggs_set_option": (C.c_int, [C.c_char_p, C.c_int32]),
    "riggs_get_option": (C.c_int, [C.c_char_p, C.POINTER(C.c_int32)]),
    "riggs_raster_binning_reset_history": (C.c_int, [_P, C.c_int64, C.c_int32, C.c_int32, C.c_int32, _P]),
    "riggs_raster_render": (C.c_int, [C.POINTER(RasterCfg), _P, _P, C.c_int64, C.c_size_t, _P, _P, _P, _P, _P, _P]),
    "riggs_raster_backward": (C.c_int, [C.POINTER(RasterCfg)] + [_P] * 11 + [_P, _P, _P, C.c_int64, _P, _P] + [_P] * 3
                              + [_P] + [_P] * 10 + [_P]),
    "riggs_fk_forward": (C.c_int, [C.c_int32] + [_P] * 8),
    "riggs_fk_backward": (C.c_int, [C.c_int32] + [_P] * 8),
    "riggs_lbs_forward": (C.c_int, [C.c_int32, C.c_int32, C.c_int32] + [_P] * 15),
    "riggs_lbs_backward": (C.c_int, [C.c_int32, C.c_int32, C.c_int32] + [_P] * 18),
    "riggs_lbs_forward_fk": (C.c_int, [C.c_int32, C.c_int32, C.c_int32] + [_P] * 15),
    "riggs_lbs_bone_table_bytes": (C.c_size_t, []),
    "riggs_lbs_backward_workspace_bytes": (C.c_size_t, [C.c_int32, C.c_int32]),
    "riggs_pose_mlp_acts_floats": (C.c_size_t, [C.c_int32, C.c_int32, C.c_int32]),
    "riggs_pose_mlp_backward_workspace_floats": (C.c_size_t, [C.c_int32, C.c_int32, C.c_int32]),
    "riggs_pose_mlp_forward": (C.c_int, [C.c_int32] * 5 + [_P] * 13),
    "riggs_pose_mlp_sync_bytes": (C.c_size_t, [C.c_int32] * 2),
    "riggs_pose_mlp_set_placement": (C.c_int, [C.c_int32]),
    "riggs_pose_mlp_set_trace": (C.c_int, [_P]),
    "riggs_pose_mlp_backward": (C.c_int, [C.c_int32] * 5 + [_P] * 13),
    "riggs_pose_mlp_backward_fk": (C.c_int, [C.c_int32] * 5 + [_P] * 7 + [C.c_int32] + [_P] * 16),
    "riggs_pose_mlp_status_word": (C.c_size_t, [C.c_int32] * 2),
    "riggs_grad_rows_row_floats": (C.c_int32, [C.c_int32, _P]),
    "riggs_grad_rows_segment_bytes": (C.c_size_t, [C.c_int32, C.c_int32, C.c_int32]),
    "riggs_grad_rows_pack": (C.c_int, [C.c_int32, _P, C.c_int32, _P, _P, C.c_float, C.c_int32, _P, _P]),
    "riggs_grad_rows_unpack": (C.c_int, [C.c_int32, C.c_int32, C.c_int32, _P, C.c_int32, _P, _P, _P, _P, _P]),
    "riggs_adam_step": (C.c_int, [C.c_int32, _P, _P, _P, _P, _P, _P, _P, C.c_double, C.c_double, C.c_double, _P]),
    "riggs_adam_step_guarded": (C.c_int, [C.c_int32, _P, _P, _P, _P, _P, _P, _P, C.c_double, C.c_double, C.c_double, _P, _P]),
    "riggs_adam_step_capturable": (C.c_int, [C.c_int32, _P, _P, _P, _P, _P, _P, _P, _P, C.c_double, C.c_double, C.c_double, _P]),
    "riggs_adam_steps_advance_coef": (C.c_int, [C.c_int32, _P, _P, _P, C.c_double, C.c_double, _P, _P]),
    "riggs_adam_step_gated_coef": (C.c_int, [C.c_int32, _P, _P, _P, _P, _P, _P, _P, _P, C.c_double, C.c_double, C.c_double, _P, _P, _P]),
    "riggs_adam_step_gated": (C.c_int, [C.c_int32, _P, _P, _P, _P, _P, _P, _P, _P, C.c_double, C.c_double, C.c_double,
                                        C.POINTER(GateStruct), _P, C.c_int32, _P]),
    "riggs_gate_flag": (C.c_int, [C.POINTER(GateStruct), _P, _P]),
    "riggs_adam_steps_advance_gated": (C.c_int, [C.c_int32, _P, C.POINTER(GateStruct), _P, _P]),
    "riggs_grad_rows_pack_gated": (C.c_int, [C.c_int32, _P, C.c_int32, _P, _P, C.c_float, C.c_int32, _P, C.POINTER(GateStruct), _P]),
    "riggs_debug_pin_cus": (C.c_int, [C.c_int32, _P, C.c_uint32, _P, _P]),
    "riggs_densify_stats": (C.c_int, [C.c_int32] + [_P] * 7),
    "riggs_l1_ssim_state_floats": (C.c_size_t, [C.c_int32] * 3),
    "riggs_l1_ssim_forward": (C.c_int, [C.c_int32] * 3 + [_P, _P, C.c_float, _P, _P, _P]),
    "riggs_l1_ssim_backward": (C.c_int, [C.c_int32] * 3 + [_P, _P, _P, C.c_float, _P, _P, _P, _P, _P]),
    "riggs_cnode_backward_blocks": (C.c_int, [C.c_int32] * 3),
    "riggs_cnode_backward_workspace_floats": (C.c_size_t, [C.c_int32] * 4),
    "riggs_cnode_forward": (C.c_int, [C.c_int32] * 7 + [_P] * 17),
    "riggs_cnode_backward": (C.c_int, [C.c_int32] * 7 + [_P] * 26),
    "riggs_skeleton_projection_state_floats": (C.c_size_t, [C.c_int32] * 3),
    "riggs_skeleton_projection_forward": (C.c_int, [C.c_int32] * 3 + [_P] * 4 + [C.c_float] * 4 + [_P] * 6),
    "riggs_skeleton_projection_backward": (C.c_int, [C.c_int32] * 3 + [_P] * 4 + [C.c_float] * 4 + [_P] * 8),
    "riggs_raster_set_trace": (C.c_int, [_P]),
    "riggs_raster_set_trace_items": (C.c_int, [C.c_uint64]),
    "riggs_mlp_forward": (C.c_int, [C.c_int32] * 5 + [_P] * 10 + [C.c_int32, _P]),
    "riggs_mlp_backward": (C.c_int, [C.c_int32] * 4 + [_P] * 8 + [C.c_int32, _P]),
    "riggs_mlp_live_rows_workspace_bytes": (C.c_size_t, [C.c_int32]),
    "riggs_mlp_live_rows": (C.c_int, [C.c_int32] * 3 + [_P] * 10),
    "riggs_mlp_rows_per_workgroup": (C.c_int32, []),
    "riggs_mlp_grad_scale": (C.c_int, [C.c_int64, _P, _P, _P, _P]),
    "riggs_mlp_l2_grad_scale": (C.c_int, [C.c_int64] + [_P] * 9),
    "riggs_mlp_cotangent": (C.c_int, [C.c_int32] * 2 + [_P] * 12),
    "riggs_mlp_wgrad_workspace_bytes": (C.c_size_t, [C.c_int32] * 4),
    "riggs_mlp_wgrad": (C.c_int, [C.c_int32] * 5 + [_P] * 6 + [C.c_size_t] + [_P] * 5 + [C.c_int32, _P]),
    "riggs_mlp_embed": (C.c_int, [C.c_int32] * 3 + [_P] * 3 + [C.c_int32, _P]),
    "riggs_mlp_pack": (C.c_int, [C.c_int32] * 4 + [_P] * 6 + [C.c_int32, _P]),
    "riggs_mlp_pack_tail": (C.c_int, [C.c_int32] * 5 + [_P] * 6 + [C.c_int32, _P]),
    "riggs_mlp_tail_bias": (C.c_int, [C.c_int32] * 2 + [_P] * 7),
    "riggs_mlp_wgrad_tail": (C.c_int, [C.c_int32] * 3 + [_P] + [C.c_int32] * 3 + [_P] * 6 + [C.c_size_t] + [_P] * 5 + [C.c_int32, _P]),
    "riggs_mlp_layout_probe": (C.c_int, [_P, _P]),
    "riggs_densify_select": (C.c_int, [C.c_int32, C.c_int32, _P, _P, _P, _P] + [C.c_float] * 5 + [_P, _P]),
    "riggs_compact_workspace_bytes": (C.c_size_t, [C.c_int32]),
    "riggs_compact_indices": (C.c_int, [C.c_int32, _P, _P, _P, _P, _P]),
    "riggs_rows_gather": (C.c_int, [C.c_int32, _P, C.c_int32, _P, _P, _P, _P, _P]),
    "riggs_split_children": (C.c_int, [C.c_int32, C.c_int32, C.c_int32, _P, _P, _P, _P, _P, C.c_float, _P, _P, _P]),
    "riggs_frame_forward": (C.c_int, [C.POINTER(Frame), _P]),
    "riggs_frame_backward": (C.c_int, [C.POINTER(Frame), C.POINTER(FrameGrads), _P]),
    "riggs_dqb_forward": (C.c_int, [C.c_int32] * 5 + [_P] * 6),
    "riggs_dqb_backward_workspace_floats": (C.c_size_t, [C.c_int32] * 3),
    "riggs_dqb_backward": (C.c_int, [C.c_int32] * 5 + [_P] * 10),
    "riggs_prof_count": (C.c_int, []),
    "riggs_prof_name": (C.c_char_p, [C.c_int32]),
    "riggs_prof_enable": (C.c_int, [C.c_uint32]),
    "riggs_prof_reset": (C.c_int, []),
    "riggs_prof_read": (C.c_int, [C.c_int32, C.POINTER(C.c_float), C.POINTER(C.c_int32)]),
    "riggs_knn_workspace_bytes": (C.c_size_t, [C.c_int32]),
    "riggs_dist2_knn3": (C.c_int, [C.c_int32, _P, _P, _P, _P]),
    "riggs_dist2_knn3_bruteforce": (C.c_int, [C.c_int32, _P, _P, _P]),
}


class RiggsHipError(RuntimeError):
    pass


def lib():
    """Load libriggs_hip.so; raise loudly when it is absent (no CPU/eager fallback exists)."""
    global _lib
    if _lib is None:
        if not os.path.exists(SO_PATH):
            raise RiggsHipError(
                "libriggs_hip.so not found at %s — build it with `python -m riggs_amd.build` "
                "(or __graft_entry__.build()).  There is no fallback path." % SO_PATH)
        # (torch's HIP runtime first: a process whose first HIP activity is this library's load — its kernels register with
        # the runtime from static initialisers — and which initialises torch.cuda afterwards ends with the library's launches
        # failing "no ROCm-capable device is detected"; the other order is the one every caller that allocates a tensor
        # before its first call takes anyway)
        import torch
        if torch.cuda.is_available():
            torch.cuda.init()
        L = C.CDLL(SO_PATH)
        for name, (res, args) in _SIGS.items():
            fn = getattr(L, name)  # AttributeError if a declared symbol is not exported
            fn.restype = res
            fn.argtypes = args
        _lib = L
    return _lib


def set_option(name: str, value: int):
    """riggs_set_option (include/riggs_hip.h): "fwd_wide_tiles", "fwd_wide_min", "bin_grouped", "cnode_bwd_atomics",
    "color_side_jobs", "preprocess_bwd_lean", "pose_mlp_layered", "fwd_hist_view_tol", "lbs_scalar"."""
    check(lib().riggs_set_option(name.encode(), int(value)), "riggs_set_option")
    OPTIONS_SET[name] = int(value)


OPTIONS_SET = {}  # what this process set through set_option (host-side decisions that follow an option read it here: no C call per frame)


def get_option(name: str) -> int:
    v = C.c_int32()
    check(lib().riggs_get_option(name.encode(), C.byref(v)), "riggs_get_option")
    return int(v.value)


def exported_symbols():
    return sorted(_SIGS.keys())


def check(rc: int, what: str):
    if rc != 0:
        msg = lib().riggs_last_error().decode(errors="replace")
        raise RiggsHipError("%s failed (rc=%d): %s" % (what, rc, msg))


def ptr(t):
    """Device pointer of a tensor (or None -> NULL)."""
    return None if t is None else t.data_ptr()


def stream_ptr():
    """The current HIP stream of the current device as an integer handle (torch._C's raw accessor: ~0.3 us instead of the
    ~15 us of building a torch.cuda.Stream object — three of those per eager frame were 7 % of its host time)."""
    import torch
    raw = getattr(torch._C, "_cuda_getCurrentRawStream", None)
    if raw is not None:
        return raw(torch._C._cuda_getDevice())
    return torch.cuda.current_stream().cuda_stream


_F32 = None


def require_cuda_f32(name, t, shape=None):
    global _F32
    if t is None:
        return None
    if _F32 is None:
        import torch
        _F32 = torch.float32
    if not t.is_cuda:
        raise RiggsHipError("%s must be a CUDA(HIP) tensor — the product path is GPU-only" % name)
    if t.dtype is not _F32:
        raise RiggsHipError("%s must be float32, got %s" % (name, t.dtype))
    if shape is not None:
        ts = t.shape
        ok = len(shape) == len(ts)
        if ok:
            for s, d in zip(shape, ts):
                if s is not None and s != d:
                    ok = False
                    break
        if not ok:
            raise RiggsHipError("%s has shape %s, expected %s" % (name, tuple(ts), shape))
    return t if t.is_contiguous() else t.contiguous()
