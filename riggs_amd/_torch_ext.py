"""The PyTorch extension front-end (csrc_torch/riggs_torch.cpp -> lib/libriggs_torch.so): the two per-iteration autograd nodes of an
unmodified train_rig.py — ``SkeletonWarp.forward`` and ``render()``'s default branch — with node, marshalling and allocations in C++
(``torch.ops.riggs.pose_deform`` / ``glue_raster``), calling the same C ABI as the ctypes nodes.  OPTIONAL: when the library is not
built, or a call is not the plain eager training frame (a flat gradient bucket is registered, a hipGraph capture is recording, the
ordered backward / ``pipe.debug`` / sparse gradient rows are on, no arena history yet), the ctypes nodes run — same kernels, same
results (tests/test_gpu_torch_ext.py).  ``enable(False)`` switches it off for the process."""
from __future__ import annotations

import os

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
PATH = os.path.join(_HERE, "lib", "libriggs_torch.so")
_state = {"loaded": None, "enabled": True}


def enable(on: bool = True):
    _state["enabled"] = bool(on)


def available() -> bool:
    """The extension is built, loads, and speaks the library's ABI version."""
    if _state["loaded"] is None:
        ok = False
        if os.path.exists(PATH):
            try:
                from . import _lib as L
                L.lib()  # (libriggs_hip.so first: the extension links against it)
                torch.ops.load_library(PATH)
                ok = int(torch.ops.riggs.abi_version()) == int(L.lib().riggs_version())
            except (OSError, RuntimeError, AttributeError):
                ok = False
        _state["loaded"] = ok
    return bool(_state["loaded"])


def active() -> bool:
    """Should an eager call take the extension's node?  (Never while a stream capture records: the captured frames' static
    gradient buffers and sparse rows are the ctypes nodes' business.)"""
    return _state["enabled"] and available() and not torch.cuda.is_current_stream_capturing()
