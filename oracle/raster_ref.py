"""ORACLE (test infrastructure, NOT product code): ctypes front-end of
oracle/raster_ref.c (CPU float32 restatement of the 3DGS tile rasterizer).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "_build", "libraster_ref.so")
_lib = None


def build(force: bool = False) -> str:
    src = os.path.join(_HERE, "raster_ref.c")
    if force or not os.path.exists(_SO) or os.path.getmtime(_SO) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "-s"])
    return _SO


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_SO):
            build()
        _lib = C.CDLL(_SO)
        _lib.rr_forward.restype = C.c_long
    return _lib


def _p(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


def _f32(a):
    return None if a is None else np.ascontiguousarray(a, dtype=np.float32)


def set_threads(n: int):
    lib().rr_set_threads(int(n))


class Saved:
    pass


def forward(means3D, opacities, view, proj, campos, tanfovx, tanfovy, H, W, bg, shs=None, colors_precomp=None,
            scales=None, rotations=None, cov3D_precomp=None, sh_degree=3, scale_modifier=1.0):
    """Returns dict(color(3,H,W), depth(H,W), alpha(H,W), radii(N)) and a Saved state."""
    L = lib()
    means3D = _f32(means3D)
    N = means3D.shape[0]
    s = Saved()
    s.N, s.H, s.W, s.deg = N, H, W, sh_degree
    s.means3D, s.opac = means3D, _f32(np.reshape(opacities, (-1,)))
    s.shs = _f32(shs)
    s.M = 0 if shs is None else s.shs.shape[1]
    s.colors_precomp, s.scales, s.rots, s.cov3D_precomp = _f32(colors_precomp), _f32(scales), _f32(rotations), _f32(cov3D_precomp)
    s.view, s.proj, s.campos, s.bg = _f32(np.reshape(view, -1)), _f32(np.reshape(proj, -1)), _f32(campos), _f32(bg)
    s.tanfovx, s.tanfovy, s.mod = float(tanfovx), float(tanfovy), float(scale_modifier)
    T = ((W + 15) // 16) * ((H + 15) // 16)
    out_color = np.zeros((3, H, W), np.float32)
    out_depth = np.zeros((H, W), np.float32)
    out_alpha = np.zeros((H, W), np.float32)
    s.radii = np.zeros(N, np.int32)
    s.depths = np.zeros(N, np.float32)
    s.xy = np.zeros((N, 2), np.float32)
    s.cov3D = np.zeros((N, 6), np.float32)
    s.conic_o = np.zeros((N, 4), np.float32)
    s.rgb = np.zeros((N, 3), np.float32)
    s.clamped = np.zeros((N, 3), np.uint8)
    s.tiles = np.zeros(N, np.uint32)
    s.ranges = np.zeros((T, 2), np.uint32)
    s.final_T = np.zeros((H, W), np.float32)
    s.n_contrib = np.zeros((H, W), np.uint32)
    cap = max(1024, 8 * N)
    while True:
        s.keys = np.zeros(cap, np.uint64)
        s.point_list = np.zeros(cap, np.uint32)
        R = L.rr_forward(
            C.c_int(N), C.c_int(sh_degree), C.c_int(s.M), C.c_int(W), C.c_int(H), _p(s.bg), _p(means3D), _p(s.shs),
            _p(s.colors_precomp), _p(s.opac), _p(s.scales), C.c_float(s.mod), _p(s.rots), _p(s.cov3D_precomp),
            _p(s.view), _p(s.proj), _p(s.campos), C.c_float(s.tanfovx), C.c_float(s.tanfovy),
            _p(out_color), _p(out_depth), _p(out_alpha), _p(s.radii), _p(s.depths), _p(s.xy), _p(s.cov3D),
            _p(s.conic_o), _p(s.rgb), _p(s.clamped), _p(s.tiles), _p(s.keys), _p(s.point_list), C.c_long(cap),
            _p(s.ranges), _p(s.final_T), _p(s.n_contrib))
        if R >= 0:
            break
        cap = -R
    s.R = int(R)
    s.keys = s.keys[: s.R]
    s.point_list = s.point_list[: s.R]
    return {"color": out_color, "depth": out_depth, "alpha": out_alpha, "radii": s.radii}, s


def backward(s: Saved, dL_dcolor, dL_ddepth=None, dL_dalpha=None):
    L = lib()
    N, M = s.N, max(s.M, 1)
    g = {
        "means3D": np.zeros((N, 3), np.float32), "means2D": np.zeros((N, 3), np.float32),
        "shs": np.zeros((N, M, 3), np.float32), "colors_precomp": np.zeros((N, 3), np.float32),
        "opacities": np.zeros((N, 1), np.float32), "scales": np.zeros((N, 3), np.float32),
        "rotations": np.zeros((N, 4), np.float32), "cov3D_precomp": np.zeros((N, 6), np.float32),
    }
    dc, dd, da = _f32(dL_dcolor), _f32(dL_ddepth), _f32(dL_dalpha)
    pl = np.ascontiguousarray(s.point_list)
    L.rr_backward(
        C.c_int(N), C.c_int(s.deg), C.c_int(s.M), C.c_int(s.W), C.c_int(s.H), _p(s.bg), _p(s.means3D), _p(s.shs),
        _p(s.colors_precomp), _p(s.opac), _p(s.scales), C.c_float(s.mod), _p(s.rots), _p(s.cov3D_precomp),
        _p(s.view), _p(s.proj), _p(s.campos), C.c_float(s.tanfovx), C.c_float(s.tanfovy),
        _p(s.radii), _p(s.depths), _p(s.xy), _p(s.cov3D), _p(s.conic_o), _p(s.rgb), _p(s.clamped), _p(pl),
        C.c_long(s.R), _p(s.ranges), _p(s.final_T), _p(s.n_contrib), _p(dc), _p(dd), _p(da),
        _p(g["means3D"]), _p(g["means2D"]), _p(g["shs"]) if s.shs is not None else None, _p(g["colors_precomp"]),
        _p(g["opacities"]), _p(g["scales"]), _p(g["rotations"]), _p(g["cov3D_precomp"]))
    return g


def dist2_knn3(points):
    pts = _f32(points)
    out = np.zeros(pts.shape[0], np.float32)
    lib().rr_dist2_knn3(C.c_int(pts.shape[0]), _p(pts), _p(out))
    return out
