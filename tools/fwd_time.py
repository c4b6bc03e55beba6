"""Event-timed duration of the compositing kernels (forward, backward) on the bench workload and on the dense-gradient scene,
eagerly issued: the quick A/B number for kernel experiments.   usage: python tools/fwd_time.py [label]"""
import ctypes as C
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from riggs_amd import _lib as L  # noqa: E402
from riggs_amd.dist import FlatGradAllReduce  # noqa: E402
from riggs_amd.rasterizer import RasterArena  # noqa: E402


def timed(scene):
    w = bench.WORKLOAD
    sc, cam, gm, sw = bench.build_workload(0, "cuda:0")
    if scene == "dense":
        from riggs_amd import synth
        from riggs_amd.gaussian_model import GaussianModel
        sc = synth.make_surface_scene(w["N"], w["J"], w["seed"])
        gm = GaussianModel.from_tensors(sc["xyz"], sc["features_dc"], sc["features_rest"], sc["scaling"], sc["rotation"],
                                        sc["opacity"], device="cuda:0")
    gimg = torch.rand(3, w["H"], w["W"], device="cuda") * 1e-6
    step = bench.make_step(cam, gm, sw, gimg, RasterArena(), 1, FlatGradAllReduce(bench.params_of(gm, sw), register=False))
    lib = L.lib()
    for _ in range(5):
        step()
    torch.cuda.synchronize()
    lib.riggs_prof_reset()
    lib.riggs_prof_enable(0xFFFFFFFF)
    for _ in range(30):
        step()
    torch.cuda.synchronize()
    lib.riggs_prof_enable(0)
    out = {}
    tot, cnt = C.c_float(), C.c_int32()
    lib.riggs_prof_name.restype = C.c_char_p
    for i in range(lib.riggs_prof_count()):
        nm = lib.riggs_prof_name(i).decode()
        L.check(lib.riggs_prof_read(i, C.byref(tot), C.byref(cnt)), "riggs_prof_read")
        if cnt.value and nm in ("render_fwd", "render_bwd", "tile_sort", "preprocess_fwd"):
            out[nm] = round(1e3 * tot.value / cnt.value, 1)
    return out


if __name__ == "__main__":
    label = sys.argv[1] if len(sys.argv) > 1 else ""
    print(label, "headline us:", timed("headline"), " dense us:", timed("dense"), flush=True)
