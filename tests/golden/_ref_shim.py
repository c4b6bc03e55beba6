"""Import harness for the RigGS reference Python (CPU, this container only).

Used ONLY by tests/golden/make_golden.py to emit golden vectors.  Nothing here
travels to the GPU box as executable reference code: the harness merely puts
/root/reference on sys.path, neutralises the hard-coded ``.cuda()`` calls and
registers stub modules for third-party packages that are absent in this image
(SURVEY.md Appendix D).  The reference sources are never copied.
"""
import sys
import types
import contextlib
import io
from typing import NamedTuple

import numpy as np
import torch

REF = "/root/reference"
CAPTURE = {}


def _to_cpu_device(kw):
    dev = kw.get("device", None)
    if dev is not None and "cuda" in str(dev):
        kw["device"] = "cpu"
    return kw


def install():
    sys.dont_write_bytecode = True
    if REF not in sys.path:
        sys.path.insert(0, REF)

    torch.Tensor.cuda = lambda self, *a, **k: self
    torch.nn.Module.cuda = lambda self, *a, **k: self

    for name in ["zeros", "ones", "zeros_like", "ones_like", "empty", "tensor",
                 "rand", "randn", "full", "arange", "normal", "eye", "linspace"]:
        orig = getattr(torch, name)

        def wrap(*a, __orig=orig, **k):
            return __orig(*a, **_to_cpu_device(k))
        setattr(torch, name, wrap)

    orig_to = torch.Tensor.to

    def to(self, *a, **k):
        a = tuple("cpu" if (isinstance(x, str) and "cuda" in x) else x for x in a)
        a = tuple(torch.device("cpu") if (isinstance(x, torch.device) and x.type == "cuda") else x for x in a)
        return orig_to(self, *a, **_to_cpu_device(k))
    torch.Tensor.to = to
    orig_mto = torch.nn.Module.to

    def mto(self, *a, **k):
        a = tuple("cpu" if (isinstance(x, str) and "cuda" in x) else x for x in a)
        return orig_mto(self, *a, **_to_cpu_device(k))
    torch.nn.Module.to = mto
    orig_device = torch.device

    def mk(name, **attrs):
        m = types.ModuleType(name)
        for k, v in attrs.items():
            setattr(m, k, v)
        sys.modules[name] = m
        return m

    def _nope(*a, **k):
        raise RuntimeError("stubbed third-party op called")

    mk("pytorch3d")
    mk("pytorch3d.ops", knn_points=_nope, ball_query=_nope)
    mk("pytorch3d.loss", chamfer_distance=_nope)
    mk("plyfile", PlyData=object, PlyElement=object)
    mk("cv2")
    mk("imageio")
    mk("openmesh")

    def dist_bruteforce(p):
        d = torch.cdist(p.double(), p.double()) ** 2
        d.fill_diagonal_(float("inf"))
        return d.topk(3, dim=1, largest=False).values.mean(1).float()
    mk("simple_knn")
    mk("simple_knn._C", distCUDA2=dist_bruteforce)

    class GaussianRasterizationSettings(NamedTuple):
        image_height: int
        image_width: int
        tanfovx: float
        tanfovy: float
        bg: torch.Tensor
        scale_modifier: float
        viewmatrix: torch.Tensor
        projmatrix: torch.Tensor
        sh_degree: int
        campos: torch.Tensor
        prefiltered: bool
        debug: bool

    class GaussianRasterizer(torch.nn.Module):
        """Capturing stand-in: records exactly what render() hands over."""

        def __init__(self, raster_settings):
            super().__init__()
            self.raster_settings = raster_settings

        def forward(self, **kw):
            CAPTURE.clear()
            CAPTURE["settings"] = self.raster_settings
            CAPTURE["kwargs"] = kw
            s = self.raster_settings
            n = kw["means3D"].shape[0]
            img = torch.zeros(3, s.image_height, s.image_width)
            return (img, torch.zeros(n, dtype=torch.int32),
                    torch.zeros(1, s.image_height, s.image_width),
                    torch.zeros(1, s.image_height, s.image_width))

    mk("diff_gaussian_rasterization",
       GaussianRasterizationSettings=GaussianRasterizationSettings,
       GaussianRasterizer=GaussianRasterizer)


@contextlib.contextmanager
def quiet():
    buf = io.StringIO()
    with contextlib.redirect_stdout(buf):
        yield
