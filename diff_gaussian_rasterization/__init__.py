"""Drop-in for the `diff_gaussian_rasterization` CUDA extension RigGS imports at
gaussian_renderer/__init__.py:14 — backed by the gfx950 HIP library (riggs_amd)."""
from riggs_amd.rasterizer import GaussianRasterizationSettings, GaussianRasterizer  # noqa: F401

__all__ = ["GaussianRasterizationSettings", "GaussianRasterizer"]
