"""simple_knn._C.distCUDA2 — HIP implementation (riggs_amd/csrc/knn.hip)."""
from riggs_amd.knn import distCUDA2  # noqa: F401
