"""Build libriggs_hip.so (gfx950) in-tree with hipcc.  No torch headers are involved: the
library is a plain C-ABI shared object (include/riggs_hip.h)."""
from __future__ import annotations

import os
import shutil
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIBDIR = os.path.join(HERE, "lib")
SO = os.path.join(LIBDIR, "libriggs_hip.so")
TORCH_SO = os.path.join(LIBDIR, "libriggs_torch.so")
TORCH_SRC = os.path.join(HERE, "csrc_torch", "riggs_torch.cpp")

# translation unit -> extra flags.  preprocess.hip must keep FP contraction off (bit-exact
# geometry vs. the CPU oracle); the compositing kernels want FMAs.
SOURCES = {
    "preprocess.hip": ["-ffp-contract=off"],
    "render.hip": ["-ffp-contract=fast"],
    "binning.hip": ["-ffp-contract=off"],  # (hosts color_job.h: the same colours, bit for bit, as preprocess.hip)
    "deform.hip": ["-ffp-contract=fast"],
    "knn.hip": ["-ffp-contract=fast"],
    "pose_mlp.hip": ["-ffp-contract=fast"],
    "optim.hip": ["-ffp-contract=off"],
    "loss.hip": ["-ffp-contract=fast"],
    "skel_loss.hip": ["-ffp-contract=off"],
    "cnode.hip": ["-ffp-contract=off"],
    "mlp.hip": ["-ffp-contract=fast"],
    "mlp_wgrad.hip": ["-ffp-contract=fast"],
    "exchange.hip": ["-ffp-contract=off"],
    "dq.hip": ["-ffp-contract=off"],
    "frame.hip": [],
    "densify.hip": ["-ffp-contract=off"],
    "capi.hip": [],
}
COMMON = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-fhip-fp32-correctly-rounded-divide-sqrt",
          "-fno-fast-math", "-Wall", "-Wno-unused-function", "-Wno-unused-result"]


def hipcc() -> str:
    for c in (os.environ.get("HIPCC"), shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if c and os.path.exists(c):
            return c
    raise RuntimeError("hipcc not found")


def _stale(target, deps):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


def build(force: bool = False, verbose: bool = False) -> str:
    os.makedirs(LIBDIR, exist_ok=True)
    objdir = os.path.join(LIBDIR, "obj")
    os.makedirs(objdir, exist_ok=True)
    headers = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".h")]
    headers.append(os.path.join(os.path.dirname(HERE), "include", "riggs_hip.h"))
    cc = hipcc()
    objs = []
    procs = []
    for src, extra in SOURCES.items():
        sp = os.path.join(CSRC, src)
        obj = os.path.join(objdir, src.replace(".hip", ".o"))
        objs.append(obj)
        if force or _stale(obj, [sp] + headers):
            cmd = [cc, "-c", sp, "-o", obj] + COMMON + extra
            if verbose:
                print(" ".join(cmd))
            procs.append((src, subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT)))
    failed = False
    for src, p in procs:
        out, _ = p.communicate()
        if p.returncode != 0:
            failed = True
            sys.stderr.write("hipcc failed for %s:\n%s\n" % (src, out.decode()))
        elif verbose and out:
            print(out.decode())
    if failed:
        raise RuntimeError("libriggs_hip build failed")
    if force or procs or _stale(SO, objs):
        cmd = [cc, "-shared", "-o", SO, "--offload-arch=gfx950"] + objs
        subprocess.check_call(cmd)
    return SO


def build_torch(force: bool = False, verbose: bool = False) -> str:
    """libriggs_torch.so: the PyTorch front-end (csrc_torch/riggs_torch.cpp — TORCH_LIBRARY ops whose autograd nodes call the C ABI
    of libriggs_hip.so), host C++ only: g++ against torch's headers and libraries, linked to libriggs_hip.so next to it ($ORIGIN).
    Optional at run time (riggs_amd/_torch_ext.py falls back to the ctypes nodes), built by __graft_entry__.build()."""
    import torch
    from torch.utils import cpp_extension as X
    build(force=False, verbose=verbose)
    header = os.path.join(os.path.dirname(HERE), "include", "riggs_hip.h")
    if not (force or _stale(TORCH_SO, [TORCH_SRC, header, SO])):
        return TORCH_SO
    rocm = os.environ.get("ROCM_PATH", "/opt/rocm")
    cmd = ["g++", "-O2", "-std=c++17", "-fPIC", "-shared", "-D__HIP_PLATFORM_AMD__=1", "-DUSE_ROCM=1",
           "-D_GLIBCXX_USE_CXX11_ABI=%d" % int(torch._C._GLIBCXX_USE_CXX11_ABI), "-Wall", "-Wno-unused-function", TORCH_SRC, "-o", TORCH_SO]
    cmd += ["-I" + i for i in X.include_paths()] + ["-I" + os.path.join(rocm, "include")]
    cmd += ["-L" + l for l in X.library_paths()] + ["-L" + LIBDIR, "-Wl,-rpath,$ORIGIN"]
    cmd += ["-lriggs_hip", "-ltorch", "-ltorch_cpu", "-lc10", "-lc10_hip", "-ltorch_hip"]
    if verbose:
        print(" ".join(cmd))
    subprocess.check_call(cmd)
    return TORCH_SO


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
    print(build_torch(force="--force" in sys.argv, verbose=True))
