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


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
