"""Compiles csrc/dq_math.h for the HOST (g++) and runs the DQB row arithmetic on the reference's golden inputs: a check of the
hand-derived backward that needs no GPU.  python tools/scratch/dq_math_check.py"""
import os
import subprocess
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
exe = "/tmp/dq_math_host"
subprocess.check_call([os.environ.get("CXX", "g++"), os.environ.get("OPT", "-O1"), "-ffp-contract=off", "-o", exe, os.path.join(ROOT, "tools/scratch/dq_math_host.cpp")])


def run(q, t, w, g_rot, g_t, nn, om):
    N, K = w.shape
    rw = (9, 4, 16)[om]
    buf = b"".join(np.ascontiguousarray(x, np.float32).tobytes() for x in (q, t, w, g_rot, g_t))
    out = np.frombuffer(subprocess.run([exe, "rows", str(N), str(K), str(int(nn)), str(om)], input=buf, capture_output=True, check=True).stdout, np.float32)
    sizes = [N * rw, N * 3, N * K * 4, N * K * 3, N * K]
    parts, o = [], 0
    for s in sizes:
        parts.append(out[o:o + s])
        o += s
    return parts


G = os.path.join(ROOT, "tests", "golden")
worst = 0.0
for name in ("dqb_interp_q", "dqb_rows3d_k3_q", "dqb_rows3d_k8_R"):
    g = np.load(os.path.join(G, name + ".npz"))
    as_q = bool(g["rot_as_q"])
    if str(g["mode"]) == "interp":
        q, t = np.stack([g["q0"], g["q1"]], 1), np.stack([g["t0"], g["t1"]], 1)
        w = np.concatenate([g["weight"], 1 - g["weight"]], 1)
        nn = 0
        want = dict(gq=np.stack([g["grad_q0"], g["grad_q1"]], 1), gt=np.stack([g["grad_t0"], g["grad_t1"]], 1))
    else:
        q, t, w, nn = g["q"], g["t"], g["weights"], 1
        want = dict(gq=g["grad_q"], gt=g["grad_t"], gw=g["grad_weights"])
    rot, t_, gq, gt, gw = run(q, t, w, g["g_rot"].reshape(w.shape[0], -1), g["g_t"], nn, 1 if as_q else 0)
    res = dict(rot=(rot, g["out_rot"]), t=(t_, g["out_t"]), gq=(gq, want["gq"]), gt=(gt, want["gt"]))
    if "gw" in want:
        res["gw"] = (gw, want["gw"])
    for k, (a, b) in res.items():
        e = float(np.abs(a.reshape(-1) - b.reshape(-1)).max()) / max(1.0, float(np.abs(b).max()))
        worst = max(worst, e)
        print("%-18s %-4s max err / max(1, |ref|) = %.3g" % (name, k, e))
print("worst", worst)
sys.exit(0 if worst < 1e-4 else 1)
