import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from oracle import dq_ref as O
from riggs_amd import dual_quaternion as DQ
np.set_printoptions(precision=4, suppress=True, linewidth=200)
rng = np.random.default_rng(3)
N, K = 4000, 2
q = rng.normal(size=(N, K, 4)); t = 0.5 * rng.normal(size=(N, K, 3)); w = rng.random((N, K)); w /= w.sum(-1, keepdims=True)
g_rot = rng.normal(size=(N, 4)); g_t = rng.normal(size=(N, 3))
rot_o, t_o, cache = O.dq_blending(q, t, w, True, norm_over_nodes=False)
gq_o, gt_o, gw_o = O.dq_blending_backward(cache, g_rot, g_t)
best = cache[1][3][0]
dev = lambda a, g=False: torch.from_numpy(np.ascontiguousarray(a)).float().cuda().requires_grad_(g)
qd, td, wd = dev(q, True), dev(t, True), dev(w, True)
rot, t_ = DQ._DQBlend.apply(qd, td, wd, False, False, 1)
((rot * dev(g_rot)).sum() + (t_ * dev(g_t)).sum()).backward()
eq = np.abs(qd.grad.cpu().numpy() - gq_o).reshape(N, -1).max(1)
ew = np.abs(wd.grad.cpu().numpy() - gw_o).reshape(N, -1).max(1)
er = np.abs(rot.detach().cpu().numpy() - rot_o).max(1)
for b in range(4):
    m = best == b
    print("best", b, "rows", m.sum(), "max err gq %.3g gw %.3g rot %.3g" % (eq[m].max(), ew[m].max(), er[m].max()), "bad rows", (eq[m] > 1e-3).sum())
bad = np.nonzero(eq > 1e-3)[0][:3]
for i in bad:
    print("row", i, "best", best[i], "q_abs-ish rot", rot_o[i], "\n hip gq", qd.grad.cpu().numpy()[i], "\n ora gq", gq_o[i], "\n hip gw", wd.grad.cpu().numpy()[i], "ora gw", gw_o[i])
    R = O.dq2qt(cache[1][0][i:i+1] * 0, False) if False else None
sgn = cache[0][4]
print("bad rows with any sgn<0:", (sgn[eq > 1e-3] < 0).any(axis=(1, 2)).sum(), "of", (eq > 1e-3).sum(), "; good rows with sgn<0:", (sgn[eq <= 1e-3] < 0).any(axis=(1, 2)).sum(), "of", (eq <= 1e-3).sum())
