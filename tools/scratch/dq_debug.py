import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from oracle import dq_ref as O
from riggs_amd import dual_quaternion as DQ
np.set_printoptions(precision=4, suppress=True, linewidth=200)
rng = np.random.default_rng(3)
N, K = 12, 1
q = rng.normal(size=(K, 4)); t = 0.5 * rng.normal(size=(K, 3)); w = np.ones((N, K))
g_rot = rng.normal(size=(N, 4)) * float(os.environ.get('GR', '1')); g_t = rng.normal(size=(N, 3)) * float(os.environ.get('GT', '1'))
rot_o, t_o, cache = O.dq_blending(q[None], t[None], w, True, norm_over_nodes=False)
gq_o, gt_o, gw_o = O.dq_blending_backward(cache, g_rot, g_t)
dev = lambda a, g=False: torch.from_numpy(np.ascontiguousarray(a)).float().cuda().requires_grad_(g)
qd, td, wd = dev(q, True), dev(t, True), dev(w, True)
rot, t_ = DQ._DQBlend.apply(qd, td, wd, True, False, 1)
((rot * dev(g_rot)).sum() + (t_ * dev(g_t)).sum()).backward()
print("best", cache[1][3][0])
print("rot hip\n", rot.detach().cpu().numpy()[:3], "\nrot ora\n", rot_o[:3])
print("gw hip", wd.grad.cpu().numpy().reshape(-1), "\ngw ora", gw_o.reshape(-1))
print("gq hip", qd.grad.cpu().numpy(), "\ngq ora", gq_o)
print("gt hip", td.grad.cpu().numpy(), "\ngt ora", gt_o)
# rows form, same data
q3 = np.repeat(q[None], N, 0); t3 = np.repeat(t[None], N, 0)
qd, td, wd = dev(q3, True), dev(t3, True), dev(w, True)
rot, t_ = DQ._DQBlend.apply(qd, td, wd, False, False, 1)
((rot * dev(g_rot)).sum() + (t_ * dev(g_t)).sum()).backward()
rot_o, t_o, cache = O.dq_blending(q3, t3, w, True, norm_over_nodes=False)
gq_o, gt_o, gw_o = O.dq_blending_backward(cache, g_rot, g_t)
print("ROWS gq hip\n", qd.grad.cpu().numpy()[:4, 0], "\ngq ora\n", gq_o[:4, 0])
print("ROWS gw hip", wd.grad.cpu().numpy().reshape(-1)[:6], "\ngw ora", gw_o.reshape(-1)[:6])
