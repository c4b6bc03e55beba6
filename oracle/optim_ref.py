"""CPU restatement of the Gaussian optimizer step and the densification statistics.

TEST INFRASTRUCTURE ONLY (tests/, __graft_entry__.smoke(), bench.py's cpu_baseline leg) — never imported by the product.

* ``adam_step``: what ``self.gaussians.optimizer.step()`` (/root/reference/train_rig.py:527) does for one parameter
  tensor.  The optimizer is ``torch.optim.Adam(l, lr=0.0, eps=1e-15)`` with one parameter per group and per-group
  learning rates (/root/reference/scene/gaussian_model.py:205-217): plain Adam — betas (0.9, 0.999), no weight decay,
  no amsgrad — in the operation order of torch's single-tensor implementation (torch/optim/adam.py,
  ``_single_tensor_adam``: lerp_, mul_/addcmul_, sqrt/div/add_, addcdiv_), all arithmetic in float32, the bias
  corrections in Python doubles.
* ``densification_stats``: /root/reference/scene/gaussian_model.py:516-518 (add_densification_stats) plus the
  max_radii2D update of /root/reference/train_rig.py:333-335.

Pinned by tests/golden/optim_adam_n67.npz, produced by running the reference's own ``GaussianModel.training_setup`` /
``optimizer.step`` / ``update_learning_rate`` / ``add_densification_stats`` on CPU (tests/golden/make_golden.py).
"""
import math

import numpy as np


def adam_step(p, g, m, v, step, lr, beta1=0.9, beta2=0.999, eps=1e-15):
    """One Adam update of (p, m, v) with gradient g at (1-based) step count ``step``.  float32 arrays, returns new ones."""
    f = np.float32
    p, g, m, v = (np.asarray(a, dtype=np.float32) for a in (p, g, m, v))
    m = m + f(1.0 - beta1) * (g - m)                      # exp_avg.lerp_(grad, 1 - beta1)   (weight < 0.5 branch)
    v = v * f(beta2) + (f(1.0 - beta2) * g) * g           # exp_avg_sq.mul_(beta2).addcmul_(grad, grad, value=1 - beta2)
    bc1 = 1.0 - beta1 ** step
    bc2 = 1.0 - beta2 ** step
    step_size = lr / bc1
    denom = np.sqrt(v) / f(math.sqrt(bc2)) + f(eps)      # (exp_avg_sq.sqrt() / bias_correction2_sqrt).add_(eps)
    p = p + f(-step_size) * (m / denom)                   # param.addcdiv_(exp_avg, denom, value=-step_size)
    return p.astype(np.float32), m.astype(np.float32), v.astype(np.float32)


def densification_stats(viewspace_grad, update_filter, xyz_gradient_accum, denom, radii=None, max_radii2D=None):
    """xyz_gradient_accum[f] += ||viewspace_grad[f, :2]||; denom[f] += 1; max_radii2D[f] = max(., radii[f])."""
    gacc = np.array(xyz_gradient_accum, dtype=np.float32).reshape(-1).copy()
    den = np.array(denom, dtype=np.float32).reshape(-1).copy()
    f = np.asarray(update_filter, dtype=bool).reshape(-1)
    g = np.asarray(viewspace_grad, dtype=np.float32)
    nrm = np.sqrt(g[:, 0] * g[:, 0] + g[:, 1] * g[:, 1]).astype(np.float32)
    gacc[f] += nrm[f]
    den[f] += np.float32(1.0)
    out = [gacc.reshape(-1, 1), den.reshape(-1, 1)]
    if max_radii2D is not None:
        mr = np.array(max_radii2D, dtype=np.float32).copy()
        mr[f] = np.maximum(mr[f], np.asarray(radii, dtype=np.float32)[f])
        out.append(mr)
    return tuple(out)
