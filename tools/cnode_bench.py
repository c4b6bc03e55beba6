"""Stage-1 control-node deformation (ControlNodeWarp.forward, time_utils.py:1133-1191) at the bench scale: the HIP kernels
against the same computation written as the reference writes it (all-pairs distances + topk standing in for pytorch3d's KNN,
gathers, einsum, autograd), forward and forward + backward.

    python tools/cnode_bench.py [N] [M] [hyper]
"""
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from riggs_amd.control_nodes import control_node_blend  # noqa: E402


def quaternion_to_matrix(q):
    r, i, j, k = torch.unbind(q, -1)
    s = 2.0 / (q * q).sum(-1)
    o = torch.stack((1 - s * (j * j + k * k), s * (i * j - k * r), s * (i * k + j * r), s * (i * j + k * r),
                     1 - s * (i * i + k * k), s * (j * k - i * r), s * (i * k - j * r), s * (j * k + i * r),
                     1 - s * (i * i + j * j)), -1)
    return o.reshape(q.shape[:-1] + (3, 3))


def torch_version(x, feature, mask, nodes, radius, weight, attrs, K, hyper, chunk=65536):
    xa = torch.cat([x, feature[:, :hyper]], -1)
    na = torch.cat([nodes[:, :3].detach(), nodes[:, 3:]], -1)
    d, idx = [], []
    for s in range(0, x.shape[0], chunk):  # (N, M) distances in chunks: 300k x 1024 floats at once is 1.2 GB
        dd = ((xa[s:s + chunk, None] - na[None]) ** 2).sum(-1)
        v, i = dd.topk(K, dim=1, largest=False)
        d.append(v)
        idx.append(i)
    d, idx = torch.cat(d), torch.cat(idx)
    w = torch.exp(-d / (2 * torch.exp(radius)[idx] ** 2)) * torch.sigmoid(weight)[idx][..., 0] + 1e-7
    w = w / w.sum(-1, keepdim=True)
    R = quaternion_to_matrix(attrs["local_rotation"] + torch.tensor([1.0, 0, 0, 0], device=x.device))
    nn = nodes[idx, :3].detach()
    Ax = torch.einsum("nkab,nkb->nka", R[idx], x[:, None] - nn) + nn + attrs["d_xyz"][idx]
    return {"d_xyz": ((Ax * w[..., None]).sum(1) - x) * mask, "d_rotation": (attrs["d_rotation"][idx] * w[..., None]).sum(1) * mask,
            "d_scaling": (attrs["d_scaling"][idx] * w[..., None]).sum(1) * mask}


def timed(fn, n=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / n


def main():
    N = int(sys.argv[1]) if len(sys.argv) > 1 else 300_000
    M = int(sys.argv[2]) if len(sys.argv) > 2 else 512
    hyper = int(sys.argv[3]) if len(sys.argv) > 3 else 8
    K = 3
    g = torch.Generator().manual_seed(0)
    x = (torch.randn(N, 3, generator=g) * 0.5).cuda()
    P = lambda t: t.cuda().requires_grad_(True)  # noqa: E731
    nodes = P(torch.cat([x[torch.randint(0, N, (M,), generator=g)].cpu(), 1e-2 + 0.02 * torch.randn(M, hyper, generator=g)], -1))
    feature = P(0.02 * torch.randn(N, hyper + 1, generator=g))
    mask = P(torch.rand(N, 1, generator=g))
    radius, weight = P(-1.9 + 0.3 * torch.randn(M, generator=g)), P(0.5 * torch.randn(M, 1, generator=g))
    attrs = {"d_xyz": P(0.1 * torch.randn(M, 3, generator=g)), "d_rotation": P(0.2 * torch.randn(M, 4, generator=g)),
             "d_scaling": P(0.05 * torch.randn(M, 3, generator=g)), "local_rotation": P(0.3 * torch.randn(M, 4, generator=g))}
    go = {k: torch.randn(N, w, device="cuda") for k, w in (("d_xyz", 3), ("d_rotation", 4), ("d_scaling", 3))}
    leaves = [nodes, feature, mask, radius, weight] + list(attrs.values())

    def hip():
        return control_node_blend(x, feature, mask, nodes, radius, weight, attrs, K=K, hyper_dim=hyper, local_frame=True, d_rot_as_res=True)

    def ref():
        return torch_version(x, feature, mask, nodes, radius, weight, attrs, K, hyper)

    def both(fn):
        for t in leaves:
            t.grad = None
        out = fn()
        torch.autograd.backward([out[k] for k in go], [go[k] for k in go])

    print("N=%d Gaussians, M=%d nodes, K=%d, hyper_dim=%d, local_frame" % (N, M, K, hyper))
    for name, fn in (("torch ops (reference style)", ref), ("HIP kernels", hip)):
        with torch.no_grad():
            f = timed(fn)
        fb = timed(lambda: both(fn))
        print("%-28s forward %8.3f ms   forward + backward %8.3f ms" % (name, f, fb), flush=True)
    a, b = hip(), ref()
    print("max |d_xyz difference| between the two: %.2e" % float((a["d_xyz"] - b["d_xyz"]).abs().max()))


if __name__ == "__main__":
    main()
