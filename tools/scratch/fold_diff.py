import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from riggs_amd import mlp as M
from tests.test_gpu_mlp import _nets, _sparse_cotangent
N = 20011
name, net, head, xe = _nets(N)[0]
g = _sparse_cotangent(N, 23, 0.2, mag=3e-7)
res = {}
for tag in ("unfold", "fold", "unfold_mysig"):
    for q in net.parameters():
        q.grad = None
    fh = M.FusedHead(net.linear, head, xe.shape[1], net.skips[0], sparse_rows=False, out_sigmoid=(tag == "fold"))
    if tag == "fold":
        out = fh(xe)
        torch.autograd.backward([out], [g])
    elif tag == "unfold":
        out = torch.sigmoid(fh(xe))
        torch.autograd.backward([out], [g])
    else:
        o = fh(xe)
        s = 1.0 / (1.0 + torch.exp(-o.detach()))
        torch.autograd.backward([o], [g * (s * (1 - s))])
    res[tag] = [q.grad.clone() for q in net.parameters()]
for k in ("fold", "unfold_mysig"):
    print(k, [("%s %.2e" % (n_, float((a - b).abs().max() / b.abs().max()))) for a, b, (n_, _) in zip(res[k], res["unfold"], net.named_parameters())][:6])
