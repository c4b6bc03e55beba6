import sys; sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
import torch
from riggs_amd import mlp as M
import test_gpu_mlp as T
N = 20011
for fmt in ("fp16", "bf16"):
    for name, net, head, xe in T._nets(N):
        g = torch.randn(N, head.weight.shape[0], device="cuda")
        out32 = head(T._hidden(net, xe)[0]); (out32 * g).sum().backward()
        g32 = {n: q.grad.clone() for n, q in net.named_parameters()}
        for q in net.parameters(): q.grad = None
        fh = M.FusedHead(net.linear, head, xe.shape[1], net.skips[0], fmt)
        out = fh(xe); (out * g).sum().backward()
        print(fmt, name, "out rel", T._rel(out.detach(), out32.detach()))
        print("   grads mean-rel:", " ".join("%s=%.4f" % (n.replace("linear.", "l").replace("weight", "w").replace("bias", "b"), T._mrel(q.grad, g32[n])) for n, q in net.named_parameters()))
        for q in net.parameters(): q.grad = None
