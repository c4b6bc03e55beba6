import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from riggs_amd import mlp as M
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
from tests.test_gpu_mlp import _nets
N = int(sys.argv[1]) if len(sys.argv) > 1 else 50003
for name, net, head, xe in _nets(N):
    fh = M.FusedHead(net.linear, head, xe.shape[1], net.skips[0], "fp16")
    pk = fh._packed()
    xb = M.embed_bf16(pk, xe)
    out0, _ = M.forward(pk, xe, False, xb)
    for rep in range(3):
        out, (acts, masks) = M.forward(pk, xe, True, xb)
        torch.cuda.synchronize()
        print(name, "rep", rep, "acts finite", bool(torch.isfinite(acts.float()).all()), "out == inference out", float((out - out0).abs().max()))
        # reference activations from the stored ones: layer l from layer l-1 (16-bit operands, fp32 accumulate)
        x16 = xb[:N, :pk.in_ch].float()
        lin = list(net.linear)
        for l in range(pk.depth):
            W = lin[l].weight.detach().to(pk.dtype).float(); b = lin[l].bias.detach().float()
            a_in = x16 if l == 0 else acts[l - 1].float()
            if l == pk.skip + 1:
                a_in = torch.cat([x16, a_in], 1)
            ref = torch.relu(a_in @ W.t() + b).to(pk.dtype).float()
            d = (acts[l].float() - ref).abs()
            bad = (d > 2e-2 * (1 + ref.abs())).nonzero()
            print("  layer", l, "max diff", float(d.max()), "bad entries", bad.shape[0], ("first " + str(bad[:4].tolist())) if bad.shape[0] else "")
