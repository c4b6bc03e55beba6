import sys, time, torch
sys.path.insert(0, "/root/repo")
from riggs_amd import mlp as M
x = torch.randn(300000, 3, device="cuda") * 0.5
pose = torch.randn(96, device="cuda")
for mr, tail in ((10, None), (4, pose)):
    for _ in range(5): M.embed_positions_bf16(x, mr, tail, fmt="fp16")
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(50): M.embed_positions_bf16(x, mr, tail, fmt="fp16")
    torch.cuda.synchronize(); print("multires %d tail %s: %.1f us" % (mr, tail is not None, (time.perf_counter() - t0) / 50 * 1e6))
