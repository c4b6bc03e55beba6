"""Skeleton projection loss (train_rig.py:309-314): the four HIP launches against the same loss written as the reference
writes it (a dozen torch ops + an all-pairs distance + autograd), forward + backward, eager and inside a captured graph.

    python tools/skel_loss_bench.py [J] [M]
"""
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from riggs_amd import synth  # noqa: E402
from riggs_amd.loss import cal_skeleton_loss, sampling_steps, camera_intrinsics  # noqa: E402


def torch_loss(d_nodes, parents, cam, t):
    fx, fy, cx, cy = camera_intrinsics(cam)
    par = parents[1:].long()
    pts = (t[:, None, None] * d_nodes[1:] + (1 - t[:, None, None]) * d_nodes[par]).reshape(-1, 3)
    V = cam.world_view_transform
    tr = pts @ V[:3, :3] + V[3, :3]
    proj = torch.stack([fy * tr[:, 1] / tr[:, 2] + cy, fx * tr[:, 0] / tr[:, 2] + cx], -1)
    d = (proj[:, None, :] - cam.thinned[None, :, :]).abs().sum(-1)
    return d.min(1).values.mean() + d.min(0).values.mean()


def timed(fn, n=200):
    for _ in range(10):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3


def main():
    J = int(sys.argv[1]) if len(sys.argv) > 1 else 24
    M = int(sys.argv[2]) if len(sys.argv) > 2 else 1500
    sc = synth.make_scene(1000, J, 5)
    cam = synth.look_at_camera(800, 800, fid=0.4).to("cuda")
    cam.thinned = torch.stack([torch.randint(150, 650, (M,)), torch.randint(150, 650, (M,))], -1).float().cuda()
    parents = sc["parents"].cuda()
    nodes = sc["joints"].cuda().clone().requires_grad_(True)
    t = sampling_steps(nodes, parents)
    print("J=%d  S=%d  P=%d points  M=%d pixels" % (J, t.shape[0], t.shape[0] * (J - 1), M))

    def run(fn):
        nodes.grad = None
        fn(nodes, parents, cam, t).backward()

    hip = lambda *a: cal_skeleton_loss(a[0], a[1], a[2], t=a[3])  # noqa: E731
    for name, fn in (("torch ops + autograd", torch_loss), ("HIP (memset + 4 launches)", hip)):
        eager = timed(lambda: run(fn))
        s = torch.cuda.Stream()
        s.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(s):
            run(fn)
        torch.cuda.current_stream().wait_stream(s)
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=s):
            run(fn)
        graphed = timed(g.replay)
        print("%-22s eager %8.1f us   captured %7.1f us   (forward + backward)" % (name, eager, graphed))


if __name__ == "__main__":
    main()
