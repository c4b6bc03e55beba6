"""Which RCCL collectives survive a hipGraph capture on this stack (one-rank communicator, each pattern in its own process:
a failure here is a segfault in hipStreamEndCapture)?   python tools/scratch/rccl_capture_probe.py"""
import os
import socket
import subprocess
import sys

PATTERNS = ["allreduce_sync", "allreduce_async_wait", "allgather_sync", "allgather_async_side_stream", "newgroup_allreduce", "two_collectives_two_groups",
            "allgather_async_compute_wait", "two_async_two_groups_same_stream"]


def child(pat):
    import torch
    import torch.distributed as dist
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda:0"))
    x = torch.ones(1 << 20, device="cuda")
    y = torch.zeros(1 << 20, device="cuda")
    g2 = dist.new_group() if "group" in pat else None
    side = torch.cuda.Stream()

    def body():
        if pat == "allreduce_sync":
            dist.all_reduce(x)
        elif pat == "allreduce_async_wait":
            w = dist.all_reduce(x, async_op=True)
            w.wait()
        elif pat == "allgather_sync":
            dist.all_gather_into_tensor(y, x)
        elif pat == "allgather_async_side_stream":
            side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side):
                w = dist.all_gather_into_tensor(y, x, async_op=True)
            x.mul_(1.0)
            with torch.cuda.stream(side):
                w.wait()
            torch.cuda.current_stream().wait_stream(side)
        elif pat == "allgather_async_compute_wait":  # no side stream of ours: the process group forks to its own and joins at wait()
            w = dist.all_gather_into_tensor(y, x, async_op=True)
            z = x * 2.0
            z.add_(1.0)
            w.wait()
        elif pat == "two_async_two_groups_same_stream":
            w1 = dist.all_gather_into_tensor(y, x, async_op=True)
            z = x * 2.0
            w2 = dist.all_reduce(z, group=g2, async_op=True)
            w1.wait()
            w2.wait()
        elif pat == "newgroup_allreduce":
            dist.all_reduce(x, group=g2)
        elif pat == "two_collectives_two_groups":
            side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side):
                w1 = dist.all_gather_into_tensor(y, x, async_op=True)
                w2 = dist.all_reduce(x, group=g2, async_op=True)
                w1.wait()
                w2.wait()
            torch.cuda.current_stream().wait_stream(side)
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        for _ in range(3):
            body()  # eager warm-up: brings the communicators up
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, stream=s):
        body()
    for _ in range(5):
        g.replay()
    torch.cuda.synchronize()
    print("OK", pat, float(x[0]), float(y[0]), flush=True)
    dist.destroy_process_group()


if __name__ == "__main__":
    if len(sys.argv) > 1:
        child(sys.argv[1])
    else:
        for p in PATTERNS:
            r = subprocess.run([sys.executable, os.path.abspath(__file__), p], capture_output=True, text=True, timeout=120)
            ok = [l for l in r.stdout.splitlines() if l.startswith("OK")]
            print(p, "->", ok[0] if ok else "FAILED rc=%d %s" % (r.returncode, (r.stderr or "").strip().splitlines()[-1:]), flush=True)
