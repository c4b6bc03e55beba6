"""Eager drop-in frame and eager training iteration with the PyTorch extension's nodes on / off (same process, same box)."""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import bench
from riggs_amd import _torch_ext as TX
sc, cam, gm, sw = bench.build_workload(0, "cuda:0")
gimg = torch.rand(3, bench.WORKLOAD["H"], bench.WORKLOAD["W"], device="cuda") * 1e-6
for on in (False, True, False, True):
    TX.enable(on)
    e = bench.eager_api_timing(cam, gm, sw, gimg)
    print("extension %s: two calls %.4f ms %s | frame entry %.4f ms" % (on, e["two_calls_ms"], e["two_calls_blocks_ms"], e["frame_entry_ms"]), flush=True)
for on in (False, True):
    TX.enable(on)
    t = bench.train_step_timing(sc, cam, gm, sw, steps=40)
    print("extension %s: train_step captured %.4f ms, eager %.4f ms" % (on, t["ms_per_step"], t["eager_ms_per_step"]), flush=True)
