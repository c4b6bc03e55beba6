import os, sys, re
sys.path.insert(0, "/root/repo")
import torch
import bench
from riggs_amd.graph import GraphedFrame
sc, cam, gm, sw = bench.build_workload(0, "cuda:0")
params = bench.params_of(gm, sw)
gf = GraphedFrame(gm, sw, cam, torch.zeros(3, device="cuda"), params)
import riggs_amd.graph as G
orig = torch.cuda.CUDAGraph
class DG(orig):
    def __new__(cls, *a, **k):
        g = orig.__new__(cls)
        return g
    def __init__(self):
        super().__init__()
        self.enable_debug_mode()
torch.cuda.CUDAGraph = DG
gf.capture()
os.makedirs("/root/repo/gpurun_out", exist_ok=True); gf.graph.debug_dump("/root/repo/gpurun_out/graph.dot")
txt = open("/root/repo/gpurun_out/graph.dot").read()
labels = re.findall(r'label="([^"]*)"', txt)
for l in labels:
    print(l.replace("\\n", " | ")[:200])
