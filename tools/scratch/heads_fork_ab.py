"""Heads-on captured iteration with the DeformMLP's backward on a side stream of the capture (GraphedTrainStep.fork_heads) against
the single-stream capture, same process, alternating."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bench
from riggs_amd.graph import GraphedTrainStep
for rep in range(2):
    for fork in (False, True):
        GraphedTrainStep.FORK_HEADS = fork
        r = bench.train_step_heads_timing("cuda:0")
        print("fork_heads", fork, r["ms_per_step"], r["heads_off_ms_per_step"], r["heads_ms"], "final loss", r["final_loss"],
              "template loss", r["template_offsets_loss"], flush=True)
GraphedTrainStep.FORK_HEADS = True
r = bench.train_step_heads_timing("cuda:0", surface=True, steps=20)
print("dense scene, fork", r["ms_per_step"], flush=True)
