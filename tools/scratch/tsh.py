import os, sys, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch, bench
print(json.dumps(bench.train_step_heads_timing("cuda:0"), indent=1))
print(json.dumps(bench.train_step_heads_timing("cuda:0", surface=True, steps=20), indent=1))
