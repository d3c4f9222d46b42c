"""bench.py in a process that has already made N torch streams (what a serving process would have): python tools/r06/bench_after_streams.py N [bench args]"""
import os
import runpy
import sys

import torch

n = int(sys.argv[1])
keep = [torch.cuda.Stream("cuda:0") for _ in range(n)]
x = torch.zeros(8, device="cuda:0")
for s in keep:
    with torch.cuda.stream(s):
        x.add_(1.0)
torch.cuda.synchronize()
root = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.argv = [os.path.join(root, "bench.py")] + sys.argv[2:]
sys.path.insert(0, root)
runpy.run_path(sys.argv[0], run_name="__main__")
