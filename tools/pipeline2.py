"""Throughput experiment: F forwards in flight (F engines with their own buffers / captured graphs, one stream each, launched
round-robin) against the one-forward-at-a-time headline.   python tools/pipeline2.py [--eager] [F ...]"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from multiagentperception_amd import synth as filler  # noqa: E402
from ptsemseg.models import get_model  # noqa: E402


def main():
    eager = "--eager" in sys.argv
    fs = [int(a) for a in sys.argv[1:] if not a.startswith("--")] or [1, 2, 3]
    dev = torch.device("cuda:0")
    preset = bench.PRESETS["cfg2"]
    B, n, S = preset["batch"], preset["agents"], preset["size"]
    def make():
        m = get_model(bench.build_cfg(preset["arch"], n, S, preset["query"]), 11)
        filler.apply_to_module(m)                  # deterministic filler: every copy has the same weights
        return m.to(dev).eval()
    x = torch.from_numpy(filler.synthetic_frames(B, n, S, S, 1234 + 2)).to(dev)
    steps = 300
    for F in fs:
        models = [make() for _ in range(F)]
        from multiagentperception_amd import ops
        streams = ops.caller_streams(dev, F) if "--pick" in sys.argv else [torch.cuda.Stream(dev) for _ in range(F)]
        for m in models:
            m.use_hip_graph = not eager
        for i in range(3 * F):
            with torch.cuda.stream(streams[i % F]):
                out = models[i % F](x, training=False, MO_flag=True, inference="softmax")
        torch.cuda.synchronize(dev)
        t0 = time.perf_counter()
        for i in range(steps):
            with torch.cuda.stream(streams[i % F]):
                out = models[i % F](x, training=False, MO_flag=True, inference="softmax")
        torch.cuda.synchronize(dev)
        el = time.perf_counter() - t0
        print("forwards in flight %d: %.4f ms / forward, %.0f agent-images/s" % (F, 1e3 * el / steps, B * n * steps / el))
        del models


if __name__ == "__main__":
    main()
