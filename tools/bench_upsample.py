"""Microbenchmark of the output kernels at cfg 2's shape (M = 20, 16 x 16 low-resolution logits, 11 classes): K9 (f32 logits),
K9 + argmax (u8 labels), K9 + argmax + confusion matrix (u8 / i64 ground truth).  Graph replay of 20 launches each."""
import os
import sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from multiagentperception_amd import ops  # noqa: E402

dev = torch.device("cuda:0")
M, h, w, n = 20, 16, 16, 11
low = torch.randn(M, h, w, 32, device=dev)
gt8 = torch.randint(0, 11, (M, 32 * h, 32 * w), dtype=torch.uint8, device=dev)
# segment-like labels (what a real evaluator sees): 64 x 64 blocks of one class
seg = torch.randint(0, 11, (M, 8, 8), device=dev).repeat_interleave(64, 1).repeat_interleave(64, 2).to(torch.uint8)
hist = torch.zeros(121, dtype=torch.int64, device=dev)
out = torch.empty(M, n, 32 * h, 32 * w, device=dev)
lab = torch.empty(M, 32 * h, 32 * w, dtype=torch.uint8, device=dev)


def timed(name, fn, nbytes):
    for _ in range(3):
        fn()
    g = torch.cuda.CUDAGraph()
    st = torch.cuda.Stream()
    st.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(st):
        with torch.cuda.graph(g, stream=st):
            for _ in range(20):
                fn()
        g.replay()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        g.replay()
        e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1000 / 20
    print("%-46s %7.1f us   %6.2f TB/s of output" % (name, us, nbytes / us / 1e6))


timed("upsample_bilinear32 (f32 logits, 231 MB)", lambda: ops.upsample_bilinear32(low, n, out=out), out.numel() * 4)
timed("upsample32_argmax (u8 labels, 5 MB)", lambda: ops.upsample32_argmax(low, n, out=lab), lab.numel())
timed("  + confusion, u8 noise labels", lambda: ops.upsample32_argmax_confusion(low, n, gt8, hist), lab.numel())
timed("  + confusion, u8 segment labels", lambda: ops.upsample32_argmax_confusion(low, n, seg, hist), lab.numel())
ws = ops.confusion_workspace(low.device, n)
timed("  + confusion, u8 noise labels, two-level flush", lambda: ops.upsample32_argmax_confusion(low, n, gt8, hist, ws=ws), lab.numel())
timed("  + confusion, u8 segment labels, two-level flush", lambda: ops.upsample32_argmax_confusion(low, n, seg, hist, ws=ws), lab.numel())
timed("  + confusion, i64 segment labels", lambda: ops.upsample32_argmax_confusion(low, n, seg.long(), hist), lab.numel())
