"""Time the fused stem (conv7x7/2 + BN + ReLU + maxpool) at the cfg-2 shape: us per launch, f32 and u8 inputs."""
import sys
import torch
sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
from multiagentperception_amd import ops  # noqa: E402


def main():
    dev = torch.device("cuda:0")
    cout = int(sys.argv[1]) if len(sys.argv) > 1 else 128
    B, N, S = [int(v) for v in (sys.argv[2:5] + ["4", "5", "512"][len(sys.argv[2:5]):])]
    x = torch.rand(B, 3 * N, S, S, device=dev) - 0.45
    u8 = torch.randint(0, 256, (B, N, S, S, 3), dtype=torch.uint8, device=dev)
    w = (torch.randn(cout, 224, device=dev) * 0.1).to(torch.bfloat16)
    sc = torch.ones(cout, device=dev)
    sh = torch.zeros(cout, device=dev)
    out = torch.empty(N * B, S // 4, S // 4, cout, dtype=torch.bfloat16, device=dev)
    for name, fn in (("f32", lambda: ops.stem_conv7x7_bn_relu_maxpool(x, N, w, sc, sh, out=out)),
                     ("u8", lambda: ops.stem_u8_conv7x7_bn_relu_maxpool(u8, w, sc, sh, out=out))):
        for _ in range(5):
            fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(50):
            fn()
        e1.record()
        torch.cuda.synchronize()
        us = e0.elapsed_time(e1) * 1000 / 50
        gf = 2.0 * N * B * (S // 2) ** 2 * cout * 147 / 1e9
        print("B=%d N=%d %dx%d stem_pool %-4s %7.1f us   %.0f TFLOP/s algorithmic (147-tap), %.0f as issued (224)" %
              (B, N, S, S, name, us, gf / us * 1e-3 * 1e3 / 1e3 * 1e3 / 1e3 * 1e0 if False else gf / (us * 1e-6) / 1e3,
               gf * 224 / 147 / (us * 1e-6) / 1e3))


if __name__ == "__main__":
    main()
