"""Phase cycles of the layer1 register-resident-weights conv (variant 50), from its p.dbg accumulators."""
import os
import sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from multiagentperception_amd import ops, _native  # noqa: E402


def main():
    dev = torch.device("cuda:0")
    variant = int(sys.argv[1]) if len(sys.argv) > 1 else 50
    C, H = (64, 128) if variant in (50, 52, 54) else (128, 64)
    M, W, G = 20, H, 2
    x = torch.randn(M, H, W, G * C, device=dev).to(torch.bfloat16)
    w = (torch.randn(G, C, 9 * C, device=dev) * 0.05).to(torch.bfloat16)
    sc = torch.ones(G * C, device=dev)
    sh = torch.zeros(G * C, device=dev)
    r = torch.randn(M, H, W, G * C, device=dev).to(torch.bfloat16)
    if variant == 54:
        w = ops.pack_wfrag_device(w, C)
    for res in (r, None):
        for _ in range(3):
            ops.conv_igemm(x, 0, C, w, C, 3, 1, G, sc, sh, residual=res, variant=variant)
        buf = torch.zeros(8192 * 8, dtype=torch.int64, device=dev)
        _native.lib().w2c_debug_conv_timeline(buf.data_ptr())
        ops.conv_igemm(x, 0, C, w, C, 3, 1, G, sc, sh, residual=res, variant=variant)
        torch.cuda.synchronize()
        full = buf.view(-1, 8).cpu()
        full = full[full[:, 7] == 1]
        t0 = full[:, 4].min()
        wl = (full[:, 4:7] - t0).double() / 100.0     # us
        print("   wall (us): kernel start spread %.1f | prologue mean %.1f | tile loop mean %.1f max %.1f | last end %.1f" % (
            wl[:, 0].max(), (wl[:, 1] - wl[:, 0]).mean(), (wl[:, 2] - wl[:, 1]).mean(), (wl[:, 2] - wl[:, 1]).max(), wl[:, 2].max()))
        b = full[:, :4].double()
        if variant == 54:       # two records (waves 0 and 3) per workgroup, 8 x 16 tiles, phases: vmcnt wait | barrier | MFMA loop | epilogue
            tiles = M * (H // 8) * (W // 16) * G / (b.shape[0] / 2)
            print("residual=%s: %d workgroups, %.1f tiles each; cycles per tile (waves 0, 3): vmcnt wait %.0f | barrier %.0f | MFMA loop %.0f "
                  "(pure 2304) | epilogue %.0f | total %.0f" % (res is not None, b.shape[0] // 2, tiles, *(b.mean(0) / tiles).tolist(), b.sum(1).mean() / tiles))
            continue
        tiles = M * (H // 4) * (W // 16) * G / (b.shape[0] * (4 if variant in (50, 52) else 1))
        print("residual=%s: %d workgroups, %.1f tiles/wave; cycles per tile: DMA issue %.0f | MFMA loop %.0f (pure 4608) | "
              "vmcnt wait %.0f | epilogue %.0f | total %.0f" % (res is not None, b.shape[0], tiles, *(b.mean(0) / tiles).tolist(),
                                                                 b.sum(1).mean() / tiles))


if __name__ == "__main__":
    main()
