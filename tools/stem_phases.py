"""Per-phase shader cycles of the fused stem (debug build: hipcc -DW2C_STEM_TIMING -> tools/libw2c_stem_phase.so).
Sums over workgroups of thread 0's clock64 deltas, printed per workgroup-step."""
import ctypes
import os
import subprocess
import sys
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SO = os.environ.get("W2C_STEM_SO", os.path.join(ROOT, "tools", "libw2c_stem_phase.so"))


def build():
    src = os.path.join(ROOT, "multiagentperception_amd", "csrc", "stem.hip")
    subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-DW2C_STEM_TIMING"]
                          + os.environ.get("W2C_STEM_DEFS", "").split()
                          + ["-x", "hip", src, os.path.join(ROOT, "tools", "ubench", "options_stub.cpp"), "-o", SO])


def main():
    if "--build" in sys.argv:
        build()
        return
    lib = ctypes.CDLL(SO)
    dev = torch.device("cuda:0")
    B, N, S, cout = 4, 5, 512, 128
    x = torch.rand(B, 3 * N, S, S, device=dev) - 0.45
    w = (torch.randn(cout, 224, device=dev) * 0.1).to(torch.bfloat16)
    sc = torch.ones(cout, device=dev)
    sh = torch.zeros(cout, device=dev)
    out = torch.empty(N * B, S // 4, S // 4, cout, dtype=torch.bfloat16, device=dev)
    vp = ctypes.c_void_p
    lib.w2c_stem_conv7x7_bn_relu_maxpool.argtypes = [vp, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, vp, vp, vp,
                                                     ctypes.c_int, vp, vp]
    if "--time" in sys.argv:           # plain launch timing of whatever build W2C_STEM_SO names (ablation builds)
        call = lambda: lib.w2c_stem_conv7x7_bn_relu_maxpool(x.data_ptr(), B, N, S, S, w.data_ptr(), sc.data_ptr(), sh.data_ptr(), cout,
                                                            out.data_ptr(), torch.cuda.current_stream().cuda_stream)
        for _ in range(5):
            call()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(50):
            call()
        e1.record()
        torch.cuda.synchronize()
        print("%s: %.1f us / launch" % (os.path.basename(SO), e0.elapsed_time(e1) * 20))
        return
    buf = (ctypes.c_ulonglong * 8)()
    lib.w2c_debug_stem_phases(None, 1)
    reps = 5
    for _ in range(reps):
        rc = lib.w2c_stem_conv7x7_bn_relu_maxpool(x.data_ptr(), B, N, S, S, w.data_ptr(), sc.data_ptr(), sh.data_ptr(), cout,
                                                  out.data_ptr(), torch.cuda.current_stream().cuda_stream)
        assert rc == 0
    torch.cuda.synchronize()
    lib.w2c_debug_stem_phases(buf, 0)
    steps = reps * (S // 2 // 8) * (N * B) * (S // 2 // 32)
    names = ["store_patch", "barrier A", "issue loads", "MFMA loop", "BN+stage", "barrier B", "pool+store", "barrier C+carry"]
    if os.environ.get("W2C_STEM_FORM") == "2":
        names = ["store_patch", "barrier", "issue loads", "MFMA rows 0-4", "BN+vmax+pack A", "hmax+stage+store A", "MFMA rows 5-8",
                 "BN..store B"]
    if os.environ.get("W2C_STEM_FORM", "3") in ("0", "3"):       # ping-pong form: group A's wave 0; its steps = half of all (+ warm-ups)
        names = ["M1 (rows 0-4)", "barrier", "V1", "barrier", "M2 (rows 5-8)", "barrier", "V2 + patch store", "barrier"]
        steps = steps / 2
    tot = 0
    for n, v in zip(names, buf):
        print("%-16s %8.0f cycles / workgroup-step" % (n, v / steps))
        tot += v / steps
    print("%-16s %8.0f   (pure MFMA per SIMD: 9 rows x 4 ct x 14 x 32 / 4 = 4032)" % ("total", tot))


if __name__ == "__main__":
    main()
