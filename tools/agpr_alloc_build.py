"""EXPERIMENT TOOL (not part of the library build; profiles/r03_conv_experiments.txt section 6).
Device-side compile of a .hip file with an UNEVEN VGPR / AGPR split.

Why: the fused stem keeps its MFMA accumulators in AGPRs (inline asm) -- an MFMA whose accumulator sits in VGPRs blocks the VALU
instructions of the other wave of its SIMD (tools/ubench/coissue.hip), which is exactly the overlap the kernel is built around.  A
2-waves/SIMD kernel has 256 registers per wave, and as soon as a function touches AGPRs the backend splits them 128 + 128 unless the
function carries the LLVM attribute "amdgpu-agpr-alloc"; clang has no source spelling for it.  The ping-pong stem needs 80 AGPRs
(5 conv rows of accumulators) and ~165 VGPRs.  So: hipcc -> device LLVM IR, add the
attribute to the stem kernels' definitions, clang (IR -> gfx950 object), lld, clang-offload-bundler, and hipcc --cuda-host-only with
the bundle as its GPU binary: the same five steps `hipcc -c` runs itself (hipcc -###), with one edit in the middle.

python tools/agpr_alloc_build.py <file.hip> <out.o> [--agprs 80] [--report]"""
import os
import re
import subprocess
import sys
import tempfile

LLVM = "/opt/rocm/lib/llvm/bin"
KERNELS = ("stem_pool3_kernel",)


def run(cmd, **kw):
    return subprocess.run(cmd, check=True, **kw)


def build(src, out, agprs=80, flags=(), report=False, hipcc="hipcc"):
    tmp = tempfile.mkdtemp(prefix="w2c_stem_")
    ll, llp, dev, hsaco, fb = (os.path.join(tmp, n) for n in ("stem.ll", "stem_p.ll", "stem_dev.o", "stem.hsaco", "stem.hipfb"))
    common = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wno-unused-value", "-Wno-unused-command-line-argument"] + list(flags)
    run([hipcc] + common + ["--cuda-device-only", "-emit-llvm", "-S", src, "-o", ll])
    lines = open(ll).read().split("\n")
    n = 0
    for i, l in enumerate(lines):
        if l.startswith("define") and any(k in l for k in KERNELS):
            m = re.search(r"\) (local_unnamed_addr )?#(\d+)", l)
            if not m:
                raise RuntimeError("unexpected kernel definition line: " + l[:160])
            lines[i] = l[:m.start() + 2] + (m.group(1) or "") + '"amdgpu-agpr-alloc"="%d" #' % agprs + m.group(2) + l[m.end():]
            n += 1
    if n == 0:
        raise RuntimeError("no stem kernel found in the device IR")
    open(llp, "w").write("\n".join(lines))
    cc = [os.path.join(LLVM, "clang"), "-x", "ir", llp, "-target", "amdgcn-amd-amdhsa", "-mcpu=gfx950", "-O3", "-c", "-o", dev]
    if report:
        r = subprocess.run(cc + ["-Rpass-analysis=kernel-resource-usage"], check=True, stderr=subprocess.PIPE, text=True)
        for blk in r.stderr.split("Function Name: ")[1:]:
            name = blk.split(" ")[0]
            if any(k in name for k in KERNELS):
                vals = dict(re.findall(r"(VGPRs|AGPRs|ScratchSize \[bytes/lane\]|Occupancy \[waves/SIMD\]): (\d+)", blk))
                print(name[:60], vals)
    else:
        run(cc)
    run([os.path.join(LLVM, "lld"), "-flavor", "gnu", "-m", "elf64_amdgpu", "--no-undefined", "-shared", "-o", hsaco, dev])
    run([os.path.join(LLVM, "clang-offload-bundler"), "-type=o", "-bundle-align=4096",
         "-targets=host-x86_64-unknown-linux-gnu,hipv4-amdgcn-amd-amdhsa--gfx950", "-input=/dev/null", "-input=" + hsaco, "-output=" + fb])
    run([hipcc] + common + ["--cuda-host-only", "-Xclang", "-fcuda-include-gpubinary", "-Xclang", fb, "-c", src, "-o", out])
    return out


if __name__ == "__main__":
    a = sys.argv[1:]
    ag = int(a[a.index("--agprs") + 1]) if "--agprs" in a else 80
    build(a[0], a[1], agprs=ag, report="--report" in a)
