"""Build libw2c_hip.so (gfx950) in-tree with hipcc.  ``python -m multiagentperception_amd._build``."""
import os
import shutil
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIBDIR = os.path.join(HERE, "lib")
LIB = os.path.join(LIBDIR, "libw2c_hip.so")
SOURCES = ["conv_igemm.hip", "conv_wgrad.hip", "bn_train.hip", "stem.hip", "stem_train.hip", "comm_attn.hip", "upsample.hip", "loss.hip"]


def _hipcc():
    for cand in (os.environ.get("HIPCC"), shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError("hipcc not found (set HIPCC=/path/to/hipcc)")


def needs_build():
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    deps = [os.path.join(CSRC, f) for f in os.listdir(CSRC)] + [os.path.join(os.path.dirname(HERE), "include", "w2c_hip.h")]
    return any(os.path.getmtime(d) > t for d in deps)


def build(force=False, verbose=False):
    if not force and not needs_build():
        return LIB
    os.makedirs(LIBDIR, exist_ok=True)
    cmd = [_hipcc(), "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared",
           "-Wno-unused-value",
           # no SLP vectorisation => no packed-f32 VALU (v_pk_fma_f32 / v_pk_mul_f32 / v_pk_add_f32) in any kernel.  Found with several
           # engines in flight (tools/inflight_lin.py): the LOW half of v_pk_fma_f32 results in the wide-K head kernel came out wrong
           # (1e-3 .. 7e-2, a few elements, even rows = low halves only) whenever its waves shared a CU with another kernel's MFMA
           # waves -- never alone, never with the scalar form.  The same IEEE results either way; packed f32 beside MFMAs is slower
           # anyway (cdna_hip_programming.md, co-issue table).
           "-fno-slp-vectorize"] + os.environ.get("W2C_EXTRA_HIPCC_FLAGS", "").split() + [os.path.join(CSRC, s) for s in SOURCES] + ["-o", LIB + ".tmp"]
    if verbose:
        print(" ".join(cmd), file=sys.stderr)
    subprocess.run(cmd, check=True)
    check_no_packed_f32(LIB + ".tmp")
    os.replace(LIB + ".tmp", LIB)
    return LIB


# The f32 head / graph kernels (comm_attn.hip) were validated -- first-forward stress in fresh processes, several engines in flight
# -- ONLY in the scalar-f32 form (see the -fno-slp-vectorize note above); -fno-slp-vectorize does not stop other passes or
# W2C_EXTRA_HIPCC_FLAGS from emitting packed f32, so the built code objects are disassembled and the build FAILS if one of these
# kernels contains a v_pk_{fma,mul,add}_f32.  (The conv kernels' epilogues do contain v_pk_fma_f32 -- from explicit
# float4 arithmetic -- and are bit-stable beside MFMA waves in every torch.equal test, so the hazard is specific to these kernels'
# packed form, not a blanket rule; tools/ubench/pkfma_mfma.hip is the stand-alone probe.)
PACKED_F32_FREE = ("linear_widek_kernel", "linear_kernel", "head_tail", "head_fc0_mfma_kernel", "comm_graph_kernel", "graph_fuse_kernel", "fuse_kernel",
                   "key_project_kernel")


def check_no_packed_f32(lib, kernels=PACKED_F32_FREE):
    import re
    import tempfile
    objdump = os.path.join(os.path.dirname(os.path.dirname(os.path.realpath(_hipcc()))), "lib", "llvm", "bin", "llvm-objdump")
    if not os.path.exists(objdump):
        objdump = "/opt/rocm/lib/llvm/bin/llvm-objdump"
    if not os.path.exists(objdump):
        print("check_no_packed_f32: llvm-objdump not found, check skipped", file=sys.stderr)
        return
    tmp = tempfile.mkdtemp(prefix="w2c_objdump_")
    try:
        local = os.path.join(tmp, "lib.so")
        shutil.copy(lib, local)
        subprocess.run([objdump, "--offloading", local], check=True, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, cwd=tmp)
        bad, seen = {}, set()
        for f in sorted(os.listdir(tmp)):
            if "amdgcn" not in f:
                continue
            dis = subprocess.run([objdump, "-d", os.path.join(tmp, f)], check=True, capture_output=True, text=True).stdout
            name = None
            for line in dis.splitlines():
                m = re.match(r"^[0-9a-f]+ <(.*)>:", line)
                if m:
                    name = m.group(1) if any(k in m.group(1) for k in kernels) else None
                    if name:
                        seen.add(name)
                elif name and re.search(r"\bv_pk_(fma|mul|add)_f32\b", line):
                    bad[name] = bad.get(name, 0) + 1
        if not seen:
            raise RuntimeError("check_no_packed_f32: none of the head / graph kernels found in the built library")
        if bad:
            raise RuntimeError("packed-f32 VALU in the f32 head / graph kernels (validated in scalar form only): %r" % (bad,))
    finally:
        shutil.rmtree(tmp, ignore_errors=True)


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
