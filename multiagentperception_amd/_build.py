"""Build libw2c_hip.so (gfx950) in-tree with hipcc.  ``python -m multiagentperception_amd._build``."""
import os
import shutil
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIBDIR = os.path.join(HERE, "lib")
LIB = os.path.join(LIBDIR, "libw2c_hip.so")
SOURCES = ["conv_igemm.hip", "conv_block.hip", "conv_wgrad.hip", "bn_train.hip", "stem.hip", "stem_train.hip", "comm_attn.hip", "upsample.hip", "loss.hip"]


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
    os.replace(LIB + ".tmp", LIB)
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
