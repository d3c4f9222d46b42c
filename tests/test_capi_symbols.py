"""CPU: the C-ABI shared library loads and exports every symbol include/w2c_hip.h declares
(no compute calls -- there is no GPU here)."""
import ctypes
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, "include", "w2c_hip.h")


def declared_functions():
    text = open(HEADER).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(w2c_[a-z0-9_]+)\s*\(", text)))


@pytest.fixture(scope="module")
def built_lib():
    from multiagentperception_amd import _build
    return _build.build()          # no-op when the in-tree .so is current; hipcc cross-compiles gfx950 without a GPU


def test_header_declares_the_expected_entry_points():
    names = declared_functions()
    for must in ("w2c_stem_conv7x7_bn_relu", "w2c_maxpool3x3s2", "w2c_conv_igemm_bf16", "w2c_linear_f32",
                 "w2c_comm_graph", "w2c_fuse_values", "w2c_upsample_bilinear32", "w2c_status_string"):
        assert must in names


def test_library_exports_every_declared_symbol(built_lib):
    import torch  # noqa: F401  (load torch's HIP runtime first, as _native.lib() does)
    handle = ctypes.CDLL(built_lib)
    for name in declared_functions():
        assert hasattr(handle, name), "libw2c_hip.so does not export %s" % name


def test_ctypes_signatures_cover_the_header(built_lib):
    from multiagentperception_amd import _native
    assert sorted(_native.SIGNATURES) == declared_functions()
    lib = _native.lib()
    assert lib.w2c_version() >= 1
    assert lib.w2c_status_string(0) == b"ok"
    assert b"invalid" in lib.w2c_status_string(-1)


def test_header_has_no_torch_types_and_is_plain_c():
    text = open(HEADER).read()
    assert "torch" not in text.lower().replace("pytorch-rocm", "") or "no torch types" in text.lower()
    assert 'extern "C"' in text
    # compiles as C
    import subprocess
    import tempfile
    with tempfile.TemporaryDirectory() as d:
        src = os.path.join(d, "t.c")
        open(src, "w").write('#include "w2c_hip.h"\nint main(void){return w2c_version!=0?0:1;}\n')
        subprocess.run(["gcc", "-std=c99", "-fsyntax-only", "-I", os.path.join(ROOT, "include"), src], check=True)
