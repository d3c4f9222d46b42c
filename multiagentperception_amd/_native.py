"""ctypes binding of libw2c_hip.so (C ABI declared in include/w2c_hip.h).

The library is REQUIRED: there is no CPU or PyTorch fallback.  Importing this
module never fails (so host-only tooling can import the package), but the first
use of ``lib()`` raises if the shared object is missing or does not export every
declared symbol.
"""
import ctypes
import os

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(HERE, "lib", "libw2c_hip.so")

_c = ctypes
_vp, _i, _f, _ll = _c.c_void_p, _c.c_int, _c.c_float, _c.c_longlong

# symbol -> argtypes (restype is int unless noted); mirrors include/w2c_hip.h
SIGNATURES = {
    "w2c_version": [],
    "w2c_status_string": [_i],
    "w2c_last_error_string": [],
    "w2c_device_arch": [_c.c_char_p, _i],
    "w2c_stem_conv7x7_bn_relu": [_vp, _i, _i, _i, _i, _vp, _vp, _vp, _i, _vp, _vp],
    "w2c_stem_conv7x7_bn_relu_maxpool": [_vp, _i, _i, _i, _i, _vp, _vp, _vp, _i, _vp, _vp],
    "w2c_stem_u8_conv7x7_bn_relu_maxpool": [_vp, _c.c_double, _c.c_double, _c.c_double, _i, _i, _i, _i, _vp, _vp, _vp, _i,
                                            _vp, _vp],
    "w2c_maxpool3x3s2": [_vp, _i, _i, _i, _i, _vp, _vp],
    "w2c_conv_igemm_bf16": [_vp, _i, _i, _i, _i, _i, _vp, _i, _i, _i, _i, _vp, _vp, _vp, _i, _vp, _i, _i, _vp, _ll, _vp],
    "w2c_conv_igemm_bf16_variant": [_vp, _i, _i, _i, _i, _i, _vp, _i, _i, _i, _i, _vp, _vp, _vp, _i, _vp, _i, _i, _vp, _i,
                                    _ll, _vp],
    "w2c_conv_igemm_bf16_splitk": [_vp, _i, _i, _i, _i, _i, _vp, _i, _i, _i, _i, _vp, _vp, _vp, _i, _vp, _i, _i, _vp, _i,
                                   _vp, _ll, _ll, _vp],
    "w2c_conv_splitk_workspace_bytes": [_i, _i, _i, _i, _i, _i, _i, _i, _i],
    "w2c_pack_wfrag_bf16": [_vp, _vp, _i, _i, _i, _vp],
    "w2c_conv3x3_wreg_supported": [_i, _i, _i, _i],
    "w2c_conv3x3_wreg_bf16": [_vp, _i, _i, _i, _i, _i, _vp, _i, _i, _vp, _vp, _vp, _i, _vp, _i, _ll, _i, _vp],
    "w2c_conv3x3_wreg_f32out": [_vp, _i, _i, _i, _i, _i, _vp, _i, _i, _vp, _vp, _i, _vp, _i, _ll, _vp],
    "w2c_debug_conv_timeline": [_vp],
    "w2c_debug_stamp": [_vp, _vp],
    "w2c_debug_install_crash_backtrace": [_i],
    "w2c_debug_conv_span": [_vp],
    "w2c_conv_igemm_fp8": [_vp, _i, _i, _i, _i, _i, _i, _vp, _i, _i, _i, _i, _vp, _vp, _vp, _i, _vp, _i, _ll, _vp, _i, _f,
                           _vp, _i, _vp],
    "w2c_conv_s2_block": [_vp, _i, _i, _i, _i, _i, _i, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _vp, _i, _vp, _i, _f, _vp, _i,
                          _vp, _i, _vp],
    "w2c_conv_s2_block_wreg_supported": [_i, _i, _i, _i],
    "w2c_conv_s2_block_wreg": [_vp, _i, _i, _i, _i, _i, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _vp, _i, _vp, _i, _i, _vp],
    "w2c_conv_s2_front_c64_supported": [_i, _i, _i, _i],
    "w2c_conv_s2_front_c64": [_vp, _i, _i, _i, _i, _vp, _vp, _vp, _vp, _vp, _vp, _i, _vp, _i, _ll, _vp, _i, _ll, _vp],
    "w2c_set_option": [_c.c_char_p, _i],
    "w2c_get_option": [_c.c_char_p],
    "w2c_conv_wgrad_workspace_bytes": [_i, _i, _i, _i, _i, _i, _i, _i],
    "w2c_conv_wgrad_bf16": [_vp, _i, _i, _i, _i, _i, _vp, _i, _i, _i, _i, _i, _vp, _vp, _ll, _vp],
    "w2c_conv_wgrad_bf16_oihw": [_vp, _i, _i, _i, _i, _i, _vp, _i, _i, _i, _i, _vp, _vp, _ll, _vp],
    "w2c_pack_conv_weights_bf16": [_vp, _i, _i, _i, _i, _vp, _vp],
    "w2c_pack_conv_weights_bf16_both": [_vp, _i, _i, _i, _vp, _vp, _vp],
    "w2c_zero_insert2_bf16": [_vp, _i, _i, _i, _i, _vp, _i, _i, _vp],
    "w2c_bn_workspace_bytes": [_ll, _i],
    "w2c_bn_train_forward": [_vp, _ll, _i, _vp, _vp, _vp, _vp, _vp, _f, _f, _vp, _i, _vp, _vp, _vp, _vp, _vp, _ll, _vp],
    "w2c_bn_train_backward": [_vp, _vp, _vp, _ll, _i, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _ll, _vp],
    "w2c_bn_train_sums": [_i, _vp, _vp, _vp, _vp, _vp, _ll, _i, _vp, _vp, _ll, _vp],
    "w2c_bn_train_forward_sums": [_vp, _ll, _i, _vp, _c.c_double, _vp, _vp, _vp, _vp, _vp, _f, _f, _vp, _i, _vp, _vp, _vp, _vp, _vp],
    "w2c_bn_train_backward_sums": [_vp, _vp, _vp, _ll, _i, _vp, _vp, _vp, _vp, _vp, _c.c_double, _vp, _vp, _vp, _vp, _vp, _vp],
    "w2c_maxpool3x3s2_train_forward": [_vp, _i, _i, _i, _i, _vp, _vp, _vp],
    "w2c_maxpool3x3s2_train_backward": [_vp, _vp, _i, _i, _i, _i, _vp, _vp],
    "w2c_debug_mx_mfma": [_vp, _vp, _vp, _vp],
    "w2c_debug_fp8_pack": [_vp, _vp, _i, _vp],
    "w2c_linear_f32": [_vp, _i, _i, _i, _i, _vp, _vp, _i, _i, _vp, _vp],
    "w2c_head_tail_f32": [_vp, _i, _i, _i, _vp, _vp, _i, _vp, _vp, _i, _vp, _vp],
    "w2c_head_tail2_f32": [_vp, _i, _i, _i, _i, _i, _vp, _vp, _vp, _vp, _i, _vp, _i, _vp, _vp, _vp, _vp, _i, _vp, _vp],
    "w2c_comm_graph": [_vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _f, _f, _i, _i, _vp, _vp, _vp, _vp, _vp, _vp],
    "w2c_comm_graph_projected": [_vp, _vp, _i, _i, _i, _i, _i, _f, _f, _i, _i, _vp, _vp, _vp, _vp, _vp],
    "w2c_comm_graph_fuse": [_vp, _vp, _i, _i, _i, _i, _i, _f, _f, _i, _i, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _vp, _i, _vp],
    "w2c_head_fc0_mfma_f32": [_vp, _i, _i, _i, _vp, _i, _i, _vp, _vp],
    "w2c_head_tail2p_f32": [_vp, _i, _ll, _vp, _i, _i, _i, _i, _i, _vp, _vp, _vp, _vp, _i, _vp, _i, _vp, _vp, _vp, _vp, _i, _vp, _vp],
    "w2c_comm_graph_fuse_u": [_vp, _vp, _i, _i, _i, _i, _i, _f, _f, _i, _i, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _vp, _i, _vp, _vp, _i, _vp, _ll, _ll, _vp],
    "w2c_fuse_values": [_vp, _i, _vp, _i, _i, _i, _i, _i, _i, _i, _vp, _i, _vp],
    "w2c_set_slots": [_vp, _i, _vp, _vp],
    "w2c_copy_to_slot": [_vp, _ll, _vp, _vp],
    "w2c_upsample_bilinear32": [_vp, _i, _i, _i, _i, _i, _vp, _vp],
    "w2c_upsample32_argmax": [_vp, _i, _i, _i, _i, _i, _vp, _vp],
    "w2c_upsample_bilinear32_backward": [_vp, _i, _i, _i, _i, _vp, _vp],
    "w2c_stem_conv7x7_train_bf16": [_vp, _i, _i, _i, _vp, _vp, _vp],
    "w2c_stem_wgrad_workspace_bytes": [_i, _i, _i],
    "w2c_stem_wgrad_bf16": [_vp, _i, _i, _i, _vp, _i, _vp, _vp, _ll, _vp],
    "w2c_cross_entropy2d_workspace_bytes": [_ll],
    "w2c_cross_entropy2d_forward": [_vp, _vp, _vp, _i, _i, _ll, _i, _i, _vp, _vp, _vp, _vp, _ll, _vp],
    "w2c_cross_entropy2d_backward": [_vp, _vp, _vp, _vp, _i, _i, _ll, _i, _vp, _vp, _vp, _vp, _vp],
    "w2c_upsample32_argmax_confusion": [_vp, _i, _i, _i, _i, _i, _vp, _i, _vp, _vp, _vp, _i, _vp],
    "w2c_confusion_matrix": [_vp, _i, _vp, _ll, _i, _vp, _vp],
    "w2c_nchw_f32_to_nhwc_bf16": [_vp, _i, _i, _i, _i, _vp, _i, _vp],
    "w2c_nhwc_bf16_to_nchw_f32": [_vp, _i, _i, _i, _i, _i, _vp, _vp],
}

_lib = None


class W2CError(RuntimeError):
    pass


def lib():
    global _lib
    if _lib is None:
        # torch wheels bundle their own libamdhip64.so; load torch FIRST so that our library's
        # NEEDED libamdhip64.so.7 resolves to that already-loaded runtime.  Loading ours first
        # pulls /opt/rocm's copy in as a second HIP runtime and every launch then fails with
        # hipErrorNoDevice.
        import torch  # noqa: F401
        if not os.path.exists(LIB_PATH):
            raise W2CError(
                "libw2c_hip.so not found at %s -- build it with "
                "`python -m multiagentperception_amd._build` (needs hipcc, gfx950). "
                "There is no CPU fallback for the When2com forward path." % LIB_PATH)
        handle = ctypes.CDLL(LIB_PATH)
        for name, argtypes in SIGNATURES.items():
            try:
                fn = getattr(handle, name)
            except AttributeError:
                raise W2CError("libw2c_hip.so does not export %s (stale build?)" % name)
            fn.argtypes = argtypes
            fn.restype = (_c.c_char_p if name in ("w2c_status_string", "w2c_last_error_string") else
                          _c.c_longlong if name in ("w2c_conv_splitk_workspace_bytes", "w2c_conv_wgrad_workspace_bytes", "w2c_bn_workspace_bytes",
                                                     "w2c_cross_entropy2d_workspace_bytes", "w2c_stem_wgrad_workspace_bytes") else _i)
        _lib = handle
    return _lib


def check(code, what):
    if code != 0:
        msg = lib().w2c_status_string(code)
        detail = lib().w2c_last_error_string() if code == -2 else b""
        raise W2CError("%s failed: %s (code %d) %s" % (what, msg.decode() if msg else "?", code,
                                                      detail.decode() if detail else ""))


def set_option(name, value):
    """change a debug / A-B switch of the library at run time (include/w2c_hip.h w2c_set_option); returns the old value"""
    old = lib().w2c_get_option(name.encode())
    check(lib().w2c_set_option(name.encode(), int(value)), "w2c_set_option(%s)" % name)
    return old
