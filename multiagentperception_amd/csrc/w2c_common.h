// Shared device/host helpers for libw2c_hip.so (gfx950 only; no dual CUDA/HIP paths).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <atomic>

#include "../../include/w2c_hip.h"

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8_t;
typedef __attribute__((ext_vector_type(16))) float f32x16_t;
typedef __attribute__((ext_vector_type(4))) float f32x4_t;
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4_t;
typedef __attribute__((ext_vector_type(2))) short s16x2_t;

// global / LDS address-space pointer casts for the LDS-DMA builtin
#define W2C_GPTR(p) ((const __attribute__((address_space(1))) void*)(p))
#define W2C_LPTR(p) ((__attribute__((address_space(3))) void*)(p))

// f32 -> bf16 round-to-nearest-even (finite inputs; activations never NaN by construction,
// NaN still maps to a NaN pattern because the mantissa carry cannot clear the exponent).
// f32 -> bf16, round-to-nearest-even: gfx950's v_cvt_pk_bf16_f32 (one instruction per PAIR; the integer
// add-and-shift idiom costs ~9 VALU per pair and made the conv / stem epilogues VALU-bound).
typedef float w2c_f32x2_t __attribute__((ext_vector_type(2)));
typedef __bf16 w2c_bf16x2_t __attribute__((ext_vector_type(2)));
__device__ __forceinline__ uint16_t f32_to_bf16(float f) { return __builtin_bit_cast(uint16_t, (__bf16)f); }
__device__ __forceinline__ float bf16_to_f32(uint16_t h) { return __uint_as_float(((uint32_t)h) << 16); }
__device__ __forceinline__ uint32_t pack_bf16x2(float lo, float hi) {
    const w2c_f32x2_t v = {lo, hi};
    return __builtin_bit_cast(uint32_t, __builtin_convertvector(v, w2c_bf16x2_t));
}

// fp8 (OCP e4m3fn, gfx950's native format) pack of four f32: v_cvt_pk_fp8_f32 rounds to nearest even; inputs are clamped
// to the format's finite range first (+-448) so an out-of-range activation saturates instead of becoming NaN.
typedef __attribute__((ext_vector_type(8))) int i32x8_t;
__device__ __forceinline__ uint32_t pack_fp8x4(float a, float b, float c, float d) {
    a = __builtin_fminf(__builtin_fmaxf(a, -448.f), 448.f);
    b = __builtin_fminf(__builtin_fmaxf(b, -448.f), 448.f);
    c = __builtin_fminf(__builtin_fmaxf(c, -448.f), 448.f);
    d = __builtin_fminf(__builtin_fmaxf(d, -448.f), 448.f);
    int w = 0;
    w = __builtin_amdgcn_cvt_pk_fp8_f32(a, b, w, false);
    w = __builtin_amdgcn_cvt_pk_fp8_f32(c, d, w, true);
    return (uint32_t)w;
}
__device__ __forceinline__ i32x8_t cat_i32x8(u32x4_t lo, u32x4_t hi) {
    return i32x8_t{(int)lo[0], (int)lo[1], (int)lo[2], (int)lo[3], (int)hi[0], (int)hi[1], (int)hi[2], (int)hi[3]};
}

// Indirect operands (include/w2c_hip.h "indirect operands"): a pointer argument with bit 0 set is the address (| 1) of a device-resident
// 8-byte slot that holds the real pointer; the kernel reads the slot when it starts.  A captured HIP graph can then run on a different
// caller-owned tensor at every replay (w2c_set_slots fills the slots in stream order just before the replay) without copying the
// tensor into a static buffer and without re-capturing.  Wave-uniform: one scalar load + branch.
#define W2C_UNTAG(p) (reinterpret_cast<uintptr_t>(p) & ~static_cast<uintptr_t>(1))
template <typename T>
__device__ __forceinline__ T* w2c_resolve(T* p) {
    const uintptr_t v = reinterpret_cast<uintptr_t>(p);
    if (v & 1) return *reinterpret_cast<T* const*>(v & ~static_cast<uintptr_t>(1));
    return p;
}

// Last HIP error text of the calling thread (for w2c_last_error_string()).
inline char* w2c_errbuf() {
    static thread_local char buf[256] = {0};
    return buf;
}
// Clear any stale (non-sticky) HIP error left by an earlier caller in this thread, so that the
// status we return belongs to OUR launch.
static inline void w2c_clear_error() { (void)hipGetLastError(); }
static inline int w2c_launch_status() {
    hipError_t e = hipGetLastError();
    if (e == hipSuccess) return W2C_OK;
    char* b = w2c_errbuf();
    const char* n = hipGetErrorName(e);
    const char* m = hipGetErrorString(e);
    int i = 0;
    for (const char* p = n; p && *p && i < 100; ++p) b[i++] = *p;
    b[i++] = ':'; b[i++] = ' ';
    for (const char* p = m; p && *p && i < 250; ++p) b[i++] = *p;
    b[i] = 0;
    return W2C_E_LAUNCH;
}

// XCD-aware, bijective remap of a linear workgroup id: hardware places block b on XCD b % 8
// (MI355X_MICROARCH.md "Workgroup dispatch"), so give every XCD a CONTIGUOUS chunk of tile ids;
// neighbouring tiles (which share an activation row-panel / a weight column-panel) then hit the
// same private L2.  Speed only -- correctness never depends on placement.
__device__ __forceinline__ int xcd_remap(int bid, int nwg) {
    const int q = nwg >> 3, r = nwg & 7;
    const int xcd = bid & 7, loc = bid >> 3;
    const int start = xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
    return start + loc;
}

// Debug / A-B switches.  The launch paths never call getenv(): the table is seeded from the environment ONCE, when the library is
// loaded, and changed at run time through the C ABI (w2c_set_option; tests and tools/ use that).  Defined in conv_igemm.hip.
enum W2COption {
    W2C_OPT_XCD2D = 0,        // 0 off | 1 auto (default) | 2 force: XCD-aware 2-D tile placement of two-group patch launches
    W2C_OPT_NO_S2PATCH,       // 1: stride-2 block fronts on the generic kernels instead of the polyphase patch kernel
    W2C_OPT_STEM_WGS,         // > 0: workgroup count of the ping-pong stem (tests: odd run lengths)
    W2C_OPT_STEM_FORM,        // 0 auto | 1 | 2 | 3: stem kernel form
    W2C_OPT_STEM_BAND,        // 8 (default) | 4
    W2C_OPT_STEM_WAVES,       // 8 (default) | 12
    W2C_OPT_WGRAD_PATCH,      // 1 (default) | 0: halo-patch weight-gradient kernel
    W2C_OPT_INWG_SPLITK,      // 1 (default) | 0: split-K convs with <= 12 splits as ONE launch (splits = waves, partials in LDS)
    W2C_OPT_WREG_MINCIN,      // 256 (default): 3x3 / s1 convs with Cin >= this go to the weights-to-registers kernel; 0 = never
    W2C_OPT_WREG_FORM,        // 0 (default): the library's per-layer choice | 80 / 81 / 83 / 93: that form wherever it fits
    W2C_OPT_REGW_FORM,        // layer1 kernel: 1 = LDS-staged epilogue | 2 = register-direct epilogue (bit-identical)
    W2C_OPT_REGH_WGS,         // > 0: workgroups per group of the two-waves-per-SIMD layer1 kernel (tests: odd run lengths)
    W2C_OPT_REGH_FORM,        // 0 (default) | 1 | 2 | 3: A/B forms of the two-waves-per-SIMD layer1 kernel (conv_regh.inl)
    W2C_OPT_L1_FORM,          // layer1 (Cin = Cout = 64) through w2c_conv3x3_wreg_bf16: 54 (default) = conv3x3_c64_regh_kernel | 0 = not offered
    W2C_OPT_S2WREG_FORM,      // stride-2 block fronts (conv_s2wreg.inl): 1 (default) .. 4: that kernel form wherever it fits | 0 = not offered (ring kernel)
    W2C_OPT_S2REGH,           // 1 (default) | 0: the first stride-2 block front (64 -> 128) on the persistent weights-stationary kernel (conv_s2regh.inl)
    W2C_OPT_COUNT
};
int w2c_option(int id);
