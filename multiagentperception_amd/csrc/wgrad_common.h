// Helpers shared by the weight-gradient kernels (conv_wgrad.hip, stem_train.hip): counted vmcnt wait and the transposed
// LDS fragment read that turns a [pixel][64 channel] tile into a k(=pixel)-contiguous MFMA operand.
#pragma once
#include "w2c_common.h"

namespace {

template <int N>
__device__ __forceinline__ void wg_wait_vmcnt() {
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

// transposed fragment: 32 channels x 16 pixels of a [pixel][64 ch] tile (128-B rows) -> MFMA operand (8 k per lane).
// (The builtin, not inline asm: the compiler then tracks lgkmcnt for the two reads.)
typedef __attribute__((ext_vector_type(4))) short w2c_s16x4_t;
typedef __attribute__((ext_vector_type(8))) short w2c_s16x8_t;
// LDS image of a tile: row = pixel (128 B = 64 channels); the four 32-byte windows of row p are stored at window index
// w ^ (p & 2): the 32 lanes serviced together by a ds_read_b64(_tr) read 4 consecutive pixels x 2 channel halves = 8 pieces of
// 32 B, which then fall on 8 distinct (row parity, window) slots of the 256-byte bank space -- conflict-free (at pitch 128 B
// unswizzled, rows p and p+2 share their banks).
__device__ __forceinline__ int wg_swz(int p) { return p & 2; }
__device__ __forceinline__ bf16x8_t tr_frag(const char* tile, int pix0, int cbase, int lane) {
    const int lhi = lane >> 5, g16 = (lane >> 4) & 1, r = (lane & 15) >> 2, q = lane & 3;
    const int p = pix0 + 8 * lhi + r;
    const int cb = (cbase + 16 * g16 + 4 * q) * 2;                 // byte offset of this lane's 4 channels inside the row
    const int w = cb >> 5, in = cb & 31;
    const char* a0 = tile + p * 128 + ((w ^ wg_swz(p)) << 5) + in;
    const char* a1 = tile + (p + 4) * 128 + ((w ^ wg_swz(p + 4)) << 5) + in;
    const w2c_s16x4_t lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) w2c_s16x4_t*)(a0));
    const w2c_s16x4_t hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) w2c_s16x4_t*)(a1));
    const w2c_s16x8_t v = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
    return __builtin_bit_cast(bf16x8_t, v);
}


}  // namespace
