// stem_train.hip -- weight gradient of the 7x7 / stride-2 / pad-3 stem convolution (3 -> 64, backbone.py:65 via the third-party
// resnet18) for the training backward.  The stem's input is the camera frame: it needs no gradient, so this
// and the training-forward variant of stem_kernel (stem.hip) are all the stem contributes to loss.backward().
//
//   dW[co][ci][ky][kx] = sum over output pixels (m, oy, ox) of  dY[m][oy][ox][co] * X[m][2oy + ky - 3][2ox + kx - 3][ci]
//
// A GEMM [64 co] x [147 taps] reduced over 1.3 M pixels (cfg 2).  Both MFMA operands must be pixel-contiguous per lane:
// dY^T comes out of a [pixel][64 co] LDS tile through ds_read_b64_tr_b16 exactly as in conv_wgrad.hip; for X the workgroup
// builds the IM2COL tile of its 8 x 16 output pixels in LDS -- [pixel][k], k = (ky*3 + ci)*8 + kx (kx = 7 is a pad slot: the
// 8 values of a (ky, ci) group are 8 consecutive input columns, one aligned 16-byte LDS write), 168 k's in three
// [128 px][64 k] tiles -- from a channel-planar bf16 copy of the 21 x 40 input patch, and reads it back transposed the same way.
// 4 waves (2 co halves x 2 groups of three 32-k tiles): 24 MFMAs per wave and block, 12 accumulator registers... (3 x 16).
// Segment partials [64][192] f32 -> workspace -> summed in lane order by stem_wgrad_reduce_kernel straight into the
// parameter's [64][3][7][7] layout: deterministic.
#include "w2c_common.h"
#include "wgrad_common.h"
#include <cstdlib>

namespace {

struct StemWgradArgs {
    const uint16_t* x;     // bf16 NHWC [M][H][W][3]
    const uint16_t* dy;    // bf16 NHWC [M][Ho][Wo][ycs], channels [0, 64)
    float* ws;             // [nseg][64][192]
    int M, H, W, Ho, Wo, ycs;
    int nseg, blocks_per_seg;
};

constexpr int SW_TH = 8, SW_TW = 16;                 // output pixels per block: 8 rows x 16 columns
constexpr int SW_PR = 2 * SW_TH + 5;                 // 21 input rows
constexpr int SW_PC = 40;                            // planar patch pitch (elements); element e of a row = input column 2*ox0 - 3 + e
constexpr int SW_KP = 192;                           // padded k (3 tiles of 64)

__global__ __launch_bounds__(256) void stem_wgrad_kernel(StemWgradArgs p) {
#if defined(__HIP_DEVICE_COMPILE__)
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* const Ys = smem;                                       // [128 px][64 co] bf16, wg_swz-swizzled       16 KB
    char* const It = smem + 16384;                               // 3 x [128 px][64 k] bf16, same swizzle       48 KB
    uint16_t* const P = reinterpret_cast<uint16_t*>(smem + 65536);   // planar patch [3][21][40] bf16             5 KB
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1;
    const int seg = blockIdx.x;
    const int lrow = lane >> 3, lpos = lane & 7;
    const __amdgpu_buffer_rsrc_t rs_y = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<uint16_t*>(p.dy), 0, (int)((size_t)p.M * p.Ho * p.Wo * p.ycs * 2), 0x00020000);

    f32x16_t acc[3];
#pragma unroll
    for (int t = 0; t < 3; ++t)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[t][e] = 0.f;

    const int tiles_x = p.Wo / SW_TW, tiles_y = p.Ho / SW_TH;
    const int nblocks = p.M * tiles_x * tiles_y;
    const int b_begin = seg * p.blocks_per_seg;
    const int b_end = min(nblocks, b_begin + p.blocks_per_seg);
    // dY DMA: instruction j of this wave moves tile pixels (wave + 4j)*8 + lrow
    int yty[4], ytx[4], ysrc[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int rloc = (wave + 4 * j) * 8 + lrow;
        yty[j] = rloc / SW_TW; ytx[j] = rloc - yty[j] * SW_TW;
        ysrc[j] = ((((lpos >> 1) ^ wg_swz(rloc)) << 1) | (lpos & 1)) * 16;
    }
    for (int blk = b_begin; blk < b_end; ++blk) {
        const int txi = blk % tiles_x, t2 = blk / tiles_x;
        const int tyi = t2 % tiles_y, img = t2 / tiles_y;
        const int oy0 = tyi * SW_TH, ox0 = txi * SW_TW;
        __syncthreads();                                         // the previous block's tiles are no longer read
        // ---- dY tile: LDS-DMA ----
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const unsigned vo = (unsigned)((((long)img * p.Ho + oy0 + yty[j]) * p.Wo + ox0 + ytx[j]) * p.ycs * 2 + ysrc[j]);
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_y, W2C_LPTR(Ys + (wave + 4 * j) * 1024), 16, vo, 0, 0, 0);
        }
        // ---- input patch -> channel-planar bf16 P[ci][r][e]: rows 2*oy0-3 .. +20, columns 2*ox0-4 .. +35 are fetched as
        // 8-byte pieces of the interleaved row (40 pixels x 3 ch x 2 B = 240 B = 30 pieces, the row start is 8-byte
        // aligned because ox0 is a multiple of 16); column 2*ox0-4 is dropped (e = column - (2*ox0 - 3)).  Image borders
        // fall on multiples of 4 pixels = 3 pieces, so a piece is entirely inside or outside. ----
        for (int it = tid; it < SW_PR * 30; it += 256) {
            const int r = it / 30, pc = it - r * 30;
            const int iy = 2 * oy0 - 3 + r;
            const int ix0 = 2 * ox0 - 4;
            const int el = pc * 4;                               // first interleaved element of the piece
            const int c0 = el / 3;                               // its pixel column (relative)
            const bool ok = ((unsigned)iy < (unsigned)p.H) & ((unsigned)(ix0 + c0) < (unsigned)p.W);
            uint2 v = make_uint2(0u, 0u);
            if (ok) v = *reinterpret_cast<const uint2*>(p.x + (((size_t)img * p.H + iy) * p.W + ix0) * 3 + el);
            const uint16_t h[4] = {(uint16_t)(v.x & 0xFFFFu), (uint16_t)(v.x >> 16), (uint16_t)(v.y & 0xFFFFu), (uint16_t)(v.y >> 16)};
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const int n = el + k, c = n / 3, ci = n - 3 * c;
                if (c >= 1) P[(ci * SW_PR + r) * SW_PC + (c - 1)] = h[k];
            }
        }
        __syncthreads();                                         // P complete (dY DMA still in flight)
        // ---- im2col: item (pixel, g = ky*3 + ci): P[ci][2ty+ky][2tx .. 2tx+7] -> 16 bytes at k = 8g of pixel's row ----
        for (int it = tid; it < 128 * 21; it += 256) {
            const int px = it & 127, gq = it >> 7;
            const int ty = px >> 4, tx = px & 15;
            const int ky = gq / 3, ci = gq - 3 * ky;
            const uint32_t* src = reinterpret_cast<const uint32_t*>(P + (ci * SW_PR + 2 * ty + ky) * SW_PC + 2 * tx);
            const uint4 v = make_uint4(src[0], src[1], src[2], src[3]);
            const int jt = gq >> 3, cidx = gq & 7;
            *reinterpret_cast<uint4*>(It + jt * 16384 + px * 128 + (((cidx >> 1) ^ wg_swz(px)) << 5) + (cidx & 1) * 16) = v;
        }
        wg_wait_vmcnt<0>();
        __syncthreads();                                         // im2col tiles and the dY tile are complete
        bf16x8_t ya[8];
#pragma unroll
        for (int kk = 0; kk < 8; ++kk) ya[kk] = tr_frag(Ys, kk * 16, wm * 32, lane);
#pragma unroll
        for (int t = 0; t < 3; ++t) {
            const int nt = wn * 3 + t;                           // 32-k tile 0..5
            const char* tile = It + (nt >> 1) * 16384;
#pragma unroll
            for (int kk = 0; kk < 8; ++kk) {
                const bf16x8_t xb = tr_frag(tile, kk * 16, (nt & 1) * 32, lane);
                acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ya[kk], xb, acc[t], 0, 0, 0);
            }
        }
    }
    // partial: D[i = co][j = k]; lane holds column k = l31, rows co = (e&3) + 8(e>>2) + 4 lhi
    const int l31 = lane & 31, lhi = lane >> 5;
    float* out = p.ws + (size_t)seg * 64 * SW_KP + (size_t)(wm * 32) * SW_KP + l31;
#pragma unroll
    for (int t = 0; t < 3; ++t)
#pragma unroll
        for (int e = 0; e < 16; ++e) {
            const int co = (e & 3) + 8 * (e >> 2) + 4 * lhi;
            out[(size_t)co * SW_KP + (wn * 3 + t) * 32] = acc[t][e];
        }
#endif
}

// dW[co][ci][ky][kx] = sum over segments of ws[seg][co][(ky*3 + ci)*8 + kx]; 16 threads per output (segments s, s+16, ...),
// combined in lane order through LDS: deterministic.
__global__ __launch_bounds__(256) void stem_wgrad_reduce_kernel(const float* __restrict__ ws, int nseg, float* __restrict__ dw) {
    __shared__ float red[16][16];
    const int col = threadIdx.x & 15, sl = threadIdx.x >> 4;
    const int o = blockIdx.x * 16 + col;                          // flat index into [64][3][7][7]
    float a = 0.f;
    if (o < 64 * 147) {
        const int co = o / 147, r = o - co * 147;
        const int ci = r / 49, r2 = r - ci * 49;
        const int ky = r2 / 7, kx = r2 - ky * 7;
        const float* src = ws + (size_t)co * SW_KP + (ky * 3 + ci) * 8 + kx;
        for (int s = sl; s < nseg; s += 16) a += src[(size_t)s * 64 * SW_KP];
    }
    red[sl][col] = a;
    __syncthreads();
    if (sl == 0 && o < 64 * 147) {
        float t = red[0][col];
#pragma unroll
        for (int j = 1; j < 16; ++j) t += red[j][col];
        dw[o] = t;
    }
}

int stem_wgrad_nseg(int M, int Ho, int Wo) {
    const long nblocks = (long)M * (Ho / SW_TH) * (Wo / SW_TW);
    long nseg = 512;                                              // two workgroups per CU
    if (nseg > nblocks) nseg = nblocks;
    return (int)nseg;
}

}  // namespace

extern "C" long long w2c_stem_wgrad_workspace_bytes(int M, int H, int W) {
    if (M <= 0 || H <= 0 || W <= 0 || (H % 16) != 0 || (W % 32) != 0) return -1;
    return (long long)stem_wgrad_nseg(M, H / 2, W / 2) * 64 * SW_KP * 4;
}

extern "C" int w2c_stem_wgrad_bf16(const uint16_t* x_nhwc3, int M, int H, int W, const uint16_t* dy, int dy_cstride, float* dw,
                                   void* workspace, long long workspace_bytes, w2c_stream_t stream) {
    w2c_clear_error();
    if (!x_nhwc3 || !dy || !dw || !workspace) return W2C_E_ARG;
    const long long need = w2c_stem_wgrad_workspace_bytes(M, H, W);
    if (need < 0 || workspace_bytes < need || dy_cstride < 64 || (dy_cstride % 8) != 0) return W2C_E_ARG;
    if ((reinterpret_cast<uintptr_t>(x_nhwc3) & 7) || (reinterpret_cast<uintptr_t>(workspace) & 15)) return W2C_E_ARG;
    if ((size_t)M * (H / 2) * (W / 2) * dy_cstride * 2 >= (1ull << 31)) return W2C_E_ARG;
    StemWgradArgs a;
    a.x = x_nhwc3; a.dy = dy; a.ws = reinterpret_cast<float*>(workspace);
    a.M = M; a.H = H; a.W = W; a.Ho = H / 2; a.Wo = W / 2; a.ycs = dy_cstride;
    a.nseg = stem_wgrad_nseg(M, a.Ho, a.Wo);
    const int nblocks = M * (a.Ho / SW_TH) * (a.Wo / SW_TW);
    a.blocks_per_seg = (nblocks + a.nseg - 1) / a.nseg;
    constexpr int lds = 65536 + 3 * SW_PR * SW_PC * 2 + 80;
    static std::atomic<unsigned long long> attr_mask{0};
    int dev = 0;
    (void)hipGetDevice(&dev);
    if (!((attr_mask.load(std::memory_order_acquire) >> (dev & 63)) & 1ull)) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&stem_wgrad_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
        attr_mask.fetch_or(1ull << (dev & 63), std::memory_order_release);
    }
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    hipLaunchKernelGGL(stem_wgrad_kernel, dim3(a.nseg), dim3(256), lds, s, a);
    int rc = w2c_launch_status();
    if (rc != W2C_OK) return rc;
    hipLaunchKernelGGL(stem_wgrad_reduce_kernel, dim3((64 * 147 + 15) / 16), dim3(256), 0, s, a.ws, a.nseg, dw);
    return w2c_launch_status();
}
