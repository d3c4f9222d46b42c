// stem.hip -- K1: agent slicing + conv 7x7 stride 2 pad 3 (3 -> 64 per trunk) + BN + ReLU on MFMA,
// and K1b: maxpool 3x3 s2 p1.
//
// Replaces divide_inputs + cat (agent.py:1088-1096,1105-1108) and resnet conv1/bn1/relu/maxpool
// (backbone.py:65-66,76-80 via the third-party resnet18).  Both trunks (u_encoder and
// query_key_net.img_encoder) read the same frames, so their stems run side by side as
// Cout = 128 and the f32 NCHW input is read from HBM exactly once.
//
// K = 7*7*3 = 147 is MFMA-unfriendly; it is laid out as 7 (ky) x 8 (kx, tap 7 = zero weight)
// x 4 (ci, channel 3 = zero) = 224 so that one 16-wide MFMA K-step is two horizontally
// adjacent input pixels (2 x 4 bf16 = 16 B): the B fragment is ONE aligned ds_read_b128 out of
// a [row][col][4ch] bf16 input patch in LDS -- no im2col buffer.  Efficiency 147/224 = 66 %.
//
// Workgroup = 256 threads, one 8x32 output tile at a time, walking a whole output row band.
// Wave w owns channel tile (w % (Cout/32)) for 256/PP pixels, and keeps its 32 channels' weights
// in registers for the whole kernel (14 fragments = 56 VGPRs): LDS holds only the input patch
// and the bf16 output tile, which is written back as whole 256-B (Cout=128) pixel rows.
#include "w2c_common.h"
#include <type_traits>

namespace {

typedef unsigned short u16x2_t __attribute__((ext_vector_type(2)));

constexpr int PATCH_ROWS = 21;     // 2*8 + 5 input rows for 8 output rows
constexpr int PATCH_COLS = 72;     // 2*32 + 6 (+2 pad) input cols for 32 output cols (+ zero tap)
constexpr int PATCH_BYTES = PATCH_ROWS * PATCH_COLS * 8;

// TRAIN = the training forward: `x` is the bf16 NHWC frame tensor the train path feeds ([M][H][W][3],
// torch channels_last of [M,3,H,W]; B = M, N = 1) and the output is the RAW convolution (train-mode BatchNorm follows as its
// own kernel): scale / shift are not read.
template <int COUT, bool TRAIN = false>
__global__ __launch_bounds__(256) void stem_kernel(const float* __restrict__ x, int B, int N, int H, int W,
                                                   const uint16_t* __restrict__ wpk,
                                                   const float* __restrict__ scale,
                                                   const float* __restrict__ shift,
                                                   uint16_t* __restrict__ y) {
    constexpr int CT = COUT / 32;           // channel tiles
    constexpr int PP = 4 / CT;              // pixel partitions across waves
    constexpr int MT = 8 / PP;              // 32-pixel tiles (= output rows) per wave
    constexpr int ROWBYTES = COUT * 2;      // staged output bytes per pixel
    constexpr int CHUNKS = ROWBYTES / 16;   // 16-B chunks per pixel
    extern __shared__ __attribute__((aligned(16))) char smem[];
    uint2* patch = reinterpret_cast<uint2*>(smem);
    char* stagebuf = smem + PATCH_BYTES;

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int ct = wave % CT, pp = wave / CT;
    const int l31 = lane & 31, lhi = lane >> 5;
    const int Ho = H >> 1, Wo = W >> 1;
    const int img = blockIdx.y;                     // agent-major image index a*B + b
    const int agent = img / B, b = img - agent * B;
    const int oy0 = blockIdx.x * 8;
    const float* xin = x + ((size_t)b * 3 * N + 3 * agent) * H * W;   // 3 planes of this agent

    // this wave's weights: A operand, i = channel, k = (kx pair, ci) within one ky
    bf16x8_t wf[7][2];
    {
        const uint16_t* wrow = wpk + (size_t)(ct * 32 + l31) * 224;
#pragma unroll
        for (int ky = 0; ky < 7; ++ky)
#pragma unroll
            for (int ks = 0; ks < 2; ++ks)
                wf[ky][ks] = *reinterpret_cast<const bf16x8_t*>(wrow + ky * 32 + (ks * 2 + lhi) * 8);
    }

    // ---- input patch staging, software-pipelined across x tiles: every thread owns FILL patch pixels;
    // the 3*FILL f32 loads of tile i+1 are issued before the MFMAs of tile i and consumed after them,
    // so the HBM round trip hides behind the matrix work instead of stalling each tile six times.
    constexpr int FILL = (PATCH_ROWS * PATCH_COLS + 255) / 256;         // 6
    float pv[FILL][3];
    unsigned pmask = 0;                                  // bit f: patch pixel f of this thread is inside the image
    auto load_patch = [&](int ox0) {
        const int iy_base = 2 * oy0 - 3, ix_base = 2 * ox0 - 3;
#pragma unroll
        for (int f = 0; f < FILL; ++f) {
            const int pidx = tid + f * 256;
            const int r = pidx / PATCH_COLS, c = pidx - r * PATCH_COLS;
            const int iy = iy_base + r, ix = ix_base + c;
            const bool ok = (pidx < PATCH_ROWS * PATCH_COLS) & (iy >= 0) & (iy < H) & (ix >= 0) & (ix < W);
            const size_t o = ok ? (size_t)iy * W + ix : 0;              // clamp: always a valid address
            // keep the RAW loaded values: masking here would make the compiler wait for the loads right away
            // (the select cannot sink across the barriers below); the mask is applied in store_patch().
            if constexpr (TRAIN) {
                const uint16_t* x16 = reinterpret_cast<const uint16_t*>(x) + ((size_t)img * H * W + o) * 3;
                pv[f][0] = bf16_to_f32(x16[0]); pv[f][1] = bf16_to_f32(x16[1]); pv[f][2] = bf16_to_f32(x16[2]);
            } else {
                pv[f][0] = xin[o]; pv[f][1] = xin[o + (size_t)H * W]; pv[f][2] = xin[o + 2 * (size_t)H * W];
            }
            pmask = ok ? (pmask | (1u << f)) : (pmask & ~(1u << f));
        }
    };
    auto store_patch = [&]() {
#pragma unroll
        for (int f = 0; f < FILL; ++f) {
            const int pidx = tid + f * 256;
            const bool ok = (pmask >> f) & 1u;
            if (pidx < PATCH_ROWS * PATCH_COLS)
                patch[pidx] = ok ? make_uint2(pack_bf16x2(pv[f][0], pv[f][1]), pack_bf16x2(pv[f][2], 0.f)) : make_uint2(0u, 0u);
        }
    };

    load_patch(0);
    for (int ox0 = 0; ox0 < Wo; ox0 += 32) {
        store_patch();                                   // (previous tile's MFMA reads are behind a barrier)
        __syncthreads();
        if (ox0 + 32 < Wo) load_patch(ox0 + 32);         // in flight during this tile's MFMAs

        f32x16_t acc[MT];
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[mt][e] = 0.f;

#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
            const int oyl = pp * MT + mt;
#pragma unroll
            for (int ky = 0; ky < 7; ++ky) {
                const uint2* prow = patch + (2 * oyl + ky) * PATCH_COLS + 2 * l31 + 2 * lhi;
#pragma unroll
                for (int ks = 0; ks < 2; ++ks) {
                    const bf16x8_t pb = *reinterpret_cast<const bf16x8_t*>(prow + ks * 4);
                    acc[mt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf[ky][ks], pb, acc[mt], 0, 0, 0);
                }
            }
        }

        // ---- epilogue: BN scale/shift + ReLU, bf16, stage [pixel][COUT] (chunk-swizzled) ----
        // D row (channel) = (e&3) + 8*(e>>2) + 4*lhi, D col (pixel) = l31
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int ch = ct * 32 + 8 * q + 4 * lhi;
            f32x4_t sc = {1.f, 1.f, 1.f, 1.f}, sh = {0.f, 0.f, 0.f, 0.f};
            if constexpr (!TRAIN) {
                sc = *reinterpret_cast<const f32x4_t*>(scale + ch);
                sh = *reinterpret_cast<const f32x4_t*>(shift + ch);
            }
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) {
                const int px = (pp * MT + mt) * 32 + l31;
                float v[4];
#pragma unroll
                for (int r = 0; r < 4; ++r) v[r] = TRAIN ? acc[mt][4 * q + r] : fmaxf(acc[mt][4 * q + r] * sc[r] + sh[r], 0.f);
                const int chunk = (ct * 4 + q) ^ (px & (CHUNKS - 1));
                *reinterpret_cast<uint2*>(stagebuf + px * ROWBYTES + chunk * 16 + lhi * 8) =
                    make_uint2(pack_bf16x2(v[0], v[1]), pack_bf16x2(v[2], v[3]));
            }
        }
        __syncthreads();
        // ---- coalesced write-back: 16 B per thread, whole pixel rows ----
        for (int id = tid; id < 256 * CHUNKS; id += 256) {
            const int px = id / CHUNKS, cg = id - px * CHUNKS;
            const uint4 v = *reinterpret_cast<const uint4*>(stagebuf + px * ROWBYTES + ((cg ^ (px & (CHUNKS - 1))) * 16));
            const int oy = oy0 + (px >> 5), ox = ox0 + (px & 31);
            *reinterpret_cast<uint4*>(y + (((size_t)img * Ho + oy) * Wo + ox) * COUT + cg * 8) = v;
        }
        // next iteration's patch fill is ordered after this iteration's MFMA reads by the
        // barrier above; its staging writes are ordered after this read-out by the barrier
        // that follows the patch fill.
    }
}

// ---------------------------------------------------------------------------------------------
// Fused stem: conv 7x7/2 + BN + ReLU + maxpool 3x3/2 p1 (backbone.py:65-66,76-80) in one kernel.
// The unfused pair writes the 64(x2)-channel half-resolution map to HBM and reads it straight
// back (cfg 2: 335 MB each way); here a workgroup (8 waves) owns a band of 8 conv rows, walks it in
// 32-column steps and per step computes a 9 x 32 conv tile (the row above the band is recomputed:
// +12.5 % MFMA work; the column left of the step is carried in LDS from the previous step), stages
// it as bf16 in LDS, pools 3x3/2 there and writes only the pooled 4 x 16 pixels.  ReLU outputs are
// >= 0 and every pooling window holds at least one in-image pixel, so out-of-image conv pixels are
// staged as 0 instead of -inf (identical maxima).
// BAND = conv rows a workgroup owns (8 or 4): BAND + 1 rows are computed per step (the row above is recomputed).
// BAND 8: 9/8 recompute but 89 KB of LDS at Cout 128 -> one workgroup per CU, whose phases (patch store, MFMA,
// BN/ReLU staging, pooling) run one after the other on every SIMD; BAND 4: 5/4 recompute, 51 KB -> two
// workgroups per CU whose phases overlap (measured with tools/stem_phases.py).
constexpr int FP_COLS = 72;
template <int BAND> constexpr int fp_rows() { return 2 * BAND + 7; }      // input rows for BAND + 1 conv rows
template <int BAND> constexpr int fp_bytes() { return fp_rows<BAND>() * FP_COLS * 8; }

// U8 = true: `x` holds the camera frames as the reference's loader reads them -- u8 RGB, [B][N][H][W][3] -- and the
// loader's transform (airsim_loader.py:521-527: RGB->BGR, float64 (v - mean)/255, cast to f32) is applied while the
// patch is staged, in double precision, so the bf16 patch is bit-identical to staging the transformed f32 frames.
struct FrameMean { double m[3]; };   // BGR means

#ifdef W2C_STEM_TIMING
__device__ unsigned long long g_stem_phase[8];
#define STEM_T(i) do { const long long t_ = clock64(); t_acc[i] += (unsigned long long)(t_ - t_prev); t_prev = t_; } while (0)
#else
#define STEM_T(i) do {} while (0)
#endif

template <int COUT, bool U8, int BAND, int NW>
__global__ __launch_bounds__(64 * NW) void stem_pool_kernel(const void* xv_arg, FrameMean mean, int B, int N, int H, int W,
                                                        const uint16_t* __restrict__ wpk,
                                                        const float* __restrict__ scale,
                                                        const float* __restrict__ shift,
                                                        uint16_t* __restrict__ y) {
    constexpr int FP_ROWS = fp_rows<BAND>(), FP_BYTES = fp_bytes<BAND>(), CR = BAND + 1;   // CR = conv rows per step
    const void* const __restrict__ xv = w2c_resolve(xv_arg);       // indirect operand: the frames' address may come from a pointer slot
    constexpr int CT = COUT / 32;           // channel tiles
    constexpr int NT = 64 * NW;             // NW = 8 waves, or 12 so that the BAND + 1 = 9 conv rows split 3/3/3 (no 4-vs-5 wait)
    constexpr int NPART = NW / CT;          // conv-row partitions across waves
    constexpr int MT_MAX = (CR + NPART - 1) / NPART;
    constexpr int ROWBYTES = COUT * 2;      // staged bytes per conv pixel
    constexpr int CHUNKS = ROWBYTES / 16;
    constexpr int SCOLS = 33;               // staged columns: [carry | 32 new]
    extern __shared__ __attribute__((aligned(16))) char smem[];
    uint2* patch = reinterpret_cast<uint2*>(smem);
    char* stg = smem + FP_BYTES;            // [CR][33][COUT] bf16, 16-B chunks swizzled by (col & (CHUNKS-1))

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int ct = wave % CT, part = wave / CT;
    const int mt_lo = (CR * part) / NPART, mt_hi = (CR * (part + 1)) / NPART;      // this wave's conv rows [mt_lo, mt_hi)
    const int l31 = lane & 31, lhi = lane >> 5;
    const int Ho = H >> 1, Wo = W >> 1, Hp = H >> 2, Wp = W >> 2;
    const int img = blockIdx.y;
    const int agent = img / B, b = img - agent * B;
    const int oy0 = blockIdx.x * BAND;
    const float* xin = U8 ? nullptr : reinterpret_cast<const float*>(xv) + ((size_t)b * 3 * N + 3 * agent) * H * W;
    const uint8_t* xin8 = U8 ? reinterpret_cast<const uint8_t*>(xv) + ((size_t)b * N + agent) * H * W * 3 : nullptr;

    bf16x8_t wf[7][2];
    {
        const uint16_t* wrow = wpk + (size_t)(ct * 32 + l31) * 224;
#pragma unroll
        for (int ky = 0; ky < 7; ++ky)
#pragma unroll
            for (int ks = 0; ks < 2; ++ks)
                wf[ky][ks] = *reinterpret_cast<const bf16x8_t*>(wrow + ky * 32 + (ks * 2 + lhi) * 8);
    }

    f32x4_t e_sc[4], e_sh[4];               // BN scale / shift of this wave's 32 channels (fixed for the kernel)
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        e_sc[q] = *reinterpret_cast<const f32x4_t*>(scale + ct * 32 + 8 * q + 4 * lhi);
        e_sh[q] = *reinterpret_cast<const f32x4_t*>(shift + ct * 32 + 8 * q + 4 * lhi);
    }
    constexpr int FILL = (FP_ROWS * FP_COLS + NT - 1) / NT;
    float pv[FILL][3];
    unsigned pmask = 0;
    auto load_patch = [&](int ox0) {
        const int iy_base = 2 * oy0 - 5, ix_base = 2 * ox0 - 3;         // conv row oy0-1 needs input row 2*(oy0-1)-3
#pragma unroll
        for (int f = 0; f < FILL; ++f) {
            const int pidx = tid + f * NT;
            const int r = pidx / FP_COLS, c = pidx - r * FP_COLS;
            const int iy = iy_base + r, ix = ix_base + c;
            const bool ok = (pidx < FP_ROWS * FP_COLS) & (iy >= 0) & (iy < H) & (ix >= 0) & (ix < W);
            const size_t o = ok ? (size_t)iy * W + ix : 0;
            if constexpr (U8) {       // network channel c (BGR) = frame channel 2-c (RGB); raw bytes, converted at store
                pv[f][0] = (float)xin8[o * 3 + 2]; pv[f][1] = (float)xin8[o * 3 + 1]; pv[f][2] = (float)xin8[o * 3];
            } else {
                pv[f][0] = xin[o]; pv[f][1] = xin[o + (size_t)H * W]; pv[f][2] = xin[o + 2 * (size_t)H * W];   // raw, masked at store
            }
            pmask = ok ? (pmask | (1u << f)) : (pmask & ~(1u << f));
        }
    };
    auto store_patch = [&]() {
#pragma unroll
        for (int f = 0; f < FILL; ++f) {
            const int pidx = tid + f * NT;
            const bool ok = (pmask >> f) & 1u;
            float v0 = pv[f][0], v1 = pv[f][1], v2 = pv[f][2];
            if constexpr (U8) {
                v0 = (float)(((double)v0 - mean.m[0]) / 255.0);
                v1 = (float)(((double)v1 - mean.m[1]) / 255.0);
                v2 = (float)(((double)v2 - mean.m[2]) / 255.0);
            }
            if (pidx < FP_ROWS * FP_COLS)
                patch[pidx] = ok ? make_uint2(pack_bf16x2(v0, v1), pack_bf16x2(v2, 0.f)) : make_uint2(0u, 0u);
        }
    };
    auto stg_addr = [&](int row, int col, int chunk) -> char* {
        return stg + ((row * SCOLS + col) * CHUNKS + (chunk ^ (col & (CHUNKS - 1)))) * 16;
    };

    // carry column (conv column -1) of the first step is outside the image: zeros
    for (int i = tid; i < CR * CHUNKS; i += NT)
        *reinterpret_cast<uint4*>(stg_addr(i / CHUNKS, 0, i % CHUNKS)) = make_uint4(0, 0, 0, 0);

    load_patch(0);
#ifdef W2C_STEM_TIMING
    unsigned long long t_acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    long long t_prev = clock64();
#endif
    for (int ox0 = 0; ox0 < Wo; ox0 += 32) {
        store_patch();
        STEM_T(0);
        __syncthreads();                    // patch visible; previous step's pooling reads + carry copy are done
        STEM_T(1);
        if (ox0 + 32 < Wo) load_patch(ox0 + 32);
        STEM_T(2);

        f32x16_t acc[MT_MAX];
#pragma unroll
        for (int m = 0; m < MT_MAX; ++m)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[m][e] = 0.f;
#pragma unroll
        for (int m = 0; m < MT_MAX; ++m) {
            const int mt = mt_lo + m;
            if (mt < mt_hi) {
#pragma unroll
                for (int ky = 0; ky < 7; ++ky) {
                    const uint2* prow = patch + (2 * mt + ky) * FP_COLS + 2 * l31 + 2 * lhi;
#pragma unroll
                    for (int ks = 0; ks < 2; ++ks) {
                        const bf16x8_t pb = *reinterpret_cast<const bf16x8_t*>(prow + ks * 4);
                        acc[m] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf[ky][ks], pb, acc[m], 0, 0, 0);
                    }
                }
            }
        }
        STEM_T(3);
        // BN + ReLU -> bf16 -> staging[mt][1 + l31][channels]; conv row oy0-1+mt < 0 is outside the image
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const f32x4_t sc = e_sc[q], sh = e_sh[q];
#pragma unroll
            for (int m = 0; m < MT_MAX; ++m) {
                const int mt = mt_lo + m;
                if (mt < mt_hi) {
                    const bool inimg = (oy0 - 1 + mt) >= 0;
                    float v[4];
#pragma unroll
                    for (int r = 0; r < 4; ++r) v[r] = inimg ? fmaxf(acc[m][4 * q + r] * sc[r] + sh[r], 0.f) : 0.f;
                    // sign bit cleared: a -0 out of fmaxf would order above every positive value in the u16 max below
                    *reinterpret_cast<uint2*>(stg_addr(mt, 1 + l31, ct * 4 + q) + lhi * 8) =
                        make_uint2(pack_bf16x2(v[0], v[1]) & 0x7FFF7FFFu, pack_bf16x2(v[2], v[3]) & 0x7FFF7FFFu);
                }
            }
        }
        STEM_T(4);
        __syncthreads();
        STEM_T(5);
        // pool 3x3/2: pooled (pyl, pxl) <- staged rows 2*pyl..2*pyl+2, staged cols 2*pxl..2*pxl+2
        for (int id = tid; id < (BAND / 2) * 16 * CHUNKS; id += NT) {
            const int cg = id % CHUNKS, pp = id / CHUNKS;
            const int pyl = pp >> 4, pxl = pp & 15;
            // staged values are ReLU outputs (>= 0, never -0: fmaxf(x, 0.f) of a negative is +0), and non-negative bf16
            // order like their bit patterns: the 3x3 max is a packed unsigned 16-bit max, no unpacking.
            u16x2_t best[4] = {{0, 0}, {0, 0}, {0, 0}, {0, 0}};
#pragma unroll
            for (int dy = 0; dy < 3; ++dy)
#pragma unroll
                for (int dx = 0; dx < 3; ++dx) {
                    const uint4 u = *reinterpret_cast<const uint4*>(stg_addr(2 * pyl + dy, 2 * pxl + dx, cg));
                    const uint32_t wv[4] = {u.x, u.y, u.z, u.w};
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const u16x2_t t = __builtin_bit_cast(u16x2_t, wv[e]);
                        best[e] = __builtin_elementwise_max(best[e], t);
                    }
                }
            uint4 o;
            o.x = __builtin_bit_cast(uint32_t, best[0]); o.y = __builtin_bit_cast(uint32_t, best[1]);
            o.z = __builtin_bit_cast(uint32_t, best[2]); o.w = __builtin_bit_cast(uint32_t, best[3]);
            const int py = (oy0 >> 1) + pyl, px = (ox0 >> 1) + pxl;
            *reinterpret_cast<uint4*>(y + (((size_t)img * Hp + py) * Wp + px) * COUT + cg * 8) = o;
        }
        STEM_T(6);
        __syncthreads();
        // carry: staged column 32 (conv column ox0+31) becomes column 0 of the next step
        for (int i = tid; i < CR * CHUNKS; i += NT) {
            const int row = i / CHUNKS, cg = i % CHUNKS;
            *reinterpret_cast<uint4*>(stg_addr(row, 0, cg)) = *reinterpret_cast<const uint4*>(stg_addr(row, 32, cg));
        }
        STEM_T(7);
        // (ordered before the next step's staging writes by the barrier after store_patch)
    }
#ifdef W2C_STEM_TIMING
    if (tid == 0)
        for (int i = 0; i < 8; ++i) atomicAdd(&g_stem_phase[i], t_acc[i]);
#endif
}

// ---------------------------------------------------------------------------------------------
// Fused stem, second form: pooling in REGISTERS.
// stem_pool_kernel above stages every conv pixel (9 x 33 x Cout bf16 = 76 KB) in LDS and pools from there; one
// workgroup fits per CU and its phases (patch store | MFMA | BN/ReLU staging | pooling) run one after the other on
// every SIMD (tools/stem_phases.py: 11.5 k cycles per step for 4.0 k of MFMA).  Here a workgroup is only Cout/32 waves:
// wave ct computes ALL 9 conv rows of the 32-column step for its 32 channels (126 MFMAs, weights in 56 registers,
// accumulators 144), applies BN + ReLU, takes the vertical 3-max in registers, packs to bf16 and takes the horizontal
// 3-max with two DPP wave shifts (lane = conv column; the column left of the step is carried through 256 B of
// wave-private LDS), then writes the 4 x 16 pooled pixels of its channels through a 4 KB wave-private staging as
// 16-byte stores.  LDS per workgroup: double-buffered input patch (26 KB) + 17 KB  =>  two to three workgroups per
// CU whose phases DO overlap, one barrier per step (patch hand-over).
__device__ __forceinline__ uint32_t lane_from_left(uint32_t x) {       // lane i <- lane i-1 (wave_shr:1)
    return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x138, 0xf, 0xf, false);
}
__device__ __forceinline__ uint32_t lane_from_right(uint32_t x) {      // lane i <- lane i+1 (wave_shl:1)
    return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x130, 0xf, 0xf, false);
}
__device__ __forceinline__ uint32_t pk_max_u16(uint32_t a, uint32_t b) {
    return __builtin_bit_cast(uint32_t, __builtin_elementwise_max(__builtin_bit_cast(u16x2_t, a), __builtin_bit_cast(u16x2_t, b)));
}

template <int COUT, bool U8>
__global__ __launch_bounds__(COUT * 2) __attribute__((amdgpu_waves_per_eu(2))) void stem_pool2_kernel(const void* xv_arg, FrameMean mean, int B, int N, int H, int W,
                                                            const uint16_t* __restrict__ wpk,
                                                            const float* __restrict__ scale,
                                                            const float* __restrict__ shift,
                                                            uint16_t* __restrict__ y) {
    constexpr int BAND = 8, CR = 9;
    const void* const __restrict__ xv = w2c_resolve(xv_arg);       // indirect operand: the frames' address may come from a pointer slot
    constexpr int FP_ROWS = fp_rows<BAND>(), FP_BYTES = fp_bytes<BAND>();
    constexpr int CT = COUT / 32, NT = 64 * CT;
    constexpr int SPX = 80;                                    // staging pitch per pooled pixel: 64 B + 16 (pixel rows 20 banks apart:
                                                               // the 8 active lanes of a ds_write_b64 group hit distinct banks; at 64 B
                                                               // they hit two -- PMC: 28 % of this kernel's LDS cycles were conflicts)
    constexpr int WAVE_LDS = 64 * SPX + 256 + 256;             // pooled staging [64 px][SPX] + carry [2 halves][128 B] + scale|shift
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int ct = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l31 = lane & 31, lhi = lane >> 5;
    const int Wo = W >> 1, Hp = H >> 2, Wp = W >> 2;
    const int img = blockIdx.y;
    const int agent = img / B, b = img - agent * B;
    const int oy0 = blockIdx.x * BAND;
    const float* xin = U8 ? nullptr : reinterpret_cast<const float*>(xv) + ((size_t)b * 3 * N + 3 * agent) * H * W;
    const uint8_t* xin8 = U8 ? reinterpret_cast<const uint8_t*>(xv) + ((size_t)b * N + agent) * H * W * 3 : nullptr;
    char* const stgw = smem + 2 * FP_BYTES + ct * WAVE_LDS;
    char* const carry = stgw + 64 * SPX;

    bf16x8_t wf[7][2];
    {
        const uint16_t* wrow = wpk + (size_t)(ct * 32 + l31) * 224;
#pragma unroll
        for (int ky = 0; ky < 7; ++ky)
#pragma unroll
            for (int ks = 0; ks < 2; ++ks)
                wf[ky][ks] = *reinterpret_cast<const bf16x8_t*>(wrow + ky * 32 + (ks * 2 + lhi) * 8);
        // BN (a*scale + shift) followed by ReLU is monotone in `a` when scale >= 0, so the 3-max can be taken on the raw
        // accumulators and BN + ReLU applied to a quarter of the values.  A channel with a negative scale gets its weights
        // negated here (exact: the accumulator is exactly negated) and |scale| below -- bit-identical to BN-then-max.
        if (scale[ct * 32 + l31] < 0.f) {
#pragma unroll
            for (int ky = 0; ky < 7; ++ky)
#pragma unroll
                for (int ks = 0; ks < 2; ++ks) {
                    u32x4_t w = __builtin_bit_cast(u32x4_t, wf[ky][ks]);
                    w ^= u32x4_t{0x80008000u, 0x80008000u, 0x80008000u, 0x80008000u};
                    wf[ky][ks] = __builtin_bit_cast(bf16x8_t, w);
                }
        }
    }
    // BN scale | shift of this wave's channels in wave-private LDS, in accumulator order: [q][lhi][4] (32 registers saved)
    char* const ssb = carry + 256;
    if (lane < 32) {
        const int q = lane >> 3, hh = (lane >> 2) & 1, k = lane & 3;
        reinterpret_cast<float*>(ssb)[(q * 2 + hh) * 4 + k] = fabsf(scale[ct * 32 + 8 * q + 4 * hh + k]);
        reinterpret_cast<float*>(ssb + 128)[(q * 2 + hh) * 4 + k] = shift[ct * 32 + 8 * q + 4 * hh + k];
    }
    const char* const ssw = ssb + lhi * 16;                    // + q*32: this lane's 4 scales; +128: shifts
    constexpr int FILL = (FP_ROWS * FP_COLS + NT - 1) / NT;
    float pv[FILL][3];
    unsigned pmask = 0;
    // a thread stages the same patch positions (row r, column c) at every step; only the image column moves (+64 per
    // step), so the row offset / row validity are computed once
    int prow_off[FILL], pcol[FILL];
#pragma unroll
    for (int f = 0; f < FILL; ++f) {
        const int pidx = tid + f * NT;
        const int r = pidx / FP_COLS, c = pidx - r * FP_COLS;
        const int iy = 2 * oy0 - 5 + r;
        const bool rok = (pidx < FP_ROWS * FP_COLS) & (iy >= 0) & (iy < H);
        prow_off[f] = rok ? iy * W : -1;
        pcol[f] = c - 3;
    }
    auto load_patch = [&](int ox0) {
#pragma unroll
        for (int f = 0; f < FILL; ++f) {
            const int pidx = tid + f * NT;
            const int ix = 2 * ox0 + pcol[f];
            const bool ok = (prow_off[f] >= 0) & (ix >= 0) & (ix < W);
            const size_t o = ok ? (size_t)(prow_off[f] + ix) : 0;
            (void)pidx;
            if constexpr (U8) {
                pv[f][0] = (float)xin8[o * 3 + 2]; pv[f][1] = (float)xin8[o * 3 + 1]; pv[f][2] = (float)xin8[o * 3];
            } else {
                pv[f][0] = xin[o]; pv[f][1] = xin[o + (size_t)H * W]; pv[f][2] = xin[o + 2 * (size_t)H * W];
            }
            pmask = ok ? (pmask | (1u << f)) : (pmask & ~(1u << f));
        }
    };
    auto store_patch = [&](uint2* patch) {
#pragma unroll
        for (int f = 0; f < FILL; ++f) {
            const int pidx = tid + f * NT;
            const bool ok = (pmask >> f) & 1u;
            float v0 = pv[f][0], v1 = pv[f][1], v2 = pv[f][2];
            if constexpr (U8) {
                v0 = (float)(((double)v0 - mean.m[0]) / 255.0);
                v1 = (float)(((double)v1 - mean.m[1]) / 255.0);
                v2 = (float)(((double)v2 - mean.m[2]) / 255.0);
            }
            if (pidx < FP_ROWS * FP_COLS)
                patch[pidx] = ok ? make_uint2(pack_bf16x2(v0, v1), pack_bf16x2(v2, 0.f)) : make_uint2(0u, 0u);
        }
    };

    // carry (vertical maxima of conv column -1) of the first step: outside the image -> 0
    if (l31 == 0) {
#pragma unroll
        for (int i = 0; i < 8; ++i) *reinterpret_cast<uint4*>(carry + lhi * 128 + i * 16) = make_uint4(0, 0, 0, 0);
    }
    load_patch(0);
    int buf = 0;
#ifdef W2C_STEM_TIMING
    unsigned long long t_acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    long long t_prev = clock64();
#endif
    for (int ox0 = 0; ox0 < Wo; ox0 += 32, buf ^= 1) {
        uint2* patch = reinterpret_cast<uint2*>(smem + buf * FP_BYTES);
        store_patch(patch);
        STEM_T(0);
        __syncthreads();                    // patch(step) visible; the buffer step+1 will overwrite was last read in step-1
        STEM_T(1);
        if (ox0 + 32 < Wo) load_patch(ox0 + 32);
        STEM_T(2);

        // ---- 9 conv rows x 32 columns x this wave's 32 channels, in two passes (rows 0-4, then 5-8) so that the live
        // accumulators (80 + 64 registers instead of 144) leave room for two waves per SIMD; BN + ReLU, then the vertical
        // 3-max in registers: pooled row p <- conv rows 2p, 2p+1, 2p+2 (row 0 = oy0-1); row 4 feeds both passes and is
        // kept as 16 post-ReLU values.  acc element e = channel (e&3) + 8(e>>2) + 4 lhi; packed dword d of pooled row p
        // holds the channels of e = 2d, 2d+1. ----
        // horizontal 3-max of two pooled rows (lane = conv column; even lanes 2j end up with pooled column j), then those
        // rows' pooled pixels -> wave-private staging [pooled row][pooled col][32 ch] -> 16-B stores
        auto finish_rows = [&](int pr0, uint32_t (&pk2)[2][8]) {
            uint4 cin[4];
            if (l31 == 0) {
#pragma unroll
                for (int i = 0; i < 4; ++i) cin[i] = *reinterpret_cast<const uint4*>(carry + lhi * 128 + pr0 * 32 + i * 16);
            }
            asm volatile("" ::: "memory");                     // the carry is read before this step overwrites it
            if (l31 == 31) {
#pragma unroll
                for (int pr = 0; pr < 2; ++pr) {
                    *reinterpret_cast<uint4*>(carry + lhi * 128 + (pr0 + pr) * 32) = make_uint4(pk2[pr][0], pk2[pr][1], pk2[pr][2], pk2[pr][3]);
                    *reinterpret_cast<uint4*>(carry + lhi * 128 + (pr0 + pr) * 32 + 16) = make_uint4(pk2[pr][4], pk2[pr][5], pk2[pr][6], pk2[pr][7]);
                }
            }
#pragma unroll
            for (int pr = 0; pr < 2; ++pr) {
#pragma unroll
                for (int d = 0; d < 8; ++d) {
                    uint32_t left = lane_from_left(pk2[pr][d]);
                    if (l31 == 0) {
                        const uint4 c = cin[pr * 2 + (d >> 2)];
                        left = (d & 3) == 0 ? c.x : (d & 3) == 1 ? c.y : (d & 3) == 2 ? c.z : c.w;
                    }
                    const uint32_t right = lane_from_right(pk2[pr][d]);
                    pk2[pr][d] = pk_max_u16(pk_max_u16(left, pk2[pr][d]), right);
                }
            }
            if ((l31 & 1) == 0) {
#pragma unroll
                for (int pr = 0; pr < 2; ++pr)
#pragma unroll
                    for (int q = 0; q < 4; ++q)    // channels 8q + 4 lhi .. +4  -> bytes (8q + 4 lhi) * 2
                        *reinterpret_cast<uint2*>(stgw + ((pr0 + pr) * 16 + (l31 >> 1)) * SPX + (8 * q + 4 * lhi) * 2) =
                            make_uint2(pk2[pr][2 * q], pk2[pr][2 * q + 1]);
            }
            asm volatile("" ::: "memory");
#pragma unroll
            for (int pr = 0; pr < 2; ++pr) {       // a pooled row: 16 pixels x 64 B = 64 lanes x 16 B
                const uint4 o = *reinterpret_cast<const uint4*>(stgw + ((pr0 + pr) * 16 + (lane >> 2)) * SPX + (lane & 3) * 16);
                const int py = (oy0 >> 1) + pr0 + pr, px = (ox0 >> 1) + (lane >> 2);
                *reinterpret_cast<uint4*>(y + (((size_t)img * Hp + py) * Wp + px) * COUT + ct * 32 + (lane & 3) * 8) = o;
            }
        };
        float keep[16];
        auto conv_rows = [&](auto first_tag, auto count_tag, f32x16_t* acc) {
            constexpr int R0 = decltype(first_tag)::value, RN = decltype(count_tag)::value;
#pragma unroll
            for (int m = 0; m < RN; ++m) {
#pragma unroll
                for (int ky = 0; ky < 7; ++ky) {
                    const uint2* prow = patch + (2 * (R0 + m) + ky) * FP_COLS + 2 * l31 + 2 * lhi;
#pragma unroll
                    for (int ks = 0; ks < 2; ++ks) {
                        const bf16x8_t pb = *reinterpret_cast<const bf16x8_t*>(prow + ks * 4);
                        if (ky == 0 && ks == 0)                 // zero C operand: no accumulator zeroing (144 v_mov per step)
                            acc[m] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf[ky][ks], pb, f32x16_t{0.f}, 0, 0, 0);
                        else
                            acc[m] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf[ky][ks], pb, acc[m], 0, 0, 0);
                    }
                }
            }
        };
        constexpr float NEG_INF = -3.0e38f;
        {
            uint32_t pk[2][8];
            f32x16_t acc[5];
            conv_rows(std::integral_constant<int, 0>{}, std::integral_constant<int, 5>{}, acc);
            STEM_T(3);
            const bool in0 = oy0 > 0;                          // conv row oy0-1 is outside the image for the first band
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const f32x4_t sc = *reinterpret_cast<const f32x4_t*>(ssw + q * 32), sh = *reinterpret_cast<const f32x4_t*>(ssw + 128 + q * 32);
                float v0[4], v1[4];
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const int e = 4 * q + k;
                    keep[e] = acc[4][e];
                    const float m0 = fmaxf(fmaxf(in0 ? acc[0][e] : NEG_INF, acc[1][e]), acc[2][e]);
                    const float m1 = fmaxf(fmaxf(acc[2][e], acc[3][e]), acc[4][e]);
                    v0[k] = fmaxf(m0 * sc[k] + sh[k], 0.f);
                    v1[k] = fmaxf(m1 * sc[k] + sh[k], 0.f);
                }
                pk[0][2 * q] = pack_bf16x2(v0[0], v0[1]) & 0x7FFF7FFFu;       // sign cleared: u16 ordering below
                pk[0][2 * q + 1] = pack_bf16x2(v0[2], v0[3]) & 0x7FFF7FFFu;
                pk[1][2 * q] = pack_bf16x2(v1[0], v1[1]) & 0x7FFF7FFFu;
                pk[1][2 * q + 1] = pack_bf16x2(v1[2], v1[3]) & 0x7FFF7FFFu;
            }
            STEM_T(4);
            finish_rows(0, pk);
            STEM_T(5);
        }
        {
            uint32_t pk[2][8];
            f32x16_t acc[4];
            conv_rows(std::integral_constant<int, 5>{}, std::integral_constant<int, 4>{}, acc);
            STEM_T(6);
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const f32x4_t sc = *reinterpret_cast<const f32x4_t*>(ssw + q * 32), sh = *reinterpret_cast<const f32x4_t*>(ssw + 128 + q * 32);
                float v2[4], v3[4];
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const int e = 4 * q + k;
                    const float m2 = fmaxf(fmaxf(keep[e], acc[0][e]), acc[1][e]);
                    const float m3 = fmaxf(fmaxf(acc[1][e], acc[2][e]), acc[3][e]);
                    v2[k] = fmaxf(m2 * sc[k] + sh[k], 0.f);
                    v3[k] = fmaxf(m3 * sc[k] + sh[k], 0.f);
                }
                pk[0][2 * q] = pack_bf16x2(v2[0], v2[1]) & 0x7FFF7FFFu;
                pk[0][2 * q + 1] = pack_bf16x2(v2[2], v2[3]) & 0x7FFF7FFFu;
                pk[1][2 * q] = pack_bf16x2(v3[0], v3[1]) & 0x7FFF7FFFu;
                pk[1][2 * q + 1] = pack_bf16x2(v3[2], v3[3]) & 0x7FFF7FFFu;
            }
            finish_rows(2, pk);
        }
        asm volatile("" ::: "memory");             // next step's staging writes stay behind these reads (same wave: in order)
        STEM_T(7);
    }
#ifdef W2C_STEM_TIMING
    if (tid == 0)
        for (int i = 0; i < 8; ++i) atomicAdd(&g_stem_phase[i], t_acc[i]);
#endif
}

// ---------------------------------------------------------------------------------------------
// Fused stem, third form: the second form's arithmetic as a persistent 8-wave PING-PONG (verdict r1 item 7).
// A workgroup is TWO groups of Cout/32 waves; wave ct of group A and wave ct of group B own the same 32 channels and sit on
// the same SIMD.  The groups take alternate 32-column steps and run their four phases
//     M1 (conv rows 0-4: 70 MFMAs) | V1 (BN, max, pack, pooled rows 0-1 out) | M2 (rows 5-8: 56 MFMAs) | V2 (pooled rows 2-3 out)
// one slot apart with a workgroup barrier between slots, so in every slot one group issues the MFMAs and the other the
// VALU / LDS / store work.  Workgroups are persistent: workgroup w walks a contiguous run of the flattened
// (image, band, step) sequence (cfg 2: 5120 steps = 20 per CU exactly; 640 band-sized workgroups on 512 slots were 1.25
// rounds); a run that starts inside a band first recomputes the step before it without storing it, only to obtain the
// carried column.  Patches rotate through three LDS buffers: the group in M2(s) converts and stages patch(s+2) (fetched at
// the top of M1(s) as 16-byte loads: 6 VMEM instructions per thread instead of 21) into the buffer the other group's
// M2(s-1) released three slots earlier.  Pooled pixels leave as 16-byte stores straight from registers
// (v_permlane32_swap pairs the half-waves' 4-channel quads): no LDS staging.  ReLU is one packed signed-16-bit max with 0
// AFTER the 3x3 max.  Output is bit-identical to the other forms (tests compare all of them).
// Measured (cfg 2, tools/bench_stem.py, tools/stem_phases.py): 100.7 us against 106.9 us for the second form.  The slots
// do NOT shrink to the MFMA time: a wave's VALU instructions and its SIMD partner's MFMAs share the SIMD's issue, and the
// time per pair of steps stays ~ (252 MFMAs x 32 cycles) + ~6 cycles per VALU instruction of both waves (16.2 k cycles;
// moving 1.3 k cycles of address arithmetic and 15 VMEM issues out of M1 lengthened V1 / V2 by the same amount).  The
// stem is bound by that sum -- 8.1 k cycles of K-padded MFMA work (147 -> 224) plus ~1.3 k VALU instructions per pair --
// not by phase alignment; what the ping-pong form gains over the second form is the persistent, evenly divided schedule.
typedef short i16x2_t __attribute__((ext_vector_type(2)));
__device__ __forceinline__ uint32_t pk_max_i16(uint32_t a, uint32_t b) {
    return __builtin_bit_cast(uint32_t, __builtin_elementwise_max(__builtin_bit_cast(i16x2_t, a), __builtin_bit_cast(i16x2_t, b)));
}

template <bool U8>
__global__ __launch_bounds__(512) void stem_pool3_kernel(const void* xv_arg, FrameMean mean, int B, int N, int H, int W,
                                                         const uint16_t* __restrict__ wpk, const float* __restrict__ scale,
                                                         const float* __restrict__ shift, uint16_t* __restrict__ y, int steps_per_wg,
                                                         int total_steps) {
    constexpr int COUT = 128, BAND = 8;
    const void* const __restrict__ xv = w2c_resolve(xv_arg);       // indirect operand: the frames' address may come from a pointer slot
    constexpr int FP_ROWS = fp_rows<BAND>(), FP_BYTES = fp_bytes<BAND>();
    constexpr int CT = COUT / 32, NT = 64 * CT;                // NT = threads per GROUP
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int grp = wave >> 2, ct = wave & 3;                  // waves ct and ct + 4 share a SIMD (cyclic wave -> SIMD placement)
    const int gtid = tid & (NT - 1);
    const int l31 = lane & 31, lhi = lane >> 5;
    const int Wo = W >> 1, Hp = H >> 2, Wp = W >> 2;
    const int nxs = Wo >> 5, nbands = (H >> 1) / BAND;
    char* const carry = smem + 3 * FP_BYTES + ct * 512;        // shared by the two groups' wave ct
    char* const ssb = carry + 256;

    // ---- this workgroup's run of steps ----
    int f_begin = blockIdx.x * steps_per_wg;
    const int f_end = min(f_begin + steps_per_wg, total_steps);
    if (f_begin >= f_end) return;                              // (whole workgroup)
    const bool warm = (f_begin % nxs) != 0;                    // starts inside a band: one unstored step for the carry
    if (warm) --f_begin;
    const int nsteps = f_end - f_begin;
    const int my_n = grp == 0 ? (nsteps + 1) >> 1 : nsteps >> 1;

    bf16x8_t wf[7][2];
    {
        const uint16_t* wrow = wpk + (size_t)(ct * 32 + l31) * 224;
#pragma unroll
        for (int ky = 0; ky < 7; ++ky)
#pragma unroll
            for (int ks = 0; ks < 2; ++ks)
                wf[ky][ks] = *reinterpret_cast<const bf16x8_t*>(wrow + ky * 32 + (ks * 2 + lhi) * 8);
        if (scale[ct * 32 + l31] < 0.f) {                      // negative BN scale: negate the weights, use |scale| (exact)
#pragma unroll
            for (int ky = 0; ky < 7; ++ky)
#pragma unroll
                for (int ks = 0; ks < 2; ++ks) {
                    u32x4_t w = __builtin_bit_cast(u32x4_t, wf[ky][ks]);
                    w ^= u32x4_t{0x80008000u, 0x80008000u, 0x80008000u, 0x80008000u};
                    wf[ky][ks] = __builtin_bit_cast(bf16x8_t, w);
                }
        }
    }
    if (grp == 0 && lane < 32) {
        const int q = lane >> 3, hh = (lane >> 2) & 1, k = lane & 3;
        reinterpret_cast<float*>(ssb)[(q * 2 + hh) * 4 + k] = fabsf(scale[ct * 32 + 8 * q + 4 * hh + k]);
        reinterpret_cast<float*>(ssb + 128)[(q * 2 + hh) * 4 + k] = shift[ct * 32 + 8 * q + 4 * hh + k];
    }
    const char* const ssw = ssb + lhi * 16;
    if constexpr (U8) {                                        // loader transform table: [c = B, G, R][256] bf16
        uint16_t* const lw = reinterpret_cast<uint16_t*>(smem + 3 * FP_BYTES + 4 * 512);
        for (int e = tid; e < 768; e += 512) {
            const int c = e >> 8, v = e & 255;
            lw[e] = f32_to_bf16((float)(((double)v - mean.m[c]) / 255.0));
        }
        __syncthreads();                                       // the prologue stages a patch through the table
    }

    // decode flattened step f -> image, band, column step
    struct Step { int img, oy0, ox0; };
    auto decode = [&](int f) {
        Step st;
        const int xs = f % nxs, t = f / nxs;
        const int band = t % nbands;
        st.img = t / nbands; st.oy0 = band * BAND; st.ox0 = xs * 32;
        return st;
    };
    // f32 frames: the same patch fetched as 16-byte loads.  An item = 4 consecutive patch columns of one row, all three
    // colour planes (3 x global_load_dwordx4, dword-aligned) -> 4 pixels x [c0 c1 c2 0] bf16 = two ds_write_b128.
    // 6 VMEM instructions per thread and step instead of 21: every VMEM issue between MFMAs costs the matrix slot ~100
    // cycles (measured: M1 4.45 k cycles for 2.24 k of MFMA with the 21 scalar loads in it).  The only items that straddle
    // the image edge are column group 0 of a band's first step (ix = -3..0) and groups 16/17 of its last step; their load
    // is moved inside the row and the lanes shifted at staging time (codes 1 / 2), so no load ever leaves the tensor.
    constexpr int NI4 = FP_ROWS * (FP_COLS / 4);
    constexpr int FILL4 = (NI4 + NT - 1) / NT;
    typedef float f32x4_u __attribute__((ext_vector_type(4), aligned(4)));
    f32x4_t pq[FILL4][3];
    unsigned pm4 = 0;                                          // per item: bits 0-3 element inside the image, bits 4-5 shift code
    bool pedge = false;                                        // (uniform) the fetched patch touches the image border
    auto load_patch4 = [&](int f) {
        const Step st = decode(f);
        pedge = st.ox0 == 0 || st.ox0 + 32 >= Wo || st.oy0 == 0 || st.oy0 + BAND >= (H >> 1);
        const int agent = st.img / B, b = st.img - agent * B;
        const float* xin = reinterpret_cast<const float*>(xv) + ((size_t)b * 3 * N + 3 * agent) * H * W;
        const int ix0 = 2 * st.ox0 - 3, iy0 = 2 * st.oy0 - 5;
        pm4 = 0;
#pragma unroll
        for (int i2 = 0; i2 < FILL4; ++i2) {
            const int it = gtid + i2 * NT;
            const int r = it / (FP_COLS / 4), c4 = it - r * (FP_COLS / 4);
            const int iy = iy0 + r;
            const bool rok = (it < NI4) & ((unsigned)iy < (unsigned)H);
            const int ixs = ix0 + 4 * c4;
            const int ixl = min(max(ixs, 0), W - 4);
            const int sh = ixl - ixs;
            const unsigned o = rok ? (unsigned)(iy * W + ixl) : (unsigned)ixl;
            pq[i2][0] = *reinterpret_cast<const f32x4_u*>(xin + o);
            pq[i2][1] = *reinterpret_cast<const f32x4_u*>(xin + (size_t)H * W + o);
            pq[i2][2] = *reinterpret_cast<const f32x4_u*>(xin + 2 * (size_t)H * W + o);
            unsigned m = 0;
#pragma unroll
            for (int k = 0; k < 4; ++k) m |= (rok & ((unsigned)(ixs + k) < (unsigned)W)) ? (1u << k) : 0u;
            m |= sh == 3 ? 0x10u : (sh == -1 ? 0x20u : 0u);
            pm4 |= m << (8 * i2);
        }
    };
    auto store_patch4 = [&](uint2* patch) {
        if (!pedge) {                                          // interior patch: nothing to mask or shift
#pragma unroll
            for (int i2 = 0; i2 < FILL4; ++i2) {
                const int it = gtid + i2 * NT;
                if (it < NI4) {
                    uint4* d = reinterpret_cast<uint4*>(patch + it * 4);
                    const f32x4_t a = pq[i2][0], b2 = pq[i2][1], c = pq[i2][2];
                    d[0] = make_uint4(pack_bf16x2(a[0], b2[0]), pack_bf16x2(c[0], 0.f), pack_bf16x2(a[1], b2[1]), pack_bf16x2(c[1], 0.f));
                    d[1] = make_uint4(pack_bf16x2(a[2], b2[2]), pack_bf16x2(c[2], 0.f), pack_bf16x2(a[3], b2[3]), pack_bf16x2(c[3], 0.f));
                }
            }
            return;
        }
#pragma unroll
        for (int i2 = 0; i2 < FILL4; ++i2) {
            const int it = gtid + i2 * NT;
            const unsigned m = pm4 >> (8 * i2);
            const bool c1 = (m & 0x10u) != 0, c2 = (m & 0x20u) != 0;
            uint32_t w0[4], w1[4];
            float e[3][4];
#pragma unroll
            for (int pl = 0; pl < 3; ++pl) {
                const f32x4_t v = pq[i2][pl];
                e[pl][0] = c2 ? v[1] : v[0];
                e[pl][1] = c2 ? v[2] : v[1];
                e[pl][2] = c2 ? v[3] : v[2];
                e[pl][3] = c1 ? v[0] : v[3];
            }
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const bool ok = (m >> k) & 1u;
                w0[k] = ok ? pack_bf16x2(e[0][k], e[1][k]) : 0u;
                w1[k] = ok ? pack_bf16x2(e[2][k], 0.f) : 0u;
            }
            if (it < NI4) {
                uint4* d = reinterpret_cast<uint4*>(patch + it * 4);      // (r * 72 + 4 c4) uint2 = it * 4
                d[0] = make_uint4(w0[0], w1[0], w0[1], w1[1]);
                d[1] = make_uint4(w0[2], w1[2], w0[3], w1[3]);
            }
        }
    };
    // u8 camera frames ([B][N][H][W][3] RGB): an item's 4 pixels are 12 consecutive bytes; with W % 64 == 0 they always sit at
    // byte 3 of a dword-aligned 16-byte window -> ONE global_load_dwordx4 per item (2 per thread-step; the per-byte form issued
    // 21), and the loader's transform (airsim_loader.py:521-527: RGB -> BGR, float64 (v - mean) / 255, f32) comes from a
    // 3 x 256-entry bf16 table in LDS built at kernel start with exactly that arithmetic (bit-identical), not from ~15 f64
    // instructions per element.  Edge items: the window is moved inside the row, b0 = byte of pixel 0 inside it (3 normally,
    // -9 for the left-edge item whose only valid pixel is its last, 7 for the right-edge item).
    const uint16_t* const lut = reinterpret_cast<const uint16_t*>(smem + 3 * FP_BYTES + 4 * 512);
    u32x4_t pw8[FILL4];
    auto load_patch8 = [&](int f) {
        const Step st = decode(f);
        const int agent = st.img / B, b = st.img - agent * B;
        const uint8_t* xin8 = reinterpret_cast<const uint8_t*>(xv) + ((size_t)b * N + agent) * H * W * 3;
        const int ix0 = 2 * st.ox0 - 3, iy0 = 2 * st.oy0 - 5;
        pedge = st.ox0 == 0 || st.ox0 + 32 >= Wo || st.oy0 == 0 || st.oy0 + BAND >= (H >> 1);
        pm4 = 0;
#pragma unroll
        for (int i2 = 0; i2 < FILL4; ++i2) {
            const int it = gtid + i2 * NT;
            const int r = it / (FP_COLS / 4), c4 = it - r * (FP_COLS / 4);
            const int iy = iy0 + r;
            const bool rok = (it < NI4) & ((unsigned)iy < (unsigned)H);
            const int ixs = ix0 + 4 * c4;
            const int win = min(max(3 * ixs - 3, 0), 3 * W - 16);
            const int b0 = 3 * ixs - win;
            const unsigned o = (rok ? (unsigned)(iy * W * 3) : 0u) + (unsigned)win;
            pw8[i2] = *reinterpret_cast<const u32x4_t*>(xin8 + o);
            unsigned m = 0;
#pragma unroll
            for (int k = 0; k < 4; ++k) m |= (rok & ((unsigned)(ixs + k) < (unsigned)W)) ? (1u << k) : 0u;
            m |= b0 == -9 ? 0x10u : (b0 == 7 ? 0x20u : 0u);
            pm4 |= m << (8 * i2);
        }
    };
    auto store_patch8 = [&](uint2* patch) {
#pragma unroll
        for (int i2 = 0; i2 < FILL4; ++i2) {
            const int it = gtid + i2 * NT;
            const u32x4_t d = pw8[i2];
            // pixel k = bytes 3 + 3k .. of the window (R, G, B in the low three bytes of px[k])
            uint32_t px[4] = {__builtin_amdgcn_alignbyte(d[1], d[0], 3), __builtin_amdgcn_alignbyte(d[2], d[1], 2),
                              __builtin_amdgcn_alignbyte(d[3], d[2], 1), d[3]};
            unsigned m = 0xFu;
            if (pedge) {
                m = pm4 >> (8 * i2);
                if (m & 0x10u) px[3] = d[0];                                      // left edge: pixel 3 = bytes 0..2
                if (m & 0x20u) {                                                  // right edge: pixels 0..2 = bytes 7, 10, 13
                    px[0] = __builtin_amdgcn_alignbyte(d[2], d[1], 3);
                    px[1] = __builtin_amdgcn_alignbyte(d[3], d[2], 2);
                    px[2] = d[3] >> 8;
                }
            }
            uint32_t w0[4], w1[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const uint32_t rr = px[k] & 255u, gg = (px[k] >> 8) & 255u, bb = (px[k] >> 16) & 255u;
                const bool ok = (m >> k) & 1u;
                w0[k] = ok ? ((uint32_t)lut[bb] | ((uint32_t)lut[256 + gg] << 16)) : 0u;
                w1[k] = ok ? (uint32_t)lut[512 + rr] : 0u;
            }
            if (it < NI4) {
                uint4* dd = reinterpret_cast<uint4*>(patch + it * 4);
                dd[0] = make_uint4(w0[0], w1[0], w0[1], w1[1]);
                dd[1] = make_uint4(w0[2], w1[2], w0[3], w1[3]);
            }
        }
    };
    auto fetch = [&](int f) { if constexpr (U8) load_patch8(f); else load_patch4(f); };
    auto stage = [&](uint2* patch) { if constexpr (U8) store_patch8(patch); else store_patch4(patch); };
    auto slot_barrier = [&]() {
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");     // this slot's LDS writes are done (global loads stay in flight)
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
    };

    // ---- prologue: each group stages the patch of its first step ----
    if (grp < nsteps) {                                        // (group B idles when the run is a single step)
        fetch(f_begin + grp);
        stage(reinterpret_cast<uint2*>(smem + grp * FP_BYTES));
    }
    slot_barrier();
    if (grp == 1) slot_barrier();                              // group B runs one slot behind
#ifdef W2C_STEM_TIMING
    unsigned long long t_acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    long long t_prev = clock64();
#endif

    for (int j = 0; j < my_n; ++j) {
        const int sl = grp + 2 * j;                            // local step index
        const Step st = decode(f_begin + sl);
        const int img = st.img, oy0 = st.oy0, ox0 = st.ox0;
        const bool do_store = !(warm && sl == 0);
        const uint2* patch = reinterpret_cast<const uint2*>(smem + (sl % 3) * FP_BYTES);
        const bool nxt = sl + 2 < nsteps;

        auto finish_rows = [&](int pr0, uint32_t (&pk2)[2][8]) {
            uint4 cin[4];
            if (l31 == 0) {
#pragma unroll
                for (int i = 0; i < 4; ++i)
                    cin[i] = ox0 == 0 ? make_uint4(0, 0, 0, 0) : *reinterpret_cast<const uint4*>(carry + lhi * 128 + pr0 * 32 + i * 16);
            }
            asm volatile("" ::: "memory");
            if (l31 == 31) {
#pragma unroll
                for (int pr = 0; pr < 2; ++pr) {
                    *reinterpret_cast<uint4*>(carry + lhi * 128 + (pr0 + pr) * 32) = make_uint4(pk2[pr][0], pk2[pr][1], pk2[pr][2], pk2[pr][3]);
                    *reinterpret_cast<uint4*>(carry + lhi * 128 + (pr0 + pr) * 32 + 16) = make_uint4(pk2[pr][4], pk2[pr][5], pk2[pr][6], pk2[pr][7]);
                }
            }
#pragma unroll
            for (int pr = 0; pr < 2; ++pr) {
#pragma unroll
                for (int d = 0; d < 8; ++d) {
                    uint32_t left = lane_from_left(pk2[pr][d]);
                    if (l31 == 0) {
                        const uint4 c = cin[pr * 2 + (d >> 2)];
                        left = (d & 3) == 0 ? c.x : (d & 3) == 1 ? c.y : (d & 3) == 2 ? c.z : c.w;
                    }
                    const uint32_t right = lane_from_right(pk2[pr][d]);
                    // bf16 patterns ordered as SIGNED 16-bit integers: a positive beats every negative and positives order
                    // like their values, so max-then-ReLU (one packed max with 0 per dword) equals the ReLU-then-max of the
                    // other forms bit for bit (an all-negative window gives some negative -> 0), for half the VALU work
                    pk2[pr][d] = pk_max_i16(pk_max_i16(pk_max_i16(left, pk2[pr][d]), right), 0u);
                }
            }
            // even lanes hold pooled column l31/2: dwords (2q, 2q+1) = channels 8q + 4 lhi .. +3.  v_permlane32_swap pairs the
            // two half-waves so that a lane ends up with 8 consecutive channels (16 B) of its pixel -- lanes 0-31 those of
            // channel group q0, lanes 32-63 those of q0 + 1 -- and the pooled pixels go out as 16-byte stores without the
            // LDS staging round trip (8 ds_write_b64 + 2 ds_read_b128 + two waits per pair of pooled rows: 14 us of the 103).
            if (do_store) {
#pragma unroll
                for (int pr = 0; pr < 2; ++pr) {
                    const int py = (oy0 >> 1) + pr0 + pr, px = (ox0 >> 1) + (l31 >> 1);
                    uint16_t* const dst = y + (((size_t)img * Hp + py) * Wp + px) * COUT + ct * 32 + lhi * 8;
#pragma unroll
                    for (int qp = 0; qp < 2; ++qp) {
                        const auto a = __builtin_amdgcn_permlane32_swap(pk2[pr][4 * qp], pk2[pr][4 * qp + 2], false, false);
                        const auto b = __builtin_amdgcn_permlane32_swap(pk2[pr][4 * qp + 1], pk2[pr][4 * qp + 3], false, false);
                        if ((l31 & 1) == 0) *reinterpret_cast<uint4*>(dst + qp * 16) = make_uint4(a[0], b[0], a[1], b[1]);
                    }
                }
            }
        };
        float keep[16];
        auto conv_rows = [&](auto first_tag, auto count_tag, f32x16_t* acc) {
            constexpr int R0 = decltype(first_tag)::value, RN = decltype(count_tag)::value;
#pragma unroll
            for (int m = 0; m < RN; ++m) {
#pragma unroll
                for (int ky = 0; ky < 7; ++ky) {
                    const uint2* prow = patch + (2 * (R0 + m) + ky) * FP_COLS + 2 * l31 + 2 * lhi;
#pragma unroll
                    for (int ks = 0; ks < 2; ++ks) {
                        const bf16x8_t pb = *reinterpret_cast<const bf16x8_t*>(prow + ks * 4);
                        if (ky == 0 && ks == 0)
                            acc[m] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf[ky][ks], pb, f32x16_t{0.f}, 0, 0, 0);
                        else
                            acc[m] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf[ky][ks], pb, acc[m], 0, 0, 0);
                    }
                }
            }
        };
        constexpr float NEG_INF = -3.0e38f;
        {
            uint32_t pk[2][8];
            f32x16_t acc[5];
            // ---- slot M1 ----
            if (nxt)
                fetch(f_begin + sl + 2);                       // consumed in M2, two slots from now
            conv_rows(std::integral_constant<int, 0>{}, std::integral_constant<int, 5>{}, acc);
            STEM_T(0);
            slot_barrier();
            STEM_T(1);
            // ---- slot V1 ----
            const bool in0 = oy0 > 0;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const f32x4_t sc = *reinterpret_cast<const f32x4_t*>(ssw + q * 32), sh = *reinterpret_cast<const f32x4_t*>(ssw + 128 + q * 32);
                float v0[4], v1[4];
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const int e = 4 * q + k;
                    keep[e] = acc[4][e];
                    const float m0 = fmaxf(fmaxf(in0 ? acc[0][e] : NEG_INF, acc[1][e]), acc[2][e]);
                    const float m1 = fmaxf(fmaxf(acc[2][e], acc[3][e]), acc[4][e]);
                    v0[k] = m0 * sc[k] + sh[k];
                    v1[k] = m1 * sc[k] + sh[k];
                }
                pk[0][2 * q] = pack_bf16x2(v0[0], v0[1]);
                pk[0][2 * q + 1] = pack_bf16x2(v0[2], v0[3]);
                pk[1][2 * q] = pack_bf16x2(v1[0], v1[1]);
                pk[1][2 * q + 1] = pack_bf16x2(v1[2], v1[3]);
            }
            finish_rows(0, pk);
            STEM_T(2);
            slot_barrier();
            STEM_T(3);
        }
        {
            uint32_t pk[2][8];
            f32x16_t acc[4];
            // ---- slot M2 (the shortest matrix slot: the patch of this group's next step is converted and staged under it;
            // its buffer was last read by the other group's M2 three slots ago) ----
            if (nxt)
                stage(reinterpret_cast<uint2*>(smem + ((sl + 2) % 3) * FP_BYTES));
            conv_rows(std::integral_constant<int, 5>{}, std::integral_constant<int, 4>{}, acc);
            STEM_T(4);
            slot_barrier();
            STEM_T(5);
            // ---- slot V2 ----
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const f32x4_t sc = *reinterpret_cast<const f32x4_t*>(ssw + q * 32), sh = *reinterpret_cast<const f32x4_t*>(ssw + 128 + q * 32);
                float v2[4], v3[4];
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const int e = 4 * q + k;
                    const float m2 = fmaxf(fmaxf(keep[e], acc[0][e]), acc[1][e]);
                    const float m3 = fmaxf(fmaxf(acc[1][e], acc[2][e]), acc[3][e]);
                    v2[k] = m2 * sc[k] + sh[k];
                    v3[k] = m3 * sc[k] + sh[k];
                }
                pk[0][2 * q] = pack_bf16x2(v2[0], v2[1]);
                pk[0][2 * q + 1] = pack_bf16x2(v2[2], v2[3]);
                pk[1][2 * q] = pack_bf16x2(v3[0], v3[1]);
                pk[1][2 * q + 1] = pack_bf16x2(v3[2], v3[3]);
            }
            finish_rows(2, pk);
            STEM_T(6);
            slot_barrier();
            STEM_T(7);
        }
    }
#ifdef W2C_STEM_TIMING
    if (tid == 0)
        for (int i = 0; i < 8; ++i) atomicAdd(&g_stem_phase[i], t_acc[i]);
#endif
    // ---- barrier parity: both groups must pass the same number of workgroup barriers ----
    {
        const int nA = (nsteps + 1) >> 1, nB = nsteps >> 1;
        const int mine = grp == 0 ? 4 * nA : 1 + 4 * nB;
        const int total_b = max(4 * nA, 1 + 4 * nB);
        for (int i = mine; i < total_b; ++i) __builtin_amdgcn_s_barrier();
    }
}

__global__ __launch_bounds__(256) void maxpool3x3s2_kernel(const uint16_t* __restrict__ x, int M, int H, int W, int C,
                                                           uint16_t* __restrict__ y) {
    const int Ho = H >> 1, Wo = W >> 1, CG = C >> 3;
    const size_t total = (size_t)M * Ho * Wo * CG;
    for (size_t id = (size_t)blockIdx.x * 256 + threadIdx.x; id < total; id += (size_t)gridDim.x * 256) {
        const int cg = (int)(id % CG);
        size_t t = id / CG;
        const int ox = (int)(t % Wo); t /= Wo;
        const int oy = (int)(t % Ho);
        const int m = (int)(t / Ho);
        float best[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) best[e] = -3.0e38f;
#pragma unroll
        for (int dy = -1; dy <= 1; ++dy) {
            const int iy = 2 * oy + dy;
            if (iy < 0 || iy >= H) continue;
#pragma unroll
            for (int dx = -1; dx <= 1; ++dx) {
                const int ix = 2 * ox + dx;
                if (ix < 0 || ix >= W) continue;
                const uint4 v = *reinterpret_cast<const uint4*>(x + (((size_t)m * H + iy) * W + ix) * C + cg * 8);
                const uint32_t wv[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    best[2 * e] = fmaxf(best[2 * e], bf16_to_f32((uint16_t)(wv[e] & 0xFFFFu)));
                    best[2 * e + 1] = fmaxf(best[2 * e + 1], bf16_to_f32((uint16_t)(wv[e] >> 16)));
                }
            }
        }
        uint4 o;
        o.x = pack_bf16x2(best[0], best[1]); o.y = pack_bf16x2(best[2], best[3]);
        o.z = pack_bf16x2(best[4], best[5]); o.w = pack_bf16x2(best[6], best[7]);
        *reinterpret_cast<uint4*>(y + (((size_t)m * Ho + oy) * Wo + ox) * C + cg * 8) = o;
    }
}

template <int COUT, bool TRAIN = false>
int launch_stem(const float* x, int B, int N, int H, int W, const uint16_t* w, const float* scale,
                const float* shift, uint16_t* y, hipStream_t s) {
    constexpr int lds = PATCH_BYTES + 256 * COUT * 2;
    static std::atomic<unsigned long long> attr_mask{0};
    int dev = 0;
    (void)hipGetDevice(&dev);
    if (!((attr_mask.load(std::memory_order_acquire) >> (dev & 63)) & 1ull)) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&stem_kernel<COUT, TRAIN>),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, lds);
        attr_mask.fetch_or(1ull << (dev & 63), std::memory_order_release);
    }
    dim3 grid((H / 2) / 8, N * B);
    hipLaunchKernelGGL((stem_kernel<COUT, TRAIN>), grid, dim3(256), lds, s, x, B, N, H, W, w, scale, shift, y);
    return w2c_launch_status();
}

}  // namespace

#ifdef W2C_STEM_TIMING
extern "C" int w2c_debug_stem_phases(unsigned long long* out8, int reset) {
    if (out8 && hipMemcpyFromSymbol(out8, HIP_SYMBOL(g_stem_phase), 64) != hipSuccess) return W2C_E_LAUNCH;
    if (reset) { unsigned long long z[8] = {0}; if (hipMemcpyToSymbol(HIP_SYMBOL(g_stem_phase), z, 64) != hipSuccess) return W2C_E_LAUNCH; }
    return W2C_OK;
}
#endif

extern "C" int w2c_stem_conv7x7_bn_relu(const float* x, int B, int N, int H, int W,
                                        const uint16_t* w, const float* scale, const float* shift, int Cout,
                                        uint16_t* y, w2c_stream_t stream) {
    w2c_clear_error();
    if (!x || !w || !scale || !shift || !y) return W2C_E_ARG;
    if (B <= 0 || N <= 0 || H <= 0 || W <= 0 || (H % 16) != 0 || (W % 64) != 0) return W2C_E_ARG;
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    if (Cout == 128) return launch_stem<128>(x, B, N, H, W, w, scale, shift, y, s);
    if (Cout == 64) return launch_stem<64>(x, B, N, H, W, w, scale, shift, y, s);
    return W2C_E_ARG;
}

extern "C" int w2c_stem_conv7x7_train_bf16(const uint16_t* x_nhwc3, int M, int H, int W, const uint16_t* w, uint16_t* y,
                                           w2c_stream_t stream) {
    w2c_clear_error();
    if (!x_nhwc3 || !w || !y) return W2C_E_ARG;
    if (M <= 0 || H <= 0 || W <= 0 || (H % 16) != 0 || (W % 64) != 0) return W2C_E_ARG;
    return launch_stem<64, true>(reinterpret_cast<const float*>(x_nhwc3), M, 1, H, W, w, nullptr, nullptr, y,
                                 reinterpret_cast<hipStream_t>(stream));
}

template <int COUT, bool U8, int BAND, int NW>
static int launch_stem_pool_band(const void* x, FrameMean mean, int B, int N, int H, int W, const uint16_t* w, const float* scale,
                                 const float* shift, uint16_t* y, hipStream_t s) {
    constexpr int lds = fp_bytes<BAND>() + (BAND + 1) * 33 * COUT * 2;
    static std::atomic<unsigned long long> attr_mask{0};
    int dev = 0;
    (void)hipGetDevice(&dev);
    if (!((attr_mask.load(std::memory_order_acquire) >> (dev & 63)) & 1ull)) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&stem_pool_kernel<COUT, U8, BAND, NW>),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, lds);
        attr_mask.fetch_or(1ull << (dev & 63), std::memory_order_release);
    }
    dim3 grid((H / 2) / BAND, N * B);
    hipLaunchKernelGGL((stem_pool_kernel<COUT, U8, BAND, NW>), grid, dim3(64 * NW), lds, s, x, mean, B, N, H, W, w, scale, shift, y);
    return w2c_launch_status();
}

template <int COUT, bool U8>
static int launch_stem_pool2(const void* x, FrameMean mean, int B, int N, int H, int W, const uint16_t* w, const float* scale,
                             const float* shift, uint16_t* y, hipStream_t s) {
    constexpr int lds = 2 * fp_bytes<8>() + (COUT / 32) * (64 * 80 + 256 + 256);
    dim3 grid((H / 2) / 8, N * B);
    hipLaunchKernelGGL((stem_pool2_kernel<COUT, U8>), grid, dim3(COUT * 2), lds, s, x, mean, B, N, H, W, w, scale, shift, y);
    return w2c_launch_status();
}

template <bool U8>
static int launch_stem_pool3(const void* x, FrameMean mean, int B, int N, int H, int W, const uint16_t* w, const float* scale,
                             const float* shift, uint16_t* y, hipStream_t s) {
    constexpr int lds = 3 * fp_bytes<8>() + 4 * 512 + (U8 ? 1536 : 0);
    static std::atomic<unsigned long long> attr_mask{0};
    static int n_cu[64] = {0};
    int dev = 0;
    (void)hipGetDevice(&dev);
    if (!((attr_mask.load(std::memory_order_acquire) >> (dev & 63)) & 1ull)) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&stem_pool3_kernel<U8>), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
        int cu = 0;
        if (hipDeviceGetAttribute(&cu, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cu <= 0) cu = 256;
        n_cu[dev & 63] = cu;
        attr_mask.fetch_or(1ull << (dev & 63), std::memory_order_release);
    }
    const int total = N * B * ((H / 2) / 8) * ((W / 2) / 32);
    int wgs = n_cu[dev & 63];
    if (const int v = w2c_option(W2C_OPT_STEM_WGS); v > 0) wgs = v;                             // tests: odd runs, mid-band starts
    int spw = (total + wgs - 1) / wgs;
    if (spw < 2) spw = 2;                                      // both groups busy
    wgs = (total + spw - 1) / spw;
    hipLaunchKernelGGL((stem_pool3_kernel<U8>), dim3(wgs), dim3(512), lds, s, x, mean, B, N, H, W, w, scale, shift, y, spw, total);
    return w2c_launch_status();
}

template <int COUT, bool U8>
static int launch_stem_pool(const void* x, FrameMean mean, int B, int N, int H, int W, const uint16_t* w, const float* scale,
                            const float* shift, uint16_t* y, hipStream_t s) {
    const int band = w2c_option(W2C_OPT_STEM_BAND);
    // measured (tools/bench_stem.py, cfg 2): BAND 8 143 us, BAND 4 154 us -- the second resident workgroup does not pay
    // for its 5/4 recompute; W2C_STEM_BAND=4 keeps the A/B reproducible.
    // 12 waves (conv rows split 3/3/3 instead of 4/5): measured 159 us vs 145 us for 8 waves -- kept for the A/B only
    // Cout = 128 (both trunks side by side): the register-pooling form (measured 110 us vs 141 us at cfg 2; ablations: MFMA +
    // fragment reads 54 us, BN/max/pack/staging VALU 53 us, patch I/O 15 us -- still serial inside a wave, two waves per
    // SIMD overlap 1.5x).  Cout = 64 (Single_agent): the LDS-pooling form is faster (95 vs 104 us).  W2C_STEM_FORM=1|2 forces.
    // Measured and dropped: a 3-waves/SIMD build (4 accumulation passes, 168 registers: 40 spilled dwords -> 186 us), 4-row
    // bands for a finer tail (1280 half-size workgroups: 112 us), staggered starts of co-resident workgroups (no change).
    const int form = w2c_option(W2C_OPT_STEM_FORM);
    // Cout = 128 default: the persistent ping-pong form (100.7 vs 106.9 us for form 2 at cfg 2)
    if constexpr (COUT == 128) {
        if ((form == 3 || form == 0) && H % 16 == 0) return launch_stem_pool3<U8>(x, mean, B, N, H, W, w, scale, shift, y, s);
    }
    if ((form == 2 || (form == 0 && COUT == 128)) && H % 16 == 0)
        return launch_stem_pool2<COUT, U8>(x, mean, B, N, H, W, w, scale, shift, y, s);
    const int nw = w2c_option(W2C_OPT_STEM_WAVES);
    if (band == 4) return launch_stem_pool_band<COUT, U8, 4, 8>(x, mean, B, N, H, W, w, scale, shift, y, s);
    if (nw == 12 && COUT == 128) return launch_stem_pool_band<COUT, U8, 8, (COUT == 128 ? 12 : 8)>(x, mean, B, N, H, W, w, scale, shift, y, s);
    return launch_stem_pool_band<COUT, U8, 8, 8>(x, mean, B, N, H, W, w, scale, shift, y, s);
}

extern "C" int w2c_stem_conv7x7_bn_relu_maxpool(const float* x, int B, int N, int H, int W,
                                                const uint16_t* w, const float* scale, const float* shift, int Cout,
                                                uint16_t* y, w2c_stream_t stream) {
    w2c_clear_error();
    if (!x || !w || !scale || !shift || !y) return W2C_E_ARG;
    if (B <= 0 || N <= 0 || H <= 0 || W <= 0 || (H % 16) != 0 || (W % 64) != 0) return W2C_E_ARG;
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    FrameMean none = {{0.0, 0.0, 0.0}};
    if (Cout == 128) return launch_stem_pool<128, false>(x, none, B, N, H, W, w, scale, shift, y, s);
    if (Cout == 64) return launch_stem_pool<64, false>(x, none, B, N, H, W, w, scale, shift, y, s);
    return W2C_E_ARG;
}

extern "C" int w2c_stem_u8_conv7x7_bn_relu_maxpool(const uint8_t* frames, double mean_b, double mean_g, double mean_r,
                                                   int B, int N, int H, int W,
                                                   const uint16_t* w, const float* scale, const float* shift, int Cout,
                                                   uint16_t* y, w2c_stream_t stream) {
    w2c_clear_error();
    if (!frames || !w || !scale || !shift || !y) return W2C_E_ARG;
    if (B <= 0 || N <= 0 || H <= 0 || W <= 0 || (H % 16) != 0 || (W % 64) != 0) return W2C_E_ARG;
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    FrameMean mean = {{mean_b, mean_g, mean_r}};
    if (Cout == 128) return launch_stem_pool<128, true>(frames, mean, B, N, H, W, w, scale, shift, y, s);
    if (Cout == 64) return launch_stem_pool<64, true>(frames, mean, B, N, H, W, w, scale, shift, y, s);
    return W2C_E_ARG;
}

extern "C" int w2c_maxpool3x3s2(const uint16_t* x, int M, int H, int W, int C, uint16_t* y, w2c_stream_t stream) {
    w2c_clear_error();
    if (!x || !y || M <= 0 || H <= 0 || W <= 0 || (H & 1) || (W & 1) || C <= 0 || (C % 8) != 0) return W2C_E_ARG;
    const size_t total = (size_t)M * (H / 2) * (W / 2) * (C / 8);
    size_t blocks = (total + 255) / 256;
    if (blocks > 256 * 16) blocks = 256 * 16;      // grid-stride beyond ~16 workgroups per CU
    hipLaunchKernelGGL(maxpool3x3s2_kernel, dim3((unsigned)blocks), dim3(256), 0,
                       reinterpret_cast<hipStream_t>(stream), x, M, H, W, C, y);
    return w2c_launch_status();
}
