// upsample.hip -- K9: bilinear x32 upsample with align_corners=False (simple_decoder,
// backbone.py:160: F.interpolate(pred, size=(32h, 32w), mode='bilinear', align_corners=False)),
// plus the two layout converters used at the module boundary / by the parity tests.
//
// HBM-write bound: 32*32 f32 outputs per input value (cfg 2: 231 MB written, 0.7 MB read).
// One workgroup = one (image, class) plane band of 32 output rows; the h x w source plane sits in
// LDS; every thread produces 4 consecutive x (one 16-B store), so a wave writes 1 KiB runs.
// Source index math follows ATen's area_pixel_compute_source_index:
//   src = (dst + 0.5) * (in/out) - 0.5, clamped at 0; i0 = floor(src), i1 = min(i0+1, in-1),
//   l1 = src - i0, l0 = 1 - l1;  out = l0y*(l0x*p00 + l1x*p01) + l1y*(l0x*p10 + l1x*p11).
#include "w2c_common.h"

namespace {

// one definition of the lerp (explicit fma placement, not left to -ffp-contract) shared by K9 and its argmax fusion
__device__ __forceinline__ float lerp2d(float ly0, float ly1, float lx0, float lx1, float p00, float p01, float p10, float p11) {
    const float top = __builtin_fmaf(lx1, p01, lx0 * p00);
    const float bot = __builtin_fmaf(lx1, p11, lx0 * p10);
    return __builtin_fmaf(ly1, bot, ly0 * top);
}

__global__ __launch_bounds__(256) void upsample32_kernel(const float* __restrict__ low, int h, int w, int lcs, int ncls,
                                                         float* out_arg) {
    float* const __restrict__ out = w2c_resolve(out_arg);       // indirect operand: the logits' address may come from a pointer slot
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float* plane = reinterpret_cast<float*>(smem);     // [h][w]
    const int H = h * 32, W = w * 32;
    const int band = blockIdx.x;                        // 32 output rows
    const int c = blockIdx.y, m = blockIdx.z;
    for (int i = threadIdx.x; i < h * w; i += 256) plane[i] = low[((size_t)m * h * w + i) * lcs + c];
    __syncthreads();
    const int xq = W >> 2;                              // float4 groups per output row
    float* obase = out + (((size_t)m * ncls + c) * H + (size_t)band * 32) * W;
    // Round 4: thread = 4 consecutive columns x a run of 16 output rows.  A 16-row half band (rows [16 k, 16 k + 16)) has ONE pair of
    // source rows (y0, y1) and a column quad ONE pair of source columns, so lerp2d's horizontal lerps (`top` at y0, `bot` at y1) are
    // formed once per thread -- 8 LDS reads and 8 lerps for 64 outputs -- and every output is lerp2d's last step alone:
    // fma(ly1, bot, ly0 * top).  Same operations in the same order as the per-pixel form: same bits.  (The per-pixel form issued 16 LDS
    // reads and ~40 VALU instructions per 16-byte store: SQ showed it stalled on issue, not on the memory system.)
    const int halves = 2 * xq;                          // (half band, column quad) items of this workgroup
    for (int id = threadIdx.x; id < halves; id += 256) {
        const int hb = id / xq, gx = id - hb * xq;
        const int oy0 = band * 32 + hb * 16;
        float sy = (oy0 + 0.5f) * 0.03125f - 0.5f;
        sy = sy < 0.f ? 0.f : sy;
        const int y0 = (int)sy;
        const int y1 = y0 + (y0 < h - 1 ? 1 : 0);
        float sx0 = (gx * 4 + 0.5f) * 0.03125f - 0.5f;
        sx0 = sx0 < 0.f ? 0.f : sx0;
        const int x0 = (int)sx0;
        const int x1 = x0 + (x0 < w - 1 ? 1 : 0);
        const float p00 = plane[y0 * w + x0], p01 = plane[y0 * w + x1], p10 = plane[y1 * w + x0], p11 = plane[y1 * w + x1];
        float top[4], bot[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            float sx = (gx * 4 + e + 0.5f) * 0.03125f - 0.5f;
            sx = sx < 0.f ? 0.f : sx;
            const float lx1 = sx - (float)x0, lx0 = 1.f - lx1;
            top[e] = __builtin_fmaf(lx1, p01, lx0 * p00);
            bot[e] = __builtin_fmaf(lx1, p11, lx0 * p10);
        }
        float* orow = obase + (size_t)(hb * 16) * W + gx * 4;
#pragma unroll
        for (int ry = 0; ry < 16; ++ry) {
            float syr = (oy0 + ry + 0.5f) * 0.03125f - 0.5f;
            syr = syr < 0.f ? 0.f : syr;
            const float ly1 = syr - (float)y0, ly0 = 1.f - ly1;
            f32x4_t o;
#pragma unroll
            for (int e = 0; e < 4; ++e) o[e] = __builtin_fmaf(ly1, bot[e], ly0 * top[e]);
            *reinterpret_cast<f32x4_t*>(orow + (size_t)ry * W) = o;
        }
    }
}

// K9 + the evaluator's class argmax (trainer.py:804 `outputs.data.max(1)[1]`) fused: the full-resolution f32
// logits (231 MB at cfg 2) are never written -- only one u8 label per pixel (5 MB).  Same source-index and lerp
// arithmetic as upsample32_kernel (lerp2d's three FMAs in its order), so labels == argmax over classes of its output, bit for
// bit; ties keep the lowest class index.
//
// Round 4 form (the first one took 47 us for 5 MB of labels: every workgroup staged the image's whole low-resolution block
// through LDS with uncoalesced 4-byte reads, and every pixel paid the full 2-D lerp per class).  Now
//   * thread = 4 consecutive output columns x ARG_ROWS = 8 output rows.  Both share their source cells: x0 changes only at
//     ox = 16 (mod 32), a multiple of 4, and an 8-row band aligned to 8 never straddles oy = 16 (mod 32) -- so the four corner
//     pixels are the same for the thread's 32 outputs;
//   * the corners' class vectors come straight from the NHWC low-resolution map: 4 corners x 3 x 16 B per thread, all in flight
//     together, L1 / L2 hits (the map is 0.7 MB); no LDS, no barrier;
//   * lerp2d = vertical lerp of the two HORIZONTAL lerps, and those do not depend on the row: top / bot are formed once per
//     (class, column) -- 4 operations for 8 rows -- and each output costs one mul + one FMA + the running argmax;
//   * labels are stored as one dword per row (a wave writes 256 contiguous bytes).
//
// CONF: the evaluator's confusion matrix (runningScore._fast_hist, metrics.py:99-108: bincount(n*gt + pred) over the
// pixels with 0 <= gt < n) fused behind the argmax -- each workgroup histograms its pixels in LDS (a wave whose 64
// lanes all hit one bin, the common case inside a segment, adds 64 with one atomic) and flushes its non-zero bins
// with one 64-bit global atomic each.  Integer atomics: the result is exact and order-independent.
constexpr int ARG_ROWS = 8;
constexpr int ARG_MAXC = 12;                        // classes handled by the register form (3 x 16 B per corner)
// MULTI: more than ARG_MAXC classes -- the classes are walked in chunks of ARG_MAXC with the rows' running (best, index) kept across
// chunks (64 more registers); the comparison order is still class-ascending, so ties keep the lowest index.
template <bool CONF, bool GT64, bool MULTI = false>
__global__ __launch_bounds__(256) void upsample32_argmax_kernel(const float* __restrict__ low, int h, int w, int lcs, int ncls,
                                                                uint8_t* labels_arg, const void* gt_arg,
                                                                unsigned long long* hist_arg) {
    uint8_t* const __restrict__ labels = w2c_resolve(labels_arg);   // indirect operands (caller-owned tensors of a captured forward)
    const void* const __restrict__ gt = w2c_resolve(gt_arg);
    unsigned long long* const __restrict__ hist = w2c_resolve(hist_arg);
    extern __shared__ __attribute__((aligned(16))) char smem[];
    unsigned* lh = reinterpret_cast<unsigned*>(smem);      // CONF: [ncls*ncls] workgroup histogram
    if (CONF) {
        for (int i = threadIdx.x; i < ncls * ncls; i += 256) lh[i] = 0;
        __syncthreads();
    }
    const int H = h * 32, W = w * 32;
    const int xq = W >> 2, nb = H / ARG_ROWS;
    const int m = blockIdx.y;
    const int items = nb * xq;                          // (band, column quad) pairs of this image, band-major
    for (int id = blockIdx.x * 256 + threadIdx.x; id < items; id += gridDim.x * 256) {
        const int band = id / xq, gx = id - band * xq;
        // source cells (ATen's area_pixel_compute_source_index, as in upsample32_kernel)
        float sy = (band * ARG_ROWS + 0.5f) * 0.03125f - 0.5f;
        sy = sy < 0.f ? 0.f : sy;
        const int y0 = (int)sy;
        const int y1 = y0 + (y0 < h - 1 ? 1 : 0);
        float sx0 = (gx * 4 + 0.5f) * 0.03125f - 0.5f;
        sx0 = sx0 < 0.f ? 0.f : sx0;
        const int x0 = (int)sx0;
        const int x1 = x0 + (x0 < w - 1 ? 1 : 0);
        const float* lm = low + (size_t)m * h * w * lcs;
        float lx1[4], lx0[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            float sx = (gx * 4 + e + 0.5f) * 0.03125f - 0.5f;
            sx = sx < 0.f ? 0.f : sx;
            lx1[e] = sx - (float)x0;
            lx0[e] = 1.f - lx1[e];
        }
        float bestr[MULTI ? ARG_ROWS : 1][4];
        int bir[MULTI ? ARG_ROWS : 1][4];
        if (MULTI) {
#pragma unroll
            for (int ry = 0; ry < ARG_ROWS; ++ry)
#pragma unroll
                for (int e = 0; e < 4; ++e) { bestr[MULTI ? ry : 0][e] = -INFINITY; bir[MULTI ? ry : 0][e] = 0; }
        }
        uint8_t* obase = labels ? labels + ((size_t)m * H + (size_t)band * ARG_ROWS) * W + gx * 4 : nullptr;
        // CONF: the ground-truth labels of the thread's 8 x 4 pixels, all loads in flight with the corner vectors (read one row at a
        // time next to their use they were 8 dependent round trips per thread: +16 us on the 12 us argmax)
        uint32_t gt8[CONF && !GT64 ? ARG_ROWS : 1];
        long long gt64[CONF && GT64 ? ARG_ROWS : 1][4];
        if (CONF) {
#pragma unroll
            for (int ry = 0; ry < ARG_ROWS; ++ry) {
                const size_t pix = ((size_t)m * H + band * ARG_ROWS + ry) * W + gx * 4;
                if (GT64) {
                    const long long* gp = reinterpret_cast<const long long*>(gt) + pix;
#pragma unroll
                    for (int e = 0; e < 4; ++e) gt64[GT64 ? ry : 0][e] = gp[e];
                } else {
                    gt8[GT64 ? 0 : ry] = *reinterpret_cast<const uint32_t*>(reinterpret_cast<const uint8_t*>(gt) + pix);
                }
            }
        }
        for (int cbase = 0; cbase < (MULTI ? ncls : 1); cbase += ARG_MAXC) {
        // the four corners' class vectors (lcs >= 4 ceil(ncls / 4) floats, 16-byte aligned rows: checked by the launcher)
        f32x4_t c00[3], c01[3], c10[3], c11[3];
#pragma unroll
        for (int v = 0; v < 3; ++v) {
            if (cbase + 4 * v < ncls) {
                c00[v] = *reinterpret_cast<const f32x4_t*>(lm + (size_t)(y0 * w + x0) * lcs + cbase + 4 * v);
                c01[v] = *reinterpret_cast<const f32x4_t*>(lm + (size_t)(y0 * w + x1) * lcs + cbase + 4 * v);
                c10[v] = *reinterpret_cast<const f32x4_t*>(lm + (size_t)(y1 * w + x0) * lcs + cbase + 4 * v);
                c11[v] = *reinterpret_cast<const f32x4_t*>(lm + (size_t)(y1 * w + x1) * lcs + cbase + 4 * v);
            } else {
                c00[v] = c01[v] = c10[v] = c11[v] = f32x4_t{0.f, 0.f, 0.f, 0.f};
            }
        }
        // horizontal lerps, once per (class, column): lerp2d's `top` and `bot`
        float top[ARG_MAXC][4], bot[ARG_MAXC][4];
#pragma unroll
        for (int c = 0; c < ARG_MAXC; ++c) {
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                top[c][e] = __builtin_fmaf(lx1[e], c01[c >> 2][c & 3], lx0[e] * c00[c >> 2][c & 3]);
                bot[c][e] = __builtin_fmaf(lx1[e], c11[c >> 2][c & 3], lx0[e] * c10[c >> 2][c & 3]);
            }
        }
        const bool last_chunk = !MULTI || cbase + ARG_MAXC >= ncls;
#pragma unroll
        for (int ry = 0; ry < ARG_ROWS; ++ry) {
            const int oy = band * ARG_ROWS + ry;
            float syr = (oy + 0.5f) * 0.03125f - 0.5f;
            syr = syr < 0.f ? 0.f : syr;
            const float ly1 = syr - (float)y0, ly0 = 1.f - ly1;
            float best[4];
            int bi[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                best[e] = MULTI ? bestr[MULTI ? ry : 0][e] : -INFINITY;
                bi[e] = MULTI ? bir[MULTI ? ry : 0][e] : 0;
            }
#pragma unroll
            for (int c = 0; c < ARG_MAXC; ++c) {
                if (cbase + c < ncls) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const float v = __builtin_fmaf(ly1, bot[c][e], ly0 * top[c][e]);
                        if (v > best[e]) { best[e] = v; bi[e] = cbase + c; }
                    }
                }
            }
            if (MULTI) {
#pragma unroll
                for (int e = 0; e < 4; ++e) { bestr[MULTI ? ry : 0][e] = best[e]; bir[MULTI ? ry : 0][e] = bi[e]; }
            }
            if (!last_chunk) continue;
            const uint32_t packed = (uint32_t)bi[0] | ((uint32_t)bi[1] << 8) | ((uint32_t)bi[2] << 16) | ((uint32_t)bi[3] << 24);
            if (labels) *reinterpret_cast<uint32_t*>(obase + (size_t)ry * W) = packed;
            if (CONF) {
                int bin[4];
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const long long g = GT64 ? gt64[GT64 ? ry : 0][e] : (long long)((gt8[GT64 ? 0 : ry] >> (8 * e)) & 0xFF);
                    bin[e] = (g >= 0 && g < ncls) ? (int)g * ncls + bi[e] : -1;
                }
                const int lane = (int)(threadIdx.x & 63);
                // the common case inside a segment: every pixel of the wave's row (64 lanes x 4 columns) falls in ONE bin -> one atomic
                const int lead = __builtin_amdgcn_readfirstlane(bin[0]);
                const int diff = (bin[0] ^ lead) | (bin[1] ^ lead) | (bin[2] ^ lead) | (bin[3] ^ lead);
                const unsigned long long active = __builtin_amdgcn_ballot_w64(true);
                if (lead >= 0 && __builtin_amdgcn_ballot_w64(diff == 0) == active) {
                    if (lane == __builtin_ctzll(active)) atomicAdd(&lh[lead], 4u * (unsigned)__builtin_popcountll(active));
                } else {
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        // one LDS atomic per bin for the wave's two most frequent-first bins (segment borders: usually that is all),
                        // per-lane atomics for the rest.  (Peeling EVERY distinct bin off with ballots is slower on noise-like labels --
                        // the synthetic benchmark's -- than the conflicting atomics it avoids: 1.328 vs 1.275 ms per evaluator step.)
                        const bool ok = bin[e] >= 0;
                        unsigned long long todo = __builtin_amdgcn_ballot_w64(ok);          // lanes that still have to be counted
#pragma unroll
                        for (int it = 0; it < 2 && todo; ++it) {                            // wave-uniform: the two most likely bins
                            const int leader = __builtin_ctzll(todo);
                            const int b0 = __builtin_amdgcn_readlane(bin[e], leader);
                            const unsigned long long same = __builtin_amdgcn_ballot_w64(ok && bin[e] == b0) & todo;
                            if (lane == leader) atomicAdd(&lh[b0], (unsigned)__builtin_popcountll(same));
                            todo &= ~same;
                        }
                        if ((todo >> lane) & 1ull) atomicAdd(&lh[bin[e]], 1u);              // whatever is left (noise-like labels): per lane
                    }
                }
            }
        }
        }   // class chunks
    }
    if (CONF) {
        __syncthreads();
        for (int i = threadIdx.x; i < ncls * ncls; i += 256)
            if (lh[i]) atomicAdd(&hist[i], (unsigned long long)lh[i]);
    }
}

// Standalone form of the same histogram for label maps that already exist (u8 predictions): hist[n*gt + pred] += 1.
template <bool GT64>
__global__ __launch_bounds__(256) void confusion_kernel(const void* __restrict__ gt, const uint8_t* __restrict__ pred, size_t n,
                                                        int ncls, unsigned long long* __restrict__ hist) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    unsigned* lh = reinterpret_cast<unsigned*>(smem);
    for (int i = threadIdx.x; i < ncls * ncls; i += 256) lh[i] = 0;
    __syncthreads();
    for (size_t id = (size_t)blockIdx.x * 256 + threadIdx.x; id < n; id += (size_t)gridDim.x * 256) {
        const long long g = GT64 ? reinterpret_cast<const long long*>(gt)[id] : (long long)reinterpret_cast<const uint8_t*>(gt)[id];
        const int pr = pred[id];
        if (g >= 0 && g < ncls && pr < ncls) atomicAdd(&lh[(int)g * ncls + pr], 1u);
    }
    __syncthreads();
    for (int i = threadIdx.x; i < ncls * ncls; i += 256)
        if (lh[i]) atomicAdd(&hist[i], (unsigned long long)lh[i]);
}

// Adjoint of upsample32_kernel (training backward): glow[m][c][y][x] = sum over the output pixels whose
// bilinear footprint touches low pixel (y, x) of wy * wx * gout[m][c][oy][ox], with the forward's exact source-index rule.
// Workgroup = (low row y, class c, image m): the <= 64 output rows that touch row y are reduced along y first (coalesced row
// reads, each thread owns columns), the W partial sums go through LDS, then every low column sums its <= 64 output columns.
// Deterministic (no atomics).  Reads each output row at most twice.
__device__ __forceinline__ float up32_weight(int o, int l, int n) {        // weight of low index l in output index o (n low pixels)
    float s = (o + 0.5f) * 0.03125f - 0.5f;
    s = s < 0.f ? 0.f : s;
    const int i0 = (int)s;
    const int i1 = i0 + (i0 < n - 1 ? 1 : 0);
    const float l1 = s - (float)i0;
    return (i0 == l ? 1.f - l1 : 0.f) + (i1 == l ? l1 : 0.f);
}
__global__ __launch_bounds__(256) void upsample32_backward_kernel(const float* __restrict__ gout, int h, int w, int ncls,
                                                                  float* __restrict__ glow) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float* t = reinterpret_cast<float*>(smem);         // [W]
    const int H = h * 32, W = w * 32;
    const int y = blockIdx.x, c = blockIdx.y, m = blockIdx.z;
    const float* g = gout + ((size_t)m * ncls + c) * H * W;
    const int o_lo = max(0, 32 * y - 16), o_hi = min(H, 32 * y + 48);
    for (int ox = threadIdx.x; ox < W; ox += 256) {
        float acc = 0.f;
        for (int oy = o_lo; oy < o_hi; ++oy) acc = __builtin_fmaf(up32_weight(oy, y, h), g[(size_t)oy * W + ox], acc);
        t[ox] = acc;
    }
    __syncthreads();
    // low column x: output columns [32x-16, 32x+48); 256 threads = w columns x (256/w) partial sums, reduced through LDS
    const int per = 256 / w;                            // threads per low column (w <= 32 -> >= 8)
    const int x = threadIdx.x / per, part = threadIdx.x - x * per;
    float acc = 0.f;
    if (x < w) {
        const int c_lo = max(0, 32 * x - 16), c_hi = min(W, 32 * x + 48);
        for (int ox = c_lo + part; ox < c_hi; ox += per) acc = __builtin_fmaf(up32_weight(ox, x, w), t[ox], acc);
    }
    __syncthreads();
    float* red = reinterpret_cast<float*>(smem) + W;   // [256]
    red[threadIdx.x] = acc;
    __syncthreads();
    if (x < w && part == 0) {
        float s2 = 0.f;
        for (int k = 0; k < per; ++k) s2 += red[x * per + k];
        glow[(((size_t)m * ncls + c) * h + y) * w + x] = s2;
    }
}

__global__ __launch_bounds__(256) void nchw_to_nhwc_kernel(const float* __restrict__ x, int M, int C, int HW,
                                                           uint16_t* __restrict__ y, int ycs) {
    const size_t total = (size_t)M * HW * C;
    for (size_t id = (size_t)blockIdx.x * 256 + threadIdx.x; id < total; id += (size_t)gridDim.x * 256) {
        const int c = (int)(id % C);
        const size_t t = id / C;
        const int p = (int)(t % HW);
        const int m = (int)(t / HW);
        y[((size_t)m * HW + p) * ycs + c] = f32_to_bf16(x[((size_t)m * C + c) * HW + p]);
    }
}

__global__ __launch_bounds__(256) void nhwc_to_nchw_kernel(const uint16_t* __restrict__ x, int xcs, int M, int C, int HW,
                                                           float* __restrict__ y) {
    const size_t total = (size_t)M * HW * C;
    for (size_t id = (size_t)blockIdx.x * 256 + threadIdx.x; id < total; id += (size_t)gridDim.x * 256) {
        const int p = (int)(id % HW);
        const size_t t = id / HW;
        const int c = (int)(t % C);
        const int m = (int)(t / C);
        y[id] = bf16_to_f32(x[((size_t)m * HW + p) * xcs + c]);
    }
}

unsigned grid_for(size_t total) {
    size_t b = (total + 255) / 256;
    return (unsigned)(b > 4096 ? 4096 : (b ? b : 1));
}

}  // namespace

extern "C" int w2c_upsample_bilinear32(const float* low, int M, int h, int w, int low_cstride, int n_classes,
                                       float* out, w2c_stream_t stream) {
    w2c_clear_error();
    if (!low || !out || M <= 0 || h <= 0 || w <= 0 || n_classes <= 0 || low_cstride < n_classes) return W2C_E_ARG;
    if ((size_t)h * w * 4 > 64 * 1024) return W2C_E_ARG;
    hipLaunchKernelGGL(upsample32_kernel, dim3(h, n_classes, M), dim3(256), (size_t)h * w * 4,
                       reinterpret_cast<hipStream_t>(stream), low, h, w, low_cstride, n_classes, out);
    return w2c_launch_status();
}

extern "C" int w2c_upsample_bilinear32_backward(const float* gout, int M, int h, int w, int n_classes, float* glow,
                                               w2c_stream_t stream) {
    w2c_clear_error();
    if (!gout || !glow || M <= 0 || h <= 0 || w <= 0 || w > 256 || n_classes <= 0) return W2C_E_ARG;
    const size_t lds = ((size_t)w * 32 + 256) * 4;
    if (lds > 64 * 1024) return W2C_E_ARG;
    hipLaunchKernelGGL(upsample32_backward_kernel, dim3(h, n_classes, M), dim3(256), lds, reinterpret_cast<hipStream_t>(stream),
                       gout, h, w, n_classes, glow);
    return w2c_launch_status();
}

// grid of the argmax kernels: (band, column-quad) items of an image in workgroups of 256, M images
// conf: every workgroup ends with one global atomic per non-zero bin of its histogram -- n^2 hot addresses -- so the confusion form
// runs ~1024 grid-striding workgroups instead of one per 256 items (cfg 2: 640 x 20: 1.5 M atomics on 121 addresses cost more than
// the argmax itself)
static dim3 argmax_grid(int M, int h, int w, bool conf = false) {
    const int items = (h * 32 / ARG_ROWS) * (w * 32 / 4);
    int gx = (items + 255) / 256;
    if (conf) {
        const int cap = (1024 + M - 1) / M;
        if (gx > cap) gx = cap;
    }
    return dim3((unsigned)gx, (unsigned)M);
}
static bool argmax_args_ok(const float* low, int low_cstride, int n_classes) {
    // the kernel reads the classes of a low-resolution pixel as 16-byte vectors
    return (low_cstride % 4) == 0 && low_cstride >= (n_classes + 3) / 4 * 4 && !(reinterpret_cast<uintptr_t>(low) & 15);
}

extern "C" int w2c_upsample32_argmax(const float* low, int M, int h, int w, int low_cstride, int n_classes,
                                     uint8_t* labels, w2c_stream_t stream) {
    w2c_clear_error();
    if (!low || !labels || M <= 0 || h <= 0 || w <= 0 || n_classes <= 0 || n_classes > 255 || low_cstride < n_classes)
        return W2C_E_ARG;
    if (!argmax_args_ok(low, low_cstride, n_classes)) return W2C_E_ARG;
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    if (n_classes <= ARG_MAXC)
        hipLaunchKernelGGL((upsample32_argmax_kernel<false, false, false>), argmax_grid(M, h, w), dim3(256), 0, s, low, h, w, low_cstride,
                           n_classes, labels, nullptr, nullptr);
    else
        hipLaunchKernelGGL((upsample32_argmax_kernel<false, false, true>), argmax_grid(M, h, w), dim3(256), 0, s, low, h, w, low_cstride,
                           n_classes, labels, nullptr, nullptr);
    return w2c_launch_status();
}

extern "C" int w2c_upsample32_argmax_confusion(const float* low, int M, int h, int w, int low_cstride, int n_classes,
                                               const void* gt, int gt_is_i64, uint8_t* labels, long long* hist,
                                               w2c_stream_t stream) {
    w2c_clear_error();
    if (!low || !gt || !hist || M <= 0 || h <= 0 || w <= 0 || n_classes <= 0 || n_classes > 64 || low_cstride < n_classes)
        return W2C_E_ARG;
    if (!argmax_args_ok(low, low_cstride, n_classes)) return W2C_E_ARG;
    // (a tagged pointer -- bit 0 set: the address of a pointer slot -- is resolved on the device; its target's alignment is the caller's to keep)
    if ((!(reinterpret_cast<uintptr_t>(gt) & 1) && (reinterpret_cast<uintptr_t>(gt) & (gt_is_i64 ? 7 : 3))) ||
        (!(reinterpret_cast<uintptr_t>(hist) & 1) && (reinterpret_cast<uintptr_t>(hist) & 7)))
        return W2C_E_ARG;
    const size_t lds = (size_t)n_classes * n_classes * 4;
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    unsigned long long* hp = reinterpret_cast<unsigned long long*>(hist);
    const dim3 grid = argmax_grid(M, h, w, true);
    const bool multi = n_classes > ARG_MAXC;
    if (gt_is_i64) {
        if (multi) hipLaunchKernelGGL((upsample32_argmax_kernel<true, true, true>), grid, dim3(256), lds, s, low, h, w, low_cstride, n_classes, labels, gt, hp);
        else hipLaunchKernelGGL((upsample32_argmax_kernel<true, true, false>), grid, dim3(256), lds, s, low, h, w, low_cstride, n_classes, labels, gt, hp);
    } else {
        if (multi) hipLaunchKernelGGL((upsample32_argmax_kernel<true, false, true>), grid, dim3(256), lds, s, low, h, w, low_cstride, n_classes, labels, gt, hp);
        else hipLaunchKernelGGL((upsample32_argmax_kernel<true, false, false>), grid, dim3(256), lds, s, low, h, w, low_cstride, n_classes, labels, gt, hp);
    }
    return w2c_launch_status();
}

extern "C" int w2c_confusion_matrix(const void* gt, int gt_is_i64, const uint8_t* pred, long long n_pixels, int n_classes,
                                    long long* hist, w2c_stream_t stream) {
    w2c_clear_error();
    if (!gt || !pred || !hist || n_pixels <= 0 || n_classes <= 0 || n_classes > 64) return W2C_E_ARG;
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    unsigned long long* hp = reinterpret_cast<unsigned long long*>(hist);
    const size_t lds = (size_t)n_classes * n_classes * 4;
    const unsigned grid = (unsigned)(((size_t)n_pixels + 256 * 16 - 1) / (256 * 16) > 2048 ? 2048 : ((size_t)n_pixels + 256 * 16 - 1) / (256 * 16));
    if (gt_is_i64)
        hipLaunchKernelGGL((confusion_kernel<true>), dim3(grid ? grid : 1), dim3(256), lds, s, gt, pred, (size_t)n_pixels, n_classes, hp);
    else
        hipLaunchKernelGGL((confusion_kernel<false>), dim3(grid ? grid : 1), dim3(256), lds, s, gt, pred, (size_t)n_pixels, n_classes, hp);
    return w2c_launch_status();
}

extern "C" int w2c_nchw_f32_to_nhwc_bf16(const float* x, int M, int C, int H, int W, uint16_t* y, int y_cstride,
                                         w2c_stream_t stream) {
    w2c_clear_error();
    if (!x || !y || M <= 0 || C <= 0 || H <= 0 || W <= 0 || y_cstride < C) return W2C_E_ARG;
    const size_t total = (size_t)M * C * H * W;
    hipLaunchKernelGGL(nchw_to_nhwc_kernel, dim3(grid_for(total)), dim3(256), 0, reinterpret_cast<hipStream_t>(stream),
                       x, M, C, H * W, y, y_cstride);
    return w2c_launch_status();
}

extern "C" int w2c_nhwc_bf16_to_nchw_f32(const uint16_t* x, int x_cstride, int M, int C, int H, int W, float* y,
                                         w2c_stream_t stream) {
    w2c_clear_error();
    if (!x || !y || M <= 0 || C <= 0 || H <= 0 || W <= 0 || x_cstride < C) return W2C_E_ARG;
    const size_t total = (size_t)M * C * H * W;
    hipLaunchKernelGGL(nhwc_to_nchw_kernel, dim3(grid_for(total)), dim3(256), 0, reinterpret_cast<hipStream_t>(stream),
                       x, x_cstride, M, C, H * W, y);
    return w2c_launch_status();
}

namespace {
struct SlotValues { const void* v[8]; };
__global__ void set_slots_kernel(const void** slots, int n, SlotValues vals) {
    if ((int)threadIdx.x < n) slots[threadIdx.x] = vals.v[threadIdx.x];
}
__global__ __launch_bounds__(256) void copy_to_slot_kernel(const uint32_t* __restrict__ src, long long n_words, uint32_t* dst_arg) {
    uint32_t* const __restrict__ dst = w2c_resolve(dst_arg);
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n_words; i += (long long)gridDim.x * 256) dst[i] = src[i];
}
}  // namespace

// Indirect operands, host side: fill up to 8 device-resident pointer slots in stream order (the values travel as kernel arguments, so
// the host may run any number of forwards ahead), and copy a small buffer to the address held in a slot.
extern "C" int w2c_set_slots(void* slots, int n, const void* const* values, w2c_stream_t stream) {
    w2c_clear_error();
    if (!slots || !values || n <= 0 || n > 8 || (reinterpret_cast<uintptr_t>(slots) & 7)) return W2C_E_ARG;
    SlotValues v{};
    for (int i = 0; i < n; ++i) v.v[i] = values[i];
    hipLaunchKernelGGL(set_slots_kernel, dim3(1), dim3(64), 0, reinterpret_cast<hipStream_t>(stream), reinterpret_cast<const void**>(slots), n, v);
    return w2c_launch_status();
}
extern "C" int w2c_copy_to_slot(const void* src, long long nbytes, void* dst, w2c_stream_t stream) {
    w2c_clear_error();
    if (!src || !dst || nbytes <= 0 || (nbytes & 3) || (reinterpret_cast<uintptr_t>(src) & 3)) return W2C_E_ARG;
    const long long nw = nbytes / 4;
    const unsigned grid = (unsigned)((nw + 255) / 256 > 64 ? 64 : (nw + 255) / 256);
    hipLaunchKernelGGL(copy_to_slot_kernel, dim3(grid), dim3(256), 0, reinterpret_cast<hipStream_t>(stream),
                       reinterpret_cast<const uint32_t*>(src), nw, reinterpret_cast<uint32_t*>(dst));
    return w2c_launch_status();
}

extern "C" int w2c_version(void) { return 1; }

extern "C" const char* w2c_last_error_string(void) { return w2c_errbuf(); }

extern "C" const char* w2c_status_string(int code) {
    switch (code) {
        case W2C_OK: return "ok";
        case W2C_E_ARG: return "invalid argument or unsupported shape";
        case W2C_E_LAUNCH: return "HIP kernel launch failed";
        default: return "unknown w2c status";
    }
}

extern "C" int w2c_device_arch(char* buf, int buflen) {
    if (!buf || buflen <= 0) return W2C_E_ARG;
    int dev = 0;
    hipDeviceProp_t prop;
    if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&prop, dev) != hipSuccess) return W2C_E_LAUNCH;
    int i = 0;
    for (; i < buflen - 1 && prop.gcnArchName[i]; ++i) buf[i] = prop.gcnArchName[i];
    buf[i] = 0;
    return W2C_OK;
}
