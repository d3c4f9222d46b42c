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
    for (int id = threadIdx.x; id < 32 * xq; id += 256) {
        const int ry = id / xq, gx = id - ry * xq;
        const int oy = band * 32 + ry;
        float sy = (oy + 0.5f) * 0.03125f - 0.5f;
        sy = sy < 0.f ? 0.f : sy;
        const int y0 = (int)sy;
        const int y1 = y0 + (y0 < h - 1 ? 1 : 0);
        const float ly1 = sy - (float)y0, ly0 = 1.f - ly1;
        f32x4_t o;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int ox = gx * 4 + e;
            float sx = (ox + 0.5f) * 0.03125f - 0.5f;
            sx = sx < 0.f ? 0.f : sx;
            const int x0 = (int)sx;
            const int x1 = x0 + (x0 < w - 1 ? 1 : 0);
            const float lx1 = sx - (float)x0, lx0 = 1.f - lx1;
            o[e] = lerp2d(ly0, ly1, lx0, lx1, plane[y0 * w + x0], plane[y0 * w + x1], plane[y1 * w + x0], plane[y1 * w + x1]);
        }
        *reinterpret_cast<f32x4_t*>(obase + (size_t)ry * W + gx * 4) = o;   // (nontemporal stores measured: no gain)
    }
}

// K9 + the evaluator's class argmax (trainer.py:804 `outputs.data.max(1)[1]`) fused: the full-resolution f32
// logits (231 MB at cfg 2) are never written -- only one u8 label per pixel (5 MB).  Same source-index and lerp
// arithmetic as upsample32_kernel, so labels == argmax over classes of its output, bit for bit; ties keep the
// lowest class index.  Workgroup = (32-row band, image): the image's whole low-res logit block sits in LDS.
//
// CONF: the evaluator's confusion matrix (runningScore._fast_hist, metrics.py:99-108: bincount(n*gt + pred) over the
// pixels with 0 <= gt < n) fused behind the argmax -- each workgroup histograms its 32-row band in LDS (a wave whose 64
// lanes all hit one bin, the common case inside a segment, adds 64 with one atomic) and flushes its non-zero bins
// with one 64-bit global atomic each.  Integer atomics: the result is exact and order-independent.
constexpr int ARG_ROWS = 8;
template <bool CONF, bool GT64>
__global__ __launch_bounds__(256) void upsample32_argmax_kernel(const float* __restrict__ low, int h, int w, int lcs, int ncls,
                                                                uint8_t* labels_arg, const void* gt_arg,
                                                                unsigned long long* hist_arg) {
    uint8_t* const __restrict__ labels = w2c_resolve(labels_arg);   // indirect operands (caller-owned tensors of a captured forward)
    const void* const __restrict__ gt = w2c_resolve(gt_arg);
    unsigned long long* const __restrict__ hist = w2c_resolve(hist_arg);
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float* blk = reinterpret_cast<float*>(smem);       // [ncls][h][w]
    unsigned* lh = reinterpret_cast<unsigned*>(smem + (size_t)ncls * h * w * 4);      // CONF: [ncls*ncls] band histogram
    if (CONF) {
        for (int i = threadIdx.x; i < ncls * ncls; i += 256) lh[i] = 0;
    }
    const int H = h * 32, W = w * 32;
    // a workgroup = ARG_ROWS output rows of one image (32-row bands were 320 workgroups at cfg 2: 1.25 per CU, 4 waves per
    // CU -- the kernel idled on latency: 46 us for 5 MB of output)
    const int band = blockIdx.x, m = blockIdx.y;
    for (int i = threadIdx.x; i < ncls * h * w; i += 256) {
        const int c = i / (h * w), p = i - c * (h * w);
        blk[i] = low[((size_t)m * h * w + p) * lcs + c];
    }
    __syncthreads();
    const int xq = W >> 2;
    uint8_t* obase = labels + ((size_t)m * H + (size_t)band * ARG_ROWS) * W;
    for (int id = threadIdx.x; id < ARG_ROWS * xq; id += 256) {
        const int ry = id / xq, gx = id - ry * xq;
        const int oy = band * ARG_ROWS + ry;
        float sy = (oy + 0.5f) * 0.03125f - 0.5f;
        sy = sy < 0.f ? 0.f : sy;
        const int y0 = (int)sy;
        const int y1 = y0 + (y0 < h - 1 ? 1 : 0);
        const float ly1 = sy - (float)y0, ly0 = 1.f - ly1;
        // the 4 pixels of a group share their source columns: x0 changes only at ox = 16 (mod 32), a multiple of 4, and the
        // left-edge clamp (sx < 0 -> 0) covers ox < 16 -- so the four corner values are read once per class, not per pixel
        float sx0 = (gx * 4 + 0.5f) * 0.03125f - 0.5f;
        sx0 = sx0 < 0.f ? 0.f : sx0;
        const int x0 = (int)sx0;
        const int x1 = x0 + (x0 < w - 1 ? 1 : 0);
        float lx1[4], lx0[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            float sx = (gx * 4 + e + 0.5f) * 0.03125f - 0.5f;
            sx = sx < 0.f ? 0.f : sx;
            lx1[e] = sx - (float)x0;
            lx0[e] = 1.f - lx1[e];
        }
        float best[4] = {-INFINITY, -INFINITY, -INFINITY, -INFINITY};
        int bi[4] = {0, 0, 0, 0};
        for (int c = 0; c < ncls; ++c) {
            const float* pl = blk + c * h * w;
            const float p00 = pl[y0 * w + x0], p01 = pl[y0 * w + x1], p10 = pl[y1 * w + x0], p11 = pl[y1 * w + x1];
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const float v = lerp2d(ly0, ly1, lx0[e], lx1[e], p00, p01, p10, p11);
                if (v > best[e]) { best[e] = v; bi[e] = c; }
            }
        }
        const uint32_t packed = (uint32_t)bi[0] | ((uint32_t)bi[1] << 8) | ((uint32_t)bi[2] << 16) | ((uint32_t)bi[3] << 24);
        if (labels) *reinterpret_cast<uint32_t*>(obase + (size_t)ry * W + gx * 4) = packed;
        if (CONF) {
            const size_t pix = ((size_t)m * H + (size_t)band * ARG_ROWS + ry) * W + gx * 4;
            long long g4[4];
            if (GT64) {
                const long long* gp = reinterpret_cast<const long long*>(gt) + pix;
#pragma unroll
                for (int e = 0; e < 4; ++e) g4[e] = gp[e];
            } else {
                const uint32_t gw = *reinterpret_cast<const uint32_t*>(reinterpret_cast<const uint8_t*>(gt) + pix);
#pragma unroll
                for (int e = 0; e < 4; ++e) g4[e] = (gw >> (8 * e)) & 0xFF;
            }
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const bool ok = g4[e] >= 0 && g4[e] < ncls;
                const int bin = ok ? (int)g4[e] * ncls + bi[e] : -1;
                // one LDS atomic per bin for the wave's two most frequent-first bins (segments are large: usually that is all),
                // per-lane atomics for the rest.  (Peeling EVERY distinct bin off with ballots is slower on noise-like labels --
                // the synthetic benchmark's -- than the conflicting atomics it avoids: 1.328 vs 1.275 ms per evaluator step.)
                unsigned long long todo = __builtin_amdgcn_ballot_w64(ok);              // lanes that still have to be counted
                const int lane = (int)(threadIdx.x & 63);
#pragma unroll
                for (int it = 0; it < 2 && todo; ++it) {                                // wave-uniform: the two most likely bins
                    const int leader = __builtin_ctzll(todo);
                    const int b0 = __builtin_amdgcn_readlane(bin, leader);
                    const unsigned long long same = __builtin_amdgcn_ballot_w64(ok && bin == b0) & todo;
                    if (lane == leader) atomicAdd(&lh[b0], (unsigned)__builtin_popcountll(same));
                    todo &= ~same;
                }
                if ((todo >> lane) & 1ull) atomicAdd(&lh[bin], 1u);                     // whatever is left (noise-like labels): per lane
            }
        }
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

extern "C" int w2c_upsample32_argmax(const float* low, int M, int h, int w, int low_cstride, int n_classes,
                                     uint8_t* labels, w2c_stream_t stream) {
    w2c_clear_error();
    if (!low || !labels || M <= 0 || h <= 0 || w <= 0 || n_classes <= 0 || n_classes > 255 || low_cstride < n_classes)
        return W2C_E_ARG;
    const size_t lds = (size_t)n_classes * h * w * 4;
    if (lds > 64 * 1024) return W2C_E_ARG;
    hipLaunchKernelGGL((upsample32_argmax_kernel<false, false>), dim3(h * (32 / ARG_ROWS), M), dim3(256), lds, reinterpret_cast<hipStream_t>(stream),
                       low, h, w, low_cstride, n_classes, labels, nullptr, nullptr);
    return w2c_launch_status();
}

extern "C" int w2c_upsample32_argmax_confusion(const float* low, int M, int h, int w, int low_cstride, int n_classes,
                                               const void* gt, int gt_is_i64, uint8_t* labels, long long* hist,
                                               w2c_stream_t stream) {
    w2c_clear_error();
    if (!low || !gt || !hist || M <= 0 || h <= 0 || w <= 0 || n_classes <= 0 || n_classes > 64 || low_cstride < n_classes)
        return W2C_E_ARG;
    // (a tagged pointer -- bit 0 set: the address of a pointer slot -- is resolved on the device; its target's alignment is the caller's to keep)
    if ((!(reinterpret_cast<uintptr_t>(gt) & 1) && (reinterpret_cast<uintptr_t>(gt) & (gt_is_i64 ? 7 : 3))) ||
        (!(reinterpret_cast<uintptr_t>(hist) & 1) && (reinterpret_cast<uintptr_t>(hist) & 7)))
        return W2C_E_ARG;
    const size_t lds = (size_t)n_classes * h * w * 4 + (size_t)n_classes * n_classes * 4;
    if (lds > 64 * 1024) return W2C_E_ARG;
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    unsigned long long* hp = reinterpret_cast<unsigned long long*>(hist);
    if (gt_is_i64)
        hipLaunchKernelGGL((upsample32_argmax_kernel<true, true>), dim3(h * (32 / ARG_ROWS), M), dim3(256), lds, s, low, h, w, low_cstride,
                           n_classes, labels, gt, hp);
    else
        hipLaunchKernelGGL((upsample32_argmax_kernel<true, false>), dim3(h * (32 / ARG_ROWS), M), dim3(256), lds, s, low, h, w, low_cstride,
                           n_classes, labels, gt, hp);
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
