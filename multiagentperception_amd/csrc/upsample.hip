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
constexpr int CONF_WS_PARTIALS = 32;                // partial histograms of the confusion kernel's two-level flush (include/w2c_hip.h)
constexpr int ARG_MAXC = 12;                        // classes handled by the register form (3 x 16 B per corner)
// MULTI: more than ARG_MAXC classes -- the classes are walked in chunks of ARG_MAXC with the rows' running (best, index) kept across
// chunks (64 more registers); the comparison order is still class-ascending, so ties keep the lowest index.
// Occupancy matters more than anything else here: cfg 2 is 640 workgroups, i.e. 2.5 per CU -- at 3 waves per SIMD (<= 168 VGPRs) they
// are all resident at once, at 2 (the confusion form took 172) the launch runs two rounds and takes twice the labels-only time.
template <bool CONF, bool GT64, bool MULTI = false>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(MULTI ? 1 : 3, 8))) void upsample32_argmax_kernel(const float* __restrict__ low, int h, int w, int lcs, int ncls,
                                                                uint8_t* labels_arg, const void* gt_arg,
                                                                unsigned long long* hist_arg, unsigned long long* ws, int ws_k,
                                                                int ws_stride) {
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
        uint32_t gt8[CONF ? ARG_ROWS : 1];
        if (CONF) {
            if (GT64) {
                // int64 labels (the loader's dtype): narrowed to a byte per pixel as they arrive (0xFF = outside [0, n)), four rows'
                // loads in flight at a time -- all eight at once are 64 registers the kernel does not have (see the occupancy note)
#pragma unroll
                for (int half = 0; half < 2; ++half) {
                    long long g64[ARG_ROWS / 2][4];
#pragma unroll
                    for (int r = 0; r < ARG_ROWS / 2; ++r) {
                        const size_t pix = ((size_t)m * H + band * ARG_ROWS + half * (ARG_ROWS / 2) + r) * W + gx * 4;
                        const long long* gp = reinterpret_cast<const long long*>(gt) + pix;
#pragma unroll
                        for (int e = 0; e < 4; ++e) g64[r][e] = gp[e];
                    }
#pragma unroll
                    for (int r = 0; r < ARG_ROWS / 2; ++r) {
                        uint32_t pk = 0;
#pragma unroll
                        for (int e = 0; e < 4; ++e)
                            pk |= ((g64[r][e] >= 0 && g64[r][e] < ncls) ? (uint32_t)g64[r][e] : 0xFFu) << (8 * e);
                        gt8[half * (ARG_ROWS / 2) + r] = pk;
                    }
                    // (the later loads must not be hoisted above this point again: neither by the IR passes nor by the scheduler)
                    asm volatile("" : "+v"(gt8[half * 4]), "+v"(gt8[half * 4 + 1]), "+v"(gt8[half * 4 + 2]), "+v"(gt8[half * 4 + 3])::"memory");
                    __builtin_amdgcn_sched_barrier(0);
                }
            } else {
#pragma unroll
                for (int ry = 0; ry < ARG_ROWS; ++ry) {
                    const size_t pix = ((size_t)m * H + band * ARG_ROWS + ry) * W + gx * 4;
                    gt8[ry] = *reinterpret_cast<const uint32_t*>(reinterpret_cast<const uint8_t*>(gt) + pix);
                }
            }
        }
        // CONF: the 32 bins of the thread's pixels, packed (bins < 144 in a byte when ncls <= 12, else < 4096 in 16 bits)
        constexpr unsigned NOBIN = MULTI ? 0xFFFFu : 0xFFu;
        unsigned binp[CONF ? (MULTI ? 2 * ARG_ROWS : ARG_ROWS) : 1];
        if (CONF) {
#pragma unroll
            for (int i = 0; i < (MULTI ? 2 * ARG_ROWS : ARG_ROWS); ++i) binp[i] = 0;
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
                if (MULTI) {
                    // this row's four bins n * gt + pred (NOBIN: gt outside [0, n)), kept for the histogram step behind the rows
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const int g = (int)((gt8[CONF ? ry : 0] >> (8 * e)) & 0xFF);        // (u8 labels: 255 >= n is outside too: n <= 64)
                        const unsigned b = g < ncls ? (unsigned)(g * ncls + bi[e]) : NOBIN;
                        binp[MULTI ? 2 * ry + (e >> 1) : 0] |= b << (16 * (e & 1));
                    }
                } else {
                    binp[MULTI ? 0 : ry] = packed;           // <= 12 classes: the row's four PREDICTIONS; the bins are formed behind the rows
                }
            }
        }
        }   // class chunks
        if (CONF && !MULTI) {
            // Histogram step, once per thread, <= 12 classes (round 5; the round-4 form added a thread's 32 pixels one LDS atomic at a time
            // unless all 32 shared ONE bin: with noise-like ground truth -- the synthetic benchmark's -- that is 2048 atomics per wave on
            // 121 addresses: SQ_LDS_BANK_CONFLICT 77 %).  An upsampled x32 PREDICTION is constant over almost every 8 x 4 block, whatever
            // the ground truth looks like: a thread whose 32 pixels share one prediction counts its ground-truth classes in REGISTERS
            // (12 byte counters in three words); a wave whose 64 threads share that prediction (the common case) sums the counters with
            // shuffles and lane g adds class g's total with ONE atomic -- <= 12 atomics per wave, all on different addresses.  Threads on
            // a prediction border fall back to per-pixel atomics.  Integer counts: the histogram is exact either way.
            const unsigned p0 = binp[0] & 0xFFu;
            const unsigned rep = p0 * 0x01010101u;
            unsigned diff = 0;
#pragma unroll
            for (int i = 0; i < ARG_ROWS; ++i) diff |= binp[i] ^ rep;
            const bool puni = diff == 0;
            const int lane = (int)(threadIdx.x & 63);
            const unsigned long long active = __builtin_amdgcn_ballot_w64(true);
            const unsigned lead = (unsigned)__builtin_amdgcn_readfirstlane((int)p0);
            const bool wave_uni = active == ~0ull && __builtin_amdgcn_ballot_w64(puni && p0 == lead) == active;
            if (puni) {
                unsigned c0 = 0, c1 = 0, c2 = 0;             // byte counters of ground-truth classes 0-3 | 4-7 | 8-11 (each <= 32)
#pragma unroll
                for (int ry = 0; ry < ARG_ROWS; ++ry)
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const unsigned g = (gt8[CONF ? ry : 0] >> (8 * e)) & 0xFFu;   // >= ncls (incl. the 0xFF of an out-of-range label): dropped
                        const unsigned inc = g < (unsigned)ncls ? 1u << ((g & 3u) * 8u) : 0u;
                        const unsigned wsel = g >> 2;
                        c0 += wsel == 0u ? inc : 0u;
                        c1 += wsel == 1u ? inc : 0u;
                        c2 += wsel == 2u ? inc : 0u;
                    }
                if (wave_uni) {
                    // 16-bit pairs {class 4w + b, class 4w + b + 2}: 64 lanes x 32 = 2048 fits
                    unsigned wv[6] = {c0 & 0x00FF00FFu, (c0 >> 8) & 0x00FF00FFu, c1 & 0x00FF00FFu, (c1 >> 8) & 0x00FF00FFu,
                                      c2 & 0x00FF00FFu, (c2 >> 8) & 0x00FF00FFu};
#pragma unroll
                    for (int i = 0; i < 6; ++i)
#pragma unroll
                        for (int off = 32; off >= 1; off >>= 1) wv[i] += (unsigned)__shfl_xor((int)wv[i], off, 64);
                    unsigned mine = 0;
#pragma unroll
                    for (int g = 0; g < ARG_MAXC; ++g) {
                        const unsigned src = wv[2 * (g >> 2) + (g & 1)];
                        const unsigned cnt = (g & 2) ? src >> 16 : src & 0xFFFFu;
                        mine = lane == g ? cnt : mine;
                    }
                    if (lane < ncls && mine) atomicAdd(&lh[lane * ncls + (int)lead], mine);
                } else {
#pragma unroll
                    for (int g = 0; g < ARG_MAXC; ++g) {
                        const unsigned src = g < 4 ? c0 : g < 8 ? c1 : c2;
                        const unsigned cnt = (src >> (8 * (g & 3))) & 0xFFu;
                        if (g < ncls && cnt) atomicAdd(&lh[g * ncls + (int)p0], cnt);
                    }
                }
            } else {
#pragma unroll
                for (int ry = 0; ry < ARG_ROWS; ++ry)
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const unsigned g = (gt8[CONF ? ry : 0] >> (8 * e)) & 0xFFu;
                        const unsigned pr = (binp[MULTI ? 0 : ry] >> (8 * e)) & 0xFFu;
                        if (g < (unsigned)ncls) atomicAdd(&lh[g * ncls + pr], 1u);
                    }
            }
        }
        if (CONF && MULTI) {
            // > 12 classes: packed 16-bit bins, per-thread / per-wave uniform fast paths (the round-4 form)
            constexpr int NP = 2 * ARG_ROWS;
            const unsigned first = binp[0] & NOBIN;
            const unsigned rep = first * 0x00010001u;
            unsigned diff = 0;
#pragma unroll
            for (int i = 0; i < NP; ++i) diff |= binp[MULTI ? i : 0] ^ rep;
            const bool uni = diff == 0;
            const int lane = (int)(threadIdx.x & 63);
            const unsigned long long active = __builtin_amdgcn_ballot_w64(true);
            const unsigned lead = (unsigned)__builtin_amdgcn_readfirstlane((int)first);
            if (__builtin_amdgcn_ballot_w64(uni && first == lead) == active) {
                if (lead != NOBIN && lane == __builtin_ctzll(active))
                    atomicAdd(&lh[lead], (unsigned)(4 * ARG_ROWS) * (unsigned)__builtin_popcountll(active));
            } else if (uni) {
                if (first != NOBIN) atomicAdd(&lh[first], (unsigned)(4 * ARG_ROWS));
            } else {
#pragma unroll
                for (int i = 0; i < NP; ++i)
#pragma unroll
                    for (int e = 0; e < 2; ++e) {
                        const unsigned b = (binp[MULTI ? i : 0] >> (16 * e)) & NOBIN;
                        if (b != NOBIN) atomicAdd(&lh[b], 1u);
                    }
            }
        }
    }
    if (CONF) {
        __syncthreads();
        const int nbins = ncls * ncls;
        if (ws) {
            // two-level flush: 640 workgroups x up to 121 64-bit atomics on the caller's 121 counters (8 cache lines) cost 5-9 us of the
            // launch; here the workgroups spread over ws_k partial histograms (a line-aligned stride apart), and the LAST workgroup to
            // finish (a counter behind the partials) moves their sums into hist and leaves the workspace zeroed for the next launch.
            // Integer sums: exact and order-independent, as before.
            __shared__ int is_last;
            const unsigned wg = blockIdx.y * gridDim.x + blockIdx.x;
            unsigned long long* part = ws + (size_t)(wg % (unsigned)ws_k) * ws_stride;
            // no __threadfence() here: on this part an agent-scope fence writes the XCD's L2 back (every workgroup paying for the dirty
            // lines of the kernels before it: +28 us).  Device-scope atomics are performed at the coherence point; a thread that has
            // its atomics' RETURN values knows they have been, and the barrier then orders them before the counter's increment.
            unsigned long long seen = 0;
            for (int i = threadIdx.x; i < nbins; i += 256)
                if (lh[i]) seen += atomicAdd(&part[i], (unsigned long long)lh[i]);
            asm volatile("" ::"v"((unsigned)seen));
            __syncthreads();
            unsigned long long* counter = ws + (size_t)ws_k * ws_stride;
            if (threadIdx.x == 0) is_last = atomicAdd(counter, 1ull) == (unsigned long long)(gridDim.x * gridDim.y) - 1ull;
            __syncthreads();
            if (is_last) {
                // thread = (bin, half of the partials): its 16 read-and-zero atomics are independent and all in flight together
                // (a rolled loop of returning atomics is one L2 round trip EACH: 32 of them cost 37 us)
                for (int i = threadIdx.x & 127; i < nbins; i += 128) {
                    const int k0 = (threadIdx.x >> 7) * (CONF_WS_PARTIALS / 2);
                    unsigned long long v[CONF_WS_PARTIALS / 2];
#pragma unroll
                    for (int k = 0; k < CONF_WS_PARTIALS / 2; ++k) v[k] = atomicExch(&ws[(size_t)(k0 + k) * ws_stride + i], 0ull);
                    unsigned long long sum = 0;
#pragma unroll
                    for (int k = 0; k < CONF_WS_PARTIALS / 2; ++k) sum += v[k];
                    if (sum) atomicAdd(&hist[i], sum);
                }
                if (threadIdx.x == 0) atomicExch(counter, 0ull);
            }
        } else {
            for (int i = threadIdx.x; i < nbins; i += 256)
                if (lh[i]) atomicAdd(&hist[i], (unsigned long long)lh[i]);
        }
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
    const size_t lds = (size_t)h * w * 4;
    hipLaunchKernelGGL(upsample32_kernel, dim3(h, n_classes, M), dim3(256), lds,
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
                           n_classes, labels, nullptr, nullptr, nullptr, 0, 0);
    else
        hipLaunchKernelGGL((upsample32_argmax_kernel<false, false, true>), argmax_grid(M, h, w), dim3(256), 0, s, low, h, w, low_cstride,
                           n_classes, labels, nullptr, nullptr, nullptr, 0, 0);
    return w2c_launch_status();
}

extern "C" int w2c_upsample32_argmax_confusion(const float* low, int M, int h, int w, int low_cstride, int n_classes,
                                               const void* gt, int gt_is_i64, uint8_t* labels, long long* hist,
                                               long long* ws, int ws_partials, w2c_stream_t stream) {
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
    // optional workspace of the two-level flush: ws_partials histograms, w2c_confusion_ws_stride(n_classes) counters apart, + 16 counters
    // behind them; zeroed by the caller ONCE, left zeroed by every launch; one workspace per stream (launches sharing one must be ordered)
    const int ws_stride = (n_classes * n_classes + 15) / 16 * 16;
    if (ws && ws_partials != 0 && ws_partials != CONF_WS_PARTIALS) return W2C_E_ARG;
    unsigned long long* wsp = (ws && ws_partials > 0) ? reinterpret_cast<unsigned long long*>(ws) : nullptr;
    if (wsp && (reinterpret_cast<uintptr_t>(ws) & 7)) return W2C_E_ARG;
    const bool multi = n_classes > ARG_MAXC;
    if (gt_is_i64) {
        if (multi) hipLaunchKernelGGL((upsample32_argmax_kernel<true, true, true>), grid, dim3(256), lds, s, low, h, w, low_cstride, n_classes, labels, gt, hp, wsp, ws_partials, ws_stride);
        else hipLaunchKernelGGL((upsample32_argmax_kernel<true, true, false>), grid, dim3(256), lds, s, low, h, w, low_cstride, n_classes, labels, gt, hp, wsp, ws_partials, ws_stride);
    } else {
        if (multi) hipLaunchKernelGGL((upsample32_argmax_kernel<true, false, true>), grid, dim3(256), lds, s, low, h, w, low_cstride, n_classes, labels, gt, hp, wsp, ws_partials, ws_stride);
        else hipLaunchKernelGGL((upsample32_argmax_kernel<true, false, false>), grid, dim3(256), lds, s, low, h, w, low_cstride, n_classes, labels, gt, hp, wsp, ws_partials, ws_stride);
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

// ---- debug: native backtrace on fatal signals (include/w2c_hip.h) ----
#include <execinfo.h>
#include <signal.h>
#include <unistd.h>
namespace {
struct sigaction g_prev_sa[65];
int g_crash_fd = 2;
void crash_backtrace_handler(int sig, siginfo_t* info, void* uc) {
    void* frames[96];
    const int n = backtrace(frames, 96);
    static const char head[] = "\n==== libw2c_hip: native backtrace of the faulting thread ====\n";
    (void)!write(g_crash_fd, head, sizeof(head) - 1);
    backtrace_symbols_fd(frames, n, g_crash_fd);
    static const char tail[] = "==== end of native backtrace ====\n";
    (void)!write(g_crash_fd, tail, sizeof(tail) - 1);
    const struct sigaction& prev = g_prev_sa[sig];
    if ((prev.sa_flags & SA_SIGINFO) && prev.sa_sigaction) {
        prev.sa_sigaction(sig, info, uc);
    } else if (!(prev.sa_flags & SA_SIGINFO) && prev.sa_handler != SIG_DFL && prev.sa_handler != SIG_IGN) {
        prev.sa_handler(sig);
    }
    signal(sig, SIG_DFL);                                // (a chained handler that returns: die with the default action)
    raise(sig);
}
}  // namespace
extern "C" int w2c_debug_install_crash_backtrace(int fd) {
    g_crash_fd = fd >= 0 ? fd : 2;
    void* warm[4];
    (void)backtrace(warm, 4);                            // loads libgcc's unwinder now, outside any signal handler
    static const int sigs[] = {SIGSEGV, SIGBUS, SIGABRT, SIGFPE, SIGILL};
    for (int sig : sigs) {
        struct sigaction sa;
        sa.sa_sigaction = crash_backtrace_handler;
        sigemptyset(&sa.sa_mask);
        sa.sa_flags = SA_SIGINFO | SA_ONSTACK | SA_NODEFER;
        struct sigaction old;
        if (sigaction(sig, &sa, &old) == 0 && old.sa_sigaction != crash_backtrace_handler) g_prev_sa[sig] = old;
    }
    return W2C_OK;
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
