// conv_igemm.hip -- 3x3 / 1x1 convolution as an MFMA implicit GEMM for gfx950 (CDNA4).
//
// Stands in for nn.Conv2d + BatchNorm2d(eval) + (residual add) + ReLU on the When2com path:
// BasicBlock convs and 1x1 downsamples of the ResNet-18 trunks (backbone.py:66-69 via the
// third-party resnet18), conv2DBatchNormRelu (models/utils.py:87-120: squeezer agent.py:54,
// policy convs agent.py:126-132) and the decoder convs (backbone.py:150-154).
//
// GEMM view:  D[pixel][cout] = sum_{tap,ci} X[pixel @ tap][ci] * Wt[cout][tap][ci]
//   rows   = output pixels (M*Ho*Wo, NHWC order => a row IS the output pixel index)
//   cols   = output channels of one group
//   K      = ksize*ksize*Cin, walked as (64-channel chunk, tap) -- tap fastest, the SAME order in every
//            kernel of this file, so results are bit-identical across tile variants: NHWC makes
//            every K-step of a row one contiguous run of BK*2 bytes, so the im2col gather is a per-lane
//            ADDRESS (row base + a wave-uniform tap offset), never a materialised matrix.
// Per workgroup (64*WM*WN threads): a BM x BN tile.
//   * operands go HBM/L2 -> LDS by LDS-DMA (global_load_lds_dwordx4: 16 B per lane; the LDS image
//     is lane-linear, so the bank swizzle is applied to the per-lane SOURCE address and again on
//     the ds_read -- cdna_hip_programming.md rule 21).  Padded taps read a zero page.
//   * STAGES-deep LDS ring: the DMA of K-step t+STAGES-1 is issued at step t, so STAGES-2 full
//     steps of MFMA work cover the L2/HBM latency; waits are counted (s_waitcnt vmcnt(N), never a
//     drain in steady state) and there is ONE raw s_barrier per K-step (it publishes tile t and
//     retires the buffer tile t-1 was read from).
//   * all ds_read_b128 of a K-step are issued ahead of its MFMAs (compiler emits counted lgkmcnt);
//     v_mfma_f32_32x32x16_bf16, f32 accumulate; A = pixels, B = channels.
//   * epilogue: per-channel scale/shift (eval BN or bias) in registers, tile staged through LDS as
//     f32 so the residual read and the bf16/f32 store are 16-byte coalesced along channels.
// LDS swizzle (rows of BK*2 bytes, 16-B chunks): BK=64: chunk c of row r at c ^ ((r>>1)&7);
// BK=32: c ^ ((r>>2)&3).  ds_read_b128 is serviced in 16-lane groups over a 256-B bank row; these
// XORs make each group's 16 (row-in-bank-row, position) pairs distinct => conflict-free
// (MI355X_MICROARCH.md LDS table).
#include "w2c_common.h"
#include <cstdlib>
#include <type_traits>

#include <atomic>
#include <cstring>

static const char* const kOptionNames[W2C_OPT_COUNT] = {"W2C_XCD2D", "W2C_NO_S2PATCH", "W2C_STEM_WGS", "W2C_STEM_FORM", "W2C_STEM_BAND",
                                                        "W2C_STEM_WAVES", "W2C_WGRAD_PATCH", "W2C_INWG_SPLITK", "W2C_WREG_MINCIN", "W2C_WREG_FORM", "W2C_REGW_FORM", "W2C_REGH_WGS", "W2C_REGH_FORM", "W2C_L1_FORM", "W2C_S2WREG_FORM", "W2C_S2REGH"};
static std::atomic<int> g_options[W2C_OPT_COUNT];
static const bool g_options_seeded = [] {
    const int defaults[W2C_OPT_COUNT] = {1, 0, 0, 0, 8, 8, 1, 1, 256, 0, 1, 0, 0, 54, 1, 1};
    for (int i = 0; i < W2C_OPT_COUNT; ++i) {
        const char* e = getenv(kOptionNames[i]);            // once, at library load
        g_options[i].store(e ? atoi(e) : defaults[i]);
    }
    return true;
}();
int w2c_option(int id) { return (id >= 0 && id < W2C_OPT_COUNT) ? g_options[id].load(std::memory_order_relaxed) : 0; }
extern "C" int w2c_set_option(const char* name, int value) {
    if (!name) return W2C_E_ARG;
    for (int i = 0; i < W2C_OPT_COUNT; ++i)
        if (!strcmp(name, kOptionNames[i])) { g_options[i].store(value); return W2C_OK; }
    return W2C_E_ARG;
}
extern "C" int w2c_get_option(const char* name) {
    if (!name) return -1;
    for (int i = 0; i < W2C_OPT_COUNT; ++i)
        if (!strcmp(name, kOptionNames[i])) return g_options[i].load();
    return -1;
}

// Debug / measurement: the next conv entry call of THIS thread makes its kernels record the launch's span -- min over workgroups of the
// start stamp, max of the end stamp (100 MHz wall clock) -- into slot[0..1] (u64; the caller presets {~0, 0}).  The pointer is a kernel
// argument, so a launch captured into a HIP graph keeps recording on every replay: bench.py times the conv family under graph
// replay, where the two trunk chains overlap, without a profiler.  Costs two atomics per workgroup when set, one scalar branch when not.
static thread_local unsigned long long* g_span_next = nullptr;
extern "C" int w2c_debug_conv_span(void* slot) {
    g_span_next = reinterpret_cast<unsigned long long*>(slot);
    return W2C_OK;
}

// A/B switch (compile time): with -DW2C_AGPR_FORM the ring / patch kernels carry one dummy "a"-constrained asm operand, which makes the
// backend select the AGPR form of every builtin MFMA in them (accumulators in AGPRs, 128 + 128 register split) instead of the VGPR form
// it picks for kernels budgeted for <= 256 registers.  Why it matters: tools/ubench/coissue.hip.
#ifdef W2C_AGPR_FORM
#define W2C_FORCE_AGPR_FORM() asm volatile("" ::"a"(0))
#else
#define W2C_FORCE_AGPR_FORM() do {} while (0)
#endif

namespace {

struct ConvArgs {
    const uint16_t* x;
    const uint16_t* w;
    const float* scale;
    const float* shift;
    const uint16_t* res;
    void* y;
    const uint16_t* zeros;
    int M, H, W, Cin, xcs;
    int Ho, Wo, Cout, ycs;
    int ks, stride, pad;
    int relu, y_f32;
    int rows;        // M*Ho*Wo
    int cin_tiles;   // Cin / BK
    int ktiles;      // ks*ks*cin_tiles
    int ntm, ntn;    // tiles along rows / cout
    unsigned long long* dbg;   // optional timeline buffer (tools/conv_timeline.py): 4 x u64 per workgroup, else null
    unsigned long long* span;  // optional launch span (bench.py's roofline pass): [0] = min over workgroups of the start stamp,
                               // [1] = max of the end stamp (100 MHz wall clock), else null
    float* ws;                 // split-K: f32 partial tiles [group][tile][split][BM][BN]
    int n_split;
    long long ygs;             // element offset between the groups' output (and residual) slabs; Cout = side by side
    uint8_t* y8;               // optional second output: fp8 e4m3 copy of the result (groups side by side), value * q8
    int y8cs;
    float q8;
    // DUAL kernels (a stride-2 BasicBlock's conv1 + its 1x1/s2 downsample from one staged input): the 1x1 conv's operands
    const uint16_t* w2;        // [groups][Cout][Cin], operand type of w
    const float* scale2;
    const float* shift2;
    uint16_t* y2;              // bf16, groups side by side, pixel stride y2cs; no ReLU, no residual
    int y2cs;
    long long y2gs;            // conv_s2regh.inl: element offset between the groups' slabs of y2 (ygs is that of y)
    int xcd2d;                 // patch kernel, 2 groups on a flattened grid: XCD -> (group, half of the tiles, half of the channel tiles)
    // conv_wreg.inl: ceil(2^32 / d) for the tile decode's four divisors (ntn, ntn / 4, tiles_x, tiles_x * tiles_y): q = umulhi(n, magic) is
    // exact while n * d < 2^32 -- a scalar multiply instead of the ~40-instruction reciprocal sequence of a run-time division, four times
    // in front of a workgroup's first memory request
    unsigned mg_ntn, mg_qn, mg_tx, mg_txy;
    unsigned mg_hw, mg_wo;     // the same for the implicit-GEMM kernels' row -> (image, oy, ox) decode (divisors Ho * Wo, Wo): set by fill_args
};
static inline unsigned w2c_magic(unsigned d) { return d <= 1 ? 0u : (unsigned)(((1ull << 32) + d - 1) / d); }
__device__ __forceinline__ int w2c_fastdiv(int n, int d, unsigned magic) { return d <= 1 ? n : (int)__umulhi((unsigned)n, magic); }
// any 0 <= n < 2^31: floor(2^32 / d) under-estimates the quotient by at most 2 -- two conditional corrections (the persistent kernels'
// tile-run bounds: n = workgroup * tiles, far above the range of the one-multiply form)
static inline unsigned w2c_magic_floor(unsigned d) { return d <= 1 ? 0u : (unsigned)((1ull << 32) / d); }
__device__ __forceinline__ int w2c_fastdiv2(int n, int d, unsigned magic) {
    if (d <= 1) return n;
    int q = (int)__umulhi((unsigned)n, magic);
    int r = n - q * d;
    if (r >= d) { ++q; r -= d; }
    if (r >= d) ++q;
    return q;
}

// element size / channels per K-step of the two operand types: a K-step is always ONE 128-byte run per row
template <bool F8> struct OpT { static constexpr int ES = F8 ? 1 : 2; static constexpr int CK = F8 ? 128 : 64; };

// 8 consecutive output channels of one pixel -> the outputs selected by (p.y, p.y_f32, p.y8)
__device__ __forceinline__ void store_out8(const ConvArgs& p, const float (&v)[8], size_t off, size_t off8) {
    if (p.y) {
        if (p.y_f32) {
            float* yo = reinterpret_cast<float*>(p.y) + off;
            *reinterpret_cast<f32x4_t*>(yo) = f32x4_t{v[0], v[1], v[2], v[3]};
            *reinterpret_cast<f32x4_t*>(yo + 4) = f32x4_t{v[4], v[5], v[6], v[7]};
        } else {
            uint4 o;
            o.x = pack_bf16x2(v[0], v[1]); o.y = pack_bf16x2(v[2], v[3]);
            o.z = pack_bf16x2(v[4], v[5]); o.w = pack_bf16x2(v[6], v[7]);
            *reinterpret_cast<uint4*>(reinterpret_cast<uint16_t*>(p.y) + off) = o;
        }
    }
    if (p.y8) {
        uint2 o;
        o.x = pack_fp8x4(v[0] * p.q8, v[1] * p.q8, v[2] * p.q8, v[3] * p.q8);
        o.y = pack_fp8x4(v[4] * p.q8, v[5] * p.q8, v[6] * p.q8, v[7] * p.q8);
        *reinterpret_cast<uint2*>(p.y8 + off8) = o;
    }
}


__device__ __forceinline__ void span_stamp(const ConvArgs& p, bool end) {
    if (p.span && threadIdx.x == 0) {
        if (end) atomicMax(p.span + 1, (unsigned long long)wall_clock64());
        else atomicMin(p.span, (unsigned long long)wall_clock64());
    }
}
__device__ __forceinline__ void dbg_stamp(const ConvArgs& p, int slot) {
    if (p.dbg && threadIdx.x == 0)
        p.dbg[((size_t)blockIdx.y * gridDim.x + blockIdx.x) * 4 + slot] = wall_clock64();
    if (slot == 0) span_stamp(p, false);
    if (slot == 3) span_stamp(p, true);
}

// Tap order of the stride-2 3x3 convs: grouped by input phase (see conv3x3s2_patch_kernel) -- every kernel that computes a
// stride-2 3x3 conv walks the taps of a chunk in THIS order, so the halo-patch kernel and the generic kernel stay bit-identical.
__device__ __forceinline__ void s2_tap(int i, int& ky, int& kx) {
    ky = (int)((0x002202111ull >> (4 * (8 - i))) & 0xF);    // i: 0..8 -> ky = 0 0 2 2 0 2 1 1 1   (9 nibbles: 64-bit constants)
    kx = (int)((0x020211021ull >> (4 * (8 - i))) & 0xF);    //             kx = 0 2 0 2 1 1 0 2 1
}

template <int N>
__device__ __forceinline__ void wait_vmcnt() {
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}
// Barrier of the LDS pipelines.  A raw s_barrier waits for nothing: the compiler is free to schedule the
// lgkmcnt wait (and the MFMAs) of this wave's last fragment reads AFTER it, so without the lgkmcnt(0) here a
// wave could pass the barrier with ds_reads still in flight while another wave's DMA -- issued right after the
// barrier -- overwrites that ring slot (seen as rare corrupted tiles at 16 waves/CU; tools/_stress_variants.py).
// The empty asm after it keeps the next step's LDS reads / DMA from being hoisted above the barrier.
__device__ __forceinline__ void pipeline_barrier() {
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
}

// SPLITK: gridDim.z workgroups share one output tile, each accumulating a contiguous range of the K-steps; partial
// tiles go to p.ws as f32 and splitk_finish_kernel sums them in split order (fixed order: deterministic) and runs
// the epilogue.  For the tail layers (<= 80 tiles under a long, weight-streaming K loop) this turns a 36-step
// latency chain into 4-5 steps on 8x the CUs.
// DUAL: a 3x3 / stride-2 / pad-1 conv whose centre tap reads exactly the pixels of the block's 1x1 / stride-2 downsample
// (input (2oy, 2ox)): at the centre-tap K-steps the staged pixel tile is multiplied with a second weight tile into a
// second accumulator set, and the epilogue runs twice.  Same MFMA sequence per output as the two separate launches
// (bit-identical), one launch and one pass over the input less per stride-2 block.
template <int BM, int BN, int WM, int WN, int BK, int STAGES, bool SPLITK = false, bool F8 = false, bool DUAL = false>
__global__ __launch_bounds__(64 * WM * WN) void conv_igemm_kernel(ConvArgs p) {
#if defined(__HIP_DEVICE_COMPILE__)   // body uses device-only buffer-descriptor builtins; the host pass only needs the stub
    W2C_FORCE_AGPR_FORM();
    constexpr int NW = WM * WN, NT = 64 * NW;
    constexpr int WTM = BM / WM, WTN = BN / WN;        // wave tile
    constexpr int MI = WTM / 32, NI = WTN / 32;        // 32x32 MFMA tiles per wave
    constexpr int ES = OpT<F8>::ES, CK = OpT<F8>::CK;  // operand bytes per element / channels per K-step
    constexpr int ROWB = BK * 2;                       // bytes per LDS row (128: 64 bf16 or 128 fp8 channels)
    constexpr int LPR = ROWB / 16;                     // lanes (16-B chunks) per row: 8 or 4
    constexpr int RPI = 64 / LPR;                      // rows per DMA wave-instruction: 8 or 16
    constexpr int A_INSTR = BM / RPI / NW;             // DMA instructions per wave per K-step
    constexpr int B_INSTR = BN / RPI / NW;
    constexpr int LOADS = A_INSTR + B_INSTR;
    constexpr int A_BYTES = BM * ROWB, B_BYTES = BN * ROWB;
    constexpr int STAGE_BYTES = A_BYTES + B_BYTES;
    constexpr int KSUB = BK / 16;                      // MFMA K-steps per stage
    constexpr int CLD = BN + 4;                        // f32 epilogue tile leading dim
    static_assert(BK == 64, "BK: one K-step = one 64-channel chunk of one tap (the K order every kernel shares)");
    static_assert(A_INSTR >= 1 && B_INSTR >= 1 && A_INSTR * RPI * NW == BM && B_INSTR * RPI * NW == BN, "DMA split");
    static_assert(STAGES >= 2 && (STAGES - 2) * LOADS < 64, "vmcnt range");
    static_assert(!DUAL || (STAGES == 2 && !SPLITK), "DUAL: the extra tile's DMAs rely on the 2-stage ring's full vmcnt drain");

    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* const B2s = smem + STAGES * STAGE_BYTES;     // DUAL: the downsample's weight tile (single buffer, used once per chunk)

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int g = blockIdx.y;

    dbg_stamp(p, 0);
    // ---- tile id (XCD-aware): n fastest so the column tiles of one row panel share an L2 ----
    const int tile = xcd_remap(blockIdx.x, p.ntm * p.ntn);
    const int tm = tile / p.ntn, tn = tile - tm * p.ntn;
    const int m0 = tm * BM, n0 = tn * BN;

    const char* xg = reinterpret_cast<const char*>(p.x) + (size_t)g * p.Cin * ES;      // group's channel slice
    const int Ktot = p.ktiles * CK;                                                    // elements per weight row
    const char* wg = reinterpret_cast<const char*>(p.w) + (size_t)g * p.Cout * Ktot * ES;   // group's weights

    // ---- DMA addressing: buffer loads to LDS (SGPR descriptor + per-lane 32-bit byte offset + scalar
    // channel-chunk offset).  A row's offset for tap (ky,kx) is base + tap_off with tap_off wave-uniform;
    // taps that fall outside the image (and rows past the end) get an offset past the descriptor's extent, so
    // the hardware bounds check writes zeros into LDS -- no zero page, no 64-bit per-lane address math. ----
    const __amdgpu_buffer_rsrc_t rs_x = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<char*>(xg), 0, (int)((size_t)p.M * p.H * p.W * p.xcs * ES), 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_w = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<char*>(wg), 0, (int)((size_t)p.Cout * Ktot * ES), 0x00020000);
    const char* wg2 = DUAL ? reinterpret_cast<const char*>(p.w2) + (size_t)g * p.Cout * p.Cin * ES : wg;
    const __amdgpu_buffer_rsrc_t rs_w2 = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<char*>(wg2), 0, (int)((size_t)p.Cout * p.Cin * ES), 0x00020000);
    const int lrow = lane / LPR;           // row within a DMA group
    const int lpos = lane % LPR;           // 16-B position within the LDS row
    auto swz = [](int r) { return BK == 64 ? ((r >> 1) & 7) : ((r >> 2) & 3); };
    int a_iy0[A_INSTR], a_ix0[A_INSTR];
    int a_base[A_INSTR];                   // byte offset of (image, iy0, ix0, chunk); may be negative before the tap is added
#pragma unroll
    for (int j = 0; j < A_INSTR; ++j) {
        const int r = (wave + NW * j) * RPI + lrow;
        const int gr = m0 + r;
        const int chunk = lpos ^ swz(r);
        if (gr < p.rows) {
            const int hw = p.Ho * p.Wo;
            const int m = w2c_fastdiv(gr, hw, p.mg_hw);              // (rows < 2^29, hw <= rows: exact; round 6)
            const int rem = gr - m * hw;
            const int oy = w2c_fastdiv(rem, p.Wo, p.mg_wo), ox = rem - oy * p.Wo;
            a_iy0[j] = oy * p.stride - p.pad;
            a_ix0[j] = ox * p.stride - p.pad;
            a_base[j] = (int)(((long)m * p.H * p.W + (long)a_iy0[j] * p.W + a_ix0[j]) * p.xcs * ES + chunk * 16);
        } else {
            a_iy0[j] = -100000; a_ix0[j] = 0; a_base[j] = 0;      // every tap out of range -> zeros
        }
    }
    unsigned b_off[B_INSTR];
#pragma unroll
    for (int j = 0; j < B_INSTR; ++j) {
        const int n = (wave + NW * j) * RPI + lrow;
        b_off[j] = (unsigned)((size_t)(n0 + n) * Ktot * ES + (lpos ^ swz(n)) * 16);
    }
    unsigned b2_off[B_INSTR];
#pragma unroll
    for (int j = 0; j < B_INSTR; ++j) {
        const int n = (wave + NW * j) * RPI + lrow;
        b2_off[j] = (unsigned)((size_t)(n0 + n) * p.Cin * ES + (lpos ^ swz(n)) * 16);
    }

    // K-step range of this workgroup and cursor (wave-uniform): tap (ky,kx) and channel chunk
    const int n_split = SPLITK ? (int)gridDim.z : 1;
    const int z = SPLITK ? (int)blockIdx.z : 0;
    const int t_begin = SPLITK ? (z * p.ktiles) / n_split : 0;
    const int t_end = SPLITK ? ((z + 1) * p.ktiles) / n_split : p.ktiles;
    // K order = (channel chunk, tap) with the TAP fastest -- the order the patch kernels need (one staged patch per
    // chunk) -- so that every kernel of this file adds the same MFMA products in the same sequence and the results do
    // not depend on which variant pick_variant() chose (i.e. on the image count: shard == unsharded batch, bit for bit)
    const int ntap = p.ks * p.ks;
    const bool s2order = p.ks == 3 && p.stride == 2;             // stride-2 3x3: phase-grouped tap order (s2_tap)
    int st_ct = t_begin / ntap;
    int st_tap = t_begin - st_ct * ntap;
    int st_ky = st_tap / p.ks, st_kx = st_tap % p.ks;
    if (s2order) s2_tap(st_tap, st_ky, st_kx);

    auto stage = [&](int buf) {
        char* As = smem + buf * STAGE_BYTES;
        char* Bs = As + A_BYTES;
        const int tap_off = (st_ky * p.W + st_kx) * p.xcs * ES;      // wave-uniform, bytes
        const int c_off = st_ct * BK * 2;                            // channel chunk, bytes (scalar offset)
#pragma unroll
        for (int j = 0; j < A_INSTR; ++j) {
            const int iy = a_iy0[j] + st_ky, ix = a_ix0[j] + st_kx;
            const bool ok = ((unsigned)iy < (unsigned)p.H) & ((unsigned)ix < (unsigned)p.W);
            const unsigned vo = ok ? (unsigned)(a_base[j] + tap_off) : 0x80000000u;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_x, W2C_LPTR(As + (wave + NW * j) * 1024), 16, vo, c_off, 0, 0);
        }
#pragma unroll
        for (int j = 0; j < B_INSTR; ++j)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_w, W2C_LPTR(Bs + (wave + NW * j) * 1024), 16, b_off[j],
                                                     (st_ky * p.ks + st_kx) * p.Cin * ES + st_ct * BK * 2, 0, 0);
        if constexpr (DUAL) {
            if (st_ky == 1 && st_kx == 1) {              // centre tap: the 1x1 conv's weight tile of this channel chunk
#pragma unroll
                for (int j = 0; j < B_INSTR; ++j)
                    __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_w2, W2C_LPTR(B2s + (wave + NW * j) * 1024), 16, b2_off[j],
                                                             st_ct * BK * 2, 0, 0);
            }
        }
        if (++st_tap == ntap) { st_tap = 0; ++st_ct; }
        if (s2order) s2_tap(st_tap, st_ky, st_kx);
        else { st_ky = st_tap / p.ks; st_kx = st_tap - st_ky * p.ks; }
    };

    f32x16_t acc[MI][NI];
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int j = 0; j < NI; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

    f32x16_t acc2[DUAL ? MI : 1][DUAL ? NI : 1];
    if constexpr (DUAL) {
#pragma unroll
        for (int i = 0; i < MI; ++i)
#pragma unroll
            for (int j = 0; j < NI; ++j)
#pragma unroll
                for (int e = 0; e < 16; ++e) acc2[i][j][e] = 0.f;
    }

    const int wm = wave / WN, wn = wave - wm * WN;
    const int l31 = lane & 31, lhi = lane >> 5;
    const int lswz = swz(l31);             // tile / wave offsets are multiples of 32: swz(row) == swz(l31)
    // epilogue constants fetched now (their latency hides under the main loop): BN scale / shift of the 8 channels this
    // thread writes out (read-out item = (tile row, 8-channel group cg = tid % (BN/8)), the same cg in every pass)
    const float* ssp = p.scale + g * p.Cout + n0 + (tid % (BN / 8)) * 8;
    const float* shp = p.shift + g * p.Cout + n0 + (tid % (BN / 8)) * 8;
    const f32x4_t e_sc0 = *reinterpret_cast<const f32x4_t*>(ssp), e_sc1 = *reinterpret_cast<const f32x4_t*>(ssp + 4);
    const f32x4_t e_sh0 = *reinterpret_cast<const f32x4_t*>(shp), e_sh1 = *reinterpret_cast<const f32x4_t*>(shp + 4);

    auto compute = [&](int buf, bool centre) {
        const char* As = smem + buf * STAGE_BYTES;
        const char* Bs = As + A_BYTES;
        if constexpr (F8) {
            // fp8 e4m3 operands: the 128-byte row is 128 channels = two K=64 MX-scaled MFMAs (all block scales 2^0:
            // per-channel / per-tensor scales live in the epilogue's scale[]).  Lane-half lhi of MFMA j takes the 32
            // bytes at chunks {4j + 2 lhi, 4j + 2 lhi + 1} of its row -- the SAME bytes-to-K assignment for the pixel
            // and the weight operand, which is all a dot product needs.
            i32x8_t a8[2][MI], b8[2][NI];
#pragma unroll
            for (int j2 = 0; j2 < 2; ++j2) {
                const int p0 = ((j2 * 4 + lhi * 2) ^ lswz) << 4, p1 = ((j2 * 4 + lhi * 2 + 1) ^ lswz) << 4;
#pragma unroll
                for (int i = 0; i < MI; ++i) {
                    const char* r = As + (wm * WTM + i * 32 + l31) * ROWB;
                    a8[j2][i] = cat_i32x8(*reinterpret_cast<const u32x4_t*>(r + p0), *reinterpret_cast<const u32x4_t*>(r + p1));
                }
#pragma unroll
                for (int j = 0; j < NI; ++j) {
                    const char* r = Bs + (wn * WTN + j * 32 + l31) * ROWB;
                    b8[j2][j] = cat_i32x8(*reinterpret_cast<const u32x4_t*>(r + p0), *reinterpret_cast<const u32x4_t*>(r + p1));
                }
            }
#pragma unroll
            for (int j2 = 0; j2 < 2; ++j2)
#pragma unroll
                for (int i = 0; i < MI; ++i)
#pragma unroll
                    for (int j = 0; j < NI; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(b8[j2][j], a8[j2][i], acc[i][j], 0, 0, 0,
                                                                                    0x7F7F7F7F, 0, 0x7F7F7F7F);
#pragma unroll
            for (int i = 0; i < MI; ++i)             // keep the MFMAs of this K-step in this K-step (see the patch kernel)
#pragma unroll
                for (int j = 0; j < NI; ++j) asm volatile("" : "+v"(acc[i][j]));
            if constexpr (DUAL) {
                if (centre) {                         // wave-uniform
#pragma unroll
                    for (int j2 = 0; j2 < 2; ++j2) {
                        const int p0 = ((j2 * 4 + lhi * 2) ^ lswz) << 4, p1 = ((j2 * 4 + lhi * 2 + 1) ^ lswz) << 4;
#pragma unroll
                        for (int j = 0; j < NI; ++j) {
                            const char* r = B2s + (wn * WTN + j * 32 + l31) * ROWB;
                            const i32x8_t w8 = cat_i32x8(*reinterpret_cast<const u32x4_t*>(r + p0), *reinterpret_cast<const u32x4_t*>(r + p1));
#pragma unroll
                            for (int i = 0; i < MI; ++i)
                                acc2[i][j] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(w8, a8[j2][i], acc2[i][j], 0, 0, 0,
                                                                                             0x7F7F7F7F, 0, 0x7F7F7F7F);
                        }
                    }
#pragma unroll
                    for (int i = 0; i < MI; ++i)
#pragma unroll
                        for (int j = 0; j < NI; ++j) asm volatile("" : "+v"(acc2[i][j]));
                }
            }
            return;
        }
        bf16x8_t a[KSUB][MI], b[KSUB][NI];
#pragma unroll
        for (int kk = 0; kk < KSUB; ++kk) {
            const int pos = ((kk * 2 + lhi) ^ lswz) << 4;
#pragma unroll
            for (int i = 0; i < MI; ++i)
                a[kk][i] = *reinterpret_cast<const bf16x8_t*>(As + (wm * WTM + i * 32 + l31) * ROWB + pos);
#pragma unroll
            for (int j = 0; j < NI; ++j)
                b[kk][j] = *reinterpret_cast<const bf16x8_t*>(Bs + (wn * WTN + j * 32 + l31) * ROWB + pos);
        }
#pragma unroll
        for (int kk = 0; kk < KSUB; ++kk)
#pragma unroll
            for (int i = 0; i < MI; ++i)
#pragma unroll
                for (int j = 0; j < NI; ++j)
                    // A = weights, B = pixels: D[channel][pixel] -- four consecutive channels of a pixel per accumulator quad
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(b[kk][j], a[kk][i], acc[i][j], 0, 0, 0);
        if constexpr (DUAL) {
            if (centre) {                             // wave-uniform: the downsample's products, same pixel fragments
#pragma unroll
                for (int kk = 0; kk < KSUB; ++kk) {
                    const int pos = ((kk * 2 + lhi) ^ lswz) << 4;
#pragma unroll
                    for (int j = 0; j < NI; ++j) {
                        const bf16x8_t w2f = *reinterpret_cast<const bf16x8_t*>(B2s + (wn * WTN + j * 32 + l31) * ROWB + pos);
#pragma unroll
                        for (int i = 0; i < MI; ++i)
                            acc2[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w2f, a[kk][i], acc2[i][j], 0, 0, 0);
                    }
                }
            }
        }
    };

    // ---- main loop: STAGES-deep ring, counted waits, one barrier per K-step ----
    const int KT = t_end - t_begin;
#pragma unroll
    for (int s = 0; s < STAGES - 1; ++s)
        if (s < KT) stage(s);
    int rd = 0;                      // buffer holding tile t
    int wr = STAGES - 1;             // buffer the next DMA goes to (== buffer of tile t-1)
    int c_tap = 0;                   // DUAL: tap index (0..8) of the tile being computed (t_begin = 0 there)
    for (int t = 0; t < KT; ++t) {
        if (t + STAGES - 2 < KT) wait_vmcnt<(STAGES - 2) * LOADS>();     // tile t (my part) has landed
        else wait_vmcnt<0>();
        pipeline_barrier();                                              // everyone's part landed; tile t-1's buffer is free
        if (t == 0) dbg_stamp(p, 1);
        if (t + STAGES - 1 < KT) stage(wr);
        compute(rd, DUAL && c_tap == 8);              // the centre tap (1,1) is the LAST tap of the stride-2 order
        if (DUAL) c_tap = (c_tap == 8) ? 0 : c_tap + 1;
        rd = (rd + 1 == STAGES) ? 0 : rd + 1;
        wr = (wr + 1 == STAGES) ? 0 : wr + 1;
    }
    // every wave done reading the ring; the epilogue reuses that LDS.  A FENCED barrier: a raw s_barrier does not
    // stop the compiler from hoisting the staging stores above it (observed as a rare corrupted tile).
    __syncthreads();
    dbg_stamp(p, 2);

    // ---- epilogue 0: residual addresses + loads FIRST (branch-free, clamped), so their HBM latency runs
    // under the LDS staging below and is paid once, not per pass ----
    constexpr int CG = BN / 8;                     // 8-channel groups per row
    constexpr int PASSES = BM * CG / NT;
    static_assert(PASSES * NT == BM * CG, "epilogue split");
    size_t e_off[PASSES], e_off8[PASSES];
    bool e_ok[PASSES];
    uint4 e_res[PASSES];
#pragma unroll
    for (int ps = 0; ps < PASSES; ++ps) {
        const int idx = ps * NT + tid;
        const int r = idx / CG, cg = idx - r * CG;
        const int gr = m0 + r;
        e_ok[ps] = gr < p.rows;
        e_off[ps] = (size_t)(e_ok[ps] ? gr : 0) * p.ycs + (size_t)g * p.ygs + n0 + cg * 8;
        e_off8[ps] = (size_t)(e_ok[ps] ? gr : 0) * p.y8cs + (size_t)g * p.Cout + n0 + cg * 8;
    }
    if (p.res && !SPLITK) {
#pragma unroll
        for (int ps = 0; ps < PASSES; ++ps) e_res[ps] = *reinterpret_cast<const uint4*>(p.res + e_off[ps]);
    } else {
#pragma unroll
        for (int ps = 0; ps < PASSES; ++ps) e_res[ps] = make_uint4(0, 0, 0, 0);
    }
    // ---- epilogue 1: scale/shift in registers, tile -> LDS as f32 [BM][CLD] ----
    float* Cs = reinterpret_cast<float*>(smem);
#pragma unroll
    for (int i = 0; i < MI; ++i) {
        const int ml = wm * WTM + i * 32 + l31;                   // D column = pixel (tile row)
#pragma unroll
        for (int j = 0; j < NI; ++j)
#pragma unroll
            for (int eg = 0; eg < 4; ++eg)                        // D rows 8 eg + 4 lhi .. +4 = four consecutive channels
                *reinterpret_cast<f32x4_t*>(Cs + ml * CLD + wn * WTN + j * 32 + eg * 8 + lhi * 4) =
                    f32x4_t{acc[i][j][eg * 4], acc[i][j][eg * 4 + 1], acc[i][j][eg * 4 + 2], acc[i][j][eg * 4 + 3]};
    }
    __syncthreads();

    if constexpr (SPLITK) {
        // partial tile -> workspace (coalesced 32-B pieces); splitk_finish_kernel (next launch on the stream) sums the
        // partial tiles in split order and runs the epilogue.  (An in-kernel "last arriver reduces" hand-off needs a
        // device-scope release/acquire per workgroup = an L2 write-back on this 8-XCD part: measured 27 -> 120 us.)
        const size_t tile_id = (size_t)g * (p.ntm * p.ntn) + tile;
        float* slab = p.ws + (tile_id * n_split + z) * (size_t)(BM * BN);
#pragma unroll
        for (int ps = 0; ps < PASSES; ++ps) {
            const int idx = ps * NT + tid;
            const int r = idx / CG, cg = idx - r * CG;
            *reinterpret_cast<f32x4_t*>(slab + r * BN + cg * 8) = *reinterpret_cast<const f32x4_t*>(Cs + r * CLD + cg * 8);
            *reinterpret_cast<f32x4_t*>(slab + r * BN + cg * 8 + 4) = *reinterpret_cast<const f32x4_t*>(Cs + r * CLD + cg * 8 + 4);
        }
        return;
    }

    // ---- epilogue 2: coalesced (+residual) (+ReLU) store, 8 channels per thread per pass ----
#pragma unroll
    for (int ps = 0; ps < PASSES; ++ps) {
        const int idx = ps * NT + tid;
        const int r = idx / CG, cg = idx - r * CG;
        const f32x4_t v0 = *reinterpret_cast<const f32x4_t*>(Cs + r * CLD + cg * 8) * e_sc0 + e_sh0;
        const f32x4_t v1 = *reinterpret_cast<const f32x4_t*>(Cs + r * CLD + cg * 8 + 4) * e_sc1 + e_sh1;
        float v[8] = {v0[0], v0[1], v0[2], v0[3], v1[0], v1[1], v1[2], v1[3]};
        const uint32_t rw[4] = {e_res[ps].x, e_res[ps].y, e_res[ps].z, e_res[ps].w};
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            v[2 * e] += bf16_to_f32((uint16_t)(rw[e] & 0xFFFFu));        // +0 when there is no residual
            v[2 * e + 1] += bf16_to_f32((uint16_t)(rw[e] >> 16));
        }
        if (p.relu) {
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] = fmaxf(v[e], 0.f);
        }
        if (e_ok[ps]) store_out8(p, v, e_off[ps], e_off8[ps]);
    }
    if constexpr (DUAL) {
        // ---- second output: the 1x1/s2 downsample (BN folded, no ReLU, no residual), bf16, groups side by side ----
        __syncthreads();                                  // every thread is done reading the first tile out of Cs
#pragma unroll
        for (int i = 0; i < MI; ++i) {
            const int ml = wm * WTM + i * 32 + l31;
#pragma unroll
            for (int j = 0; j < NI; ++j)
#pragma unroll
                for (int eg = 0; eg < 4; ++eg)
                    *reinterpret_cast<f32x4_t*>(Cs + ml * CLD + wn * WTN + j * 32 + eg * 8 + lhi * 4) =
                        f32x4_t{acc2[i][j][eg * 4], acc2[i][j][eg * 4 + 1], acc2[i][j][eg * 4 + 2], acc2[i][j][eg * 4 + 3]};
        }
        __syncthreads();
        const float* ssp2 = p.scale2 + g * p.Cout + n0 + (tid % (BN / 8)) * 8;
        const float* shp2 = p.shift2 + g * p.Cout + n0 + (tid % (BN / 8)) * 8;
        const f32x4_t sc0 = *reinterpret_cast<const f32x4_t*>(ssp2), sc1 = *reinterpret_cast<const f32x4_t*>(ssp2 + 4);
        const f32x4_t sh0 = *reinterpret_cast<const f32x4_t*>(shp2), sh1 = *reinterpret_cast<const f32x4_t*>(shp2 + 4);
#pragma unroll
        for (int ps = 0; ps < PASSES; ++ps) {
            const int idx = ps * NT + tid;
            const int r = idx / CG, cg = idx - r * CG;
            const f32x4_t v0 = *reinterpret_cast<const f32x4_t*>(Cs + r * CLD + cg * 8) * sc0 + sh0;
            const f32x4_t v1 = *reinterpret_cast<const f32x4_t*>(Cs + r * CLD + cg * 8 + 4) * sc1 + sh1;
            if (e_ok[ps]) {
                uint4 o;
                o.x = pack_bf16x2(v0[0], v0[1]); o.y = pack_bf16x2(v0[2], v0[3]);
                o.z = pack_bf16x2(v1[0], v1[1]); o.w = pack_bf16x2(v1[2], v1[3]);
                *reinterpret_cast<uint4*>(p.y2 + (size_t)(m0 + r) * p.y2cs + (size_t)g * p.Cout + n0 + cg * 8) = o;
            }
        }
    }
    dbg_stamp(p, 3);
#endif
}

// =====================================================================================================
// Patch-staged 3x3 stride-1 convolution.
//
// The generic kernel above re-reads every input pixel once per tap (9x) and every weight tile once
// per row tile; on the big layers that L2->LDS traffic (~25 TB/s aggregate, measured by ablation)
// costs as much time as the MFMAs.  Here a workgroup owns a 2-D tile of TH x TW output pixels of ONE
// image (256 pixels) and stages, per 64-channel chunk, the (TH+2) x (TW+2) input PATCH (halo
// included, zero outside the image) in LDS exactly once; all 9 taps read their A fragments from that
// patch at a shifted pixel index, so activation traffic drops ~8x and only the weight tiles (16 KB per
// tap for BN=128) stream through a STAGES-deep ring.  8 waves (4 x 2), wave tile 64 x (BN/2),
// one barrier per tap, counted vmcnt, patch double-buffered across channel chunks.
// LDS patch image: one 128-B row per patch pixel, 16-B chunk c of pixel q at c ^ ((q>>1)&7) -- the
// ds_read_b128 lane groups see 16 consecutive pixels (mod 16 distinct) => conflict-free for TW=32.
// (Measured and rejected in round 2, profiles/r02_conv_experiments.txt: a software-pipelined form that issues the fragment
//  reads of K-step t+1 before the MFMAs of K-step t from a second register set.  Correct and bit-identical, but the second
//  set costs 64-124 registers -- 92 -> 156 / 176 -> 300 -- i.e. ONE workgroup per CU instead of two, and the co-resident
//  workgroup is what hides this kernel's latencies: layer2 55 -> 84 us, layer3 53 -> 99 us, layer4 58 -> 92 us.)
template <int TH, int TW, int BN, int WM, int WN, int STAGES, int PB, bool F8 = false>
__global__ __launch_bounds__(64 * WM * WN) void conv3x3_patch_kernel(ConvArgs p) {
#if defined(__HIP_DEVICE_COMPILE__)   // body uses device-only buffer-descriptor builtins; the host pass only needs the stub
    W2C_FORCE_AGPR_FORM();
    constexpr int ES = OpT<F8>::ES, CK = OpT<F8>::CK;  // operand bytes per element / channels per 128-byte K-step
    constexpr int BM = TH * TW;
    constexpr int NW = WM * WN, NT = 64 * NW;
    constexpr int WTM = BM / WM, WTN = BN / WN;
    constexpr int MI = WTM / 32, NI = WTN / 32;
    constexpr int PW = TW + 2, PH = TH + 2, NP = PH * PW;
    constexpr int P_INSTR = (NP + 8 * NW - 1) / (8 * NW);   // patch DMA instructions per wave per chunk (8 pixels each,
                                                            // lanes past the last patch pixel are predicated off)
    constexpr int PATCH_BYTES = NP * 128;
    constexpr int B_BYTES = BN * 128;
    constexpr int B_INSTR = BN / 8 / NW;               // weight DMA instructions per wave per tap
    constexpr int CLD = BN + 4;
    static_assert(MI >= 1 && NI >= 1 && MI * 32 * WM == BM && NI * 32 * WN == BN && B_INSTR >= 1, "shape");
    static_assert(STAGES >= 2 && (STAGES - 2) * B_INSTR + P_INSTR < 64, "vmcnt range");

    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* patch0 = smem;
    char* bring = smem + PB * PATCH_BYTES;          // PB == 1: single channel chunk (Cin == 64), no patch double buffer

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lrow = lane >> 3, lpos = lane & 7;

    dbg_stamp(p, 0);
    // ---- tile decode: n fastest, then x, y, image ----
    const int tiles_x = p.W / TW, tiles_y = p.H / TH;
    int g, tsp, tn;
    if (p.xcd2d) {
        // weight-heavy layers (layer4: 4.7 MB of weights and 5.2 MB of input per group, 4 MB of L2 per XCD): with the tile
        // ids dealt out contiguously every XCD streams ALL the weights of both groups.  Here XCD x (= block id mod 8) owns
        // one group, one half of its spatial tiles and one half of its channel tiles: 2.6 + 2.35 MB per XCD instead of
        // 1.3 + 9.4.  Placement only -- the result does not depend on it.
        const int xcd = blockIdx.x & 7, loc = blockIdx.x >> 3;
        const int hn = p.ntn >> 1, hm = p.ntm >> 1;
        g = xcd >> 2;
        const int tm_l = w2c_fastdiv(loc, hn, p.mg_qn);               // (magic-number divisions, round 6: see ConvArgs)
        tn = (xcd & 1) * hn + (loc - tm_l * hn);
        tsp = ((xcd >> 1) & 1) * hm + tm_l;
    } else {
        g = blockIdx.y;
        const int tile = xcd_remap(blockIdx.x, p.ntm * p.ntn);
        tsp = w2c_fastdiv(tile, p.ntn, p.mg_ntn);
        tn = tile - tsp * p.ntn;
    }
    const int img = w2c_fastdiv(tsp, tiles_x * tiles_y, p.mg_txy);
    const int trem = tsp - img * (tiles_x * tiles_y);
    const int tyi = w2c_fastdiv(trem, tiles_x, p.mg_tx);
    const int txi = trem - tyi * tiles_x;
    const int y0 = tyi * TH, x0 = txi * TW, n0 = tn * BN;

    const char* xg = reinterpret_cast<const char*>(p.x) + (size_t)g * p.Cin * ES;
    const int Ktot = 9 * p.Cin;
    const char* wg = reinterpret_cast<const char*>(p.w) + (size_t)g * p.Cout * Ktot * ES;
    const int nchunks = p.Cin / CK;
    const int KT = nchunks * 9;

    // ---- DMA addressing: buffer loads to LDS.  Each lane's byte offset inside the tensor is fixed for the
    // whole tile (a VGPR), the K-step / channel-chunk advance is a scalar offset, the base lives in an SGPR
    // descriptor: no per-step VALU address math (measured: ~500 of ~1560 cycles per K-step went into issuing
    // flat 64-bit-address LDS-DMA, tools/conv_phases.py).  Halo pixels outside the image carry an offset
    // past the descriptor's extent, so the hardware bounds check writes zeros -- no zero page, no select.
    const __amdgpu_buffer_rsrc_t rs_x = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<char*>(xg), 0, (int)((size_t)p.M * p.H * p.W * p.xcs * ES), 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_w = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<char*>(wg), 0, (int)((size_t)p.Cout * Ktot * ES), 0x00020000);
    // Rounds whose first pixel is past the patch are skipped by the whole wave (wave-uniform test), so the
    // number of VMEM ops a wave has in flight is EXACTLY known: the last round exists only for some waves.
    const bool p_last = (wave + NW * (P_INSTR - 1)) * 8 < NP;       // does this wave issue round P_INSTR-1 ?
    unsigned pa_off[P_INSTR];                                       // byte offset of this lane's 16 B, or out of range
#pragma unroll
    for (int j = 0; j < P_INSTR; ++j) {
        const int q = (wave + NW * j) * 8 + lrow;                    // patch pixel index
        const int py = q / PW, px = q - py * PW;
        const int chunk = lpos ^ ((px >> 1) & 7);                    // swizzle keyed on the patch COLUMN (see load_frags)
        const int iy = y0 - 1 + py, ix = x0 - 1 + px;
        const bool ok = ((unsigned)iy < (unsigned)p.H) & ((unsigned)ix < (unsigned)p.W);
        pa_off[j] = ok ? (unsigned)(((img * p.H + iy) * p.W + ix) * p.xcs * ES + chunk * 16) : 0x80000000u;   // (< 2 GiB: fill_args)
    }
    auto issue_patch = [&](int cc, int buf) {
        char* dst = patch0 + buf * PATCH_BYTES;
#pragma unroll
        for (int j = 0; j < P_INSTR; ++j) {
            if ((wave + NW * j) * 8 < NP) {                          // wave-uniform
                if ((wave + NW * j) * 8 + lrow < NP)   // lanes past the last patch pixel do not write LDS (no padding)
                    __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_x, W2C_LPTR(dst + (wave + NW * j) * 1024), 16,
                                                             pa_off[j], cc * 128, 0, 0);
            }
        }
    };
    // wait until at most (K weight-tile DMAs + one patch chunk) of this wave's VMEM ops are outstanding
    auto wait_tiles_and_patch = [&](auto ktag) {
        constexpr int K = decltype(ktag)::value;
        if (p_last) wait_vmcnt<K * B_INSTR + P_INSTR>();
        else wait_vmcnt<K * B_INSTR + P_INSTR - 1>();
    };
    // ---- weight tile addresses ----
    unsigned b_off[B_INSTR];
#pragma unroll
    for (int j = 0; j < B_INSTR; ++j) {
        const int n = (wave + NW * j) * 8 + lrow;
        b_off[j] = (unsigned)((size_t)(n0 + n) * Ktot * ES + (lpos ^ ((n >> 1) & 7)) * 16);
    }
    int st_tap = 0, st_cc = 0;                                  // cursor of the next weight tile to issue
    auto issue_b = [&](int buf) {
        char* dst = bring + buf * B_BYTES;
        const int koff = st_tap * p.Cin * ES + st_cc * 128;     // scalar byte offset of the K-step
#pragma unroll
        for (int j = 0; j < B_INSTR; ++j)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_w, W2C_LPTR(dst + (wave + NW * j) * 1024), 16, b_off[j], koff, 0, 0);
        if (++st_tap == 9) { st_tap = 0; ++st_cc; }
    };

    f32x16_t acc[MI][NI];
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int j = 0; j < NI; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

    const int wm = wave / WN, wn = wave - wm * WN;
    const int l31 = lane & 31, lhi = lane >> 5;
    int pp0[MI], pc0[MI];                                       // patch pixel / patch column of this lane's row at tap (0,0)
#pragma unroll
    for (int i = 0; i < MI; ++i) {
        const int tp = wm * WTM + i * 32 + l31;
        pc0[i] = tp % TW;
        pp0[i] = (tp / TW) * PW + pc0[i];
    }
    const int bswz = (l31 >> 1) & 7;
    // epilogue constants (latency hidden by the main loop): BN scale / shift of the 8 channels this thread writes out
    // (read-out item = (tile row, 8-channel group cg = tid % (BN/8)), the same cg in every pass)
    const float* ssp = p.scale + g * p.Cout + n0 + (tid % (BN / 8)) * 8;
    const float* shp = p.shift + g * p.Cout + n0 + (tid % (BN / 8)) * 8;
    const f32x4_t e_sc0 = *reinterpret_cast<const f32x4_t*>(ssp), e_sc1 = *reinterpret_cast<const f32x4_t*>(ssp + 4);
    const f32x4_t e_sh0 = *reinterpret_cast<const f32x4_t*>(shp), e_sh1 = *reinterpret_cast<const f32x4_t*>(shp + 4);

    struct Frags {
        bf16x8_t fa[F8 ? 1 : 4][MI], fb[F8 ? 1 : 4][NI];       // fragment registers of ONE K-step (tap)
        i32x8_t fa8[F8 ? 2 : 1][MI], fb8[F8 ? 2 : 1][NI];      // fp8 form: two K=64 MFMAs per 128-byte row
    };
    Frags fr0, fr1;                                             // fr1 only lives in the PIPE form
    // LDS slot of chunk c of patch pixel (row, col): c ^ ((col >> 1) & 7), at byte (row*PW + col)*128.  Keyed on the COLUMN:
    // a wave's 32 pixels lie on two patch rows when TW = 16, and ds_read_b128 services lanes {0-3,12-15,20-27} (etc.)
    // together -- columns {c..c+3, c+12..c+15} of one row and {c+4..c+11} of the next, a complete residue system mod 16 =>
    // 16 distinct 16-B slots of the 256-B bank row.  (Keyed on the linear pixel index the two rows collide: PMC showed
    // 33 % of the LDS cycles of these kernels were bank conflicts.)
    auto load_frags = [&](Frags& fr, const char* patch, const char* Bs, int dk, int kx) {
        auto& fa = fr.fa; auto& fb = fr.fb; auto& fa8 = fr.fa8; auto& fb8 = fr.fb8;
        const char* arow[MI];
        int aswz[MI];
#pragma unroll
        for (int i = 0; i < MI; ++i) {
            const int pp = pp0[i] + dk;
            arow[i] = patch + pp * 128;
            aswz[i] = ((pc0[i] + kx) >> 1) & 7;
        }
        if constexpr (F8) {
#pragma unroll
            for (int j2 = 0; j2 < 2; ++j2) {
                const int c0 = j2 * 4 + lhi * 2;
#pragma unroll
                for (int i = 0; i < MI; ++i)
                    fa8[j2][i] = cat_i32x8(*reinterpret_cast<const u32x4_t*>(arow[i] + ((c0 ^ aswz[i]) << 4)),
                                           *reinterpret_cast<const u32x4_t*>(arow[i] + (((c0 + 1) ^ aswz[i]) << 4)));
#pragma unroll
                for (int j = 0; j < NI; ++j) {
                    const char* r = Bs + (wn * WTN + j * 32 + l31) * 128;
                    fb8[j2][j] = cat_i32x8(*reinterpret_cast<const u32x4_t*>(r + ((c0 ^ bswz) << 4)),
                                           *reinterpret_cast<const u32x4_t*>(r + (((c0 + 1) ^ bswz) << 4)));
                }
            }
            return;
        }
#pragma unroll
        for (int kk = 0; kk < (F8 ? 1 : 4); ++kk) {
#pragma unroll
            for (int i = 0; i < MI; ++i)
                fa[kk][i] = *reinterpret_cast<const bf16x8_t*>(arow[i] + (((kk * 2 + lhi) ^ aswz[i]) << 4));
#pragma unroll
            for (int j = 0; j < NI; ++j)
                fb[kk][j] = *reinterpret_cast<const bf16x8_t*>(Bs + (wn * WTN + j * 32 + l31) * 128 +
                                                               (((kk * 2 + lhi) ^ bswz) << 4));
        }
    };
    auto mfma_all = [&](Frags& fr) {
        auto& fa = fr.fa; auto& fb = fr.fb; auto& fa8 = fr.fa8; auto& fb8 = fr.fb8;
        if constexpr (F8) {
#pragma unroll
            for (int j2 = 0; j2 < 2; ++j2)
#pragma unroll
                for (int i = 0; i < MI; ++i)
#pragma unroll
                    for (int j = 0; j < NI; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(fb8[j2][j], fa8[j2][i], acc[i][j], 0, 0, 0,
                                                                                    0x7F7F7F7F, 0, 0x7F7F7F7F);
            // pin this step's MFMAs here: left alone, the compiler sinks the (pure) MFMA calls of up to 17 K-steps to the
            // end of the tap loop and keeps all their fragments live -- 512 VGPRs + scratch, 1 workgroup per CU (3x slower)
#pragma unroll
            for (int i = 0; i < MI; ++i)
#pragma unroll
                for (int j = 0; j < NI; ++j) asm volatile("" : "+v"(acc[i][j]));
            return;
        }
#pragma unroll
        for (int kk = 0; kk < (F8 ? 1 : 4); ++kk)
#pragma unroll
            for (int i = 0; i < MI; ++i)
#pragma unroll
                for (int j = 0; j < NI; ++j)
                    // A = weights, B = pixels: D[channel][pixel], so a lane holds 4 CONSECUTIVE CHANNELS of one pixel per
                    // accumulator quad and the epilogue stages them with one ds_write_b128 instead of four ds_write_b32
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fb[kk][j], fa[kk][i], acc[i][j], 0, 0, 0);
    };
    using KS2 = std::integral_constant<int, STAGES - 2>;

    // ---- lock-step pipeline: patch(cc+1) prefetched at tap 0 of chunk cc; weight tiles STAGES-1 ahead ----
    // (Measured and rejected on MI355X, see profiles/r01_b_conv_variant_sweep.txt: an 8-wave "ping-pong"
    //  variant with wave groups half an epoch apart, a runtime tap loop, a persistent cross-tile-
    //  prefetching version with a 32-row multi-pass epilogue, and asm-pinned MFMAs with the step's DMA pieces
    //  interleaved between its four MFMA groups (l2 58 -> 63 us, l3 60 -> 67 us) -- all slower than this form.)
#ifdef W2C_PHASE_TIMING
    long long ph[5] = {0, 0, 0, 0, 0};
#endif
    issue_patch(0, 0);
#pragma unroll
    for (int s = 0; s < STAGES - 1; ++s) issue_b(s);           // KT >= 9 > STAGES-1
    int rd = 0, wr = STAGES - 1, t = 0;
    for (int cc = 0; cc < nchunks; ++cc) {
        const char* patch = patch0 + (PB == 2 ? (cc & 1) : 0) * PATCH_BYTES;
        const bool more = cc + 1 < nchunks;
#pragma unroll
        for (int tap = 0; tap < 9; ++tap, ++t) {
            // wait until weight tile t (and everything older, incl. this chunk's patch) has landed.
            // Younger loads allowed in flight: STAGES-2 weight tiles, plus patch(cc+1) while it is
            // younger than tile t (it was issued right after tile t0+STAGES-1 at tap 0).
#ifdef W2C_PHASE_TIMING
            const long long c0 = clock64();
#endif
            if (t + STAGES - 2 < KT) {
                if (more && tap >= 1 && tap <= STAGES - 1) wait_tiles_and_patch(KS2{});
                else wait_vmcnt<(STAGES - 2) * B_INSTR>();
            } else {
                wait_vmcnt<0>();
            }
#ifdef W2C_PHASE_TIMING
            const long long c1 = clock64();
#endif
            pipeline_barrier();
#ifdef W2C_PHASE_TIMING
            const long long c2 = clock64();
#endif
            if (t == 0) dbg_stamp(p, 1);
#ifdef W2C_READS_FIRST     // measured: fragment reads ahead of the DMA issue is SLOWER (l2 v30 55 -> 66 us at equal clocks)
            load_frags(fr0, patch, bring + rd * B_BYTES, (tap / 3) * PW + (tap % 3), tap % 3);
#endif
            if (t + STAGES - 1 < KT) issue_b(wr);
            if (tap == 0 && more) issue_patch(cc + 1, (cc + 1) & 1);
#ifdef W2C_PHASE_TIMING
            const long long c3 = clock64();
#endif
#ifndef W2C_READS_FIRST
            load_frags(fr0, patch, bring + rd * B_BYTES, (tap / 3) * PW + (tap % 3), tap % 3);
#endif
#ifdef W2C_PHASE_TIMING
            asm volatile("" ::: "memory");
            const long long c4 = clock64();
#endif
            mfma_all(fr0);
#ifdef W2C_PHASE_TIMING
            const long long c5 = clock64();
            ph[0] += c1 - c0; ph[1] += c2 - c1; ph[2] += c3 - c2; ph[3] += c4 - c3; ph[4] += c5 - c4;
#endif
            rd = (rd + 1 == STAGES) ? 0 : rd + 1;
            wr = (wr + 1 == STAGES) ? 0 : wr + 1;
        }
    }
    __syncthreads();     // fenced: the epilogue's staging stores must not be hoisted above this barrier
    dbg_stamp(p, 2);
#ifdef W2C_PHASE_TIMING
    if (p.dbg && (threadIdx.x & 63) == 0) {          // lane 0 of every wave: [WG][wave][5] after the 4 stamps region
        unsigned long long* o = p.dbg + (1u << 19) + (((size_t)blockIdx.y * gridDim.x + blockIdx.x) * NW + wave) * 5;
        for (int k = 0; k < 5; ++k) o[k] = (unsigned long long)ph[k];
    }
#endif

    // ---- epilogue (same contract as the generic kernel; tile row r -> pixel (y0 + r/TW, x0 + r%TW)) ----
    constexpr int CG = BN / 8;
    constexpr int PASSES = BM * CG / NT;
    static_assert(PASSES * NT == BM * CG, "epilogue split");
    size_t e_off[PASSES], e_off8[PASSES];
    uint4 e_res[PASSES];
#pragma unroll
    for (int ps = 0; ps < PASSES; ++ps) {
        const int idx = ps * NT + tid;
        const int r = idx / CG, cg = idx - r * CG;
        const int oy = y0 + r / TW, ox = x0 + r % TW;
        e_off[ps] = (((size_t)img * p.H + oy) * p.W + ox) * p.ycs + (size_t)g * p.ygs + n0 + cg * 8;
        e_off8[ps] = (((size_t)img * p.H + oy) * p.W + ox) * p.y8cs + (size_t)g * p.Cout + n0 + cg * 8;
    }
    if (p.res) {                                   // all residual loads in flight together, under the LDS staging
#pragma unroll
        for (int ps = 0; ps < PASSES; ++ps) e_res[ps] = *reinterpret_cast<const uint4*>(p.res + e_off[ps]);
    } else {
#pragma unroll
        for (int ps = 0; ps < PASSES; ++ps) e_res[ps] = make_uint4(0, 0, 0, 0);
    }
    float* Cs = reinterpret_cast<float*>(smem);
#pragma unroll
    for (int i = 0; i < MI; ++i) {
        const int ml = wm * WTM + i * 32 + l31;                   // D column = pixel
#pragma unroll
        for (int j = 0; j < NI; ++j)
#pragma unroll
            for (int eg = 0; eg < 4; ++eg)                        // D rows 8 eg + 4 lhi .. +4 = four consecutive channels
                *reinterpret_cast<f32x4_t*>(Cs + ml * CLD + wn * WTN + j * 32 + eg * 8 + lhi * 4) =
                    f32x4_t{acc[i][j][eg * 4], acc[i][j][eg * 4 + 1], acc[i][j][eg * 4 + 2], acc[i][j][eg * 4 + 3]};
    }
    __syncthreads();
#pragma unroll
    for (int ps = 0; ps < PASSES; ++ps) {
        const int idx = ps * NT + tid;
        const int r = idx / CG, cg = idx - r * CG;
        const f32x4_t v0 = *reinterpret_cast<const f32x4_t*>(Cs + r * CLD + cg * 8) * e_sc0 + e_sh0;
        const f32x4_t v1 = *reinterpret_cast<const f32x4_t*>(Cs + r * CLD + cg * 8 + 4) * e_sc1 + e_sh1;
        float v[8] = {v0[0], v0[1], v0[2], v0[3], v1[0], v1[1], v1[2], v1[3]};
        const uint32_t rw[4] = {e_res[ps].x, e_res[ps].y, e_res[ps].z, e_res[ps].w};
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            v[2 * e] += bf16_to_f32((uint16_t)(rw[e] & 0xFFFFu));
            v[2 * e + 1] += bf16_to_f32((uint16_t)(rw[e] >> 16));
        }
        if (p.relu) {
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] = fmaxf(v[e], 0.f);
        }
        store_out8(p, v, e_off[ps], e_off8[ps]);
    }
    dbg_stamp(p, 3);
#endif
}


// =====================================================================================================
// Stride-2 block front on halo patches: conv1 3x3 / s2 / p1 (+ the 1x1 / s2 downsample) of a BasicBlock.
//
// A stride-2 3x3 conv is a 2x2 stride-1 conv over the four POLYPHASE components of its input: input row 2oy-1+ky is
// phase py = (ky != 1) of pixel-block row oy-1 (ky = 0) or oy (ky = 1, 2), same along x.  So per 64-channel chunk the
// workgroup stages FOUR small patches instead of re-gathering its rows 9 times -- patch(py,px) = the (TH+1) x (TW+1) blocks
// around the tile, one pixel (2by+py, 2bx+px) each; the per-lane DMA source offsets make the stride-2 gather free -- and
// every tap reads its A fragments from the patch of its phase at a block shift (dby, dbx) in {0,1}^2:
//     phase (1,1): taps (0,0) (0,2) (2,0) (2,2)      phase (1,0): (0,1) (2,1)      phase (0,1): (1,0) (1,2)
//     phase (0,0): tap (1,1)  -- whose pixels (2oy, 2ox) are exactly the 1x1/s2 downsample's: a 10th K-step per chunk
//                                multiplies the same fragments with the downsample's weight tile into a second accumulator.
// Activation traffic L2->LDS: 4 x 162 pixels per chunk instead of 9 x 128 rows re-gathered, 9 (10) weight tiles as before;
// 8 waves (4 x 2, wave tile 32 px x 32 ch), patches double-buffered across phases (every phase has >= 2 steps, so the next
// patch is issued two steps ahead), 3-deep weight ring, counted vmcnt, one barrier per K-step; 66 KB of LDS -> two
// workgroups per CU.  K order: (chunk, taps in the phase order above) -- the generic kernel walks stride-2 convs in the
// same order (s2_tap_order), so the two are bit-identical.

template <int BN, int RS, bool F8>
__global__ __launch_bounds__(512) void conv3x3s2_patch_kernel(ConvArgs p) {
#if defined(__HIP_DEVICE_COMPILE__)
    W2C_FORCE_AGPR_FORM();
    constexpr int ES = OpT<F8>::ES, CK = OpT<F8>::CK;
    constexpr int TH = 8, TW = 16, BM = TH * TW, WM = 4, WN = 2, NW = 8, NT = 512;
    constexpr int D = RS - 1;                                          // weight tile t+D is issued at step t
    static_assert(RS == 2 || RS == 3, "ring depth");
    constexpr int WTN = BN / WN, NI = WTN / 32;
    constexpr int PWB = TW + 2, PHB = TH + 1, NP = PHB * PWB;          // blocks per patch row (17 used, even pitch) x rows
    constexpr int P_INSTR = (NP + 8 * NW - 1) / (8 * NW);
    constexpr int PATCH_BYTES = ((NP * 128 + 1023) / 1024) * 1024;
    constexpr int B_BYTES = BN * 128;
    constexpr int B_INSTR = BN / 8 / NW;
    constexpr int CLD = BN + 4;
    static_assert(NI >= 1 && B_INSTR >= 1 && NI * 32 * WN == BN, "shape");
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* const patch0 = smem;
    char* const bring = smem + 2 * PATCH_BYTES;

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int g = blockIdx.y;
    const int lrow = lane >> 3, lpos = lane & 7;
    const bool dual = p.w2 != nullptr;
    const int SPC = dual ? 10 : 9;                                     // K-steps per channel chunk

    dbg_stamp(p, 0);
    const int tiles_x = p.Wo / TW, tiles_y = p.Ho / TH;
    const int tile = xcd_remap(blockIdx.x, p.ntm * p.ntn);
    const int tsp = tile / p.ntn, tn = tile - tsp * p.ntn;
    const int txi = tsp % tiles_x;
    const int tyi = (tsp / tiles_x) % tiles_y;
    const int img = tsp / (tiles_x * tiles_y);
    const int oy0 = tyi * TH, ox0 = txi * TW, n0 = tn * BN;

    const char* xg = reinterpret_cast<const char*>(p.x) + (size_t)g * p.Cin * ES;
    const int Ktot = 9 * p.Cin;
    const char* wg = reinterpret_cast<const char*>(p.w) + (size_t)g * p.Cout * Ktot * ES;
    const char* wg2 = dual ? reinterpret_cast<const char*>(p.w2) + (size_t)g * p.Cout * p.Cin * ES : wg;
    const int nchunks = p.Cin / CK;
    const int KT = nchunks * SPC;
    const __amdgpu_buffer_rsrc_t rs_x = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<char*>(xg), 0, (int)((size_t)p.M * p.H * p.W * p.xcs * ES), 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_w = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<char*>(wg), 0, (int)((size_t)p.Cout * Ktot * ES), 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_w2 = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<char*>(wg2), 0, (int)((size_t)p.Cout * p.Cin * ES), 0x00020000);

    // patch DMA: LDS patch pixel q = (block row b_r, block col b_c) <- input pixel (2(oy0-1+b_r) + py, 2(ox0-1+b_c) + px);
    // the phase shift (py, px) is a scalar offset, the block base a per-lane one; blocks outside the image are out of range
    const bool p_last = (wave + NW * (P_INSTR - 1)) * 8 < NP;
    unsigned pa_off[P_INSTR];
#pragma unroll
    for (int j = 0; j < P_INSTR; ++j) {
        const int q = (wave + NW * j) * 8 + lrow;
        const int b_r = q / PWB, b_c = q - b_r * PWB;
        const int chunk = lpos ^ ((b_c >> 1) & 7);
        const int by = oy0 - 1 + b_r, bx = ox0 - 1 + b_c;
        const bool ok = (by >= 0) & (bx >= 0) & (b_c <= TW) & (2 * by + 1 < p.H) & (2 * bx + 1 < p.W);
        pa_off[j] = ok ? (unsigned)((((long)img * p.H + 2 * by) * p.W + 2 * bx) * p.xcs * ES + chunk * 16) : 0x80000000u;
    }
    auto issue_patch = [&](int cc, int ph, int buf) {          // ph: 0 -> (1,1), 1 -> (1,0), 2 -> (0,1), 3 -> (0,0)
        const int py = ph < 2 ? 1 : 0, px = (ph == 0 || ph == 2) ? 1 : 0;
        const int soff = (py * p.W + px) * p.xcs * ES + cc * 128;
        char* dst = patch0 + buf * PATCH_BYTES;
#pragma unroll
        for (int j = 0; j < P_INSTR; ++j) {
            if ((wave + NW * j) * 8 < NP) {
                if ((wave + NW * j) * 8 + lrow < NP)
                    __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_x, W2C_LPTR(dst + (wave + NW * j) * 1024), 16, pa_off[j], soff, 0, 0);
            }
        }
    };
    unsigned b_off[B_INSTR], b2_off[B_INSTR];
#pragma unroll
    for (int j = 0; j < B_INSTR; ++j) {
        const int n = (wave + NW * j) * 8 + lrow;
        b_off[j] = (unsigned)((size_t)(n0 + n) * Ktot * ES + (lpos ^ ((n >> 1) & 7)) * 16);
        b2_off[j] = (unsigned)((size_t)(n0 + n) * p.Cin * ES + (lpos ^ ((n >> 1) & 7)) * 16);
    }
    int st_i = 0, st_cc = 0;                                   // cursor of the next weight tile: step index in the chunk, chunk
    auto issue_b = [&](int slot) {
        char* dst = bring + slot * B_BYTES;
        if (st_i < 9) {
            int tky, tkx;
            s2_tap(st_i, tky, tkx);
            const int koff = (tky * 3 + tkx) * p.Cin * ES + st_cc * 128;
#pragma unroll
            for (int j = 0; j < B_INSTR; ++j)
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_w, W2C_LPTR(dst + (wave + NW * j) * 1024), 16, b_off[j], koff, 0, 0);
        } else {
#pragma unroll
            for (int j = 0; j < B_INSTR; ++j)
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_w2, W2C_LPTR(dst + (wave + NW * j) * 1024), 16, b2_off[j], st_cc * 128, 0, 0);
        }
        if (++st_i == SPC) { st_i = 0; ++st_cc; }
    };

    f32x16_t acc[NI], acc2[NI];
#pragma unroll
    for (int j = 0; j < NI; ++j)
#pragma unroll
        for (int e = 0; e < 16; ++e) { acc[j][e] = 0.f; acc2[j][e] = 0.f; }

    const int wm = wave / WN, wn = wave - wm * WN;
    const int l31 = lane & 31, lhi = lane >> 5;
    const int tp = wm * 32 + l31;                              // this lane's output pixel of the tile
    const int tr = tp / TW, tc = tp - tr * TW;
    const int bswz = (l31 >> 1) & 7;
    const float* ssp = p.scale + g * p.Cout + n0 + (tid % (BN / 8)) * 8;
    const float* shp = p.shift + g * p.Cout + n0 + (tid % (BN / 8)) * 8;
    const f32x4_t e_sc0 = *reinterpret_cast<const f32x4_t*>(ssp), e_sc1 = *reinterpret_cast<const f32x4_t*>(ssp + 4);
    const f32x4_t e_sh0 = *reinterpret_cast<const f32x4_t*>(shp), e_sh1 = *reinterpret_cast<const f32x4_t*>(shp + 4);

    // one K-step: fragments of (patch buffer, block shift) x weight slot -> MFMAs into `a`
    auto kstep = [&](const char* patch, const char* Bs, int dby, int dbx, f32x16_t (&a)[NI]) {
        const int col = tc + dbx;
        const char* arow = patch + ((tr + dby) * PWB + col) * 128;
        const int aswz = (col >> 1) & 7;
        if constexpr (F8) {
            i32x8_t fa[2], fb[2][NI];
#pragma unroll
            for (int j2 = 0; j2 < 2; ++j2) {
                const int c0 = j2 * 4 + lhi * 2;
                fa[j2] = cat_i32x8(*reinterpret_cast<const u32x4_t*>(arow + ((c0 ^ aswz) << 4)),
                                   *reinterpret_cast<const u32x4_t*>(arow + (((c0 + 1) ^ aswz) << 4)));
#pragma unroll
                for (int j = 0; j < NI; ++j) {
                    const char* r = Bs + (wn * WTN + j * 32 + l31) * 128;
                    fb[j2][j] = cat_i32x8(*reinterpret_cast<const u32x4_t*>(r + ((c0 ^ bswz) << 4)),
                                          *reinterpret_cast<const u32x4_t*>(r + (((c0 + 1) ^ bswz) << 4)));
                }
            }
#pragma unroll
            for (int j2 = 0; j2 < 2; ++j2)
#pragma unroll
                for (int j = 0; j < NI; ++j)
                    a[j] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(fb[j2][j], fa[j2], a[j], 0, 0, 0, 0x7F7F7F7F, 0, 0x7F7F7F7F);
#pragma unroll
            for (int j = 0; j < NI; ++j) asm volatile("" : "+v"(a[j]));
        } else {
            bf16x8_t fa[4], fb[4][NI];
#pragma unroll
            for (int kk = 0; kk < 4; ++kk) {
                fa[kk] = *reinterpret_cast<const bf16x8_t*>(arow + (((kk * 2 + lhi) ^ aswz) << 4));
#pragma unroll
                for (int j = 0; j < NI; ++j)
                    fb[kk][j] = *reinterpret_cast<const bf16x8_t*>(Bs + (wn * WTN + j * 32 + l31) * 128 + (((kk * 2 + lhi) ^ bswz) << 4));
            }
#pragma unroll
            for (int kk = 0; kk < 4; ++kk)
#pragma unroll
                for (int j = 0; j < NI; ++j) a[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fb[kk][j], fa[kk], a[j], 0, 0, 0);
        }
    };

    // ---- pipeline.  Per chunk, step i: 0-3 phase (1,1) | 4,5 phase (1,0) | 6,7 phase (0,1) | 8 phase (0,0) | 9 downsample.
    // Phase ph of a chunk uses patch buffer ph & 1; patch(ph+1) is issued at the first step of phase ph (right after that
    // step's weight tile), i.e. >= 2 steps ahead when the downsample step exists.  In-order vmcnt: at the top of step i a
    // wave may leave in flight the weight tile of step i+1 and a patch issued at step i-1 or i-2 (not needed before i+1).
    issue_patch(0, 0, 0);
    issue_b(0);
    if (D == 2) issue_b(1);
    int t = 0;
    auto step = [&](auto itag, int cc) {
        constexpr int i = decltype(itag)::value;
        constexpr bool start = (i == 0 || i == 4 || i == 6 || i == 8);
        // a patch issued at a phase-start step s (after that step's weight tile W(s+D)) is younger than W(t) iff s >= t-D, and
        // not needed before its own phase starts:
        constexpr bool patch_in_flight = (i == 1 || (D == 2 && i == 2) || i == 5 || i == 7 || i == 9);
        if (i == 0 && !dual && t > 0) {
            wait_vmcnt<0>();                                   // 9-step chunks: this phase's patch was issued one step ago, after W(t+D-1)
        } else if (t + D - 1 < KT) {
            if (patch_in_flight && !(i == 9 && cc + 1 >= nchunks)) {
                if (p_last) wait_vmcnt<(D - 1) * B_INSTR + P_INSTR>(); else wait_vmcnt<(D - 1) * B_INSTR + P_INSTR - 1>();
            } else {
                wait_vmcnt<(D - 1) * B_INSTR>();
            }
        } else {
            wait_vmcnt<0>();
        }
        pipeline_barrier();
        if (t == 0) dbg_stamp(p, 1);
        if (t + D < KT) issue_b((t + D) % RS);
        if (start) {                                          // next phase's patch: same buffer as the phase that just ended
            constexpr int ph = (i == 0) ? 0 : (i == 4) ? 1 : (i == 6) ? 2 : 3;
            if (ph < 3) issue_patch(cc, ph + 1, (ph + 1) & 1);
            else if (cc + 1 < nchunks) issue_patch(cc + 1, 0, 0);
        }
        constexpr int ph_i = (i < 4) ? 0 : (i < 6) ? 1 : (i < 8) ? 2 : 3;
        constexpr int ky = (i < 9) ? ((const int[9]){0, 0, 2, 2, 0, 2, 1, 1, 1})[i < 9 ? i : 8] : 1;
        constexpr int kx = (i < 9) ? ((const int[9]){0, 2, 0, 2, 1, 1, 0, 2, 1})[i < 9 ? i : 8] : 1;
        constexpr int dby = (ky == 0) ? 0 : 1, dbx = (kx == 0) ? 0 : 1;
        const char* patch = patch0 + (ph_i & 1) * PATCH_BYTES;
        const char* Bs = bring + (t % RS) * B_BYTES;
        if (i < 9) kstep(patch, Bs, dby, dbx, acc);
        else kstep(patch, Bs, 1, 1, acc2);
        ++t;
    };
    for (int cc = 0; cc < nchunks; ++cc) {
        step(std::integral_constant<int, 0>{}, cc); step(std::integral_constant<int, 1>{}, cc);
        step(std::integral_constant<int, 2>{}, cc); step(std::integral_constant<int, 3>{}, cc);
        step(std::integral_constant<int, 4>{}, cc); step(std::integral_constant<int, 5>{}, cc);
        step(std::integral_constant<int, 6>{}, cc); step(std::integral_constant<int, 7>{}, cc);
        step(std::integral_constant<int, 8>{}, cc);
        if (dual) step(std::integral_constant<int, 9>{}, cc);
    }
    __syncthreads();
    dbg_stamp(p, 2);

    // ---- epilogues: conv1 (scale/shift, ReLU, bf16 and/or fp8), then the downsample (scale2/shift2, bf16) ----
    constexpr int CG = BN / 8;
    constexpr int PASSES = BM * CG / NT;
    static_assert(PASSES * NT == BM * CG, "epilogue split");
    float* Cs = reinterpret_cast<float*>(smem);
    const int ml = wm * 32 + l31;
    size_t e_pix[PASSES];
#pragma unroll
    for (int ps = 0; ps < PASSES; ++ps) {
        const int r = (ps * NT + tid) / CG;
        e_pix[ps] = ((size_t)img * p.Ho + oy0 + r / TW) * p.Wo + ox0 + r % TW;
    }
#pragma unroll
    for (int j = 0; j < NI; ++j)
#pragma unroll
        for (int eg = 0; eg < 4; ++eg)
            *reinterpret_cast<f32x4_t*>(Cs + ml * CLD + wn * WTN + j * 32 + eg * 8 + lhi * 4) =
                f32x4_t{acc[j][eg * 4], acc[j][eg * 4 + 1], acc[j][eg * 4 + 2], acc[j][eg * 4 + 3]};
    __syncthreads();
#pragma unroll
    for (int ps = 0; ps < PASSES; ++ps) {
        const int idx = ps * NT + tid;
        const int r = idx / CG, cg = idx - r * CG;
        const f32x4_t v0 = *reinterpret_cast<const f32x4_t*>(Cs + r * CLD + cg * 8) * e_sc0 + e_sh0;
        const f32x4_t v1 = *reinterpret_cast<const f32x4_t*>(Cs + r * CLD + cg * 8 + 4) * e_sc1 + e_sh1;
        float v[8] = {v0[0], v0[1], v0[2], v0[3], v1[0], v1[1], v1[2], v1[3]};
        if (p.relu) {
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] = fmaxf(v[e], 0.f);
        }
        const size_t ch = (size_t)g * p.Cout + n0 + cg * 8;
        store_out8(p, v, e_pix[ps] * p.ycs + (size_t)g * p.ygs + n0 + cg * 8, e_pix[ps] * p.y8cs + ch);
    }
    if (dual) {
        __syncthreads();
#pragma unroll
        for (int j = 0; j < NI; ++j)
#pragma unroll
            for (int eg = 0; eg < 4; ++eg)
                *reinterpret_cast<f32x4_t*>(Cs + ml * CLD + wn * WTN + j * 32 + eg * 8 + lhi * 4) =
                    f32x4_t{acc2[j][eg * 4], acc2[j][eg * 4 + 1], acc2[j][eg * 4 + 2], acc2[j][eg * 4 + 3]};
        __syncthreads();
        const float* ssp2 = p.scale2 + g * p.Cout + n0 + (tid % (BN / 8)) * 8;
        const float* shp2 = p.shift2 + g * p.Cout + n0 + (tid % (BN / 8)) * 8;
        const f32x4_t sc0 = *reinterpret_cast<const f32x4_t*>(ssp2), sc1 = *reinterpret_cast<const f32x4_t*>(ssp2 + 4);
        const f32x4_t sh0 = *reinterpret_cast<const f32x4_t*>(shp2), sh1 = *reinterpret_cast<const f32x4_t*>(shp2 + 4);
#pragma unroll
        for (int ps = 0; ps < PASSES; ++ps) {
            const int idx = ps * NT + tid;
            const int r = idx / CG, cg = idx - r * CG;
            const f32x4_t v0 = *reinterpret_cast<const f32x4_t*>(Cs + r * CLD + cg * 8) * sc0 + sh0;
            const f32x4_t v1 = *reinterpret_cast<const f32x4_t*>(Cs + r * CLD + cg * 8 + 4) * sc1 + sh1;
            uint4 o;
            o.x = pack_bf16x2(v0[0], v0[1]); o.y = pack_bf16x2(v0[2], v0[3]);
            o.z = pack_bf16x2(v1[0], v1[1]); o.w = pack_bf16x2(v1[2], v1[3]);
            *reinterpret_cast<uint4*>(p.y2 + e_pix[ps] * p.y2cs + (size_t)g * p.Cout + n0 + cg * 8) = o;
        }
    }
    dbg_stamp(p, 3);
#endif
}

template <int BN, int RS, bool F8>
int launch_s2patch(ConvArgs& a, int groups, hipStream_t s) {
    constexpr int CK = OpT<F8>::CK;
    if (a.ks != 3 || a.stride != 2 || a.Cin % CK != 0 || a.Cout % BN != 0 || a.Ho % 8 != 0 || a.Wo % 16 != 0 || (a.H & 1) || (a.W & 1) ||
        a.res)                                                 // (a block's conv1 has no residual; the kernel has no path for one)
        return W2C_E_ARG;
    a.ntm = a.M * (a.Ho / 8) * (a.Wo / 16);
    a.ntn = a.Cout / BN;
    constexpr int patch = ((9 * 18 * 128 + 1023) / 1024) * 1024;
    constexpr int ring = 2 * patch + RS * BN * 128;
    constexpr int epi = 128 * (BN + 4) * 4;
    constexpr int lds = ring > epi ? ring : epi;
    static_assert(lds <= 80 * 1024, "two workgroups per CU");
    static std::atomic<unsigned long long> attr_mask{0};
    int dev = 0;
    (void)hipGetDevice(&dev);
    if (!((attr_mask.load(std::memory_order_acquire) >> (dev & 63)) & 1ull)) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&conv3x3s2_patch_kernel<BN, RS, F8>),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, lds);
        attr_mask.fetch_or(1ull << (dev & 63), std::memory_order_release);
    }
    hipLaunchKernelGGL((conv3x3s2_patch_kernel<BN, RS, F8>), dim3(a.ntm * a.ntn, groups), dim3(512), lds, s, a);
    return w2c_launch_status();
}

// =====================================================================================================
// Layer1 kernel: 3x3 stride-1, Cin = Cout = 64 per group, weights stationary in REGISTERS.
//
// K = 576 is only 9 K-steps of the ring kernels above: a 128-pixel tile is ~1 us of MFMA work per wave under
// 9 barriers, a cold prologue and a 3 us epilogue -- layer1 ran at 25 % of the MFMA peak and 2x its HBM time.
// Here the whole 64 x 576 weight matrix of the group (73 KB) lives in the wave's registers for the kernel's
// lifetime (72 MFMA A-fragments = 288 of the 512 VGPR+AGPR of a 1-wave-per-SIMD kernel; MFMA A/B operands may be
// AGPRs on gfx950), and each WAVE is an independent persistent worker:
//   * it owns a contiguous run of 4 x 16-pixel tiles (XCD-contiguous, so the halo rows of neighbouring waves meet
//     in one L2) and two private 6 x 18-pixel patch buffers in LDS;
//   * the patch of tile t+1 and the residual tile of tile t arrive by LDS-DMA (buffer_load ... lds; out-of-image
//     halo pixels are out-of-range offsets -> zeros) while tile t's 144 MFMAs run; B fragments (pixels) are one
//     ds_read_b128 each, 0.5 per MFMA;
//   * NO workgroup barrier anywhere: every LDS byte a wave reads was written by that wave (DMA + s_waitcnt vmcnt,
//     or its own in-order ds_writes);
//   * epilogue: the dead patch buffer becomes the f32 staging of 32 pixels x 64 channels (row pitch 272 B:
//     conflict-free ds_write_b128), read back as 8-channel groups, + scale/shift + residual (from LDS) + ReLU,
//     16-byte coalesced bf16 stores.
// vmcnt bookkeeping per tile: [patch(t+1) x14, residual(t) x8] are issued at the top of tile t and waited for
// (vmcnt(0)) before epilogue(t), ~4.6k MFMA cycles later; that wait also retires the stores of tile t-1.
template <bool HAS_RES>
__global__ __launch_bounds__(256) void conv3x3_c64_regw_kernel(ConvArgs p) {
#if defined(__HIP_DEVICE_COMPILE__)
    constexpr int PW = 18;                         // patch width (16 + 2 halo); 6 rows
    constexpr int PATCH_BYTES = 14 * 1024;         // 14 DMA instructions x 8 pixels x 128 B (108 pixels used)
    constexpr int RES_BYTES = 8 * 1024;            // 64 pixels x 128 B
    constexpr int WAVE_LDS = 2 * PATCH_BYTES + RES_BYTES;
    constexpr int SPITCH = 272;                    // staging row pitch, bytes
    constexpr int WPITCH = 1152 + 16;              // prologue weight rows in LDS: 16-B pad => conflict-free fragment reads
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l31 = lane & 31, lhi = lane >> 5;
    const int g = blockIdx.y;
    char* const wl = smem + wave * WAVE_LDS;
    char* const resbuf = wl + 2 * PATCH_BYTES;

    // ---- weights -> registers: A operand, row = output channel, k = (tap, cin).  The register CLASS is pinned by the
    // inline-asm MFMAs below: channel tile 0 (144 regs) + the 64 accumulators live in AGPRs, channel tile 1 in VGPRs.
    // Left to the compiler, all 288 land in VGPR-class values that are spilled to / re-read from AGPRs around every use
    // and the B-fragment reads lose their double buffer (measured 102 us vs 72 us for the ring kernel).
    // A fragment is 16 B out of a 1152-B weight row: gathering it straight from global memory costs 72 uncoalesced
    // loads per lane, so the group's 73 KB go through LDS once (coalesced in, fragment-shaped out). ----
    const unsigned long long wall0 = p.dbg ? wall_clock64() : 0;
    span_stamp(p, false);
    u32x4_t wa[9][4], wv[9][4];
    {
        const uint16_t* wg = p.w + (size_t)g * 64 * 576;
        for (int c = tid; c < 64 * 72; c += 256) {
            const int row = c / 72, col = c - row * 72;
            *reinterpret_cast<uint4*>(smem + row * WPITCH + col * 16) = *reinterpret_cast<const uint4*>(wg + row * 576 + col * 8);
        }
        if (tid < 64) {                            // BN scale | shift of the group: 512 B after the waves' regions
            reinterpret_cast<float*>(smem + 4 * WAVE_LDS)[tid] = p.scale[g * 64 + tid];
            reinterpret_cast<float*>(smem + 4 * WAVE_LDS)[64 + tid] = p.shift[g * 64 + tid];
        }
        __syncthreads();
#pragma unroll
        for (int tap = 0; tap < 9; ++tap)
#pragma unroll
            for (int kc = 0; kc < 4; ++kc) {
                wa[tap][kc] = *reinterpret_cast<const u32x4_t*>(smem + l31 * WPITCH + tap * 128 + kc * 32 + lhi * 16);
                wv[tap][kc] = *reinterpret_cast<const u32x4_t*>(smem + (32 + l31) * WPITCH + tap * 128 + kc * 32 + lhi * 16);
            }
        __syncthreads();                           // the only workgroup barriers of the kernel: LDS is reused below
    }

    // ---- tiles of this wave: a contiguous run, XCD-contiguous across the grid ----
    const int ntx = p.W >> 4, nty = p.H >> 2, tpi = ntx * nty;
    const int T = p.M * tpi;
    const int nwg = gridDim.x;
    const int b = blockIdx.x;
    const int logical = (nwg % 8 == 0) ? (b & 7) * (nwg >> 3) + (b >> 3) : b;
    const int wid = logical * 4 + wave, nw = nwg * 4;
    const int t_begin = __builtin_amdgcn_readfirstlane((int)(((long)wid * T) / nw));      // wave-uniform, and the
    const int t_end = __builtin_amdgcn_readfirstlane((int)(((long)(wid + 1) * T) / nw));    // compiler should know it
    if (t_begin >= t_end) return;

    // ---- DMA constants.  Patch instruction j moves pixels q = 8j + lane/8 (q = row*18 + col), lane%8 = LDS chunk
    // position; the bank swizzle (chunk c of a pixel in patch column x sits at c ^ ((x>>1)&7)) is applied to the SOURCE chunk. ----
    const size_t x_bytes = (size_t)p.M * p.H * p.W * p.xcs * 2;
    const size_t y_bytes = (size_t)p.M * p.H * p.W * p.ycs * (p.y_f32 ? 4 : 2);
    const size_t r_bytes = (size_t)p.M * p.H * p.W * p.ycs * 2;
    const __amdgpu_buffer_rsrc_t rs_x = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<uint16_t*>(p.x + (size_t)g * 64), 0, (int)(x_bytes - (size_t)g * 128), 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_r = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<uint16_t*>((HAS_RES ? p.res : p.x) + (size_t)g * 64), 0, (int)((HAS_RES ? r_bytes : x_bytes) - (size_t)g * 128),
        0x00020000);
    const __amdgpu_buffer_rsrc_t rs_y = __builtin_amdgcn_make_buffer_rsrc(p.y, 0, (int)y_bytes, 0x00020000);
    int off_rel[14];
    unsigned m_top = 0, m_bot = 0, m_left = 0, m_right = 0, m_inval = 0;
#pragma unroll
    for (int j = 0; j < 14; ++j) {
        const int q = 8 * j + (lane >> 3);
        const int dy = (q * 3641) >> 16, dx = q - dy * PW;          // q / 18, q % 18 for q < 128
        const int chunk = (lane & 7) ^ ((dx >> 1) & 7);             // swizzle keyed on the patch COLUMN (conflict-free, see above)
        off_rel[j] = ((dy * p.W + dx) * p.xcs + chunk * 8) * 2;
        m_top |= (dy == 0 ? 1u : 0u) << j;
        m_bot |= (dy == 5 ? 1u : 0u) << j;
        m_left |= (dx == 0 ? 1u : 0u) << j;
        m_right |= (dx == 17 ? 1u : 0u) << j;
        m_inval |= (q >= 6 * PW ? 1u : 0u) << j;
    }
    const unsigned r_lane = (unsigned)(((lane >> 3) * p.ycs + (lane & 7) * 8) * 2);        // residual DMA: pixel lane/8, chunk lane%8
    const int cg = lane & 7;                                                               // read-out: 8-channel group
    const unsigned y_lane = (unsigned)(((lane >> 3) * p.ycs + g * 64 + cg * 8) * (p.y_f32 ? 4 : 2));

    auto tile_coords = [&](int t, int& img, int& y0, int& x0) {
        img = t / tpi;
        const int r = t - img * tpi;
        const int tx = r / nty;                    // column-major: a wave's consecutive tiles are vertically adjacent, so
        x0 = tx * 16;                              // 2 of the 6 patch rows of the next tile were just read by this wave
        y0 = (r - tx * nty) * 4;                   // (measured: layer1 L2-miss read traffic 1.47x -> see profiles/)
    };
    // patch DMA of one tile = 14 instructions; `pbase` / `pbad` are its wave-uniform base offset and halo mask
    int pbase = 0;
    unsigned pbad = 0;
    auto patch_setup = [&](int t, bool live) {
        int img, y0, x0;
        tile_coords(t, img, y0, x0);
        pbase = (((img * p.H + y0 - 1) * p.W) + x0 - 1) * p.xcs * 2;          // bytes; negative only where the halo lanes are off
        pbad = !live ? 0xFFFFFFFFu
                     : (m_inval | (y0 == 0 ? m_top : 0u) | (y0 + 4 == p.H ? m_bot : 0u) | (x0 == 0 ? m_left : 0u) |
                        (x0 + 16 == p.W ? m_right : 0u));
    };
    auto patch_piece = [&](int j, char* dst) {
        const unsigned vo = ((pbad >> j) & 1u) ? 0x80000000u : (unsigned)(pbase + off_rel[j]);
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_x, W2C_LPTR(dst + j * 1024), 16, vo, 0, 0, 0);
    };
    int rbase = 0;
    auto residual_piece = [&](int j) {       // instruction j: pixels 8j..8j+7 of the 4 x 16 tile (row j/2, cols 8(j&1)..)
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_r, W2C_LPTR(resbuf + j * 1024), 16, r_lane,
                                                 rbase + ((j >> 1) * p.W + 8 * (j & 1)) * p.ycs * 2, 0, 0);
    };

    // B-fragment geometry: MFMA pixel tile pt = rows 2pt, 2pt+1 of the 4 x 16 tile; lane's pixel = (l31>>4, l31&15)
    int qb[2];
#pragma unroll
    for (int pt = 0; pt < 2; ++pt) qb[pt] = (pt * 2 + (l31 >> 4)) * PW + (l31 & 15);
    const char* const ssb = smem + 4 * WAVE_LDS + cg * 32;     // this lane's 8 channels' scale (+256: shift), re-read per tile

    int cur = 0;
    patch_setup(t_begin, true);
#pragma unroll
    for (int j = 0; j < 14; ++j) patch_piece(j, wl);
    unsigned long long ph[4] = {0, 0, 0, 0};       // debug (p.dbg): cycles at tile top | MFMA loop | vmcnt wait | epilogue
    const unsigned long long wall1 = p.dbg ? wall_clock64() : 0;
    long long tp = p.dbg ? clock64() : 0;
    auto stamp = [&](int i) {
        if (p.dbg) { const long long n = clock64(); ph[i] += (unsigned long long)(n - tp); tp = n; }
    };
    int img, y0, x0;
    tile_coords(t_begin, img, y0, x0);
    for (int t = t_begin; t < t_end; ++t) {
        char* const pc = wl + cur * PATCH_BYTES;
        char* const pn = wl + (cur ^ 1) * PATCH_BYTES;
        // tile t -> (img, y0, x0) incrementally (the divisions of tile_coords cost ~500 cycles per tile on one wave)
        if (t != t_begin) {
            y0 += 4;
            if (y0 == p.H) { y0 = 0; x0 += 16; if (x0 == p.W) { x0 = 0; ++img; } }
        }
        rbase = ((img * p.H + y0) * p.W + x0) * p.ycs * 2;
        {   // patch(t+1): base offset + halo mask; on the last tile every lane is off (zeros into the idle buffer)
            int xn = x0, yn = y0 + 4, in = img;
            if (yn == p.H) { yn = 0; xn += 16; if (xn == p.W) { xn = 0; ++in; } }
            const bool live = t + 1 < t_end;
            pbase = (((in * p.H + yn - 1) * p.W) + xn - 1) * p.xcs * 2;
            pbad = !live ? 0xFFFFFFFFu
                         : (m_inval | (yn == 0 ? m_top : 0u) | (yn + 4 == p.H ? m_bot : 0u) | (xn == 0 ? m_left : 0u) |
                            (xn + 16 == p.W ? m_right : 0u));
        }
        // patch(t) must have landed.  First tile: everything issued so far.  Later tiles: its 14 pieces were issued during
        // tile t-1 and are older than that tile's output stores (8, or 16 for f32 output), which may stay in flight --
        // vmcnt retires in issue order on gfx9 (the compiler's own waitcnt insertion relies on the same).
        if (t == t_begin) wait_vmcnt<0>();
        else if (p.y_f32) wait_vmcnt<16>();
        else wait_vmcnt<8>();
        asm volatile("" ::: "memory");
        stamp(0);

        // 36 K-steps (tap, 16-channel chunk), fully unrolled; the B fragments of step s+1 are read before the MFMAs of
        // step s are issued, and the 8 + 14 LDS-DMA instructions of residual(t) / patch(t+1) are spread one per K-step
        // through the MFMA stream (issued back to back at the tile top they cost ~95 cycles each with the MFMA pipe idle).
        f32x16_t acc[2][2];
        // fragment address of (tap, kc) for pixel q = qb + ky*18 + kx:  q*128 + ((2kc | lhi) ^ ((q>>1)&7))*16
        //                                                             = fb + ((kc << 5) ^ fx),  fb, fx per (tap, pt).
        // qv is laundered through an empty asm once per tile so the 72 addresses are NOT hoisted out of the tile loop
        // (they would cost 72 registers; recomputed they are ~2 VALU per read in the MFMA shadow).
        int qv0 = qb[0], qv1 = qb[1];
        asm volatile("" : "+v"(qv0), "+v"(qv1));
        int fb[2], fx[2];
        auto tap_setup = [&](int tap) {
            const int ky = tap / 3, kx = tap - 3 * ky;
            const int q0 = qv0 + ky * PW + kx, q1 = qv1 + ky * PW + kx;
            const int cs = ((((l31 & 15) + kx) >> 1) ^ lhi) & 7;      // column-keyed swizzle, both pixel tiles share the column
            fb[0] = q0 * 128; fx[0] = cs << 4;
            fb[1] = q1 * 128; fx[1] = cs << 4;
        };
        auto frag = [&](int kc, int pt) { return *reinterpret_cast<const u32x4_t*>(pc + fb[pt] + ((kc << 5) ^ fx[pt])); };
        u32x4_t bx[2][2];
        tap_setup(0);
        bx[0][0] = frag(0, 0);
        bx[0][1] = frag(0, 1);
#pragma unroll
        for (int step = 0; step < 36; ++step) {
            const int tap = step >> 2, kc = step & 3, cb = step & 1;
            if (step + 1 < 36) {
                if (((step + 1) & 3) == 0) tap_setup((step + 1) >> 2);
                bx[cb ^ 1][0] = frag((step + 1) & 3, 0);
                bx[cb ^ 1][1] = frag((step + 1) & 3, 1);
            }
            if (HAS_RES && step >= 1 && step <= 8) residual_piece(step - 1);
            if (step >= 9 && step <= 22) patch_piece(step - 9, pn);
            if (step == 0) {
                asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, 0" : "=a"(acc[0][0]) : "a"(wa[tap][kc]), "v"(bx[cb][0]));
                asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, 0" : "=a"(acc[0][1]) : "v"(wv[tap][kc]), "v"(bx[cb][0]));
                asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, 0" : "=a"(acc[1][0]) : "a"(wa[tap][kc]), "v"(bx[cb][1]));
                asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, 0" : "=a"(acc[1][1]) : "v"(wv[tap][kc]), "v"(bx[cb][1]));
            } else {
                asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+a"(acc[0][0]) : "a"(wa[tap][kc]), "v"(bx[cb][0]));
                asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+a"(acc[0][1]) : "v"(wv[tap][kc]), "v"(bx[cb][0]));
                asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+a"(acc[1][0]) : "a"(wa[tap][kc]), "v"(bx[cb][1]));
                asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+a"(acc[1][1]) : "v"(wv[tap][kc]), "v"(bx[cb][1]));
            }
        }
        // the MFMAs are opaque to the compiler's hazard recogniser: cover the XDL-write -> VALU/DS-read wait states by hand
        asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");
        stamp(1);

        // ---- epilogue ----
        char* const stage = pc;                    // the patch of tile t is dead: f32 staging, 32 pixels x 64 channels
        const f32x4_t e_sc0 = *reinterpret_cast<const f32x4_t*>(ssb), e_sc1 = *reinterpret_cast<const f32x4_t*>(ssb + 16);
        const f32x4_t e_sh0 = *reinterpret_cast<const f32x4_t*>(ssb + 256), e_sh1 = *reinterpret_cast<const f32x4_t*>(ssb + 272);
#pragma unroll
        for (int pt = 0; pt < 2; ++pt) {
            // D layout: lane = pixel l31, acc element e = channel (e&3) + 8(e>>2) + 4 lhi of the 32-channel tile ct
#pragma unroll
            for (int ct = 0; ct < 2; ++ct)
#pragma unroll
                for (int eg = 0; eg < 4; ++eg)
                    *reinterpret_cast<f32x4_t*>(stage + l31 * SPITCH + (ct * 32 + eg * 8 + lhi * 4) * 4) =
                        f32x4_t{acc[pt][ct][eg * 4], acc[pt][ct][eg * 4 + 1], acc[pt][ct][eg * 4 + 2], acc[pt][ct][eg * 4 + 3]};
            if (pt == 0) {     // residual(t) (issued at K-steps 1-8) must have landed; the 14 younger pieces of patch(t+1) may
                asm volatile("" ::: "memory");      // still be in flight (HBM latency > the 14 K-steps since their issue)
                if (HAS_RES) wait_vmcnt<14>();
                asm volatile("" ::: "memory");
                stamp(2);
            }
            f32x4_t v0[4], v1[4];
            uint4 rr[4];
#pragma unroll
            for (int it = 0; it < 4; ++it) {       // read-out item: pixel it*8 + lane/8 of the MFMA tile, channels cg*8..+8
                const char* sp = stage + (it * 8 + (lane >> 3)) * SPITCH + cg * 32;
                v0[it] = *reinterpret_cast<const f32x4_t*>(sp);
                v1[it] = *reinterpret_cast<const f32x4_t*>(sp + 16);
                if constexpr (HAS_RES)             // tile pixel = (pt*2 + it/2, (it&1)*8 + lane/8)
                    rr[it] = *reinterpret_cast<const uint4*>(resbuf + ((pt * 2 + (it >> 1)) * 16 + (it & 1) * 8 + (lane >> 3)) * 128 + cg * 16);
            }
#pragma unroll
            for (int it = 0; it < 4; ++it) {
                f32x4_t a0 = v0[it] * e_sc0 + e_sh0;
                f32x4_t a1 = v1[it] * e_sc1 + e_sh1;
                if constexpr (HAS_RES) {
                    const uint32_t rw[4] = {rr[it].x, rr[it].y, rr[it].z, rr[it].w};
                    a0 += f32x4_t{__uint_as_float(rw[0] << 16), __uint_as_float(rw[0] & 0xFFFF0000u),
                                  __uint_as_float(rw[1] << 16), __uint_as_float(rw[1] & 0xFFFF0000u)};
                    a1 += f32x4_t{__uint_as_float(rw[2] << 16), __uint_as_float(rw[2] & 0xFFFF0000u),
                                  __uint_as_float(rw[3] << 16), __uint_as_float(rw[3] & 0xFFFF0000u)};
                }
                const int pix = ((pt * 2 + (it >> 1)) * p.W + (it & 1) * 8);                     // tile-relative pixel, uniform
                const int tile_pix = (img * p.H + y0) * p.W + x0;
                if (p.y_f32) {
                    if (p.relu) {
#pragma unroll
                        for (int e = 0; e < 4; ++e) { a0[e] = fmaxf(a0[e], 0.f); a1[e] = fmaxf(a1[e], 0.f); }
                    }
                    const int so = (tile_pix + pix) * p.ycs * 4;
                    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4_t, a0), rs_y, y_lane, so, 0);
                    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4_t, a1), rs_y, y_lane + 16, so, 0);
                } else {
                    // ReLU after the bf16 rounding, on the packed pairs: rounding keeps the sign, and a signed 16-bit
                    // max with 0 clears exactly the negative bf16 (incl. -0) -- same bits as fmaxf before the rounding
                    uint32_t ow[4] = {pack_bf16x2(a0[0], a0[1]), pack_bf16x2(a0[2], a0[3]), pack_bf16x2(a1[0], a1[1]),
                                      pack_bf16x2(a1[2], a1[3])};
                    if (p.relu) {
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            const s16x2_t h = __builtin_bit_cast(s16x2_t, ow[e]);
                            ow[e] = __builtin_bit_cast(uint32_t, __builtin_elementwise_max(h, s16x2_t{0, 0}));
                        }
                    }
                    const u32x4_t o = {ow[0], ow[1], ow[2], ow[3]};
                    __builtin_amdgcn_raw_buffer_store_b128(o, rs_y, y_lane, (tile_pix + pix) * p.ycs * 2, 0);
                }
            }
        }
        asm volatile("" ::: "memory");
        stamp(3);
        cur ^= 1;
    }
    if (p.dbg && lane == 0 && wave == 0) {         // 8 x u64 per workgroup: 4 phase sums (shader cycles), 3 wall stamps (100 MHz)
        unsigned long long* d = p.dbg + ((size_t)blockIdx.y * gridDim.x + blockIdx.x) * 8;
        for (int i = 0; i < 4; ++i) d[i] = ph[i];
        d[4] = wall0; d[5] = wall1; d[6] = wall_clock64(); d[7] = 1;
    }
    span_stamp(p, true);
#endif
}

template <bool HAS_RES>
__global__ __launch_bounds__(256) void conv3x3_c64_regw2_kernel(ConvArgs p) {
#if defined(__HIP_DEVICE_COMPILE__)
    constexpr int PW = 18;                         // patch width (16 + 2 halo); 6 rows
    constexpr int PATCH_BYTES = 14 * 1024;         // 14 DMA instructions x 8 pixels x 128 B (108 pixels used)
    constexpr int RES_BYTES = 8 * 1024;            // 64 pixels x 128 B
    constexpr int WAVE_LDS = 2 * PATCH_BYTES + RES_BYTES;
    constexpr int SPITCH = 272;                    // staging row pitch, bytes
    constexpr int WPITCH = 1152 + 16;              // prologue weight rows in LDS: 16-B pad => conflict-free fragment reads
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l31 = lane & 31, lhi = lane >> 5;
    const int g = blockIdx.y;
    char* const wl = smem + wave * WAVE_LDS;
    char* const resbuf = wl + 2 * PATCH_BYTES;

    // ---- weights -> registers: A operand, row = output channel, k = (tap, cin).  The register CLASS is pinned by the
    // inline-asm MFMAs below: channel tile 0 (144 regs) + the 64 accumulators live in AGPRs, channel tile 1 in VGPRs.
    // Left to the compiler, all 288 land in VGPR-class values that are spilled to / re-read from AGPRs around every use
    // and the B-fragment reads lose their double buffer (measured 102 us vs 72 us for the ring kernel).
    // A fragment is 16 B out of a 1152-B weight row: gathering it straight from global memory costs 72 uncoalesced
    // loads per lane, so the group's 73 KB go through LDS once (coalesced in, fragment-shaped out). ----
    const unsigned long long wall0 = p.dbg ? wall_clock64() : 0;
    span_stamp(p, false);
    u32x4_t wa[9][4], wv[9][4];
    {
        const uint16_t* wg = p.w + (size_t)g * 64 * 576;
        for (int c = tid; c < 64 * 72; c += 256) {
            const int row = c / 72, col = c - row * 72;
            *reinterpret_cast<uint4*>(smem + row * WPITCH + col * 16) = *reinterpret_cast<const uint4*>(wg + row * 576 + col * 8);
        }
        if (tid < 64) {
            // BN scale / shift of the group in the accumulators' layout: entry (ct, j, lhi) = {scale[4], shift[4]} of channels
            // ct*32 + 8j + 4 lhi .. +3 -- what one lane half holds in accumulator elements 4j .. 4j+3 of channel tile ct
            const int ct = tid >> 5, j = (tid >> 3) & 3, lh = (tid >> 2) & 1, k = tid & 3;
            float* const e = reinterpret_cast<float*>(smem + 4 * WAVE_LDS) + ((ct * 4 + j) * 2 + lh) * 8;
            e[k] = p.scale[g * 64 + tid];
            e[4 + k] = p.shift[g * 64 + tid];
        }
        __syncthreads();
#pragma unroll
        for (int tap = 0; tap < 9; ++tap)
#pragma unroll
            for (int kc = 0; kc < 4; ++kc) {
                wa[tap][kc] = *reinterpret_cast<const u32x4_t*>(smem + l31 * WPITCH + tap * 128 + kc * 32 + lhi * 16);
                wv[tap][kc] = *reinterpret_cast<const u32x4_t*>(smem + (32 + l31) * WPITCH + tap * 128 + kc * 32 + lhi * 16);
            }
        __syncthreads();                           // the only workgroup barriers of the kernel: LDS is reused below
    }

    // ---- tiles of this wave: a contiguous run, XCD-contiguous across the grid ----
    const int ntx = p.W >> 4, nty = p.H >> 2, tpi = ntx * nty;
    const int T = p.M * tpi;
    const int nwg = gridDim.x;
    const int b = blockIdx.x;
    const int logical = (nwg % 8 == 0) ? (b & 7) * (nwg >> 3) + (b >> 3) : b;
    const int wid = logical * 4 + wave, nw = nwg * 4;
    const int t_begin = __builtin_amdgcn_readfirstlane((int)(((long)wid * T) / nw));      // wave-uniform, and the
    const int t_end = __builtin_amdgcn_readfirstlane((int)(((long)(wid + 1) * T) / nw));    // compiler should know it
    if (t_begin >= t_end) return;

    // ---- DMA constants.  Patch instruction j moves pixels q = 8j + lane/8 (q = row*18 + col), lane%8 = LDS chunk
    // position; the bank swizzle (chunk c of a pixel in patch column x sits at c ^ ((x>>1)&7)) is applied to the SOURCE chunk. ----
    const size_t x_bytes = (size_t)p.M * p.H * p.W * p.xcs * 2;
    const size_t y_bytes = (size_t)p.M * p.H * p.W * p.ycs * (p.y_f32 ? 4 : 2);
    const size_t r_bytes = (size_t)p.M * p.H * p.W * p.ycs * 2;
    const __amdgpu_buffer_rsrc_t rs_x = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<uint16_t*>(p.x + (size_t)g * 64), 0, (int)(x_bytes - (size_t)g * 128), 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_r = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<uint16_t*>((HAS_RES ? p.res : p.x) + (size_t)g * 64), 0, (int)((HAS_RES ? r_bytes : x_bytes) - (size_t)g * 128),
        0x00020000);
    const __amdgpu_buffer_rsrc_t rs_y = __builtin_amdgcn_make_buffer_rsrc(p.y, 0, (int)y_bytes, 0x00020000);
    int off_rel[14];
    unsigned m_top = 0, m_bot = 0, m_left = 0, m_right = 0, m_inval = 0;
#pragma unroll
    for (int j = 0; j < 14; ++j) {
        const int q = 8 * j + (lane >> 3);
        const int dy = (q * 3641) >> 16, dx = q - dy * PW;          // q / 18, q % 18 for q < 128
        const int chunk = (lane & 7) ^ ((dx >> 1) & 7);             // swizzle keyed on the patch COLUMN (conflict-free, see above)
        off_rel[j] = ((dy * p.W + dx) * p.xcs + chunk * 8) * 2;
        m_top |= (dy == 0 ? 1u : 0u) << j;
        m_bot |= (dy == 5 ? 1u : 0u) << j;
        m_left |= (dx == 0 ? 1u : 0u) << j;
        m_right |= (dx == 17 ? 1u : 0u) << j;
        m_inval |= (q >= 6 * PW ? 1u : 0u) << j;
    }
    // residual DMA: instruction j moves pixels 8j .. 8j+7 (pixel lane/8), 16-byte chunk c of pixel q lands at position c ^ f(q),
    // f(q) = (q & 7) ^ ((q >> 3) & 1) = (lane/8) ^ (j & 1) -- keyed so that the epilogue's 8-byte reads (one pixel per lane, same channel
    // quad: rows 128 B apart) spread over the banks instead of hitting two of them
    const unsigned r_lane0 = (unsigned)(((lane >> 3) * p.ycs + ((lane & 7) ^ (lane >> 3)) * 8) * 2);
    const unsigned r_lane1 = (unsigned)(((lane >> 3) * p.ycs + ((lane & 7) ^ (lane >> 3) ^ 1) * 8) * 2);
    // epilogue, register-direct: lane (l31, lhi) holds pixel (2pt + l31/16, l31%16) of the tile; after the half-wave swap it stores the
    // 8 channels ct*32 + 8(2jp + lhi) .. +7 of that pixel
    const int e_px = (l31 >> 4) * p.W + (l31 & 15);                                        // + 2 pt W
    const unsigned y_lane = (unsigned)((e_px * p.ycs + g * 64 + lhi * 8) * 2);
    const int r_px = (l31 >> 4) * 16 + (l31 & 15);                                         // + 32 pt: pixel index in the residual tile
    const int r_f = (r_px & 7) ^ ((r_px >> 3) & 1);                                        // f(q) of the DMA above (+ 32 pt does not change it)

    auto tile_coords = [&](int t, int& img, int& y0, int& x0) {
        img = t / tpi;
        const int r = t - img * tpi;
        const int tx = r / nty;                    // column-major: a wave's consecutive tiles are vertically adjacent, so
        x0 = tx * 16;                              // 2 of the 6 patch rows of the next tile were just read by this wave
        y0 = (r - tx * nty) * 4;                   // (measured: layer1 L2-miss read traffic 1.47x -> see profiles/)
    };
    // patch DMA of one tile = 14 instructions; `pbase` / `pbad` are its wave-uniform base offset and halo mask
    int pbase = 0;
    unsigned pbad = 0;
    auto patch_setup = [&](int t, bool live) {
        int img, y0, x0;
        tile_coords(t, img, y0, x0);
        pbase = (((img * p.H + y0 - 1) * p.W) + x0 - 1) * p.xcs * 2;          // bytes; negative only where the halo lanes are off
        pbad = !live ? 0xFFFFFFFFu
                     : (m_inval | (y0 == 0 ? m_top : 0u) | (y0 + 4 == p.H ? m_bot : 0u) | (x0 == 0 ? m_left : 0u) |
                        (x0 + 16 == p.W ? m_right : 0u));
    };
    auto patch_piece = [&](int j, char* dst) {
        const unsigned vo = ((pbad >> j) & 1u) ? 0x80000000u : (unsigned)(pbase + off_rel[j]);
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_x, W2C_LPTR(dst + j * 1024), 16, vo, 0, 0, 0);
    };
    int rbase = 0;
    auto residual_piece = [&](int j) {       // instruction j: pixels 8j..8j+7 of the 4 x 16 tile (row j/2, cols 8(j&1)..)
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_r, W2C_LPTR(resbuf + j * 1024), 16, (j & 1) ? r_lane1 : r_lane0,
                                                 rbase + ((j >> 1) * p.W + 8 * (j & 1)) * p.ycs * 2, 0, 0);
    };

    // B-fragment geometry: MFMA pixel tile pt = rows 2pt, 2pt+1 of the 4 x 16 tile; lane's pixel = (l31>>4, l31&15)
    int qb[2];
#pragma unroll
    for (int pt = 0; pt < 2; ++pt) qb[pt] = (pt * 2 + (l31 >> 4)) * PW + (l31 & 15);
    const char* const ssd = smem + 4 * WAVE_LDS + lhi * 32;    // + (ct*4 + j) * 64: {scale[4], shift[4]} of this lane half's quad

    int cur = 0;
    patch_setup(t_begin, true);
#pragma unroll
    for (int j = 0; j < 14; ++j) patch_piece(j, wl);
    unsigned long long ph[4] = {0, 0, 0, 0};       // debug (p.dbg): cycles at tile top | MFMA loop | vmcnt wait | epilogue
    const unsigned long long wall1 = p.dbg ? wall_clock64() : 0;
    long long tp = p.dbg ? clock64() : 0;
    auto stamp = [&](int i) {
        if (p.dbg) { const long long n = clock64(); ph[i] += (unsigned long long)(n - tp); tp = n; }
    };
    int img, y0, x0;
    tile_coords(t_begin, img, y0, x0);
    for (int t = t_begin; t < t_end; ++t) {
        char* const pc = wl + cur * PATCH_BYTES;
        char* const pn = wl + (cur ^ 1) * PATCH_BYTES;
        // tile t -> (img, y0, x0) incrementally (the divisions of tile_coords cost ~500 cycles per tile on one wave)
        if (t != t_begin) {
            y0 += 4;
            if (y0 == p.H) { y0 = 0; x0 += 16; if (x0 == p.W) { x0 = 0; ++img; } }
        }
        rbase = ((img * p.H + y0) * p.W + x0) * p.ycs * 2;
        {   // patch(t+1): base offset + halo mask; on the last tile every lane is off (zeros into the idle buffer)
            int xn = x0, yn = y0 + 4, in = img;
            if (yn == p.H) { yn = 0; xn += 16; if (xn == p.W) { xn = 0; ++in; } }
            const bool live = t + 1 < t_end;
            pbase = (((in * p.H + yn - 1) * p.W) + xn - 1) * p.xcs * 2;
            pbad = !live ? 0xFFFFFFFFu
                         : (m_inval | (yn == 0 ? m_top : 0u) | (yn + 4 == p.H ? m_bot : 0u) | (xn == 0 ? m_left : 0u) |
                            (xn + 16 == p.W ? m_right : 0u));
        }
        // patch(t) must have landed.  First tile: everything issued so far.  Later tiles: its 14 pieces were issued during
        // tile t-1 and are older than that tile's output stores (8, or 16 for f32 output), which may stay in flight --
        // vmcnt retires in issue order on gfx9 (the compiler's own waitcnt insertion relies on the same).
        if (t == t_begin) wait_vmcnt<0>();
        else if (p.y_f32) wait_vmcnt<16>();
        else wait_vmcnt<8>();
        asm volatile("" ::: "memory");
        stamp(0);

        // 36 K-steps (tap, 16-channel chunk), fully unrolled; the B fragments of step s+1 are read before the MFMAs of
        // step s are issued, and the 8 + 14 LDS-DMA instructions of residual(t) / patch(t+1) are spread one per K-step
        // through the MFMA stream (issued back to back at the tile top they cost ~95 cycles each with the MFMA pipe idle).
        f32x16_t acc[2][2];
        // fragment address of (tap, kc) for pixel q = qb + ky*18 + kx:  q*128 + ((2kc | lhi) ^ ((q>>1)&7))*16
        //                                                             = fb + ((kc << 5) ^ fx),  fb, fx per (tap, pt).
        // qv is laundered through an empty asm once per tile so the 72 addresses are NOT hoisted out of the tile loop
        // (they would cost 72 registers; recomputed they are ~2 VALU per read in the MFMA shadow).
        int qv0 = qb[0], qv1 = qb[1];
        asm volatile("" : "+v"(qv0), "+v"(qv1));
        int fb[2], fx[2];
        auto tap_setup = [&](int tap) {
            const int ky = tap / 3, kx = tap - 3 * ky;
            const int q0 = qv0 + ky * PW + kx, q1 = qv1 + ky * PW + kx;
            const int cs = ((((l31 & 15) + kx) >> 1) ^ lhi) & 7;      // column-keyed swizzle, both pixel tiles share the column
            fb[0] = q0 * 128; fx[0] = cs << 4;
            fb[1] = q1 * 128; fx[1] = cs << 4;
        };
        auto frag = [&](int kc, int pt) { return *reinterpret_cast<const u32x4_t*>(pc + fb[pt] + ((kc << 5) ^ fx[pt])); };
        u32x4_t bx[2][2];
        tap_setup(0);
        bx[0][0] = frag(0, 0);
        bx[0][1] = frag(0, 1);
#pragma unroll
        for (int step = 0; step < 36; ++step) {
            const int tap = step >> 2, kc = step & 3, cb = step & 1;
            if (step + 1 < 36) {
                if (((step + 1) & 3) == 0) tap_setup((step + 1) >> 2);
                bx[cb ^ 1][0] = frag((step + 1) & 3, 0);
                bx[cb ^ 1][1] = frag((step + 1) & 3, 1);
            }
            if (HAS_RES && step >= 1 && step <= 8) residual_piece(step - 1);
            if (step >= 9 && step <= 22) patch_piece(step - 9, pn);
            if (step == 0) {
                asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, 0" : "=a"(acc[0][0]) : "a"(wa[tap][kc]), "v"(bx[cb][0]));
                asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, 0" : "=a"(acc[0][1]) : "v"(wv[tap][kc]), "v"(bx[cb][0]));
                asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, 0" : "=a"(acc[1][0]) : "a"(wa[tap][kc]), "v"(bx[cb][1]));
                asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, 0" : "=a"(acc[1][1]) : "v"(wv[tap][kc]), "v"(bx[cb][1]));
            } else {
                asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+a"(acc[0][0]) : "a"(wa[tap][kc]), "v"(bx[cb][0]));
                asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+a"(acc[0][1]) : "v"(wv[tap][kc]), "v"(bx[cb][0]));
                asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+a"(acc[1][0]) : "a"(wa[tap][kc]), "v"(bx[cb][1]));
                asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+a"(acc[1][1]) : "v"(wv[tap][kc]), "v"(bx[cb][1]));
            }
        }
        // the MFMAs are opaque to the compiler's hazard recogniser: cover the XDL-write -> VALU/DS-read wait states by hand
        asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");
        stamp(1);

        // ---- epilogue, register-direct: no LDS staging.  Per (pixel tile pt, channel tile ct) the lane's four channel quads are
        // scaled / shifted (+ residual, read as 8 bytes per quad), rounded to bf16, ReLU'd on the packed pairs, and v_permlane32_swap
        // pairs quad j of the lower half-wave with quad j of the upper one (and quad j+1 likewise), so that every lane ends up with 8
        // consecutive channels of its pixel = one 16-byte store.  Same arithmetic, same order as the staged form: same bits. ----
        if (HAS_RES) {
            asm volatile("" ::: "memory");
            wait_vmcnt<14>();                      // residual(t) landed; the 14 younger pieces of patch(t+1) may still be in flight
            asm volatile("" ::: "memory");
        }
        stamp(2);
        const int tile_pix = (img * p.H + y0) * p.W + x0;
#pragma unroll
        for (int pt = 0; pt < 2; ++pt) {
            const char* const rrow = resbuf + (pt * 32 + r_px) * 128 + lhi * 8;
            const int so = (tile_pix + 2 * pt * p.W) * p.ycs * 2;
#pragma unroll
            for (int ct = 0; ct < 2; ++ct) {
                // one accumulator tile out of the AGPRs at a time: this (empty, volatile, memory-clobbering) asm "rewrites" the tile,
                // so its 16 reads cannot be hoisted above the previous tile's stores -- all 64 at once do not fit the 256 VGPRs
                asm volatile("" : "+a"(acc[pt][ct])::"memory");
#pragma unroll
                for (int jp = 0; jp < 2; ++jp) {
                    uint32_t pk[2][2];
#pragma unroll
                    for (int jj = 0; jj < 2; ++jj) {
                        const int j = 2 * jp + jj;
                        const f32x4_t scv = *reinterpret_cast<const f32x4_t*>(ssd + (ct * 4 + j) * 64);
                        const f32x4_t shv = *reinterpret_cast<const f32x4_t*>(ssd + (ct * 4 + j) * 64 + 16);
                        f32x4_t a = f32x4_t{acc[pt][ct][j * 4], acc[pt][ct][j * 4 + 1], acc[pt][ct][j * 4 + 2], acc[pt][ct][j * 4 + 3]} * scv + shv;
                        if constexpr (HAS_RES) {
                            const uint2 rr = *reinterpret_cast<const uint2*>(rrow + (((ct * 4 + j) ^ r_f) << 4));
                            a += f32x4_t{__uint_as_float(rr.x << 16), __uint_as_float(rr.x & 0xFFFF0000u),
                                         __uint_as_float(rr.y << 16), __uint_as_float(rr.y & 0xFFFF0000u)};
                        }
                        if (p.y_f32) {             // f32 output (tests, the decoder head's shape never comes here): the quad as it is
                            if (p.relu) {
#pragma unroll
                                for (int e = 0; e < 4; ++e) a[e] = fmaxf(a[e], 0.f);
                            }
                            __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4_t, a), rs_y,
                                                                   (unsigned)((e_px * p.ycs + g * 64 + ct * 32 + 8 * j + 4 * lhi) * 4),
                                                                   (tile_pix + 2 * pt * p.W) * p.ycs * 4, 0);
                            continue;
                        }
                        pk[jj][0] = pack_bf16x2(a[0], a[1]);
                        pk[jj][1] = pack_bf16x2(a[2], a[3]);
                        if (p.relu) {
#pragma unroll
                            for (int e = 0; e < 2; ++e) {
                                const s16x2_t h = __builtin_bit_cast(s16x2_t, pk[jj][e]);
                                pk[jj][e] = __builtin_bit_cast(uint32_t, __builtin_elementwise_max(h, s16x2_t{0, 0}));
                            }
                        }
                    }
                    if (p.y_f32) continue;
                    const auto sa = __builtin_amdgcn_permlane32_swap(pk[0][0], pk[1][0], false, false);
                    const auto sb = __builtin_amdgcn_permlane32_swap(pk[0][1], pk[1][1], false, false);
                    const u32x4_t o = {sa[0], sb[0], sa[1], sb[1]};
                    __builtin_amdgcn_raw_buffer_store_b128(o, rs_y, y_lane + (ct * 32 + jp * 16) * 2, so, 0);
                }
            }
        }
        asm volatile("" ::: "memory");
        stamp(3);
        cur ^= 1;
    }
    if (p.dbg && lane == 0 && wave == 0) {         // 8 x u64 per workgroup: 4 phase sums (shader cycles), 3 wall stamps (100 MHz)
        unsigned long long* d = p.dbg + ((size_t)blockIdx.y * gridDim.x + blockIdx.x) * 8;
        for (int i = 0; i < 4; ++i) d[i] = ph[i];
        d[4] = wall0; d[5] = wall1; d[6] = wall_clock64(); d[7] = 1;
    }
    span_stamp(p, true);
#endif
}

template <bool HAS_RES, int VER = 1>
int launch_regw(ConvArgs& a, int groups, hipStream_t s) {
    constexpr int lds = 4 * (2 * 14 * 1024 + 8 * 1024) + 512;    // 4 waves' patch/residual buffers + scale/shift
    static std::atomic<unsigned long long> attr_mask{0};
    int dev = 0;
    (void)hipGetDevice(&dev);
    static int n_cu[64] = {0};
    if (!((attr_mask.load(std::memory_order_acquire) >> (dev & 63)) & 1ull)) {
        (void)hipFuncSetAttribute(VER == 2 ? reinterpret_cast<const void*>(&conv3x3_c64_regw2_kernel<HAS_RES>)
                                           : reinterpret_cast<const void*>(&conv3x3_c64_regw_kernel<HAS_RES>),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, lds);
        hipDeviceProp_t prop;
        n_cu[dev & 63] = (hipGetDeviceProperties(&prop, dev) == hipSuccess && prop.multiProcessorCount > 0)
                             ? prop.multiProcessorCount : 256;
        attr_mask.fetch_or(1ull << (dev & 63), std::memory_order_release);
    }
    const long tiles = (long)a.M * (a.H / 4) * (a.W / 16);
    int wgs = (n_cu[dev & 63] + groups - 1) / groups;            // one 4-wave workgroup per CU, split over the groups
    wgs = (wgs + 7) / 8 * 8;
    if ((long)wgs * 4 > tiles) wgs = (int)((tiles + 3) / 4);
    if (VER == 2) hipLaunchKernelGGL((conv3x3_c64_regw2_kernel<HAS_RES>), dim3(wgs, groups), dim3(256), lds, s, a);
    else hipLaunchKernelGGL((conv3x3_c64_regw_kernel<HAS_RES>), dim3(wgs, groups), dim3(256), lds, s, a);
    return w2c_launch_status();
}

int launch_regw2_any(ConvArgs& a, int groups, hipStream_t s) {       // register-direct epilogue
    if (a.ks != 3 || a.stride != 1 || a.Cin != 64 || a.Cout != 64 || a.H % 4 != 0 || a.W % 16 != 0 || a.ygs != 64 || a.y8 || !a.y) return W2C_E_ARG;
    if ((size_t)a.M * a.H * a.W * a.xcs * 2 >= (1ull << 31) || (size_t)a.M * a.H * a.W * a.ycs * 2 >= (1ull << 31)) return W2C_E_ARG;
    return a.res ? launch_regw<true, 2>(a, groups, s) : launch_regw<false, 2>(a, groups, s);
}

int launch_regw_any(ConvArgs& a, int groups, hipStream_t s) {
    if (a.ks != 3 || a.stride != 1 || a.Cin != 64 || a.Cout != 64 || a.H % 4 != 0 || a.W % 16 != 0 || a.ygs != 64 || a.y8 || !a.y) return W2C_E_ARG;
    if ((size_t)a.M * a.H * a.W * a.xcs * 2 >= (1ull << 31) || (size_t)a.M * a.H * a.W * a.ycs * 2 >= (1ull << 31)) return W2C_E_ARG;
    return a.res ? launch_regw<true>(a, groups, s) : launch_regw<false>(a, groups, s);
}


template <int TH, int TW, int BN, int WM, int WN, int STAGES, int PB = 2, bool F8 = false>
int launch_patch(ConvArgs& a, int groups, hipStream_t s) {
    constexpr int CK = OpT<F8>::CK;
    if (a.ks != 3 || a.stride != 1 || a.Cin % CK != 0 || a.Cout % BN != 0 || a.H % TH != 0 || a.W % TW != 0)
        return W2C_E_ARG;
    if (PB == 1 && a.Cin != CK) return W2C_E_ARG;
    a.ntm = a.M * (a.H / TH) * (a.W / TW);
    a.ntn = a.Cout / BN;
    if ((long)a.ntm * a.ntn * ((a.H / TH) * (a.W / TW) > a.ntn ? (a.H / TH) * (a.W / TW) : a.ntn) >= (1ll << 32)) return W2C_E_ARG;   // (fast-division range)
    a.mg_ntn = w2c_magic((unsigned)a.ntn);
    a.mg_qn = w2c_magic((unsigned)(a.ntn >> 1));
    a.mg_tx = w2c_magic((unsigned)(a.W / TW));
    a.mg_txy = w2c_magic((unsigned)((a.H / TH) * (a.W / TW)));
    constexpr int ring = PB * (TH + 2) * (TW + 2) * 128 + STAGES * BN * 128;
    constexpr int epi = TH * TW * (BN + 4) * 4;
    constexpr int lds = ring > epi ? ring : epi;
    static_assert(lds <= 160 * 1024, "LDS");
    static std::atomic<unsigned long long> attr_mask{0};
    int dev = 0;
    (void)hipGetDevice(&dev);
    if (!((attr_mask.load(std::memory_order_acquire) >> (dev & 63)) & 1ull)) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&conv3x3_patch_kernel<TH, TW, BN, WM, WN, STAGES, PB, F8>),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        attr_mask.fetch_or(1ull << (dev & 63), std::memory_order_release);
    }
    dim3 grid(a.ntm * a.ntn, groups);
    const int xcd2d_mode = w2c_option(W2C_OPT_XCD2D);            // (tests and tools/ab_xcd2d.sh switch it through w2c_set_option)
    // weights of a group larger than half an XCD's L2 and at least as large as its input: split both operands over the XCDs
    const long wbytes = (long)a.Cout * 9 * a.Cin * OpT<F8>::ES, xbytes = (long)a.M * a.H * a.W * a.Cin * OpT<F8>::ES;
    a.xcd2d = (xcd2d_mode == 2 || (xcd2d_mode == 1 && wbytes >= (2 << 20) && 2 * wbytes >= xbytes)) && groups == 2 && !(a.ntm & 1) &&
              !(a.ntn & 1) && (a.ntm * a.ntn) % 4 == 0;
    if (a.xcd2d) grid = dim3(a.ntm * a.ntn * 2, 1);
    hipLaunchKernelGGL((conv3x3_patch_kernel<TH, TW, BN, WM, WN, STAGES, PB, F8>), grid, dim3(64 * WM * WN), lds, s, a);
    return w2c_launch_status();
}

#include "conv_wreg.inl"
#include "conv_regh.inl"
#include "conv_s2wreg.inl"
#include "conv_s2regh.inl"

template <int BM, int BN, int BK, int STAGES>
constexpr int conv_lds_bytes() {
    constexpr int ring = STAGES * (BM + BN) * BK * 2;
    constexpr int epi = BM * (BN + 4) * 4;
    return ring > epi ? ring : epi;
}

// The epilogue of a split-K conv on the summed partials of 8 consecutive channels of one pixel: BN scale/shift (+ residual) (+ ReLU),
// store.  ONE body for the finish kernel and the in-workgroup form: the two must round alike.
__device__ __forceinline__ void splitk_epilogue8(const ConvArgs& p, f32x4_t v0, f32x4_t v1, int ch, size_t off) {
    v0 = v0 * *reinterpret_cast<const f32x4_t*>(p.scale + ch) + *reinterpret_cast<const f32x4_t*>(p.shift + ch);
    v1 = v1 * *reinterpret_cast<const f32x4_t*>(p.scale + ch + 4) + *reinterpret_cast<const f32x4_t*>(p.shift + ch + 4);
    float v[8] = {v0[0], v0[1], v0[2], v0[3], v1[0], v1[1], v1[2], v1[3]};
    if (p.res) {
        const uint4 rr = *reinterpret_cast<const uint4*>(p.res + off);
        const uint32_t rw[4] = {rr.x, rr.y, rr.z, rr.w};
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            v[2 * e] += bf16_to_f32((uint16_t)(rw[e] & 0xFFFFu));
            v[2 * e + 1] += bf16_to_f32((uint16_t)(rw[e] >> 16));
        }
    }
    if (p.relu) {
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] = fmaxf(v[e], 0.f);
    }
    if (p.y_f32) {
        float* yo = reinterpret_cast<float*>(p.y) + off;
        *reinterpret_cast<f32x4_t*>(yo) = f32x4_t{v[0], v[1], v[2], v[3]};
        *reinterpret_cast<f32x4_t*>(yo + 4) = f32x4_t{v[4], v[5], v[6], v[7]};
    } else {
        uint4 o;
        o.x = pack_bf16x2(v[0], v[1]); o.y = pack_bf16x2(v[2], v[3]);
        o.z = pack_bf16x2(v[4], v[5]); o.w = pack_bf16x2(v[6], v[7]);
        *reinterpret_cast<uint4*>(reinterpret_cast<uint16_t*>(p.y) + off) = o;
    }
}

// Second half of a split-K conv: out = act(scale * sum_z partial_z + shift (+ residual)); one thread per 8 channels.
template <int BM, int BN>
__global__ __launch_bounds__(256) void splitk_finish_kernel(ConvArgs p) {
    constexpr int CG = BN / 8;
    const int g = blockIdx.y;
    const long total = (long)p.ntm * p.ntn * BM * CG;
    const long id = (long)blockIdx.x * 256 + threadIdx.x;
    span_stamp(p, true);                                     // (a launch span ends with its finish kernel; thread 0 may leave early)
    if (id >= total) return;
    const int cg = (int)(id % CG);
    const long t1 = id / CG;
    const int r = (int)(t1 % BM);
    const int tile = (int)(t1 / BM);                         // == xcd_remap'ed tile id used by the producer
    const int tm = tile / p.ntn, tn = tile - tm * p.ntn;
    const int gr = tm * BM + r;
    if (gr >= p.rows) return;
    const float* sp = p.ws + (((size_t)g * (p.ntm * p.ntn) + tile) * p.n_split) * (size_t)(BM * BN) + r * BN + cg * 8;
    f32x4_t v0 = {0.f, 0.f, 0.f, 0.f}, v1 = v0;
    for (int z = 0; z < p.n_split; ++z) {
        v0 += *reinterpret_cast<const f32x4_t*>(sp + (size_t)z * (BM * BN));
        v1 += *reinterpret_cast<const f32x4_t*>(sp + (size_t)z * (BM * BN) + 4);
    }
    splitk_epilogue8(p, v0, v1, g * p.Cout + tn * BN + cg * 8, (size_t)gr * p.ycs + (size_t)g * p.ygs + tn * BN + cg * 8);
    span_stamp(p, true);
}

// Split-K INSIDE one workgroup, for the tail layers (policy conv2..5, the decoder's last 3x3: <= 1280 output rows under a
// 2304-long K): wave z of the workgroup is split z of the split-K launch above -- the same K-step range [z kt / S, (z+1) kt / S), the
// same MFMA sequence per 32x32 block -- and the S partial tiles meet in LDS instead of an f32 workspace in HBM, where 256 threads add
// them in split order and run the finish kernel's epilogue.  One launch instead of two, no workspace round trip, and bit-identical
// to the two-launch form (same products, same order of additions).  Operands go global -> VGPR fragments directly (the 16 bytes a
// lane feeds the MFMA are 16 contiguous bytes of one pixel / one filter row): with 3..5 K-steps per wave there is no ring to
// amortise, only latency to overlap -- all loads of K-step t+1 are in flight under the MFMAs of K-step t.
// Tile = BM_ x 32 outputs, LDS = S x BM_ x 36 x 4 bytes, kept under 80 KB so that the workgroup fits beside ONE workgroup of the
// other launch chain's conv kernels on a CU (they take 68..79 KB each): a bigger footprint would wait for a fully drained CU.
template <int BM_, int UF, int MAXT>
__global__ __launch_bounds__(MAXT) void conv_inwg_splitk_kernel(ConvArgs p) {
#if defined(__HIP_DEVICE_COMPILE__)
    constexpr int BN_ = 32, MI = BM_ / 32, CLD = BN_ + 4, CG = BN_ / 8;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float* const part = reinterpret_cast<float*>(smem);          // [split][BM_][CLD]
    span_stamp(p, false);
    const int tid = threadIdx.x, lane = tid & 63;
    const int z = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int n_split = p.n_split;
    const int g = blockIdx.y;
    const int tile = blockIdx.x;
    const int tm = tile / p.ntn, tn = tile - tm * p.ntn;
    const int m0 = tm * BM_, n0 = tn * BN_;
    const int l31 = lane & 31, lhi = lane >> 5;

    const char* xg = reinterpret_cast<const char*>(p.x) + (size_t)g * p.Cin * 2;
    const int Ktot = p.ktiles * 64;
    const char* wg = reinterpret_cast<const char*>(p.w) + (size_t)g * p.Cout * Ktot * 2;
    const __amdgpu_buffer_rsrc_t rs_x = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<char*>(xg), 0, (int)((size_t)p.M * p.H * p.W * p.xcs * 2), 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_w = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<char*>(wg), 0, (int)((size_t)p.Cout * Ktot * 2), 0x00020000);
    int a_iy0[MI], a_ix0[MI], a_base[MI];
#pragma unroll
    for (int i = 0; i < MI; ++i) {
        const int gr = m0 + i * 32 + l31;
        if (gr < p.rows) {
            const int hw = p.Ho * p.Wo;
            const int m = w2c_fastdiv(gr, hw, p.mg_hw);              // (rows < 2^29, hw <= rows: exact; round 6)
            const int rem = gr - m * hw;
            const int oy = w2c_fastdiv(rem, p.Wo, p.mg_wo), ox = rem - oy * p.Wo;
            a_iy0[i] = oy * p.stride - p.pad;
            a_ix0[i] = ox * p.stride - p.pad;
            a_base[i] = (int)(((long)m * p.H * p.W + (long)a_iy0[i] * p.W + a_ix0[i]) * p.xcs * 2 + lhi * 16);
        } else {
            a_iy0[i] = -100000; a_ix0[i] = 0; a_base[i] = 0;
        }
    }
    const unsigned b_off = (unsigned)((size_t)(n0 + l31) * Ktot * 2 + lhi * 16);

    const int t_begin = (z * p.ktiles) / n_split, t_end = ((z + 1) * p.ktiles) / n_split;
    const int ntap = p.ks * p.ks;
    const bool s2order = p.ks == 3 && p.stride == 2;
    int st_ct = t_begin / ntap;
    int st_tap = t_begin - st_ct * ntap;
    int st_ky = st_tap / p.ks, st_kx = st_tap % p.ks;
    if (s2order) s2_tap(st_tap, st_ky, st_kx);

    struct Frags { u32x4_t a[4][MI], b[4]; };
    auto load = [&](Frags& f, bool live) {                        // K-step (st_ct, st_tap): 16 bytes per lane per MFMA operand
        const int tap_off = (st_ky * p.W + st_kx) * p.xcs * 2;
        const int c_off = st_ct * 128;
        const int w_off = (st_ky * p.ks + st_kx) * p.Cin * 2 + c_off;
#pragma unroll
        for (int i = 0; i < MI; ++i) {
            const int iy = a_iy0[i] + st_ky, ix = a_ix0[i] + st_kx;
            const bool ok = live & ((unsigned)iy < (unsigned)p.H) & ((unsigned)ix < (unsigned)p.W);
            const unsigned vo = ok ? (unsigned)(a_base[i] + tap_off) : 0x80000000u;     // out of range: the bounds check returns 0
#pragma unroll
            for (int kk = 0; kk < 4; ++kk) f.a[kk][i] = __builtin_amdgcn_raw_buffer_load_b128(rs_x, vo + kk * 32, c_off, 0);
        }
        const unsigned wo = live ? b_off : 0x80000000u;
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) f.b[kk] = __builtin_amdgcn_raw_buffer_load_b128(rs_w, wo + kk * 32, w_off, 0);
        if (++st_tap == ntap) { st_tap = 0; ++st_ct; }
        if (s2order) s2_tap(st_tap, st_ky, st_kx);
        else { st_ky = st_tap / p.ks; st_kx = st_tap - st_ky * p.ks; }
    };
    f32x16_t acc[MI];
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[i][e] = 0.f;
    auto compute = [&](const Frags& f) {
#pragma unroll
        for (int kk = 0; kk < 4; ++kk)
#pragma unroll
            for (int i = 0; i < MI; ++i)
                acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8_t, f.b[kk]),
                                                                 __builtin_bit_cast(bf16x8_t, f.a[kk][i]), acc[i], 0, 0, 0);
    };
    const int KT = t_end - t_begin;
    if constexpr (UF > 0) if (KT > UF) __builtin_trap();      // (the host picks UF >= the longest split)
    if constexpr (UF > 0) {
        // short range (the dispatched case: 3..5 K-steps per wave): EVERY load of the wave is issued before its first MFMA -- one
        // memory latency per wave instead of one per K-step.  Slots past the range load nothing (offset out of range) and are skipped.
        Frags f[UF > 0 ? UF : 1];
#pragma unroll
        for (int i = 0; i < UF; ++i) load(f[i], i < KT);
#pragma unroll
        for (int i = 0; i < UF; ++i)
            if (i < KT) compute(f[i]);
    } else {
        Frags f0, f1;
        int t = t_begin;
        load(f0, true);
        while (true) {
            if (t + 1 < t_end) load(f1, true);
            compute(f0);
            if (++t >= t_end) break;
            if (t + 1 < t_end) load(f0, true);
            compute(f1);
            if (++t >= t_end) break;
        }
    }
    // partial tile of split z -> LDS, f32 [BM_][CLD] (D column = pixel, D rows 8 eg + 4 lhi .. + 4 = four consecutive channels)
    float* const mine = part + (size_t)z * (BM_ * CLD);
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int eg = 0; eg < 4; ++eg)
            *reinterpret_cast<f32x4_t*>(mine + (i * 32 + l31) * CLD + eg * 8 + lhi * 4) =
                f32x4_t{acc[i][eg * 4], acc[i][eg * 4 + 1], acc[i][eg * 4 + 2], acc[i][eg * 4 + 3]};
    __syncthreads();
    // read-out item id -> (row r = id % BM_, channel group cg = id / BM_): a wave's lanes walk consecutive ROWS of one channel group,
    // so the 16 lanes ds_read_b128 services together read 16-byte runs at 144-byte (CLD floats) steps -- 36 r mod 64 dwords is a
    // bijection on any 16 rows that differ mod 16: every bank once.  (Rounds 3-5 mapped r = id / CG, cg = id % CG: four lanes per row,
    // rows {0, 3, 5, 6} of a service group collide pairwise -- SQ_LDS_BANK_CONFLICT 25 % of the LDS cycles of these launches.)  The
    // order of the additions per output is unchanged: same bits.
    for (int id = tid; id < BM_ * CG; id += (int)blockDim.x) {      // (fewer than 4 splits: fewer threads than read-out items)
        const int cg = id / BM_, r = id - cg * BM_;
        const int gr = m0 + r;
        if (gr >= p.rows) continue;
        const float* sp = part + r * CLD + cg * 8;
        f32x4_t v0 = {0.f, 0.f, 0.f, 0.f}, v1 = v0;
        for (int s = 0; s < n_split; ++s) {
            v0 += *reinterpret_cast<const f32x4_t*>(sp + (size_t)s * (BM_ * CLD));
            v1 += *reinterpret_cast<const f32x4_t*>(sp + (size_t)s * (BM_ * CLD) + 4);
        }
        splitk_epilogue8(p, v0, v1, g * p.Cout + n0 + cg * 8, (size_t)gr * p.ycs + (size_t)g * p.ygs + n0 + cg * 8);
    }
    span_stamp(p, true);
#endif
}

template <int BM_, int UF, int MAXT>
int launch_conv_inwg_splitk(ConvArgs& a, int groups, int ksplit, hipStream_t s) {
    a.cin_tiles = a.Cin / 64;
    a.ktiles = a.ks * a.ks * a.cin_tiles;
    a.ntm = (a.rows + BM_ - 1) / BM_;
    a.ntn = a.Cout / 32;
    a.n_split = ksplit;
    const int lds = ksplit * BM_ * 36 * 4;
    static std::atomic<unsigned long long> attr_mask{0};
    int dev = 0;
    (void)hipGetDevice(&dev);
    if (!((attr_mask.load(std::memory_order_acquire) >> (dev & 63)) & 1ull)) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_inwg_splitk_kernel<BM_, UF, MAXT>),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, 80 * 1024);
        attr_mask.fetch_or(1ull << (dev & 63), std::memory_order_release);
    }
    hipLaunchKernelGGL((conv_inwg_splitk_kernel<BM_, UF, MAXT>), dim3(a.ntm * a.ntn, groups), dim3(64 * ksplit), lds, s, a);
    return w2c_launch_status();
}

// split-K launch of the generic kernel (grid.z = ksplit) + its finish kernel
template <int BM, int BN, int WM, int WN>
int launch_conv_splitk(ConvArgs& a, int groups, int ksplit, hipStream_t s) {
    constexpr int BK = 64, STAGES = 2;
    if (a.Cin % BK != 0 || a.Cout % BN != 0) return W2C_E_ARG;
    a.cin_tiles = a.Cin / BK;
    a.ktiles = a.ks * a.ks * a.cin_tiles;
    a.ntm = (a.rows + BM - 1) / BM;
    a.ntn = a.Cout / BN;
    if (ksplit < 1 || ksplit > a.ktiles) return W2C_E_ARG;
    constexpr int lds = conv_lds_bytes<BM, BN, BK, STAGES>();
    static_assert(lds <= 64 * 1024, "LDS");
    a.n_split = ksplit;
    dim3 grid(a.ntm * a.ntn, groups, ksplit);
    hipLaunchKernelGGL((conv_igemm_kernel<BM, BN, WM, WN, BK, STAGES, true>), grid, dim3(64 * WM * WN), lds, s, a);
    const int rc = w2c_launch_status();
    if (rc != W2C_OK) return rc;
    const long threads = (long)a.ntm * a.ntn * BM * (BN / 8);
    hipLaunchKernelGGL((splitk_finish_kernel<BM, BN>), dim3((unsigned)((threads + 255) / 256), groups), dim3(256), 0, s, a);
    return w2c_launch_status();
}

// tile shape + split count for a split-K launch; ksplit == 1 means "not worth splitting".
// The split count is a function of the LAYER (channels, taps, output map size) only, never of the image count M or of the
// number of groups launched together: the
// partial sums of a split are added in split order, so a split that moved with M would make a shard of the agents
// round differently from the unsharded batch.  It is sized for a nominal 16-image batch (~2 workgroups per CU, >= 3
// K-steps per split); other batch sizes get the same arithmetic with more or fewer workgroups.
struct SplitPlan { int bm, bn, tiles, ksplit; };
SplitPlan plan_splitk(const ConvArgs& a, int groups, int want) {
    SplitPlan sp;
    sp.bn = (a.Cout % 64 == 0) ? 64 : 32;
    sp.bm = sp.bn == 64 ? 64 : 128;
    sp.tiles = (int)(((long)a.rows + sp.bm - 1) / sp.bm) * (a.Cout / sp.bn) * groups;
    const int kt = a.ks * a.ks * (a.Cin / 64);
    int ks = want;
    if (ks <= 0) {
        const long rows16 = 16L * a.Ho * a.Wo;
        const long tiles16 = ((rows16 + sp.bm - 1) / sp.bm) * (a.Cout / sp.bn);     // per group: running two trunks side by
                                                                                     // side or one at a time rounds alike
        ks = (int)((512 + tiles16 - 1) / tiles16);
        if (ks > kt / 3) ks = kt / 3;
        if (tiles16 >= 256) ks = 1;
    }
    if (ks < 1) ks = 1;
    if (ks > kt) ks = kt;
    sp.ksplit = ks;
    return sp;
}

template <int BM, int BN, int WM, int WN, int BK, int STAGES, bool F8 = false, bool DUAL = false>
int launch_conv(ConvArgs& a, int groups, hipStream_t s) {
    constexpr int CK = OpT<F8>::CK;
    if (a.Cin % CK != 0 || a.Cout % BN != 0) return W2C_E_ARG;
    if (DUAL && (a.ks != 3 || a.stride != 2 || !a.w2 || !a.y2)) return W2C_E_ARG;
    a.cin_tiles = a.Cin / CK;
    a.ktiles = a.ks * a.ks * a.cin_tiles;
    a.ntm = (a.rows + BM - 1) / BM;
    a.ntn = a.Cout / BN;
    constexpr int ring = STAGES * (BM + BN) * BK * 2 + (DUAL ? BN * BK * 2 : 0);      // DUAL: + the 1x1 conv's weight tile
    constexpr int lds = ring > conv_lds_bytes<BM, BN, BK, STAGES>() ? ring : conv_lds_bytes<BM, BN, BK, STAGES>();
    static_assert(lds <= 160 * 1024, "LDS");
    // dynamic LDS above 64 KiB needs the attribute once per device; keep a per-device bit.
    static std::atomic<unsigned long long> attr_mask{0};
    int dev = 0;
    (void)hipGetDevice(&dev);
    if (!((attr_mask.load(std::memory_order_acquire) >> (dev & 63)) & 1ull)) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_igemm_kernel<BM, BN, WM, WN, BK, STAGES, false, F8, DUAL>),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, lds);
        attr_mask.fetch_or(1ull << (dev & 63), std::memory_order_release);
    }
    dim3 grid(a.ntm * a.ntn, groups);
    hipLaunchKernelGGL((conv_igemm_kernel<BM, BN, WM, WN, BK, STAGES, false, F8, DUAL>), grid, dim3(64 * WM * WN), lds, s, a);
    return w2c_launch_status();
}

// stride-2 BasicBlock front (conv1 3x3/s2 + downsample 1x1/s2) in one launch; tile chosen like pick_variant's generic rules
template <bool F8>
int launch_dual(ConvArgs& a, int groups, int variant, hipStream_t s) {
    const long rows = a.rows;
    (void)rows;
    if (variant < 0) variant = 6;       // two accumulator sets: 64x64 tiles keep 2 waves per SIMD (128x128: 288 registers, 1 wave)
    switch (variant) {
        case 0: return launch_conv<128, 128, 2, 2, 64, 2, F8, true>(a, groups, s);
        case 3: return launch_conv<128, 64, 2, 2, 64, 2, F8, true>(a, groups, s);
        case 6: return launch_conv<64, 64, 2, 2, 64, 2, F8, true>(a, groups, s);
        default: return W2C_E_ARG;
    }
}

// Variant table (index = `variant` of w2c_conv_igemm_bf16_variant; tools/bench_conv.py sweeps it).  Only the variants
// pick_variant() dispatches are built; the ~30 other tile / ring shapes measured in round 1
// (profiles/r01_b_conv_variant_sweep.txt: BK=32 rings, 8-wave 256-row tiles, 8x32 / 16x16 / 4x32 patches, deeper rings,
// the layer2 register-resident sibling) lost everywhere and were removed.
int launch_variant(int variant, ConvArgs& a, int groups, hipStream_t s) {
    switch (variant) {
        // generic implicit GEMM (stride 2, 1x1, narrow maps): BM x BN, 4 waves, 2-deep ring
        case 0: return launch_conv<128, 128, 2, 2, 64, 2>(a, groups, s);
        case 3: return launch_conv<128, 64, 2, 2, 64, 2>(a, groups, s);
        case 6: return launch_conv<64, 64, 2, 2, 64, 2>(a, groups, s);
        case 8: return launch_conv<128, 32, 4, 1, 64, 2>(a, groups, s);
        // patch-staged 3x3 stride-1 kernels: 8 x 16-pixel tiles, <= 80 KB of LDS so TWO workgroups share a CU (one's
        // prologue / epilogue under the other's MFMAs)
        case 30: return launch_patch<8, 16, 128, 2, 2, 2>(a, groups, s);     // 4 waves, 128 output channels per tile
        case 36: return launch_patch<8, 16, 64, 4, 2, 3>(a, groups, s);      // 8 waves, 64 channels, 3-deep weight ring
        // Cin == 64 (one channel chunk): a single patch buffer -> ~39 KB of LDS -> four workgroups per CU
        case 38: return launch_patch<8, 16, 64, 2, 2, 2, 1>(a, groups, s);
        // (round-3 experiment, removed: this kernel with 128-pixel x 64-channel WAVE tiles -- 4 waves on 16 x 16 x 128 / 2 waves on
        //  8 x 16 x 128, one wave per SIMD -- 707-710 TFLOP/s on 8x deeper K against 906-991 / 792-871 for 36 / 30: the tile shape
        //  alone buys nothing while the reads and the MFMAs of a step run one after the other; profiles/r03_wreg_kernel.txt section 1)
        // weights-to-registers kernels (conv_wreg.inl; `w` must be in w2c_pack_wfrag_bf16 order): NN channel blocks x KS K groups
        case 80: return launch_wreg<2, 2>(a, groups, s);
        case 81: return launch_wreg<1, 4>(a, groups, s);
        case 83: return launch_wreg<1, 2>(a, groups, s);
        case 93: return launch_wreg<1, 4, 0, 4>(a, groups, s);   // 81 with the weights 4 / 2 K-steps ahead instead of 8
        case 94: return launch_wreg<1, 4, 0, 2>(a, groups, s);
        case 193: return launch_wreg<1, 4, 0, 4, 2, true>(a, groups, s);      // 93 writing f32 (w2c_conv3x3_wreg_f32out)
        case 95: return launch_wreg<1, 4, 0, 4, 1>(a, groups, s);             // 93 with 32 channels per wave: twice the workgroups (bit-identical)
        case 195: return launch_wreg<1, 4, 0, 4, 1, true>(a, groups, s);
        // (round 4, removed in round 5: 32 channels per wave -- 8 waves on the same workgroup tile, bit-identical -- for the small launches:
        //  +-8 % alone, slower in every forward (profiles/r04_rank_shapes.txt): those launches are not short of waves)
        // layer1 (Cin = Cout = 64): weights stationary in registers, one persistent wave per SIMD, no barriers
        case 50: return launch_regw_any(a, groups, s);
        case 52: return launch_regw2_any(a, groups, s);
        // layer1, two waves per SIMD (conv_regh.inl; `w` in w2c_pack_wfrag_bf16 order)
        case 54: return launch_regh_any(a, groups, s);
        // stride-2 3x3 on polyphase halo patches (here without the fused downsample)
        case 60: return launch_s2patch<64, 3, false>(a, groups, s);
        case 61: return launch_s2patch<128, 2, false>(a, groups, s);
        case 62: return launch_s2patch<64, 2, false>(a, groups, s);
        default: return W2C_E_ARG;
    }
}

// fp8 (e4m3, MX-scaled K=64 MFMA) siblings: identical byte geometry -- a K-step is still one 128-byte run per row, now
// 128 channels -- so the same tile shapes apply with half the K-steps.
int launch_variant_f8(int variant, ConvArgs& a, int groups, hipStream_t s) {
    switch (variant) {
        case 0: return launch_conv<128, 128, 2, 2, 64, 2, true>(a, groups, s);
        case 3: return launch_conv<128, 64, 2, 2, 64, 2, true>(a, groups, s);
        case 6: return launch_conv<64, 64, 2, 2, 64, 2, true>(a, groups, s);
        case 30: return launch_patch<8, 16, 128, 2, 2, 2, 2, true>(a, groups, s);
        case 36: return launch_patch<8, 16, 64, 4, 2, 3, 2, true>(a, groups, s);
        case 38: return launch_patch<8, 16, 64, 2, 2, 2, 1, true>(a, groups, s);      // Cin == 128: one chunk
        case 40: return launch_patch<8, 16, 128, 2, 2, 2, 1, true>(a, groups, s);     // Cin == 128, 128 channels per tile
        case 60: return launch_s2patch<64, 3, true>(a, groups, s);
        case 61: return launch_s2patch<128, 2, true>(a, groups, s);
        default: return W2C_E_ARG;
    }
}

int pick_variant_f8(const ConvArgs& a, int groups) {
    const long rows = a.rows;
    const int Cout = a.Cout;
    if (a.stride == 2 && a.ks == 3 && a.Ho % 8 == 0 && a.Wo % 16 == 0 && Cout % 64 == 0 && !(a.H & 1) && !(a.W & 1) && !a.res &&
        !w2c_option(W2C_OPT_NO_S2PATCH))
        return 60;
    if (a.ks == 3 && a.stride == 1 && a.H % 8 == 0 && a.W % 16 == 0) {
        const long tiles = (long)a.M * (a.H / 8) * (a.W / 16) * groups;
        if (a.Cin == 128 && Cout % 128 == 0 && tiles * (Cout / 128) >= 256) return 40;
        if (a.Cin == 128 && Cout % 64 == 0 && tiles * (Cout / 64) >= 64) return 38;
        if (Cout % 64 == 0 && tiles * (Cout / 64) >= 64) return 36;
    }
    if (Cout % 128 == 0 && (rows / 128) * (Cout / 128) * groups >= 512) return 0;
    if (Cout % 64 == 0 && (rows / 128) * (Cout / 64) * groups >= 512) return 3;
    return 6;
}

// Per-layer kernel choice, from the per-layer sweeps of tools/bench_conv.py on MI355X
// (profiles/r01_b_conv_variant_sweep.txt; run-to-run noise of single cells is ~+-8 %).  Stride-1 3x3 convs go to
// the patch-staged kernel with 128-pixel (8x16) tiles sized so that >= 2 workgroups share a CU:
//   Cin == Cout == 64 (layer1), >= 4 tiles per wave -> v50: register-resident weights, persistent waves, no barriers
//   Cin == 64  otherwise                    -> v38: single patch buffer, 39 KB LDS, four workgroups per CU
//   Cin == 128 (layer2)                    -> v30: 128 output channels per tile, 4 waves
//   deeper                                  -> v36: 64 output channels per tile, 8 waves, 3-deep weight ring
// everything else (stride 2, 1x1, maps narrower than 16 pixels) to the generic implicit GEMM, whose tile is chosen
// to keep >= ~2 workgroups per CU.
int pick_variant(const ConvArgs& a, int groups) {
    const long rows = a.rows;
    const int Cout = a.Cout;
    if (a.ks == 3 && a.stride == 1 && a.H % 8 == 0 && a.W % 16 == 0) {
        const long tiles = (long)a.M * (a.H / 8) * (a.W / 16) * groups;
        // layer1 at full size: weights stationary in registers (v50) once every wave of the chip gets >= 4 tiles
        // (its 7 us weight prologue is per launch); smaller problems stay on the ring kernel
        if (a.Cin == 64 && Cout == 64 && a.ygs == 64 && a.H % 4 == 0 && !a.y_f32 && !a.y8 && a.y && (long)a.M * (a.H / 4) * (a.W / 16) * groups >= 4096 &&
            (size_t)a.M * a.H * a.W * a.xcs * 2 < (1ull << 31) && (size_t)a.M * a.H * a.W * a.ycs * 2 < (1ull << 31))
            return w2c_option(W2C_OPT_REGW_FORM) == 2 ? 52 : 50;
        if (a.Cin == 64 && Cout % 64 == 0 && tiles * (Cout / 64) >= 64) return 38;
        if (a.Cin == 128 && Cout % 128 == 0 && tiles * (Cout / 128) >= 256) return 30;
        if (Cout % 64 == 0 && tiles * (Cout / 64) >= 64) return 36;
    }
    if (a.stride == 2 && a.ks == 3 && a.Ho % 8 == 0 && a.Wo % 16 == 0 && Cout % 64 == 0 && !(a.H & 1) && !(a.W & 1) && !a.ws &&
        !a.res && !w2c_option(W2C_OPT_NO_S2PATCH))
        return 60;
    // stride-2 3x3: the 128x64 tile beats 128x128 at every cfg-2 shape (tools/bench_s2_block.py: 46.1 / 36.9 / 36.5 us vs
    // 47.7 / 42.8 / 38.1 us) -- twice the workgroups for a kernel whose K-step is dominated by the 9x re-gather of its rows
    if (a.stride == 2 && a.ks == 3 && Cout % 64 == 0 && (rows / 128) * (Cout / 64) * groups >= 512) return 3;
    if (Cout % 128 == 0 && (rows / 128) * (Cout / 128) * groups >= 512) return 0;
    if (Cout % 64 == 0 && (rows / 128) * (Cout / 64) * groups >= 512) return 3;
    if (Cout % 64 == 0) return 6;
    return 8;
}

int fill_args(ConvArgs& a, const void* x, int M, int H, int W, int Cin, int x_cstride,
              const void* w, int Cout, int ksize, int stride, int groups,
              const float* scale, const float* shift, const uint16_t* residual, int relu,
              void* y, int y_cstride, int y_is_f32, const void* zero_page, long long y_group_stride, bool f8 = false,
              uint8_t* y8 = nullptr, int y8_cstride = 0, float y8_scale = 1.f) {
    const int es = f8 ? 1 : 2, ck = f8 ? 128 : 64;
    if (!x || !w || !scale || !shift || (!y && !y8) || !zero_page) return W2C_E_ARG;
    if (M <= 0 || H <= 0 || W <= 0 || groups <= 0) return W2C_E_ARG;
    if (Cin <= 0 || (Cin % ck) != 0 || Cout <= 0 || (Cout % 32) != 0) return W2C_E_ARG;
    if (!((ksize == 3) || (ksize == 1)) || !((stride == 1) || (stride == 2))) return W2C_E_ARG;
    if (y_group_stride == 0) y_group_stride = Cout;
    if (x_cstride < groups * Cin || ((y || residual) && y_cstride < (y_group_stride == Cout ? groups : 1) * Cout)) return W2C_E_ARG;
    if ((x_cstride * es % 16) != 0 || ((y || residual) && (y_cstride % 8) != 0) || (y_group_stride % 8) != 0) return W2C_E_ARG;   // 16-byte vector access
    if (y8 && (y8_cstride < groups * Cout || (y8_cstride % 8) != 0 || !(y8_scale > 0.f))) return W2C_E_ARG;
    a.x = reinterpret_cast<const uint16_t*>(x); a.w = reinterpret_cast<const uint16_t*>(w);
    a.scale = scale; a.shift = shift; a.res = residual; a.y = y;
    a.zeros = reinterpret_cast<const uint16_t*>(zero_page);
    a.M = M; a.H = H; a.W = W; a.Cin = Cin; a.xcs = x_cstride;
    a.ks = ksize; a.stride = stride; a.pad = ksize == 3 ? 1 : 0;
    a.Ho = (H + 2 * a.pad - ksize) / stride + 1;
    a.Wo = (W + 2 * a.pad - ksize) / stride + 1;
    a.Cout = Cout; a.ycs = y_cstride; a.relu = relu; a.y_f32 = y_is_f32;      // ycs is also the residual's pixel stride
    // the kernels address x and w through buffer descriptors with 32-bit byte offsets (extent < 2 GiB; an offset of
    // 0x80000000 is the "padding -> zeros" marker) and count output rows in an int: refuse what would overflow
    // instead of silently reading zeros.  Callers chunk M (ops.conv_igemm does).
    if ((size_t)M * H * W * x_cstride * es >= (1ull << 31) || (size_t)Cout * ksize * ksize * Cin * es >= (1ull << 31) ||
        (size_t)M * a.Ho * a.Wo >= (1ull << 29))
        return W2C_E_ARG;
    a.rows = M * a.Ho * a.Wo;
    a.mg_hw = w2c_magic((unsigned)(a.Ho * a.Wo));
    a.mg_wo = w2c_magic((unsigned)a.Wo);
    a.mg_ntn = a.mg_qn = a.mg_tx = a.mg_txy = 0;
    a.dbg = nullptr;
    a.span = g_span_next;            // (w2c_debug_conv_span: the next conv call of this thread records its launch span)
    g_span_next = nullptr;
    a.ws = nullptr;
    a.n_split = 1;
    a.ygs = y_group_stride;
    a.y8 = y8; a.y8cs = y8_cstride; a.q8 = y8 ? 1.f / y8_scale : 1.f;
    a.w2 = nullptr; a.scale2 = nullptr; a.shift2 = nullptr; a.y2 = nullptr; a.y2cs = 0; a.y2gs = 0;
    return W2C_OK;
}

}  // namespace

extern "C" int w2c_conv_igemm_bf16(const uint16_t* x, int M, int H, int W, int Cin, int x_cstride,
                                   const uint16_t* w, int Cout, int ksize, int stride, int groups,
                                   const float* scale, const float* shift,
                                   const uint16_t* residual, int relu,
                                   void* y, int y_cstride, int y_is_f32,
                                   const void* zero_page, long long y_group_stride, w2c_stream_t stream) {
    w2c_clear_error();
    ConvArgs a;
    int rc = fill_args(a, x, M, H, W, Cin, x_cstride, w, Cout, ksize, stride, groups, scale, shift, residual, relu,
                       y, y_cstride, y_is_f32, zero_page, y_group_stride);
    if (rc != W2C_OK) return rc;
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    return launch_variant(pick_variant(a, groups), a, groups, s);
}

// [groups][Cout][9*Cin] (K = tap * Cin + c) -> fragment order of conv_wreg.inl:
// [groups][Cout/32][chunk*9 + tap][k slice 0..3][half][channel % 32][8 elements]; one thread per 16 bytes
__global__ void pack_wfrag_kernel(const uint4* __restrict__ w, uint4* __restrict__ out, int Cout, int Cin, long n16) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n16) return;
    // destination index -> (g, nb, cc, tap, kk, half, c32)
    long r = i;
    const int c32 = (int)(r & 31); r >>= 5;
    const int half = (int)(r & 1); r >>= 1;
    const int kk = (int)(r & 3); r >>= 2;
    const int tap = (int)(r % 9); r /= 9;
    const int nch = Cin >> 6;
    const int cc = (int)(r % nch); r /= nch;
    const int nbn = Cout >> 5;
    const int nb = (int)(r % nbn);
    const long g = r / nbn;
    const long src = ((g * Cout + nb * 32 + c32) * 9 + tap) * (long)Cin + cc * 64 + kk * 16 + half * 8;    // elements
    out[i] = w[src >> 3];
}

extern "C" int w2c_pack_wfrag_bf16(const uint16_t* w, uint16_t* wfrag, int groups, int Cout, int Cin, w2c_stream_t stream) {
    if (!w || !wfrag || groups <= 0 || Cout <= 0 || Cin <= 0 || (Cout % 32) != 0 || (Cin % 64) != 0) return W2C_E_ARG;
    w2c_clear_error();
    const long n16 = (long)groups * Cout * 9 * Cin / 8;
    hipLaunchKernelGGL(pack_wfrag_kernel, dim3((unsigned)((n16 + 255) / 256)), dim3(256), 0, reinterpret_cast<hipStream_t>(stream),
                       reinterpret_cast<const uint4*>(w), reinterpret_cast<uint4*>(wfrag), Cout, Cin, n16);
    return w2c_launch_status();
}

// Which layers go to the weights-to-registers kernel, from the cfg-2 layer sweeps (profiles/r03_wreg_kernel.txt): the deep
// layers (Cin >= 256: layer3, layer4, policy conv1 / conv2, decoder conv0) gain 8-30 %; layer2 (Cin = 128: 18 K-steps) does not.
// A function of the layer geometry only.
static int wreg_form(int H, int W, int Cin, int Cout) {
    if (H <= 0 || W <= 0 || (H % 8) != 0 || (W % 16) != 0 || (Cin % 64) != 0 || (Cout % 64) != 0) return 0;
    // layer1 (Cin = Cout = 64): the two-waves-per-SIMD register-resident kernel (conv_regh.inl), bit-identical to the ring kernels
    if (Cin == 64 && Cout == 64) return w2c_option(W2C_OPT_L1_FORM) == 54 ? 54 : 0;
    const int mincin = w2c_option(W2C_OPT_WREG_MINCIN);
    // (layer2, Cin = 128, stays on the ring kernel: with K = 1152 the K-group reduction of this kernel family costs as much as the
    //  K loop -- also in a persistent form that prefetches across tiles, profiles/r06_conv_experiments.txt section 3)
    if (mincin <= 0 || Cin < mincin) return 0;
    const int forced = w2c_option(W2C_OPT_WREG_FORM);
    if (forced == 80 && (Cout % 128) != 0) return 93;
    if (forced) return forced;
    return 93;                                       // 1 channel block x 4 K groups, weights 4 K-steps ahead
}
extern "C" int w2c_conv3x3_wreg_supported(int H, int W, int Cin, int Cout) { return wreg_form(H, W, Cin, Cout) != 0; }
// launches of the default form with fewer 128-pixel x 64-channel workgroups than this take the 32-channel-per-wave form (twice the
// workgroups, same K groups and reduction order: the same bits, so the choice may depend on the image count)
// (round 6, tools/r06/ab.sh, one rank's share of cfg 3 = 8 images, ms per forward: 0.5608 / 0.5610 -> 0.5473 / 0.5488 with the threshold at
//  128 workgroups -- its layer4 / squeezer / policy conv launches are 64-128 workgroups of ~20 us on a 256-CU chip; 0.5537 / 0.5551 at 256,
//  where layer3's 256-workgroup launches switch too; the sharded step of that rank 0.6129 / 0.6137 -> 0.5991 / 0.5990; cfg 4's rank 0.8966
//  -> 0.8791; cfg 2, whose smallest launch of this family is 160 workgroups, is untouched.)
constexpr long kWregSmallLaunch = 128;
static bool wreg_small(int M, int H, int W, int Cout, int groups) {
    return (long)M * (H / 8) * (W / 16) * (Cout / 64) * groups <= kWregSmallLaunch;
}

extern "C" int w2c_conv3x3_wreg_bf16(const uint16_t* x, int M, int H, int W, int Cin, int x_cstride,
                                     const uint16_t* wfrag, int Cout, int groups,
                                     const float* scale, const float* shift, const uint16_t* residual, int relu,
                                     uint16_t* y, int y_cstride, long long y_group_stride, int form, w2c_stream_t stream) {
    ConvArgs a;
    // (the zero page is unused by this kernel: out-of-image halo pixels are out-of-range buffer offsets)
    const int rc = fill_args(a, x, M, H, W, Cin, x_cstride, wfrag, Cout, 3, 1, groups, scale, shift, residual, relu, y, y_cstride, 0,
                             /*zero_page=*/x, y_group_stride);
    if (rc != W2C_OK) return rc;
    // 32-bit element offsets in the epilogue: the whole output window (all groups' slabs) must lie within 2^31 elements of y
    if ((size_t)M * H * W * y_cstride * 2 >= (1ull << 31) || y_group_stride < 0 ||
        (unsigned long long)(groups - 1) * (unsigned long long)(y_group_stride ? y_group_stride : Cout) + (size_t)M * H * W * y_cstride >= (1ull << 31))
        return W2C_E_ARG;
    if (form == 0) {
        form = wreg_form(H, W, Cin, Cout);
        if (form == 93 && wreg_small(M, H, W, Cout, groups)) form = 95;
    }
    if (form != 80 && form != 81 && form != 83 && form != 93 && form != 94 && form != 95 && form != 54) return W2C_E_ARG;
    w2c_clear_error();
    return launch_variant(form, a, groups, reinterpret_cast<hipStream_t>(stream));
}

// The default form (93) with an f32 output tensor and no residual: the decoder's first conv on every agent's value map (U = conv0 without
// bias / ReLU, fused after the communication graph by linearity).  Same K groups and reduction order as w2c_conv3x3_wreg_bf16: the f32
// values are what that entry point rounds to bf16.
extern "C" int w2c_conv3x3_wreg_f32out(const uint16_t* x, int M, int H, int W, int Cin, int x_cstride,
                                       const uint16_t* wfrag, int Cout, int groups,
                                       const float* scale, const float* shift, int relu,
                                       float* y, int y_cstride, long long y_group_stride, w2c_stream_t stream) {
    ConvArgs a;
    const int rc = fill_args(a, x, M, H, W, Cin, x_cstride, wfrag, Cout, 3, 1, groups, scale, shift, nullptr, relu, y, y_cstride, 1,
                             /*zero_page=*/x, y_group_stride);
    if (rc != W2C_OK) return rc;
    if (wreg_form(H, W, Cin, Cout) != 93) return W2C_E_ARG;
    if ((size_t)M * H * W * y_cstride >= (1ull << 31) || y_group_stride < 0 ||
        (unsigned long long)(groups - 1) * (unsigned long long)(y_group_stride ? y_group_stride : Cout) + (size_t)M * H * W * y_cstride >= (1ull << 31))
        return W2C_E_ARG;
    w2c_clear_error();
    return launch_variant(wreg_small(M, H, W, Cout, groups) ? 195 : 193, a, groups, reinterpret_cast<hipStream_t>(stream));
}

extern "C" long long w2c_conv_splitk_workspace_bytes(int M, int H, int W, int Cin, int Cout, int ksize, int stride,
                                                     int groups, int ksplit) {
    if (M <= 0 || H <= 0 || W <= 0 || Cin <= 0 || Cin % 64 || Cout <= 0 || Cout % 32 || groups <= 0) return -1;
    if (!((ksize == 3) || (ksize == 1)) || !((stride == 1) || (stride == 2))) return -1;
    ConvArgs a;
    a.ks = ksize; a.Cin = Cin; a.Cout = Cout;
    const int pad = ksize == 3 ? 1 : 0;
    a.Ho = (H + 2 * pad - ksize) / stride + 1;
    a.Wo = (W + 2 * pad - ksize) / stride + 1;
    a.rows = M * a.Ho * a.Wo;
    const SplitPlan sp = plan_splitk(a, groups, ksplit);
    if (sp.ksplit <= 1) return 0;
    return (long long)sp.tiles * sp.ksplit * sp.bm * sp.bn * 4;
}

extern "C" int w2c_conv_igemm_bf16_splitk(const uint16_t* x, int M, int H, int W, int Cin, int x_cstride,
                                          const uint16_t* w, int Cout, int ksize, int stride, int groups,
                                          const float* scale, const float* shift,
                                          const uint16_t* residual, int relu,
                                          void* y, int y_cstride, int y_is_f32,
                                          const void* zero_page, int ksplit,
                                          void* workspace, long long workspace_bytes, long long y_group_stride, w2c_stream_t stream) {
    w2c_clear_error();
    ConvArgs a;
    int rc = fill_args(a, x, M, H, W, Cin, x_cstride, w, Cout, ksize, stride, groups, scale, shift, residual, relu,
                       y, y_cstride, y_is_f32, zero_page, y_group_stride);
    if (rc != W2C_OK) return rc;
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    const SplitPlan sp = plan_splitk(a, groups, ksplit);
    if (sp.ksplit <= 1) return launch_variant(pick_variant(a, groups), a, groups, s);
    // up to 12 splits: the splits are the waves of one workgroup and the partial tiles meet in LDS (same bits, one launch, no workspace)
    if (sp.ksplit <= 12 && a.Cin % 64 == 0 && a.Cout % 32 == 0 && w2c_option(W2C_OPT_INWG_SPLITK))
    {
        // 64-row tiles while the partial tiles fit the 80 KB budget (<= 8 splits), 32-row tiles above; a wave with <= 3 K-steps
        // issues all of its loads up front.  (Measured, tools/bench_tail.py, us per conv inside a graph, two launches -> one:
        // policy conv3 15.8 -> 14.5, conv4 16.2 -> 13.8, conv5 11.1 -> 10.1, decoder's last conv 15.1 -> 10.4.  32-row tiles with 5
        // K-steps up front: 19.9 -- the bound is the bytes a CU can pull per second, ~47 GB/s per workgroup whatever the ring
        // depth, so halving the tile doubles the weight traffic and loses.)
        const int kt = a.ks * a.ks * (a.Cin / 64);
        const int per_wave = (kt + sp.ksplit - 1) / sp.ksplit;           // K-steps of the longest split
        if (sp.ksplit <= 8) return launch_conv_inwg_splitk<64, 0, 512>(a, groups, sp.ksplit, s);
        return per_wave <= 3 ? launch_conv_inwg_splitk<32, 3, 768>(a, groups, sp.ksplit, s)
                             : launch_conv_inwg_splitk<32, 0, 768>(a, groups, sp.ksplit, s);
    }
    const long long need = (long long)sp.tiles * sp.ksplit * sp.bm * sp.bn * 4;
    if (!workspace || workspace_bytes < need || (reinterpret_cast<uintptr_t>(workspace) & 15)) return W2C_E_ARG;
    a.ws = reinterpret_cast<float*>(workspace);
    return sp.bn == 64 ? launch_conv_splitk<64, 64, 2, 2>(a, groups, sp.ksplit, s)
                       : launch_conv_splitk<128, 32, 4, 1>(a, groups, sp.ksplit, s);
}

// Debug: next w2c_conv_igemm_bf16_variant call of THIS thread writes a workgroup timeline (4 x u64
// wall_clock64 stamps per workgroup: start, first tile landed, main loop done, end) to `buf`.
static thread_local unsigned long long* g_dbg_next = nullptr;
extern "C" int w2c_debug_conv_timeline(void* buf) {
    g_dbg_next = reinterpret_cast<unsigned long long*>(buf);
    return W2C_OK;
}

// Debug: one-thread kernel that writes the 100 MHz wall clock to slot[0] -- a time stamp in stream order (tools/chain_stamps.py:
// when does each launch chain of a captured forward start?).
// slot[32] = the shader-clock counter at the same instant: (d slot[32] / d slot[0]) x 100 MHz = the clock the chip held between two stamps
// (the counter is per XCD: the XCC id rides in the top 4 bits, a reader compares stamps of the same XCD only)
__global__ void stamp_kernel(unsigned long long* slot) {
#if defined(__HIP_DEVICE_COMPILE__)
    const unsigned long long xcc = __builtin_amdgcn_s_getreg((3 << 11) | 20) & 0xFu;       // hwreg(HW_REG_XCC_ID, 0, 4)
    slot[0] = wall_clock64();
    slot[32] = (xcc << 60) | ((unsigned long long)clock64() & ((1ull << 60) - 1));
#endif
}
extern "C" int w2c_debug_stamp(void* slot, w2c_stream_t stream) {
    w2c_clear_error();
    stamp_kernel<<<1, 1, 0, reinterpret_cast<hipStream_t>(stream)>>>(reinterpret_cast<unsigned long long*>(slot));
    return w2c_launch_status();
}

extern "C" int w2c_conv_igemm_bf16_variant(const uint16_t* x, int M, int H, int W, int Cin, int x_cstride,
                                           const uint16_t* w, int Cout, int ksize, int stride, int groups,
                                           const float* scale, const float* shift,
                                           const uint16_t* residual, int relu,
                                           void* y, int y_cstride, int y_is_f32,
                                           const void* zero_page, int variant, long long y_group_stride, w2c_stream_t stream) {
    w2c_clear_error();
    ConvArgs a;
    int rc = fill_args(a, x, M, H, W, Cin, x_cstride, w, Cout, ksize, stride, groups, scale, shift, residual, relu,
                       y, y_cstride, y_is_f32, zero_page, y_group_stride);
    if (rc != W2C_OK) return rc;
    a.dbg = g_dbg_next;
    g_dbg_next = nullptr;
    return launch_variant(variant, a, groups, reinterpret_cast<hipStream_t>(stream));
}

extern "C" int w2c_conv_igemm_fp8(const void* x, int x_is_fp8, int M, int H, int W, int Cin, int x_cstride,
                                  const void* w, int Cout, int ksize, int stride, int groups,
                                  const float* scale, const float* shift,
                                  const uint16_t* residual, int relu,
                                  uint16_t* y_bf16, int y_cstride, long long y_group_stride,
                                  uint8_t* y_fp8, int y8_cstride, float y8_scale,
                                  const void* zero_page, int variant, w2c_stream_t stream) {
    w2c_clear_error();
    ConvArgs a;
    int rc = fill_args(a, x, M, H, W, Cin, x_cstride, w, Cout, ksize, stride, groups, scale, shift, residual, relu,
                       y_bf16, y_cstride, 0, zero_page, y_group_stride, x_is_fp8 != 0, y_fp8, y8_cstride, y8_scale);
    if (rc != W2C_OK) return rc;
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    if (x_is_fp8) return launch_variant_f8(variant >= 0 ? variant : pick_variant_f8(a, groups), a, groups, s);
    return launch_variant(variant >= 0 ? variant : pick_variant(a, groups), a, groups, s);
}

extern "C" int w2c_conv_s2_block(const void* x, int x_is_fp8, int M, int H, int W, int Cin, int x_cstride,
                                 const void* w3, const float* scale3, const float* shift3,
                                 const void* w1, const float* scale1, const float* shift1,
                                 int Cout, int groups,
                                 uint16_t* t_bf16, int t_cstride, uint8_t* t_fp8, int t8_cstride, float t8_scale,
                                 uint16_t* idt_bf16, int idt_cstride,
                                 const void* zero_page, int variant, w2c_stream_t stream) {
    w2c_clear_error();
    ConvArgs a;
    int rc = fill_args(a, x, M, H, W, Cin, x_cstride, w3, Cout, 3, 2, groups, scale3, shift3, nullptr, 1,
                       t_bf16, t_cstride, 0, zero_page, 0, x_is_fp8 != 0, t_fp8, t8_cstride, t8_scale);
    if (rc != W2C_OK) return rc;
    if (!w1 || !scale1 || !shift1 || !idt_bf16 || idt_cstride < groups * Cout || (idt_cstride % 8) != 0) return W2C_E_ARG;
    a.w2 = reinterpret_cast<const uint16_t*>(w1); a.scale2 = scale1; a.shift2 = shift1; a.y2 = idt_bf16; a.y2cs = idt_cstride;
    a.dbg = g_dbg_next;
    g_dbg_next = nullptr;
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    // maps that tile into 8 x 16 output pixels go to the polyphase halo-patch kernel (variant 60), the rest to the generic
    // DUAL kernel; both walk K in the same order, so the choice never shows in the bits
    const bool patch_ok = a.Ho % 8 == 0 && a.Wo % 16 == 0 && Cout % 64 == 0 && !(H & 1) && !(W & 1);
    if (variant == 60 || (variant < 0 && patch_ok && !w2c_option(W2C_OPT_NO_S2PATCH)))
        return x_is_fp8 ? launch_s2patch<64, 3, true>(a, groups, s) : launch_s2patch<64, 3, false>(a, groups, s);
    if (variant == 61) return x_is_fp8 ? launch_s2patch<128, 2, true>(a, groups, s) : launch_s2patch<128, 2, false>(a, groups, s);
    if (variant == 62) return x_is_fp8 ? launch_s2patch<64, 2, true>(a, groups, s) : launch_s2patch<64, 2, false>(a, groups, s);
    return x_is_fp8 ? launch_dual<true>(a, groups, variant, s) : launch_dual<false>(a, groups, variant, s);
}

// Stride-2 block front on the weights-to-registers structure (conv_s2wreg.inl).  A function of the layer geometry only.
// NOT offered by default (W2C_S2WREG_FORM=1 turns it on): alone it beats the polyphase ring kernel on layer3.0 / layer4.0 (cfg 2, one
// group: 26.0 vs 29.5, 25.0 vs 29.5 us; layer2.0 32.1 vs 32.6) but its workgroups live as long (prologue 2.5-3 us + reduction +
// downsample pass + stores around a main loop that is 1.3-1.7x faster), so beside the other trunk chain the forward does not move
// (1.0484 vs 1.0485 ms, 3 interleaved pairs; rank shapes of cfg 3 / cfg 4 the same) -- profiles/r04_s2_front_wreg.txt.
// Round 4, third session: DEFAULT (form 1) since the persistent two-group launches went to one workgroup per CU: with the
// front of the forward burning less power the faster main loop shows -- 4 interleaved pairs, bench.py --steps 100: 1.0222 1.0074 1.0164
// 1.0187 (ring kernel) vs 1.0054 1.0049 1.0065 1.0009 ms (form 1); layer2.0 (Cin = 64) goes to conv_s2regh.inl before this is asked.
static int s2wreg_form(int H, int W, int Cin, int Cout) {
    if (H <= 0 || W <= 0 || (H & 1) || (W & 1) || ((H / 2) % 8) != 0 || ((W / 2) % 16) != 0 || (Cin % 64) != 0 || Cin > 256 || (Cout % 64) != 0)
        return 0;
    const int f = w2c_option(W2C_OPT_S2WREG_FORM);
    if (f == 3 && (Cout % 128) != 0) return 1;
    return (f >= 1 && f <= 4) ? f : 0;
}
extern "C" int w2c_conv_s2_block_wreg_supported(int H, int W, int Cin, int Cout) { return s2wreg_form(H, W, Cin, Cout) != 0; }

extern "C" int w2c_conv_s2_block_wreg(const uint16_t* x, int M, int H, int W, int Cin, int x_cstride,
                                      const uint16_t* w3frag, const float* scale3, const float* shift3,
                                      const uint16_t* w1frag, const float* scale1, const float* shift1,
                                      int Cout, int groups, uint16_t* t_bf16, int t_cstride, uint16_t* idt_bf16, int idt_cstride,
                                      int form, w2c_stream_t stream) {
    w2c_clear_error();
    ConvArgs a;
    const int rc = fill_args(a, x, M, H, W, Cin, x_cstride, w3frag, Cout, 3, 2, groups, scale3, shift3, nullptr, 1, t_bf16, t_cstride, 0,
                             /*zero_page=*/x, 0);
    if (rc != W2C_OK) return rc;
    if (!t_bf16 || !w1frag || !scale1 || !shift1 || !idt_bf16 || idt_cstride < groups * Cout || (idt_cstride % 8) != 0) return W2C_E_ARG;
    // 32-bit element offsets in the epilogue
    if ((size_t)M * a.Ho * a.Wo * t_cstride >= (1ull << 31) || (size_t)M * a.Ho * a.Wo * idt_cstride >= (1ull << 31)) return W2C_E_ARG;
    a.w2 = w1frag; a.scale2 = scale1; a.shift2 = shift1; a.y2 = idt_bf16; a.y2cs = idt_cstride;
    a.dbg = g_dbg_next;
    g_dbg_next = nullptr;
    if (form == 0) form = s2wreg_form(H, W, Cin, Cout);
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    if (form >= 16) { a.n_split = form >> 4; form &= 15; }      // (debug: which point the timeline's third stamp marks)
    switch (form) {
        case 1: return launch_s2wreg<1, 4, 4>(a, groups, s);
        case 2: return launch_s2wreg<1, 4, 2>(a, groups, s);
        case 3: return launch_s2wreg<2, 2, 2>(a, groups, s);
        case 4: return launch_s2wreg<1, 2, 2>(a, groups, s);
        default: return W2C_E_ARG;
    }
}

// Front of the first stride-2 BasicBlock (Cin = 64 -> Cout = 128 per group) on the persistent weights-stationary kernel (conv_s2regh.inl).
// Offered from the layer geometry only (never from M or the group count: a sharded batch takes the same kernel as the whole one).
extern "C" int w2c_conv_s2_front_c64_supported(int H, int W, int Cin, int Cout) {
    return w2c_option(W2C_OPT_S2REGH) != 0 && H > 0 && W > 0 && !(H & 1) && !(W & 1) && ((H / 2) % 8) == 0 && ((W / 2) % 8) == 0 && Cin == 64 &&
           Cout == 128;
}
extern "C" int w2c_conv_s2_front_c64(const uint16_t* x, int M, int H, int W, int x_cstride,
                                     const uint16_t* w3frag, const float* scale3, const float* shift3,
                                     const uint16_t* w1frag, const float* scale1, const float* shift1, int groups,
                                     uint16_t* t_bf16, int t_cstride, long long t_group_stride,
                                     uint16_t* idt_bf16, int idt_cstride, long long idt_group_stride, w2c_stream_t stream) {
    w2c_clear_error();
    ConvArgs a;
    if (t_group_stride <= 0 || idt_group_stride <= 0 || (idt_group_stride % 8) != 0) return W2C_E_ARG;
    const int rc = fill_args(a, x, M, H, W, 64, x_cstride, w3frag, 128, 3, 2, groups, scale3, shift3, nullptr, 1, t_bf16, t_cstride, 0,
                             /*zero_page=*/x, t_group_stride);
    if (rc != W2C_OK) return rc;
    if (!t_bf16 || !w1frag || !scale1 || !shift1 || !idt_bf16 || (idt_cstride % 8) != 0 || idt_cstride < (idt_group_stride == 128 ? groups : 1) * 128)
        return W2C_E_ARG;
    a.w2 = w1frag; a.scale2 = scale1; a.shift2 = shift1; a.y2 = idt_bf16; a.y2cs = idt_cstride; a.y2gs = idt_group_stride;
    return launch_s2regh(a, groups, reinterpret_cast<hipStream_t>(stream));
}

// One wave: C[32][32] (f32, row-major) = A[32][64] * B[32][64]^T with e4m3 operands through the MX-scaled MFMA with unit
// block scales -- the primitive the fp8 conv kernels are built on (unit test of the fragment convention).
__global__ void mx_mfma_probe_kernel(const uint8_t* A, const uint8_t* B, float* C) {
#if defined(__HIP_DEVICE_COMPILE__)
    const int lane = threadIdx.x, l31 = lane & 31, lhi = lane >> 5;
    const u32x4_t* ap = reinterpret_cast<const u32x4_t*>(A + l31 * 64 + lhi * 32);
    const u32x4_t* bp = reinterpret_cast<const u32x4_t*>(B + l31 * 64 + lhi * 32);
    f32x16_t acc;
    for (int e = 0; e < 16; ++e) acc[e] = 0.f;
    acc = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(cat_i32x8(ap[0], ap[1]), cat_i32x8(bp[0], bp[1]), acc, 0, 0, 0,
                                                          0x7F7F7F7F, 0, 0x7F7F7F7F);
    for (int e = 0; e < 16; ++e) C[((e & 3) + 8 * (e >> 2) + 4 * lhi) * 32 + l31] = acc[e];    // D[row of A][row of B]
#endif
}
__global__ void fp8_pack_probe_kernel(const float* x, uint8_t* y, int n) {
    const int i = (blockIdx.x * 256 + threadIdx.x) * 4;
    if (i + 3 < n) *reinterpret_cast<uint32_t*>(y + i) = pack_fp8x4(x[i], x[i + 1], x[i + 2], x[i + 3]);
}
extern "C" int w2c_debug_mx_mfma(const uint8_t* a, const uint8_t* b, float* c, w2c_stream_t stream) {
    w2c_clear_error();
    if (!a || !b || !c) return W2C_E_ARG;
    hipLaunchKernelGGL(mx_mfma_probe_kernel, dim3(1), dim3(64), 0, reinterpret_cast<hipStream_t>(stream), a, b, c);
    return w2c_launch_status();
}
extern "C" int w2c_debug_fp8_pack(const float* x, uint8_t* y, int n, w2c_stream_t stream) {
    w2c_clear_error();
    if (!x || !y || n <= 0 || (n & 3)) return W2C_E_ARG;
    hipLaunchKernelGGL(fp8_pack_probe_kernel, dim3((n / 4 + 255) / 256), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), x, y, n);
    return w2c_launch_status();
}
