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
//   K      = ksize*ksize*Cin, walked as (tap, 64-channel chunk): NHWC makes every K-step of a
//            row one contiguous 128-byte run, so the im2col gather is a per-lane ADDRESS, never
//            a materialised matrix.
// Per workgroup (256 threads = 4 waves, one per SIMD): a BM x BN tile, K-step 64.
//   * operands go HBM/L2 -> LDS by LDS-DMA (global_load_lds_dwordx4: 16 B per lane, the LDS
//     image is lane-linear, so the bank swizzle is applied to the per-lane SOURCE address and
//     again on the ds_read -- cdna_hip_programming.md rule 21).  Padded taps read a zero page.
//   * two LDS stages; the DMA of K-step t+1 is in flight while step t's MFMAs run; one barrier
//     per K-step.
//   * v_mfma_f32_32x32x16_bf16, f32 accumulate; A = pixels, B = channels.
//   * epilogue: per-channel scale/shift (eval BN or bias) in registers, tile staged through LDS as
//     f32 so the residual read and the bf16/f32 store are 16-byte coalesced along channels.
// LDS swizzle: rows are 128 B (64 bf16); chunk c (16 B) of row r lives at position
// c ^ ((r >> 1) & 7).  ds_read_b128 is serviced in 16-lane groups over a 256-B bank row (= two
// of our rows); this XOR makes each group's 16 (row parity, position) pairs distinct => no
// bank conflicts (MI355X_MICROARCH.md LDS table).
#include "w2c_common.h"

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
    int cin_tiles;   // Cin / 64
    int ktiles;      // ks*ks*cin_tiles
    int ntm, ntn;    // tiles along rows / cout
};

constexpr int BK = 64;                 // K-step (bf16 elements) = 128 B per row
constexpr int ROWB = BK * 2;           // bytes per LDS row

template <int BM, int BN, int WM, int WN>
__global__ __launch_bounds__(256) void conv_igemm_kernel(ConvArgs p) {
    constexpr int WTM = BM / WM, WTN = BN / WN;        // wave tile
    constexpr int MI = WTM / 32, NI = WTN / 32;        // 32x32 MFMA tiles per wave
    constexpr int A_INSTR = BM / 32;                   // LDS-DMA instructions per wave per K-step (8 rows each)
    constexpr int B_INSTR = BN / 32;
    constexpr int A_BYTES = BM * ROWB, B_BYTES = BN * ROWB;
    constexpr int STAGE_BYTES = A_BYTES + B_BYTES;
    constexpr int CLD = BN + 4;                        // f32 epilogue tile leading dim
    static_assert(WM * WN == 4, "4 waves");
    static_assert(2 * STAGE_BYTES >= 0, "");

    extern __shared__ __attribute__((aligned(16))) char smem[];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int g = blockIdx.y;

    // ---- tile id (XCD-aware): n fastest so the column tiles of one row panel share an L2 ----
    const int tile = xcd_remap(blockIdx.x, p.ntm * p.ntn);
    const int tm = tile / p.ntn, tn = tile - tm * p.ntn;
    const int m0 = tm * BM, n0 = tn * BN;

    const uint16_t* xg = p.x + (size_t)g * p.Cin;                      // group's channel slice
    const uint16_t* wg = p.w + (size_t)g * p.Cout * (p.ktiles * BK);   // group's weights
    const int Ktot = p.ktiles * BK;

    // ---- per-thread gather state for its A rows (fixed for the whole K loop) ----
    const int lrow = lane >> 3;            // row within an 8-row DMA group
    const int lpos = lane & 7;             // 16-B position within the 128-B LDS row
    int a_iy0[A_INSTR], a_ix0[A_INSTR], a_chunk[A_INSTR];
    long a_img[A_INSTR];                   // element offset of image start, or -1 when the row is past the end
#pragma unroll
    for (int j = 0; j < A_INSTR; ++j) {
        const int r = (wave + 4 * j) * 8 + lrow;
        const int gr = m0 + r;
        a_chunk[j] = lpos ^ ((r >> 1) & 7);
        if (gr < p.rows) {
            const int hw = p.Ho * p.Wo;
            const int m = gr / hw;
            const int rem = gr - m * hw;
            const int oy = rem / p.Wo, ox = rem - oy * p.Wo;
            a_iy0[j] = oy * p.stride - p.pad;
            a_ix0[j] = ox * p.stride - p.pad;
            a_img[j] = (long)m * p.H * p.W;
        } else {
            a_iy0[j] = 0; a_ix0[j] = 0; a_img[j] = -1;
        }
    }
    const uint16_t* b_src[B_INSTR];
#pragma unroll
    for (int j = 0; j < B_INSTR; ++j) {
        const int n = (wave + 4 * j) * 8 + lrow;
        const int chunk = lpos ^ ((n >> 1) & 7);
        b_src[j] = wg + (size_t)(n0 + n) * Ktot + chunk * 8;
    }

    // K-step cursor (wave-uniform): tap (ky,kx) and channel chunk c0
    int st_ky = 0, st_kx = 0, st_ct = 0, st_kt = 0;

    auto stage = [&](int buf) {
        char* As = smem + buf * STAGE_BYTES;
        char* Bs = As + A_BYTES;
        const int c0 = st_ct * BK;
#pragma unroll
        for (int j = 0; j < A_INSTR; ++j) {
            const int iy = a_iy0[j] + st_ky, ix = a_ix0[j] + st_kx;
            const bool ok = (a_img[j] >= 0) & (iy >= 0) & (iy < p.H) & (ix >= 0) & (ix < p.W);
            const uint16_t* src = ok ? xg + (size_t)(a_img[j] + (long)iy * p.W + ix) * p.xcs + c0 + a_chunk[j] * 8
                                     : p.zeros + a_chunk[j] * 8;
            __builtin_amdgcn_global_load_lds(W2C_GPTR(src), W2C_LPTR(As + (wave + 4 * j) * 1024), 16, 0, 0);
        }
#pragma unroll
        for (int j = 0; j < B_INSTR; ++j) {
            __builtin_amdgcn_global_load_lds(W2C_GPTR(b_src[j] + (size_t)st_kt * BK),
                                             W2C_LPTR(Bs + (wave + 4 * j) * 1024), 16, 0, 0);
        }
        // advance cursor
        ++st_kt;
        if (++st_ct == p.cin_tiles) {
            st_ct = 0;
            if (++st_kx == p.ks) { st_kx = 0; ++st_ky; }
        }
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
    const int swz = (l31 >> 1) & 7;        // (row >> 1) & 7 for every row this lane reads (tile offsets are multiples of 32)

    auto compute = [&](int buf) {
        const char* As = smem + buf * STAGE_BYTES;
        const char* Bs = As + A_BYTES;
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) {               // four K=16 sub-steps
            const int pos = ((kk * 2 + lhi) ^ swz) << 4;
            bf16x8_t a[MI], b[NI];
#pragma unroll
            for (int i = 0; i < MI; ++i)
                a[i] = *reinterpret_cast<const bf16x8_t*>(As + (wm * WTM + i * 32 + l31) * ROWB + pos);
#pragma unroll
            for (int j = 0; j < NI; ++j)
                b[j] = *reinterpret_cast<const bf16x8_t*>(Bs + (wn * WTN + j * 32 + l31) * ROWB + pos);
#pragma unroll
            for (int i = 0; i < MI; ++i)
#pragma unroll
                for (int j = 0; j < NI; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i], b[j], acc[i][j], 0, 0, 0);
        }
    };

    // ---- main loop: DMA of step t+1 overlaps the MFMAs of step t; one barrier per step ----
    stage(0);
    __syncthreads();           // (emits vmcnt(0): the DMA is a pending LDS write)
    int cur = 0;
    for (int t = 0; t < p.ktiles - 1; ++t) {
        stage(cur ^ 1);
        compute(cur);
        __syncthreads();
        cur ^= 1;
    }
    compute(cur);
    __syncthreads();           // every wave done reading the staging buffers; reuse LDS for the epilogue

    // ---- epilogue 1: scale/shift in registers, tile -> LDS as f32 [BM][CLD] ----
    float* Cs = reinterpret_cast<float*>(smem);
#pragma unroll
    for (int j = 0; j < NI; ++j) {
        const int nl = wn * WTN + j * 32 + l31;
        const float sc = p.scale[g * p.Cout + n0 + nl];
        const float sh = p.shift[g * p.Cout + n0 + nl];
#pragma unroll
        for (int i = 0; i < MI; ++i) {
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int ml = wm * WTM + i * 32 + (e & 3) + 8 * (e >> 2) + 4 * lhi;   // C/D row map of 32x32 MFMA
                Cs[ml * CLD + nl] = acc[i][j][e] * sc + sh;
            }
        }
    }
    __syncthreads();

    // ---- epilogue 2: coalesced (+residual) (+ReLU) store, 8 channels per thread per pass ----
    constexpr int CG = BN / 8;                     // 8-channel groups per row
    constexpr int PASSES = BM * CG / 256;
#pragma unroll
    for (int ps = 0; ps < PASSES; ++ps) {
        const int idx = ps * 256 + tid;
        const int r = idx / CG, cg = idx - r * CG;
        const int gr = m0 + r;
        if (gr >= p.rows) continue;
        const f32x4_t v0 = *reinterpret_cast<const f32x4_t*>(Cs + r * CLD + cg * 8);
        const f32x4_t v1 = *reinterpret_cast<const f32x4_t*>(Cs + r * CLD + cg * 8 + 4);
        float v[8] = {v0[0], v0[1], v0[2], v0[3], v1[0], v1[1], v1[2], v1[3]};
        const size_t off = (size_t)gr * p.ycs + (size_t)g * p.Cout + n0 + cg * 8;
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
}

template <int BM, int BN, int WM, int WN>
constexpr int conv_lds_bytes() {
    constexpr int stage2 = 2 * (BM + BN) * ROWB;
    constexpr int epi = BM * (BN + 4) * 4;
    return stage2 > epi ? stage2 : epi;
}

template <int BM, int BN, int WM, int WN>
int launch_conv(ConvArgs& a, int groups, hipStream_t s) {
    a.ntm = (a.rows + BM - 1) / BM;
    a.ntn = a.Cout / BN;
    constexpr int lds = conv_lds_bytes<BM, BN, WM, WN>();
    // dynamic LDS above 64 KiB needs the attribute once per device; keep a per-device bit.
    static unsigned long long attr_mask = 0;   // benign race: every thread writes the same attribute
    int dev = 0;
    (void)hipGetDevice(&dev);
    if (!((attr_mask >> (dev & 63)) & 1ull)) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_igemm_kernel<BM, BN, WM, WN>),
                            hipFuncAttributeMaxDynamicSharedMemorySize, lds);
        attr_mask |= 1ull << (dev & 63);
    }
    dim3 grid(a.ntm * a.ntn, groups);
    hipLaunchKernelGGL((conv_igemm_kernel<BM, BN, WM, WN>), grid, dim3(256), lds, s, a);
    return w2c_launch_status();
}

}  // namespace

extern "C" int w2c_conv_igemm_bf16(const uint16_t* x, int M, int H, int W, int Cin, int x_cstride,
                                   const uint16_t* w, int Cout, int ksize, int stride, int groups,
                                   const float* scale, const float* shift,
                                   const uint16_t* residual, int relu,
                                   void* y, int y_cstride, int y_is_f32,
                                   const void* zero_page, w2c_stream_t stream) {
    w2c_clear_error();
    if (!x || !w || !scale || !shift || !y || !zero_page) return W2C_E_ARG;
    if (M <= 0 || H <= 0 || W <= 0 || groups <= 0) return W2C_E_ARG;
    if (Cin <= 0 || (Cin % 64) != 0 || Cout <= 0 || (Cout % 32) != 0) return W2C_E_ARG;
    if (!((ksize == 3) || (ksize == 1)) || !((stride == 1) || (stride == 2))) return W2C_E_ARG;
    if (x_cstride < groups * Cin || y_cstride < groups * Cout) return W2C_E_ARG;
    if ((x_cstride % 8) != 0 || (y_cstride % 8) != 0) return W2C_E_ARG;   // 16-byte vector access
    ConvArgs a;
    a.x = x; a.w = w; a.scale = scale; a.shift = shift; a.res = residual; a.y = y;
    a.zeros = reinterpret_cast<const uint16_t*>(zero_page);
    a.M = M; a.H = H; a.W = W; a.Cin = Cin; a.xcs = x_cstride;
    a.ks = ksize; a.stride = stride; a.pad = ksize == 3 ? 1 : 0;
    a.Ho = (H + 2 * a.pad - ksize) / stride + 1;
    a.Wo = (W + 2 * a.pad - ksize) / stride + 1;
    a.Cout = Cout; a.ycs = y_cstride; a.relu = relu; a.y_f32 = y_is_f32;
    a.rows = M * a.Ho * a.Wo;
    a.cin_tiles = Cin / 64;
    a.ktiles = ksize * ksize * a.cin_tiles;
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    // Tile choice: fill >= ~2 workgroups per CU (256 CUs) where the layer allows it.
    const long rows = a.rows;
    if (Cout % 128 == 0 && (rows / 128) * (Cout / 128) * groups >= 512) return launch_conv<128, 128, 2, 2>(a, groups, s);
    if (Cout % 64 == 0 && (rows / 128) * (Cout / 64) * groups >= 512) return launch_conv<128, 64, 2, 2>(a, groups, s);
    if (Cout % 64 == 0) return launch_conv<64, 64, 2, 2>(a, groups, s);
    return launch_conv<128, 32, 4, 1>(a, groups, s);
}
