// conv_wgrad.hip -- weight gradient of the path's 3x3 / 1x1 convolutions (the backward of
// trainer.py:669-673's loss.backward() for nn.Conv2d, models/utils.py:87-120, backbone.py:66-69,150-154).
//
//   dW[g][co][tap][ci] = sum over output pixels p of  dY[p][g*Cout + co] * X[p @ tap][g*Cin + ci]
//
// A GEMM whose reduction runs over PIXELS -- the outer dimension of the NHWC tensors -- while an MFMA operand lane wants
// 8 consecutive reduction elements.  Tiles are staged in LDS exactly as the forward kernel stages them ([pixel][64 channels],
// 128-byte rows, LDS-DMA with buffer addressing, out-of-image taps = out-of-range offsets -> zeros) and read back
// TRANSPOSED by gfx950's ds_read_b64_tr_b16: within a 16-lane group, lane 4r+q supplies the address of 4 contiguous
// channels of pixel r; lane l receives channel l of pixels 0..3 -- the k-contiguous fragment, no shuffles.
// Workgroup (4 waves, 2 x 2): one (group, 64 co, 64 ci) weight tile over one segment of the output pixels, all taps:
// per 64-pixel block the dY tile is staged once (its fragments stay in registers across the taps) and the X tile once per
// tap (double-buffered); 9 accumulator sets per wave.  Segments' partial tiles go to a workspace and are summed in segment
// order by a second launch (deterministic -- no float atomics).
#include "w2c_common.h"
#include "wgrad_common.h"
#include <cstdlib>

namespace {

struct WgradArgs {
    const uint16_t* x;
    const uint16_t* dy;
    float* ws;         // [nseg][G][Cout][taps][Cin]
    float* dw;         // [G][Cout][taps][Cin]
    int M, H, W, Cin, xcs;
    int Ho, Wo, Cout, ycs;
    int ks, stride, pad;
    int rows;          // M*Ho*Wo
    int nseg, blocks_per_seg;   // 128-pixel blocks per segment
    int nct_o, nct_i;  // 64-channel tiles along Cout / Cin
};

template <int TAPS>
__global__ __launch_bounds__(256) void conv_wgrad_kernel(WgradArgs p) {
#if defined(__HIP_DEVICE_COMPILE__)
    constexpr int BP = 128;                      // pixels per block (8 MFMA k-steps per tap: one barrier per 256 MFMA cycles per wave)
    constexpr int NJ = BP / 32;                  // DMA instructions per wave per tile (8 pixel rows each, 4 waves)
    __shared__ __attribute__((aligned(16))) char smem[BP * 128 * 3];      // dY tile | X tile x 2
    char* const Ys = smem;
    char* const Xs = smem + BP * 128;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1;
    const int g = blockIdx.z;
    const int seg = blockIdx.y;
    const int to = blockIdx.x / p.nct_i, ti = blockIdx.x - to * p.nct_i;
    const int lrow = lane >> 3, lpos = lane & 7;

    const char* xg = reinterpret_cast<const char*>(p.x) + ((size_t)g * p.Cin + ti * 64) * 2;
    const char* yg = reinterpret_cast<const char*>(p.dy) + ((size_t)g * p.Cout + to * 64) * 2;
    const __amdgpu_buffer_rsrc_t rs_x = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<char*>(xg), 0, (int)((size_t)p.M * p.H * p.W * p.xcs * 2), 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_y = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<char*>(yg), 0, (int)((size_t)p.rows * p.ycs * 2), 0x00020000);

    f32x16_t acc[TAPS];
#pragma unroll
    for (int t = 0; t < TAPS; ++t)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[t][e] = 0.f;

    const int b_begin = seg * p.blocks_per_seg;
    const int nblocks = (p.rows + BP - 1) / BP;
    const int b_end = min(nblocks, b_begin + p.blocks_per_seg);
    for (int blk = b_begin; blk < b_end; ++blk) {
        const int p0 = blk * BP;
        // this lane's NJ pixel rows of the block (DMA instruction j moves rows (wave + 4j)*8 + lrow); LDS chunk position lpos
        // holds the SOURCE chunk ((lpos>>1) ^ swz(row)) : lpos&1 (the swizzle is applied on the source side, see wg_swz)
        int iy0[NJ], ix0[NJ];
        unsigned ybase[NJ];
        int xbase[NJ];
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
            const int rloc = (wave + 4 * j) * 8 + lrow;
            const int pr = p0 + rloc;
            const int src16 = ((((lpos >> 1) ^ wg_swz(rloc)) << 1) | (lpos & 1)) * 16;
            if (pr < p.rows) {
                const int hw = p.Ho * p.Wo;
                const int m = pr / hw, rem = pr - m * hw;
                const int oy = rem / p.Wo, ox = rem - oy * p.Wo;
                iy0[j] = oy * p.stride - p.pad;
                ix0[j] = ox * p.stride - p.pad;
                ybase[j] = (unsigned)((size_t)pr * p.ycs * 2 + src16);
                xbase[j] = (int)((((long)m * p.H + iy0[j]) * p.W + ix0[j]) * p.xcs * 2 + src16);
            } else {
                iy0[j] = -100000; ix0[j] = 0; ybase[j] = 0x80000000u; xbase[j] = 0;
            }
        }
        auto stage_x = [&](int tap, int buf) {
            const int ky = TAPS == 1 ? 0 : tap / 3, kx = TAPS == 1 ? 0 : tap - 3 * (tap / 3);
            const int tap_off = (ky * p.W + kx) * p.xcs * 2;
#pragma unroll
            for (int j = 0; j < NJ; ++j) {
                const int iy = iy0[j] + ky, ix = ix0[j] + kx;
                const bool ok = ((unsigned)iy < (unsigned)p.H) & ((unsigned)ix < (unsigned)p.W);
                const unsigned vo = ok ? (unsigned)(xbase[j] + tap_off) : 0x80000000u;
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_x, W2C_LPTR(Xs + buf * (BP * 128) + (wave + 4 * j) * 1024), 16, vo, 0, 0, 0);
            }
        };
        __syncthreads();                                   // previous block's tiles are no longer read
#pragma unroll
        for (int j = 0; j < NJ; ++j)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_y, W2C_LPTR(Ys + (wave + 4 * j) * 1024), 16, ybase[j], 0, 0, 0);
        stage_x(0, 0);
        bf16x8_t ya[BP / 16];
#pragma unroll
        for (int tap = 0; tap < TAPS; ++tap) {
            wg_wait_vmcnt<0>();
            __syncthreads();                               // tile(tap) (and dY) landed for everyone; the other X buffer is free
            if (tap + 1 < TAPS) stage_x(tap + 1, (tap + 1) & 1);
            if (tap == 0) {
#pragma unroll
                for (int kk = 0; kk < BP / 16; ++kk) ya[kk] = tr_frag(Ys, kk * 16, wm * 32, lane);
            }
            const char* xt = Xs + (tap & 1) * (BP * 128);
#pragma unroll
            for (int kk = 0; kk < BP / 16; ++kk) {
                const bf16x8_t xb = tr_frag(xt, kk * 16, wn * 32, lane);
                acc[tap] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ya[kk], xb, acc[tap], 0, 0, 0);
            }
        }
    }
    // partial tile -> workspace: D[i = co][j = ci]; lane holds column ci = l31, rows co = (e&3) + 8(e>>2) + 4 lhi
    const int l31 = lane & 31, lhi = lane >> 5;
    const size_t wsz = (size_t)gridDim.z * p.Cout * TAPS * p.Cin;
    float* out = p.ws + (size_t)seg * wsz + ((size_t)g * p.Cout + to * 64 + wm * 32) * TAPS * p.Cin + ti * 64 + wn * 32 + l31;
#pragma unroll
    for (int tap = 0; tap < TAPS; ++tap)
#pragma unroll
        for (int e = 0; e < 16; ++e) {
            const int co = (e & 3) + 8 * (e >> 2) + 4 * lhi;
            out[((size_t)co * TAPS + tap) * p.Cin] = acc[tap][e];
        }
#endif
}

// 3x3 / stride-1 form on HALO PATCHES (the forward patch kernel's idea): a block is an 8 x 16 spatial tile of one image; its
// (8+2) x (16+2) input patch is staged ONCE (23 LDS-DMA instructions per workgroup) and all 9 taps read their transposed
// fragments from it at a shifted pixel index -- the generic form above re-stages a 128-pixel X tile per tap (9 x 16
// instructions, and a vmcnt(0) + barrier per tap that exposes the DMA latency nine times per block).  Patch + dY tile are
// double-buffered across blocks: one barrier per block, 72 MFMAs per wave between barriers.  78 KB of LDS: two workgroups per
// CU.  Same partial-tile / segment / reduction contract as conv_wgrad_kernel (the pixel order inside a segment differs, so the
// two forms round differently; each is deterministic).
__global__ __launch_bounds__(256) void conv_wgrad_patch_kernel(WgradArgs p) {
#if defined(__HIP_DEVICE_COMPILE__)
    constexpr int TH = 8, TW = 16, PW = TW + 2, NP = (TH + 2) * PW;          // 180 patch pixels
    constexpr int NPI = (NP + 7) / 8;                                          // 23 DMA instructions (8 pixel rows each)
    constexpr int PATCH_B = NPI * 1024, Y_B = TH * TW * 128, BUF_B = PATCH_B + Y_B;
    constexpr int PJ = (NPI + 3) / 4;                                          // patch instructions per wave (last: waves 0-2)
    extern __shared__ __attribute__((aligned(16))) char smem[];               // [2][patch | dY tile]
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1;
    const int g = blockIdx.z, seg = blockIdx.y;
    const int to = blockIdx.x / p.nct_i, ti = blockIdx.x - to * p.nct_i;
    const int lrow = lane >> 3, lpos = lane & 7;

    const char* xg = reinterpret_cast<const char*>(p.x) + ((size_t)g * p.Cin + ti * 64) * 2;
    const char* yg = reinterpret_cast<const char*>(p.dy) + ((size_t)g * p.Cout + to * 64) * 2;
    const __amdgpu_buffer_rsrc_t rs_x = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<char*>(xg), 0, (int)((size_t)p.M * p.H * p.W * p.xcs * 2), 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_y = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<char*>(yg), 0, (int)((size_t)p.rows * p.ycs * 2), 0x00020000);

    f32x16_t acc[9];
#pragma unroll
    for (int t = 0; t < 9; ++t)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[t][e] = 0.f;

    const int tiles_x = p.Wo / TW, tiles_y = p.Ho / TH;
    const int nblocks = p.M * tiles_x * tiles_y;
    const int b_begin = seg * p.blocks_per_seg;
    const int b_end = min(nblocks, b_begin + p.blocks_per_seg);
    // this lane's fixed positions: patch pixel of DMA instruction j, tile pixel of dY instruction j
    int ppy[PJ], ppx[PJ], psrc[PJ];
#pragma unroll
    for (int j = 0; j < PJ; ++j) {
        const int q = (wave + 4 * j) * 8 + lrow;
        ppy[j] = q / PW; ppx[j] = q - ppy[j] * PW;
        psrc[j] = q < NP ? ((((lpos >> 1) ^ wg_swz(q)) << 1) | (lpos & 1)) * 16 : -1;
    }
    int yty[4], ytx[4], ysrc[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int rloc = (wave + 4 * j) * 8 + lrow;
        yty[j] = rloc / TW; ytx[j] = rloc - yty[j] * TW;
        ysrc[j] = ((((lpos >> 1) ^ wg_swz(rloc)) << 1) | (lpos & 1)) * 16;
    }
    auto issue = [&](int blk, int buf) {
        const int txi = blk % tiles_x, t2 = blk / tiles_x;
        const int tyi = t2 % tiles_y, img = t2 / tiles_y;
        const int y0 = tyi * TH, x0 = txi * TW;
        char* pd = smem + buf * BUF_B;
#pragma unroll
        for (int j = 0; j < PJ; ++j) {
            if (wave + 4 * j < NPI) {                                        // wave-uniform
                const int iy = y0 - 1 + ppy[j], ix = x0 - 1 + ppx[j];
                const bool ok = (psrc[j] >= 0) & ((unsigned)iy < (unsigned)p.H) & ((unsigned)ix < (unsigned)p.W);
                const unsigned vo = ok ? (unsigned)((((long)img * p.H + iy) * p.W + ix) * p.xcs * 2 + psrc[j]) : 0x80000000u;
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_x, W2C_LPTR(pd + (wave + 4 * j) * 1024), 16, vo, 0, 0, 0);
            }
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const unsigned vo = (unsigned)((((long)img * p.Ho + y0 + yty[j]) * p.Wo + x0 + ytx[j]) * p.ycs * 2 + ysrc[j]);
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_y, W2C_LPTR(pd + PATCH_B + (wave + 4 * j) * 1024), 16, vo, 0, 0, 0);
        }
    };
    if (b_begin < b_end) issue(b_begin, 0);
    int buf = 0;
    for (int blk = b_begin; blk < b_end; ++blk, buf ^= 1) {
        wg_wait_vmcnt<0>();
        __syncthreads();                     // block `blk` landed for everyone; everyone is done reading the other buffer
        if (blk + 1 < b_end) issue(blk + 1, buf ^ 1);
        const char* patch = smem + buf * BUF_B;
        const char* Ys = patch + PATCH_B;
        bf16x8_t ya[8];
#pragma unroll
        for (int kk = 0; kk < 8; ++kk) ya[kk] = tr_frag(Ys, kk * 16, wm * 32, lane);
#pragma unroll
        for (int tap = 0; tap < 9; ++tap) {
            const int ky = tap / 3, kx = tap - 3 * ky;
#pragma unroll
            for (int kk = 0; kk < 8; ++kk) {                                  // tile row kk = 16 consecutive patch pixels
                const bf16x8_t xb = tr_frag(patch, (kk + ky) * PW + kx, wn * 32, lane);
                acc[tap] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ya[kk], xb, acc[tap], 0, 0, 0);
            }
        }
    }
    const int l31 = lane & 31, lhi = lane >> 5;
    const size_t wsz = (size_t)gridDim.z * p.Cout * 9 * p.Cin;
    float* out = p.ws + (size_t)seg * wsz + ((size_t)g * p.Cout + to * 64 + wm * 32) * 9 * p.Cin + ti * 64 + wn * 32 + l31;
#pragma unroll
    for (int tap = 0; tap < 9; ++tap)
#pragma unroll
        for (int e = 0; e < 16; ++e) {
            const int co = (e & 3) + 8 * (e >> 2) + 4 * lhi;
            out[((size_t)co * 9 + tap) * p.Cin] = acc[tap][e];
        }
#endif
}

// Sum of the nseg partial tiles, 4 consecutive floats per column index i.  SL = 1: one thread walks the segments of its
// column (fine for the handful of segments of the wide layers).  SL = 16: layer1-type convs have ONE 64x64 tile and ~400
// segments -- 9 216 columns x 435 dependent loads ran at 0.6 TB/s (109 us for 64 MB, 2.5x the MFMA kernel); here 16 threads
// share a column (segments sl, sl+16, ...: independent loads, 4 in flight each), combined through LDS in lane order.
// Either way the order of additions is fixed by (nseg, SL): deterministic.
template <int SL>
__device__ __forceinline__ bool wgrad_seg_sum(const float* __restrict__ ws, long n4, int nseg, long& i, f32x4_t& s) {
    if constexpr (SL == 1) {
        i = (long)blockIdx.x * 256 + threadIdx.x;
        if (i >= n4) return false;
        s = reinterpret_cast<const f32x4_t*>(ws)[i];
        for (int k = 1; k < nseg; ++k) s += reinterpret_cast<const f32x4_t*>(ws)[(size_t)k * n4 + i];
        return true;
    } else {
        constexpr int COLS = 256 / SL;
        __shared__ f32x4_t red[SL][COLS];
        const int col = threadIdx.x % COLS, sl = threadIdx.x / COLS;
        i = (long)blockIdx.x * COLS + col;
        const bool in = i < n4;
        f32x4_t a0 = {0.f, 0.f, 0.f, 0.f}, a1 = a0, a2 = a0, a3 = a0;
        if (in) {
            const f32x4_t* base = reinterpret_cast<const f32x4_t*>(ws) + i;
            int k = sl;
            for (; k + 3 * SL < nseg; k += 4 * SL) {
                a0 += base[(size_t)k * n4];
                a1 += base[(size_t)(k + SL) * n4];
                a2 += base[(size_t)(k + 2 * SL) * n4];
                a3 += base[(size_t)(k + 3 * SL) * n4];
            }
            for (; k < nseg; k += SL) a0 += base[(size_t)k * n4];
        }
        red[sl][col] = (a0 + a1) + (a2 + a3);
        __syncthreads();
        if (sl != 0 || !in) return false;
        s = red[0][col];
#pragma unroll
        for (int j = 1; j < SL; ++j) s += red[j][col];
        return true;
    }
}

template <int SL>
__global__ __launch_bounds__(256) void wgrad_reduce_kernel(const float* __restrict__ ws, float* __restrict__ dw, long n4, int nseg) {
    long i;
    f32x4_t s;
    if (!wgrad_seg_sum<SL>(ws, n4, nseg, i, s)) return;
    reinterpret_cast<f32x4_t*>(dw)[i] = s;
}

// the same reduction, written in nn.Conv2d's parameter layout [Cout][Cin][ky][kx] (groups = 1) so that the optimiser reads the
// gradient where autograd expects it, with no permute copy in between.  i indexes 4 consecutive ci of one (co, tap).
template <int SL>
__global__ __launch_bounds__(256) void wgrad_reduce_oihw_kernel(const float* __restrict__ ws, float* __restrict__ dw, long n4, int nseg,
                                                                 int taps, int cin) {
    long i;
    f32x4_t s;
    if (!wgrad_seg_sum<SL>(ws, n4, nseg, i, s)) return;
    const long e = i * 4;                          // = (co * taps + tap) * cin + ci
    const int ci = (int)(e % cin);
    const long ct = e / cin;
    const int tap = (int)(ct % taps);
    const long co = ct / taps;
    float* d = dw + (co * cin + ci) * taps + tap;
#pragma unroll
    for (int j = 0; j < 4; ++j) d[(long)j * taps] = s[j];
}

// nn.Conv2d's f32 parameter [Cout][Cin][ky][kx] -> the kernels' packed bf16 operand, one launch (the torch formulation is a
// permute copy + a cast (+ a flip): three passes and three launches per conv and step):
//   mode 0 (forward)  out[co][tap][ci]
//   mode 1 (dgrad)    out[ci][taps-1-tap][co]      (flipped taps, transposed channels)
// A workgroup moves a 32(co) x 32(ci) x taps block through LDS: reads are runs of 32*taps consecutive floats, writes runs of
// 32 consecutive bf16.
template <int T>
__global__ __launch_bounds__(256) void pack_weights_kernel(const float* __restrict__ w, uint16_t* __restrict__ out,
                                                           uint16_t* __restrict__ out_dgrad, int Cout, int Cin, int mode) {
    constexpr int PITCH = 32 * T + 1;        // odd pitch per co row: the transposed (mode 1) read walks co with the lane index --
                                             // at pitch 288 words all 64 lanes hit two banks (the first version: 15 us per call,
                                             // 1.3 ms per training step)
    __shared__ float tile[32 * PITCH];
    const int co0 = blockIdx.y * 32, ci0 = blockIdx.x * 32;
    constexpr int NIT = 32 * 32 * T / 256;   // all of a thread's loads in flight together (a rolled loop waited for each one: 15 us)
    float v[NIT];
#pragma unroll
    for (int k = 0; k < NIT; ++k) {
        const int r = threadIdx.x + k * 256;
        const int co_l = r / (32 * T), rem = r - co_l * (32 * T);
        v[k] = w[((long)(co0 + co_l) * Cin + ci0) * T + rem];
    }
#pragma unroll
    for (int k = 0; k < NIT; ++k) {
        const int r = threadIdx.x + k * 256;
        const int co_l = r / (32 * T), rem = r - co_l * (32 * T);
        tile[co_l * PITCH + rem] = v[k];
    }
    __syncthreads();
#pragma unroll 4
    for (int o = threadIdx.x; o < 32 * 32 * T; o += 256) {
        const int a = o / (32 * T), t = (o >> 5) % T, b = o & 31;
        if (mode != 1) out[((long)(co0 + a) * T + t) * Cin + ci0 + b] = f32_to_bf16(tile[a * PITCH + b * T + t]);
        if (mode != 0) out_dgrad[((long)(ci0 + a) * T + t) * Cout + co0 + b] = f32_to_bf16(tile[b * PITCH + a * T + (T - 1 - t)]);
    }
}

// dY (stride-2 conv output grid) -> zero-inserted tensor on the input grid: u[m][2oy][2ox] = dy[m][oy][ox], 0 elsewhere;
// the stride-2 conv's input gradient is then a stride-1 conv of u with the flipped, transposed weights.
__global__ __launch_bounds__(256) void zero_insert2_kernel(const uint4* __restrict__ dy, uint4* __restrict__ u, int M, int Ho, int Wo, int H,
                                                           int W, int c8) {
    const size_t total = (size_t)M * H * W * c8;
    for (size_t id = (size_t)blockIdx.x * 256 + threadIdx.x; id < total; id += (size_t)gridDim.x * 256) {
        const int c = (int)(id % c8);
        size_t t = id / c8;
        const int ix = (int)(t % W);
        t /= W;
        const int iy = (int)(t % H);
        const int m = (int)(t / H);
        uint4 v = make_uint4(0, 0, 0, 0);
        if (!(iy & 1) && !(ix & 1) && (iy >> 1) < Ho && (ix >> 1) < Wo)
            v = dy[(((size_t)m * Ho + (iy >> 1)) * Wo + (ix >> 1)) * c8 + c];
        u[id] = v;
    }
}

}  // namespace

extern "C" long long w2c_conv_wgrad_workspace_bytes(int M, int H, int W, int Cin, int Cout, int ksize, int stride, int groups) {
    if (M <= 0 || H <= 0 || W <= 0 || Cin <= 0 || Cin % 64 || Cout <= 0 || Cout % 64 || groups <= 0) return -1;
    if (!((ksize == 3) || (ksize == 1)) || !((stride == 1) || (stride == 2))) return -1;
    const int pad = ksize == 3 ? 1 : 0;
    const long rows = (long)M * ((H + 2 * pad - ksize) / stride + 1) * ((W + 2 * pad - ksize) / stride + 1);
    const long nblocks = (rows + 127) / 128;
    const long tiles = (long)(Cout / 64) * (Cin / 64) * groups;
    // pixel segments: enough workgroups to fill the chip (~768), but every workgroup must amortise its 64x64x9 f32 partial
    // tile (147 KB written, then re-read by the reduction) over >= 8 pixel blocks, and a conv's partials stay <= 64 MB.
    // (First version: 1536 workgroups whatever the layer -> 226 MB of partials per layer4 / layer1 conv and a reduction that
    // cost twice the MFMA kernel; second: >= 32 blocks per workgroup -> ~160 workgroups, 0.6 per CU, 122 us per conv.)
    long nseg = (768 + tiles - 1) / tiles;
    if (nseg > nblocks / 4) nseg = nblocks / 4;
    const long per_seg = (long)groups * Cout * ksize * ksize * Cin * 4;
    if (nseg * per_seg > (64L << 20)) nseg = (64L << 20) / per_seg;      // <= 64 MB of partials per conv
    if (nseg < 1) nseg = 1;
    return nseg * (long long)groups * Cout * ksize * ksize * Cin * 4;
}

static int wgrad_impl(const uint16_t* x, int M, int H, int W, int Cin, int x_cstride,
                      const uint16_t* dy, int Cout, int dy_cstride, int ksize, int stride, int groups,
                      float* dw, void* workspace, long long workspace_bytes, w2c_stream_t stream, bool oihw) {
    w2c_clear_error();
    if (!x || !dy || !dw || !workspace) return W2C_E_ARG;
    const long long need = w2c_conv_wgrad_workspace_bytes(M, H, W, Cin, Cout, ksize, stride, groups);
    if (need < 0 || workspace_bytes < need || (reinterpret_cast<uintptr_t>(workspace) & 15) || (reinterpret_cast<uintptr_t>(dw) & 15))
        return W2C_E_ARG;
    if (x_cstride < groups * Cin || dy_cstride < groups * Cout || (x_cstride % 8) || (dy_cstride % 8)) return W2C_E_ARG;
    WgradArgs a;
    a.x = x; a.dy = dy; a.ws = reinterpret_cast<float*>(workspace); a.dw = dw;
    a.M = M; a.H = H; a.W = W; a.Cin = Cin; a.xcs = x_cstride;
    a.ks = ksize; a.stride = stride; a.pad = ksize == 3 ? 1 : 0;
    a.Ho = (H + 2 * a.pad - ksize) / stride + 1;
    a.Wo = (W + 2 * a.pad - ksize) / stride + 1;
    a.Cout = Cout; a.ycs = dy_cstride;
    if ((size_t)M * H * W * x_cstride * 2 >= (1ull << 31) || (size_t)M * a.Ho * a.Wo * dy_cstride * 2 >= (1ull << 31)) return W2C_E_ARG;
    a.rows = M * a.Ho * a.Wo;
    const long per = (long)groups * Cout * ksize * ksize * Cin * 4;
    a.nseg = (int)(need / per);
    const int nblocks = (a.rows + 127) / 128;
    a.blocks_per_seg = (nblocks + a.nseg - 1) / a.nseg;
    a.nct_o = Cout / 64; a.nct_i = Cin / 64;
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    dim3 grid(a.nct_o * a.nct_i, a.nseg, groups);
    if (ksize == 3 && stride == 1 && a.Ho % 8 == 0 && a.Wo % 16 == 0 && w2c_option(W2C_OPT_WGRAD_PATCH) != 0) {
        constexpr int lds = 2 * (23 * 1024 + 128 * 128);
        static std::atomic<unsigned long long> attr_mask{0};
        int dev = 0;
        (void)hipGetDevice(&dev);
        if (!((attr_mask.load(std::memory_order_acquire) >> (dev & 63)) & 1ull)) {
            (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_wgrad_patch_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
            attr_mask.fetch_or(1ull << (dev & 63), std::memory_order_release);
        }
        hipLaunchKernelGGL(conv_wgrad_patch_kernel, grid, dim3(256), lds, s, a);
    } else if (ksize == 3) hipLaunchKernelGGL((conv_wgrad_kernel<9>), grid, dim3(256), 0, s, a);
    else hipLaunchKernelGGL((conv_wgrad_kernel<1>), grid, dim3(256), 0, s, a);
    int rc = w2c_launch_status();
    if (rc != W2C_OK) return rc;
    const long n4 = per / 16;
    if (a.nseg >= 32) {                      // many segments of a small tile set: 16 threads per column
        const dim3 rg((unsigned)((n4 + 15) / 16));
        if (oihw) hipLaunchKernelGGL((wgrad_reduce_oihw_kernel<16>), rg, dim3(256), 0, s, a.ws, dw, n4, a.nseg, ksize * ksize, Cin);
        else hipLaunchKernelGGL((wgrad_reduce_kernel<16>), rg, dim3(256), 0, s, a.ws, dw, n4, a.nseg);
    } else {
        const dim3 rg((unsigned)((n4 + 255) / 256));
        if (oihw) hipLaunchKernelGGL((wgrad_reduce_oihw_kernel<1>), rg, dim3(256), 0, s, a.ws, dw, n4, a.nseg, ksize * ksize, Cin);
        else hipLaunchKernelGGL((wgrad_reduce_kernel<1>), rg, dim3(256), 0, s, a.ws, dw, n4, a.nseg);
    }
    return w2c_launch_status();
}

extern "C" int w2c_conv_wgrad_bf16(const uint16_t* x, int M, int H, int W, int Cin, int x_cstride,
                                   const uint16_t* dy, int Cout, int dy_cstride, int ksize, int stride, int groups,
                                   float* dw, void* workspace, long long workspace_bytes, w2c_stream_t stream) {
    return wgrad_impl(x, M, H, W, Cin, x_cstride, dy, Cout, dy_cstride, ksize, stride, groups, dw, workspace, workspace_bytes, stream, false);
}

extern "C" int w2c_conv_wgrad_bf16_oihw(const uint16_t* x, int M, int H, int W, int Cin, int x_cstride,
                                        const uint16_t* dy, int Cout, int dy_cstride, int ksize, int stride,
                                        float* dw, void* workspace, long long workspace_bytes, w2c_stream_t stream) {
    return wgrad_impl(x, M, H, W, Cin, x_cstride, dy, Cout, dy_cstride, ksize, stride, 1, dw, workspace, workspace_bytes, stream, true);
}

static int pack_impl(const float* w_oihw, int Cout, int Cin, int ksize, int mode, uint16_t* out, uint16_t* out_dgrad, w2c_stream_t stream) {
    w2c_clear_error();
    if (!w_oihw || Cout <= 0 || Cin <= 0 || (Cout % 32) || (Cin % 32) || (ksize != 1 && ksize != 3)) return W2C_E_ARG;
    if ((mode != 1 && !out) || (mode != 0 && !out_dgrad)) return W2C_E_ARG;
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    dim3 grid(Cin / 32, Cout / 32);
    if (ksize == 3) hipLaunchKernelGGL((pack_weights_kernel<9>), grid, dim3(256), 0, s, w_oihw, out, out_dgrad, Cout, Cin, mode);
    else hipLaunchKernelGGL((pack_weights_kernel<1>), grid, dim3(256), 0, s, w_oihw, out, out_dgrad, Cout, Cin, mode);
    return w2c_launch_status();
}

extern "C" int w2c_pack_conv_weights_bf16(const float* w_oihw, int Cout, int Cin, int ksize, int mode, uint16_t* out, w2c_stream_t stream) {
    if (mode != 0 && mode != 1) return W2C_E_ARG;
    return pack_impl(w_oihw, Cout, Cin, ksize, mode, mode == 0 ? out : nullptr, mode == 1 ? out : nullptr, stream);
}

extern "C" int w2c_pack_conv_weights_bf16_both(const float* w_oihw, int Cout, int Cin, int ksize, uint16_t* out_fwd, uint16_t* out_dgrad,
                                               w2c_stream_t stream) {
    return pack_impl(w_oihw, Cout, Cin, ksize, 2, out_fwd, out_dgrad, stream);
}

extern "C" int w2c_zero_insert2_bf16(const uint16_t* dy, int M, int Ho, int Wo, int C, uint16_t* u, int H, int W, w2c_stream_t stream) {
    w2c_clear_error();
    if (!dy || !u || M <= 0 || Ho <= 0 || Wo <= 0 || C <= 0 || (C % 8) || H < 2 * Ho - 1 || W < 2 * Wo - 1 || H > 2 * Ho || W > 2 * Wo)
        return W2C_E_ARG;
    const size_t total = (size_t)M * H * W * (C / 8);
    size_t blocks = (total + 255) / 256;
    if (blocks > 8192) blocks = 8192;
    hipLaunchKernelGGL(zero_insert2_kernel, dim3((unsigned)blocks), dim3(256), 0, reinterpret_cast<hipStream_t>(stream),
                       reinterpret_cast<const uint4*>(dy), reinterpret_cast<uint4*>(u), M, Ho, Wo, H, W, C / 8);
    return w2c_launch_status();
}
