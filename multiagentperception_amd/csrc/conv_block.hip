// conv_block.hip -- a whole ResNet BasicBlock with Cin = Cout = 64 (layer1) as ONE kernel for gfx950 (CDNA4).
//
// Stands in for   y = relu(bn2(conv2(relu(bn1(conv1(x))))) + x)   -- the two stride-1 BasicBlocks of layer1 of the third-party
// resnet18 the reference instantiates at backbone.py:63-69 (conv3x3 - BN - ReLU - conv3x3 - BN - add - ReLU).  As two launches
// of conv3x3_c64_regw_kernel (conv_igemm.hip) the block moves 5 activation maps through HBM (x, t, t, x, y: 420 MB per block at
// cfg 2) for 97 GFLOP and runs against the memory system; here the intermediate map t never leaves the CU: 2 maps, 168 MB.
//
// Geometry: "flattened strips".  A workgroup owns a strip = rows [y0, y1) x columns [c0, c0+Wc) of one (image, group).  Rows of the
// strip are laid end to end with pitch P = Wc + 8 positions (position p of a row <-> image column c0 - 2 + p: two halo columns on
// the left, the rest of the pitch on the right), so a 3x3 tap is the constant index shift ky*P + kx - 1 and an MFMA pixel tile is
// simply 32 CONSECUTIVE positions -- tiles straddle rows, nothing is recomputed along x except the pitch padding (8 of 136), and
// along y only the two halo rows of t per strip.  With X(q) = GX + q the index of x-position q (row y0-2 + q/P) and T likewise
// (row y0-1 + q/P):
//     t[q] = relu(bn1(sum_taps w1[ky][kx] . x[X(q) + ky*P + kx - 1]))   forced to 0 outside the image (conv2's zero padding)
//     y[q] = relu(bn2(sum_taps w2[ky][kx] . t[T(q) + ky*P + kx - 1]) + x[X(q) + 2P])
// x and t live in two LDS RINGS of 128-byte positions (64 bf16 channels); 16-byte chunk c of ring position r sits at
// c ^ ((r >> 1) & 7), which makes the 32-consecutive-position ds_read_b128 fragment reads conflict-free at every tap shift.
//
// Roles: four waves, one per SIMD, each with the whole 512-register file, all 64 x 576 weights of ITS conv in registers
// (288; MFMA A operands may be AGPRs) -- waves 0,1 run conv1 (tiles 2s, 2s+1 at step s), waves 2,3 run conv2 L steps behind
// (tiles 2(s-L), 2(s-L)+1), L = ceil((2P + 65) / 64) so that every t position a conv2 tile reads was written in an earlier step.
// One s_barrier per step publishes the step's t tiles and the x pieces that landed; x arrives by LDS-DMA (buffer_load ... lds,
// 8 positions per instruction, out-of-image lanes carry an out-of-range offset -> zeros), two pieces per wave and step, two
// steps ahead of their first use; the residual is read back from the x ring (it is still there: conv2 trails conv1 by less than
// the ring).  conv2's output tile goes through a wave-private 4 KB staging so the global stores are whole 128-byte rows.
// tools/model_block_fused.py executes exactly these index formulas (ring sizes, lag, piece schedule) on poisoned numpy rings.
//
// Numerics: the same MFMA sequence per output element as conv3x3_c64_regw_kernel (tap-major, 16-channel steps), the same
// f32 epilogue (fma(acc, scale, shift) (+ residual), ReLU, round-to-nearest-even bf16) and t rounded to bf16 exactly where the
// two-launch form stores it: results are bit-identical to the two launches (tests/test_kernels_gpu.py).
#include "w2c_common.h"
#include <cstdlib>

namespace {

constexpr int GXY = 8;             // guard positions in front of index 0 of both rings (tap (0,0) of position 0 reads index GXY - 1)

struct BlockArgs {
    const uint16_t* x;
    const uint16_t* w1;
    const uint16_t* w2;
    const float* sc1;
    const float* sh1;
    const float* sc2;
    const float* sh2;
    uint16_t* y;
    int M, H, W, xcs, ycs, G;
    int Wc, ncs, P, L, NX, NT;     // column-strip width, column strips per image, pitch, conv2 lag (steps), ring sizes (positions)
    int NS, U;                     // strips in the launch, units (image, group, column strip)
    unsigned magP;                 // ceil(2^32 / P): q / P == umulhi(q, magP) for every q < 2^24
    unsigned long long* dbg;
};

template <int N>
__device__ __forceinline__ void blk_wait_vmcnt() {
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

template <int AB>
__global__ __launch_bounds__(256) void conv_block_c64_fused_kernel(BlockArgs p) {
#if defined(__HIP_DEVICE_COMPILE__)
    constexpr int WPITCH = 1152 + 16;              // weight rows in LDS during the prologue: 16-B pad => conflict-free fragment reads
    constexpr int STAGE_BYTES = 32 * 128;          // conv2 output staging per wave
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int role = wave >> 1, wsel = wave & 1;   // role 0: conv1 (x ring -> t ring), role 1: conv2 (t ring -> y)
    const int l31 = lane & 31, lhi = lane >> 5;
    const int P = p.P, L = p.L, NX = p.NX, NT = p.NT;
    const int tring_off = NX * 128;
    const int stage_off = (NX + NT) * 128 + (role * 2 + wsel) * STAGE_BYTES;      // every wave has one (conv1 waves: a dump for idle epilogues)
    const unsigned lds_base = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(size_t)W2C_LPTR(smem));

    // ring this wave's MFMA loop reads: byte offset of the ring in LDS and its size in positions
    const int rd_off = role ? tring_off : 0;
    const int rd_n = role ? NT : NX;

    const size_t x_bytes = (size_t)p.M * p.H * p.W * p.xcs * 2;
    const size_t y_bytes = (size_t)p.M * p.H * p.W * p.ycs * 2;
    // x is read by LDS-DMA issued from INLINE ASM with a hand-built descriptor: hipcc orders every LDS access that follows a
    // compiler-visible buffer_load ... lds behind s_waitcnt vmcnt(0) (it cannot tell the rings' slots apart), which would park
    // every step on the HBM latency of the pieces it has just issued.  The waits for these pieces are the counted ones below.
    const unsigned long long xaddr = reinterpret_cast<unsigned long long>(p.x);
    const u32x4_t srd_x = {(unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)xaddr),
                           (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(xaddr >> 32)),
                           (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)x_bytes), 0x00020000u};
    const __amdgpu_buffer_rsrc_t rs_y = __builtin_amdgcn_make_buffer_rsrc(p.y, 0, (int)y_bytes, 0x00020000);

    // DMA lane constants: a piece = 8 consecutive ring positions; lane -> position l = lane / 8, 16-B slot lane % 8; the slot's
    // SOURCE chunk is slot ^ key(ring position), key = ((piece & 1) << 2) | (l >> 1) (ring sizes are multiples of 16 positions)
    const int dl = lane >> 3;
    const int d_lane_even = dl * p.xcs * 2 + (((lane & 7) ^ (dl >> 1)) << 4);
    const int d_lane_odd = dl * p.xcs * 2 + (((lane & 7) ^ (4 | (dl >> 1))) << 4);

    // Register classes are pinned by the inline-asm MFMAs: channel tile 0 and the accumulators in AGPRs, channel tile 1 in VGPRs
    // (the conv3x3_c64_regw_kernel split).  A value that is "a" in one statement and "v" in another is shuttled through
    // v_accvgpr_write before every use -- and hipcc pads no VALU-write -> MFMA-read hazard inside asm.
    u32x4_t wa[9][4];          // channel tile 0 (channels 0..31): AGPRs
    u32x4_t wbv[9][4];         // channel tile 1: VGPRs
    int cur_g = -1;
    // BN scale | shift of both convs: 4 x 64 f32 behind the staging buffers, re-read per tile (64 registers otherwise)
    float* const cst = reinterpret_cast<float*>(smem + (NX + NT) * 128 + 4 * STAGE_BYTES);
    const float* const my_sc = cst + role * 128 + lhi * 4;       // this lane's channels: ct*32 + eg*8 + lhi*4 + i

    for (int sid = blockIdx.x; sid < p.NS; sid += gridDim.x) {
        // ---- strip decode (wave-uniform) ----
        const int sbase_n = p.NS / p.U, srem = p.NS - sbase_n * p.U;
        int u, si, sn;
        if (sid < srem * (sbase_n + 1)) { u = sid / (sbase_n + 1); si = sid - u * (sbase_n + 1); sn = sbase_n + 1; }
        else { const int s2 = sid - srem * (sbase_n + 1); u = srem + s2 / sbase_n; si = s2 - (u - srem) * sbase_n; sn = sbase_n; }
        const int upg = p.M * p.ncs;
        const int g = u / upg;
        const int ur = u - g * upg;
        const int img = ur / p.ncs;
        const int c0 = (ur - img * p.ncs) * p.Wc;
        const int y0 = (int)(((long)si * p.H) / sn), y1 = (int)(((long)(si + 1) * p.H) / sn);
        const int R = y1 - y0;
        const int n1 = ((R + 2) * P + 31) >> 5, n2 = (R * P + 31) >> 5;
        int S = (n1 + 1) >> 1;                      // + 1: the epilogue of a tile runs in the step after its MFMAs
        if (((n2 + 1) >> 1) + L > S) S = ((n2 + 1) >> 1) + L;
        S += 1;

        // ---- weights of this wave's conv -> registers (through LDS: coalesced in, fragment-shaped out) ----
        if (g != cur_g) {
            __syncthreads();                       // previous strip: every wave is done with the rings
            const uint16_t* w1g = p.w1 + (size_t)g * 64 * 576;
            const uint16_t* w2g = p.w2 + (size_t)g * 64 * 576;
            for (int c = tid; c < 64 * 72; c += 256) {
                const int row = c / 72, col = c - row * 72;
                *reinterpret_cast<uint4*>(smem + row * WPITCH + col * 16) = *reinterpret_cast<const uint4*>(w1g + row * 576 + col * 8);
                *reinterpret_cast<uint4*>(smem + (64 + row) * WPITCH + col * 16) = *reinterpret_cast<const uint4*>(w2g + row * 576 + col * 8);
            }
            __syncthreads();
            const char* wsrc = smem + role * 64 * WPITCH;
#pragma unroll
            for (int tap = 0; tap < 9; ++tap)
#pragma unroll
                for (int kc = 0; kc < 4; ++kc) {
                    wa[tap][kc] = *reinterpret_cast<const u32x4_t*>(wsrc + l31 * WPITCH + tap * 128 + kc * 32 + lhi * 16);
                    const u32x4_t w1t = *reinterpret_cast<const u32x4_t*>(wsrc + (32 + l31) * WPITCH + tap * 128 + kc * 32 + lhi * 16);
                    wbv[tap][kc] = w1t;
                }
            cur_g = g;
            __syncthreads();                       // weights read out: the constants below land in what was the weight image
            if (tid < 64) {
                cst[tid] = p.sc1[g * 64 + tid]; cst[64 + tid] = p.sh1[g * 64 + tid];
                cst[128 + tid] = p.sc2[g * 64 + tid]; cst[192 + tid] = p.sh2[g * 64 + tid];
            }
        }
        __syncthreads();                           // weights read out (or: previous strip drained) -- the rings may be written

        // ---- x pieces.  Piece i covers X indices 8i..8i+7, X = GXY + q, q = xrow*P + pos; image row y0-2+xrow, column c0-2+pos ----
        const int NSL = NX >> 3;                   // ring slots (pieces)
        auto issue_piece = [&](int i, int slot) {
            const int q0 = 8 * i - GXY;
            const int xrow = (int)__umulhi((unsigned)q0, p.magP);
            const int pos0 = q0 - xrow * P;
            const int row = y0 - 2 + xrow;
            int col0 = c0 - 2 + pos0;
            const int sb = ((img * p.H + row) * p.W + col0) * p.xcs * 2 + g * 128;     // may be negative / meaningless for dead lanes
            if ((unsigned)row >= (unsigned)p.H) col0 = -(1 << 20);                      // dead row: every lane fails the column test
            const bool ok = (unsigned)(col0 + dl) < (unsigned)p.W;
            const unsigned vo = ok ? (unsigned)(sb + ((i & 1) ? d_lane_odd : d_lane_even)) : 0x80000000u;
            const unsigned lds_dst = lds_base + ((unsigned)slot << 10);
            unsigned keep;
            asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %1\n\ts_nop 0\n\tbuffer_load_dwordx4 %2, %3, 0 offen lds\n\ts_mov_b32 m0, %0"
                         : "=&s"(keep) : "s"(lds_dst), "v"(vo), "s"(srd_x) : "memory");
        };
        const int npre = (P >> 2) + 17;            // pieces 1..npre are in LDS before step 0 (they cover steps 0 and 1)
        for (int i = 1 + wave; i <= npre; i += 4) issue_piece(i, i % NSL);
        blk_wait_vmcnt<0>();
        __syncthreads();
        int pslot = (npre + 2 * wave + 1) % NSL;   // slot of this wave's first piece of step 0 (pieces npre + 8s + 2w + 1, + 2)

        // ---- cursors (wave-uniform), advanced by 64 positions / 2 tiles per step ----
        // rd_s  : ring position of tap (0,0) of position 0 of the CURRENT tile in the ring this wave's MFMAs read
        // aux_s : role 0: T ring position of position 0 of the current tile (where its t goes, one step later);
        //         role 1: X ring position of the residual of the current tile (read during the MFMA step, used one step later)
        // (conv2 starts L steps late: its cursors start 64 L positions back, modulo the ring)
        int rd_s = role ? (GXY - 1 + 32 * wsel + (NT << 4) - 64 * L) % NT : (GXY - 1 + 32 * wsel) % NX;
        int aux_s = role ? (GXY + 32 * wsel + 2 * P + (NX << 4) - 64 * L) % NX : (GXY + 32 * wsel) % NT;
        int aux_prev = 0;                          // aux_s of the previous step (role 0: where the PREVIOUS tile's t goes)
        int tile = role ? wsel - 2 * L : wsel;     // conv1: 2s + wsel ; conv2: 2(s - L) + wsel
        const int ntile = role ? n2 : n1;
        int s = 0;

        unsigned long long ph[4] = {0, 0, 0, 0};   // debug (p.dbg): cycles in the tile body | - | vmcnt wait | barrier
        long long tp = p.dbg ? clock64() : 0;
        auto stamp = [&](int i) {
            if (p.dbg) { const long long n = clock64(); ph[i] += (unsigned long long)(n - tp); tp = n; }
        };

        // One step.  (ca0, ca1): accumulators of the CURRENT tile (MFMAs of this step); (pa0, pa1): those of the tile of the previous
        // step, whose epilogue is spread through this step's MFMA stream -- a wave alone on its SIMD has nobody else to fill the
        // matrix pipe while it does VALU / LDS work, and a wave's instructions issue in order: whatever should run under an MFMA
        // has to sit right behind it in the stream.  So every tap is written as 8 x (one MFMA + a "gap" of <= ~6 other
        // instructions), each gap closed by a sched_barrier so the compiler keeps it there:
        //   gaps 0-1: LDS address of tap+1's fragments      gaps 2-3: its four ds_read_b128 (a full tap ahead of their MFMAs)
        //   gaps 4-7: one (channel tile, 8-channel group) quad of the previous tile's epilogue (taps 0-7), or its way out of the
        //             staging buffer to global memory (tap 8); the two x pieces of the step ride in taps 1 and 2.
        // Both roles run the SAME instruction stream (no branches inside): conv1 adds a zero residual and masks t outside the image,
        // conv2 adds the x fragments and masks nothing; an idle epilogue (no previous tile) writes to the wave's own staging buffer
        // and its stores carry an out-of-range offset.  rc / rp: residual fragments of the current / previous conv2 tile.  The caller
        // alternates the two register sets (static names).
        auto step = [&](f32x16_t& ca0, f32x16_t& ca1, f32x16_t& pa0, f32x16_t& pa1, uint2 (&rc)[8], uint2 (&rp)[8]) {
            const int ptile = tile - 2;
            const bool cur_on = (tile >= 0) & (tile < ntile);
            const bool prev_on = (ptile >= 0) & (ptile < ntile);
            const int i0 = npre + 8 * s + 2 * wave + 1;          // this step's two x pieces (needed from step s + 2 on)
            bool stored = false;
            if (cur_on | prev_on) {
                // ---- previous tile: where its epilogue writes (LDS), what it masks ----
                const int pq = 32 * ptile + l31;
                unsigned dst = stage_off + l31 * 128 + lhi * 8;   // conv2 (and idle conv1 epilogues): the wave's staging rows
                int dkey = l31 & 7;
                bool tvalid = true;
                if (role == 0) {
                    const int rr = (int)__umulhi((unsigned)pq, p.magP);
                    const int pos = pq - rr * P;
                    tvalid = ((unsigned)(y0 - 1 + rr) < (unsigned)p.H) & ((unsigned)(c0 - 2 + pos) < (unsigned)p.W);
                    if (prev_on) {
                        int r = aux_prev + l31;
                        r = (int)__builtin_elementwise_min((unsigned)r, (unsigned)(r - NT));
                        dst = tring_off + (r << 7) + lhi * 8;
                        dkey = (r >> 1) & 7;
                    }
                } else if (cur_on) {
                    // residual of the CURRENT conv2 tile: 4 channels (8 bytes) per (ct, eg) from the x ring, kept for the next step
                    int r = aux_s + l31;
                    r = (int)__builtin_elementwise_min((unsigned)r, (unsigned)(r - NX));
                    const char* ra = smem + (r << 7) + lhi * 8;
                    const int key = (r >> 1) & 7;
#pragma unroll
                    for (int c = 0; c < 8; ++c) rc[c] = *reinterpret_cast<const uint2*>(ra + ((c ^ key) << 4));
                }
                // conv2 tile out of the staging (tap 8): this lane's four 16-byte pieces and where they go
                const int cg = lane & 7;
                unsigned st_off[4];
#pragma unroll
                for (int it = 0; it < 4; ++it) {
                    const int ps = it * 8 + (lane >> 3);
                    const int oq = 32 * ptile + ps;
                    const int orow = (int)__umulhi((unsigned)oq, p.magP);
                    const int pos = oq - orow * P;
                    const bool ok = (role != 0) & prev_on & ((unsigned)(pos - 2) < (unsigned)p.Wc) & (orow < R);
                    const unsigned off = (unsigned)((((img * p.H + y0 + orow) * p.W + c0 - 2 + pos) * p.ycs + g * 64 + cg * 8) * 2);
                    st_off[it] = ok ? off : 0x80000000u;
                }

                const char* fb = nullptr;
                int fx = 0, tr = 0;
                u32x4_t bx[2][4];
                auto frag = [&](int kc) { return *reinterpret_cast<const u32x4_t*>(fb + ((kc << 5) ^ fx)); };
                {   // tap 0's fragments (exposed once per tile)
                    int r = l31 + rd_s;
                    r = (int)__builtin_elementwise_min((unsigned)r, (unsigned)(r - rd_n));
                    fb = smem + rd_off + (r << 7);
                    fx = (((r >> 1) ^ lhi) & 7) << 4;
#pragma unroll
                    for (int kc = 0; kc < 4; ++kc) bx[0][kc] = frag(kc);
                }
                f32x4_t scn = *reinterpret_cast<const f32x4_t*>(my_sc), shn = *reinterpret_cast<const f32x4_t*>(my_sc + 64);
                f32x4_t scq, shq;
                float v[4];
                uint32_t o0 = 0, o1 = 0;
                u32x4_t so[4];
#define W2C_SB __builtin_amdgcn_sched_barrier(0)
#pragma unroll
                for (int tap = 0; tap < 9; ++tap) {
                    const int cb = tap & 1;
                    const int ct = (tap >> 2) & 1, eg = tap & 3;                 // the quad of this tap (taps 0..7)
                    const f32x16_t& pa = ct ? pa1 : pa0;
                    const int nky = (tap + 1) / 3, nkx = (tap + 1) - 3 * nky;     // next tap
                    // ---- MFMA 0 (channel tile 0) | gap 0: next tap's ring position ----
                    // "=&a": without the early clobber hipcc may place the accumulator ON TOP of the weight fragment (seen:
                    // v_mfma a[0:15], a[0:3], ...).  s_nop 1: if the register allocator parks an "a" operand in VGPRs it copies
                    // it back with v_accvgpr_write right in front of the statement, and nothing inside asm is hazard-padded.
                    if (tap == 0) asm volatile("s_nop 1\n\tv_mfma_f32_32x32x16_bf16 %0, %1, %2, 0" : "=&a"(ca0) : "a"(wa[0][0]), "v"(bx[0][0]));
                    else asm volatile("s_nop 1\n\tv_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+a"(ca0) : "a"(wa[tap][0]), "v"(bx[cb][0]));
                    if (tap < 8) {
                        tr = l31 + (rd_s + nky * P + nkx);
                        tr = (int)__builtin_elementwise_min((unsigned)tr, (unsigned)(tr - rd_n));
                    }
                    W2C_SB;
                    asm volatile("s_nop 1\n\tv_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+a"(ca0) : "a"(wa[tap][1]), "v"(bx[cb][1]));
                    if (tap < 8) {
                        fb = smem + rd_off + (tr << 7);
                        fx = (((tr >> 1) ^ lhi) & 7) << 4;
                    }
                    W2C_SB;
                    asm volatile("s_nop 1\n\tv_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+a"(ca0) : "a"(wa[tap][2]), "v"(bx[cb][2]));
                    if (tap < 8) { bx[cb ^ 1][0] = frag(0); bx[cb ^ 1][1] = frag(1); }
                    W2C_SB;
                    asm volatile("s_nop 1\n\tv_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+a"(ca0) : "a"(wa[tap][3]), "v"(bx[cb][3]));
                    if (tap < 8) { bx[cb ^ 1][2] = frag(2); bx[cb ^ 1][3] = frag(3); }
                    W2C_SB;
                    // ---- channel tile 1 | gaps 4-7: the epilogue quad (taps 0..7) or the staged tile's way out (tap 8) ----
                    if (tap == 0) asm volatile("s_nop 1\n\tv_mfma_f32_32x32x16_bf16 %0, %1, %2, 0" : "=&a"(ca1) : "v"(wbv[0][0]), "v"(bx[0][0]));
                    else asm volatile("s_nop 1\n\tv_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+a"(ca1) : "v"(wbv[tap][0]), "v"(bx[cb][0]));
                    if (!(AB & 1)) {
                        if (tap < 8) {
                            scq = scn; shq = shn;
#pragma unroll
                            for (int i = 0; i < 4; ++i) v[i] = pa[eg * 4 + i];
                        } else {
                            so[0] = *reinterpret_cast<const u32x4_t*>(smem + stage_off + (0 * 8 + (lane >> 3)) * 128 + ((cg ^ ((lane >> 3) & 7)) << 4));
                            so[1] = *reinterpret_cast<const u32x4_t*>(smem + stage_off + (1 * 8 + (lane >> 3)) * 128 + ((cg ^ ((lane >> 3) & 7)) << 4));
                        }
                    }
                    W2C_SB;
                    asm volatile("s_nop 1\n\tv_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+a"(ca1) : "v"(wbv[tap][1]), "v"(bx[cb][1]));
                    if (!(AB & 1)) {
                        if (tap < 8) {
#pragma unroll
                            for (int i = 0; i < 4; ++i) v[i] = __builtin_fmaf(v[i], scq[i], shq[i]);
                            const uint2 rw = rp[ct * 4 + eg];
                            v[0] += __uint_as_float(rw.x << 16); v[1] += __uint_as_float(rw.x & 0xFFFF0000u);
                        } else {
                            so[2] = *reinterpret_cast<const u32x4_t*>(smem + stage_off + (2 * 8 + (lane >> 3)) * 128 + ((cg ^ ((lane >> 3) & 7)) << 4));
                            so[3] = *reinterpret_cast<const u32x4_t*>(smem + stage_off + (3 * 8 + (lane >> 3)) * 128 + ((cg ^ ((lane >> 3) & 7)) << 4));
                        }
                    }
                    if (tap == 1 && !(AB & 2)) issue_piece(i0, pslot);
                    W2C_SB;
                    asm volatile("s_nop 1\n\tv_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+a"(ca1) : "v"(wbv[tap][2]), "v"(bx[cb][2]));
                    if (!(AB & 1)) {
                        if (tap < 8) {
                            const uint2 rw = rp[ct * 4 + eg];
                            v[2] += __uint_as_float(rw.y << 16); v[3] += __uint_as_float(rw.y & 0xFFFF0000u);
                            o0 = pack_bf16x2(v[0], v[1]); o1 = pack_bf16x2(v[2], v[3]);
                            // ReLU on the packed pairs (a signed 16-bit max with 0 clears exactly the negative bf16, like the two-launch form)
                            o0 = __builtin_bit_cast(uint32_t, __builtin_elementwise_max(__builtin_bit_cast(s16x2_t, o0), s16x2_t{0, 0}));
                            o1 = __builtin_bit_cast(uint32_t, __builtin_elementwise_max(__builtin_bit_cast(s16x2_t, o1), s16x2_t{0, 0}));
                        } else {
                            __builtin_amdgcn_raw_buffer_store_b128(so[0], rs_y, st_off[0], 0, 0);
                            __builtin_amdgcn_raw_buffer_store_b128(so[1], rs_y, st_off[1], 0, 0);
                        }
                    }
                    if (tap == 2 && !(AB & 2)) issue_piece(i0 + 1, (pslot + 1 == NSL) ? 0 : pslot + 1);
                    W2C_SB;
                    asm volatile("s_nop 1\n\tv_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+a"(ca1) : "v"(wbv[tap][3]), "v"(bx[cb][3]));
                    if (!(AB & 1)) {
                        if (tap < 8) {
                            if (!tvalid) { o0 = 0; o1 = 0; }                    // conv1: outside the image = conv2's zero padding
                            *reinterpret_cast<uint2*>(smem + dst + (((ct * 4 + eg) ^ dkey) << 4)) = make_uint2(o0, o1);
                            if (tap < 7) {                                      // BN scale / shift of the next quad
                                const int nct = ((tap + 1) >> 2) & 1, neg = (tap + 1) & 3;
                                scn = *reinterpret_cast<const f32x4_t*>(my_sc + nct * 32 + neg * 8);
                                shn = *reinterpret_cast<const f32x4_t*>(my_sc + 64 + nct * 32 + neg * 8);
                            }
                        } else {
                            __builtin_amdgcn_raw_buffer_store_b128(so[2], rs_y, st_off[2], 0, 0);
                            __builtin_amdgcn_raw_buffer_store_b128(so[3], rs_y, st_off[3], 0, 0);
                            stored = true;
                        }
                    }
                    W2C_SB;
                }
#undef W2C_SB
            } else if (!(AB & 2)) {
                issue_piece(i0, pslot);
                issue_piece(i0 + 1, (pslot + 1 == NSL) ? 0 : pslot + 1);
            }
            pslot += 8;
            if (pslot >= NSL) pslot -= NSL;
            stamp(0);
            // end of step: everything this wave issued in EARLIER steps has landed (this step's own 2 pieces (+4 stores) may still
            // fly), its LDS writes are done, then the barrier publishes them
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            if (stored) blk_wait_vmcnt<6>(); else blk_wait_vmcnt<2>();
            stamp(2);
            if (!(AB & 8)) __builtin_amdgcn_s_barrier();
            asm volatile("" ::: "memory");
            stamp(3);
            tile += 2;
            ++s;
            aux_prev = aux_s;
            rd_s += 64; if (rd_s >= rd_n) rd_s -= rd_n;
            const int aux_n = role ? NX : NT;
            aux_s += 64; if (aux_s >= aux_n) aux_s -= aux_n;
        };

        f32x16_t accA0, accA1, accB0, accB1;
        uint2 resA[8], resB[8];
#pragma unroll
        for (int c = 0; c < 8; ++c) { resA[c] = make_uint2(0, 0); resB[c] = make_uint2(0, 0); }
        while (s < S) {
            step(accA0, accA1, accB0, accB1, resA, resB);
            if (s < S) step(accB0, accB1, accA0, accA1, resB, resA);
        }
        blk_wait_vmcnt<0>();                       // strip done: drain before the rings are re-indexed
        if (p.dbg && lane == 0) {
            unsigned long long* d = p.dbg + ((size_t)blockIdx.x * 4 + wave) * 8;
            for (int i = 0; i < 4; ++i) d[i] = ph[i];
            d[4] = (unsigned long long)S; d[5] = (unsigned long long)R;
        }
    }
#endif
}

}  // namespace

// Debug: the next w2c_conv_block_c64 call of THIS thread writes per-wave phase cycle sums (8 x u64 per wave: MFMA loop, epilogue,
// vmcnt wait, barrier, steps, rows) to `buf` (tools/block_phases.py).
static thread_local unsigned long long* g_blk_dbg_next = nullptr;
extern "C" int w2c_debug_block_phases(void* buf) {
    g_blk_dbg_next = reinterpret_cast<unsigned long long*>(buf);
    return W2C_OK;
}

extern "C" int w2c_conv_block_c64(const uint16_t* x, int M, int H, int W, int x_cstride,
                                  const uint16_t* w1, const float* scale1, const float* shift1,
                                  const uint16_t* w2, const float* scale2, const float* shift2,
                                  int groups, uint16_t* y, int y_cstride, int max_workgroups, w2c_stream_t stream) {
    w2c_clear_error();
    if (!x || !w1 || !w2 || !scale1 || !shift1 || !scale2 || !shift2 || !y || x == y) return W2C_E_ARG;
    if (M <= 0 || H <= 0 || W <= 0 || groups <= 0) return W2C_E_ARG;
    if (x_cstride < groups * 64 || y_cstride < groups * 64 || (x_cstride % 8) || (y_cstride % 8)) return W2C_E_ARG;
    if ((W % 8) != 0 || (W > 128 && (W % 128) != 0) || H > 512) return W2C_E_ARG;
    if ((size_t)M * H * W * x_cstride * 2 >= (1ull << 31) || (size_t)M * H * W * y_cstride * 2 >= (1ull << 31)) return W2C_E_ARG;
    BlockArgs a;
    a.x = x; a.w1 = w1; a.w2 = w2; a.sc1 = scale1; a.sh1 = shift1; a.sc2 = scale2; a.sh2 = shift2; a.y = y;
    a.M = M; a.H = H; a.W = W; a.xcs = x_cstride; a.ycs = y_cstride; a.G = groups;
    a.Wc = W <= 128 ? W : 128;
    a.ncs = W / a.Wc;
    a.P = a.Wc + 8;
    a.L = (2 * a.P + 129 + 63) / 64;               // tools/model_block_fused.py: lag, ring sizes
    a.NT = (64 * (a.L - 1) + 65 + 15) / 16 * 16;
    a.NX = (64 * a.L + 200 + 15) / 16 * 16;
    a.magP = (unsigned)(((1ull << 32) + a.P - 1) / a.P);
    // timing ablations (-DW2C_BLOCK_ABLATIONS builds only; WRONG results): 1 no epilogue, 2 no per-step DMA, 8 no barrier
#ifdef W2C_BLOCK_ABLATIONS
    static const int ablate_env = [] { const char* e = getenv("W2C_BLOCK_ABLATE"); return e ? atoi(e) : 0; }();
#else
    constexpr int ablate_env = 0;
#endif
    a.dbg = g_blk_dbg_next;
    g_blk_dbg_next = nullptr;
    static int n_cu[64] = {0};
    int dev = 0;
    (void)hipGetDevice(&dev);
    constexpr int WBYTES = 2 * 64 * (1152 + 16);
    const int ring_bytes = (a.NX + a.NT) * 128 + 4 * 32 * 128 + 1024;
    const int lds = ring_bytes > WBYTES ? ring_bytes : WBYTES;
    if (lds > 160 * 1024) return W2C_E_ARG;
    static std::atomic<unsigned long long> attr_mask{0};
    if (!((attr_mask.load(std::memory_order_acquire) >> (dev & 63)) & 1ull)) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_block_c64_fused_kernel<0>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
#ifdef W2C_BLOCK_ABLATIONS
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_block_c64_fused_kernel<11>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_block_c64_fused_kernel<27>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_block_c64_fused_kernel<1>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
#endif
        hipDeviceProp_t prop;
        n_cu[dev & 63] = (hipGetDeviceProperties(&prop, dev) == hipSuccess && prop.multiProcessorCount > 0)
                             ? prop.multiProcessorCount : 256;
        attr_mask.fetch_or(1ull << (dev & 63), std::memory_order_release);
    }
    int cus = n_cu[dev & 63];
    if (max_workgroups > 0 && max_workgroups < cus) cus = max_workgroups;
    // strips: units (image, group, column strip) are cut into equal row ranges so that every CU gets one strip when there are
    // fewer units than CUs (never shorter than 4 rows: each strip recomputes two halo rows of t); else one strip per unit
    a.U = M * groups * a.ncs;
    if (a.U >= cus) a.NS = a.U;
    else {
        int per = cus / a.U;                       // floor: the first (cus % U) units get one strip more
        const int cap = H / 4 > 0 ? H / 4 : 1;
        if (per >= cap) a.NS = a.U * cap;
        else a.NS = cus;
    }
    const int grid = a.NS < cus ? a.NS : cus;
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    (void)ablate_env;
#ifdef W2C_BLOCK_ABLATIONS
    if (ablate_env == 11) { hipLaunchKernelGGL(conv_block_c64_fused_kernel<11>, dim3(grid), dim3(256), lds, st, a); return w2c_launch_status(); }
    if (ablate_env == 27) { hipLaunchKernelGGL(conv_block_c64_fused_kernel<27>, dim3(grid), dim3(256), lds, st, a); return w2c_launch_status(); }
    if (ablate_env == 1) { hipLaunchKernelGGL(conv_block_c64_fused_kernel<1>, dim3(grid), dim3(256), lds, st, a); return w2c_launch_status(); }
#endif
    hipLaunchKernelGGL(conv_block_c64_fused_kernel<0>, dim3(grid), dim3(256), lds, st, a);
    return w2c_launch_status();
}
