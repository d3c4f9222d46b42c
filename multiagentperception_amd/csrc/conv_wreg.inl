// =====================================================================================================
// 3x3 stride-1 convolution, "weights to registers" form (included by conv_igemm.hip, inside its namespace).
//
// Why another kernel.  In the ring kernels above every MFMA operand comes out of LDS: a 32x32 (v36) or 64x64 (v30) WAVE tile
// costs 2 / 1 ds_read_b128 per MFMA, the LDS pipe is as busy as the matrix pipe (SQ: lds% ~ mfma%), a K-step is only 4-16 MFMAs
// per wave between two workgroup barriers, and each K-step pays 8-16 LDS-DMA issues (~100 cycles each) for its weight tile.
// Big workgroup tiles would fix the ratio but do not fit these layers: at cfg 2 a layer4 conv is 5120 pixels x 512 channels --
// 2560 outputs per SIMD of the chip.
// Here the WAVE tile is 128 pixels x 64 channels (8 accumulators in AGPRs, 0.5 ds_read_b128 per MFMA) and the parallelism comes
// from splitting K over the waves of a workgroup instead of shrinking the tile:
//   * workgroup = NN x KS waves on ONE 8 x 16-pixel tile: wave (nw, kg) owns channels [64 nw, +64) and, of EVERY K-step
//     (t = chunk * 9 + tap, 64 input channels per chunk), the 16-deep k slices [kg * 4 / KS, +4 / KS) -- all waves walk the same
//     taps in step, so the chunk barriers cost no waiting (a split by taps leaves one K group a tap behind per chunk: -11 %);
//   * the WEIGHTS never touch LDS: they are pre-packed in MFMA A-fragment order (ops.pack_wfrag: one contiguous 1 KB block per
//     (32 channels, K-step, k slice), lane-linear), so a slice's weights are 2 perfectly coalesced 16-byte-per-lane buffer loads
//     straight into VGPRs, issued two K-steps ahead (a ring of three register sets) -- no DMA issue cost, no weight ring in LDS,
//     no per-K-step barrier;
//   * the input patch (10 x 18 pixels x 64 channels, halo included) is staged by LDS-DMA as in the ring kernels, in a ring of
//     THREE buffers: one workgroup barrier per 64-channel chunk (before its last tap: patch cc+1 has landed, buffer cc-1 is free
//     for patch cc+2), and the fragment read-ahead runs across the chunk boundary;
//   * B fragments: one ds_read_b128 per two MFMAs, the reads of slice n+1 interleaved 1 : 2 with the MFMAs of slice n
//     (sched_group_barrier), 4 reads in flight;
//   * after the loop the KS partial tiles are reduced through LDS in log2(KS) pairwise rounds (each wave ends up owning
//     128 / KS pixels x 64 channels, fixed summation order) and written register-direct: scale / shift (+ residual, loaded
//     before the exchange) + ReLU, bf16 pack, v_permlane32_swap pairs the half-waves' channel quads into 16-byte stores.
// LDS: 72 KB of patch buffers -> two workgroups per CU; 128 AGPRs + <= 128 VGPRs -> two waves per SIMD.
// The K order differs from the ring kernels' (partial sums per K group): results agree to f32 rounding, not bit for bit.
// NB = 32-channel blocks per wave (2: the wave tile above; 1: 128 pixels x 32 channels -- twice the workgroups on the same pixel tiles, for
// launches that cannot fill the chip: <= 128 workgroups of the NB = 2 form (a rank's share of a sharded batch, Single_agent at small
// batches) are one ~20 us workgroup life on half the CUs; the library switches by the launch's workgroup count (conv_igemm.hip
// wreg_small).  Same K groups, same reduction order: bit-identical to NB = 2, so the choice may depend on the image count.
// F32OUT: the output tensor is f32 (the decoder's first conv taken through the fusion by linearity: U = conv0_nobias(V), engine.DecoderPlan);
// a lane stores its two channel quads as they sit in the accumulators (two 16-byte stores 8 channels apart): no pack, no half-wave swap.
template <int NN, int KS, int ABL = 0, int DW = 8, int NB = 2, bool F32OUT = false>      // ABL: timing ablations (wrong results): 1 no weight loads, 2 no fragment reads
__global__ __launch_bounds__(64 * NN * KS, 2) void conv3x3_wreg_kernel(ConvArgs p) {
#if defined(__HIP_DEVICE_COMPILE__)
    constexpr int PW = 18, NP = 180, NPIECE = 23;  // patch: 10 x 18 pixels, 128 B each, DMA'd in 1 KB pieces of 8 pixels
    constexpr int NW = NN * KS;
    constexpr int P_INSTR = (NPIECE + NW - 1) / NW;
    constexpr int PATCH_STRIDE = P_INSTR * NW * 1024;   // every wave issues P_INSTR pieces; those past the patch (and the last piece's
                                                        // 4 dead pixels) are all-out-of-range loads that write zeros into the pad
    constexpr int KK = 4 / KS;                     // k slices (16 deep) per K-step per wave
    // weights are loaded D K-steps ahead into a ring of R register sets (R divides 9, D < R).  A K-step is 8 KK MFMAs = 256 KK cycles
    // of matrix pipe per wave: with one slice per step (KS = 4) two steps ahead is only ~0.5-1k cycles -- less than an L2 round
    // trip under load -- so that form runs a 9-deep ring (72 VGPRs), DW steps ahead
    constexpr int R = (KK == 1 && DW > 2) ? 9 : 3, D = KK == 1 ? DW : 2;
    constexpr int CW = 32 * NB;                    // channels per wave
    constexpr int NB_SYNC = 8 * NB * KK > 56 ? 56 : 8 * NB * KK;   // weight loads a wave issues between a patch and the chunk sync that needs it (>=)
    static_assert(KS == 1 || KS == 2 || KS == 4, "K split");
    // one dummy "a" operand makes the backend pick the AGPR form of every builtin MFMA here (accumulators in AGPRs)
    asm volatile("" ::"a"(0));

    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l31 = lane & 31, lhi = lane >> 5;
    const int nw = wave % NN, kg = wave / NN;

    dbg_stamp(p, 0);
    const int tiles_x = p.W >> 4, tiles_y = p.H >> 3;
    const int g = blockIdx.y;
    int tsp, tn;
    if (p.xcd2d) {
        // weight-heavy layers (layer4: 4.7 MB of weights per conv against 4 MB of L2 per XCD; every wave streams its weights from
        // L2): XCD x (= block id mod 8) owns one half of the pixel tiles and one quarter of the channel tiles -- 1.2 MB of weights +
        // 2.6 MB of input per XCD instead of all the weights through every L2.  Fabric reads per layer4 launch 64 -> ~36 MB (PMC);
        // time-neutral (the Infinity Cache holds everything).  Placement only: the result does not depend on it.
        const int xcd = blockIdx.x & 7, loc = blockIdx.x >> 3;
        const int qn = p.ntn >> 2, hm = p.ntm >> 1;
        const int tm_l = w2c_fastdiv(loc, qn, p.mg_qn);
        tn = (xcd >> 1) * qn + (loc - tm_l * qn);
        tsp = (xcd & 1) * hm + tm_l;
    } else {
        const int tile = xcd_remap(blockIdx.x, p.ntm * p.ntn);
        tsp = w2c_fastdiv(tile, p.ntn, p.mg_ntn);
        tn = tile - tsp * p.ntn;
    }
    const int img = w2c_fastdiv(tsp, tiles_x * tiles_y, p.mg_txy);
    const int trem = tsp - img * (tiles_x * tiles_y);
    const int tyi = w2c_fastdiv(trem, tiles_x, p.mg_tx), txi = trem - tyi * tiles_x;
    const int y0 = tyi * 8, x0 = txi * 16;
    const int n0 = tn * (NN * CW) + nw * CW;        // this wave's first output channel inside the group
    const int nchunks = p.Cin >> 6, KT = nchunks * 9;

    // BN scale | shift of this wave's 64 channels: parked in LDS (512 B per wave behind the patch / exchange area) for the epilogue
    constexpr int XR = 8192 * NB;                  // exchange region of a wave (round 1: two pixel blocks x NB accumulators x 4 KB)
    constexpr int SS_BASE = (3 * PATCH_STRIDE > NW * XR ? 3 * PATCH_STRIDE : NW * XR);
    float* const ssw = reinterpret_cast<float*>(smem + SS_BASE + wave * 512) + lhi * 4;
    // (parked by two dword LDS-DMA requests behind the first patches, below: a load + ds_write here put the load's whole round trip in
    //  front of the workgroup's first patch request)

    // ---- input patch: LDS-DMA from inline asm (hidden from the compiler's vmcnt bookkeeping: counted by hand below) ----
    const unsigned lds_base = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(size_t)W2C_LPTR(smem));
    const unsigned long long xaddr = reinterpret_cast<unsigned long long>(p.x) + (unsigned long long)g * p.Cin * 2;
    const unsigned x_bytes = (unsigned)((size_t)p.M * p.H * p.W * p.xcs * 2);
    const u32x4_t srd_x = {(unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)xaddr),
                           (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(xaddr >> 32)),
                           (unsigned)__builtin_amdgcn_readfirstlane((int)x_bytes), 0x00020000u};
    unsigned pa_off[P_INSTR];                      // this lane's source byte offset per piece (out of range => zeros)
#pragma unroll
    for (int j = 0; j < P_INSTR; ++j) {
        const int q = (wave + NW * j) * 8 + (lane >> 3);
        const int py = q / PW, px = q - py * PW;
        const int chunk = (lane & 7) ^ ((px >> 1) & 7);             // swizzle keyed on the patch column (see the ring kernel)
        const int iy = y0 - 1 + py, ix = x0 - 1 + px;
        const bool ok = (q < NP) & ((unsigned)iy < (unsigned)p.H) & ((unsigned)ix < (unsigned)p.W);
        pa_off[j] = ok ? (unsigned)(((img * p.H + iy) * p.W + ix) * p.xcs * 2 + chunk * 16) : 0x80000000u;   // (< 2 GiB: launch_wreg)
    }
    auto issue_patch = [&](int cc, int buf) {
        const unsigned soff = (unsigned)cc * 128u;
#pragma unroll
        for (int j = 0; j < P_INSTR; ++j) {
            const unsigned dst = lds_base + (unsigned)buf * PATCH_STRIDE + (unsigned)(wave + NW * j) * 1024u;
            unsigned keep;
            asm volatile("s_nop 4\n\ts_mov_b32 %0, m0\n\ts_mov_b32 m0, %1\n\ts_nop 0\n\tbuffer_load_dwordx4 %2, %3, %4 offen lds\n\ts_mov_b32 m0, %0"
                         : "=&s"(keep) : "s"(dst), "v"(pa_off[j]), "s"(srd_x), "s"(soff) : "memory");
        }
    };

    // ---- weights: fragment-packed, coalesced 16-byte loads straight to registers (compiler-counted) ----
    const char* wbase = reinterpret_cast<const char*>(p.w) + (size_t)g * (p.Cout >> 5) * KT * 4096;
    const __amdgpu_buffer_rsrc_t rs_w = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(wbase), 0, (p.Cout >> 5) * KT * 4096, 0x00020000);
    const int wv = lane * 16 + kg * KK * 1024;      // this lane's 16 bytes of this wave's first k slice inside a (32 channels, K-step) block
    const int nb0 = n0 >> 5;
    const int ws0 = nb0 * KT * 4096, ws1 = ws0 + KT * 4096;
    auto load_a = [&](u32x4_t (&A)[NB][KK], int t) {
#pragma unroll
        for (int q = 0; q < KK; ++q) {
            A[0][q] = __builtin_bit_cast(u32x4_t, __builtin_amdgcn_raw_buffer_load_b128(rs_w, wv + q * 1024, ws0 + t * 4096, 0));
            if constexpr (NB == 2)
                A[NB - 1][q] = __builtin_bit_cast(u32x4_t, __builtin_amdgcn_raw_buffer_load_b128(rs_w, wv + q * 1024, ws1 + t * 4096, 0));
        }
    };

    // KS == 4 (one slice per tap): the first tap of the first chunk takes the constant 0 as its C operand instead of 128 accumulator
    // writes in front of the loop (v_accvgpr_write is a VALU instruction: 0.25 us of a SIMD that also hosts another workgroup's MFMAs)
    constexpr bool ZERO_C = (KK == 1);
    f32x16_t acc[4][NB];
    if constexpr (!ZERO_C) {
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < NB; ++j)
#pragma unroll
                for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;
    }

    // Pixel-block order of this wave's accumulators: acc[i] is pixel block blk(i) of the tile (rows 2 blk, 2 blk + 1).  The order is
    // chosen per K group so that the reduction after the loop is the SAME code in every wave (no accumulator flows through a
    // wave-dependent branch -- that cost 20-56 spilled registers): acc[0 .. 4/KS) are the blocks the wave ends up owning, the
    // rest is what it sends, in the order the receiving partner keeps them.
    int blkoff[4];                                  // wave-uniform LDS byte offsets of the blocks inside a patch
    {
        const int b0 = kg & 1, b1 = (kg >> 1) & 1;
        auto fin = [&](int k) { return ((k & 1) ? 2 : 0) + ((k >> 1) ? 1 : 0); };
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            int blk;
            if constexpr (KS == 1) blk = i;
            else if constexpr (KS == 2) blk = (i + 2 * kg) & 3;
            else blk = fin(kg ^ (i == 0 ? 0 : i == 1 ? 2 : i == 2 ? 1 : 3));
            blkoff[i] = __builtin_amdgcn_readfirstlane(blk * (2 * PW * 128));
        }
        (void)b0; (void)b1;
    }
    const int pc0 = l31 & 15;
    const int pp0 = (l31 >> 4) * PW + pc0;         // patch pixel of this lane's row of pixel block 0 at tap (0, 0)
    // LDS byte offset (inside a patch buffer) of this lane's 16 bytes of slice q at the three kx: the swizzle depends on kx only
    int boff[3][KK];
#pragma unroll
    for (int kx = 0; kx < 3; ++kx)
#pragma unroll
        for (int q = 0; q < KK; ++q)
            boff[kx][q] = (pp0 + kx) * 128 + (((((kg * KK + q) << 1) | lhi) ^ (((pc0 + kx) >> 1) & 7)) << 4);
    // B fragments (pixels) of one k slice: 4 pixel blocks x ds_read_b128; tap and q are compile-time, the buffer base is not
    auto read_b = [&](bf16x8_t (&fb)[4], const char* pbuf, auto tapc, auto qc) {
        constexpr int tap = decltype(tapc)::value, q = decltype(qc)::value;
        constexpr int ky = tap / 3, kx = tap % 3;
        const char* r = pbuf + boff[kx][q] + ky * (PW * 128);
#pragma unroll
        for (int i = 0; i < 4; ++i) fb[i] = *reinterpret_cast<const bf16x8_t*>(r + blkoff[i]);
    };
    auto mfma8 = [&](const u32x4_t (&A)[NB][KK], const bf16x8_t (&fb)[4], int q, auto firstc) {
        constexpr bool FIRST = decltype(firstc)::value;
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < NB; ++j) {
                if constexpr (FIRST)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8_t, A[j][q]), fb[i],
                                                                        f32x16_t{0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f}, 0, 0, 0);
                else
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8_t, A[j][q]), fb[i], acc[i][j], 0, 0, 0);
            }
    };

    // ---- main loop.  VMEM queue of a wave, in issue order:  P(0) P(1) A(0) A(1) | A(2) .. A(9) [sync 0] P(2) A(10) | ...
    // chunk sync cc (before the last tap of chunk cc) waits until at most the weight loads issued after P(cc+1) are outstanding. ----
    u32x4_t AR[R][NB][KK];
    bf16x8_t fb[2][4];
    issue_patch(0, 0);
    if (nchunks > 1) issue_patch(1, 1);
    {
        // BN scale | shift of this wave's channels -> LDS, one dword per lane each, as LDS-DMA (counted like the patch pieces: they are
        // older than every weight load, so each hand-counted wait below covers them)
        const unsigned long long sca = reinterpret_cast<unsigned long long>(p.scale), sha = reinterpret_cast<unsigned long long>(p.shift);
        const u32x4_t srd_sc = {(unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)sca), (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(sca >> 32)),
                                0x7FFFFFFFu, 0x00020000u};
        const u32x4_t srd_sh = {(unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)sha), (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(sha >> 32)),
                                0x7FFFFFFFu, 0x00020000u};
        const unsigned voff = (NB == 2 || lane < CW) ? (unsigned)(g * p.Cout + n0 + lane) * 4u : 0x80000000u;
        const unsigned dst = lds_base + SS_BASE + (unsigned)wave * 512u;
        unsigned keep;
        asm volatile("s_nop 4\n\ts_mov_b32 %0, m0\n\ts_mov_b32 m0, %1\n\ts_nop 0\n\tbuffer_load_dword %2, %3, 0 offen lds\n\t"
                     "s_add_u32 m0, m0, 256\n\ts_nop 0\n\tbuffer_load_dword %2, %4, 0 offen lds\n\ts_mov_b32 m0, %0"
                     : "=&s"(keep) : "s"(dst), "v"(voff), "s"(srd_sc), "s"(srd_sh) : "memory");
    }
#pragma unroll
    for (int d = 0; d < D; ++d) load_a(AR[d], d);
    if (nchunks > 1) wait_vmcnt<P_INSTR + NB * KK * D>(); else wait_vmcnt<NB * KK * D>();
    pipeline_barrier();
    dbg_stamp(p, 1);
    read_b(fb[0], smem, std::integral_constant<int, 0>{}, std::integral_constant<int, 0>{});
    if constexpr (ABL & 2) read_b(fb[1], smem, std::integral_constant<int, 0>{}, std::integral_constant<int, 0>{});

    // one chunk = 9 taps, fully unrolled; PAR = parity of the chunk's first slice in the fb[] double buffer
    auto chunk_body = [&](auto parc, int cc, int bcur, int bnext, auto firstcc) {
        constexpr int PAR = decltype(parc)::value;
        constexpr bool FIRSTC = decltype(firstcc)::value;           // the tile's first chunk (ZERO_C forms only)
        const char* pcur = smem + bcur * PATCH_STRIDE;
        const char* pnext = smem + bnext * PATCH_STRIDE;
        const int t0 = cc * 9;
        auto tap_body = [&](auto tapc) {
            constexpr int tap = decltype(tapc)::value;
            if constexpr (tap == 8) {                 // chunk sync: patch cc+1 is complete everywhere; every wave is past chunk cc-1
                wait_vmcnt<NB_SYNC>();
                pipeline_barrier();
                if (cc + 2 < nchunks) {
                    int b2 = bnext + 1; if (b2 == 3) b2 = 0;
                    issue_patch(cc + 2, b2);
                }
            }
            // pin the schedule per tap: nothing moves across this point, the weight loads of tap + D go FIRST (left alone, the
            // scheduler sinks them next to their first use two taps later and every slice waits out an L2 round trip)
            __builtin_amdgcn_sched_barrier(0);
            if constexpr (!(ABL & 1)) {
                load_a(AR[(tap + D) % R], t0 + tap + D);   // past the last K-step: other rows or zeros, never used
                __builtin_amdgcn_sched_group_barrier(0x020, NB * KK, 0);
            }
            auto slice = [&](auto qc) {
                constexpr int q = decltype(qc)::value;
                constexpr int n = tap * KK + q;
                bf16x8_t (&cur)[4] = fb[(PAR + n) & 1];
                bf16x8_t (&nxt)[4] = fb[(PAR + n + 1) & 1];
                if constexpr (!(ABL & 2)) {
                    if constexpr (q + 1 < KK) read_b(nxt, pcur, std::integral_constant<int, tap>{}, std::integral_constant<int, (q + 1) % KK>{});
                    else if constexpr (tap < 8) read_b(nxt, pcur, std::integral_constant<int, (tap + 1) % 9>{}, std::integral_constant<int, 0>{});
                    else read_b(nxt, pnext, std::integral_constant<int, 0>{}, std::integral_constant<int, 0>{});
                }
                mfma8(AR[tap % R], cur, q, std::integral_constant<bool, FIRSTC && tap == 0 && q == 0>{});
                if constexpr (!(ABL & 2)) {
#pragma unroll
                    for (int z = 0; z < 4; ++z) {
                        __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
                        __builtin_amdgcn_sched_group_barrier(0x008, NB, 0);
                    }
                }
            };
            slice(std::integral_constant<int, 0>{});
            if constexpr (KK > 1) slice(std::integral_constant<int, 1>{});
            if constexpr (KK > 2) { slice(std::integral_constant<int, 2>{}); slice(std::integral_constant<int, 3>{}); }
        };
        tap_body(std::integral_constant<int, 0>{}); tap_body(std::integral_constant<int, 1>{}); tap_body(std::integral_constant<int, 2>{});
        tap_body(std::integral_constant<int, 3>{}); tap_body(std::integral_constant<int, 4>{}); tap_body(std::integral_constant<int, 5>{});
        tap_body(std::integral_constant<int, 6>{}); tap_body(std::integral_constant<int, 7>{}); tap_body(std::integral_constant<int, 8>{});
    };
    {
        int bcur = 0, bnext = 1;
        auto adv = [&]() { bcur = bnext; bnext = bnext + 1 == 3 ? 0 : bnext + 1; };
        using No_ = std::integral_constant<bool, false>;
        if constexpr ((9 * KK) % 2 == 0) {
            for (int cc = 0; cc < nchunks; ++cc) { chunk_body(std::integral_constant<int, 0>{}, cc, bcur, bnext, No_{}); adv(); }
        } else {
            // (odd slice count per chunk: the fragment double buffer's parity alternates from chunk to chunk)
            chunk_body(std::integral_constant<int, 0>{}, 0, bcur, bnext, std::integral_constant<bool, ZERO_C>{}); adv();
            int cc = 1;
            for (; cc + 1 < nchunks; cc += 2) {
                chunk_body(std::integral_constant<int, 1>{}, cc, bcur, bnext, No_{}); adv();
                chunk_body(std::integral_constant<int, 0>{}, cc + 1, bcur, bnext, No_{}); adv();
            }
            if (cc < nchunks) chunk_body(std::integral_constant<int, 1>{}, cc, bcur, bnext, No_{});
        }
    }
    // ---- epilogue operands, issued before the exchange so that their latency hides under it: this wave will own pixel blocks
    // acc[0 .. 4 / KS) (see the reduction below); residual in the STORE layout (16 bytes = 8 consecutive channels per lane) ----
    constexpr int CNT = 4 / KS;
    uint4 rres[CNT][NB][2];
    unsigned eoff[CNT];                            // element offsets (the tensors are < 2 GiB: fill_args / ops.conv_igemm)
#pragma unroll
    for (int ii = 0; ii < CNT; ++ii) {
        const int oy = y0 + blkoff[ii] / (PW * 128) + (l31 >> 4), ox = x0 + pc0;
        eoff[ii] = (unsigned)((((size_t)img * p.H + oy) * p.W + ox) * p.ycs + (size_t)g * p.ygs + n0 + lhi * (F32OUT ? 4 : 8));
    }
#pragma unroll
    for (int j = 0; j < NB; ++j)
#pragma unroll
        for (int m = 0; m < 2; ++m)
#pragma unroll
            for (int ii = 0; ii < CNT; ++ii)
                rres[ii][j][m] = (!F32OUT && p.res) ? *reinterpret_cast<const uint4*>(p.res + eoff[ii] + j * 32 + m * 16) : make_uint4(0, 0, 0, 0);
    pipeline_barrier();                              // every wave is done with the patch buffers: LDS is reused below
    dbg_stamp(p, 2);

    // ---- K-group reduction through LDS: raw accumulator dumps, lane-linear (conflict-free 16-byte accesses) ----
    // (one accumulator block at a time -- sched_barrier: left alone the scheduler reads all 64-128 values out of the AGPRs first
    //  and spills)
    auto dump = [&](auto i0c, auto cntc, char* dst) {
        constexpr int I0 = decltype(i0c)::value, CNTD = decltype(cntc)::value;
#pragma unroll
        for (int i = 0; i < CNTD; ++i)
#pragma unroll
            for (int j = 0; j < NB; ++j) {
#pragma unroll
                for (int eg = 0; eg < 4; ++eg)
                    *reinterpret_cast<f32x4_t*>(dst + ((i * NB + j) * 4 + eg) * 1024 + lane * 16) =
                        f32x4_t{acc[I0 + i][j][eg * 4], acc[I0 + i][j][eg * 4 + 1], acc[I0 + i][j][eg * 4 + 2], acc[I0 + i][j][eg * 4 + 3]};
                __builtin_amdgcn_sched_barrier(0);
            }
    };
    auto addin = [&](auto i0c, auto cntc, const char* src) {
        constexpr int I0 = decltype(i0c)::value, CNTD = decltype(cntc)::value;
#pragma unroll
        for (int i = 0; i < CNTD; ++i)
#pragma unroll
            for (int j = 0; j < NB; ++j) {
#pragma unroll
                for (int eg = 0; eg < 4; ++eg) {
                    const f32x4_t v = *reinterpret_cast<const f32x4_t*>(src + ((i * NB + j) * 4 + eg) * 1024 + lane * 16);
#pragma unroll
                    for (int e = 0; e < 4; ++e) acc[I0 + i][j][eg * 4 + e] += v[e];
                }
                __builtin_amdgcn_sched_barrier(0);
            }
    };
    // ---- output: pixel blocks [I0, I0 + CNT) x this wave's 64 channels, register-direct ----
    auto finalize = [&]() {
#pragma unroll
        for (int j = 0; j < NB; ++j)
#pragma unroll
            for (int m = 0; m < 2; ++m) {
                const f32x4_t sc0 = *reinterpret_cast<const f32x4_t*>(ssw + j * 32 + m * 16);
                const f32x4_t sc1 = *reinterpret_cast<const f32x4_t*>(ssw + j * 32 + m * 16 + 8);
                const f32x4_t sh0 = *reinterpret_cast<const f32x4_t*>(ssw + 64 + j * 32 + m * 16);
                const f32x4_t sh1 = *reinterpret_cast<const f32x4_t*>(ssw + 64 + j * 32 + m * 16 + 8);
#pragma unroll
                for (int ii = 0; ii < CNT; ++ii) {
                    const int i = ii;
                    f32x4_t v0 = f32x4_t{acc[i][j][m * 8], acc[i][j][m * 8 + 1], acc[i][j][m * 8 + 2], acc[i][j][m * 8 + 3]} * sc0 + sh0;
                    f32x4_t v1 = f32x4_t{acc[i][j][m * 8 + 4], acc[i][j][m * 8 + 5], acc[i][j][m * 8 + 6], acc[i][j][m * 8 + 7]} * sc1 + sh1;
                    if (!F32OUT && p.res) {
                        // the lane's 16 residual bytes are in the STORE layout; the same half-wave swap (it is its own inverse)
                        // brings them into the accumulator layout (two channel quads 8 channels apart)
                        const uint4 r = rres[ii][j][m];
                        const auto ra = __builtin_amdgcn_permlane32_swap(r.x, r.z, false, false);
                        const auto rb = __builtin_amdgcn_permlane32_swap(r.y, r.w, false, false);
                        v0 += f32x4_t{__uint_as_float(ra[0] << 16), __uint_as_float(ra[0] & 0xFFFF0000u),
                                      __uint_as_float(rb[0] << 16), __uint_as_float(rb[0] & 0xFFFF0000u)};
                        v1 += f32x4_t{__uint_as_float(ra[1] << 16), __uint_as_float(ra[1] & 0xFFFF0000u),
                                      __uint_as_float(rb[1] << 16), __uint_as_float(rb[1] & 0xFFFF0000u)};
                    }
                    if (p.relu) {
#pragma unroll
                        for (int e = 0; e < 4; ++e) { v0[e] = fmaxf(v0[e], 0.f); v1[e] = fmaxf(v1[e], 0.f); }
                    }
                    if constexpr (F32OUT) {
                        float* const yo = reinterpret_cast<float*>(p.y) + eoff[ii] + j * 32 + m * 16;
                        *reinterpret_cast<f32x4_t*>(yo) = v0;
                        *reinterpret_cast<f32x4_t*>(yo + 8) = v1;
                        continue;
                    }
                    const uint32_t a0 = pack_bf16x2(v0[0], v0[1]), a1 = pack_bf16x2(v0[2], v0[3]);
                    const uint32_t b0 = pack_bf16x2(v1[0], v1[1]), b1 = pack_bf16x2(v1[2], v1[3]);
                    const auto sa = __builtin_amdgcn_permlane32_swap(a0, b0, false, false);
                    const auto sb = __builtin_amdgcn_permlane32_swap(a1, b1, false, false);
                    *reinterpret_cast<uint4*>(reinterpret_cast<uint16_t*>(p.y) + eoff[ii] + j * 32 + m * 16) = make_uint4(sa[0], sb[0], sa[1], sb[1]);
                }
                __builtin_amdgcn_sched_barrier(0);
            }
    };
    using I0_ = std::integral_constant<int, 0>; using I1_ = std::integral_constant<int, 1>;
    using I2_ = std::integral_constant<int, 2>;
    auto xr = [&](int kgx) { return smem + (kgx * NN + nw) * XR; };           // exchange region of wave (nw, kgx)
    if constexpr (KS >= 2) {
        dump(I2_{}, I2_{}, xr(kg));                                            // round 1 with K group kg ^ 1
        pipeline_barrier();
        addin(I0_{}, I2_{}, xr(kg ^ 1));
    }
    if constexpr (KS == 4) {
        dump(I1_{}, I1_{}, xr(kg ^ 1));                                        // round 2 with kg ^ 2, through the region this wave has just read
        pipeline_barrier();
        addin(I0_{}, I1_{}, xr(kg ^ 3));                                       // (the partner wrote into the region IT read: (kg ^ 2) ^ 1)
    }
    finalize();
    dbg_stamp(p, 3);
#endif
}

template <int NN, int KS, int ABL = 0, int DW = 8, int NB = 2, bool F32OUT = false>
int launch_wreg(ConvArgs& a, int groups, hipStream_t s) {
    if (a.ks != 3 || a.stride != 1 || a.Cin % 64 != 0 || a.Cout % (NN * 32 * NB) != 0 || a.H % 8 != 0 || a.W % 16 != 0 || (a.y_f32 != 0) != F32OUT || a.y8 ||
        !a.y || a.ws || (F32OUT && a.res))
        return W2C_E_ARG;
    a.ntm = a.M * (a.H / 8) * (a.W / 16);
    a.ntn = a.Cout / (NN * 32 * NB);
    if ((long)a.ntm * a.ntn * ((a.H / 8) * (a.W / 16) > a.ntn ? (a.H / 8) * (a.W / 16) : a.ntn) >= (1ll << 32)) return W2C_E_ARG;   // (fast-division range)
    a.mg_ntn = w2c_magic((unsigned)a.ntn);
    a.mg_qn = w2c_magic((unsigned)(a.ntn >> 2));
    a.mg_tx = w2c_magic((unsigned)(a.W / 16));
    a.mg_txy = w2c_magic((unsigned)((a.H / 8) * (a.W / 16)));
    constexpr int patch = 3 * ((23 + NN * KS - 1) / (NN * KS)) * NN * KS * 1024;
    constexpr int xchg = KS > 1 ? NN * KS * 8192 * NB : 0;
    constexpr int lds = (patch > xchg ? patch : xchg) + NN * KS * 512;
    static_assert(lds <= 160 * 1024, "LDS");
    static std::atomic<unsigned long long> attr_mask{0};
    int dev = 0;
    (void)hipGetDevice(&dev);
    if (!((attr_mask.load(std::memory_order_acquire) >> (dev & 63)) & 1ull)) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&conv3x3_wreg_kernel<NN, KS, ABL, DW, NB, F32OUT>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        attr_mask.fetch_or(1ull << (dev & 63), std::memory_order_release);
    }
    const int xcd2d_mode = w2c_option(W2C_OPT_XCD2D);
    const long wbytes = (long)a.Cout * 9 * a.Cin * 2;
    a.xcd2d = (xcd2d_mode == 2 || (xcd2d_mode == 1 && wbytes >= (2 << 20))) && !(a.ntm & 1) && !(a.ntn & 3);
    hipLaunchKernelGGL((conv3x3_wreg_kernel<NN, KS, ABL, DW, NB, F32OUT>), dim3(a.ntm * a.ntn, groups), dim3(64 * NN * KS), lds, s, a);
    return w2c_launch_status();
}
