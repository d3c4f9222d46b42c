// =====================================================================================================
// Stride-2 BasicBlock front, "weights to registers" form (round 4; included by conv_igemm.hip, inside its namespace).
//
// conv1 (3x3 / s2 / p1, BN, ReLU) + the block's 1x1 / s2 downsample (BN) from one launch, as conv3x3s2_patch_kernel does, but on
// the structure of conv_wreg.inl instead of the ring kernel's: that kernel reads every MFMA operand out of LDS (2 ds_read_b128 per
// MFMA on 32 x 32 wave tiles -- the LDS pipe, not the matrix pipe, bounds it: SQ MFMA busy 20 %), pays a workgroup barrier and a
// weight-tile DMA per 4 MFMAs of a wave, and stages the input once per 64-channel tile of the output.
//   * polyphase patches as there: a 3x3 / s2 conv is a 2x2 / s1 conv over the four phase images of its input; per 64-channel chunk
//     the workgroup stages patch(py, px) = the (8+1) x (16+1) pixel BLOCKS around its 8 x 16 output tile, one pixel
//     (2 by + py, 2 bx + px) of each, by LDS-DMA (the stride-2 gather is in the per-lane source offsets), and tap (ky, kx) reads
//     phase (ky != 1, kx != 1) at the block shift (ky != 0, kx != 0).  K order (chunk, then s2_tap_order): taps (0,0) (0,2) (2,0) (2,2)
//     of phase (1,1), (0,1) (2,1) of (1,0), (1,0) (1,2) of (0,1), (1,1) of (0,0);
//   * every phase has its OWN buffer (4 buffers = one whole chunk resident, 76.5 KB: phases (0,*) only ever read block rows 1..8 and
//     keep just those): the sync before the LAST tap of a phase (its successor's patch has landed, everybody is past its
//     predecessor) refills the predecessor's buffer for its next use three phases later -- at least 3 taps of MFMA work ahead;
//   * wave tile 128 pixels x 64 channels, K split over the KS waves of a workgroup, weights as MFMA A fragments straight from global
//     memory into registers D taps ahead (the stride-1 packing, ops.pack_wfrag, indexed in the stride-2 tap order), pixel fragments
//     one ds_read_b128 per two MFMAs, pairwise K-group reduction through LDS, register-direct epilogue: all as in conv_wreg.inl;
//   * the downsample runs as a second, short pass after conv1's reduction: the phase-(0,0) patches of all chunks are staged again
//     (from L2) over the dead patch area, and every wave computes the FULL-K 1x1 product of the 128 / KS pixels it owns after the
//     reduction (8 MFMAs per chunk, the same MFMA sequence per output as the ring kernel: the downsample is bit-identical to it),
//     so no second reduction; both results are stored at the end.
// conv1 sums K in per-K-group partial sums: equal to the ring kernel's result to f32 rounding, not bit for bit (like conv_wreg.inl);
// the result does not depend on M, the group count or the launch geometry.
constexpr int s2w_tapidx(int i) {       // s2_tap_order step -> ky * 3 + kx (the packed weights' tap index)
    return i == 0 ? 0 : i == 1 ? 2 : i == 2 ? 6 : i == 3 ? 8 : i == 4 ? 1 : i == 5 ? 7 : i == 6 ? 3 : i == 7 ? 5 : 4;
}
constexpr int s2w_phase(int i) { return i < 4 ? 0 : i < 6 ? 1 : i < 8 ? 2 : 3; }
constexpr int s2w_dby(int i) { return (i == 0 || i == 1 || i == 4) ? 0 : 1; }      // ky != 0
constexpr int s2w_dbx(int i) { return (i == 0 || i == 2 || i == 6) ? 0 : 1; }      // kx != 0
constexpr int s2w_pbase(int ph) {       // LDS byte offset of a phase buffer's (virtual) block row 0
    return ph == 0 ? 0 : ph == 1 ? 9 * 2304 : ph == 2 ? 17 * 2304 : 25 * 2304;
}

template <int NN, int KS, int DW = 4>
__global__ __launch_bounds__(64 * NN * KS, 2) void conv3x3s2_wreg_kernel(ConvArgs p) {
#if defined(__HIP_DEVICE_COMPILE__)
    constexpr int PW = 18, NP = 162, NPIECE = 21;    // phase patch: 9 x 18 blocks (17 used per row), 128 B each, DMA'd in 1 KB pieces
    constexpr int ROWB = PW * 128;
    constexpr int NW = NN * KS;
    constexpr int P_INSTR = (NPIECE + NW - 1) / NW;   // every wave issues exactly P_INSTR pieces per patch (surplus ones repeat a piece):
                                                      // the vmcnt arithmetic below counts them
    constexpr int KK = 4 / KS;
    constexpr int R = (KK == 1 && DW > 2) ? 9 : 3, D = KK == 1 ? DW : 2;
    constexpr int PATCH_END = 34 * ROWB;              // 9 + 9 + 8 + 8 block rows
    constexpr int SS_BASE = PATCH_END > NW * 16384 ? PATCH_END : NW * 16384;
    static_assert(KS == 2 || KS == 4, "K split");
    static_assert(ROWB == 2304, "s2w_pbase");
    asm volatile("" ::"a"(0));                        // AGPR form of the builtin MFMAs (see conv_wreg.inl)

    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l31 = lane & 31, lhi = lane >> 5, lrow = lane >> 3;
    const int nw = wave % NN, kg = wave / NN;

    dbg_stamp(p, 0);
    const int tiles_x = p.Wo >> 4, tiles_y = p.Ho >> 3;
    const int g = blockIdx.y;
    const int tile = xcd_remap(blockIdx.x, p.ntm * p.ntn);
    const int tsp = w2c_fastdiv(tile, p.ntn, p.mg_ntn), tn = tile - tsp * p.ntn;          // (magic-number divisions: conv_wreg.inl)
    const int img = w2c_fastdiv(tsp, tiles_x * tiles_y, p.mg_txy);
    const int trem = tsp - img * (tiles_x * tiles_y);
    const int tyi = w2c_fastdiv(trem, tiles_x, p.mg_tx), txi = trem - tyi * tiles_x;
    const int oy0 = tyi * 8, ox0 = txi * 16;
    const int n0 = tn * (NN * 64) + nw * 64;
    const int nchunks = p.Cin >> 6, KT = nchunks * 9;

    // BN scale | shift of this wave's 64 channels, conv1's and the downsample's: parked in LDS (1 KB per channel block; the K groups
    // of a block write the same values) for the epilogue
    const bool dual = p.w2 != nullptr;
    float* const ssb = reinterpret_cast<float*>(smem + SS_BASE + nw * 1024);
    float* const ssw = ssb + lhi * 4;
    const unsigned lds_base = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(size_t)W2C_LPTR(smem));
    {
        // as dword LDS-DMA requests, the OLDEST vector-memory operations of the wave (every hand-counted wait below covers them): a
        // load + ds_write here put the load's round trip in front of the workgroup's first patch request (conv_wreg.inl, round 6)
        const unsigned voff = (unsigned)(g * p.Cout + n0 + lane) * 4u;
        auto park = [&](const float* src, unsigned dst) {
            const unsigned long long a = reinterpret_cast<unsigned long long>(src);
            const u32x4_t srd = {(unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)a), (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(a >> 32)),
                                 0x7FFFFFFFu, 0x00020000u};
            unsigned keep;
            asm volatile("s_nop 4\n\ts_mov_b32 %0, m0\n\ts_mov_b32 m0, %1\n\ts_nop 0\n\tbuffer_load_dword %2, %3, 0 offen lds\n\ts_mov_b32 m0, %0"
                         : "=&s"(keep) : "s"(dst), "v"(voff), "s"(srd) : "memory");
        };
        const unsigned d0 = lds_base + SS_BASE + (unsigned)nw * 1024u;
        park(p.scale, d0);
        park(p.shift, d0 + 256u);
        if (dual) {
            park(p.scale2, d0 + 512u);
            park(p.shift2, d0 + 768u);
        }
    }

    // ---- phase patches: LDS-DMA from inline asm, counted by hand ----
    const unsigned long long xaddr = reinterpret_cast<unsigned long long>(p.x) + (unsigned long long)g * p.Cin * 2;
    const unsigned x_bytes = (unsigned)((size_t)p.M * p.H * p.W * p.xcs * 2);
    const u32x4_t srd_x = {(unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)xaddr),
                           (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(xaddr >> 32)),
                           (unsigned)__builtin_amdgcn_readfirstlane((int)x_bytes), 0x00020000u};
    // block (b_r, b_c) of a patch <- input pixel (2 (oy0 - 1 + b_r) + py, 2 (ox0 - 1 + b_c) + px): the phase shift is a scalar offset.
    // Blocks above / left of the image are the padding (out-of-range offset -> zeros); below / right never leave it (H, W even).
    auto src_off = [&](int q) -> unsigned {
        const int b_r = q / PW, b_c = q - b_r * PW;
        const int chunk = (lane & 7) ^ ((b_c >> 1) & 7);            // swizzle keyed on the patch column, as in conv_wreg.inl
        const int by = oy0 - 1 + b_r, bx = ox0 - 1 + b_c;
        const bool ok = (by >= 0) & (bx >= 0) & (b_c <= 16);
        return ok ? (unsigned)(((img * p.H + 2 * by) * p.W + 2 * bx) * p.xcs * 2 + chunk * 16) : 0x80000000u;   // (< 2 GiB: the launcher)
    };
    auto piece_of = [&](int j) { const int pc = wave + NW * j; return pc < NPIECE ? pc : NPIECE - 1; };
    unsigned pa_off[P_INSTR];
#pragma unroll
    for (int j = 0; j < P_INSTR; ++j) pa_off[j] = src_off(piece_of(j) * 8 + lrow);
    auto dma = [&](unsigned dst, unsigned voff, unsigned soff) {
        unsigned keep;
        asm volatile("s_nop 4\n\ts_mov_b32 %0, m0\n\ts_mov_b32 m0, %1\n\ts_nop 0\n\tbuffer_load_dwordx4 %2, %3, %4 offen lds\n\ts_mov_b32 m0, %0"
                     : "=&s"(keep) : "s"(dst), "v"(voff), "s"(srd_x), "s"(soff) : "memory");
    };
    // patch of (chunk cc, phase ph) -> phase buffer `buf` (buf != ph only for the downsample's patches, see the last chunk below)
    auto issue_patch = [&](int cc, auto phc, auto bufc) {
        constexpr int ph = decltype(phc)::value, buf = decltype(bufc)::value;
        constexpr int py = ph < 2 ? 1 : 0, px = (ph == 0 || ph == 2) ? 1 : 0;
        const unsigned soff = (unsigned)((py * p.W + px) * p.xcs * 2 + cc * 128);
#pragma unroll
        for (int j = 0; j < P_INSTR; ++j) {
            // phases (0,*) keep block rows 1..8 only (their virtual row 0 is the previous buffer's last row): pieces 0 and 1 do not
            // exist there -- a wave that would issue one repeats its next piece instead (same bytes to the same place)
            int jj = j;
            if constexpr (py == 0 && P_INSTR > 1) { if (j + 1 < P_INSTR && piece_of(j) < 2) jj = j + 1; }
            const int pc = piece_of(jj);
            const int q = pc * 8 + lrow;
            const unsigned voff = jj == j ? pa_off[j] : pa_off[(j + 1 < P_INSTR) ? j + 1 : j];
            if (q < NP && (py == 1 || q >= PW)) dma(lds_base + s2w_pbase(buf) + (unsigned)pc * 1024u, voff, soff);
        }
    };

    // ---- weights: fragment-packed (stride-1 packing: K-step cc * 9 + ky * 3 + kx), coalesced 16-byte loads to registers ----
    const char* wbase = reinterpret_cast<const char*>(p.w) + (size_t)g * (p.Cout >> 5) * KT * 4096;
    const __amdgpu_buffer_rsrc_t rs_w = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(wbase), 0, (p.Cout >> 5) * KT * 4096, 0x00020000);
    const int wv = lane * 16 + kg * KK * 1024;
    const int nb0 = n0 >> 5;
    const int ws0 = nb0 * KT * 4096, ws1 = ws0 + KT * 4096;
    auto load_a = [&](u32x4_t (&A)[2][KK], int t) {
#pragma unroll
        for (int q = 0; q < KK; ++q) {
            A[0][q] = __builtin_bit_cast(u32x4_t, __builtin_amdgcn_raw_buffer_load_b128(rs_w, wv + q * 1024, ws0 + t * 4096, 0));
            A[1][q] = __builtin_bit_cast(u32x4_t, __builtin_amdgcn_raw_buffer_load_b128(rs_w, wv + q * 1024, ws1 + t * 4096, 0));
        }
    };

    f32x16_t acc[4][2];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

    // pixel-block order of the accumulators per K group (conv_wreg.inl): acc[0 .. 4/KS) are the blocks this wave owns after the reduction
    int blkoff[4];
    {
        auto fin = [&](int k) { return ((k & 1) ? 2 : 0) + ((k >> 1) ? 1 : 0); };
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            int blk;
            if constexpr (KS == 2) blk = (i + 2 * kg) & 3;
            else blk = fin(kg ^ (i == 0 ? 0 : i == 1 ? 2 : i == 2 ? 1 : 3));
            blkoff[i] = __builtin_amdgcn_readfirstlane(blk * (2 * ROWB));
        }
    }
    const int pc0 = l31 & 15;
    const int pp0 = (l31 >> 4) * PW + pc0;
    int boff[2][KK];                               // [dbx][slice]: this lane's 16 bytes inside a patch at block shift (0, dbx)
#pragma unroll
    for (int dbx = 0; dbx < 2; ++dbx)
#pragma unroll
        for (int q = 0; q < KK; ++q)
            boff[dbx][q] = (pp0 + dbx) * 128 + (((((kg * KK + q) << 1) | lhi) ^ (((pc0 + dbx) >> 1) & 7)) << 4);
    auto read_b = [&](bf16x8_t (&fb)[4], auto ic, auto qc) {
        constexpr int i = decltype(ic)::value, q = decltype(qc)::value;
        const char* r = smem + s2w_pbase(s2w_phase(i)) + s2w_dby(i) * ROWB + boff[s2w_dbx(i)][q];
#pragma unroll
        for (int b = 0; b < 4; ++b) fb[b] = *reinterpret_cast<const bf16x8_t*>(r + blkoff[b]);
    };
    auto mfma8 = [&](const u32x4_t (&A)[2][KK], const bf16x8_t (&fb)[4], int q) {
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j)
                acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8_t, A[j][q]), fb[i], acc[i][j], 0, 0, 0);
    };

    // ---- main loop.  VMEM queue of a wave, in issue order (P = P_INSTR pieces, A = one weight set of 2 KK loads):
    //   P(0,0) P(0,1) P(0,2) A x D | A A A [sync tap 3] P(cc,3) A A [sync 5] P(cc+1,0) A A [sync 7] P(cc+1,1) A [sync 8] P(cc+1,2) A | ...
    // the sync before the last tap of phase s needs the patch of phase s + 1: everything issued after that patch may stay in flight.
    using I0_ = std::integral_constant<int, 0>; using I1_ = std::integral_constant<int, 1>;
    using I2_ = std::integral_constant<int, 2>; using I3_ = std::integral_constant<int, 3>;
    u32x4_t AR[R][2][KK];
    bf16x8_t fb[2][4];
    issue_patch(0, I0_{}, I0_{});
    issue_patch(0, I1_{}, I1_{});
    issue_patch(0, I2_{}, I2_{});
#pragma unroll
    for (int d = 0; d < D; ++d) load_a(AR[d], s2w_tapidx(d));
    wait_vmcnt<2 * P_INSTR + 2 * KK * D>();
    pipeline_barrier();
    dbg_stamp(p, 1);
    read_b(fb[0], I0_{}, I0_{});

    auto chunk_body = [&](auto parc, int cc) {
        constexpr int PAR = decltype(parc)::value;
        const int t0 = cc * 9;
        const bool more = cc + 1 < nchunks;
        // LAST chunk: the phase buffers that fall free take the phase-(0,0) patches of the EARLIER chunks again (from L2) -- the
        // downsample pass behind the loop finds chunk c < nchunks - 1 in buffer c and the last chunk where the loop left it
        const bool ds1 = dual && nchunks >= 2, ds2 = dual && nchunks >= 3, ds3 = dual && nchunks >= 4;
        auto tap_body = [&](auto tapc) {
            constexpr int tap = decltype(tapc)::value;
            constexpr int ph = s2w_phase(tap);
            if constexpr (tap == 3) {                 // needs P(cc,1); younger: 5 A, P(cc,2)
                wait_vmcnt<5 * 2 * KK + P_INSTR>();
                pipeline_barrier();
                issue_patch(cc, I3_{}, I3_{});        // buffer (0,0): everybody is past the previous chunk
            } else if constexpr (tap == 5) {          // needs P(cc,2); younger: 6 A, P(cc,3)
                wait_vmcnt<6 * 2 * KK + P_INSTR>();
                pipeline_barrier();
                if (more) issue_patch(cc + 1, I0_{}, I0_{});
                else if (ds1) issue_patch(0, I3_{}, I0_{});
            } else if constexpr (tap == 7) {          // needs P(cc,3); younger: 4 A (+ the patch issued at tap 5)
                if (more || ds1) wait_vmcnt<4 * 2 * KK + P_INSTR>(); else wait_vmcnt<4 * 2 * KK>();
                pipeline_barrier();
                if (more) issue_patch(cc + 1, I1_{}, I1_{});
                else if (ds2) issue_patch(1, I3_{}, I1_{});
            } else if constexpr (tap == 8) {          // needs P(cc+1,0); younger: 3 A, P(cc+1,1)
                if (more) wait_vmcnt<3 * 2 * KK + P_INSTR>();
                pipeline_barrier();
                if (more) issue_patch(cc + 1, I2_{}, I2_{});
                else if (ds3) issue_patch(2, I3_{}, I2_{});
            }
            (void)ph;
            __builtin_amdgcn_sched_barrier(0);
            {
                constexpr int nx = tap + D;            // weight set D taps ahead: step nx of this chunk or nx - 9 of the next
                load_a(AR[nx % R], nx < 9 ? t0 + s2w_tapidx(nx % 9) : t0 + 9 + s2w_tapidx(nx % 9));   // past the end: zeros, unused
                __builtin_amdgcn_sched_group_barrier(0x020, 2 * KK, 0);
            }
            auto slice = [&](auto qc) {
                constexpr int q = decltype(qc)::value;
                constexpr int n = tap * KK + q;
                bf16x8_t (&cur)[4] = fb[(PAR + n) & 1];
                bf16x8_t (&nxt)[4] = fb[(PAR + n + 1) & 1];
                if constexpr (q + 1 < KK) read_b(nxt, std::integral_constant<int, tap>{}, std::integral_constant<int, (q + 1) % KK>{});
                else read_b(nxt, std::integral_constant<int, (tap + 1) % 9>{}, I0_{});     // (after the last tap of all: stale data, unused)
                mfma8(AR[tap % R], cur, q);
#pragma unroll
                for (int z = 0; z < 4; ++z) {
                    __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
                    __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);
                }
            };
            slice(I0_{});
            if constexpr (KK > 1) slice(I1_{});
        };
        tap_body(std::integral_constant<int, 0>{}); tap_body(std::integral_constant<int, 1>{}); tap_body(std::integral_constant<int, 2>{});
        tap_body(std::integral_constant<int, 3>{}); tap_body(std::integral_constant<int, 4>{}); tap_body(std::integral_constant<int, 5>{});
        tap_body(std::integral_constant<int, 6>{}); tap_body(std::integral_constant<int, 7>{}); tap_body(std::integral_constant<int, 8>{});
    };
    if constexpr ((9 * KK) % 2 == 0) {
        for (int cc = 0; cc < nchunks; ++cc) chunk_body(I0_{}, cc);
    } else {
        int cc = 0;
        for (; cc + 1 < nchunks; cc += 2) { chunk_body(I0_{}, cc); chunk_body(I1_{}, cc + 1); }
        if (cc < nchunks) chunk_body(I0_{}, cc);
    }

    // every weight set stays a live value to the end of the loop: the sync arithmetic above counts the loads of the taps past the last
    // K-step too, so the compiler must not drop them from a statically-last chunk body as dead
#pragma unroll
    for (int r = 0; r < D; ++r)                    // (the sets loaded by the last D taps: steps 9 .. 9 + D - 1 land in sets 0 .. D - 1)
#pragma unroll
        for (int q = 0; q < KK; ++q) asm volatile("" ::"v"(AR[r][0][q]), "v"(AR[r][1][q]));

    // ---- output addressing: this wave owns pixel blocks acc[0 .. CNT) ----
    constexpr int CNT = 4 / KS;
    unsigned eoff[CNT], e2off[CNT];
#pragma unroll
    for (int ii = 0; ii < CNT; ++ii) {
        const size_t pix = ((size_t)img * p.Ho + oy0 + blkoff[ii] / ROWB + (l31 >> 4)) * p.Wo + ox0 + pc0;
        eoff[ii] = (unsigned)(pix * p.ycs + (size_t)g * p.ygs + n0 + lhi * 8);
        e2off[ii] = (unsigned)(pix * p.y2cs + (size_t)g * p.Cout + n0 + lhi * 8);
    }
    // ---- downsample pass: the 1x1 / s2 conv of this wave's OWN pixel blocks, full K (no K split, so no reduction), on the phase-(0,0)
    // patches sitting in the phase buffers.  Accumulators in VGPRs (the loop's weight ring and fragment registers are dead; the
    // loop's 128 AGPR accumulators are not, and the kernel's AGPR budget is exactly those): MFMAs from inline asm. ----
    f32x16_t acc2[CNT][2];
#pragma unroll
    for (int ii = 0; ii < CNT; ++ii)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc2[ii][j][e] = 0.f;
    if (dual) {
        const char* w2base = reinterpret_cast<const char*>(p.w2) + (size_t)g * p.Cout * p.Cin * 2;
        const __amdgpu_buffer_rsrc_t rs_w2 = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(w2base), 0, p.Cout * p.Cin * 2, 0x00020000);
        auto load_a2 = [&](u32x4_t (&A2)[2][4], int c) {
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int kk = 0; kk < 4; ++kk)
                    A2[j][kk] = __builtin_bit_cast(u32x4_t, __builtin_amdgcn_raw_buffer_load_b128(
                        rs_w2, lane * 16 + kk * 1024, ((nb0 + j) * nchunks + c) * 4096, 0));
        };
        u32x4_t A2[2][2][4];
        load_a2(A2[0], 0);                               // (lands under the wait for the last patches)
        int b2off[4];
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) b2off[kk] = ROWB + (pp0 + 1) * 128 + ((((kk << 1) | lhi) ^ (((pc0 + 1) >> 1) & 7)) << 4);
        wait_vmcnt<0>();
        pipeline_barrier();                              // the downsample patches of the earlier chunks have landed everywhere
        if (p.n_split == 2) dbg_stamp(p, 2);
        auto pass = [&](const u32x4_t (&A)[2][4], int c) {
            const int buf = c + 1 < nchunks ? c : 3;
            const char* pb = smem + (buf == 0 ? s2w_pbase(0) : buf == 1 ? s2w_pbase(1) : buf == 2 ? s2w_pbase(2) : s2w_pbase(3));
            bf16x8_t f2[CNT][4];
#pragma unroll
            for (int ii = 0; ii < CNT; ++ii)
#pragma unroll
                for (int kk = 0; kk < 4; ++kk) f2[ii][kk] = *reinterpret_cast<const bf16x8_t*>(pb + blkoff[ii] + b2off[kk]);
#pragma unroll
            for (int kk = 0; kk < 4; ++kk)
#pragma unroll
                for (int ii = 0; ii < CNT; ++ii)
#pragma unroll
                    for (int j = 0; j < 2; ++j)
                        asm volatile("s_nop 1\n\tv_mfma_f32_32x32x16_bf16 %0, %1, %2, %0"        // (s_nop: an operand may be fresh from a VALU copy)
                                     : "+v"(acc2[ii][j]) : "v"(A[j][kk]), "v"(f2[ii][kk]));
            // the compiler does not know what the asm statements are: whatever VALU it places behind them (a copy of an accumulator at
            // the loop edge, the epilogue) must find the results written -- 18 wait states behind a 16-pass MFMA
            asm volatile("s_nop 15\n\ts_nop 7" ::: "memory");
        };
        for (int c = 0; c < nchunks; c += 2) {
            if (c + 1 < nchunks) load_a2(A2[1], c + 1);
            pass(A2[0], c);
            if (c + 1 < nchunks) {
                if (c + 2 < nchunks) load_a2(A2[0], c + 2);
                pass(A2[1], c + 1);
            }
        }
    } else {
        wait_vmcnt<0>();
    }
    pipeline_barrier();                              // every wave is done with the patch buffers: LDS is reused below
    if (p.n_split <= 1 || (p.n_split == 2 && !dual)) dbg_stamp(p, 2);

    // ---- K-group reduction through LDS (conv_wreg.inl) ----
    auto dump = [&](auto i0c, auto cntc, char* dst) {
        constexpr int I0 = decltype(i0c)::value, CNTD = decltype(cntc)::value;
#pragma unroll
        for (int i = 0; i < CNTD; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j) {
#pragma unroll
                for (int eg = 0; eg < 4; ++eg)
                    *reinterpret_cast<f32x4_t*>(dst + ((i * 2 + j) * 4 + eg) * 1024 + lane * 16) =
                        f32x4_t{acc[I0 + i][j][eg * 4], acc[I0 + i][j][eg * 4 + 1], acc[I0 + i][j][eg * 4 + 2], acc[I0 + i][j][eg * 4 + 3]};
                __builtin_amdgcn_sched_barrier(0);
            }
    };
    auto addin = [&](auto i0c, auto cntc, const char* src) {
        constexpr int I0 = decltype(i0c)::value, CNTD = decltype(cntc)::value;
#pragma unroll
        for (int i = 0; i < CNTD; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j) {
#pragma unroll
                for (int eg = 0; eg < 4; ++eg) {
                    const f32x4_t v = *reinterpret_cast<const f32x4_t*>(src + ((i * 2 + j) * 4 + eg) * 1024 + lane * 16);
#pragma unroll
                    for (int e = 0; e < 4; ++e) acc[I0 + i][j][eg * 4 + e] += v[e];
                }
                __builtin_amdgcn_sched_barrier(0);
            }
    };
    auto xr = [&](int kgx) { return smem + (kgx * NN + nw) * 16384; };
    dump(I2_{}, I2_{}, xr(kg));
    pipeline_barrier();
    addin(I0_{}, I2_{}, xr(kg ^ 1));
    if constexpr (KS == 4) {
        dump(I1_{}, I1_{}, xr(kg ^ 1));
        pipeline_barrier();
        addin(I0_{}, I1_{}, xr(kg ^ 3));
    }
    if (p.n_split == 3) dbg_stamp(p, 2);

    // ---- stores, register-direct: scale / shift (+ ReLU), bf16 pack, v_permlane32_swap pairs the half-waves' channel quads ----
    auto store_tiles = [&](const f32x16_t (&a)[2], const float* sc, const float* sh, bool relu, uint16_t* out, unsigned off) {
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int m = 0; m < 2; ++m) {
                const f32x4_t sc0 = *reinterpret_cast<const f32x4_t*>(sc + j * 32 + m * 16);
                const f32x4_t sc1 = *reinterpret_cast<const f32x4_t*>(sc + j * 32 + m * 16 + 8);
                const f32x4_t sh0 = *reinterpret_cast<const f32x4_t*>(sh + j * 32 + m * 16);
                const f32x4_t sh1 = *reinterpret_cast<const f32x4_t*>(sh + j * 32 + m * 16 + 8);
                f32x4_t v0 = f32x4_t{a[j][m * 8], a[j][m * 8 + 1], a[j][m * 8 + 2], a[j][m * 8 + 3]} * sc0 + sh0;
                f32x4_t v1 = f32x4_t{a[j][m * 8 + 4], a[j][m * 8 + 5], a[j][m * 8 + 6], a[j][m * 8 + 7]} * sc1 + sh1;
                if (relu) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) { v0[e] = fmaxf(v0[e], 0.f); v1[e] = fmaxf(v1[e], 0.f); }
                }
                const uint32_t a0 = pack_bf16x2(v0[0], v0[1]), a1 = pack_bf16x2(v0[2], v0[3]);
                const uint32_t b0 = pack_bf16x2(v1[0], v1[1]), b1 = pack_bf16x2(v1[2], v1[3]);
                const auto sa = __builtin_amdgcn_permlane32_swap(a0, b0, false, false);
                const auto sb = __builtin_amdgcn_permlane32_swap(a1, b1, false, false);
                *reinterpret_cast<uint4*>(out + off + j * 32 + m * 16) = make_uint4(sa[0], sb[0], sa[1], sb[1]);
                __builtin_amdgcn_sched_barrier(0);
            }
    };
#pragma unroll
    for (int ii = 0; ii < CNT; ++ii) store_tiles(acc[ii], ssw, ssw + 64, p.relu != 0, reinterpret_cast<uint16_t*>(p.y), eoff[ii]);
    if (dual) {
#pragma unroll
        for (int ii = 0; ii < CNT; ++ii) store_tiles(acc2[ii], ssw + 128, ssw + 192, false, p.y2, e2off[ii]);
    }
    dbg_stamp(p, 3);
#endif
}

template <int NN, int KS, int DW = 4>
int launch_s2wreg(ConvArgs& a, int groups, hipStream_t s) {
    if (a.ks != 3 || a.stride != 2 || a.Cin % 64 != 0 || a.Cin > 256 || a.Cout % (NN * 64) != 0 || a.Ho % 8 != 0 || a.Wo % 16 != 0 ||
        (a.H & 1) || (a.W & 1) || a.res || a.y_f32 || a.y8 || !a.y || a.ws)
        return W2C_E_ARG;
    a.ntm = a.M * (a.Ho / 8) * (a.Wo / 16);
    a.ntn = a.Cout / (NN * 64);
    if ((long)a.ntm * a.ntn * ((a.Ho / 8) * (a.Wo / 16) > a.ntn ? (a.Ho / 8) * (a.Wo / 16) : a.ntn) >= (1ll << 32)) return W2C_E_ARG;   // (fast-division range)
    a.mg_ntn = w2c_magic((unsigned)a.ntn);
    a.mg_qn = 0;
    a.mg_tx = w2c_magic((unsigned)(a.Wo / 16));
    a.mg_txy = w2c_magic((unsigned)((a.Ho / 8) * (a.Wo / 16)));
    constexpr int patch = 34 * 2304;
    constexpr int xchg = NN * KS * 16384;
    constexpr int lds = (patch > xchg ? patch : xchg) + NN * 1024;
    static_assert(lds <= 80 * 1024 || NN * KS > 4, "two workgroups per CU");
    static std::atomic<unsigned long long> attr_mask{0};
    int dev = 0;
    (void)hipGetDevice(&dev);
    if (!((attr_mask.load(std::memory_order_acquire) >> (dev & 63)) & 1ull)) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&conv3x3s2_wreg_kernel<NN, KS, DW>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        attr_mask.fetch_or(1ull << (dev & 63), std::memory_order_release);
    }
    hipLaunchKernelGGL((conv3x3s2_wreg_kernel<NN, KS, DW>), dim3(a.ntm * a.ntn, groups), dim3(64 * NN * KS), lds, s, a);
    return w2c_launch_status();
}
