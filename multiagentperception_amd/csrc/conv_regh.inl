// =====================================================================================================
// Layer1 kernel, round-4 form: 3x3 stride-1, Cin = Cout = 64 per group, weights stationary in registers, TWO waves per SIMD
// (included by conv_igemm.hip, inside its namespace).
//
// Why another layer1 kernel.  conv3x3_c64_regw_kernel keeps a group's whole 64 x 576 weight matrix in ONE wave (288 registers), so
// a SIMD hosts one wave and everything that is not an MFMA -- tile top, the wait for the residual, the 2.4 k-cycle epilogue: 38 %
// of a wave's time (tools/regw_phases.py) -- leaves the matrix pipe idle; hiding it inside one instruction stream means a
// hand-interleaved two-pass loop.  Here the hardware does the interleaving:
//   * a wave owns HALF of the output channels (32 x 576 weights = 144 VGPRs + 32 accumulator AGPRs: <= 256 registers), so TWO waves
//     share a SIMD and one's epilogue / DMA issue / waits run under the other's MFMAs;
//   * workgroup = 4 waves on ONE 8 x 16-pixel tile: wave (ph, ch) computes pixel rows [4 ph, +4) x channels [32 ch, +32); the
//     10 x 18-pixel halo patch is staged ONCE for the four of them (23 LDS-DMA pieces of 1 KB, six per wave; halo = out-of-range
//     offsets -> zeros), double-buffered across tiles: ONE workgroup barrier per tile; two workgroups per CU (49 KB of LDS each) are
//     not synchronised with each other, which is what de-phases the two waves of a SIMD;
//   * the workgroup is a persistent worker on a contiguous, column-major run of tiles (vertical neighbours share 2 of 10 patch rows
//     in L2);
//   * weights come fragment-packed (w2c_pack_wfrag_bf16): 36 perfectly coalesced 16-byte loads per lane straight into VGPRs -- the
//     older form's 6.7 us LDS-staged weight prologue is gone;
//   * B fragments (pixels): one ds_read_b128 per MFMA (the 64-channel wave of the older form needed 0.5): 4 SIMDs x 1 read per 32
//     cycles = 50 % of the LDS read rate at full MFMA rate;
//   * LDS-DMA is issued from inline asm (invisible to the compiler's vmcnt bookkeeping, waited for with hand-counted vmcnt: a
//     compiler-visible LDS-DMA puts vmcnt(0) in front of every later LDS read), spread through the MFMA stream;
//   * epilogue register-direct: residual loaded in the STORE layout (16 B per lane), brought into the accumulator layout by
//     v_permlane32_swap (its own inverse), scale / shift / + residual / bf16 pack / ReLU on the packed pairs / swap / 16-byte stores.
// Same MFMA sequence per output element (K walked tap-major, 16 channels per step) and the same epilogue arithmetic in the same
// order as conv3x3_c64_regw_kernel and the ring kernels: results are bit-identical (tests/test_kernels_gpu.py).
// VMEM queue of a wave per tile, in issue order: [6 patch pieces of tile t+1 (asm, K-steps 2-12)] [4 residual loads of tile t (K-steps
// 14-17)] [4 stores of tile t]; the top of tile t+1 waits vmcnt(4): everything but the stores.
// Measured (cfg 2, both trunks, tools/regw_phases.py): per tile and wave ~450 cycles at the tile top + 3.5 k in the MFMA loop (2.3 k of
// MFMA) + 2.2 k of epilogue = 6.5-6.7 k against 9.3-9.4 k per 64-pixel tile of the one-wave-per-SIMD form; 48 / 53 us per launch
// against 55 / 61 in the forward.  What bounds it now is one wave's serial instruction stream (loop + epilogue), not the matrix pipe
// (4.6 k per tile pair): ~540 instructions per wave and tile, of which 72 are MFMAs.
template <bool HAS_RES, int NBUF = 2, int RES_STEP = 14, int OPT = 0>
__global__ __launch_bounds__(256, 2) void conv3x3_c64_regh_kernel(ConvArgs p) {
#if defined(__HIP_DEVICE_COMPILE__)
    constexpr int PW = 18;                         // patch width (16 + 2 halo); 10 rows
    constexpr int NP = 180;                        // patch pixels
    constexpr int PATCH_BYTES = 24 * 1024;         // 4 waves x 6 pieces x 1 KB (pieces 22.5 .. 24 are pad: zeros)
    constexpr int PF = NBUF - 1;                   // patches in flight ahead of the tile being computed
    static_assert(NBUF == 2 || NBUF == 3, "patch ring depth");
    constexpr int BD = (OPT & 1) ? 2 : 1;          // B fragments are read BD K-steps ahead of their MFMAs (ring of BD + 1 register sets)
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l31 = lane & 31, lhi = lane >> 5;
    const int ph = wave >> 1, ch = wave & 1;
    const int g = blockIdx.y;
    span_stamp(p, false);
    if constexpr ((OPT & 6) != 0) {
        // the two waves of a SIMD run the same code from the same start: left alone they stay in phase (both in their MFMA loop, then
        // both in their epilogue).  HW_ID[3:0] = the wave's slot on its SIMD: odd slots get a static priority (OPT & 2) or a late start (OPT & 4)
        const unsigned slot = __builtin_amdgcn_s_getreg((3 << 11) | (0 << 6) | 4) & 1u;      // hwreg(HW_REG_HW_ID, 0, 4)
        if (OPT & 2) { if (slot) __builtin_amdgcn_s_setprio(1); }
        if (OPT & 4) { if (slot) { for (int i = 0; i < 24; ++i) __builtin_amdgcn_s_sleep(127); } }     // ~24 x 127 x 64 clocks
    }

    // ---- tiles of this workgroup: a contiguous run, XCD-contiguous across the grid ----
    const int ntx = p.W >> 4, nty = p.H >> 3, tpi = ntx * nty;
    const int T = p.M * tpi;
    const int nwg = gridDim.x;
    const int b = blockIdx.x;
    const int logical = (nwg % 8 == 0) ? (b & 7) * (nwg >> 3) + (b >> 3) : b;
    // (round 6: multiply-high divisions -- (logical + 1) * T < 2^31 is checked by the launcher; two 64-bit run-time divisions here were
    //  ~0.3 us in front of the workgroup's first memory request)
    const int t_begin = __builtin_amdgcn_readfirstlane(w2c_fastdiv2(logical * T, nwg, p.mg_ntn));
    const int t_end = __builtin_amdgcn_readfirstlane(w2c_fastdiv2((logical + 1) * T, nwg, p.mg_ntn));
    if (t_begin >= t_end) { span_stamp(p, true); return; }      // workgroup-uniform: nobody is left waiting at a barrier

    // ---- weights -> registers: this wave's 32 channels x 576, fragment order (block (g, ch, tap, kc) = 1 KB, lane-linear) ----
    u32x4_t wr[9][4];
    {
        const uint16_t* wf = p.w + ((size_t)(g * 2 + ch) * 36) * 512 + lane * 8;
#pragma unroll
        for (int tap = 0; tap < 9; ++tap)
#pragma unroll
            for (int kc = 0; kc < 4; ++kc) wr[tap][kc] = *reinterpret_cast<const u32x4_t*>(wf + (tap * 4 + kc) * 512);
    }
    // BN scale | shift of the group, per channel quad {scale[4], shift[4]}: 512 B behind the patch buffers (read after the first barrier)
    float* const ss = reinterpret_cast<float*>(smem + NBUF * PATCH_BYTES);
    if (tid < 64) {
        ss[(tid >> 2) * 8 + (tid & 3)] = p.scale[g * 64 + tid];
        ss[(tid >> 2) * 8 + 4 + (tid & 3)] = p.shift[g * 64 + tid];
    }

    // ---- patch DMA (inline asm).  Piece j = wave + 4 i moves pixels q = 8 j + lane / 8 (q = row * 18 + col), lane % 8 = LDS chunk
    // position; the bank swizzle (chunk c of a pixel in patch column x sits at c ^ ((x >> 1) & 7)) is applied to the SOURCE chunk ----
    const unsigned lds_base = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(size_t)W2C_LPTR(smem));
    // The descriptor's base is shifted one row + one pixel BEFORE the group's first channel, so that the wave-uniform tile offset
    // (first patch pixel = (y0 - 1, x0 - 1)) is never negative and can ride in SOFFSET; the per-lane VOFFSET is the lane's offset inside
    // the patch, or 0x80000000 for a halo pixel outside the image / a pad lane (range-checked against num_records -> zeros).
    const unsigned shift_b = (unsigned)((p.W + 1) * p.xcs * 2);
    const unsigned long long xaddr = reinterpret_cast<unsigned long long>(p.x) + (unsigned long long)g * 128 - shift_b;
    const unsigned x_bytes = (unsigned)((size_t)p.M * p.H * p.W * p.xcs * 2 - (size_t)g * 128);
    const u32x4_t srd_x = {(unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)xaddr),
                           (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(xaddr >> 32)),
                           (unsigned)__builtin_amdgcn_readfirstlane((int)(x_bytes + shift_b)), 0x00020000u};
    int off_rel[6];
    unsigned hmask = 0;                            // one register: bit i (+ 6 k) = piece i is a halo pixel of class k (top | bottom | left | right | pad)
#pragma unroll
    for (int i = 0; i < 6; ++i) {
        const int q = 8 * (wave + 4 * i) + (lane >> 3);
        const int dy = (q * 3641) >> 16, dx = q - dy * PW;          // q / 18, q % 18 for q < 192
        const int chunk = (lane & 7) ^ ((dx >> 1) & 7);
        off_rel[i] = ((dy * p.W + dx) * p.xcs + chunk * 8) * 2;
        hmask |= (dy == 0 ? 1u : 0u) << i;
        hmask |= (dy == 9 ? 1u : 0u) << (i + 6);
        hmask |= (dx == 0 ? 1u : 0u) << (i + 12);
        hmask |= (dx == 17 ? 1u : 0u) << (i + 18);
        hmask |= (q >= NP ? 1u : 0u) << (i + 24);
    }
    unsigned pbase = 0;                            // SOFFSET of the next patch: byte offset of its first pixel from the shifted base
    unsigned vo[6];                                // VOFFSETs of this wave's six pieces of the next patch
    auto patch_piece = [&](int i, int buf) {
        const unsigned dst = lds_base + (unsigned)buf * PATCH_BYTES + (unsigned)(wave + 4 * i) * 1024u;
        asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, %3 offen lds"
                     :: "s"(dst), "v"(vo[i]), "s"(srd_x), "s"(pbase) : "memory", "m0");
    };
    auto patch_setup = [&](int img, int y0, int x0, bool live) {
        pbase = (unsigned)((((img * p.H + y0) * p.W) + x0) * p.xcs * 2);      // (y0 - 1, x0 - 1) from the shifted base
        // wave-uniform selection of the lane masks: which halo classes are outside the image for this tile
        const unsigned sel = 0x3F000000u | (y0 == 0 ? 0x3Fu : 0u) | (y0 + 8 == p.H ? 0xFC0u : 0u) | (x0 == 0 ? 0x3F000u : 0u) |
                             (x0 + 16 == p.W ? 0xFC0000u : 0u);
        const unsigned m = hmask & sel;
        const unsigned pbad = !live ? 0x3Fu : ((m | (m >> 6) | (m >> 12) | (m >> 18) | (m >> 24)) & 0x3Fu);
#pragma unroll
        for (int i = 0; i < 6; ++i) vo[i] = ((pbad >> i) & 1u) ? 0x80000000u : (unsigned)off_rel[i];
    };

    // ---- output / residual addressing (store layout): lane (l31, lhi) -> pixel (4 ph + 2 pt + l31 / 16, l31 % 16), 8 channels
    // 32 ch + 16 jp + 8 lhi .. + 7 per (pt, jp) ----
    const size_t y_bytes = (size_t)p.M * p.H * p.W * p.ycs * 2;
    const __amdgpu_buffer_rsrc_t rs_y = __builtin_amdgcn_make_buffer_rsrc(p.y, 0, (int)y_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_r = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint16_t*>(HAS_RES ? p.res : p.x), 0,
                                                                          (int)(HAS_RES ? y_bytes : (size_t)x_bytes), 0x00020000);
    const int e_px = (4 * ph + (l31 >> 4)) * p.W + (l31 & 15);                              // + 2 pt W
    const unsigned y_lane = (unsigned)((e_px * p.ycs + g * 64 + ch * 32 + lhi * 8) * 2);

    // B-fragment geometry: MFMA pixel block pt = rows 4 ph + 2 pt, + 1 of the tile; lane's pixel = (l31 >> 4, l31 & 15)
    const int qb0 = (4 * ph + (l31 >> 4)) * PW + (l31 & 15);
    int foff[3][4];
#pragma unroll
    for (int kx = 0; kx < 3; ++kx)
#pragma unroll
        for (int kc = 0; kc < 4; ++kc) foff[kx][kc] = (kc << 5) ^ ((((((l31 & 15) + kx) >> 1) ^ lhi) & 7) << 4);
    const char* const ssd = reinterpret_cast<const char*>(ss) + (ch * 8 + lhi) * 32;        // + j * 64: quad 8 ch + 2 j + lhi

    int img = w2c_fastdiv2(t_begin, tpi, p.mg_txy), y0, x0;
    {
        const int r = t_begin - img * tpi;
        const int tx = w2c_fastdiv2(r, nty, p.mg_tx);   // column-major: consecutive tiles are vertically adjacent
        x0 = tx * 16;
        y0 = (r - tx * nty) * 8;
    }
    auto advance = [&](int& im, int& yy, int& xx) {
        yy += 8;
        if (yy == p.H) { yy = 0; xx += 16; if (xx == p.W) { xx = 0; ++im; } }
    };
    // prologue: patches t_begin .. t_begin + PF - 1; (in, yn, xn) = the next tile whose patch is to be issued
    int in = img, yn = y0, xn = x0;
#pragma unroll
    for (int k = 0; k < PF; ++k) {
        patch_setup(in, yn, xn, t_begin + k < t_end);
#pragma unroll
        for (int i = 0; i < 6; ++i) patch_piece(i, k);
        advance(in, yn, xn);
    }
    int cur = 0;
    // debug (p.dbg, tools/regw_phases.py): wave 0's cycles in  vmcnt wait | barrier | MFMA loop | epilogue, 3 wall-clock stamps
    unsigned long long dph[4] = {0, 0, 0, 0};
    const unsigned long long wall0 = p.dbg ? wall_clock64() : 0;
    long long dtp = p.dbg ? clock64() : 0;
    auto stamp = [&](int i) {
        if (p.dbg) { const long long n = clock64(); dph[i] += (unsigned long long)(n - dtp); dtp = n; }
    };

    for (int t = t_begin; t < t_end; ++t) {
        const char* const pc = smem + cur * PATCH_BYTES;
        int nxt = cur + PF; if (nxt >= NBUF) nxt -= NBUF;
        if (t != t_begin) advance(img, y0, x0);
        // patch(t + PF): base offset + halo mask; past the run every lane is off (zeros into the idle buffer)
        patch_setup(in, yn, xn, t + PF < t_end);
        advance(in, yn, xn);
        // This wave's pieces of patch(t) have landed -- vmcnt retires in issue order, so allow exactly the operations issued after
        // them that may still be outstanding: per tile body 6 pieces of patch(t + PF) and 4 stores (the residual loads of a tile have
        // returned before its stores are issued: the stores carry their data) -- then the workgroup barrier: the whole patch is there,
        // and every wave is done reading the buffer of patch(t - 1), which the pieces of patch(t + PF) issued below overwrite.
        if (t == t_begin) wait_vmcnt<(PF - 1) * 6>();
        else if (PF == 2 && t == t_begin + 1) wait_vmcnt<6 + 4>();
        else wait_vmcnt<(PF == 1 ? 4 : 2 * 4 + 6)>();
        stamp(0);
        pipeline_barrier();
        stamp(1);

        f32x16_t acc[2];
        // fragment address of (tap, kc) for pixel q = qb + ky * 18 + kx:  q * 128 + ((2 kc | lhi) ^ ((x >> 1) & 7)) * 16, x = patch column
        //   = [pc + qb * 128]  +  foff[kx][kc]  +  (ky * 18 + kx) * 128 (an immediate):   12 loop-invariant lane offsets, one add per read
        // pixel block 1 is two patch rows below block 0: + 2 * 18 * 128 bytes, an immediate too
        const char* const fbase0 = pc + qb0 * 128;
        auto frag = [&](int tap, int kc, int pt) {
            const int ky = tap / 3, kx = tap - 3 * ky;
            return *reinterpret_cast<const u32x4_t*>(fbase0 + foff[kx][kc] + ((ky + 2 * pt) * PW + kx) * 128);
        };
        u32x4_t bx[BD + 1][2];
        uint4 rres[2][2];
        const int tile_pix = (img * p.H + y0) * p.W + x0;
#pragma unroll
        for (int d = 0; d < BD; ++d) {
            bx[d][0] = frag(d >> 2, d & 3, 0);
            bx[d][1] = frag(d >> 2, d & 3, 1);
        }
#pragma unroll
        for (int step = 0; step < 36; ++step) {
            const int tap = step >> 2, kc = step & 3, cb = step % (BD + 1);
            if (step + BD < 36 && !(OPT & 8)) {            // (OPT & 8 / 16 / 32: timing ablations, wrong results)
                bx[(step + BD) % (BD + 1)][0] = frag((step + BD) >> 2, (step + BD) & 3, 0);
                bx[(step + BD) % (BD + 1)][1] = frag((step + BD) >> 2, (step + BD) & 3, 1);
            }
            if (step >= 2 && step <= 12 && (step & 1) == 0 && !(OPT & 16)) patch_piece((step - 2) >> 1, nxt);
            if (HAS_RES && step >= RES_STEP && step < RES_STEP + 4) {      // residual(t), store layout: 4 x 16 B per lane
                const int k = step - RES_STEP, pt = k >> 1, jp = k & 1;
                rres[pt][jp] = __builtin_bit_cast(uint4, __builtin_amdgcn_raw_buffer_load_b128(
                                   rs_r, y_lane + jp * 32, (tile_pix + 2 * pt * p.W) * p.ycs * 2, 0));
            }
            // register classes pinned here: taps 0-5 of the weights (96 registers) + the 32 accumulators live in AGPRs, taps 6-8 (48) in
            // VGPRs -- a 2-waves-per-SIMD kernel that touches AGPRs gets a fixed 128 + 128 split from the backend
            if (step == 0) {
                asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, 0" : "=a"(acc[0]) : "a"(wr[tap][kc]), "v"(bx[cb][0]));
                asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, 0" : "=a"(acc[1]) : "a"(wr[tap][kc]), "v"(bx[cb][1]));
            } else if (tap < 6) {
                asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+a"(acc[0]) : "a"(wr[tap][kc]), "v"(bx[cb][0]));
                asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+a"(acc[1]) : "a"(wr[tap][kc]), "v"(bx[cb][1]));
            } else {
                asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+a"(acc[0]) : "v"(wr[tap][kc]), "v"(bx[cb][0]));
                asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+a"(acc[1]) : "v"(wr[tap][kc]), "v"(bx[cb][1]));
            }
        }
        // the MFMAs are opaque to the compiler's hazard recogniser: cover the XDL-write -> VALU-read wait states by hand
        asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");
        stamp(2);

        // ---- epilogue, register-direct (same arithmetic, same order as conv3x3_c64_regw2_kernel).  The four channel quads' scale /
        // shift come out of LDS ONCE per tile, all eight reads in flight together (read next to their use they cost eight separate
        // lgkmcnt(0) round trips beside the partner wave's fragment reads: ~1 k cycles of a 2.2 k-cycle epilogue) ----
        // (two quads = 16 registers at a time: all four do not fit beside the residual)
        // ReLU on the packed bf16 pairs as ONE v_pk_max_i16 against a wave-uniform operand: 0 (ReLU) or 0x8000 per half (no-op)
        const s16x2_t relu_lo = p.relu ? s16x2_t{0, 0} : s16x2_t{(short)-32768, (short)-32768};
#pragma unroll
        for (int jp = 0; jp < 2; ++jp) {
            asm volatile("" : "+a"(acc[0]), "+a"(acc[1])::"memory");      // half of the accumulators out of the AGPRs at a time
            f32x4_t scv[2], shv[2];
#pragma unroll
            for (int jj = 0; jj < 2; ++jj) {
                scv[jj] = *reinterpret_cast<const f32x4_t*>(ssd + (2 * jp + jj) * 64);
                shv[jj] = *reinterpret_cast<const f32x4_t*>(ssd + (2 * jp + jj) * 64 + 16);
            }
#pragma unroll
            for (int pt = 0; pt < 2; ++pt) {
                const int so = (tile_pix + 2 * pt * p.W) * p.ycs * 2;
                uint32_t pk[2][2];
                uint32_t ra[2] = {0, 0}, rb[2] = {0, 0};
                if constexpr (HAS_RES) {
                    const uint4 r = rres[pt][jp];
                    const auto sa = __builtin_amdgcn_permlane32_swap(r.x, r.z, false, false);
                    const auto sb = __builtin_amdgcn_permlane32_swap(r.y, r.w, false, false);
                    ra[0] = sa[0]; ra[1] = sa[1]; rb[0] = sb[0]; rb[1] = sb[1];
                }
#pragma unroll
                for (int jj = 0; jj < 2; ++jj) {
                    const int j = 2 * jp + jj;
                    f32x4_t a = f32x4_t{acc[pt][j * 4], acc[pt][j * 4 + 1], acc[pt][j * 4 + 2], acc[pt][j * 4 + 3]} * scv[jj] + shv[jj];
                    if constexpr (HAS_RES)
                        a += f32x4_t{__uint_as_float(ra[jj] << 16), __uint_as_float(ra[jj] & 0xFFFF0000u),
                                     __uint_as_float(rb[jj] << 16), __uint_as_float(rb[jj] & 0xFFFF0000u)};
#pragma unroll
                    for (int e = 0; e < 2; ++e) {
                        const s16x2_t h = __builtin_bit_cast(s16x2_t, pack_bf16x2(a[2 * e], a[2 * e + 1]));
                        pk[jj][e] = __builtin_bit_cast(uint32_t, __builtin_elementwise_max(h, relu_lo));
                    }
                }
                const auto sa = __builtin_amdgcn_permlane32_swap(pk[0][0], pk[1][0], false, false);
                const auto sb = __builtin_amdgcn_permlane32_swap(pk[0][1], pk[1][1], false, false);
                const u32x4_t o = {sa[0], sb[0], sa[1], sb[1]};
                if (!(OPT & 32) || t == t_begin) __builtin_amdgcn_raw_buffer_store_b128(o, rs_y, y_lane + jp * 32, so, 0);
            }
        }
        asm volatile("" ::: "memory");
        stamp(3);
        cur = cur + 1 == NBUF ? 0 : cur + 1;
    }
    if (p.dbg && lane == 0 && (wave == 0 || wave == 3)) {
        unsigned long long* d = p.dbg + (((size_t)blockIdx.y * gridDim.x + blockIdx.x) * 2 + (wave == 3)) * 8;
        for (int i = 0; i < 4; ++i) d[i] = dph[i];
        d[4] = wall0; d[5] = wall0; d[6] = wall_clock64(); d[7] = 1;
    }
    span_stamp(p, true);
#endif
}

template <bool HAS_RES, int NBUF = 2, int RES_STEP = 14, int OPT = 0>
int launch_regh(ConvArgs& a, int groups, hipStream_t s) {
    constexpr int lds = NBUF * 24 * 1024 + 512;
    static std::atomic<unsigned long long> attr_mask{0};
    static int n_cu[64] = {0};
    int dev = 0;
    (void)hipGetDevice(&dev);
    if (!((attr_mask.load(std::memory_order_acquire) >> (dev & 63)) & 1ull)) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&conv3x3_c64_regh_kernel<HAS_RES, NBUF, RES_STEP, OPT>),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, lds);
        hipDeviceProp_t prop;
        n_cu[dev & 63] = (hipGetDeviceProperties(&prop, dev) == hipSuccess && prop.multiProcessorCount > 0) ? prop.multiProcessorCount : 256;
        attr_mask.fetch_or(1ull << (dev & 63), std::memory_order_release);
    }
    const long tiles = (long)a.M * (a.H / 8) * (a.W / 16);
    // one group: two 4-wave workgroups per CU.  Two groups (both trunks' layer1): ONE workgroup per CU in total -- these launches move
    // 168 / 252 MB at 3.5-4.6 TB/s, the second wave per SIMD buys 2 % of the launch (53.2 -> 54.3 us) and costs power the rest of the
    // forward pays for: 1.054 -> 1.040 ms per forward with half the workgroups (profiles/r04_s2_front_c64.txt, W2C_REGH_WGS A/B)
    long wgs = groups == 1 ? 2L * n_cu[dev & 63] : (n_cu[dev & 63] + groups - 1) / groups;
    const int opt = w2c_option(W2C_OPT_REGH_WGS);
    if (opt > 0) wgs = opt;
    if (wgs > tiles) wgs = tiles;
    if ((wgs + 1) * tiles >= (1ll << 31)) return W2C_E_ARG;                  // (the kernel's 32-bit tile-run arithmetic)
    a.mg_ntn = w2c_magic_floor((unsigned)wgs);                                // divisors of the kernel's run decode: workgroups,
    a.mg_txy = w2c_magic_floor((unsigned)((a.H / 8) * (a.W / 16)));           // tiles per image,
    a.mg_tx = w2c_magic_floor((unsigned)(a.H / 8));                           // tile rows (column-major tile order)
    hipLaunchKernelGGL((conv3x3_c64_regh_kernel<HAS_RES, NBUF, RES_STEP, OPT>), dim3((unsigned)wgs, groups), dim3(256), lds, s, a);
    return w2c_launch_status();
}

// `a.w` must be in w2c_pack_wfrag_bf16 order
int launch_regh_any(ConvArgs& a, int groups, hipStream_t s) {
    if (a.ks != 3 || a.stride != 1 || a.Cin != 64 || a.Cout != 64 || a.H % 8 != 0 || a.W % 16 != 0 || a.ygs != 64 || a.y8 || !a.y || a.y_f32 ||
        a.ws)
        return W2C_E_ARG;
    if ((size_t)a.M * a.H * a.W * a.xcs * 2 >= (1ull << 31) || (size_t)a.M * a.H * a.W * a.ycs * 2 >= (1ull << 31)) return W2C_E_ARG;
    // A/B forms, all bit-identical (measured, cfg 2: profiles/r04_layer1_regh.txt): the default is a 2-deep patch ring, B fragments one
    // K-step ahead, residual loads at K-step 14.  (The kernel's OPT bits 8 / 16 / 32 are timing ablations with wrong results: not reachable.)
    switch (w2c_option(W2C_OPT_REGH_FORM)) {
        case 1: return a.res ? launch_regh<true, 2, 14, 1>(a, groups, s) : launch_regh<false, 2, 14, 1>(a, groups, s);   // B fragments 2 steps ahead
        case 2: return a.res ? launch_regh<true, 2, 14, 2>(a, groups, s) : launch_regh<false, 2, 14, 2>(a, groups, s);   // static priority by wave slot
        case 5: return a.res ? launch_regh<true, 2, 4>(a, groups, s) : launch_regh<false>(a, groups, s);                 // residual at K-step 4
        case 6: return a.res ? launch_regh<true, 2, 28>(a, groups, s) : launch_regh<false>(a, groups, s);                // ... 28
        case 7: return a.res ? launch_regh<true, 3, 14>(a, groups, s) : launch_regh<false, 3>(a, groups, s);             // 3-deep patch ring
        default: return a.res ? launch_regh<true>(a, groups, s) : launch_regh<false>(a, groups, s);
    }
}
