// =====================================================================================================
// Front of the FIRST stride-2 BasicBlock (layer2.0 of the third-party resnet18, backbone.py:66-69): conv1 3x3 / s2 / p1 (64 -> 128,
// BN, ReLU) + the block's 1x1 / s2 downsample (64 -> 128, BN) from one pass over the input, on the structure of conv_regh.inl
// (included by conv_igemm.hip, inside its namespace; round 4, third session).
//
// Why another stride-2 kernel.  The polyphase ring kernel (conv3x3s2_patch_kernel) and the weights-to-registers form (conv_s2wreg.inl)
// spend a third of a workgroup's life in a prologue with nothing to overlap (one memory round trip for the first patch) and another
// sixth in the epilogue; with Cin = 64 a tile's whole K is 9 taps, so there is no long loop to amortise them over: 31-33 us per trunk
// alone (SQ: MFMA busy 19 %, waves parked 47 %), 59 us for the two trunks' fronts side by side at the head of the two launch chains
// = 2.8 TB/s on a launch that moves 168 MB.  This layer's weights are small enough to be STATIONARY: a wave that owns 32 of the 128
// output channels holds 32 x 576 (conv1) + 32 x 64 (downsample) weights in 160 registers, so the kernel can be what layer1's is:
//   * persistent workgroups (two per CU, 77 KB of LDS each, not synchronised with each other: one's epilogue runs under the other's
//     MFMAs) on contiguous runs of 8 x 8-pixel OUTPUT tiles; workgroup = 4 waves, wave w = output channels [32 w, +32) of all 64 pixels
//     (2 MFMA pixel blocks of 4 x 8);
//   * the (17 x 17)-pixel input patch of a tile is staged ONCE for the four waves, as the four PHASE images of the stride-2 conv
//     (phase (py, px) = input pixels (2 by + py, 2 bx + px): 9 x 9, 9 x 8, 8 x 9, 8 x 8 blocks -- a 3x3 / s2 conv is a 2x2 / s1 conv
//     over them, so a tap's fragment read is a stride-1 read inside one phase), 37 LDS-DMA pieces of 1 KB issued from inline asm with
//     hand-counted vmcnt, double-buffered across tiles: patch(t + 1) travels under the MFMAs of tile t, ONE barrier per tile;
//   * K order = every stride-2 kernel's here (s2_tap order, 16 channels per step): conv1's and the downsample's results are
//     bit-identical to w2c_conv_s2_block's (tests/test_kernels_gpu.py);
//   * the downsample is 4 more K-steps on the phase-(0, 0) fragments (the centre tap's) with the second weight set, after conv1's
//     epilogue (same accumulators: no second accumulator set to hold);
//   * epilogues register-direct as in conv_regh.inl (scale / shift, bf16 pack, ReLU on the packed pairs, v_permlane32_swap, 16-byte
//     stores); the two outputs go to per-group slabs (group strides are arguments), so that each trunk's chain continues on a compact
//     one-group tensor.
// Both trunks' fronts run as ONE two-group launch before the chains fork (engine.TrunkPlan.after_stem).
// VMEM queue of a wave per tile, in issue order: [10 patch pieces of tile t + 1 (asm, K-steps 2-20)] [4 loads of the downsample's weights]
// [4 stores of t] [4 stores of idt]; the top of tile t + 1 waits vmcnt(8): everything but the stores.
template <int OPT = 0>
__global__ __launch_bounds__(256, 2) void conv3x3s2_c64_regh_kernel(ConvArgs p) {
#if defined(__HIP_DEVICE_COMPILE__)
    constexpr int NPIECE = 37;                     // 296 patch pixels (81 + 7 pad | 72 | 72 | 64), 8 per piece
    constexpr int PATCH_BYTES = NPIECE * 1024;
    constexpr int DUMP = 2 * PATCH_BYTES;          // 1 KB: where the three surplus piece slots (4 waves x 10 = 40) write their zeros
    constexpr int SS = DUMP + 1024;                // 2 KB: {scale[4], shift[4]} per channel quad, conv1's then the downsample's
    constexpr int NPW = 10;                        // pieces per wave and patch
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l31 = lane & 31, lhi = lane >> 5;
    const int g = blockIdx.y;
    span_stamp(p, false);

    // ---- tiles of this workgroup: a contiguous run, XCD-contiguous across the grid ----
    const int ntx = p.Wo >> 3, nty = p.Ho >> 3, tpi = ntx * nty;
    const int T = p.M * tpi;
    const int nwg = gridDim.x;
    const int b = blockIdx.x;
    const int logical = (nwg % 8 == 0) ? (b & 7) * (nwg >> 3) + (b >> 3) : b;
    // (round 6: multiply-high divisions -- (logical + 1) * T < 2^31 is checked by the launcher; two 64-bit run-time divisions here were
    //  ~0.3 us in front of the workgroup's first memory request)
    const int t_begin = __builtin_amdgcn_readfirstlane(w2c_fastdiv2(logical * T, nwg, p.mg_ntn));
    const int t_end = __builtin_amdgcn_readfirstlane(w2c_fastdiv2((logical + 1) * T, nwg, p.mg_ntn));
    if (t_begin >= t_end) { span_stamp(p, true); return; }      // workgroup-uniform

    // ---- weights -> registers, fragment order (w2c_pack_wfrag_bf16 / ops.pack_w1frag): block (g, w, tap, kc) = 1 KB, lane-linear.
    // wr[i] = the tap walked i-th (s2_tap order) ----
    u32x4_t wr[9][4];
    {
        const uint16_t* wf = p.w + ((size_t)(g * 4 + wave) * 36) * 512 + lane * 8;
        // (in two batches, the AGPR-resident taps pinned there before anything else is computed: with all 36 loads' destinations live in
        //  VGPRs at once the allocator spills the per-lane patch offsets, and reloads them behind a vmcnt(0) in front of every DMA)
#pragma unroll
        for (int i = 0; i < 6; ++i)
#pragma unroll
            for (int kc = 0; kc < 4; ++kc) wr[i][kc] = *reinterpret_cast<const u32x4_t*>(wf + (s2w_tapidx(i) * 4 + kc) * 512);
#pragma unroll
        for (int i = 0; i < 6; ++i)
#pragma unroll
            for (int kc = 0; kc < 4; ++kc) asm volatile("" : "+a"(wr[i][kc]));
#pragma unroll
        for (int i = 6; i < 9; ++i)
#pragma unroll
            for (int kc = 0; kc < 4; ++kc) wr[i][kc] = *reinterpret_cast<const u32x4_t*>(wf + (s2w_tapidx(i) * 4 + kc) * 512);
#pragma unroll
        for (int i = 6; i < 9; ++i)
#pragma unroll
            for (int kc = 0; kc < 4; ++kc) asm volatile("" : "+v"(wr[i][kc]));
    }
    // the downsample's 32 x 64 weights (16 registers) are NOT resident: 160 + 32 accumulators + the loop's own registers do not fit 256;
    // they are fetched (4 KB per wave, L2 hits) at the end of every tile's conv1 loop into the registers the B fragments have just left
    const __amdgpu_buffer_rsrc_t rs_w1 = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint16_t*>(p.w2), 0, (int)(gridDim.y * 16384), 0x00020000);
    int w1_soff = (g * 4 + wave) * 4096;
    float* const ss = reinterpret_cast<float*>(smem + SS);
    if (tid < 128) {
        ss[(tid >> 2) * 8 + (tid & 3)] = p.scale[g * 128 + tid];
        ss[(tid >> 2) * 8 + 4 + (tid & 3)] = p.shift[g * 128 + tid];
        ss[256 + (tid >> 2) * 8 + (tid & 3)] = p.scale2[g * 128 + tid];
        ss[256 + (tid >> 2) * 8 + 4 + (tid & 3)] = p.shift2[g * 128 + tid];
    }

    // ---- patch DMA.  Flat patch pixel q = 8 j + lane / 8 of piece j = wave + 4 i; q -> (phase, block row, block column):
    //   phase 0 = (py 1, px 1): q in [0, 88), pitch 9, 9 rows (81 real)      phase 1 = (1, 0): [88, 160), pitch 8, 9 rows
    //   phase 2 = (0, 1): [160, 232), pitch 9, 8 rows                         phase 3 = (0, 0): [232, 296), pitch 8, 8 rows
    // block (br, bc) of phase (py, px) = input pixel (2 oy0 - 1 + 2 br + (1 - py), 2 ox0 - 1 + 2 bc + (1 - px)); the 16-byte chunk c of
    // a pixel sits at LDS slot c ^ key, key = ((bc >> 1) & 3) | ((br & 1) << 2) (a quarter-wave's 16 fragment reads -- 2 block rows
    // x 8 columns -- then cover all 64 banks once) ----
    const unsigned lds_base = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(size_t)W2C_LPTR(smem));
    const unsigned shift_b = (unsigned)((p.W + 1) * p.xcs * 2);
    const unsigned long long xaddr = reinterpret_cast<unsigned long long>(p.x) + (unsigned long long)g * 128 - shift_b;
    const unsigned x_bytes = (unsigned)((size_t)p.M * p.H * p.W * p.xcs * 2 - (size_t)g * 128);
    const u32x4_t srd_x = {(unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)xaddr),
                           (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(xaddr >> 32)),
                           (unsigned)__builtin_amdgcn_readfirstlane((int)(x_bytes + shift_b)), 0x00020000u};
    int off_rel[NPW];
    unsigned hmask = 0;                            // bit i: piece i is a top-halo pixel | bit 10 + i: left halo | bit 20 + i: pad / surplus
#pragma unroll
    for (int i = 0; i < NPW; ++i) {
        const int q = 8 * (wave + 4 * i) + (lane >> 3);
        int ph, idx;
        if (q < 88) { ph = 0; idx = q; }
        else if (q < 160) { ph = 1; idx = q - 88; }
        else if (q < 232) { ph = 2; idx = q - 160; }
        else { ph = 3; idx = q - 232; }
        const int py = ph < 2 ? 1 : 0, px = (ph == 0 || ph == 2) ? 1 : 0;
        const int pitch = px ? 9 : 8, rows = py ? 9 : 8;
        const int br = idx / pitch, bc = idx - br * pitch;
        const bool real = (q < 296) & (br < rows);
        const int key = ((bc >> 1) & 3) | ((br & 1) << 2);
        const int chunk = (lane & 7) ^ key;
        const int ry = 2 * br + (1 - py), rx = 2 * bc + (1 - px);
        off_rel[i] = ((ry * p.W + rx) * p.xcs + chunk * 8) * 2;
        hmask |= ((py == 1 && br == 0) ? 1u : 0u) << i;
        hmask |= ((px == 1 && bc == 0) ? 1u : 0u) << (i + 10);
        hmask |= (real ? 0u : 1u) << (i + 20);
    }
    unsigned pbase = 0;                            // SOFFSET of the next patch: its first pixel (2 oy0 - 1, 2 ox0 - 1) from the shifted base
    unsigned pbad = 0;                             // per lane: bit i = piece i of the next patch is outside the image / pad (-> zeros)
    auto patch_piece = [&](int i, int buf) {
        const int j = wave + 4 * i;
        const unsigned dst = lds_base + (j < NPIECE ? (unsigned)buf * PATCH_BYTES + (unsigned)j * 1024u : (unsigned)DUMP);
        const unsigned vo = ((pbad >> i) & 1u) ? 0x80000000u : (unsigned)off_rel[i];
        asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, %3 offen lds"
                     :: "s"(dst), "v"(vo), "s"(srd_x), "s"(pbase) : "memory", "m0");
    };
    auto patch_setup = [&](int img, int oy0, int ox0, bool live) {
        pbase = (unsigned)((((img * p.H + 2 * oy0) * p.W) + 2 * ox0) * p.xcs * 2);
        const unsigned sel = 0x3FF00000u | (oy0 == 0 ? 0x3FFu : 0u) | (ox0 == 0 ? 0xFFC00u : 0u);
        const unsigned m = hmask & sel;
        pbad = !live ? 0x3FFu : ((m | (m >> 10) | (m >> 20)) & 0x3FFu);
    };

    // ---- outputs: per-group slabs; lane (l31, lhi) -> pixel (4 pt + l31 / 8, l31 % 8) of the tile, 8 channels 32 w + 16 jp + 8 lhi .. + 7 ----
    const int r = l31 >> 3, c = l31 & 7;
    const size_t t_bytes = (size_t)p.M * p.Ho * p.Wo * p.ycs * 2, i_bytes = (size_t)p.M * p.Ho * p.Wo * p.y2cs * 2;
    const __amdgpu_buffer_rsrc_t rs_t = __builtin_amdgcn_make_buffer_rsrc(reinterpret_cast<uint16_t*>(p.y) + (size_t)g * p.ygs, 0, (int)t_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_i = __builtin_amdgcn_make_buffer_rsrc(p.y2 + (size_t)g * p.y2gs, 0, (int)i_bytes, 0x00020000);
    const unsigned t_lane = (unsigned)(((r * p.Wo + c) * p.ycs + wave * 32 + lhi * 8) * 2);
    const unsigned i_lane = (unsigned)(((r * p.Wo + c) * p.y2cs + wave * 32 + lhi * 8) * 2);

    // ---- B-fragment geometry: byte offset inside a patch buffer of this lane's 16 bytes of K slice kc at tap (phase, dy, dx), pixel block pt:
    //   (PB[phase] + (4 pt + dy) pitch + dx) * 128   (immediate)   +   (r pitch + c) * 128   +   (((2 kc | lhi) ^ key) << 4) ----
    // ((2 kc | lhi) ^ key) << 4 = kb ^ (kc << 5) with kb = (key ^ lhi) << 4 < 128, and the pixel term is a multiple of 128: six lane
    // offsets (pitch 9: the four (dy, dx); pitch 8: dy = 0 | 1, dx = 0), one v_xor with a constant per K-step
    auto kb = [&](int dy, int dx) { return ((((((c + dx) >> 1) & 3) | (((r + dy) & 1) << 2)) ^ lhi) << 4); };
    const int lo9[2][2] = {{(r * 9 + c) * 128 + kb(0, 0), (r * 9 + c) * 128 + kb(0, 1)}, {(r * 9 + c) * 128 + kb(1, 0), (r * 9 + c) * 128 + kb(1, 1)}};
    const int lo8[2] = {(r * 8 + c) * 128 + kb(0, 0), (r * 8 + c) * 128 + kb(1, 0)};
    const char* const ssd1 = reinterpret_cast<const char*>(ss) + (wave * 8 + lhi) * 32;        // + j * 64: quad 8 w + 2 j + lhi
    const char* const ssd2 = ssd1 + 1024;

    int img = w2c_fastdiv2(t_begin, tpi, p.mg_txy), oy0, ox0;
    {
        const int rr = t_begin - img * tpi;
        const int ty = w2c_fastdiv2(rr, ntx, p.mg_tx);
        oy0 = ty * 8;
        ox0 = (rr - ty * ntx) * 8;
    }
    auto advance = [&](int& im, int& yy, int& xx) {
        xx += 8;
        if (xx == p.Wo) { xx = 0; yy += 8; if (yy == p.Ho) { yy = 0; ++im; } }
    };
    int in = img, yn = oy0, xn = ox0;
    patch_setup(in, yn, xn, true);
#pragma unroll
    for (int i = 0; i < NPW; ++i) patch_piece(i, 0);
    advance(in, yn, xn);
    int cur = 0;

    for (int t = t_begin; t < t_end; ++t) {
        const int nxt = cur ^ 1;
        if (t != t_begin) advance(img, oy0, ox0);
        patch_setup(in, yn, xn, t + 1 < t_end);
        advance(in, yn, xn);
        // this wave's pieces of patch(t) have landed (vmcnt retires in order: only the 8 stores of tile t - 1 may still be out), then the
        // workgroup barrier: the whole patch is there and every wave is done reading patch(t - 1), whose buffer patch(t + 1) overwrites
        if (t == t_begin) wait_vmcnt<0>(); else wait_vmcnt<8>();
        pipeline_barrier();

        f32x16_t acc[2];
        // lane offsets of this tile's buffer: the buffer's offset goes in BEFORE the per-K-step xor (PATCH_BYTES has no bit below 1 << 10,
        // so (lo + pcoff) ^ (kc << 5) = (lo ^ (kc << 5)) + pcoff) -- written the other way round the 24 loop-invariant values lo ^ (kc << 5)
        // are hoisted out of the tile loop and cost 24 registers
        const int pcoff = cur * PATCH_BYTES;
        const int a9[2][2] = {{lo9[0][0] + pcoff, lo9[0][1] + pcoff}, {lo9[1][0] + pcoff, lo9[1][1] + pcoff}};
        const int a8[2] = {lo8[0] + pcoff, lo8[1] + pcoff};
        auto frag = [&](int i, int kc, int pt) {
            const int ky = s2w_tapidx(i) / 3, kx = s2w_tapidx(i) % 3;
            const int dy = ky == 2 ? 1 : 0, dx = kx == 2 ? 1 : 0;
            const int ph = s2w_phase(i);
            const int pitch = (ph == 0 || ph == 2) ? 9 : 8;
            const int pb = ph == 0 ? 0 : ph == 1 ? 88 : ph == 2 ? 160 : 232;
            const int lo = (pitch == 9 ? a9[dy][dx] : a8[dy]) ^ (kc << 5);
            return *reinterpret_cast<const u32x4_t*>(smem + lo + (pb + (4 * pt + dy) * pitch + dx) * 128);
        };
        u32x4_t bx[2][2];
        const int tile_pix = (img * p.Ho + oy0) * p.Wo + ox0;
        bx[0][0] = frag(0, 0, 0);
        bx[0][1] = frag(0, 0, 1);
#pragma unroll
        for (int step = 0; step < 36; ++step) {
            const int i = step >> 2, kc = step & 3, cb = step & 1;
            if (step + 1 < 36) {
                bx[cb ^ 1][0] = frag((step + 1) >> 2, (step + 1) & 3, 0);
                bx[cb ^ 1][1] = frag((step + 1) >> 2, (step + 1) & 3, 1);
            }
            if (step >= 2 && step <= 20 && (step & 1) == 0) patch_piece((step - 2) >> 1, nxt);
            // register classes pinned as in conv_regh.inl: taps 0-5 + the accumulators in AGPRs, taps 6-8 and the downsample's weights in VGPRs
            if (step == 0) {
                asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, 0" : "=a"(acc[0]) : "a"(wr[i][kc]), "v"(bx[cb][0]));
                asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, 0" : "=a"(acc[1]) : "a"(wr[i][kc]), "v"(bx[cb][1]));
            } else if (i < 6) {
                asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+a"(acc[0]) : "a"(wr[i][kc]), "v"(bx[cb][0]));
                asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+a"(acc[1]) : "a"(wr[i][kc]), "v"(bx[cb][1]));
            } else {
                asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+a"(acc[0]) : "v"(wr[i][kc]), "v"(bx[cb][0]));
                asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+a"(acc[1]) : "v"(wr[i][kc]), "v"(bx[cb][1]));
            }
        }
        // the MFMAs are opaque to the compiler's hazard recogniser: cover the XDL-write -> VALU-read wait states by hand
        asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");
        u32x4_t w1r[4];
        asm volatile("" : "+s"(w1_soff));            // (opaque: keeps these loads inside the loop -- hoisted they are 16 more live registers)
#pragma unroll
        for (int kc = 0; kc < 4; ++kc) w1r[kc] = __builtin_bit_cast(u32x4_t, __builtin_amdgcn_raw_buffer_load_b128(rs_w1, lane * 16 + kc * 1024, w1_soff, 0));

        // ---- epilogue, register-direct (the arithmetic and its order are conv_regh.inl's = the ring kernels') ----
        auto epilogue = [&](const char* ssd, bool relu, __amdgpu_buffer_rsrc_t rs, unsigned lane_off, int cs) {
            const s16x2_t relu_lo = relu ? s16x2_t{0, 0} : s16x2_t{(short)-32768, (short)-32768};
#pragma unroll
            for (int jp = 0; jp < 2; ++jp) {
                asm volatile("" : "+a"(acc[0]), "+a"(acc[1])::"memory");
                f32x4_t scv[2], shv[2];
#pragma unroll
                for (int jj = 0; jj < 2; ++jj) {
                    scv[jj] = *reinterpret_cast<const f32x4_t*>(ssd + (2 * jp + jj) * 64);
                    shv[jj] = *reinterpret_cast<const f32x4_t*>(ssd + (2 * jp + jj) * 64 + 16);
                }
#pragma unroll
                for (int pt = 0; pt < 2; ++pt) {
                    const int so = (tile_pix + 4 * pt * p.Wo) * cs * 2;
                    uint32_t pk[2][2];
#pragma unroll
                    for (int jj = 0; jj < 2; ++jj) {
                        const int j = 2 * jp + jj;
                        const f32x4_t a = f32x4_t{acc[pt][j * 4], acc[pt][j * 4 + 1], acc[pt][j * 4 + 2], acc[pt][j * 4 + 3]} * scv[jj] + shv[jj];
#pragma unroll
                        for (int e = 0; e < 2; ++e) {
                            const s16x2_t h = __builtin_bit_cast(s16x2_t, pack_bf16x2(a[2 * e], a[2 * e + 1]));
                            pk[jj][e] = __builtin_bit_cast(uint32_t, __builtin_elementwise_max(h, relu_lo));
                        }
                    }
                    const auto sa = __builtin_amdgcn_permlane32_swap(pk[0][0], pk[1][0], false, false);
                    const auto sb = __builtin_amdgcn_permlane32_swap(pk[0][1], pk[1][1], false, false);
                    const u32x4_t o = {sa[0], sb[0], sa[1], sb[1]};
                    __builtin_amdgcn_raw_buffer_store_b128(o, rs, lane_off + jp * 32, so, 0);
                }
            }
        };
        epilogue(ssd1, p.relu != 0, rs_t, t_lane, p.ycs);
        asm volatile("" ::: "memory");

        // ---- downsample: the centre tap's fragments (phase (0, 0)) x the 1x1 weights, 4 K-steps into the same accumulators ----
        bx[0][0] = frag(8, 0, 0);
        bx[0][1] = frag(8, 0, 1);
#pragma unroll
        for (int kc = 0; kc < 4; ++kc) {
            const int cb = kc & 1;
            if (kc + 1 < 4) {
                bx[cb ^ 1][0] = frag(8, kc + 1, 0);
                bx[cb ^ 1][1] = frag(8, kc + 1, 1);
            }
            if (kc == 0) {
                asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, 0" : "=a"(acc[0]) : "v"(w1r[kc]), "v"(bx[cb][0]));
                asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, 0" : "=a"(acc[1]) : "v"(w1r[kc]), "v"(bx[cb][1]));
            } else {
                asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+a"(acc[0]) : "v"(w1r[kc]), "v"(bx[cb][0]));
                asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+a"(acc[1]) : "v"(w1r[kc]), "v"(bx[cb][1]));
            }
        }
        asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");
        epilogue(ssd2, false, rs_i, i_lane, p.y2cs);
        asm volatile("" ::: "memory");
        cur = nxt;
    }
    span_stamp(p, true);
#endif
}

int launch_s2regh(ConvArgs& a, int groups, hipStream_t s) {
    constexpr int lds = 2 * 37 * 1024 + 1024 + 2048;
    static_assert(lds <= 80 * 1024, "two workgroups per CU");
    if (a.ks != 3 || a.stride != 2 || a.Cin != 64 || a.Cout != 128 || (a.H & 1) || (a.W & 1) || a.Ho % 8 != 0 || a.Wo % 8 != 0 || a.res || a.y_f32 ||
        a.y8 || !a.y || !a.y2 || !a.w2 || a.ws)
        return W2C_E_ARG;
    if ((size_t)a.M * a.H * a.W * a.xcs * 2 >= (1ull << 31) || (size_t)a.M * a.Ho * a.Wo * a.ycs * 2 >= (1ull << 31) ||
        (size_t)a.M * a.Ho * a.Wo * a.y2cs * 2 >= (1ull << 31))
        return W2C_E_ARG;
    static std::atomic<unsigned long long> attr_mask{0};
    static int n_cu[64] = {0};
    int dev = 0;
    (void)hipGetDevice(&dev);
    if (!((attr_mask.load(std::memory_order_acquire) >> (dev & 63)) & 1ull)) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&conv3x3s2_c64_regh_kernel<0>), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
        hipDeviceProp_t prop;
        n_cu[dev & 63] = (hipGetDeviceProperties(&prop, dev) == hipSuccess && prop.multiProcessorCount > 0) ? prop.multiProcessorCount : 256;
        attr_mask.fetch_or(1ull << (dev & 63), std::memory_order_release);
    }
    const long tiles = (long)a.M * (a.Ho / 8) * (a.Wo / 8);
    // one group: two 4-wave workgroups per CU; several groups: ONE workgroup per CU in total -- the launch is HBM-bound then (4.2 TB/s
    // at cfg 2) and half the workgroups are 3-5 % faster (tools/bench_s2_front_c64.py: 37.9 vs 39.8 us, M = 8: 20.3 vs 21.1, 1024^2: 31.1 vs 32.7)
    long wgs = groups == 1 ? 2L * n_cu[dev & 63] : (n_cu[dev & 63] + groups - 1) / groups;
    const int opt = w2c_option(W2C_OPT_REGH_WGS);
    if (opt > 0) wgs = opt;
    if (wgs > tiles) wgs = tiles;
    if ((wgs + 1) * tiles >= (1ll << 31)) return W2C_E_ARG;                  // (the kernel's 32-bit tile-run arithmetic)
    a.mg_ntn = w2c_magic_floor((unsigned)wgs);
    a.mg_txy = w2c_magic_floor((unsigned)((a.Ho / 8) * (a.Wo / 8)));
    a.mg_tx = w2c_magic_floor((unsigned)(a.Wo / 8));
    hipLaunchKernelGGL((conv3x3s2_c64_regh_kernel<0>), dim3((unsigned)wgs, groups), dim3(256), lds, s, a);
    return w2c_launch_status();
}
