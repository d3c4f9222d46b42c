// Do a wave's MFMAs and ANOTHER wave's VALU instructions on the same SIMD overlap?  (gfx950; the stem's ping-pong form assumes they do.)
// Workgroup = 8 waves = 2 per SIMD: waves 0-3 run a v_mfma_f32_32x32x16_bf16 stream, waves 4-7 a VALU stream of R instructions per MFMA
// of the other wave.  Timed: MFMA waves alone, VALU waves alone, both.  256 workgroups, one per CU.
// hipcc --offload-arch=gfx950 -O3 tools/ubench/coissue.hip -o /tmp/coissue && /tmp/coissue
#include <hip/hip_runtime.h>
#include <cstdio>
typedef __attribute__((ext_vector_type(16))) float f32x16_t;
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4_t;

// what: bit 0 = MFMA waves work, bit 1 = VALU waves work.  KIND: 0 = v_fma_f32, 1 = v_pk_max / cvt_pk mix, 2 = DPP row shifts
template <int R, int KIND, bool ACCV>
__global__ __launch_bounds__(512) void k(float* out, unsigned long long* cyc, int iters, int what) {
    const int wave = threadIdx.x >> 6;
    float s = 0;
    const long long t0 = clock64();
    if (wave < 4) {
        if (what & 1) {
            u32x4_t a0 = {0x3f803f80u + threadIdx.x, 0x3f803f80u, 0x3f803f80u, 0x3f803f80u}, b0 = a0;
            f32x16_t c0, c1;
            for (int i = 0; i < 16; ++i) { c0[i] = 0; c1[i] = 0; }
            if (ACCV) asm volatile("" : "+v"(c0), "+v"(c1));
            else asm volatile("" : "+a"(c0), "+a"(c1));
            for (int it = 0; it < iters; ++it) {
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    if (ACCV) {                      // accumulators in VGPRs: the form hipcc emits for the builtin in a <= 256-register kernel
                        asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(c0) : "v"(a0), "v"(b0));
                        asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(c1) : "v"(a0), "v"(b0));
                    } else {
                        asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+a"(c0) : "v"(a0), "v"(b0));
                        asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+a"(c1) : "v"(a0), "v"(b0));
                    }
                }
            }
            for (int i = 0; i < 16; ++i) s += c0[i] + c1[i];
        }
    } else if (what & 2) {
        float x0 = threadIdx.x, x1 = x0 + 1, x2 = x0 + 2, x3 = x0 + 3, x4 = x0 * 0.5f, x5 = x1 * 0.5f, x6 = x2 * 0.5f, x7 = x3 * 0.5f;
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int u = 0; u < 16 * R / 8; ++u) {          // 16 MFMAs per iteration on the other wave -> 16 R VALU here, 8 per round
                if (KIND == 0) {
                    asm volatile("v_fma_f32 %0, %0, %1, %1\n\tv_fma_f32 %2, %2, %3, %3\n\tv_fma_f32 %4, %4, %5, %5\n\tv_fma_f32 %6, %6, %7, %7\n\t"
                                 "v_fma_f32 %1, %1, %0, %0\n\tv_fma_f32 %3, %3, %2, %2\n\tv_fma_f32 %5, %5, %4, %4\n\tv_fma_f32 %7, %7, %6, %6"
                                 : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7));
                } else if (KIND == 1) {
                    asm volatile("v_cvt_pk_bf16_f32 %0, %0, %1\n\tv_pk_max_i16 %2, %2, %3\n\tv_max3_f32 %4, %4, %5, %6\n\tv_cvt_pk_bf16_f32 %7, %7, %6\n\t"
                                 "v_pk_max_i16 %1, %1, %0\n\tv_max3_f32 %3, %3, %2, %4\n\tv_fma_f32 %5, %5, %4, %4\n\tv_max_f32 %6, %6, %7"
                                 : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7));
                } else if (KIND == 5) {      // whole-wave shifts (wave_shr:1 / wave_shl:1), as the stem's horizontal 3-max uses
                    asm volatile("s_nop 1\n\tv_mov_b32_dpp %0, %1 wave_shr:1 row_mask:0xf bank_mask:0xf\n\tv_max_f32 %2, %2, %0\n\t"
                                 "v_mov_b32_dpp %3, %4 wave_shl:1 row_mask:0xf bank_mask:0xf\n\tv_max_f32 %5, %5, %3\n\t"
                                 "v_mov_b32_dpp %6, %7 wave_shr:1 row_mask:0xf bank_mask:0xf\n\tv_max_f32 %1, %1, %6\n\t"
                                 "v_mov_b32_dpp %4, %2 wave_shl:1 row_mask:0xf bank_mask:0xf\n\tv_max_f32 %7, %7, %4"
                                 : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7));
                } else if (KIND == 3) {      // reads of this wave's own AGPRs (an accumulator read-out), 4 of 8 instructions
                    asm volatile("v_accvgpr_read_b32 %0, a0\n\tv_max_f32 %1, %1, %0\n\tv_accvgpr_read_b32 %2, a1\n\tv_max_f32 %3, %3, %2\n\t"
                                 "v_accvgpr_read_b32 %4, a2\n\tv_max_f32 %5, %5, %4\n\tv_accvgpr_read_b32 %6, a3\n\tv_max_f32 %7, %7, %6"
                                 : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7) :: "a0", "a1", "a2", "a3");
                } else if (KIND == 4) {      // all 8 are AGPR reads
                    asm volatile("v_accvgpr_read_b32 %0, a0\n\tv_accvgpr_read_b32 %1, a1\n\tv_accvgpr_read_b32 %2, a2\n\tv_accvgpr_read_b32 %3, a3\n\t"
                                 "v_accvgpr_read_b32 %4, a4\n\tv_accvgpr_read_b32 %5, a5\n\tv_accvgpr_read_b32 %6, a6\n\tv_accvgpr_read_b32 %7, a7"
                                 : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7) :: "a0", "a1", "a2", "a3", "a4", "a5", "a6", "a7");
                } else {
                    asm volatile("v_mov_b32_dpp %0, %1 row_shr:1 row_mask:0xf bank_mask:0xf\n\tv_max_f32 %2, %2, %0\n\t"
                                 "v_mov_b32_dpp %3, %4 row_shl:1 row_mask:0xf bank_mask:0xf\n\tv_max_f32 %5, %5, %3\n\t"
                                 "v_mov_b32_dpp %6, %7 row_shr:1 row_mask:0xf bank_mask:0xf\n\tv_max_f32 %1, %1, %6\n\t"
                                 "v_mov_b32_dpp %4, %2 row_shl:1 row_mask:0xf bank_mask:0xf\n\tv_max_f32 %7, %7, %4"
                                 : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7));
                }
            }
        }
        s = x0 + x1 + x2 + x3 + x4 + x5 + x6 + x7;
    }
    asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");
    const long long t1 = clock64();
    out[blockIdx.x * 512 + threadIdx.x] = s;
    if ((threadIdx.x & 63) == 0) cyc[blockIdx.x * 8 + wave] = (unsigned long long)(t1 - t0);
}

template <int R, int KIND, bool ACCV = false>
void run(const char* name) {
    float* out; unsigned long long* cyc;
    hipMalloc(&out, 256 * 512 * 4); hipMalloc(&cyc, 256 * 8 * 8);
    const int iters = 2000;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    float ms[4] = {0, 0, 0, 0};
    unsigned long long h[8];
    double cy[4][2] = {};
    for (int what = 1; what <= 3; ++what) {
        k<R, KIND, ACCV><<<256, 512>>>(out, cyc, 10, what);
        hipDeviceSynchronize();
        hipEventRecord(e0);
        k<R, KIND, ACCV><<<256, 512>>>(out, cyc, iters, what);
        hipEventRecord(e1); hipEventSynchronize(e1);
        hipEventElapsedTime(&ms[what], e0, e1);
        hipMemcpy(h, cyc, sizeof(h), hipMemcpyDeviceToHost);
        cy[what][0] = (double)h[0] / (iters * 16.0);        // cycles per MFMA (MFMA wave 0)
        cy[what][1] = (double)h[4] / (iters * 16.0 * R);    // cycles per VALU instruction (VALU wave 4)
    }
    printf("%-34s R=%d: MFMA alone %.3f ms (%.1f cyc/MFMA) | VALU alone %.3f ms (%.2f cyc/inst) | both %.3f ms (%.1f cyc/MFMA, %.2f cyc/inst)  -> overlap %.0f %%\n",
           name, R, ms[1], cy[1][0], ms[2], cy[2][1], ms[3], cy[3][0], cy[3][1], 100.0 * (ms[1] + ms[2] - ms[3]) / (ms[1] < ms[2] ? ms[1] : ms[2]));
    hipFree(out); hipFree(cyc);
}

int main() {
    run<2, 0>("v_fma_f32");
    run<4, 0>("v_fma_f32");
    run<6, 0>("v_fma_f32");
    run<8, 0>("v_fma_f32");
    run<4, 1>("cvt_pk / pk_max / max3 mix");
    run<6, 1>("cvt_pk / pk_max / max3 mix");
    run<4, 0, true>("v_fma_f32 | MFMA acc in VGPRs");
    run<6, 0, true>("v_fma_f32 | MFMA acc in VGPRs");
    run<4, 1, true>("cvt/pk_max/max3 | MFMA acc in VGPRs");
    run<4, 5>("DPP wave_shr / wave_shl + max");
    run<4, 3>("accvgpr_read + max (own AGPRs)");
    run<4, 4>("accvgpr_read only");
    run<2, 4>("accvgpr_read only");
    run<4, 2>("DPP row shift + max");
    run<6, 2>("DPP row shift + max");
    return 0;
}
