// Calibration microbenchmark (gfx950): cycles per v_mfma_f32_32x32x16_bf16 on one wave per SIMD for
//   - N independent accumulator chains taken round-robin (N = 1, 2, 4)
//   - runs of R same-accumulator MFMAs back to back, chains alternating (R = 4)
//   - the A operand in AGPRs or VGPRs
// hipcc --offload-arch=gfx950 -O3 tools/ubench/mfma_rate.hip -o /tmp/mfma_rate && /tmp/mfma_rate
#include <hip/hip_runtime.h>
#include <cstdio>
typedef __attribute__((ext_vector_type(16))) float f32x16_t;
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4_t;

// RND: operands are pseudo-random bf16 values in +-[0.5, 1) (different in every lane and register) instead of the constant 1.0 --
// the MFMA issue rate in CYCLES is the same, the clock the chip holds under the load is not (power): this is the ceiling a kernel
// working on real data can reach.
__device__ inline unsigned rnd_bf16x2(unsigned x) {
    x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16;
    return (x & 0x807F807Fu) | 0x3F003F00u;
}
template <int MODE, bool RND = false>
__global__ __launch_bounds__(256) void k(float* out, unsigned long long* cyc, int iters) {
    u32x4_t a0 = {0x3f803f80u + threadIdx.x, 0x3f803f80u, 0x3f803f80u, 0x3f803f80u}, a1 = a0, b0 = a0, b1 = a0;
    if (RND) {
        const unsigned h = (blockIdx.x * 256 + threadIdx.x) * 16;
        for (int i = 0; i < 4; ++i) { a0[i] = rnd_bf16x2(h + i); a1[i] = rnd_bf16x2(h + 4 + i); b0[i] = rnd_bf16x2(h + 8 + i); b1[i] = rnd_bf16x2(h + 12 + i); }
    }
    f32x16_t c0, c1, c2, c3;
    for (int i = 0; i < 16; ++i) { c0[i] = 0; c1[i] = 0; c2[i] = 0; c3[i] = 0; }
    asm volatile("" : "+a"(c0), "+a"(c1), "+a"(c2), "+a"(c3));
    const long long t0 = clock64();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            if (MODE == 0) {          // 1 chain, A in AGPR
                asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+a"(c0) : "a"(a0), "v"(b0));
                asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+a"(c0) : "a"(a1), "v"(b1));
                asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+a"(c0) : "a"(a0), "v"(b1));
                asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+a"(c0) : "a"(a1), "v"(b0));
            } else if (MODE == 1) {   // 2 chains alternating, A in AGPR / VGPR like the fused block
                asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+a"(c0) : "a"(a0), "v"(b0));
                asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+a"(c1) : "v"(a1), "v"(b0));
                asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+a"(c0) : "a"(a0), "v"(b1));
                asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+a"(c1) : "v"(a1), "v"(b1));
            } else if (MODE == 2) {   // 4 chains round-robin
                asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+a"(c0) : "a"(a0), "v"(b0));
                asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+a"(c1) : "v"(a1), "v"(b0));
                asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+a"(c2) : "a"(a0), "v"(b1));
                asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+a"(c3) : "v"(a1), "v"(b1));
            } else if (MODE == 3) {   // 4 chains, all operands VGPR, accumulators VGPR
                asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(c0) : "v"(a0), "v"(b0));
                asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(c1) : "v"(a1), "v"(b0));
                asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(c2) : "v"(a0), "v"(b1));
                asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(c3) : "v"(a1), "v"(b1));
            } else if (MODE == 4) {   // 2 chains, runs of 4
                asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+a"(c0) : "a"(a0), "v"(b0));
                asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+a"(c0) : "a"(a0), "v"(b1));
                asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+a"(c0) : "a"(a0), "v"(b0));
                asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+a"(c0) : "a"(a0), "v"(b1));
                asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+a"(c1) : "v"(a1), "v"(b0));
                asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+a"(c1) : "v"(a1), "v"(b1));
                asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+a"(c1) : "v"(a1), "v"(b0));
                asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+a"(c1) : "v"(a1), "v"(b1));
            } else if (MODE == 5) {   // 2 chains alternating, all VGPR operands + VGPR accumulators
                asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(c0) : "v"(a0), "v"(b0));
                asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(c1) : "v"(a1), "v"(b0));
                asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(c0) : "v"(a0), "v"(b1));
                asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(c1) : "v"(a1), "v"(b1));
            } else if (MODE == 6) {   // 16x16x32: 4 chains
                typedef __attribute__((ext_vector_type(4))) float f32x4_t;
                f32x4_t d0 = {c0[0], c0[1], c0[2], c0[3]}, d1 = d0, d2 = d0, d3 = d0;
                asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+v"(d0) : "v"(a0), "v"(b0));
                asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+v"(d1) : "v"(a1), "v"(b0));
                asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+v"(d2) : "v"(a0), "v"(b1));
                asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+v"(d3) : "v"(a1), "v"(b1));
                c0[0] = d0[0] + d1[0] + d2[0] + d3[0];
            }
        }
    }
    asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");
    const long long t1 = clock64();
    float s = 0;
    for (int i = 0; i < 16; ++i) s += c0[i] + c1[i] + c2[i] + c3[i];
    out[blockIdx.x * 256 + threadIdx.x] = s;
    if (threadIdx.x == 0) cyc[blockIdx.x] = (unsigned long long)(t1 - t0);
}

template <int MODE, bool RND = false>
void run(const char* name, int nmfma_per_u) {
    float* out; unsigned long long* cyc;
    hipMalloc(&out, 256 * 256 * 4); hipMalloc(&cyc, 256 * 8);
    const int iters = 2000;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    k<MODE, RND><<<256, 256>>>(out, cyc, 10);
    hipEventRecord(e0);
    k<MODE, RND><<<256, 256>>>(out, cyc, iters);
    hipEventRecord(e1); hipDeviceSynchronize();
    float ms; hipEventElapsedTime(&ms, e0, e1);
    unsigned long long h[256]; hipMemcpy(h, cyc, 256 * 8, hipMemcpyDeviceToHost);
    double n = (double)iters * 8 * nmfma_per_u;
    double tf = n * 256 * 4 * 32768.0 / (ms * 1e-3) / 1e12;
    printf("%-52s %6.1f cycles/MFMA (clock64)  %7.1f ns/MFMA  %7.0f TFLOP/s chip  => %.2f GHz\n", name, h[0] / n, ms * 1e6 / n, tf, (h[0] / n) / (ms * 1e6 / n));
}


// LDS-fed variants: B fragments come from ds_read_b128 issued FD steps ahead (the fused BasicBlock's inner loop)
template <int MODE>
__global__ __launch_bounds__(256) void kl(float* out, unsigned long long* cyc, int iters) {
    __shared__ __attribute__((aligned(16))) char lds[65536];
    for (int i = threadIdx.x; i < 65536 / 16; i += 256) reinterpret_cast<u32x4_t*>(lds)[i] = u32x4_t{0x3f803f80u, 0x3f803f80u, 0x3f803f80u, (unsigned)i};
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    u32x4_t a0 = {0x3f803f80u + threadIdx.x, 0x3f803f80u, 0x3f803f80u, 0x3f803f80u}, a1 = a0;
    f32x16_t c0, c1, c2, c3;
    for (int i = 0; i < 16; ++i) { c0[i] = 0; c1[i] = 0; c2[i] = 0; c3[i] = 0; }
    asm volatile("" : "+a"(c0), "+a"(c1), "+a"(c2), "+a"(c3));
    const char* base = lds + wave * 16384 + (lane & 31) * 128 + ((((lane & 31) >> 1) ^ (lane >> 5)) & 7) * 16;
    const long long t0 = clock64();
    for (int it = 0; it < iters; ++it) {
        u32x4_t bx[4];
        int off = (it & 3) * 4096;
        asm volatile("" : "+v"(off));
#pragma unroll
        for (int f = 0; f < 3; ++f) bx[f] = *reinterpret_cast<const u32x4_t*>(base + off + ((f << 5)));
#pragma unroll
        for (int st = 0; st < 32; ++st) {
            if (st + 3 < 32) bx[(st + 3) & 3] = *reinterpret_cast<const u32x4_t*>(base + off + ((((st + 3) & 3) << 5) ^ ((st + 3) >> 2 << 7)));
            if (MODE == 0) {          // 2 chains, 1 read per 2 MFMAs
                asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+a"(c0) : "a"(a0), "v"(bx[st & 3]));
                asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+a"(c1) : "v"(a1), "v"(bx[st & 3]));
            } else if (MODE == 1) {   // + s_nop 1 in front of the agpr-A MFMA
                asm volatile("s_nop 1\n\tv_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+a"(c0) : "a"(a0), "v"(bx[st & 3]));
                asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+a"(c1) : "v"(a1), "v"(bx[st & 3]));
            } else if (MODE == 2) {   // 4 MFMAs per read (two reads feed four chains every other step): the regw ratio 0.5 reads / MFMA
                asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+a"(c0) : "a"(a0), "v"(bx[st & 3]));
                asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+a"(c1) : "v"(a1), "v"(bx[st & 3]));
                asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+a"(c2) : "a"(a0), "v"(bx[st & 3]));
                asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+a"(c3) : "v"(a1), "v"(bx[st & 3]));
            }
        }
    }
    asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");
    const long long t1 = clock64();
    float s = 0;
    for (int i = 0; i < 16; ++i) s += c0[i] + c1[i] + c2[i] + c3[i];
    out[blockIdx.x * 256 + threadIdx.x] = s;
    if (threadIdx.x == 0) cyc[blockIdx.x] = (unsigned long long)(t1 - t0);
}
template <int MODE>
void runl(const char* name, int nmfma_per_it) {
    float* out; unsigned long long* cyc;
    hipMalloc(&out, 256 * 256 * 4); hipMalloc(&cyc, 256 * 8);
    const int iters = 1000;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    kl<MODE><<<256, 256>>>(out, cyc, 10);
    hipEventRecord(e0);
    kl<MODE><<<256, 256>>>(out, cyc, iters);
    hipEventRecord(e1); hipDeviceSynchronize();
    float ms; hipEventElapsedTime(&ms, e0, e1);
    unsigned long long h[256]; hipMemcpy(h, cyc, 256 * 8, hipMemcpyDeviceToHost);
    double n = (double)iters * nmfma_per_it;
    printf("%-52s %6.1f cycles/MFMA (clock64)  %7.1f ns/MFMA\n", name, h[0] / n, ms * 1e6 / n);
}

int main() {
    run<0>("1 chain (A agpr)", 4);
    run<1>("2 chains alternating (A agpr / vgpr)", 4);
    run<2>("4 chains round-robin (A agpr / vgpr)", 4);
    run<3>("4 chains, everything in VGPRs", 4);
    run<4>("2 chains, runs of 4 back to back", 8);
    run<5>("2 chains alternating, everything in VGPRs", 4);
    run<6>("16x16x32, 4 chains (VGPR)", 4);
    run<2, true>("4 chains round-robin, RANDOM operands", 4);
    run<0, true>("1 chain, RANDOM operands", 4);
    run<2>("4 chains round-robin, constant operands (again)", 4);
    runl<0>("LDS-fed, 2 chains, 1 ds_read_b128 per 2 MFMAs", 64);
    runl<1>("LDS-fed, same + s_nop 1", 64);
    runl<2>("LDS-fed, 4 chains, 1 ds_read_b128 per 4 MFMAs", 128);
    return 0;
}
