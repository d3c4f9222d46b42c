// Probe: does v_pk_fma_f32 return wrong LOW halves when its waves share a CU with another kernel's MFMA
// waves?  Round 3 saw a few wrong keys / queries from the wide-K head kernel -- always the low half of an SLP-packed v_pk_fma_f32 pair,
// only while that kernel ran beside the conv kernels -- rebuilt the library without SLP vectorisation and pinned the behaviour with
// tests; the CAUSE was never isolated.  This program isolates the instruction: a checker kernel runs dependent chains of v_pk_fma_f32 and,
// lane for lane, the same chain as two scalar v_fma_f32 (identical IEEE results by definition), with its operands (a) in registers and
// (b) streamed from global memory like the head kernel's weights, and counts bitwise mismatches -- alone, and beside a kernel that keeps
// every SIMD's matrix pipe busy (v_mfma_f32_32x32x16_bf16, accumulators in AGPRs or in VGPRs) on a second stream, the two kernels sized
// so that their waves share CUs.
//   hipcc --offload-arch=gfx950 -O3 tools/ubench/pkfma_mfma.hip -o /tmp/pkfma && /tmp/pkfma
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef __attribute__((ext_vector_type(16))) float f32x16_t;
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4_t;
typedef __attribute__((ext_vector_type(2))) float f32x2_t;

template <bool AGPR>
__global__ __launch_bounds__(256) void mfma_hog(float* sink, int iters) {
    u32x4_t a = {0x3f803f80u + threadIdx.x, 0x3f813f80u, 0x3f803f82u, 0x3f803f80u}, b = {0x3f003f00u, 0x3f013f00u, 0x3f003f02u, 0x3f003f00u};
    f32x16_t c0, c1;
    for (int i = 0; i < 16; ++i) { c0[i] = 0; c1[i] = 0; }
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 16; ++u) {
            if (AGPR) {
                asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+a"(c0) : "v"(a), "v"(b));
                asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+a"(c1) : "v"(a), "v"(b));
            } else {
                asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(c0) : "v"(a), "v"(b));
                asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(c1) : "v"(a), "v"(b));
            }
        }
    }
    asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");
    if (c0[0] + c1[3] == 12345.678f) sink[0] = c0[1];
}

// MEM: the multiplier pairs come from global memory (16 B per lane and step, two slices in flight), as in linear_widek_kernel
// SEL (round 5): the operand-select forms the SLP-packed build of linear_widek_kernel actually contains (tools/r05: its ISA has 120
// v_pk_fma_f32, every one with a broadcast of ONE multiplier dword to both halves): 0 = plain, 1 = op_sel_hi:[1,0,1] (both halves read
// src1's LOW dword), 2 = op_sel:[0,1,0] (both halves read src1's HIGH dword -- the low half crosses over).
template <bool MEM, int SEL = 0>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2, 2))) void pk_checker(const float* __restrict__ w, int steps,
                                                                                             unsigned long long* bad, float* sink) {
    const int tid = blockIdx.x * 256 + threadIdx.x;
    f32x2_t accp[4];
    float accs[4][2];
    for (int r = 0; r < 4; ++r) { accp[r] = f32x2_t{0.f, 0.f}; accs[r][0] = accs[r][1] = 0.f; }
    float x0 = 1.0f + (tid & 1023) * 9.5367431640625e-07f, x1 = 0.5f + (tid & 511) * 1.9073486328125e-06f;
    const float* wp = w + (size_t)(tid & 4095) * 4;
    for (int s = 0; s < steps; ++s) {
        float w4[4];
        if (MEM) {
            const float4 v = *reinterpret_cast<const float4*>(wp + (size_t)(s & 255) * 16384);
            w4[0] = v.x; w4[1] = v.y; w4[2] = v.z; w4[3] = v.w;
        } else {
            w4[0] = 0.999f + 1e-4f * (s & 7); w4[1] = -1.001f + 2e-4f * (s & 3); w4[2] = 0.5f; w4[3] = -0.25f;
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const f32x2_t xv = {x0 + 0.125f * r, x1 - 0.0625f * r};
            const f32x2_t wv = {w4[r], w4[(r + 1) & 3]};
            if (SEL == 0) asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(accp[r]) : "v"(xv), "v"(wv));
            if (SEL == 1) asm volatile("v_pk_fma_f32 %0, %1, %2, %0 op_sel_hi:[1,0,1]" : "+v"(accp[r]) : "v"(xv), "v"(wv));
            if (SEL == 2) asm volatile("v_pk_fma_f32 %0, %1, %2, %0 op_sel:[0,1,0]" : "+v"(accp[r]) : "v"(xv), "v"(wv));
            const float m0 = SEL == 2 ? wv[1] : wv[0], m1 = SEL == 1 ? wv[0] : wv[1];
            asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(accs[r][0]) : "v"(xv[0]), "v"(m0));
            asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(accs[r][1]) : "v"(xv[1]), "v"(m1));
        }
        x0 = x0 * 0.99951171875f + 4.8828125e-4f;
        x1 = x1 * 1.00048828125f - 2.44140625e-4f;
    }
    unsigned lo = 0, hi = 0;
    for (int r = 0; r < 4; ++r) {
        lo += __float_as_uint(accp[r][0]) != __float_as_uint(accs[r][0]);
        hi += __float_as_uint(accp[r][1]) != __float_as_uint(accs[r][1]);
    }
    if (lo) atomicAdd(bad, (unsigned long long)lo);
    if (hi) atomicAdd(bad + 1, (unsigned long long)hi);
    if (accp[0][0] == 12345.678f) sink[0] = accp[1][1];
}

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

template <bool MEM, int SEL = 0>
static void run(const char* name, int hog, hipStream_t s_chk, hipStream_t s_hog, const float* w, unsigned long long* bad, float* sink) {
    CK(hipMemset(bad, 0, 16));
    const int rounds = 40;
    for (int r = 0; r < rounds; ++r) {
        if (hog == 1) hipLaunchKernelGGL(mfma_hog<true>, dim3(512), dim3(256), 0, s_hog, sink, 400);      // 2 workgroups per CU: 8 waves, 2 per SIMD
        if (hog == 2) hipLaunchKernelGGL(mfma_hog<false>, dim3(512), dim3(256), 0, s_hog, sink, 400);
        for (int k = 0; k < 4; ++k) hipLaunchKernelGGL((pk_checker<MEM, SEL>), dim3(512), dim3(256), 0, s_chk, w, 3000, bad, sink);
    }
    CK(hipDeviceSynchronize());
    unsigned long long h[2];
    CK(hipMemcpy(h, bad, 16, hipMemcpyDeviceToHost));
    printf("%-58s low-half mismatches %llu, high-half %llu  (of %llu pairs)\n", name, h[0], h[1],
           (unsigned long long)rounds * 4 * 512 * 256 * 4);
}

int main() {
    hipStream_t s0, s1;
    CK(hipStreamCreate(&s0));
    CK(hipStreamCreate(&s1));
    float *w, *sink;
    unsigned long long* bad;
    CK(hipMalloc(&w, (size_t)256 * 16384 * 4 + 65536));
    CK(hipMalloc(&sink, 64));
    CK(hipMalloc(&bad, 16));
    float* hw = (float*)malloc((size_t)256 * 16384 * 4 + 65536);
    for (size_t i = 0; i < (size_t)256 * 16384 + 16384; ++i) hw[i] = 0.5f + (float)((i * 2654435761u) & 0xFFFF) / 65536.0f;
    CK(hipMemcpy(w, hw, (size_t)256 * 16384 * 4 + 65536, hipMemcpyHostToDevice));
    run<false>("register operands, alone", 0, s0, s1, w, bad, sink);
    run<false>("register operands, beside MFMA waves (AGPR accumulators)", 1, s0, s1, w, bad, sink);
    run<false>("register operands, beside MFMA waves (VGPR accumulators)", 2, s0, s1, w, bad, sink);
    run<true>("operands from global memory, alone", 0, s0, s1, w, bad, sink);
    run<true>("operands from global memory, beside MFMA (AGPR acc.)", 1, s0, s1, w, bad, sink);
    run<true>("operands from global memory, beside MFMA (VGPR acc.)", 2, s0, s1, w, bad, sink);
    run<true, 1>("global operands, op_sel_hi:[1,0,1] (low dword to both), alone", 0, s0, s1, w, bad, sink);
    run<true, 1>("global operands, op_sel_hi:[1,0,1], beside MFMA (AGPR acc.)", 1, s0, s1, w, bad, sink);
    run<true, 1>("global operands, op_sel_hi:[1,0,1], beside MFMA (VGPR acc.)", 2, s0, s1, w, bad, sink);
    run<true, 2>("global operands, op_sel:[0,1,0] (high dword to both), alone", 0, s0, s1, w, bad, sink);
    run<true, 2>("global operands, op_sel:[0,1,0], beside MFMA (AGPR acc.)", 1, s0, s1, w, bad, sink);
    run<true, 2>("global operands, op_sel:[0,1,0], beside MFMA (VGPR acc.)", 2, s0, s1, w, bad, sink);
    run<false, 2>("register operands, op_sel:[0,1,0], beside MFMA (AGPR acc.)", 1, s0, s1, w, bad, sink);
    return 0;
}
