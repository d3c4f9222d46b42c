// comm_attn.hip -- K5 (key/query heads), K6 (communication graph) and K7 (feature fusion).
//
//  K5  w2c_linear_f32      km_generator / linear MLP layers (agent.py:150-159, 167-178)
//  K6  w2c_comm_graph      MIMOGeneralDotProductAttention scores + softmax over KEYS
//                          (agent.py:256,268,274), +0.001*I tie-break (agent.py:1164-1167),
//                          argmax_select / activated_select coefficients (agent.py:1036-1078),
//                          and the diagonal-masked variant of MIMOcomWho (agent.py:299-343)
//  K7  w2c_fuse_values     sum_k coef[b,k,q] * V[b,k] (agent.py:276-284) + agents2batch
//                          (agent.py:1080-1086), without the reference's [B,Nk,Nq,C,h,w] temporary
//
// These are latency / HBM bound (SURVEY.md section 2 kernel table): f32 VALU math, wave64
// shuffle reductions, 16-byte coalesced bf16 traffic for V.  No MFMA here on purpose.
#include "w2c_common.h"

namespace {

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    return v;
}

// ---------------------------------------------------------------- K5: y = act(x W^T + b)
// One wave per output column o; the wave keeps all M rows' partial sums in registers
// (MR rows per pass), lanes stride over K four elements at a time.
template <int MR, bool XBF16>
__global__ __launch_bounds__(256) void linear_kernel(const void* __restrict__ xv, int x_stride, int M, int K,
                                                     const float* __restrict__ w, const float* __restrict__ bias,
                                                     int O, int relu, float* __restrict__ y) {
    const int lane = threadIdx.x & 63;
    const int o = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (o >= O) return;
    const float* wrow = w + (size_t)o * K;
    for (int m0 = 0; m0 < M; m0 += MR) {
        float acc[MR];
#pragma unroll
        for (int r = 0; r < MR; ++r) acc[r] = 0.f;
        for (int k = lane * 4; k < K; k += 256) {
            const f32x4_t wv = *reinterpret_cast<const f32x4_t*>(wrow + k);
#pragma unroll
            for (int r = 0; r < MR; ++r) {
                if (m0 + r < M) {
                    float x0, x1, x2, x3;
                    if (XBF16) {
                        const uint2 u = *reinterpret_cast<const uint2*>(
                            reinterpret_cast<const uint16_t*>(xv) + (size_t)(m0 + r) * x_stride + k);
                        x0 = bf16_to_f32((uint16_t)(u.x & 0xFFFFu)); x1 = bf16_to_f32((uint16_t)(u.x >> 16));
                        x2 = bf16_to_f32((uint16_t)(u.y & 0xFFFFu)); x3 = bf16_to_f32((uint16_t)(u.y >> 16));
                    } else {
                        const f32x4_t u = *reinterpret_cast<const f32x4_t*>(
                            reinterpret_cast<const float*>(xv) + (size_t)(m0 + r) * x_stride + k);
                        x0 = u[0]; x1 = u[1]; x2 = u[2]; x3 = u[3];
                    }
                    acc[r] = fmaf(x0, wv[0], acc[r]);
                    acc[r] = fmaf(x1, wv[1], acc[r]);
                    acc[r] = fmaf(x2, wv[2], acc[r]);
                    acc[r] = fmaf(x3, wv[3], acc[r]);
                }
            }
        }
#pragma unroll
        for (int r = 0; r < MR; ++r) {
            const float s = wave_sum(acc[r]);
            if (lane == 0 && m0 + r < M) {
                float v = s + bias[o];
                if (relu) v = fmaxf(v, 0.f);
                y[(size_t)(m0 + r) * O + o] = v;
            }
        }
    }
}

// Wide-K form (K >= 1024, i.e. fc.0 on the flattened policy map): one WORKGROUP per output column;
// the 256 threads split K (coalesced 16-B reads of the weight row and of every input row, which is
// L2-resident: M*K*2 bytes), MR rows accumulate per pass, then wave shuffle + cross-wave LDS
// reduction.  The one-wave-per-output form above serialises K/256 dependent iterations per row and
// ran 76 us for M=20, K=4096, O=256 (r01_a profile); this form is bound by the 4 MB weight read.
// MR = 20 rows per pass: cfg 2's 20 agent-images take ONE pass over the weight row (MR = 8 took three: 16.3 -> 14.7 us under graph replay, tools/bench_linear.py; the rest is the 20 wave reductions).
// (amdgpu_waves_per_eu(2, 2): two K-slices of row loads in flight need ~110 VGPRs; two waves per SIMD is all this launch has anyway)
template <int MR, bool XBF16>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2, 2))) void linear_widek_kernel(const void* __restrict__ xv, int x_stride, int M, int K,
                                                           const float* __restrict__ w, const float* __restrict__ bias,
                                                           int O, int relu, float* __restrict__ y) {
    __shared__ float red[4][MR];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int o = blockIdx.x;
    const float* wrow = w + (size_t)o * K;
    for (int m0 = 0; m0 < M; m0 += MR) {
        float acc[MR];
#pragma unroll
        for (int r = 0; r < MR; ++r) acc[r] = 0.f;
        // All MR row loads of a K-slice (and of the next one) are issued before the first is used: written as a plain load-use loop the
        // compiler sinks every load next to its FMAs and waits vmcnt(0) after each pair -- 40 dependent L2 round trips per workgroup.
        auto slice = [&](int k, f32x4_t& wv, uint2 (&xb)[MR], f32x4_t (&xf)[MR]) {
            wv = *reinterpret_cast<const f32x4_t*>(wrow + k);
#pragma unroll
            for (int r = 0; r < MR; ++r) {
                const int m = m0 + r < M ? m0 + r : M - 1;          // clamp: branch-free, duplicates discarded below
                if (XBF16) xb[r] = *reinterpret_cast<const uint2*>(reinterpret_cast<const uint16_t*>(xv) + (size_t)m * x_stride + k);
                else xf[r] = *reinterpret_cast<const f32x4_t*>(reinterpret_cast<const float*>(xv) + (size_t)m * x_stride + k);
            }
        };
        auto fma_slice = [&](const f32x4_t& wv, const uint2 (&xb)[MR], const f32x4_t (&xf)[MR]) {
#pragma unroll
            for (int r = 0; r < MR; ++r) {
                float x0, x1, x2, x3;
                if (XBF16) {
                    x0 = bf16_to_f32((uint16_t)(xb[r].x & 0xFFFFu)); x1 = bf16_to_f32((uint16_t)(xb[r].x >> 16));
                    x2 = bf16_to_f32((uint16_t)(xb[r].y & 0xFFFFu)); x3 = bf16_to_f32((uint16_t)(xb[r].y >> 16));
                } else {
                    x0 = xf[r][0]; x1 = xf[r][1]; x2 = xf[r][2]; x3 = xf[r][3];
                }
                acc[r] = fmaf(x0, wv[0], acc[r]);
                acc[r] = fmaf(x1, wv[1], acc[r]);
                acc[r] = fmaf(x2, wv[2], acc[r]);
                acc[r] = fmaf(x3, wv[3], acc[r]);
            }
        };
        f32x4_t wa, wb;
        uint2 xba[MR], xbb[MR];
        f32x4_t xfa[MR], xfb[MR];                          // (the unused operand type's arrays are dead code)
        int k = tid * 4;
        for (; k + 1024 < K; k += 2048) {                   // two K-slices per trip, both sets of loads in flight
            slice(k, wa, xba, xfa);
            slice(k + 1024, wb, xbb, xfb);
            asm volatile("" ::: "memory");
            fma_slice(wa, xba, xfa);
            fma_slice(wb, xbb, xfb);
        }
        if (k < K) {
            slice(k, wa, xba, xfa);
            asm volatile("" ::: "memory");
            fma_slice(wa, xba, xfa);
        }
        // the MR butterflies level by level (MR independent ds_bpermute per level in flight; row by row, with lane 0's store in
        // between, the compiler waits lgkmcnt(0) after each of the 6 MR shuffles), then lane 0 stores all.  Same order of additions.
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            float t[MR];
#pragma unroll
            for (int r = 0; r < MR; ++r) t[r] = __shfl_xor(acc[r], o, 64);
#pragma unroll
            for (int r = 0; r < MR; ++r) acc[r] += t[r];
        }
        if (lane == 0) {
#pragma unroll
            for (int r = 0; r < MR; ++r) red[wave][r] = acc[r];
        }
        __syncthreads();
        if (tid < MR && m0 + tid < M) {
            float v = red[0][tid] + red[1][tid] + red[2][tid] + red[3][tid] + bias[o];
            if (relu) v = fmaxf(v, 0.f);
            y[(size_t)(m0 + tid) * O + o] = v;
        }
        __syncthreads();
    }
}

// Head tail: out = W2 relu(W1 h0 + b1) + b2 for one head (km_generator / linear fc.2, fc.4;
// agent.py:152-155).  Workgroup = (image row m, 256-output slice); h0 / h1 live in LDS; weights are packed
// K-MAJOR ([K][O]) so consecutive threads read consecutive outputs of one k (coalesced, L2-resident).
// Latency-bound by construction (tiny), so every loop keeps 16 independent loads in flight; each slice
// recomputes the 128-wide hidden layer (32 K MACs) rather than synchronising through memory.
// n_part > 0: h0 is not the finished fc.0 output but n_part split-K partial sums of it (w2c_head_fc0_mfma_f32), part_stride floats
// apart: h0[k] = relu(sum_p part_p[k] + b0[k]), summed in part order (deterministic).
__device__ __forceinline__ void head_tail_body(const float* __restrict__ h0, int h0_stride, int K1,
                                               const float* __restrict__ w1t, const float* __restrict__ b1, int H1,
                                               const float* __restrict__ w2t, const float* __restrict__ b2, int O,
                                               float* __restrict__ out, int n_part = 0, long part_stride = 0,
                                               const float* __restrict__ b0 = nullptr) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float* s_h0 = reinterpret_cast<float*>(smem);          // [K1]
    float* s_h1 = s_h0 + K1;                               // [H1]
    float* s_part = s_h1 + H1;                             // [256]
    const int m = blockIdx.x, tid = threadIdx.x;
    if (n_part > 0) {
        for (int k = tid; k < K1; k += 256) {
            const float* src = h0 + (size_t)m * h0_stride + k;
            float v = 0.f;
            int pp = 0;
            for (; pp + 8 <= n_part; pp += 8) {                        // 8 partials in flight, added in part order
                float t[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) t[u] = src[(size_t)(pp + u) * part_stride];
#pragma unroll
                for (int u = 0; u < 8; ++u) v += t[u];
            }
            for (; pp < n_part; ++pp) v += src[(size_t)pp * part_stride];
            s_h0[k] = fmaxf(v + b0[k], 0.f);
        }
    } else {
        for (int k = tid; k < K1; k += 256) s_h0[k] = h0[(size_t)m * h0_stride + k];
    }
    __syncthreads();
    // layer 1: thread = (output j, K-partition)
    const int parts = 256 / H1;                            // H1 <= 256
    {
        const int j = tid % H1, part = tid / H1;
        float a[4] = {0.f, 0.f, 0.f, 0.f};
        if (part < parts) {
            const int kn = (K1 - part + parts - 1) / parts;           // this partition's k count: k = part + i*parts
            int i = 0;
            for (; i + 31 < kn; i += 32) {                             // 32 loads in flight (same summation order as the 16-wide step)
                float wv[32];
#pragma unroll
                for (int u = 0; u < 32; ++u) wv[u] = w1t[(size_t)(part + (i + u) * parts) * H1 + j];
#pragma unroll
                for (int u = 0; u < 32; ++u) a[u & 3] = fmaf(wv[u], s_h0[part + (i + u) * parts], a[u & 3]);
            }
            for (; i + 15 < kn; i += 16) {
                float wv[16];
#pragma unroll
                for (int u = 0; u < 16; ++u) wv[u] = w1t[(size_t)(part + (i + u) * parts) * H1 + j];
#pragma unroll
                for (int u = 0; u < 16; ++u) a[u & 3] = fmaf(wv[u], s_h0[part + (i + u) * parts], a[u & 3]);
            }
            for (; i < kn; ++i) a[0] = fmaf(w1t[(size_t)(part + i * parts) * H1 + j], s_h0[part + i * parts], a[0]);
        }
        s_part[tid] = (a[0] + a[1]) + (a[2] + a[3]);
    }
    __syncthreads();
    if (tid < H1) {
        float v = b1[tid];
        for (int pp = 0; pp < parts; ++pp) v += s_part[pp * H1 + tid];
        s_h1[tid] = fmaxf(v, 0.f);
    }
    __syncthreads();
    // layer 2: thread = one output of this workgroup's slice, K = H1
    const int o = blockIdx.y * 256 + tid;
    if (o < O) {
        float a[4] = {0.f, 0.f, 0.f, 0.f};
        int k = 0;
        for (; k + 31 < H1; k += 32) {
            float wv[32];
#pragma unroll
            for (int u = 0; u < 32; ++u) wv[u] = w2t[(size_t)(k + u) * O + o];
#pragma unroll
            for (int u = 0; u < 32; ++u) a[u & 3] = fmaf(wv[u], s_h1[k + u], a[u & 3]);
        }
        for (; k + 15 < H1; k += 16) {
            float wv[16];
#pragma unroll
            for (int u = 0; u < 16; ++u) wv[u] = w2t[(size_t)(k + u) * O + o];
#pragma unroll
            for (int u = 0; u < 16; ++u) a[u & 3] = fmaf(wv[u], s_h1[k + u], a[u & 3]);
        }
        for (; k < H1; ++k) a[0] = fmaf(w2t[(size_t)k * O + o], s_h1[k], a[0]);
        out[(size_t)m * O + o] = (a[0] + a[1]) + (a[2] + a[3]) + b2[o];
    }
}

__global__ __launch_bounds__(256) void head_tail_kernel(const float* __restrict__ h0, int h0_stride, int K1,
                                                        const float* __restrict__ w1t, const float* __restrict__ b1, int H1,
                                                        const float* __restrict__ w2t, const float* __restrict__ b2, int O,
                                                        float* __restrict__ out) {
    head_tail_body(h0, h0_stride, K1, w1t, b1, H1, w2t, b2, O, out);
}

// both heads (key and query tails read disjoint column ranges of the same fc.0 output) in ONE launch: blockIdx.z = head.
struct HeadTailSet { const float* w1t; const float* b1; const float* w2t; const float* b2; float* out; int col_off; int O; };
__global__ __launch_bounds__(256) void head_tail2_kernel(const float* __restrict__ h0, int h0_stride, int K1, int H1,
                                                         HeadTailSet a, HeadTailSet b, int n_part, long part_stride,
                                                         const float* __restrict__ b0) {
    const HeadTailSet& s = blockIdx.z == 0 ? a : b;
    if ((int)blockIdx.y * 256 >= s.O) return;              // the narrower head has fewer 256-output slices (whole workgroup)
    head_tail_body(h0 + s.col_off, h0_stride, K1, s.w1t, s.b1, H1, s.w2t, s.b2, s.O, s.out, n_part, part_stride,
                   b0 ? b0 + s.col_off : nullptr);
}

// The two heads' tails on split-K partials of fc.0, WIDE form (round 4): 1024 threads per (row, head) instead of 256, so that each of
// the three dependent stages -- sum of the partials, fc.2 (K = 256), fc.4 (K = 128) -- is ONE round of loads per thread (the 256-thread
// body above walks 4 + 4 + 4 rounds of 32 loads: ~10 us of pure latency at the end of the policy chain).  Same three stages, fixed
// summation orders (deterministic); taken when 1024 % K1 == 0, 1024 % H1 == 0, n_part % (1024 / K1) == 0 and both heads have <= 64 outputs.
__global__ __launch_bounds__(1024) void head_tail2w_kernel(const float* __restrict__ part, int n_part, long part_stride,
                                                           const float* __restrict__ b0, int h0_stride, int K1, int H1,
                                                           HeadTailSet a, HeadTailSet b) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float* s_h0 = reinterpret_cast<float*>(smem);          // [K1]
    float* s_h1 = s_h0 + K1;                               // [H1]
    float* s_part = s_h1 + H1;                             // [1024]
    const HeadTailSet& s = blockIdx.y == 0 ? a : b;
    const int m = blockIdx.x, tid = threadIdx.x;
    // The weights of stages 1 and 2 do not depend on the activations: this thread's 32 + 16 values are requested up front, so that their
    // memory round trips run under stage 0's instead of two more behind it (three dependent round trips -> one; the dispatched shape:
    // K1 / (1024 / H1) == 32).  Same values, same fmaf order.
    // K1 and H1 divide 1024 (the launcher's condition for this kernel): powers of two -- every division below is a shift (round 6: eight
    // per-lane run-time divisions were ~0.5 us of this 8 us launch at the end of the policy chain)
    const int shH = __builtin_ctz((unsigned)H1), shK = __builtin_ctz((unsigned)K1);
    const int P1 = 1024 >> shH, j1 = tid & (H1 - 1), p1 = tid >> shH;
    const int kn1 = (K1 - p1 + P1 - 1) >> (10 - shH);
    const bool pre = kn1 == 32;
    float wv1[32], wv2[16];
    if (pre) {
#pragma unroll
        for (int u = 0; u < 32; ++u) wv1[u] = s.w1t[(size_t)(p1 + u * P1) * H1 + j1];
    }
    {
        const int o = tid & 63, p = tid >> 6;
#pragma unroll
        for (int u = 0; u < 16; ++u) {
            const int k = p + u * 16;
            wv2[u] = (o < s.O && k < H1) ? s.w2t[(size_t)k * s.O + o] : 0.f;
        }
    }
    {   // stage 0: h0[k] = relu(sum_p part_p[m][col_off + k] + b0): thread = (k, group of n_part / G consecutive partials)
        const int G = 1024 >> shK, per = n_part >> (10 - shK);
        const int k = tid & (K1 - 1), g = tid >> shK;
        const float* src = part + (size_t)m * h0_stride + s.col_off + k + (size_t)g * per * part_stride;
        float v = 0.f;
        int pp = 0;
        for (; pp + 4 <= per; pp += 4) {
            float t[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) t[u] = src[(size_t)(pp + u) * part_stride];
#pragma unroll
            for (int u = 0; u < 4; ++u) v += t[u];
        }
        for (; pp < per; ++pp) v += src[(size_t)pp * part_stride];
        s_part[tid] = v;
        __syncthreads();
        if (tid < K1) {
            float t = s_part[tid];
            for (int gg = 1; gg < G; ++gg) t += s_part[gg * K1 + tid];
            s_h0[tid] = fmaxf(t + b0[s.col_off + tid], 0.f);
        }
        __syncthreads();
    }
    {   // stage 1: h1 = relu(W1 h0 + b1): thread = (output j, K partition): k = p + i P
        const int P = 1024 >> shH, j = tid & (H1 - 1), p = tid >> shH;
        const int kn = (K1 - p + P - 1) >> (10 - shH);
        float acc[4] = {0.f, 0.f, 0.f, 0.f};
        int i = 0;
        if (pre) {
#pragma unroll
            for (int u = 0; u < 32; ++u) acc[u & 3] = fmaf(wv1[u], s_h0[p + u * P], acc[u & 3]);
            i = 32;
        }
        for (; i + 32 <= kn; i += 32) {
            float wv[32];
#pragma unroll
            for (int u = 0; u < 32; ++u) wv[u] = s.w1t[(size_t)(p + (i + u) * P) * H1 + j];
#pragma unroll
            for (int u = 0; u < 32; ++u) acc[u & 3] = fmaf(wv[u], s_h0[p + (i + u) * P], acc[u & 3]);
        }
        for (; i < kn; ++i) acc[0] = fmaf(s.w1t[(size_t)(p + i * P) * H1 + j], s_h0[p + i * P], acc[0]);
        s_part[tid] = (acc[0] + acc[1]) + (acc[2] + acc[3]);
        __syncthreads();
        if (tid < H1) {
            float v = s.b1[tid];
            for (int pp = 0; pp < P; ++pp) v += s_part[pp * H1 + tid];
            s_h1[tid] = fmaxf(v, 0.f);
        }
        __syncthreads();
    }
    {   // stage 2: out = W2 h1 + b2 (O <= 64): thread = (output o, K partition of 16)
        const int o = tid & 63, p = tid >> 6;              // 16 partitions
        float acc = 0.f;
        if (o < s.O) {
#pragma unroll
            for (int u = 0; u < 16; ++u) {
                const int k = p + u * 16;
                if (k < H1) acc = fmaf(wv2[u], s_h1[k], acc);
            }
        }
        s_part[tid] = acc;
        __syncthreads();
        if (tid < s.O) {
            float v = s.b2[tid];
            for (int pp = 0; pp < 16; ++pp) v += s_part[pp * 64 + tid];
            s.out[(size_t)m * s.O + tid] = v;
        }
    }
}

// ---------------------------------------------------------------- K5, round 4: fc.0 of the heads on the f32 matrix pipe
// y[m][o] = sum_k x[m][k] W[o][k]  (M = 20 .. 128 rows, K = 4096, O = 512: 84 MFLOP, 8 MB of f32 weights).  The VALU form above
// (linear_widek_kernel: one workgroup per output column, every workgroup re-reads all M rows of x) took 25-28 us at the END of the
// policy chain, the forward's critical path.  v_mfma_f32_32x32x2_f32 is exact f32 (an fmaf chain per output, MI355X_MICROARCH.md) at the
// f32 vector rate, which is plenty here: the kernel is bound by streaming the weights once.
//   grid = (O / 32 column tiles, KS K-splits); workgroup = 4 waves, wave w takes the 64-deep K slice [ks * 256 + 64 w, +64);
//   weights arrive FRAGMENT-PACKED (HeadPlan): block (column tile, 8 k) = 64 lanes x 16 B, lane (o % 32, half) holds
//   W[o][8 q + 4 half + 0..3]: one coalesced 1 KB load feeds 4 MFMAs; x: lane (row % 32, half) loads the 4 bf16 at the same k;
//   MFMA e of a block contracts k in {8 q + e, 8 q + 4 + e} -- the same pairing for both operands, which is all a dot product needs;
//   the 4 waves' partial tiles meet in LDS, are added in wave order and stored as ONE partial per (K-split, row, column):
//   part[ks][m][o].  The tail kernel adds the KS partials in order (+ bias, ReLU).  Deterministic; independent of M (an output's
//   sum order is fixed by (ks, wave, q, e)), so a rank's shard rounds like the unsharded batch.
template <int RB>                                          // 32-row blocks of x per workgroup (grid.z workgroups cover M rows)
__global__ __launch_bounds__(256) void head_fc0_mfma_kernel(const uint16_t* __restrict__ x, int x_stride, int M, int K,
                                                            const float* __restrict__ wf, int O, float* __restrict__ part) {
#if defined(__HIP_DEVICE_COMPILE__)
    __shared__ __attribute__((aligned(16))) float red[4][RB][16][64];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l31 = lane & 31, lhi = lane >> 5;
    const int ot = blockIdx.x, ks = blockIdx.y, KS = gridDim.y;
    const int mb = blockIdx.z * (32 * RB);                   // first row of this workgroup's row block (any M: grid.z row blocks)
    const int kw = K / KS / 4;                               // K slice of a wave (multiple of 8)
    const int k0 = ks * (K / KS) + wave * kw;
    const int nq = kw >> 3;
    const f32x4_t* wp = reinterpret_cast<const f32x4_t*>(wf) + ((size_t)ot * (K >> 3) + (k0 >> 3)) * 64 + lane;
    const uint16_t* xp[RB];
#pragma unroll
    for (int rb = 0; rb < RB; ++rb) {
        const int mr = mb + rb * 32 + l31;
        const int m = mr < M ? mr : M - 1;                                // clamp: rows past M are computed and dropped
        xp[rb] = x + (size_t)m * x_stride + k0 + 4 * lhi;
    }
    f32x16_t acc[RB];
#pragma unroll
    for (int rb = 0; rb < RB; ++rb)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[rb][e] = 0.f;
    constexpr int QD = RB == 1 ? 8 : 4;                      // blocks (8 k each) of loads in flight: a 64-deep K slice is ONE round for M <= 32
    for (int q = 0; q < nq; q += QD) {
        f32x4_t wv[QD];
        uint2 xv[QD][RB];
#pragma unroll
        for (int u = 0; u < QD; ++u) {
            wv[u] = wp[(size_t)(q + u) * 64];
#pragma unroll
            for (int rb = 0; rb < RB; ++rb) xv[u][rb] = *reinterpret_cast<const uint2*>(xp[rb] + (q + u) * 8);
        }
#pragma unroll
        for (int u = 0; u < QD; ++u)
#pragma unroll
            for (int rb = 0; rb < RB; ++rb) {
                const float xe[4] = {__uint_as_float(xv[u][rb].x << 16), __uint_as_float(xv[u][rb].x & 0xFFFF0000u),
                                     __uint_as_float(xv[u][rb].y << 16), __uint_as_float(xv[u][rb].y & 0xFFFF0000u)};
#pragma unroll
                for (int e = 0; e < 4; ++e) acc[rb] = __builtin_amdgcn_mfma_f32_32x32x2f32(wv[u][e], xe[e], acc[rb], 0, 0, 0);
            }
    }
    // D layout: lane = row l31 of the block, element r = column (r & 3) + 8 (r >> 2) + 4 lhi of the tile
#pragma unroll
    for (int rb = 0; rb < RB; ++rb)
#pragma unroll
        for (int r = 0; r < 16; ++r) red[wave][rb][r][lane] = acc[rb][r];
    __syncthreads();
    for (int i = tid; i < RB * 16 * 64; i += 256) {
        const int ln = i & 63, r = (i >> 6) & 15, rb = i >> 10;
        const float v = ((red[0][rb][r][ln] + red[1][rb][r][ln]) + red[2][rb][r][ln]) + red[3][rb][r][ln];
        const int m = mb + rb * 32 + (ln & 31), o = ot * 32 + (r & 3) + 8 * (r >> 2) + 4 * (ln >> 5);
        if (m < M) part[((size_t)ks * M + m) * O + o] = v;
    }
#endif
}

// ---------------------------------------------------------------- K6: communication graph
// score[k][q] = key[k] . (Wq query[q] + bq) = (Wq^T key[k]) . query[q] + key[k] . bq
// so only T[k] = Wq^T key[k] (Dq values) and t0[k] = key[k].bq are formed: N*Dk*Dq MACs and no
// [N][Dk] projected-query buffer.
//   key_project_kernel : one workgroup per (agent k, sample b) row -> T row of Dq+1 floats.
//                        thread = (column j, K-partition): consecutive lanes read consecutive
//                        columns of a Wq row (coalesced); partitions reduced through LDS.
//   comm_graph_kernel  : one workgroup per sample b; one wave per query column, lanes = keys,
//                        softmax / argmax by wave64 shuffles.  N <= 64.
constexpr int MAXN = 64;

__global__ __launch_bounds__(256) void key_project_kernel(const float* __restrict__ key, const float* __restrict__ wq,
                                                          const float* __restrict__ bq, int Dq, int Dk,
                                                          float* __restrict__ T) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float* red = reinterpret_cast<float*>(smem);          // [parts][Dq]
    const int row = blockIdx.x;                            // agent-major k*B + b
    const float* krow = key + (size_t)row * Dk;
    const int tid = threadIdx.x;
    const int parts = 256 / Dq;                            // Dq <= 256
    const int j = tid % Dq, part = tid / Dq;
    if (part < parts) {
        float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;      // 4 independent chains: loads stay in flight
        int d = part;
        for (; d + 3 * parts < Dk; d += 4 * parts) {
            s0 = fmaf(wq[(size_t)d * Dq + j], krow[d], s0);
            s1 = fmaf(wq[(size_t)(d + parts) * Dq + j], krow[d + parts], s1);
            s2 = fmaf(wq[(size_t)(d + 2 * parts) * Dq + j], krow[d + 2 * parts], s2);
            s3 = fmaf(wq[(size_t)(d + 3 * parts) * Dq + j], krow[d + 3 * parts], s3);
        }
        for (; d < Dk; d += parts) s0 = fmaf(wq[(size_t)d * Dq + j], krow[d], s0);
        red[part * Dq + j] = (s0 + s1) + (s2 + s3);
    }
    // bias column: every wave takes a strided share, wave 0 finishes
    float sb = 0.f;
    for (int d = tid; d < Dk; d += 256) sb = fmaf(bq[d], krow[d], sb);
    sb = wave_sum(sb);
    float* redb = red + parts * Dq;
    if ((tid & 63) == 0) redb[tid >> 6] = sb;
    __syncthreads();
    if (tid < Dq) {
        float s = 0.f;
        for (int pp = 0; pp < parts; ++pp) s += red[pp * Dq + tid];
        T[(size_t)row * (Dq + 1) + tid] = s;
    }
    if (tid == 0) T[(size_t)row * (Dq + 1) + Dq] = redb[0] + redb[1] + redb[2] + redb[3];
}

__device__ __forceinline__ void wave_argmax(float& best, int& bidx) {
    // first maximal index, like torch.argmax / Tensor.max(dim)[1] on CPU
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const float ob = __shfl_xor(best, o, 64);
        const int oi = __shfl_xor(bidx, o, 64);
        if (ob > best || (ob == best && oi < bidx)) { best = ob; bidx = oi; }
    }
}

// One query column of the graph for sample b (lanes = keys): -> this lane's prob_action entry `pr`, fusion coefficient `cf`,
// the column's action, and whether the entry counts towards num_connect.  Shared by comm_graph_kernel and graph_fuse_kernel so
// the two produce the same bits.
__device__ __forceinline__ bool graph_column(const float* __restrict__ query, const float* __restrict__ T, int B, int N, int Dq,
                                             int who, int mode, float thres, float tie_bias, int b, int q, int ql, int lane,
                                             float& pr, float& cf, int& act) {
    const int k = lane;
    float s = -INFINITY;
    if (k < N) {
        const float* trow = T + (size_t)(k * B + b) * (Dq + 1);
        float d = trow[Dq];
        if (query) {
            const float* qrow = query + (size_t)(ql * B + b) * Dq;   // rows of the produced query agents only
            // 16 key / 16 query values in flight per step (a plain j loop issues one dependent pair of loads at a time: 32 L2 round
            // trips per column); the FMAs stay in j order
            int j = 0;
            for (; j + 16 <= Dq; j += 16) {
                float tv[16], qv[16];
#pragma unroll
                for (int u = 0; u < 16; ++u) { tv[u] = trow[j + u]; qv[u] = qrow[j + u]; }
#pragma unroll
                for (int u = 0; u < 16; ++u) d = fmaf(tv[u], qv[u], d);
            }
            for (; j < Dq; ++j) d = fmaf(trow[j], qrow[j], d);
        } else {
            for (int j = 0; j < Dq; ++j) d += trow[j];                 // all-ones query (agent.py:1143,1370)
        }
        s = (who && k == q) ? -INFINITY : d;                           // who: diagonal stripped (agent.py:310-318)
    }
    const float mx = wave_max(s);
    const float e = (k < N && s > -INFINITY) ? expf(s - mx) : 0.f;
    const float den = wave_sum(e);
    const float p0 = e / den;                                           // softmax over keys (agent.py:274)
    pr = (k < N && !who && k == q) ? p0 + tie_bias : p0;                // prob_action (agent.py:1164-1167)
    float best = (k < N) ? pr : -INFINITY;
    int bidx = (k < N) ? k : 0x7fffffff;
    wave_argmax(best, bidx);
    if (mode == 0) cf = p0;                                             // softmax / training (agent.py:1155-1161)
    else if (mode == 1) cf = (k == bidx) ? 1.f : 0.f;                   // argmax_select (agent.py:1040-1041)
    else cf = (pr > thres) ? pr : 0.f;                                  // activated_select (agent.py:1062)
    act = bidx;
    if (mode != 0 && !who) {
        // MIMOcom returns argmax over keys of the connect matrix (agent.py:1189,1200);
        // MIMOcomWho always argmax(prob_action) (agent.py:1389,1408,1419)
        float cb = (k < N) ? cf : -INFINITY;
        int ci = (k < N) ? k : 0x7fffffff;
        wave_argmax(cb, ci);
        act = ci;
    }
    return k < N && k != q && cf != 0.f;
}

__global__ __launch_bounds__(256) void comm_graph_kernel(const float* __restrict__ query, const float* __restrict__ T,
                                                         int B, int N, int Dq, int who, int mode, float thres,
                                                         float tie_bias, int q_lo, int q_n,
                                                         float* __restrict__ prob, float* __restrict__ coef,
                                                         int64_t* __restrict__ action, int32_t* __restrict__ nnz) {
    __shared__ int cnt;
    const int b = blockIdx.x;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    if (tid == 0) cnt = 0;
    __syncthreads();
    int local_nnz = 0;
    for (int ql = wave; ql < q_n; ql += 4) {               // one wave per query column
        float pr, cf;
        int act;
        if (graph_column(query, T, B, N, Dq, who, mode, thres, tie_bias, b, q_lo + ql, ql, lane, pr, cf, act)) ++local_nnz;
        if (lane < N) {
            const size_t o = ((size_t)b * N + lane) * q_n + ql;
            prob[o] = pr;
            coef[o] = cf;
        }
        if (lane == 0) action[(size_t)b * q_n + ql] = act;
    }
    local_nnz = (int)wave_sum((float)local_nnz);
    if (lane == 0 && local_nnz) atomicAdd(&cnt, local_nnz);
    __syncthreads();
    if (tid == 0) nnz[b] = cnt;
}

// ---------------------------------------------------------------- K7: fusion
// thread = 8 channels (16 B) of one (b, pixel); loops queries, skipping zero coefficients
// (wave-uniform branch: coef depends on (b,k,q) only and a workgroup never spans two b).
__global__ __launch_bounds__(256) void fuse_kernel(const uint16_t* __restrict__ v, int vcs, const float* __restrict__ coef,
                                                   int B, int N, int q_lo, int q_n, int hw, int C, int append_own,
                                                   uint16_t* __restrict__ out, int ocs) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float* cs = reinterpret_cast<float*>(smem);          // [N][q_n] for this b
    const int b = blockIdx.y;
    for (int i = threadIdx.x; i < N * q_n; i += 256) cs[i] = coef[(size_t)b * N * q_n + i];
    __syncthreads();
    const int CG = C >> 3;
    const int total = hw * CG;
    if (N <= 8 && q_n <= 8) {
        // small graphs (cfg 2: 5 x 5): every key's 16 bytes are loaded ONCE, all together, and reused by all queries -- the general
        // loop below re-reads them per query behind a store (q_n dependent load rounds).  Same fmaf order per output: same bits.
        for (int id = blockIdx.x * 256 + threadIdx.x; id < total; id += gridDim.x * 256) {
            const int px = id / CG, cg = id - px * CG;
            uint4 vk[8];
#pragma unroll
            for (int k = 0; k < 8; ++k)
                vk[k] = k < N ? *reinterpret_cast<const uint4*>(v + ((size_t)(k * B + b) * hw + px) * vcs + cg * 8) : make_uint4(0, 0, 0, 0);
#pragma unroll
            for (int ql = 0; ql < 8; ++ql) {
                if (ql >= q_n) break;
                float acc[8];
#pragma unroll
                for (int e = 0; e < 8; ++e) acc[e] = 0.f;
#pragma unroll
                for (int k = 0; k < 8; ++k) {
                    if (k >= N) break;
                    const float c = cs[k * q_n + ql];
                    if (c == 0.f) continue;
                    const uint32_t wv[4] = {vk[k].x, vk[k].y, vk[k].z, vk[k].w};
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        acc[2 * e] = fmaf(c, bf16_to_f32((uint16_t)(wv[e] & 0xFFFFu)), acc[2 * e]);
                        acc[2 * e + 1] = fmaf(c, bf16_to_f32((uint16_t)(wv[e] >> 16)), acc[2 * e + 1]);
                    }
                }
                uint4 o;
                o.x = pack_bf16x2(acc[0], acc[1]); o.y = pack_bf16x2(acc[2], acc[3]);
                o.z = pack_bf16x2(acc[4], acc[5]); o.w = pack_bf16x2(acc[6], acc[7]);
                uint16_t* orow = out + ((size_t)(ql * B + b) * hw + px) * ocs;
                *reinterpret_cast<uint4*>(orow + cg * 8) = o;
                if (append_own) {
                    const uint4 own = *reinterpret_cast<const uint4*>(v + ((size_t)((q_lo + ql) * B + b) * hw + px) * vcs + cg * 8);
                    *reinterpret_cast<uint4*>(orow + C + cg * 8) = own;
                }
            }
        }
        return;
    }
    for (int id = blockIdx.x * 256 + threadIdx.x; id < total; id += gridDim.x * 256) {
        const int px = id / CG, cg = id - px * CG;
        for (int ql = 0; ql < q_n; ++ql) {
            float acc[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) acc[e] = 0.f;
            for (int k = 0; k < N; ++k) {
                const float c = cs[k * q_n + ql];
                if (c == 0.f) continue;
                const uint4 u = *reinterpret_cast<const uint4*>(v + ((size_t)(k * B + b) * hw + px) * vcs + cg * 8);
                const uint32_t wv[4] = {u.x, u.y, u.z, u.w};
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    acc[2 * e] = fmaf(c, bf16_to_f32((uint16_t)(wv[e] & 0xFFFFu)), acc[2 * e]);
                    acc[2 * e + 1] = fmaf(c, bf16_to_f32((uint16_t)(wv[e] >> 16)), acc[2 * e + 1]);
                }
            }
            uint4 o;
            o.x = pack_bf16x2(acc[0], acc[1]); o.y = pack_bf16x2(acc[2], acc[3]);
            o.z = pack_bf16x2(acc[4], acc[5]); o.w = pack_bf16x2(acc[6], acc[7]);
            uint16_t* orow = out + ((size_t)(ql * B + b) * hw + px) * ocs;
            *reinterpret_cast<uint4*>(orow + cg * 8) = o;
            if (append_own) {
                const uint4 own = *reinterpret_cast<const uint4*>(v + ((size_t)((q_lo + ql) * B + b) * hw + px) * vcs + cg * 8);
                *reinterpret_cast<uint4*>(orow + C + cg * 8) = own;
            }
        }
    }
}

// K6 + K7 in one launch: every workgroup of sample b recomputes b's graph (N x q_n dot products of Dq+1 values: ~2 us of
// latency, far less than a kernel boundary plus a 4-workgroup launch of its own) into LDS and fuses its share of the pixels with
// it; workgroup x = 0 of each sample also writes prob / coef / action / nnz.  Same arithmetic as the two separate kernels.
__global__ __launch_bounds__(256) void graph_fuse_kernel(const float* __restrict__ query, const float* __restrict__ T,
                                                         int B, int N, int Dq, int who, int mode, float thres, float tie_bias,
                                                         int q_lo, int q_n, float* __restrict__ prob, float* __restrict__ coef,
                                                         int64_t* __restrict__ action, int32_t* __restrict__ nnz,
                                                         const uint16_t* __restrict__ v, int vcs, int hw, int C, int append_own,
                                                         uint16_t* __restrict__ out, int ocs) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float* cs = reinterpret_cast<float*>(smem);          // [N][q_n] for this b, then one int
    int* cnt = reinterpret_cast<int*>(cs + N * q_n);
    const int b = blockIdx.y;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const bool writer = blockIdx.x == 0;
    if (tid == 0) *cnt = 0;
    __syncthreads();
    int local_nnz = 0;
    for (int ql = wave; ql < q_n; ql += 4) {
        float pr, cf;
        int act;
        if (graph_column(query, T, B, N, Dq, who, mode, thres, tie_bias, b, q_lo + ql, ql, lane, pr, cf, act)) ++local_nnz;
        if (lane < N) {
            cs[lane * q_n + ql] = cf;
            if (writer) {
                const size_t o = ((size_t)b * N + lane) * q_n + ql;
                prob[o] = pr;
                coef[o] = cf;
            }
        }
        if (writer && lane == 0) action[(size_t)b * q_n + ql] = act;
    }
    if (writer) {
        local_nnz = (int)wave_sum((float)local_nnz);
        if (lane == 0 && local_nnz) atomicAdd(cnt, local_nnz);
    }
    __syncthreads();
    if (writer && tid == 0) nnz[b] = *cnt;
    const int CG = C >> 3;
    const int total = hw * CG;
    for (int id = blockIdx.x * 256 + threadIdx.x; id < total; id += gridDim.x * 256) {
        const int px = id / CG, cg = id - px * CG;
        for (int ql = 0; ql < q_n; ++ql) {
            float acc[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) acc[e] = 0.f;
            for (int k = 0; k < N; ++k) {
                const float c = cs[k * q_n + ql];
                if (c == 0.f) continue;
                const uint4 u = *reinterpret_cast<const uint4*>(v + ((size_t)(k * B + b) * hw + px) * vcs + cg * 8);
                const uint32_t wv[4] = {u.x, u.y, u.z, u.w};
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    acc[2 * e] = fmaf(c, bf16_to_f32((uint16_t)(wv[e] & 0xFFFFu)), acc[2 * e]);
                    acc[2 * e + 1] = fmaf(c, bf16_to_f32((uint16_t)(wv[e] >> 16)), acc[2 * e + 1]);
                }
            }
            uint4 o;
            o.x = pack_bf16x2(acc[0], acc[1]); o.y = pack_bf16x2(acc[2], acc[3]);
            o.z = pack_bf16x2(acc[4], acc[5]); o.w = pack_bf16x2(acc[6], acc[7]);
            uint16_t* orow = out + ((size_t)(ql * B + b) * hw + px) * ocs;
            *reinterpret_cast<uint4*>(orow + cg * 8) = o;
            if (append_own) {
                const uint4 own = *reinterpret_cast<const uint4*>(v + ((size_t)((q_lo + ql) * B + b) * hw + px) * vcs + cg * 8);
                *reinterpret_cast<uint4*>(orow + C + cg * 8) = own;
            }
        }
    }
}

// K6 + K7 + the decoder's first conv by LINEARITY (round 4).  conv0 of the decoder (backbone.py:150-152: conv3x3 + bias, then ReLU) is
// linear before its bias, and the fused map is a linear combination of the agents' value maps (agent.py:276-284), so
//     conv0(sum_k P[k,q] V[k]) = sum_k P[k,q] U[k],   U[k] = conv0_nobias(V[k])
// (MIMOcomWho, decoder input cat(fused, V[q]), agent.py:1382: + U_own[q], the conv of V[q] with the second half of the filters).
// The U maps (f32) depend on the value encoder alone and are computed while the policy chain -- the critical path -- is still
// running; after the join only this kernel is left in front of the decoder's last conv: graph column(s) as in graph_fuse_kernel, then
//     y[q] = relu(sum_k coef[k,q] U[k] (+ U_own[q]) + bias)  -> bf16.
// The weighted sum runs in f32 over f32 maps and is rounded ONCE (the V-fusing form rounds the fused map to bf16 before the conv).
// thread = 4 channels (16 B of f32) of one (b, pixel): every key's 16 bytes are loaded once and reused by all queries.
__global__ __launch_bounds__(256) void graph_fuse_u_kernel(const float* __restrict__ query, const float* __restrict__ T,
                                                           int B, int N, int Dq, int who, int mode, float thres, float tie_bias,
                                                           int q_lo, int q_n, float* __restrict__ prob, float* __restrict__ coef,
                                                           int64_t* __restrict__ action, int32_t* __restrict__ nnz,
                                                           const float* __restrict__ u, int ucs, int hw, int C,
                                                           const float* __restrict__ u_own, int own_cs,
                                                           const float* __restrict__ bias, uint16_t* __restrict__ out, int ocs,
                                                           char* pack2_arg, long act_off, long nnz_off) {
    // pack2 (optional, indirect-capable): a second, caller-owned copy of the packed prob | action | nnz (same layout as the buffer
    // prob / action / nnz point into: action at act_off bytes, nnz at nnz_off) -- the captured forward's outputs land in the caller's
    // tensor without a copy launch of their own
    char* const pack2 = w2c_resolve(pack2_arg);
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float* cs = reinterpret_cast<float*>(smem);          // [N][q_n] for this b, then one int
    int* cnt = reinterpret_cast<int*>(cs + N * q_n);
    const int b = blockIdx.y;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const bool writer = blockIdx.x == 0;
    if (tid == 0) *cnt = 0;
    // small graphs (N <= 8): the keys' 16 bytes of this thread's FIRST item do not depend on the graph -- they are requested here, so that
    // their memory round trip runs under the graph column's (loads of q / T, dot products, softmax) instead of behind it.  Every key is
    // loaded (the thresholded modes' unused maps are valid memory: zeros or the gathered map); which ones are USED is still decided by the
    // coefficients below: same fmaf sequence, same bits.
    const int CGp = C >> 2;
    const int id0 = blockIdx.x * 256 + threadIdx.x;
    f32x4_t uk0[8];
    if (N <= 8 && id0 < hw * CGp) {
        // (C / 4 is a power of two for every decoder of the path: a shift instead of a per-lane run-time division in front of the launch's
        //  first loads; any other width takes the division)
        const int px = (CGp & (CGp - 1)) == 0 ? id0 >> __builtin_ctz((unsigned)CGp) : id0 / CGp, cg = id0 - px * CGp;
#pragma unroll
        for (int k = 0; k < 8; ++k)
            uk0[k] = k < N ? *reinterpret_cast<const f32x4_t*>(u + ((size_t)(k * B + b) * hw + px) * ucs + cg * 4) : f32x4_t{0.f, 0.f, 0.f, 0.f};
    }
    __syncthreads();
    int local_nnz = 0;
    for (int ql = wave; ql < q_n; ql += 4) {
        float pr, cf;
        int act;
        if (graph_column(query, T, B, N, Dq, who, mode, thres, tie_bias, b, q_lo + ql, ql, lane, pr, cf, act)) ++local_nnz;
        if (lane < N) {
            cs[lane * q_n + ql] = cf;
            if (writer) {
                const size_t o = ((size_t)b * N + lane) * q_n + ql;
                prob[o] = pr;
                coef[o] = cf;
                if (pack2) reinterpret_cast<float*>(pack2)[o] = pr;
            }
        }
        if (writer && lane == 0) {
            action[(size_t)b * q_n + ql] = act;
            if (pack2) reinterpret_cast<int64_t*>(pack2 + act_off)[(size_t)b * q_n + ql] = act;
        }
    }
    if (writer) {
        local_nnz = (int)wave_sum((float)local_nnz);
        if (lane == 0 && local_nnz) atomicAdd(cnt, local_nnz);
    }
    __syncthreads();
    if (writer && tid == 0) {
        nnz[b] = *cnt;
        if (pack2) reinterpret_cast<int32_t*>(pack2 + nnz_off)[b] = *cnt;
    }
    const int CG = C >> 2;
    const int total = hw * CG;
    const bool cg_p2 = (CG & (CG - 1)) == 0;
    const int cg_sh = __builtin_ctz((unsigned)CG);
    for (int id = blockIdx.x * 256 + threadIdx.x; id < total; id += gridDim.x * 256) {
        const int px = cg_p2 ? id >> cg_sh : id / CG, cg = id - px * CG;
        const f32x4_t bv = *reinterpret_cast<const f32x4_t*>(bias + cg * 4);
        if (N <= 8) {
            // small graphs (cfg 2: 5 agents): all keys' 16 bytes in flight together; a key no query uses is not loaded (wave-uniform)
            f32x4_t uk[8];
            if (id == id0) {                                      // the prefetched item
#pragma unroll
                for (int k = 0; k < 8; ++k) uk[k] = uk0[k];
            } else {
#pragma unroll
                for (int k = 0; k < 8; ++k) {
                    bool used = false;
                    if (k < N)
                        for (int ql = 0; ql < q_n; ++ql) used |= cs[k * q_n + ql] != 0.f;
                    uk[k] = used ? *reinterpret_cast<const f32x4_t*>(u + ((size_t)(k * B + b) * hw + px) * ucs + cg * 4) : f32x4_t{0.f, 0.f, 0.f, 0.f};
                }
            }
            for (int ql = 0; ql < q_n; ++ql) {
                f32x4_t acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int k = 0; k < 8; ++k) {
                    if (k >= N) break;
                    const float c = cs[k * q_n + ql];
                    if (c == 0.f) continue;                           // same skip as the general loop: same fmaf sequence, same bits
#pragma unroll
                    for (int e = 0; e < 4; ++e) acc[e] = fmaf(c, uk[k][e], acc[e]);
                }
                if (u_own) {                                           // rows of the LOCAL queries: (ql * B + b)
                    const f32x4_t o4 = *reinterpret_cast<const f32x4_t*>(u_own + ((size_t)(ql * B + b) * hw + px) * own_cs + cg * 4);
#pragma unroll
                    for (int e = 0; e < 4; ++e) acc[e] += o4[e];
                }
#pragma unroll
                for (int e = 0; e < 4; ++e) acc[e] = fmaxf(acc[e] + bv[e], 0.f);
                uint2 o;
                o.x = pack_bf16x2(acc[0], acc[1]); o.y = pack_bf16x2(acc[2], acc[3]);
                *reinterpret_cast<uint2*>(out + ((size_t)(ql * B + b) * hw + px) * ocs + cg * 4) = o;
            }
            continue;
        }
        for (int ql = 0; ql < q_n; ++ql) {
            f32x4_t acc = {0.f, 0.f, 0.f, 0.f};
            for (int k = 0; k < N; ++k) {
                const float c = cs[k * q_n + ql];
                if (c == 0.f) continue;
                const f32x4_t v4 = *reinterpret_cast<const f32x4_t*>(u + ((size_t)(k * B + b) * hw + px) * ucs + cg * 4);
#pragma unroll
                for (int e = 0; e < 4; ++e) acc[e] = fmaf(c, v4[e], acc[e]);
            }
            if (u_own) {
                const f32x4_t o4 = *reinterpret_cast<const f32x4_t*>(u_own + ((size_t)(ql * B + b) * hw + px) * own_cs + cg * 4);
#pragma unroll
                for (int e = 0; e < 4; ++e) acc[e] += o4[e];
            }
#pragma unroll
            for (int e = 0; e < 4; ++e) acc[e] = fmaxf(acc[e] + bv[e], 0.f);
            uint2 o;
            o.x = pack_bf16x2(acc[0], acc[1]); o.y = pack_bf16x2(acc[2], acc[3]);
            *reinterpret_cast<uint2*>(out + ((size_t)(ql * B + b) * hw + px) * ocs + cg * 4) = o;
        }
    }
}

}  // namespace

extern "C" int w2c_linear_f32(const void* x, int x_is_bf16, int x_stride, int M, int K,
                              const float* w, const float* b, int O, int relu, float* y, w2c_stream_t stream) {
    w2c_clear_error();
    if (!x || !w || !b || !y || M <= 0 || K <= 0 || O <= 0 || (K % 4) != 0 || (x_stride % 4) != 0) return W2C_E_ARG;
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    if (K >= 1024) {
        if (x_is_bf16)
            hipLaunchKernelGGL((linear_widek_kernel<20, true>), dim3(O), dim3(256), 0, s, x, x_stride, M, K, w, b, O, relu, y);
        else
            hipLaunchKernelGGL((linear_widek_kernel<20, false>), dim3(O), dim3(256), 0, s, x, x_stride, M, K, w, b, O, relu, y);
        return w2c_launch_status();
    }
    dim3 grid((O + 3) / 4);
    if (x_is_bf16)
        hipLaunchKernelGGL((linear_kernel<16, true>), grid, dim3(256), 0, s, x, x_stride, M, K, w, b, O, relu, y);
    else
        hipLaunchKernelGGL((linear_kernel<16, false>), grid, dim3(256), 0, s, x, x_stride, M, K, w, b, O, relu, y);
    return w2c_launch_status();
}

extern "C" int w2c_head_tail_f32(const float* h0, int h0_stride, int M, int K1, const float* w1t, const float* b1, int H1,
                                 const float* w2t, const float* b2, int O, float* out, w2c_stream_t stream) {
    w2c_clear_error();
    if (!h0 || !w1t || !b1 || !w2t || !b2 || !out || M <= 0 || K1 <= 0 || H1 <= 0 || H1 > 256 || O <= 0) return W2C_E_ARG;
    if (h0_stride < K1) return W2C_E_ARG;
    const size_t lds = (size_t)(K1 + H1 + 256) * 4;
    hipLaunchKernelGGL(head_tail_kernel, dim3(M, (O + 255) / 256), dim3(256), lds, reinterpret_cast<hipStream_t>(stream),
                       h0, h0_stride, K1, w1t, b1, H1, w2t, b2, O, out);
    return w2c_launch_status();
}

extern "C" int w2c_head_tail2_f32(const float* h0, int h0_stride, int M, int K1, int H1,
                                  int col_off_a, const float* w1t_a, const float* b1_a, const float* w2t_a, const float* b2_a, int O_a,
                                  float* out_a,
                                  int col_off_b, const float* w1t_b, const float* b1_b, const float* w2t_b, const float* b2_b, int O_b,
                                  float* out_b, w2c_stream_t stream) {
    w2c_clear_error();
    if (!h0 || !w1t_a || !b1_a || !w2t_a || !b2_a || !out_a || !w1t_b || !b1_b || !w2t_b || !b2_b || !out_b) return W2C_E_ARG;
    if (M <= 0 || K1 <= 0 || H1 <= 0 || H1 > 256 || O_a <= 0 || O_b <= 0 || col_off_a < 0 || col_off_b < 0) return W2C_E_ARG;
    if (h0_stride < col_off_a + K1 || h0_stride < col_off_b + K1) return W2C_E_ARG;
    const size_t lds = (size_t)(K1 + H1 + 256) * 4;
    const int omax = O_a > O_b ? O_a : O_b;
    HeadTailSet a{w1t_a, b1_a, w2t_a, b2_a, out_a, col_off_a, O_a}, b{w1t_b, b1_b, w2t_b, b2_b, out_b, col_off_b, O_b};
    hipLaunchKernelGGL(head_tail2_kernel, dim3(M, (omax + 255) / 256, 2), dim3(256), lds, reinterpret_cast<hipStream_t>(stream),
                       h0, h0_stride, K1, H1, a, b, 0, 0L, static_cast<const float*>(nullptr));
    return w2c_launch_status();
}

extern "C" int w2c_head_fc0_mfma_f32(const uint16_t* x, int x_stride, int M, int K, const float* wfrag, int O, int ksplit,
                                     float* part, w2c_stream_t stream) {
    w2c_clear_error();
    if (!x || !wfrag || !part || M <= 0 || K <= 0 || O <= 0 || (O % 32) != 0 || ksplit <= 0) return W2C_E_ARG;
    if ((K % (ksplit * 256)) != 0 || (x_stride % 4) != 0 || (reinterpret_cast<uintptr_t>(x) & 7) || (reinterpret_cast<uintptr_t>(wfrag) & 15))
        return W2C_E_ARG;
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    if (M <= 32) hipLaunchKernelGGL((head_fc0_mfma_kernel<1>), dim3(O / 32, ksplit), dim3(256), 0, s, x, x_stride, M, K, wfrag, O, part);
    else hipLaunchKernelGGL((head_fc0_mfma_kernel<2>), dim3(O / 32, ksplit, (M + 63) / 64), dim3(256), 0, s, x, x_stride, M, K, wfrag, O, part);
    return w2c_launch_status();
}

extern "C" int w2c_head_tail2p_f32(const float* part, int n_part, long long part_stride, const float* b0, int h0_stride, int M, int K1, int H1,
                                   int col_off_a, const float* w1t_a, const float* b1_a, const float* w2t_a, const float* b2_a, int O_a,
                                   float* out_a,
                                   int col_off_b, const float* w1t_b, const float* b1_b, const float* w2t_b, const float* b2_b, int O_b,
                                   float* out_b, w2c_stream_t stream) {
    w2c_clear_error();
    if (!part || !b0 || n_part <= 0 || part_stride <= 0) return W2C_E_ARG;
    if (!w1t_a || !b1_a || !w2t_a || !b2_a || !out_a || !w1t_b || !b1_b || !w2t_b || !b2_b || !out_b) return W2C_E_ARG;
    if (M <= 0 || K1 <= 0 || H1 <= 0 || H1 > 256 || O_a <= 0 || O_b <= 0 || col_off_a < 0 || col_off_b < 0) return W2C_E_ARG;
    if (h0_stride < col_off_a + K1 || h0_stride < col_off_b + K1) return W2C_E_ARG;
    const size_t lds = (size_t)(K1 + H1 + 256) * 4;
    const int omax = O_a > O_b ? O_a : O_b;
    HeadTailSet a{w1t_a, b1_a, w2t_a, b2_a, out_a, col_off_a, O_a}, b{w1t_b, b1_b, w2t_b, b2_b, out_b, col_off_b, O_b};
    if (omax <= 64 && K1 <= 1024 && (1024 % K1) == 0 && (1024 % H1) == 0 && H1 <= 256 && (n_part % (1024 / K1)) == 0) {
        hipLaunchKernelGGL(head_tail2w_kernel, dim3(M, 2), dim3(1024), (size_t)(K1 + H1 + 1024) * 4, reinterpret_cast<hipStream_t>(stream),
                           part, n_part, (long)part_stride, b0, h0_stride, K1, H1, a, b);
        return w2c_launch_status();
    }
    hipLaunchKernelGGL(head_tail2_kernel, dim3(M, (omax + 255) / 256, 2), dim3(256), lds, reinterpret_cast<hipStream_t>(stream),
                       part, h0_stride, K1, H1, a, b, n_part, (long)part_stride, b0);
    return w2c_launch_status();
}

extern "C" int w2c_comm_graph(const float* query, const float* key, const float* wq, const float* bq,
                              int B, int N, int Dq, int Dk, int who, int mode, float thres, float tie_bias,
                              int q_lo, int q_n, float* workspace,
                              float* prob, float* coef, int64_t* action, int32_t* nnz_offdiag,
                              w2c_stream_t stream) {
    w2c_clear_error();
    if (!key || !wq || !bq || !workspace || !prob || !coef || !action || !nnz_offdiag) return W2C_E_ARG;
    if (B <= 0 || N <= 0 || N > MAXN || Dq <= 0 || Dq > 256 || Dk <= 0 || mode < 0 || mode > 2) return W2C_E_ARG;
    if (q_lo < 0 || q_n <= 0 || q_lo + q_n > N) return W2C_E_ARG;
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    const size_t lds = (size_t)((256 / Dq) * Dq + 4) * 4;
    hipLaunchKernelGGL(key_project_kernel, dim3(N * B), dim3(256), lds, s, key, wq, bq, Dq, Dk, workspace);
    hipLaunchKernelGGL(comm_graph_kernel, dim3(B), dim3(256), 0, s, query, workspace, B, N, Dq, who, mode, thres,
                       tie_bias, q_lo, q_n, prob, coef, action, nnz_offdiag);
    return w2c_launch_status();
}

extern "C" int w2c_comm_graph_projected(const float* query, const float* tproj, int B, int N, int Dq, int who, int mode,
                                        float thres, float tie_bias, int q_lo, int q_n,
                                        float* prob, float* coef, int64_t* action, int32_t* nnz_offdiag,
                                        w2c_stream_t stream) {
    w2c_clear_error();
    if (!tproj || !prob || !coef || !action || !nnz_offdiag) return W2C_E_ARG;
    if (B <= 0 || N <= 0 || N > MAXN || Dq <= 0 || mode < 0 || mode > 2) return W2C_E_ARG;
    if (q_lo < 0 || q_n <= 0 || q_lo + q_n > N) return W2C_E_ARG;
    hipLaunchKernelGGL(comm_graph_kernel, dim3(B), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), query, tproj, B, N, Dq,
                       who, mode, thres, tie_bias, q_lo, q_n, prob, coef, action, nnz_offdiag);
    return w2c_launch_status();
}

extern "C" int w2c_fuse_values(const uint16_t* v, int v_cstride, const float* coef, int B, int N, int q_lo, int q_n,
                               int hw, int C, int append_own, uint16_t* out, int out_cstride, w2c_stream_t stream) {
    w2c_clear_error();
    if (!v || !coef || !out || B <= 0 || N <= 0 || q_n <= 0 || q_lo < 0 || q_lo + q_n > N) return W2C_E_ARG;
    if (hw <= 0 || C <= 0 || (C % 8) != 0 || (v_cstride % 8) != 0 || (out_cstride % 8) != 0) return W2C_E_ARG;
    if (v_cstride < C || out_cstride < (append_own ? 2 * C : C)) return W2C_E_ARG;
    const int total = hw * (C / 8);
    int bx = (total + 255) / 256;
    if (bx > 1024) bx = 1024;
    const size_t lds = (size_t)N * q_n * 4;
    hipLaunchKernelGGL(fuse_kernel, dim3(bx, B), dim3(256), lds, reinterpret_cast<hipStream_t>(stream),
                       v, v_cstride, coef, B, N, q_lo, q_n, hw, C, append_own, out, out_cstride);
    return w2c_launch_status();
}

extern "C" int w2c_comm_graph_fuse(const float* query, const float* tproj, int B, int N, int Dq, int who, int mode,
                                   float thres, float tie_bias, int q_lo, int q_n,
                                   float* prob, float* coef, int64_t* action, int32_t* nnz_offdiag,
                                   const uint16_t* v, int v_cstride, int hw, int C, int append_own,
                                   uint16_t* out, int out_cstride, w2c_stream_t stream) {
    w2c_clear_error();
    if (!tproj || !prob || !coef || !action || !nnz_offdiag || !v || !out) return W2C_E_ARG;
    if (B <= 0 || N <= 0 || N > MAXN || Dq <= 0 || mode < 0 || mode > 2) return W2C_E_ARG;
    if (q_lo < 0 || q_n <= 0 || q_lo + q_n > N) return W2C_E_ARG;
    if (hw <= 0 || C <= 0 || (C % 8) != 0 || (v_cstride % 8) != 0 || (out_cstride % 8) != 0) return W2C_E_ARG;
    if (v_cstride < C || out_cstride < (append_own ? 2 * C : C)) return W2C_E_ARG;
    const int total = hw * (C / 8);
    int bx = (total + 255) / 256;
    if (bx > 1024) bx = 1024;
    const size_t lds = (size_t)N * q_n * 4 + 16;
    hipLaunchKernelGGL(graph_fuse_kernel, dim3(bx, B), dim3(256), lds, reinterpret_cast<hipStream_t>(stream), query, tproj, B, N, Dq,
                       who, mode, thres, tie_bias, q_lo, q_n, prob, coef, action, nnz_offdiag, v, v_cstride, hw, C, append_own,
                       out, out_cstride);
    return w2c_launch_status();
}

extern "C" int w2c_comm_graph_fuse_u(const float* query, const float* tproj, int B, int N, int Dq, int who, int mode,
                                     float thres, float tie_bias, int q_lo, int q_n,
                                     float* prob, float* coef, int64_t* action, int32_t* nnz_offdiag,
                                     const float* u, int u_cstride, int hw, int C, const float* u_own, int own_cstride,
                                     const float* bias,
                                     uint16_t* out, int out_cstride, void* pack2, long long act_off, long long nnz_off, w2c_stream_t stream) {
    w2c_clear_error();
    if (!tproj || !prob || !coef || !action || !nnz_offdiag || !u || !bias || !out) return W2C_E_ARG;
    if (B <= 0 || N <= 0 || N > MAXN || Dq <= 0 || mode < 0 || mode > 2) return W2C_E_ARG;
    if (q_lo < 0 || q_n <= 0 || q_lo + q_n > N) return W2C_E_ARG;
    if (hw <= 0 || C <= 0 || (C % 4) != 0 || (u_cstride % 4) != 0 || (out_cstride % 4) != 0 || u_cstride < C || out_cstride < C) return W2C_E_ARG;
    if (u_own && ((own_cstride % 4) != 0 || own_cstride < C || (reinterpret_cast<uintptr_t>(u_own) & 15))) return W2C_E_ARG;
    if ((reinterpret_cast<uintptr_t>(u) & 15) || (reinterpret_cast<uintptr_t>(bias) & 15) || (reinterpret_cast<uintptr_t>(out) & 7)) return W2C_E_ARG;
    const int total = hw * (C / 4);
    int bx = (total + 255) / 256;
    if (bx > 1024) bx = 1024;
    const size_t lds = (size_t)N * q_n * 4 + 16;
    hipLaunchKernelGGL(graph_fuse_u_kernel, dim3(bx, B), dim3(256), lds, reinterpret_cast<hipStream_t>(stream), query, tproj, B, N, Dq,
                       who, mode, thres, tie_bias, q_lo, q_n, prob, coef, action, nnz_offdiag, u, u_cstride, hw, C, u_own, own_cstride,
                       bias, out, out_cstride, reinterpret_cast<char*>(pack2), (long)act_off, (long)nnz_off);
    return w2c_launch_status();
}
