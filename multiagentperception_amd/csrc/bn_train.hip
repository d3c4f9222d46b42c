// bn_train.hip -- train-mode BatchNorm2d fused with ReLU and the residual add, forward and backward (SURVEY 8f rank 3,
// stage 2).  The reference trains with batch statistics over the agent-concatenated batch (agent.py:1108-1111: all N*B
// images of a forward go through one nn.BatchNorm2d call; models/utils.py:87-120, third-party resnet18 BasicBlock).
//
//   forward   mean_c, var_c over the P = M*H*W pixels of x (bf16 NHWC, the conv's raw output);
//             running_mean/var <- (1-m)*running + m*(mean, unbiased var);   y = act(gamma*(x-mean)*rstd + beta (+ residual))
//   backward  dyr = dy * [y > 0] (ReLU);  dbeta = sum dyr;  dgamma = sum dyr*xhat;  d_residual = dyr;
//             dx = gamma*rstd*(dyr - dbeta/P - xhat*dgamma/P)
//
// All HBM-bound streaming kernels: 16 B (8 channels) per lane, channels innermost, per-channel constants in registers.
// Reductions are two-level and deterministic: per-workgroup partial sums (f32) -> a one-workgroup finalize kernel that adds
// them in order in f64 (no atomics).
#include "w2c_common.h"

namespace {

constexpr int MAX_CHUNKS = 1024;

__device__ __forceinline__ void unpack8(const uint4 v, float (&f)[8]) {
    const uint32_t w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        f[2 * e] = __uint_as_float(w[e] << 16);
        f[2 * e + 1] = __uint_as_float(w[e] & 0xFFFF0000u);
    }
}
__device__ __forceinline__ uint4 pack8(const float (&f)[8]) {
    return make_uint4(pack_bf16x2(f[0], f[1]), pack_bf16x2(f[2], f[3]), pack_bf16x2(f[4], f[5]), pack_bf16x2(f[6], f[7]));
}

// partial[chunk][0][c] = sum_p a_p[c], partial[chunk][1][c] = sum_p b_p[c] over the chunk's pixels, where
//   MODE 0 (forward stats):   a = x, b = x*x
//   MODE 1 (backward sums):   a = dyr, b = dyr * xhat      (dyr = dy * [y > 0] when y != null)
template <int MODE>
__global__ __launch_bounds__(256) void bn_reduce_kernel(const uint16_t* __restrict__ x, const uint16_t* __restrict__ dy,
                                                        const uint16_t* __restrict__ y, const float* __restrict__ mean,
                                                        const float* __restrict__ rstd, long P, int C, int pix_per_chunk,
                                                        float* __restrict__ partial) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int cg = C >> 3;                                 // 8-channel groups
    const int rows = 256 / cg;                             // pixels per sweep (C <= 2048 -> cg <= 256)
    const int tg = threadIdx.x % cg, tr = threadIdx.x / cg;
    const long p0 = (long)blockIdx.x * pix_per_chunk;
    const long p1 = p0 + pix_per_chunk < P ? p0 + pix_per_chunk : P;
    float sa[8], sb[8], mu[8], rs[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) { sa[e] = 0.f; sb[e] = 0.f; mu[e] = 0.f; rs[e] = 1.f; }
    if (MODE == 1) {
#pragma unroll
        for (int e = 0; e < 8; ++e) { mu[e] = mean[tg * 8 + e]; rs[e] = rstd[tg * 8 + e]; }
    }
    if (tr < rows) {
        for (long p = p0 + tr; p < p1; p += rows) {
            float xv[8];
            unpack8(*reinterpret_cast<const uint4*>(x + p * C + tg * 8), xv);
            if (MODE == 0) {
#pragma unroll
                for (int e = 0; e < 8; ++e) { sa[e] += xv[e]; sb[e] = __builtin_fmaf(xv[e], xv[e], sb[e]); }
            } else {
                float gv[8];
                unpack8(*reinterpret_cast<const uint4*>(dy + p * C + tg * 8), gv);
                if (y) {
                    float yv[8];
                    unpack8(*reinterpret_cast<const uint4*>(y + p * C + tg * 8), yv);
#pragma unroll
                    for (int e = 0; e < 8; ++e) gv[e] = yv[e] > 0.f ? gv[e] : 0.f;
                }
#pragma unroll
                for (int e = 0; e < 8; ++e) { sa[e] += gv[e]; sb[e] = __builtin_fmaf(gv[e], (xv[e] - mu[e]) * rs[e], sb[e]); }
            }
        }
    }
    // reduce the `rows` partial rows through LDS: [tr][2][C]
    float* red = reinterpret_cast<float*>(smem);
    if (tr < rows) {
#pragma unroll
        for (int e = 0; e < 8; ++e) { red[(tr * 2 + 0) * C + tg * 8 + e] = sa[e]; red[(tr * 2 + 1) * C + tg * 8 + e] = sb[e]; }
    }
    __syncthreads();
    for (int i = threadIdx.x; i < 2 * C; i += 256) {
        float s = 0.f;
        for (int r = 0; r < rows; ++r) s += red[r * 2 * C + i];
        partial[(size_t)blockIdx.x * 2 * C + i] = s;
    }
}

// sums over the chunks of partial[k][0][c] and partial[k][1][c], one wave per channel, result valid in every lane
__device__ __forceinline__ void wave_sum2(const float* __restrict__ partial, int nchunk, int C, int c, double& s, double& q) {
    const int lane = threadIdx.x & 63;
    s = 0.0; q = 0.0;
    for (int k = lane; k < nchunk; k += 64) { s += partial[(size_t)k * 2 * C + c]; q += partial[(size_t)k * 2 * C + C + c]; }
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) {
        s += __shfl_xor(s, off, 64);
        q += __shfl_xor(q, off, 64);
    }
}

// forward finalize: mean / rstd, the folded a = gamma*rstd, b = beta - mean*a, and the running-stat update (in place)
__global__ __launch_bounds__(256) void bn_fwd_finalize_kernel(const float* __restrict__ partial, int nchunk, long P, int C, float eps,
                                                              float momentum, const float* __restrict__ gamma,
                                                              const float* __restrict__ beta, float* __restrict__ running_mean,
                                                              float* __restrict__ running_var, long long* __restrict__ num_batches_tracked,
                                                              float* __restrict__ mean, float* __restrict__ rstd, float* __restrict__ a,
                                                              float* __restrict__ b) {
    // one wave per channel: lane l adds chunks l, l+64, ... in order (f64), then a fixed shuffle tree -- deterministic
    const int c = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (c < C) {
        double s, q;
        wave_sum2(partial, nchunk, C, c, s, q);
        if ((threadIdx.x & 63) != 0) return;
        const double m = s / (double)P;
        double var = q / (double)P - m * m;
        var = var < 0.0 ? 0.0 : var;
        const float r = (float)(1.0 / sqrt(var + (double)eps));
        mean[c] = (float)m;
        rstd[c] = r;
        const float av = gamma[c] * r;
        a[c] = av;
        b[c] = beta[c] - (float)m * av;
        if (num_batches_tracked && c == 0) *num_batches_tracked += 1;      // nn.BatchNorm2d's step counter (one launch less per layer)
        if (running_mean) {
            const double unbiased = P > 1 ? var * (double)P / (double)(P - 1) : var;
            running_mean[c] = (1.f - momentum) * running_mean[c] + momentum * (float)m;
            running_var[c] = (1.f - momentum) * running_var[c] + momentum * (float)unbiased;
        }
    }
}

// backward finalize: dgamma, dbeta and the three per-channel constants of dx = k1*dyr + k2*x + k3
//   dx = A*(dyr - dbeta/P - xhat*dgamma/P),  A = gamma*rstd,  xhat = (x - mean)*rstd
//      = A*dyr + (-A*rstd*dgamma/P)*x + (-A*dbeta/P + A*rstd*mean*dgamma/P)
__global__ __launch_bounds__(256) void bn_bwd_finalize_kernel(const float* __restrict__ partial, int nchunk, long P, int C,
                                                              const float* __restrict__ gamma, const float* __restrict__ mean,
                                                              const float* __restrict__ rstd, float* __restrict__ dgamma,
                                                              float* __restrict__ dbeta, float* __restrict__ k1, float* __restrict__ k2,
                                                              float* __restrict__ k3) {
    const int c = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (c < C) {
        double s, q;
        wave_sum2(partial, nchunk, C, c, s, q);
        if ((threadIdx.x & 63) != 0) return;
        dbeta[c] = (float)s;
        dgamma[c] = (float)q;
        const double A = (double)gamma[c] * rstd[c];
        k1[c] = (float)A;
        k2[c] = (float)(-A * rstd[c] * q / (double)P);
        k3[c] = (float)(-A * s / (double)P + A * rstd[c] * mean[c] * q / (double)P);
    }
}

// ---- cross-rank (synchronised) BatchNorm, round 4: the two-phase forms.  Train-mode BatchNorm takes its statistics over the
// agent-concatenated batch (agent.py:1108-1111); when the agents are sharded over ranks the per-channel sums must be added over the ranks
// between the reduction and the finalize.  Phase A leaves the LOCAL sums as f64 [2][C] (+ the caller all-reduces them, with the pixel
// count); phase B finalizes from the GLOBAL sums.
__global__ __launch_bounds__(256) void bn_sums_kernel(const float* __restrict__ partial, int nchunk, int C, double* __restrict__ sums) {
    const int c = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (c < C) {
        double s, q;
        wave_sum2(partial, nchunk, C, c, s, q);
        if ((threadIdx.x & 63) == 0) { sums[c] = s; sums[C + c] = q; }
    }
}
__global__ __launch_bounds__(256) void bn_fwd_finalize_sums_kernel(const double* __restrict__ sums, double Ptot, int C, float eps, float momentum,
                                                                   const float* __restrict__ gamma, const float* __restrict__ beta,
                                                                   float* __restrict__ running_mean, float* __restrict__ running_var,
                                                                   long long* __restrict__ num_batches_tracked, float* __restrict__ mean,
                                                                   float* __restrict__ rstd, float* __restrict__ a, float* __restrict__ b) {
    const int c = blockIdx.x * 256 + threadIdx.x;
    if (c >= C) return;
    const double m = sums[c] / Ptot;
    double var = sums[C + c] / Ptot - m * m;
    var = var < 0.0 ? 0.0 : var;
    const float r = (float)(1.0 / sqrt(var + (double)eps));
    mean[c] = (float)m;
    rstd[c] = r;
    const float av = gamma[c] * r;
    a[c] = av;
    b[c] = beta[c] - (float)m * av;
    if (num_batches_tracked && c == 0) *num_batches_tracked += 1;
    if (running_mean) {
        const double unbiased = Ptot > 1.0 ? var * Ptot / (Ptot - 1.0) : var;
        running_mean[c] = (1.f - momentum) * running_mean[c] + momentum * (float)m;
        running_var[c] = (1.f - momentum) * running_var[c] + momentum * (float)unbiased;
    }
}
// dgamma / dbeta are this rank's OWN sums (the parameter gradients are all-reduced with every other gradient afterwards); the three
// constants of dx = k1*dyr + k2*x + k3 come from the GLOBAL sums and the global pixel count
__global__ __launch_bounds__(256) void bn_bwd_finalize_sums_kernel(const double* __restrict__ loc, const double* __restrict__ glob, double Ptot,
                                                                   int C, const float* __restrict__ gamma, const float* __restrict__ mean,
                                                                   const float* __restrict__ rstd, float* __restrict__ dgamma,
                                                                   float* __restrict__ dbeta, float* __restrict__ k1, float* __restrict__ k2,
                                                                   float* __restrict__ k3) {
    const int c = blockIdx.x * 256 + threadIdx.x;
    if (c >= C) return;
    dbeta[c] = (float)loc[c];
    dgamma[c] = (float)loc[C + c];
    const double s = glob[c], q = glob[C + c];
    const double A = (double)gamma[c] * rstd[c];
    k1[c] = (float)A;
    k2[c] = (float)(-A * rstd[c] * q / Ptot);
    k3[c] = (float)(-A * s / Ptot + A * rstd[c] * mean[c] * q / Ptot);
}

// y = act(a*x + b (+ residual))
__global__ __launch_bounds__(256) void bn_apply_kernel(const uint16_t* __restrict__ x, const float* __restrict__ a,
                                                       const float* __restrict__ b, const uint16_t* __restrict__ res, int relu, long P,
                                                       int C, uint16_t* __restrict__ y) {
    const int cg = C >> 3;
    const long total = P * cg;
    for (long id = (long)blockIdx.x * 256 + threadIdx.x; id < total; id += (long)gridDim.x * 256) {
        const int g = (int)(id % cg);
        float xv[8], rv[8], o[8];
        unpack8(*reinterpret_cast<const uint4*>(x + id * 8), xv);
        if (res) unpack8(*reinterpret_cast<const uint4*>(res + id * 8), rv);
        const f32x4_t a0 = *reinterpret_cast<const f32x4_t*>(a + g * 8), a1 = *reinterpret_cast<const f32x4_t*>(a + g * 8 + 4);
        const f32x4_t b0 = *reinterpret_cast<const f32x4_t*>(b + g * 8), b1 = *reinterpret_cast<const f32x4_t*>(b + g * 8 + 4);
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            float v = __builtin_fmaf(xv[e], e < 4 ? a0[e] : a1[e - 4], e < 4 ? b0[e] : b1[e - 4]);
            if (res) v += rv[e];
            o[e] = relu ? fmaxf(v, 0.f) : v;
        }
        *reinterpret_cast<uint4*>(y + id * 8) = pack8(o);
    }
}

// dx = k1*dyr + k2*x + k3;  dres = dyr (optional);  dyr = dy * [y > 0] when y != null
__global__ __launch_bounds__(256) void bn_bwd_apply_kernel(const uint16_t* __restrict__ dy, const uint16_t* __restrict__ y,
                                                           const uint16_t* __restrict__ x, const float* __restrict__ k1,
                                                           const float* __restrict__ k2, const float* __restrict__ k3, long P, int C,
                                                           uint16_t* __restrict__ dx, uint16_t* __restrict__ dres) {
    const int cg = C >> 3;
    const long total = P * cg;
    for (long id = (long)blockIdx.x * 256 + threadIdx.x; id < total; id += (long)gridDim.x * 256) {
        const int g = (int)(id % cg);
        float gv[8], xv[8], o[8];
        unpack8(*reinterpret_cast<const uint4*>(dy + id * 8), gv);
        unpack8(*reinterpret_cast<const uint4*>(x + id * 8), xv);
        if (y) {
            float yv[8];
            unpack8(*reinterpret_cast<const uint4*>(y + id * 8), yv);
#pragma unroll
            for (int e = 0; e < 8; ++e) gv[e] = yv[e] > 0.f ? gv[e] : 0.f;
        }
#pragma unroll
        for (int e = 0; e < 8; ++e) o[e] = __builtin_fmaf(k1[g * 8 + e], gv[e], __builtin_fmaf(k2[g * 8 + e], xv[e], k3[g * 8 + e]));
        *reinterpret_cast<uint4*>(dx + id * 8) = pack8(o);
        if (dres) *reinterpret_cast<uint4*>(dres + id * 8) = pack8(gv);
    }
}

int chunking(long P, int C, int* pix_per_chunk) {
    const int rows = 256 / (C >> 3);
    long target = (P + 767) / 768;                        // ~768 workgroups
    long ppc = ((target + rows - 1) / rows) * rows;
    if (ppc < rows * 4) ppc = rows * 4;
    long n = (P + ppc - 1) / ppc;
    while (n > MAX_CHUNKS) { ppc *= 2; n = (P + ppc - 1) / ppc; }
    *pix_per_chunk = (int)ppc;
    return (int)n;
}

unsigned ew_grid(long total) {
    long b = (total + 255) / 256;
    return (unsigned)(b > 8192 ? 8192 : (b ? b : 1));
}

}  // namespace

extern "C" long long w2c_bn_workspace_bytes(long long P, int C) {
    if (P <= 0 || C <= 0 || (C % 8) || C > 2048) return -1;
    return (long long)MAX_CHUNKS * 2 * C * 4;
}

extern "C" int w2c_bn_train_forward(const uint16_t* x, long long P, int C, const float* gamma, const float* beta,
                                    float* running_mean, float* running_var, long long* num_batches_tracked, float momentum, float eps,
                                    const uint16_t* residual, int relu, uint16_t* y,
                                    float* mean, float* rstd, float* ab /* [2][C] scratch */, void* workspace,
                                    long long workspace_bytes, w2c_stream_t stream) {
    w2c_clear_error();
    if (!x || !gamma || !beta || !y || !mean || !rstd || !ab || !workspace || P <= 0 || C <= 0 || (C % 8) || C > 2048) return W2C_E_ARG;
    if (workspace_bytes < w2c_bn_workspace_bytes(P, C)) return W2C_E_ARG;
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    int ppc;
    const int n = chunking(P, C, &ppc);
    float* part = reinterpret_cast<float*>(workspace);
    const size_t lds = (size_t)(256 / (C >> 3)) * 2 * C * 4;
    hipLaunchKernelGGL((bn_reduce_kernel<0>), dim3(n), dim3(256), lds, s, x, nullptr, nullptr, nullptr, nullptr, (long)P, C, ppc, part);
    hipLaunchKernelGGL(bn_fwd_finalize_kernel, dim3((C + 3) / 4), dim3(256), 0, s, part, n, (long)P, C, eps, momentum, gamma, beta,
                       running_mean, running_var, num_batches_tracked, mean, rstd, ab, ab + C);
    hipLaunchKernelGGL(bn_apply_kernel, dim3(ew_grid(P * (C >> 3))), dim3(256), 0, s, x, ab, ab + C, residual, relu, (long)P, C, y);
    return w2c_launch_status();
}

extern "C" int w2c_bn_train_backward(const uint16_t* dy, const uint16_t* y_or_null, const uint16_t* x, long long P, int C,
                                     const float* gamma, const float* mean, const float* rstd,
                                     uint16_t* dx, uint16_t* dres_or_null, float* dgamma, float* dbeta,
                                     float* k123 /* [3][C] scratch */, void* workspace, long long workspace_bytes, w2c_stream_t stream) {
    w2c_clear_error();
    if (!dy || !x || !gamma || !mean || !rstd || !dx || !dgamma || !dbeta || !k123 || !workspace || P <= 0 || C <= 0 || (C % 8) || C > 2048)
        return W2C_E_ARG;
    if (workspace_bytes < w2c_bn_workspace_bytes(P, C)) return W2C_E_ARG;
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    int ppc;
    const int n = chunking(P, C, &ppc);
    float* part = reinterpret_cast<float*>(workspace);
    const size_t lds = (size_t)(256 / (C >> 3)) * 2 * C * 4;
    hipLaunchKernelGGL((bn_reduce_kernel<1>), dim3(n), dim3(256), lds, s, x, dy, y_or_null, mean, rstd, (long)P, C, ppc, part);
    hipLaunchKernelGGL(bn_bwd_finalize_kernel, dim3((C + 3) / 4), dim3(256), 0, s, part, n, (long)P, C, gamma, mean, rstd, dgamma, dbeta,
                       k123, k123 + C, k123 + 2 * C);
    hipLaunchKernelGGL(bn_bwd_apply_kernel, dim3(ew_grid(P * (C >> 3))), dim3(256), 0, s, dy, y_or_null, x, k123, k123 + C, k123 + 2 * C,
                       (long)P, C, dx, dres_or_null);
    return w2c_launch_status();
}

// ---- maxpool 3x3 / stride 2 / pad 1 for the training path (backbone.py:66 via the third-party resnet18's maxpool) ----
// forward: y and, per output element, the tap index 3*ky+kx of the FIRST maximum in scan order (what nn.MaxPool2d's backward
// routes the gradient to); backward: every input element sums dy over the <= 4 windows that selected it (a gather: no atomics).
namespace {

__global__ __launch_bounds__(256) void maxpool_train_fwd_kernel(const uint16_t* __restrict__ x, int M, int H, int W, int C,
                                                                uint16_t* __restrict__ y, uint8_t* __restrict__ idx) {
    const int Ho = H >> 1, Wo = W >> 1, cg = C >> 3;
    const long total = (long)M * Ho * Wo * cg;
    for (long id = (long)blockIdx.x * 256 + threadIdx.x; id < total; id += (long)gridDim.x * 256) {
        const int g = (int)(id % cg);
        long t = id / cg;
        const int ox = (int)(t % Wo); t /= Wo;
        const int oy = (int)(t % Ho);
        const int m = (int)(t / Ho);
        float best[8];
        int bi[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) { best[e] = -INFINITY; bi[e] = 0; }
#pragma unroll
        for (int ky = 0; ky < 3; ++ky)
#pragma unroll
            for (int kx = 0; kx < 3; ++kx) {
                const int iy = 2 * oy - 1 + ky, ix = 2 * ox - 1 + kx;
                if ((unsigned)iy < (unsigned)H && (unsigned)ix < (unsigned)W) {
                    float v[8];
                    unpack8(*reinterpret_cast<const uint4*>(x + (((size_t)m * H + iy) * W + ix) * C + g * 8), v);
#pragma unroll
                    for (int e = 0; e < 8; ++e)
                        if (v[e] > best[e] || (bi[e] == 0 && best[e] == -INFINITY)) { best[e] = v[e]; bi[e] = ky * 3 + kx; }
                }
            }
        *reinterpret_cast<uint4*>(y + id * 8) = pack8(best);
        uint2 o;
        o.x = (uint32_t)bi[0] | ((uint32_t)bi[1] << 8) | ((uint32_t)bi[2] << 16) | ((uint32_t)bi[3] << 24);
        o.y = (uint32_t)bi[4] | ((uint32_t)bi[5] << 8) | ((uint32_t)bi[6] << 16) | ((uint32_t)bi[7] << 24);
        *reinterpret_cast<uint2*>(idx + id * 8) = o;
    }
}

__global__ __launch_bounds__(256) void maxpool_train_bwd_kernel(const uint16_t* __restrict__ dy, const uint8_t* __restrict__ idx, int M, int H,
                                                                int W, int C, uint16_t* __restrict__ dx) {
    const int Ho = H >> 1, Wo = W >> 1, cg = C >> 3;
    const long total = (long)M * H * W * cg;
    for (long id = (long)blockIdx.x * 256 + threadIdx.x; id < total; id += (long)gridDim.x * 256) {
        const int g = (int)(id % cg);
        long t = id / cg;
        const int ix = (int)(t % W); t /= W;
        const int iy = (int)(t % H);
        const int m = (int)(t / H);
        float acc[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) acc[e] = 0.f;
        // windows (oy, ox) with 2oy-1 <= iy <= 2oy+1: oy = iy/2 (even iy) or {iy/2, iy/2+1} (odd iy); same for x
        for (int oy = iy >> 1; oy <= ((iy + 1) >> 1); ++oy) {
            if (oy >= Ho) continue;
            const int ky = iy - (2 * oy - 1);
            for (int ox = ix >> 1; ox <= ((ix + 1) >> 1); ++ox) {
                if (ox >= Wo) continue;
                const int tap = ky * 3 + (ix - (2 * ox - 1));
                const size_t o = ((((size_t)m * Ho + oy) * Wo + ox) * cg + g) * 8;
                const uint2 iv = *reinterpret_cast<const uint2*>(idx + o);
                float gv[8];
                unpack8(*reinterpret_cast<const uint4*>(dy + o), gv);
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    const int sel = (int)(((e < 4 ? iv.x : iv.y) >> (8 * (e & 3))) & 0xFF);
                    acc[e] += sel == tap ? gv[e] : 0.f;
                }
            }
        }
        *reinterpret_cast<uint4*>(dx + id * 8) = pack8(acc);
    }
}

}  // namespace

extern "C" int w2c_maxpool3x3s2_train_forward(const uint16_t* x, int M, int H, int W, int C, uint16_t* y, uint8_t* idx, w2c_stream_t stream) {
    w2c_clear_error();
    if (!x || !y || !idx || M <= 0 || H <= 0 || W <= 0 || (H & 1) || (W & 1) || C <= 0 || (C % 8)) return W2C_E_ARG;
    hipLaunchKernelGGL(maxpool_train_fwd_kernel, dim3(ew_grid((long)M * (H / 2) * (W / 2) * (C / 8))), dim3(256), 0,
                       reinterpret_cast<hipStream_t>(stream), x, M, H, W, C, y, idx);
    return w2c_launch_status();
}

extern "C" int w2c_maxpool3x3s2_train_backward(const uint16_t* dy, const uint8_t* idx, int M, int H, int W, int C, uint16_t* dx,
                                               w2c_stream_t stream) {
    w2c_clear_error();
    if (!dy || !dx || !idx || M <= 0 || H <= 0 || W <= 0 || (H & 1) || (W & 1) || C <= 0 || (C % 8)) return W2C_E_ARG;
    hipLaunchKernelGGL(maxpool_train_bwd_kernel, dim3(ew_grid((long)M * H * W * (C / 8))), dim3(256), 0,
                       reinterpret_cast<hipStream_t>(stream), dy, idx, M, H, W, C, dx);
    return w2c_launch_status();
}

// ---- two-phase (cross-rank) forms: see bn_sums_kernel.  mode 0: sums = (sum x, sum x^2); mode 1: (sum dyr, sum dyr * xhat)
extern "C" int w2c_bn_train_sums(int mode, const uint16_t* x, const uint16_t* dy, const uint16_t* y_or_null, const float* mean,
                                 const float* rstd, long long P, int C, double* sums, void* workspace, long long workspace_bytes,
                                 w2c_stream_t stream) {
    w2c_clear_error();
    if (!x || !sums || !workspace || P <= 0 || C <= 0 || (C % 8) || C > 2048 || (mode != 0 && mode != 1)) return W2C_E_ARG;
    if (mode == 1 && (!dy || !mean || !rstd)) return W2C_E_ARG;
    if (workspace_bytes < w2c_bn_workspace_bytes(P, C)) return W2C_E_ARG;
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    int ppc;
    const int n = chunking(P, C, &ppc);
    float* part = reinterpret_cast<float*>(workspace);
    const size_t lds = (size_t)(256 / (C >> 3)) * 2 * C * 4;
    if (mode == 0) hipLaunchKernelGGL((bn_reduce_kernel<0>), dim3(n), dim3(256), lds, s, x, nullptr, nullptr, nullptr, nullptr, (long)P, C, ppc, part);
    else hipLaunchKernelGGL((bn_reduce_kernel<1>), dim3(n), dim3(256), lds, s, x, dy, y_or_null, mean, rstd, (long)P, C, ppc, part);
    hipLaunchKernelGGL(bn_sums_kernel, dim3((C + 3) / 4), dim3(256), 0, s, part, n, C, sums);
    return w2c_launch_status();
}

extern "C" int w2c_bn_train_forward_sums(const uint16_t* x, long long P, int C, const double* sums_global, double P_total,
                                         const float* gamma, const float* beta, float* running_mean, float* running_var,
                                         long long* num_batches_tracked, float momentum, float eps, const uint16_t* residual, int relu,
                                         uint16_t* y, float* mean, float* rstd, float* ab, w2c_stream_t stream) {
    w2c_clear_error();
    if (!x || !sums_global || !gamma || !beta || !y || !mean || !rstd || !ab || P <= 0 || P_total < (double)P || C <= 0 || (C % 8) || C > 2048)
        return W2C_E_ARG;
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    hipLaunchKernelGGL(bn_fwd_finalize_sums_kernel, dim3((C + 255) / 256), dim3(256), 0, s, sums_global, P_total, C, eps, momentum, gamma, beta,
                       running_mean, running_var, num_batches_tracked, mean, rstd, ab, ab + C);
    hipLaunchKernelGGL(bn_apply_kernel, dim3(ew_grid(P * (C >> 3))), dim3(256), 0, s, x, ab, ab + C, residual, relu, (long)P, C, y);
    return w2c_launch_status();
}

extern "C" int w2c_bn_train_backward_sums(const uint16_t* dy, const uint16_t* y_or_null, const uint16_t* x, long long P, int C,
                                          const float* gamma, const float* mean, const float* rstd, const double* sums_local,
                                          const double* sums_global, double P_total, uint16_t* dx, uint16_t* dres_or_null,
                                          float* dgamma, float* dbeta, float* k123, w2c_stream_t stream) {
    w2c_clear_error();
    if (!dy || !x || !gamma || !mean || !rstd || !sums_local || !sums_global || !dx || !dgamma || !dbeta || !k123 || P <= 0 ||
        P_total < (double)P || C <= 0 || (C % 8) || C > 2048)
        return W2C_E_ARG;
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    hipLaunchKernelGGL(bn_bwd_finalize_sums_kernel, dim3((C + 255) / 256), dim3(256), 0, s, sums_local, sums_global, P_total, C, gamma, mean, rstd,
                       dgamma, dbeta, k123, k123 + C, k123 + 2 * C);
    hipLaunchKernelGGL(bn_bwd_apply_kernel, dim3(ew_grid(P * (C >> 3))), dim3(256), 0, s, dy, y_or_null, x, k123, k123 + C, k123 + 2 * C,
                       (long)P, C, dx, dres_or_null);
    return w2c_launch_status();
}
