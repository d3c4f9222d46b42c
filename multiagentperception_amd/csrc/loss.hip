// cross_entropy2d (reference ptsemseg/loss/loss.py:5-18 = F.cross_entropy over the NCHW logits with
// ignore_index 250, optional class weights, mean over the kept pixels) forward and backward on gfx950.
//
// HBM-bound: the forward reads every logit once (online max / sum-of-exponentials, one pass for any class count) and
// writes 4 B per pixel (the log-sum-exp, kept for the backward); the backward reads logits + lse and writes the gradient
// -- 2 * 4 * C bytes per pixel, against the 5 passes (log_softmax, nll_loss, their two backwards, the NHWC transpose
// copy of loss.py:13) of the stock path.  A lane owns a pixel, so for a fixed class the 64 lanes of a wave touch 64
// consecutive floats of the NCHW plane: every access is a coalesced 256-B row.
// Deterministic: per-block partial sums (f64) are combined by one block in index order.
#include "w2c_common.h"

namespace {

constexpr int CE_THREADS = 256;

__device__ __forceinline__ double wave_sum_f64(double v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, 64);
    return v;
}

// partials: [nblocks][3] doubles = (sum of w_t * loss, sum of w_t, number of out-of-range targets)
__global__ __launch_bounds__(CE_THREADS) void ce_forward_kernel(const float* __restrict__ logits, const long long* __restrict__ target,
                                                                 const float* __restrict__ weight, int C, long long HW, long long P,
                                                                 int ignore_index, float* __restrict__ lse, float* __restrict__ loss_px,
                                                                 double* __restrict__ partials) {
    double s_loss = 0.0, s_w = 0.0, s_bad = 0.0;
    for (long long i = (long long)blockIdx.x * CE_THREADS + threadIdx.x; i < P; i += (long long)gridDim.x * CE_THREADS) {
        const long long n = i / HW, p = i - n * HW;
        const float* x = logits + n * C * HW + p;
        const long long t = target[i];
        float m = -INFINITY, s = 0.f, xt = 0.f;
        for (int c = 0; c < C; ++c) {
            const float v = x[(long long)c * HW];
            if (c == t) xt = v;
            if (v > m) { s = s * __expf(m - v) + 1.f; m = v; }      // (s = 0 at the first class: 0 * exp(-inf - v) = 0)
            else s += __expf(v - m);
        }
        const float l = m + __logf(s);
        lse[i] = l;
        float li = 0.f;
        if (t != ignore_index) {
            if (t >= 0 && t < C) {
                const float w = weight ? weight[t] : 1.f;
                li = w * (l - xt);
                s_loss += (double)li;
                s_w += (double)w;
            } else s_bad += 1.0;
        }
        if (loss_px) loss_px[i] = li;
    }
    __shared__ double red[3][CE_THREADS / 64];
    s_loss = wave_sum_f64(s_loss); s_w = wave_sum_f64(s_w); s_bad = wave_sum_f64(s_bad);
    const int wv = threadIdx.x >> 6;
    if ((threadIdx.x & 63) == 0) { red[0][wv] = s_loss; red[1][wv] = s_w; red[2][wv] = s_bad; }
    __syncthreads();
    if (threadIdx.x < 3) {
        double a = 0.0;
        for (int k = 0; k < CE_THREADS / 64; ++k) a += red[threadIdx.x][k];
        partials[(long long)blockIdx.x * 3 + threadIdx.x] = a;
    }
}

// out[0] = loss (mean over the kept weight, or the sum), out[1] = the denominator (sum of kept weights), out[2] = number of
// targets outside [0, C) that are not ignore_index (PyTorch device-asserts on those; here they are dropped and counted)
__global__ __launch_bounds__(64) void ce_finalize_kernel(const double* __restrict__ partials, int nblocks, int size_average,
                                                          float* __restrict__ out) {
    double a[3] = {0.0, 0.0, 0.0};
    for (int b = threadIdx.x; b < nblocks; b += 64) {
        a[0] += partials[3ll * b]; a[1] += partials[3ll * b + 1]; a[2] += partials[3ll * b + 2];
    }
    // lane l holds blocks l, l+64, ...: a fixed tree, independent of timing
#pragma unroll
    for (int k = 0; k < 3; ++k) a[k] = wave_sum_f64(a[k]);
    if (threadIdx.x == 0) {
        out[0] = (float)(size_average ? a[0] / a[1] : a[0]);
        out[1] = (float)a[1];
        out[2] = (float)a[2];
    }
}

// d logits[n,c,p] = g * w_t * (softmax_c - [c == t]),  g = gout (scalar on the device, optional) * gpx[i] (per pixel, optional)
//                   / denom (device scalar, when the forward averaged)
__global__ __launch_bounds__(CE_THREADS) void ce_backward_kernel(const float* __restrict__ logits, const long long* __restrict__ target,
                                                                  const float* __restrict__ weight, const float* __restrict__ lse,
                                                                  int C, long long HW, long long P, int ignore_index,
                                                                  const float* __restrict__ denom, const float* __restrict__ gout,
                                                                  const float* __restrict__ gpx, float* __restrict__ dlogits) {
    float g0 = gout ? gout[0] : 1.f;
    if (denom) g0 /= denom[0];
    for (long long i = (long long)blockIdx.x * CE_THREADS + threadIdx.x; i < P; i += (long long)gridDim.x * CE_THREADS) {
        const long long n = i / HW, p = i - n * HW;
        const float* x = logits + n * C * HW + p;
        float* d = dlogits + n * C * HW + p;
        const long long t = target[i];
        const bool keep = t != ignore_index && t >= 0 && t < C;
        float g = 0.f;
        if (keep) {
            g = g0 * (weight ? weight[t] : 1.f);
            if (gpx) g *= gpx[i];
        }
        const float l = lse[i];
        for (int c = 0; c < C; ++c) {
            float v = 0.f;
            if (keep) v = g * (__expf(x[(long long)c * HW] - l) - (c == t ? 1.f : 0.f));
            d[(long long)c * HW] = v;
        }
    }
}

int ce_blocks(long long P) {
    long long b = (P + CE_THREADS - 1) / CE_THREADS;
    if (b > 4096) b = 4096;              // 16 workgroups per CU: enough loads in flight for HBM, few partials to combine
    if (b < 1) b = 1;
    return (int)b;
}

}  // namespace

extern "C" long long w2c_cross_entropy2d_workspace_bytes(long long n_pixels) { return (long long)ce_blocks(n_pixels) * 3 * sizeof(double); }

extern "C" int w2c_cross_entropy2d_forward(const float* logits, const long long* target, const float* weight, int N, int C, long long HW,
                                           int ignore_index, int size_average, float* lse, float* loss_px, float* out3,
                                           void* workspace, long long workspace_bytes, void* stream) {
    if (!logits || !target || !lse || !out3 || !workspace) return W2C_E_ARG;
    if (N <= 0 || C <= 0 || HW <= 0) return W2C_E_ARG;
    const long long P = (long long)N * HW;
    if (workspace_bytes < w2c_cross_entropy2d_workspace_bytes(P)) return W2C_E_ARG;
    const int nb = ce_blocks(P);
    w2c_clear_error();
    hipLaunchKernelGGL(ce_forward_kernel, dim3(nb), dim3(CE_THREADS), 0, (hipStream_t)stream, (const float*)logits,
                       (const long long*)target, (const float*)weight, C, HW, P, ignore_index, (float*)lse, (float*)loss_px,
                       (double*)workspace);
    hipLaunchKernelGGL(ce_finalize_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, (const double*)workspace, nb, size_average,
                       (float*)out3);
    return w2c_launch_status();
}

extern "C" int w2c_cross_entropy2d_backward(const float* logits, const long long* target, const float* weight, const float* lse, int N,
                                            int C, long long HW, int ignore_index, const float* denom, const float* gout,
                                            const float* gpx, float* dlogits, void* stream) {
    if (!logits || !target || !lse || !dlogits) return W2C_E_ARG;
    if (N <= 0 || C <= 0 || HW <= 0) return W2C_E_ARG;
    const long long P = (long long)N * HW;
    w2c_clear_error();
    hipLaunchKernelGGL(ce_backward_kernel, dim3(ce_blocks(P) * 4), dim3(CE_THREADS), 0, (hipStream_t)stream, (const float*)logits,
                       (const long long*)target, (const float*)weight, (const float*)lse, C, HW, P, ignore_index,
                       (const float*)denom, (const float*)gout, (const float*)gpx, (float*)dlogits);
    return w2c_launch_status();
}
