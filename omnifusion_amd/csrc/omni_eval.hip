// omni_eval.hip — on-device evaluation metrics (the first "next" row):
// median scaling + the seven depth metrics of the reference's eval loop, without a device->host round trip
// per batch.   Replaces /root/reference/test.py:151-176 (compute_eval_metrics) and /root/reference/metrics.py:7-26.
//
//   omni_masked_median_f32   x[mask>0].median()  — exact: 4-pass radix select on the order-preserving key of the float
//                            bits (torch.median returns the LOWER middle element for an even count: rank (n-1)/2)
//   omni_depth_metrics_f32   pred *= scale (in place, as test.py:162 does), then the masked sums of
//                            |p-g|/g, (p-g)^2/g, (p-g)^2, (log p - log g)^2 [own mask p,g > 1e-7], delta < 1.25^{1,2,3}
// Reductions are two-stage (per-block partials in double, then one block in fixed order): deterministic.
#include "omni_internal.h"

namespace {

__device__ __forceinline__ unsigned fkey(float v)
{
    const unsigned b = __float_as_uint(v);
    return (b & 0x80000000u) ? ~b : (b | 0x80000000u);      // ascending unsigned order == ascending float order
}

// state[0] = prefix key bits decided so far, state[1] = remaining rank within the prefix bucket, state[2] = count
__global__ __launch_bounds__(256) void median_hist_kernel(const float* __restrict__ x, const float* __restrict__ mask, size_t n,
                                                          int pass, const unsigned* __restrict__ state, unsigned* __restrict__ hist)
{
    __shared__ unsigned h[256];
    h[threadIdx.x] = 0;
    __syncthreads();
    const int shift = 24 - 8 * pass;
    const unsigned prefix = state[0];
    const unsigned pmask = pass == 0 ? 0u : (0xffffffffu << (shift + 8));
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
        if (mask[i] > 0.0f) {
            const unsigned k = fkey(x[i]);
            if ((k & pmask) == prefix) atomicAdd(&h[(k >> shift) & 255u], 1u);
        }
    }
    __syncthreads();
    if (h[threadIdx.x]) atomicAdd(&hist[threadIdx.x], h[threadIdx.x]);
}

__global__ void median_pick_kernel(int pass, unsigned* __restrict__ state, unsigned* __restrict__ hist, float* __restrict__ out)
{
    if (threadIdx.x != 0) return;
    unsigned rank = state[1];
    if (pass == 0) {                                         // total count -> rank of the lower median
        unsigned cnt = 0;
        for (int b = 0; b < 256; ++b) cnt += hist[b];
        state[2] = cnt;
        rank = cnt ? (cnt - 1) / 2 : 0;
    }
    unsigned acc = 0; int b = 0;
    for (; b < 256; ++b) { if (acc + hist[b] > rank) break; acc += hist[b]; }
    if (b == 256) b = 255;
    const int shift = 24 - 8 * pass;
    state[0] |= (unsigned)b << shift;
    state[1] = rank - acc;
    for (int q = 0; q < 256; ++q) hist[q] = 0;
    if (pass == 3) {
        const unsigned k = state[0];
        const unsigned bits = (k & 0x80000000u) ? (k & 0x7fffffffu) : ~k;
        *out = state[2] ? __uint_as_float(bits) : __uint_as_float(0x7fc00000u);     // empty selection -> NaN (like torch)
    }
}

constexpr int NMET = 9;     // abs_rel, sq_rel, rms_sq_lin, rms_sq_log, d1, d2, d3, N, N_log
__global__ __launch_bounds__(256) void metrics_partial_kernel(float* __restrict__ pred, const float* __restrict__ gt,
                                                              const float* __restrict__ mask, const float* __restrict__ scale_num,
                                                              const float* __restrict__ scale_den, size_t n, double* __restrict__ part)
{
    __shared__ double red[4][NMET];
    const float sc = (scale_num && scale_den) ? (*scale_num / *scale_den) : 1.0f;
    double s[NMET];
#pragma unroll
    for (int k = 0; k < NMET; ++k) s[k] = 0.0;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
        const float p = pred[i] * sc;
        pred[i] = p;                                         // test.py:162 scales the prediction in place
        const float g = gt[i];
        if (mask[i] > 0.0f) {
            const float d = p - g;
            s[0] += (double)(fabsf(d) / g); s[1] += (double)((d * d) / g); s[2] += (double)(d * d);
            const float r = fmaxf(p / g, g / p);
            s[4] += r < 1.25f ? 1.0 : 0.0; s[5] += r < 1.5625f ? 1.0 : 0.0; s[6] += r < 1.953125f ? 1.0 : 0.0;
            s[7] += 1.0;
            if (p > 1e-7f && g > 1e-7f) { const float l = logf(p) - logf(g); s[3] += (double)(l * l); s[8] += 1.0; }
        }
    }
#pragma unroll
    for (int k = 0; k < NMET; ++k)
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) s[k] += __shfl_xor(s[k], o);
    const int wave = threadIdx.x >> 6;
    if ((threadIdx.x & 63) == 0)
#pragma unroll
        for (int k = 0; k < NMET; ++k) red[wave][k] = s[k];
    __syncthreads();
    if (threadIdx.x < NMET) part[(size_t)blockIdx.x * NMET + threadIdx.x] = (red[0][threadIdx.x] + red[1][threadIdx.x]) + (red[2][threadIdx.x] + red[3][threadIdx.x]);
}
__global__ void metrics_final_kernel(const double* __restrict__ part, int nblocks, float* __restrict__ out)
{
    if (threadIdx.x >= NMET) return;
    double s = 0.0;
    for (int b = 0; b < nblocks; ++b) s += part[(size_t)b * NMET + threadIdx.x];
    __shared__ double tot[NMET];
    tot[threadIdx.x] = s;
    __syncthreads();
    const double N = tot[7], NL = tot[8];
    float v;
    if (threadIdx.x == 3) v = (float)(NL > 0 ? tot[3] / NL : __longlong_as_double(0x7ff8000000000000ll));
    else if (threadIdx.x >= 7) v = (float)tot[threadIdx.x];
    else v = (float)(N > 0 ? tot[threadIdx.x] / N : __longlong_as_double(0x7ff8000000000000ll));
    out[threadIdx.x] = v;
}
}  // namespace

extern "C" {
// ws: at least 260 unsigned (state[4] + hist[256]).  out: device float.
int omni_masked_median_f32(const float* x, const float* mask, size_t n, unsigned* ws, float* out, omni_stream_t stream)
{
    if (!x || !mask || !ws || !out) OMNI_FAIL(OMNI_ERR_INVALID, "omni_masked_median: null pointer");
    hipStream_t s = (hipStream_t)stream;
    OMNI_HIP(hipMemsetAsync(ws, 0, 260 * sizeof(unsigned), s));
    const int blocks = (int)((n + 255) / 256 < 2048 ? (n + 255) / 256 : 2048);
    for (int pass = 0; pass < 4; ++pass) {
        if (blocks > 0) hipLaunchKernelGGL(median_hist_kernel, dim3(blocks), dim3(256), 0, s, x, mask, n, pass, (const unsigned*)ws, ws + 4);
        hipLaunchKernelGGL(median_pick_kernel, dim3(1), dim3(64), 0, s, pass, ws, ws + 4, out);
    }
    OMNI_HIP(hipGetLastError());
    return OMNI_OK;
}
// pred is scaled in place by *scale_num / *scale_den (device scalars; both NULL: no scaling).  ws: 2048*9 doubles.
// out[9] (device): abs_rel, sq_rel, rms_sq_lin, rms_sq_log, d1, d2, d3, N, N_log
int omni_depth_metrics_f32(float* pred, const float* gt, const float* mask, const float* scale_num, const float* scale_den,
                           size_t n, double* ws, float* out, omni_stream_t stream)
{
    if (!pred || !gt || !mask || !ws || !out) OMNI_FAIL(OMNI_ERR_INVALID, "omni_depth_metrics: null pointer");
    hipStream_t s = (hipStream_t)stream;
    int blocks = (int)((n + 255) / 256 < 2048 ? (n + 255) / 256 : 2048);
    if (blocks < 1) blocks = 1;
    hipLaunchKernelGGL(metrics_partial_kernel, dim3(blocks), dim3(256), 0, s, pred, gt, mask, scale_num, scale_den, n, ws);
    hipLaunchKernelGGL(metrics_final_kernel, dim3(1), dim3(64), 0, s, (const double*)ws, blocks, out);
    OMNI_HIP(hipGetLastError());
    return OMNI_OK;
}
}
