// Stand-alone reproducer for DESIGN 5b cause 2: v_pk_mul_f32 / v_pk_add_f32 in one kernel returning wrong values WHILE another stream issues
// dense v_mfma_f32_32x32x16_f16 (three per product block, as the f16x3 convolutions do).  hipcc --offload-arch=gfx950 -O3 tools/pkfp32_repro.hip -o /tmp/pk && /tmp/pk
// Prints the fraction of victim launches with at least one wrong element, with and without the matrix kernel beside them.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef float f16v __attribute__((ext_vector_type(16)));
typedef float f2 __attribute__((ext_vector_type(2)));
__global__ __launch_bounds__(256) void mfma_kernel(float* sink, int iters, int per_block)   // per_block MFMAs per product block: 3 (f16x3) or 1
{
    h8 a, b; for (int i = 0; i < 8; ++i) { a[i] = (_Float16)(0.001f * (threadIdx.x + i)); b[i] = (_Float16)(0.002f * (threadIdx.x ^ i)); }
    f16v acc0 = {}, acc1 = {};
    for (int it = 0; it < iters; ++it) {
        acc1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc1, 0, 0, 0);
        if (per_block == 3) { acc0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(b, a, acc0, 0, 0, 0); acc1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, a, acc1, 0, 0, 0); }
    }
    sink[blockIdx.x * 256 + threadIdx.x] = acc0[0] + acc1[3];
}
__global__ __launch_bounds__(256) void victim_kernel(const f2* __restrict__ x, const f2* __restrict__ y, f2* __restrict__ out, int n, int rounds)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    f2 a = x[i], b = y[i], r = {0.f, 0.f};
    for (int k = 0; k < rounds; ++k) {                            // packed products and sums, as hipcc's SLP vectoriser emits them for bilinear weights
        f2 p; asm volatile("v_pk_mul_f32 %0, %1, %2" : "=v"(p) : "v"(a), "v"(b));
        asm volatile("v_pk_add_f32 %0, %1, %2" : "=v"(r) : "v"(r), "v"(p));
        a[0] += 1.0f; a[1] -= 1.0f;
    }
    out[i] = r;
}
int main()
{
    const int n = 1 << 20, rounds = 64, launches = 400;
    std::vector<f2> hx(n), hy(n), want(n), got(n);
    for (int i = 0; i < n; ++i) { hx[i] = {1.0f + (i % 977) * 0.25f, 3.0f - (i % 311) * 0.5f}; hy[i] = {0.5f + (i % 13), 2.0f - (i % 7)}; }
    for (int i = 0; i < n; ++i) { f2 a = hx[i], r = {0, 0}; for (int k = 0; k < rounds; ++k) { r[0] += a[0] * hy[i][0]; r[1] += a[1] * hy[i][1]; a[0] += 1.0f; a[1] -= 1.0f; } want[i] = r; }
    f2 *dx, *dy, *dout; float* sink;
    hipMalloc(&dx, n * 8); hipMalloc(&dy, n * 8); hipMalloc(&dout, n * 8); hipMalloc(&sink, 2048 * 256 * 4);
    hipMemcpy(dx, hx.data(), n * 8, hipMemcpyHostToDevice); hipMemcpy(dy, hy.data(), n * 8, hipMemcpyHostToDevice);
    hipStream_t sa, sb; hipStreamCreateWithFlags(&sa, hipStreamNonBlocking); hipStreamCreateWithFlags(&sb, hipStreamNonBlocking);
    for (int mode = 0; mode < 3; ++mode) {                         // 0: victim alone | 1: beside 3 MFMAs per block | 2: beside 1 MFMA per block
        int bad_launches = 0; long long bad_elems = 0;
        for (int l = 0; l < launches; ++l) {
            if (mode) hipLaunchKernelGGL(mfma_kernel, dim3(2048), dim3(256), 0, sa, sink, 4000, mode == 1 ? 3 : 1);
            hipLaunchKernelGGL(victim_kernel, dim3(n / 256), dim3(256), 0, sb, dx, dy, dout, n, rounds);
            hipStreamSynchronize(sb);
            hipMemcpy(got.data(), dout, n * 8, hipMemcpyDeviceToHost);
            long long bad = 0;
            for (int i = 0; i < n; ++i) bad += (got[i][0] != want[i][0]) || (got[i][1] != want[i][1]);
            bad_launches += bad != 0; bad_elems += bad;
            if (mode) hipStreamSynchronize(sa);
        }
        printf("%-44s: %d of %d victim launches wrong, %lld wrong elements\n", mode == 0 ? "packed fp32 alone" : mode == 1 ? "beside v_mfma x3 per block on another stream" : "beside v_mfma x1 per block on another stream", bad_launches, launches, bad_elems);
    }
    return 0;
}
