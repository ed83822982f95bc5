// Issue rate of packed fp32 arithmetic on gfx950 (plain forms, inline assembly): does v_pk_mul_f32 / v_pk_fma_f32 cost one VALU slot (two results per
// lane and slot) or two?  16 independent chains per thread, 4096 trips; 1 / 2 / 4 waves per SIMD.  Stand-alone: hipcc --offload-arch=gfx950 -O3 tools/pk_rate.hip -o /tmp/pk_rate
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f2v __attribute__((ext_vector_type(2)));
template <int MODE> __global__ __launch_bounds__(256) void rate_kernel(float* out, float s, int trips)
{
    f2v a[8]; float b[16];
    for (int i = 0; i < 8; ++i) { a[i] = f2v{(float)threadIdx.x + i, 1.0f + i}; }
    for (int i = 0; i < 16; ++i) b[i] = (float)threadIdx.x + i;
    const f2v s2 = {s, s};
    for (int t = 0; t < trips; ++t) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            if (MODE == 0) {                                      // 16 scalar multiplies
#pragma unroll
                for (int i = 0; i < 16; ++i) asm volatile("v_mul_f32 %0, %1, %0" : "+v"(b[i]) : "v"(s));
            } else if (MODE == 1) {                               // 8 packed multiplies = the same 16 products
#pragma unroll
                for (int i = 0; i < 8; ++i) asm volatile("v_pk_mul_f32 %0, %1, %0" : "+v"(a[i]) : "v"(s2));
            } else if (MODE == 2) {                               // 16 scalar fma
#pragma unroll
                for (int i = 0; i < 16; ++i) asm volatile("v_fma_f32 %0, %1, %0, %0" : "+v"(b[i]) : "v"(s));
            } else {                                              // 8 packed fma
#pragma unroll
                for (int i = 0; i < 8; ++i) asm volatile("v_pk_fma_f32 %0, %1, %0, %0" : "+v"(a[i]) : "v"(s2));
            }
        }
    }
    float acc = 0.0f;
    for (int i = 0; i < 8; ++i) acc += a[i].x + a[i].y;
    for (int i = 0; i < 16; ++i) acc += b[i];
    if (acc == 12345.678f) out[threadIdx.x] = acc;
}
template <int MODE> static void run(const char* name, float* out, int blocks)
{
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int trips = 4096;
    hipLaunchKernelGGL(rate_kernel<MODE>, dim3(blocks), dim3(256), 0, 0, out, 1.0000001f, trips);
    hipEventRecord(e0);
    for (int i = 0; i < 5; ++i) hipLaunchKernelGGL(rate_kernel<MODE>, dim3(blocks), dim3(256), 0, 0, out, 1.0000001f, trips);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1); ms /= 5;
    const double products = (double)blocks * 256 * trips * 4 * 16;
    printf("%-22s blocks %5d (%d wave(s) per SIMD): %8.1f us  %7.2f G results/s  = %.2f results per lane-slot at 2.4 GHz x 256 CU x 64 lanes\n", name, blocks, blocks / 256, ms * 1e3,
           products / ms * 1e-6, products / (ms * 1e-3) / (2.4e9 * 256 * 64));
}
// dependent chains: the same number of v_mul_f32 per thread spread over NCH independent chains (1 = every instruction waits for the one before it)
template <int NCH> __global__ __launch_bounds__(256) void chain_kernel(float* out, float s, int trips)
{
    float b[NCH];
    for (int i = 0; i < NCH; ++i) b[i] = (float)threadIdx.x + i;
    for (int t = 0; t < trips; ++t) {
#pragma unroll
        for (int r = 0; r < 64 / NCH; ++r)
#pragma unroll
            for (int i = 0; i < NCH; ++i) asm volatile("v_mul_f32 %0, %1, %0" : "+v"(b[i]) : "v"(s));
    }
    float acc = 0.0f;
    for (int i = 0; i < NCH; ++i) acc += b[i];
    if (acc == 12345.678f) out[threadIdx.x] = acc;
}
template <int NCH> static void run_chain(float* out, int blocks)
{
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int trips = 4096;
    hipLaunchKernelGGL(chain_kernel<NCH>, dim3(blocks), dim3(256), 0, 0, out, 1.0000001f, trips);
    hipEventRecord(e0);
    for (int i = 0; i < 5; ++i) hipLaunchKernelGGL(chain_kernel<NCH>, dim3(blocks), dim3(256), 0, 0, out, 1.0000001f, trips);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1); ms /= 5;
    printf("v_mul_f32, %2d chain(s) per thread, %d wave(s) per SIMD: %8.1f us = %.2f ns per instruction and wave\n", NCH, blocks / 256, ms * 1e3, ms * 1e6 / (trips * 64.0));
}
int main()
{
    float* out; hipMalloc(&out, 1 << 20);
    for (int blocks : {256, 512, 1024}) {
        run<0>("v_mul_f32 x16", out, blocks); run<1>("v_pk_mul_f32 x8", out, blocks);
        run<2>("v_fma_f32 x16", out, blocks); run<3>("v_pk_fma_f32 x8", out, blocks);
    }
    for (int blocks : {256, 512}) { run_chain<1>(out, blocks); run_chain<2>(out, blocks); run_chain<4>(out, blocks); run_chain<16>(out, blocks); }
    return 0;
}
