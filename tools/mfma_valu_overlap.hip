// Do matrix instructions and vector arithmetic of DIFFERENT waves on one SIMD overlap on gfx950?  256 blocks x 8 waves (two per SIMD): waves 0-3 issue 32x32x16 fp16
// matrix instructions (four independent accumulators), waves 4-7 issue v_fma_f32 (16 independent chains).  Times: matrix waves alone, vector waves alone, both.
// overlap -> both ~ max; one shared issue / execution resource -> both ~ sum.  Stand-alone: hipcc --offload-arch=gfx950 -O3 tools/mfma_valu_overlap.hip -o /tmp/mvo
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f16v __attribute__((ext_vector_type(16)));
typedef _Float16 h8v __attribute__((ext_vector_type(8)));
template <int OP> __global__ __launch_bounds__(512) void overlap_kernel(float* out, int mode, int mfma_trips, int valu_trips, float s)
{
    const int wave = threadIdx.x >> 6;
    if (wave < 4) {
        if (!(mode & 1)) return;
        f16v acc[4]; for (int i = 0; i < 4; ++i) acc[i] = (f16v)(0.0f);
        h8v a = (h8v)((_Float16)(float)(threadIdx.x & 3)), b = (h8v)((_Float16)s);
        for (int t = 0; t < mfma_trips; ++t) {
#pragma unroll
            for (int r = 0; r < 2; ++r)
#pragma unroll
                for (int i = 0; i < 4; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc[i], 0, 0, 0);
        }
        float x = 0.0f; for (int i = 0; i < 4; ++i) x += acc[i][0] + acc[i][7];
        if (x == 12345.678f) out[threadIdx.x] = x;
    } else {
        if (!(mode & 2)) return;
        float b[16]; for (int i = 0; i < 16; ++i) b[i] = (float)threadIdx.x + i;
        for (int t = 0; t < valu_trips; ++t) {
#pragma unroll
            for (int r = 0; r < 4; ++r)
#pragma unroll
                for (int i = 0; i < 16; ++i) {
                    if (OP == 1)      asm volatile("v_mul_f32 %0, %1, %0" : "+v"(b[i]) : "v"(s));
                    else if (OP == 2) asm volatile("v_mul_lo_u32 %0, %1, %0" : "+v"(b[i]) : "v"(s));          // (quarter-rate integer multiply: address arithmetic)
                    else if (OP == 3) asm volatile("v_cvt_f16_f32 %0, %0" : "+v"(b[i]));
                    else              asm volatile("v_fma_f32 %0, %1, %0, %0" : "+v"(b[i]) : "v"(s));
                }
        }
        float x = 0.0f; for (int i = 0; i < 16; ++i) x += b[i];
        if (x == 12345.678f) out[threadIdx.x] = x;
    }
}
template <int OP> static float run(float* out, int mode, int mt, int vt)
{
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(overlap_kernel<OP>, dim3(256), dim3(512), 0, 0, out, mode, mt, vt, 1.0000001f);
    hipEventRecord(e0);
    for (int i = 0; i < 5; ++i) hipLaunchKernelGGL(overlap_kernel<OP>, dim3(256), dim3(512), 0, 0, out, mode, mt, vt, 1.0000001f);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1); return ms / 5 * 1e3f;
}
template <int OP> static void sweep(float* out, const char* name)
{
    const int mt = 2000;                       // 16 000 matrix instructions per matrix wave
    for (int vt : {2000 / (OP == 2 ? 4 : 1), 1000 / (OP == 2 ? 4 : 1), 500 / (OP == 2 ? 4 : 1), 250 / (OP == 2 ? 4 : 1)}) {    // 128 000 ... 16 000 vector instructions per vector wave (a quarter of that for the quarter-rate multiply)
        const float tm = run<OP>(out, 1, mt, vt), tv = run<OP>(out, 2, mt, vt), tb = run<OP>(out, 3, mt, vt);
        printf("%s x %6d per vector wave: matrix waves alone %7.1f us (%.1f ns per instruction and SIMD) | vector waves alone %7.1f us (%.2f ns per instruction) | both %7.1f us = %.2f x max, %.2f x sum\n",
               name, vt * 64, tm, tm * 1e3 / (mt * 8), tv, tv * 1e3 / (vt * 64), tb, tb / (tm > tv ? tm : tv), tb / (tm + tv));
    }
}
int main()
{
    float* out; hipMalloc(&out, 1 << 20);
    sweep<0>(out, "v_fma_f32"); sweep<1>(out, "v_mul_f32"); sweep<2>(out, "v_mul_lo_u32"); sweep<3>(out, "v_cvt_f16_f32");
    return 0;
}
