// Investigation: which packed instructions / operand selections return wrong results beside an MFMA kernel.  Each variant executes ONE
// packed instruction per loop trip on lane-dependent operands and checks both halves against scalar arithmetic in the same lane; mismatches are
// counted per lane.  Run by tools/pk/pk_probe.py beside a convolution of the product library.
#include <hip/hip_runtime.h>
typedef float f2 __attribute__((ext_vector_type(2)));
typedef _Float16 h2 __attribute__((ext_vector_type(2)));
#define PROBE32(NAME, ASM, EXP_LO, EXP_HI)                                                                               \
__global__ __launch_bounds__(256) void NAME(unsigned* __restrict__ bad, int iters)                                      \
{                                                                                                                       \
    const int lane = threadIdx.x & 63; unsigned nb = 0;                                                                 \
    f2 a = {1.0f + 0.001f * (float)(threadIdx.x + blockIdx.x), 2.0f + 0.002f * (float)lane};                           \
    f2 b = {0.5f + 0.003f * (float)lane, 3.0f - 0.001f * (float)(blockIdx.x & 255)};                                   \
    f2 c = {0.25f, -0.75f};                                                                                             \
    for (int i = 0; i < iters; ++i) {                                                                                   \
        f2 r;                                                                                                           \
        asm volatile(ASM : "=v"(r) : "v"(a), "v"(b), "v"(c));                                                           \
        const float lo = EXP_LO, hi = EXP_HI;                                                                           \
        nb += (r.x != lo) + (r.y != hi);                                                                                \
        a.x += 0.0625f; b.y -= 0.03125f;                                                                                \
        asm volatile("" : "+v"(a), "+v"(b));                                                                            \
    }                                                                                                                   \
    if (nb) atomicAdd(bad + lane, nb);                                                                                  \
}
// (no FMA contraction: the file is compiled with -ffp-contract=off)
PROBE32(mul_plain,      "v_pk_mul_f32 %0, %1, %2",                                   a.x * b.x, a.y * b.y)
PROBE32(mul_src1_cross, "v_pk_mul_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,0]",      a.x * b.y, a.y * b.x)
PROBE32(mul_src0_cross, "v_pk_mul_f32 %0, %1, %2 op_sel:[1,0] op_sel_hi:[0,1]",      a.y * b.x, a.x * b.y)
PROBE32(mul_src1_hihi,  "v_pk_mul_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,1]",      a.x * b.y, a.y * b.y)
PROBE32(mul_src1_lolo,  "v_pk_mul_f32 %0, %1, %2 op_sel:[0,0] op_sel_hi:[1,0]",      a.x * b.x, a.y * b.x)
PROBE32(add_src1_cross, "v_pk_add_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,0]",      a.x + b.y, a.y + b.x)
PROBE32(add_src0_cross, "v_pk_add_f32 %0, %1, %2 op_sel:[1,0] op_sel_hi:[0,1]",      a.y + b.x, a.x + b.y)
PROBE32(fma_plain,      "v_pk_fma_f32 %0, %1, %2, %3",                               __builtin_fmaf(a.x, b.x, c.x), __builtin_fmaf(a.y, b.y, c.y))
PROBE32(fma_src1_cross, "v_pk_fma_f32 %0, %1, %2, %3 op_sel:[0,1,0] op_sel_hi:[1,0,1]", __builtin_fmaf(a.x, b.y, c.x), __builtin_fmaf(a.y, b.x, c.y))
PROBE32(fma_src2_cross, "v_pk_fma_f32 %0, %1, %2, %3 op_sel:[0,0,1] op_sel_hi:[1,1,0]", __builtin_fmaf(a.x, b.x, c.y), __builtin_fmaf(a.y, b.y, c.x))
PROBE32(mov_cross,      "v_pk_mov_b32 %0, %1, %2 op_sel:[1,0]",                      a.y, b.x)
#define PROBE16(NAME, ASM, EXP_LO, EXP_HI)                                                                              \
__global__ __launch_bounds__(256) void NAME(unsigned* __restrict__ bad, int iters)                                      \
{                                                                                                                       \
    const int lane = threadIdx.x & 63; unsigned nb = 0;                                                                 \
    h2 a = {(_Float16)(1.0f + 0.01f * (float)lane), (_Float16)(2.0f + 0.02f * (float)(threadIdx.x >> 6))};            \
    h2 b = {(_Float16)(0.5f + 0.03f * (float)lane), (_Float16)(3.0f - 0.01f * (float)(blockIdx.x & 63))};             \
    for (int i = 0; i < iters; ++i) {                                                                                   \
        h2 r;                                                                                                           \
        asm volatile(ASM : "=v"(r) : "v"(a), "v"(b));                                                                   \
        const _Float16 lo = EXP_LO, hi = EXP_HI;                                                                        \
        nb += (r.x != lo) + (r.y != hi);                                                                                \
        a.x += (_Float16)0.0625f; b.y -= (_Float16)0.03125f;                                                            \
        asm volatile("" : "+v"(a), "+v"(b));                                                                            \
    }                                                                                                                   \
    if (nb) atomicAdd(bad + lane, nb);                                                                                  \
}
PROBE16(mul16_src1_cross, "v_pk_mul_f16 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,0]",    a.x * b.y, a.y * b.x)
PROBE16(add16_src1_cross, "v_pk_add_f16 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,0]",    a.x + b.y, a.y + b.x)

#define ENTRY(NAME) if (which == n++) { if (name) *name = #NAME; if (bad) hipLaunchKernelGGL(NAME, dim3(blocks), dim3(256), 0, (hipStream_t)stream, bad, iters); return 0; }
extern "C" int pk_probe(int which, unsigned* bad, int blocks, int iters, void* stream, const char** name)
{
    int n = 0;
    ENTRY(mul_plain) ENTRY(mul_src1_cross) ENTRY(mul_src0_cross) ENTRY(mul_src1_hihi) ENTRY(mul_src1_lolo) ENTRY(add_src1_cross) ENTRY(add_src0_cross)
    ENTRY(fma_plain) ENTRY(fma_src1_cross) ENTRY(fma_src2_cross) ENTRY(mov_cross) ENTRY(mul16_src1_cross) ENTRY(add16_src1_cross)
    return -1;
}
