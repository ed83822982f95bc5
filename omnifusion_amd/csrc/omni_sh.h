// omni_sh.h — the split-half ("SH") activation layout shared by the network kernels (device side).
//
// A tensor [..., C] (C % 32 == 0) stores, per group of 32 channels, 32 hi halfs followed by 32 lo halfs (128 bytes = the
// footprint of 32 floats):  x = hi + lo * 2^-11,  hi = fp16(x) (0 below the fp16 normal range 2^-14),
// lo = fp16((x - hi) * 2^11) — 22-23 significant bits, and exactly the operand format of the f16x3 matrix products.
#pragma once
#include <hip/hip_runtime.h>

typedef float omni_f4v __attribute__((ext_vector_type(4)));
typedef _Float16 omni_h4v __attribute__((ext_vector_type(4)));

// Range guard: fp16(x) is inf for |x| > 65504 and the lo half would then be NaN, so values are SATURATED to the
// fp16 range before the split (a trained checkpoint with activation outliers degrades instead of turning into NaN) and a sticky
// per-device flag records that it happened — `omni_sh_overflow()` reads and clears it, the Python model exposes it as
// `spherical_fusion.overflowed()`.  One copy per translation unit (the library is built without relocatable device code).
static __device__ unsigned sh_overflow_flag;
constexpr float SH_MAX = 65504.0f;

__device__ __forceinline__ void sh_split4(const omni_f4v x, omni_h4v& hi, omni_h4v& lo)
{
    const float m = fmaxf(fmaxf(fabsf(x[0]), fabsf(x[1])), fmaxf(fabsf(x[2]), fabsf(x[3])));
    if (__builtin_expect(m > SH_MAX, 0)) atomicOr(&sh_overflow_flag, 1u);           // NaN compares false: propagates like in fp32
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        const float xs = fabsf(x[e]) > SH_MAX ? copysignf(SH_MAX, x[e]) : x[e];     // NaN compares false and stays NaN
        const _Float16 h = (fabsf(xs) < 6.103515625e-05f) ? (_Float16)0.0f : (_Float16)xs;
        hi[e] = h; lo[e] = (_Float16)((xs - (float)h) * 2048.0f);
    }
}
// host side of the flag, one definition per translation unit that includes this header and wants to report it
#define OMNI_SH_OVERFLOW_ACCESSOR(fn)                                                                          \
    int fn(unsigned* out, int reset) {                                                                         \
        unsigned v = 0;                                                                                        \
        if (hipMemcpyFromSymbol(&v, HIP_SYMBOL(sh_overflow_flag), sizeof(v)) != hipSuccess) return -1;         \
        if (reset && v) { const unsigned z = 0; if (hipMemcpyToSymbol(HIP_SYMBOL(sh_overflow_flag), &z, sizeof(z)) != hipSuccess) return -1; } \
        *out |= v; return 0; }
__device__ __forceinline__ omni_f4v sh_join4(const omni_h4v hi, const omni_h4v lo)
{
    omni_f4v x;
#pragma unroll
    for (int e = 0; e < 4; ++e) x[e] = fmaf((float)lo[e], 4.8828125e-4f, (float)hi[e]);
    return x;
}
// byte offset of the hi halfs of channels [c, c+4); e = flat element index (pixel * C + c), c % 4 == 0, C % 32 == 0
__device__ __forceinline__ size_t sh_off(size_t e) { return (e & ~(size_t)31) * 4 + (e & 31) * 2; }

// four consecutive channels at flat element index e of an fp32 (SH = false) or SH (true) tensor
template <bool SH> __device__ __forceinline__ omni_f4v act_load4(const void* base, size_t e)
{
    if (SH) {
        const unsigned char* p = (const unsigned char*)base + sh_off(e);
        return sh_join4(*reinterpret_cast<const omni_h4v*>(p), *reinterpret_cast<const omni_h4v*>(p + 64));
    }
    return *reinterpret_cast<const omni_f4v*>((const float*)base + e);
}
template <bool SH> __device__ __forceinline__ void act_store4(void* base, size_t e, const omni_f4v v)
{
    if (SH) {
        omni_h4v hi, lo; sh_split4(v, hi, lo);
        unsigned char* p = (unsigned char*)base + sh_off(e);
        *reinterpret_cast<omni_h4v*>(p) = hi; *reinterpret_cast<omni_h4v*>(p + 64) = lo;
    } else {
        *reinterpret_cast<omni_f4v*>((float*)base + e) = v;
    }
}
