// Investigation only: as taps_dump.hip, but shaped like p2e_kernel — a wave-uniform loop over the candidate patches, TWO per trip, their constants
// indexed out of a by-value table (scalar loads inside the loop), so that the SLP vectoriser pairs the two patches' arithmetic (SGPR-pair operands).
#include <hip/hip_runtime.h>
#define NPATCH 18
struct TapArgs2 { float kx, ky, half_h, half_w, fw, fh; int H, W; unsigned mask; float sp[NPATCH], cp[NPATCH], sl0[NPATCH], cl0[NPATCH]; };
#define NF 16
struct T2 { float cd, sd, cos_c, rc, nx, ny, X, Y, wa, wb, wc, wd, q0, q1, q2; };
__device__ __forceinline__ void taps(const TapArgs2& a, int n, float slat, float clat, float slon, float clon, T2& t)
{
    const float sl0 = a.sl0[n], cl0 = a.cl0[n], sp = a.sp[n], cp = a.cp[n];
    t.cd = clon * cl0 + slon * sl0;
    t.sd = slon * cl0 - clon * sl0;
    t.q0 = sp * slat; t.q1 = cp * clat; t.q2 = t.q1 * t.cd;
    t.cos_c = t.q0 + t.q2;
    float rc = __builtin_amdgcn_rcpf(t.cos_c);
    rc = fmaf(fmaf(-t.cos_c, rc, 1.0f), rc, rc);
    t.rc = rc;
    float nx = (clat * t.sd) * rc;
    float ny = (cp * slat - sp * clat * t.cd) * rc;
    nx = nx * a.kx; ny = ny * a.ky;
    t.nx = nx; t.ny = ny;
    t.X = (nx + 1.0f) * a.half_h;
    t.Y = (ny + 1.0f) * a.half_w;
    const bool valid = (t.X < a.fw) && (t.X > 0.0f) && (t.Y < a.fh) && (t.Y > 0.0f) && (t.cos_c > 0.0f);
    const float fx = floorf(t.X), fy = floorf(t.Y);
    const float x1f = fminf(fx + 1.0f, a.fw - 1.0f), y1f = fminf(fy + 1.0f, a.fh - 1.0f);
    float wa = (x1f - t.X) * (y1f - t.Y), wb = (x1f - t.X) * (t.Y - fy), wc = (t.X - fx) * (y1f - t.Y), wd = (t.X - fx) * (t.Y - fy);
    t.wa = valid ? wa : 0.0f; t.wb = valid ? wb : 0.0f; t.wc = valid ? wc : 0.0f; t.wd = valid ? wd : 0.0f;
}
__global__ __launch_bounds__(256) void taps_dump2_kernel(TapArgs2 a, const float2* __restrict__ row_trig, const float2* __restrict__ col_trig, float* __restrict__ out)
{
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int tiles = a.W / 64;
    const int i = __builtin_amdgcn_readfirstlane((int)(blockIdx.x / tiles) * 4 + wave), j = (blockIdx.x % tiles) * 64 + lane;
    if (i >= a.H) return;
    const float2 rt = row_trig[i], ct = col_trig[j];
    unsigned m = (unsigned)__builtin_amdgcn_readfirstlane((int)a.mask);
    float acc = 0.0f, l1 = 0.0f;
    while (m) {
        const int n0 = __builtin_ctz(m); m &= m - 1;
        const bool two = m != 0u;
        const int n1 = two ? __builtin_ctz(m) : n0;
        if (two) m &= m - 1;
        T2 t[2];
        taps(a, n0, rt.x, rt.y, ct.x, ct.y, t[0]);
        taps(a, n1, rt.x, rt.y, ct.x, ct.y, t[1]);
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            if (u == 1 && !two) break;
            const int n = u ? n1 : n0;
            float* o = out + (((size_t)i * a.W + j) * NPATCH + n) * NF;
            o[0] = t[u].cd; o[1] = t[u].sd; o[2] = t[u].cos_c; o[3] = t[u].rc; o[4] = t[u].nx; o[5] = t[u].ny; o[6] = t[u].X; o[7] = t[u].Y;
            o[8] = t[u].wa; o[9] = t[u].wb; o[10] = t[u].wc; o[11] = t[u].wd;
            const float ws = (t[u].wa + t[u].wb) + (t[u].wc + t[u].wd);
            l1 += ws; acc = fmaf(t[u].X, t[u].wd, fmaf(t[u].Y, t[u].wc, fmaf(t[u].nx, t[u].wb, fmaf(t[u].ny, t[u].wa, acc))));
            o[12] = t[u].q0; o[13] = t[u].q1; o[14] = t[u].q2; o[15] = acc;
        }
    }
}
extern "C" int taps_dump2(TapArgs2 a, const void* row_trig, const void* col_trig, void* out, void* stream)
{
    hipLaunchKernelGGL(taps_dump2_kernel, dim3((a.H / 4) * (a.W / 64)), dim3(256), 0, (hipStream_t)stream, a, (const float2*)row_trig, (const float2*)col_trig, (float*)out);
    return (int)hipGetLastError();
}
