// omni_equi2pers.hip — ERP -> N tangent-plane patches (gfx950).
//
// Replaces /root/reference/equi_pers/equi2pers_v3.py:20-122:
//   K1  CPU grid build + repeat(bs) + H2D (:24-109)        -> evaluated in-kernel, closed form
//   K2  F.grid_sample(bilinear, border, align_corners=True) (:111) -> gather below
//   K3  F.unfold + reshape to [B,C,ph,pw,N] (:112-113)      -> fused into the store (LDS transpose)
//   K4  uv2xyz rays (:13-18,115-118), uv (:120-121)         -> omni_equi2pers_aux
//   K5  dead erp_mask work (:49-74)                         -> dropped
//
// Geometry.  The reference evaluates (fp32)  rho=sqrt(x^2+y^2), c=atan(rho),
//   lat = asin(cos c sin p1 + y sin c cos p1 / rho),
//   lon = l0 + atan2(x sin c, rho cos p1 cos c - y sin p1 sin c)             (:95-100)
// With sin c = rho/sqrt(1+rho^2), cos c = 1/sqrt(1+rho^2) this is identically
//   lat = asin((sin p1 + y cos p1) / sqrt(1+x^2+y^2)),   lon = l0 + atan2(x, cos p1 - y sin p1)
// which needs two transcendental calls instead of six (the reference's 0/0 at an exact patch
// centre, quirk q4, is reproduced explicitly in e2p_lonlat) and differs from the reference's fp32 chain by coordinate round-off only (a few 1e-4 px
// away from the poles; tests/ carry the tolerance).
//
// Memory behaviour: one thread owns one (or four consecutive) sample position(s) and loops
// over all B*C image planes, so geometry is evaluated once and amortised; ERP reads are
// 4-byte gathers with wave-level locality (a wave walks a short curve on the ERP), patch
// writes are fully coalesced.  HBM-bound: algorithmic bytes B*C*(H*W + ph*pw*N)*sizeof(T).
#include <stdio.h>
#include <stdlib.h>
#include <type_traits>
#include <vector>
#include <algorithm>
#include <utility>
#include "omni_internal.h"
#include "omni_spgather.h"

namespace {

struct E2PArgs {
    const void* erp; void* pers;
    int B, C, H, W, ph, pw;
    float fovx, fovy;          // fov_w/360, fov_h/180  (equi2pers_v3.py:24)
    float stepx, stepy;        // linspace(0,1,P) step (:29)
    float sx_scale, sy_scale;  // (W-1)/2, (H-1)/2  (grid_sample align_corners=True)
    int dbg;                   // tuning hook (OMNI_E2P_DBG): 1 = suppress stores, 2 = suppress box loads
    const float2* ixy;         // per-geometry table of clamped sampling coordinates [N][ph][pw] (e2p_lds_kernel), or null
    long long* trace;          // debug build, OMNI_E2P_DBG bit 16: per-block time stamps (omni_debug_set_trace)
    int store_mode;            // option e2p_store: 0 plain | 1 non-temporal (default)
    int dbg_skip_fb;           // timing experiment only (option e2p_ref_lds = 2)
    PatchTab tab;
};

struct Tap {                    // one bilinear footprint on the ERP, branch-free to fetch
    int r0, r1;                 // element offsets of the two tap rows (see e2p_tap)
    int sel;                    // PAIR: 1 when the 2-wide load was shifted left by one (x0 == W-1)
                                // !PAIR: column step dx (0 when x0+1 is outside)
    float w00, w01, w10, w11;   // ATen's nw, ne, sw, se weights
};

constexpr float PI_F = 3.14159265358979323846f;
constexpr float PI_2_F = 1.57079632679489661923f;

__device__ __forceinline__ float lin01(int idx, int steps, float step)
{
    // torch.linspace(0, 1, steps)[idx] in fp32 (two-sided), equi2pers_v3.py:29
    // (ATen evaluates the upper half as ONE fma: linspace(0,1,15)[7] = 0.49999997, not 0.5)
    return (idx < (steps >> 1)) ? step * (float)idx : fmaf(-step, (float)(steps - 1 - idx), 1.0f);
}

// inverse gnomonic for sample (h, w) of patch n -> unwrapped lon, lat and the pieces xyz needs
__device__ __forceinline__ void e2p_lonlat(const E2PArgs& a, int n, int h, int w,
                                           float& lon, float& lat, float& x, float& q, float& t, float& inv)
{
    const float sw = lin01(w, a.pw, a.stepx), sh = lin01(h, a.ph, a.stepy);
    x = ((sw * 2.0f - 1.0f) * PI_F) * a.fovx;                 // :86-89
    const float y = ((sh * 2.0f - 1.0f) * PI_2_F) * a.fovy;
    const float sp = a.tab.sphi[n], cp = a.tab.cphi[n];
    q = cp - y * sp;
    t = sp + y * cp;
    inv = 1.0f / sqrtf(1.0f + x * x + y * y);
    float sl = t * inv;
    sl = fminf(1.0f, fmaxf(-1.0f, sl));
    lat = asinf(sl);
    lon = a.tab.lam0[n] + atan2f(x, q);
    // Reference quirk q4: at x == y == 0 (the centre sample when BOTH patch dims are odd and their
    // linspace midpoints are exactly 0.5) the reference divides 0/0 at :99 -> lat = NaN while
    // lon = l0 + atan2(0, 0) = l0.  ATen then clips the NaN row coordinate to 0, so that sample reads
    // the top ERP row, and xyz is NaN.  Reproduced, not fixed: it defines parity.
    if (x == 0.0f && y == 0.0f) { lat = __builtin_nanf(""); t = lat; }
}

__device__ __forceinline__ void e2p_uv(float lon, float lat, float& u, float& v)
{
    v = lat / PI_2_F;                                          // :101
    u = lon / PI_F;                                            // :102
    if (u > 1.0f) u -= 2.0f;                                   // :103
    if (u < -1.0f) u += 2.0f;                                  // :104
}

// Footprint of sample (h, w) of patch n.  ATen's grid_sampler skips taps that fall outside the
// image; a clipped coordinate is integral there, so such a tap also has weight exactly 0.  The
// outside tap is therefore ALIASED onto the in-range pixel of the same row/column (never onto a
// pixel ATen would not have read), which keeps every load unconditional and in bounds.
template <bool PAIR>
__device__ __forceinline__ Tap e2p_tap(const E2PArgs& a, int n, int h, int w)
{
    float lon, lat, x, q, t, inv, u, v;
    e2p_lonlat(a, n, h, w, lon, lat, x, q, t, inv);
    e2p_uv(lon, lat, u, v);
    // ATen grid_sampler: unnormalise (align_corners) then clip (border)
    float ix = (u + 1.0f) * a.sx_scale, iy = (v + 1.0f) * a.sy_scale;
    ix = fminf((float)(a.W - 1), fmaxf(ix, 0.0f));
    iy = fminf((float)(a.H - 1), fmaxf(iy, 0.0f));
    const float fx = floorf(ix), fy = floorf(iy);
    const int x0 = (int)fx, y0 = (int)fy;
    const float tx = ix - fx, ty = iy - fy, ex = 1.0f - tx, ey = 1.0f - ty;
    Tap p;
    p.w00 = ey * ex; p.w01 = ey * tx; p.w10 = ty * ex; p.w11 = ty * tx;
    const int y1 = min(y0 + 1, a.H - 1);
    if (PAIR) {                                   // 8-byte loads of (xb, xb+1), xb = min(x0, W-2)
        const int xb = min(x0, a.W - 2);
        p.sel = x0 - xb;
        p.r0 = y0 * a.W + xb; p.r1 = y1 * a.W + xb;
    } else {
        p.sel = (x0 + 1 < a.W) ? 1 : 0;
        p.r0 = y0 * a.W + x0; p.r1 = y1 * a.W + x0;
    }
    return p;
}

template <typename T> struct Pair;
template <> struct Pair<float> {
    struct __attribute__((packed, aligned(4))) U { float x, y; };     // 4-byte aligned 8-byte load
    static __device__ __forceinline__ void ld(const float* p, float& x, float& y)
    { const U v = *reinterpret_cast<const U*>(p); x = v.x; y = v.y; }
};
template <> struct Pair<__half> {
    static __device__ __forceinline__ void ld(const __half* p, float& x, float& y)
    { unsigned u; __builtin_memcpy(&u, p, 4); const __half2 h = *reinterpret_cast<const __half2*>(&u);
      x = __low2float(h); y = __high2float(h); }
};

// ATen's bilinear sum nw*w00 + ne*w01 + sw*w10 + se*w11, associated COLUMN-wise — (v00 w00 + v10 w10) + (v01 w01 + v11 w11) — so that a tap
// pair (x0, x0+1) read as one 8-byte value goes through two packed operations (v_pk_mul_f32, v_pk_fma_f32) and one add without
// any register shuffling.  EVERY equi2pers kernel (gather, LDS box, reference layout, fallback) uses this one function: same bits.
__device__ __forceinline__ float e2p_blend(float v00, float v01, float v10, float v11, float w00, float w01, float w10, float w11)
{
    return fmaf(v10, w10, v00 * w00) + fmaf(v11, w11, v01 * w01);
}

template <typename T, bool PAIR>
__device__ __forceinline__ float e2p_fetch(const T* __restrict__ img, const Tap& p)
{
    float v00, v01, v10, v11;
    if (PAIR) {
        float ax, ay, bx, by;
        Pair<T>::ld(img + p.r0, ax, ay);
        Pair<T>::ld(img + p.r1, bx, by);
        v00 = p.sel ? ay : ax; v01 = ay; v10 = p.sel ? by : bx; v11 = by;
    } else {
        v00 = Store<T>::ld(img + p.r0); v01 = Store<T>::ld(img + p.r0 + p.sel);
        v10 = Store<T>::ld(img + p.r1); v11 = Store<T>::ld(img + p.r1 + p.sel);
    }
    return e2p_blend(v00, v01, v10, v11, p.w00, p.w01, p.w10, p.w11);
}

// ------------------------------------------------------------------ planar output [B,N,C,ph,pw]
// A wave owns 256 consecutive elements of patch n's flattened (h,w) plane; lane l owns elements
// l, l+64, l+128, l+192 of them, so every load instruction covers 64 CONSECUTIVE samples (a short
// run of the ERP: 3-4 cache lines per tap row) and every store instruction writes one contiguous
// 256-byte run.  Geometry is evaluated once per sample and amortised over all B*C image planes.
template <typename T, bool PAIR, int SPT, int UNR>
__global__ __launch_bounds__(256) void e2p_planar_kernel(E2PArgs a, int blocks_per_patch, int nblocks)
{
    const unsigned lb = omni_xcd_remap(blockIdx.x, nblocks);
    const int n = lb / blocks_per_patch;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int e0 = ((lb % blocks_per_patch) * 4 + wave) * (64 * SPT) + lane;
    const int plane = a.ph * a.pw;
    if (e0 >= plane) return;
    Tap tp[SPT];
#pragma unroll
    for (int k = 0; k < SPT; ++k) {
        const int e = min(e0 + 64 * k, plane - 1);
        tp[k] = e2p_tap<PAIR>(a, n, e / a.pw, e % a.pw);
    }
    const T* erp = (const T*)a.erp;
    T* out = (T*)a.pers + (size_t)n * a.C * plane + e0;
    const size_t img_plane = (size_t)a.H * a.W;
    const size_t out_bstride = (size_t)a.tab.N * a.C * plane;
    const int planes = a.B * a.C;
    // UNR image planes per trip: all their gathers are issued before the first result is consumed
    for (int p0 = 0; p0 < planes; p0 += UNR) {
        float r[UNR][SPT];
#pragma unroll
        for (int u = 0; u < UNR; ++u) {
            const int p = min(p0 + u, planes - 1);
            const T* img = erp + (size_t)p * img_plane;
#pragma unroll
            for (int k = 0; k < SPT; ++k) r[u][k] = e2p_fetch<T, PAIR>(img, tp[k]);
        }
#pragma unroll
        for (int u = 0; u < UNR; ++u) {
            const int p = p0 + u;
            if (p < planes) {
                const int b = p / a.C, c = p - b * a.C;
                T* dst = out + (size_t)b * out_bstride + (size_t)c * plane;
#pragma unroll
                for (int k = 0; k < SPT; ++k)
                    if (e0 + 64 * k < plane) Store<T>::st(dst + 64 * k, r[u][k]);
            }
        }
    }
}

// ------------------------------------------------------------------ planar output, LDS-staged ERP footprint
// The gather kernel above is bound by the vector L1: a 64-lane gather costs ~27 tag look-ups for ~0.5 KB of
// useful data (profiles/r01a_resample_pmc.txt).  Here a block owns a 32x32 sample tile of one patch, finds the
// bounding box of the tile's bilinear footprint on the ERP (block reduction; columns measured relative to the
// tile's first sample so that a tile straddling the +-pi seam still has a narrow box), streams that box into
// LDS with fully coalesced 16-byte loads (64 useful bytes per L1 access) and takes the four taps of every
// sample from LDS (ds_read2_b32).  The box of plane p+1 is in flight in registers while plane p is computed
// (double-buffered LDS, one barrier per plane).  Tiles whose box does not fit (the pole itself lies inside, or
// the ERP row pitch is not a multiple of 4) fall back to the direct gathers — wave-uniform branch, same taps.
constexpr int E2P_BOXF = 3968;                    // floats per LDS buffer: 2 buffers + 80 B < 32 KiB -> 5 blocks / CU

__device__ __forceinline__ int wave_min(int v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = min(v, __shfl_xor(v, o));
    return v;
}
__device__ __forceinline__ int wave_max(int v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = max(v, __shfl_xor(v, o));
    return v;
}


// clamped sampling coordinates of patch sample (n, h, w): the closed-form geometry (two transcendentals per sample) followed
// by grid_sample's align_corners=True scaling and border clamp (equi2pers_v3.py:95-104,111)
__device__ __forceinline__ void e2p_sample_xy(const E2PArgs& a, int n, int h, int w, float& ix, float& iy)
{
    float lon, lat, x, q, tt, inv, u, v;
    e2p_lonlat(a, n, h, w, lon, lat, x, q, tt, inv);
    e2p_uv(lon, lat, u, v);
    ix = (u + 1.0f) * a.sx_scale; iy = (v + 1.0f) * a.sy_scale;
    ix = fminf((float)(a.W - 1), fmaxf(ix, 0.0f));
    iy = fminf((float)(a.H - 1), fmaxf(iy, 0.0f));
}
__global__ __launch_bounds__(256) void e2p_ixy_kernel(E2PArgs a, float2* __restrict__ tab, int total)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= total) return;
    const int w = i % a.pw, h = (i / a.pw) % a.ph, n = i / (a.pw * a.ph);
    float ix, iy;
    e2p_sample_xy(a, n, h, w, ix, iy);
    tab[i] = make_float2(ix, iy);
}

// Grid: blocks [0, ntiles) own one (patch, tile) each and run the LDS path; a tile that does not fit returns at once
// and is covered by blocks [ntiles, ntiles + nfb*B): one block per (listed tile, batch item), direct gathers, so the
// few pole tiles are spread over B times more blocks instead of serialising B*C planes in one straggler.
// flags_out != nullptr: geometry-setup mode, only records which tiles need the gather path.
// BWD: the transposed operator — a.pers holds g_pers (read), a.erp g_erp (zeroed by the host, accumulated here): every tile
// accumulates its footprint box in LDS (ds_add_f32) and flushes it with coalesced global atomics, 16 bytes per lane
template <int TS, bool BWD = false>               // tile side in samples: 32 (4 samples per thread) or 16 (1)
__global__ __launch_bounds__(256) void e2p_lds_kernel(E2PArgs a, int tiles_x, int tiles_per_patch, int ntiles,
                                                      const int* __restrict__ fb, unsigned char* flags_out)
{
    // ONE __shared__ object: with a second one hipcc waits vmcnt(0) before every ds_read while an LDS-DMA is in flight
    __shared__ __attribute__((aligned(16))) float lds_all[2 * E2P_BOXF + 20];
    float (*box)[E2P_BOXF] = reinterpret_cast<float (*)[E2P_BOXF]>(lds_all);
    int (*red)[4] = reinterpret_cast<int (*)[4]>(lds_all + 2 * E2P_BOXF);
    int& sh_xc = *reinterpret_cast<int*>(lds_all + 2 * E2P_BOXF + 16);
    const bool fb_block = (int)blockIdx.x >= ntiles;
    int fb_b = 0;
    unsigned lb;
    if (fb_block) { const int idx = blockIdx.x - ntiles; lb = fb[idx / a.B]; fb_b = idx % a.B; }
    else lb = omni_xcd_remap(blockIdx.x, ntiles);
    const int n = lb / tiles_per_patch;
    const int tile = lb % tiles_per_patch;
    constexpr int SPT = TS * TS / 256, RSTEP = 256 / TS;        // samples per thread, row step between them
    const int th0 = (tile / tiles_x) * TS, tw0 = (tile % tiles_x) * TS;
    const int t = threadIdx.x, wave = t >> 6;
    const int col = t % TS, rowb = t / TS;
    const int W = a.W, H = a.H;

    // ---- taps of this thread's 4 samples (rows rowb + 8k of the tile, column col)
    int x0[SPT], y0[SPT], y1[SPT], s1[SPT];
    float w00[SPT], w01[SPT], w10[SPT], w11[SPT];
    const int w = min(tw0 + col, a.pw - 1);
#pragma unroll
    for (int k = 0; k < SPT; ++k) {
        const int h = min(th0 + rowb + RSTEP * k, a.ph - 1);
        float ix, iy;
        if (a.ixy) {                                     // configuration constant: evaluated once per geometry handle by the
            const float2 c = a.ixy[((size_t)n * a.ph + h) * a.pw + w];   // same device function (bit-identical), 8 bytes per sample
            ix = c.x; iy = c.y;
        } else {
            e2p_sample_xy(a, n, h, w, ix, iy);
        }
        const float fx = floorf(ix), fy = floorf(iy);
        x0[k] = (int)fx; y0[k] = (int)fy;
        const float tx = ix - fx, ty = iy - fy, ex = 1.0f - tx, ey = 1.0f - ty;
        w00[k] = ey * ex; w01[k] = ey * tx; w10[k] = ty * ex; w11[k] = ty * tx;
        y1[k] = min(y0[k] + 1, H - 1);
        s1[k] = (x0[k] + 1 < W) ? 1 : 0;          // +1 column outside: alias onto x0 (its weight is exactly 0)
    }
    // ---- footprint box: rows [ymin, ymax], columns relative to the tile's first sample (seam-safe)
    if (t == 0) sh_xc = x0[0];
    __syncthreads();
    const int xc = sh_xc, half = W >> 1;
    int dx[SPT];
    int ymin = y0[0], ymax = y1[0];
#pragma unroll
    for (int k = 0; k < SPT; ++k) {
        int d = x0[k] - xc;
        if (d >= half) d -= W;
        if (d < -half) d += W;
        dx[k] = d;
        ymin = min(ymin, y0[k]); ymax = max(ymax, y1[k]);
    }
    int dmin = dx[0], dmax = dx[0];
#pragma unroll
    for (int k = 1; k < SPT; ++k) { dmin = min(dmin, dx[k]); dmax = max(dmax, dx[k]); }
    ymin = wave_min(ymin); ymax = wave_max(ymax); dmin = wave_min(dmin); dmax = wave_max(dmax);
    if ((t & 63) == 0) { red[wave][0] = ymin; red[wave][1] = ymax; red[wave][2] = dmin; red[wave][3] = dmax; }
    __syncthreads();
    ymin = min(min(red[0][0], red[1][0]), min(red[2][0], red[3][0]));
    ymax = max(max(red[0][1], red[1][1]), max(red[2][1], red[3][1]));
    dmin = min(min(red[0][2], red[1][2]), min(red[2][2], red[3][2]));
    dmax = max(max(red[0][3], red[1][3]), max(red[2][3], red[3][3]));
    int xs = xc + dmin;                             // absolute first column of the box (may wrap)
    if (xs < 0) xs += W;
    if (xs >= W) xs -= W;
    const int xs4 = xs & ~3, shift = xs - xs4;
    int bw = (dmax - dmin + 2 + shift + 3) & ~3;             // columns x0..x0+1 of every sample, whole 16-byte chunks
    if (((bw >> 2) & 1) == 0) bw += 4;                       // odd number of 16-byte chunks per row: consecutive box rows start
                                                             // 4, 12, 20, 28 banks apart (polar patches walk the box by rows)
    const int bh = ymax - ymin + 1;
    const int bw4 = bw >> 2, nchunk = bh * bw4;
    const bool fits = ((W & 3) == 0) && (bw <= W) && (bh * bw <= E2P_BOXF);
    const bool full = (th0 + TS <= a.ph) && (tw0 + TS <= a.pw);

    const float* erp = (const float*)a.erp;
    const int plane = a.ph * a.pw;
    const size_t img_plane = (size_t)H * W;
    const size_t out_bstride = (size_t)a.tab.N * a.C * plane;
    // this thread's 4 output elements: e0 + 8k rows
    float* out = (float*)a.pers + (size_t)n * a.C * plane + (size_t)(th0 + rowb) * a.pw + (tw0 + col);
    const int ostep = RSTEP * a.pw;

    if (flags_out) { if (t == 0) flags_out[lb] = (fits && full) ? 0 : 1; return; }
    if (!fb_block && !(fits && full)) return;          // covered by the fallback blocks of this launch
    if (OMNI_DBG(a, 4) && fb_block) return;
    if (OMNI_DBG(a, 8) && !fb_block) return;
    if (fb_block && blockIdx.y > 0) return;            // the fallback blocks walk every plane themselves
    if (!fb_block) {
        int r0[SPT], r1[SPT];
#pragma unroll
        for (int k = 0; k < SPT; ++k) {
            const int c0 = dx[k] - dmin + shift;
            r0[k] = (y0[k] - ymin) * bw + c0;
            r1[k] = (y1[k] - ymin) * bw + c0;
        }
        // Plane loop for a box of NJ x 256 16-byte chunks at most (NJ is block-uniform).  Threads past the
        // last chunk re-load / re-store the last chunk (identical data, same address): no exec masking.
        // Box fill by LDS-DMA (global_load_lds_dwordx4): a wave's 64 lanes deposit 64 consecutive 16-byte chunks
        // straight into the box (the chunk order IS the LDS order), no VGPR staging and no ds_write issue slots.
        // The DMA of plane p+1 is in flight behind the gathers of plane p; one barrier per plane.  The loop is kept
        // free of per-lane conditions: the scalar unit is shared by the whole CU and ~100 scalar instructions per wave
        // and plane (exec-mask juggling, 64-bit pointer updates) were costing as much as the gathers themselves.
        typedef const __attribute__((address_space(1))) void* gptr_t;
        typedef __attribute__((address_space(3))) void* lptr_t;
        const int lane = t & 63;
        const int nj = (nchunk + 255) >> 8;                          // block-uniform number of chunk columns (1..4)
        int goff[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int qc = min(wave * 64 + 256 * j + lane, nchunk - 1);   // lanes past the end re-fetch the last chunk ...
            const int r = qc / bw4, cx = qc - r * bw4;
            int gx = xs4 + 4 * cx;
            if (gx >= W) gx -= W;
            goff[j] = (ymin + r) * W + gx;
        }
        // ... into their own (unused) slot, which must still lie inside the buffer: slots = E2P_BOXF/4 = 992 < 1024
        const bool tail_ok = (wave * 64 + 256 * 3 + lane) < E2P_BOXF / 4;
        auto dma = [&](const float* img, float* buf) {
            __builtin_amdgcn_global_load_lds((gptr_t)(img + goff[0]), (lptr_t)(buf + (wave * 64) * 4), 16, 0, 0);
            if (nj > 1) __builtin_amdgcn_global_load_lds((gptr_t)(img + goff[1]), (lptr_t)(buf + (wave * 64 + 256) * 4), 16, 0, 0);
            if (nj > 2) __builtin_amdgcn_global_load_lds((gptr_t)(img + goff[2]), (lptr_t)(buf + (wave * 64 + 512) * 4), 16, 0, 0);
            if (nj > 3 && tail_ok) __builtin_amdgcn_global_load_lds((gptr_t)(img + goff[3]), (lptr_t)(buf + (wave * 64 + 768) * 4), 16, 0, 0);
        };
        float* const box0 = &box[0][0];
        if (BWD) {
            // ---- transposed trip per plane: zero my chunks | barrier | 4 x 4 ds_add_f32 | barrier | flush my chunks (global atomics)
            float* gerp = (float*)a.erp;
            const float* src = (const float*)a.pers + (size_t)n * a.C * plane + (size_t)(th0 + rowb) * a.pw + (tw0 + col);
            const size_t bskip = out_bstride - (size_t)a.C * plane;
            int cc = 0;
            const int planes = a.B * a.C;
            const float4 zero4 = make_float4(0.f, 0.f, 0.f, 0.f);
            for (int p = 0; p < planes; ++p) {
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const int qc = wave * 64 + 256 * j + lane;
                    if (j < nj && qc < nchunk) *reinterpret_cast<float4*>(box0 + qc * 4) = zero4;
                }
                __syncthreads();
#pragma unroll
                for (int k = 0; k < SPT; ++k) {
                    const float g = src[k * ostep];
                    atomicAdd(box0 + r0[k], g * w00[k]);
                    atomicAdd(box0 + r0[k] + s1[k], g * w01[k]);          // s1 == 0: the +1 column is outside and its weight exactly 0
                    atomicAdd(box0 + r1[k], g * w10[k]);
                    atomicAdd(box0 + r1[k] + s1[k], g * w11[k]);
                }
                __syncthreads();
                float* ge = gerp + (size_t)p * img_plane;
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const int qc = wave * 64 + 256 * j + lane;
                    if (j < nj && qc < nchunk) {
                        const float4 v = *reinterpret_cast<const float4*>(box0 + qc * 4);
                        float* q = ge + goff[j];
                        if (v.x != 0.0f) atomicAdd(q, v.x);
                        if (v.y != 0.0f) atomicAdd(q + 1, v.y);
                        if (v.z != 0.0f) atomicAdd(q + 2, v.z);
                        if (v.w != 0.0f) atomicAdd(q + 3, v.w);
                    }
                }
                src += plane;
                if (++cc == a.C) { cc = 0; src += bskip; }
            }
            return;
        }
        // blockIdx.y owns a contiguous range of the B*C image planes (small launches — few tiles, e.g. 18 patches of
        // 128^2 — are split over the planes so that the chip is filled; the geometry prologue is repeated per range)
        const int planes_all = a.B * a.C;
        const int per = (planes_all + (int)gridDim.y - 1) / (int)gridDim.y;
        const int p_begin = (int)blockIdx.y * per, planes = min(planes_all, p_begin + per);
        if (p_begin >= planes) return;
        const float* img = erp + (size_t)p_begin * img_plane;
        dma(img, box0 + (p_begin & 1) * E2P_BOXF);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        float* dst = out + (size_t)(p_begin / a.C) * out_bstride + (size_t)(p_begin % a.C) * plane;
        const size_t bskip = out_bstride - (size_t)a.C * plane;
        int cc = p_begin % a.C;
        for (int p = p_begin; p < planes; ++p) {
            const float* cur = box0 + (p & 1) * E2P_BOXF;
            if (p + 1 < planes) { img += img_plane; dma(img, box0 + ((p + 1) & 1) * E2P_BOXF); }
            float r[SPT];
#pragma unroll
            for (int k = 0; k < SPT; ++k) {
                const float a0 = cur[r0[k]], a1 = cur[r0[k] + 1];              // one ds_read2_b32 per tap row
                const float b0 = cur[r1[k]], b1 = cur[r1[k] + 1];
                r[k] = e2p_blend(a0, s1[k] ? a1 : a0, b0, s1[k] ? b1 : b0, w00[k], w01[k], w10[k], w11[k]);
            }
#pragma unroll
            for (int k = 0; k < SPT; ++k) dst[k * ostep] = r[k];
            dst += plane;
            if (++cc == a.C) { cc = 0; dst += bskip; }
            // counted wait: the DMA pieces are older than this trip's 4 stores, which may stay in flight across the
            // barrier (a plain __syncthreads() would drain them: its fence waits vmcnt(0) while an LDS-DMA is pending)
            if (SPT == 4) asm volatile("s_waitcnt vmcnt(4) lgkmcnt(0)" ::: "memory");
            else          asm volatile("s_waitcnt vmcnt(1) lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
        }
    } else {
        // direct gathers (same taps): tiles containing a pole, ragged tiles, odd row pitch
        bool ok[SPT];
#pragma unroll
        for (int k = 0; k < SPT; ++k) ok[k] = (th0 + rowb + RSTEP * k < a.ph) && (tw0 + col < a.pw);
        if (BWD) {                                                // direct global atomics for this (tile, batch item)
            float* gerp = (float*)a.erp;
            const float* srcb = (const float*)a.pers + (size_t)n * a.C * plane + (size_t)(th0 + rowb) * a.pw + (tw0 + col)
                              + (size_t)fb_b * out_bstride;
            for (int c = 0; c < a.C; ++c) {
                float* ge = gerp + ((size_t)fb_b * a.C + c) * img_plane;
#pragma unroll
                for (int k = 0; k < SPT; ++k) {
                    if (!ok[k] || !(w00[k] == w00[k])) continue;              // outside a ragged tile / NaN sample (q4)
                    const float g = srcb[(size_t)c * plane + k * ostep];
                    const int g0 = y0[k] * W + x0[k], g1 = y1[k] * W + x0[k];
                    atomicAdd(ge + g0, g * w00[k]); atomicAdd(ge + g0 + s1[k], g * w01[k]);
                    atomicAdd(ge + g1, g * w10[k]); atomicAdd(ge + g1 + s1[k], g * w11[k]);
                }
            }
            return;
        }
        float* dstb = out + (size_t)fb_b * out_bstride;
        for (int c = 0; c < a.C; ++c) {
            const float* img = erp + ((size_t)fb_b * a.C + c) * img_plane;
            float* dst = dstb + (size_t)c * plane;
#pragma unroll
            for (int k = 0; k < SPT; ++k) {
                const int g0 = y0[k] * W + x0[k], g1 = y1[k] * W + x0[k];
                const float v00 = img[g0], v01 = img[g0 + s1[k]], v10 = img[g1], v11 = img[g1 + s1[k]];
                const float r = e2p_blend(v00, v01, v10, v11, w00[k], w01[k], w10[k], w11[k]);
                if (ok[k]) dst[k * ostep] = r;
            }
        }
    }
}

// ------------------------------------------------------------------ planar output, one wave per small sample tile (default)
// e2p_lds_kernel above synchronises a 256-thread block once per image plane and keeps two boxes in flight; measured 46 us for the
// 164 MB of BASELINE cfg 4 (3.6 TB/s), its waves parked at the barrier / the DMA wait most of the time.  Here ONE WAVE owns a
// tile of 8 x 32 samples (4 horizontally adjacent ones per lane: one 16-byte store per lane and plane) and streams the B*C image planes through a ring of NB LDS slots: per plane the bounding box of the tile's taps on
// the ERP arrives by LDS-DMA (buffer_load ... lds, 1 KiB pieces), the wave waits with a COUNTED s_waitcnt for exactly the
// pieces of the stage it is about to read (the NB-1 younger stages and the stores stay in flight), takes the taps from LDS and
// stores 4 samples per lane.  No barrier, no block-level reduction: the box of every tile is a constant of the geometry (table
// built once per handle from the same sampling coordinates).  The stage loop is instantiated per NJ = pieces per box, so every
// wait count and piece loop is a compile-time constant.  Tiles whose box exceeds the slot (the pole inside or next to the
// tile) are listed per geometry and handled by extra blocks of the same launch with direct gathers, one per (tile, batch item).
constexpr int E2B_NPX = 4;                      // samples per lane of the 8 x 32 tile (the kernels take NPX = 4 | 2 as a template parameter: 8 x 32 | 4 x 32 samples)
constexpr int E2B_NJMAX = 8;                    // 1-KiB DMA pieces per box at most
constexpr int E2B_RING_KB = 12;                 // LDS ring per wave (13 waves per CU by LDS; NJ <= 3: 4 slots, <= 6: 2 slots, else 1)

typedef __amdgpu_buffer_rsrc_t e2b_rsrc_t;
typedef __attribute__((address_space(3))) void* e2b_lptr_t;
__device__ __forceinline__ void e2b_dma16(e2b_rsrc_t rs, unsigned char* lds, unsigned voff, unsigned soff)
{
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (e2b_lptr_t)lds, 16, (int)voff, (int)soff, 0, 0);
}
template <int N> __device__ __forceinline__ void e2b_wait_vm()
{
    static_assert(N >= 0 && N <= 63, "vmcnt is a 6-bit counter");
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}
template <typename T> struct E2BPair;
template <> struct E2BPair<float> {
    static __device__ __forceinline__ void ld(const unsigned char* b, int o, float& x, float& y)
    { const float* p = reinterpret_cast<const float*>(b) + o; x = p[0]; y = p[1]; }
};
template <> struct E2BPair<__half> {
    static __device__ __forceinline__ void ld(const unsigned char* b, int o, float& x, float& y)
    {   // halfs o, o+1: one ds_read2_b32 of the two 32-bit words around them + a byte-align
        const unsigned* p = reinterpret_cast<const unsigned*>(b) + (o >> 1);
        const unsigned w0 = p[0], w1 = p[1];
        const unsigned v = (o & 1) ? __builtin_amdgcn_alignbyte(w1, w0, 2u) : w0;
        const __half2 h = *reinterpret_cast<const __half2*>(&v);
        x = __low2float(h); y = __high2float(h);
    }
};

// clamped sampling coordinate of sample (n, h, w): from the per-geometry table, or evaluated on the fly (same function, same bits)
__device__ __forceinline__ void e2b_xy(const E2PArgs& a, int n, int h, int w, float& ix, float& iy)
{
    if (a.ixy) { const float2 c = a.ixy[((size_t)n * a.ph + h) * a.pw + w]; ix = c.x; iy = c.y; }
    else e2p_sample_xy(a, n, h, w, ix, iy);
}

// table entry of one tile: x = bw4 | bh << 12 | fits << 31 (bw4 = 16-byte chunks per box row, bh = box rows),
//                          y = xs4 | ymin << 16 (first box column, chunk-aligned, the box wraps at the seam; first box row)
// the tile is E2B_TH = 8 rows x E2B_TW = 32 columns of samples, 4 per lane; every lane stores 4 adjacent samples of ONE row: one 16-byte
// (fp16: 8-byte) store per lane and plane.  Two lane -> sample maps (e2p_box_kernel's ROWMAP); for the second one the 4x4 block (4 rows x 4
// columns) held by each quad of lanes is transposed with DPP moves before the store.
constexpr int E2B_TW = 32;                       // (tile height: 2 NPX = 8 or 4 sample rows, a template parameter)

__device__ __forceinline__ float e2b_dpp_xor1(float v) { return __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(v), 0xB1, 0xF, 0xF, true)); }   // quad_perm [1,0,3,2]
__device__ __forceinline__ float e2b_dpp_xor2(float v) { return __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(v), 0x4E, 0xF, 0xF, true)); }   // quad_perm [2,3,0,1]
// r[k] of lane i (i = lane % 4)  ->  r[k] = what lane k of the quad held in r[i]
__device__ __forceinline__ void e2b_quad_transpose(float (&r)[4], int lane)
{
    const bool o1 = lane & 1, o2 = lane & 2;
#pragma unroll
    for (int q = 0; q < 2; ++q) {                                  // 2x2 blocks: exchange M[2p][2q+1] <-> M[2p+1][2q]
        const float y = e2b_dpp_xor1(o1 ? r[2 * q] : r[2 * q + 1]);
        r[2 * q + 1] = o1 ? r[2 * q + 1] : y;
        r[2 * q] = o1 ? y : r[2 * q];
    }
#pragma unroll
    for (int q = 0; q < 2; ++q) {                                  // off-diagonal 2x2 blocks: M[p][q+2] <-> M[p+2][q]
        const float y = e2b_dpp_xor2(o2 ? r[q] : r[q + 2]);
        r[q + 2] = o2 ? r[q + 2] : y;
        r[q] = o2 ? y : r[q];
    }
}

// the 4 x 32 tile (NPX = 2): lane l holds (row l/32, column l%32) and (row l/32 + 2, same column); after the exchange with lane l ^ 1 an even lane
// holds its first row's columns (c, c+1), an odd lane its second row's columns (c-1, c): two adjacent samples of one row per lane
__device__ __forceinline__ void e2b_pair_transpose(float (&r)[2], int lane)
{
    const bool odd = lane & 1;
    const float y = e2b_dpp_xor1(odd ? r[0] : r[1]);
    r[0] = odd ? y : r[0];
    r[1] = odd ? r[1] : y;
}
__device__ __forceinline__ void e2b_transpose(float (&r)[4], int lane) { e2b_quad_transpose(r, lane); }
__device__ __forceinline__ void e2b_transpose(float (&r)[2], int lane) { e2b_pair_transpose(r, lane); }

template <int NPX>
__global__ __launch_bounds__(256) void e2b_tiles_kernel(E2PArgs a, uint2* __restrict__ ent, int tiles_x, int tiles_pp, int ntiles, int epc,
                                                        int cap_chunks, int odd_pitch, int* __restrict__ stats)
{
    const int wid = (int)((blockIdx.x * 256 + threadIdx.x) >> 6), lane = threadIdx.x & 63;
    if (wid >= ntiles) return;
    const int n = wid / tiles_pp, t = wid - n * tiles_pp;
    const int th0 = (t / tiles_x) * (2 * NPX), tw0 = (t % tiles_x) * E2B_TW;
    const int w = min(tw0 + (lane & 31), a.pw - 1);
    const int W = a.W, H = a.H, half = W >> 1;
    int x0[NPX], ymin = 0x7fffffff, ymax = -1;
#pragma unroll
    for (int k = 0; k < NPX; ++k) {
        const int h = min(th0 + (lane >> 5) + 2 * k, a.ph - 1);
        float ix, iy;
        e2b_xy(a, n, h, w, ix, iy);
        const int y0 = (int)floorf(iy);                            // (NaN -> 0: ATen clips the NaN row coordinate of quirk q4 to 0)
        x0[k] = (int)floorf(ix);
        ymin = min(ymin, y0); ymax = max(ymax, min(y0 + 1, H - 1));
    }
    const int xc = __shfl(x0[0], 0);
    int dmin = 0x7fffffff, dmax = -0x7fffffff;
#pragma unroll
    for (int k = 0; k < NPX; ++k) {
        int d = x0[k] - xc;
        if (d >= half) d -= W;
        if (d < -half) d += W;
        dmin = min(dmin, d); dmax = max(dmax, d);
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        ymin = min(ymin, __shfl_xor(ymin, o)); ymax = max(ymax, __shfl_xor(ymax, o));
        dmin = min(dmin, __shfl_xor(dmin, o)); dmax = max(dmax, __shfl_xor(dmax, o));
    }
    int xs = xc + dmin;
    if (xs < 0) xs += W;
    if (xs >= W) xs -= W;
    const int xs4 = xs / epc * epc, shift = xs - xs4;
    int bw4 = (dmax - dmin + 2 + shift + epc - 1) / epc;           // columns x0 .. x0+1 of every sample, whole 16-byte chunks
    if (odd_pitch && (bw4 & 1) == 0 && (bw4 + 1) * epc <= W) ++bw4;   // odd number of chunks per box row: consecutive rows start 4, 12, 20, 28 banks apart
    const int bh = ymax - ymin + 1;
    const bool fits = (W % epc == 0) && bw4 * epc <= W && bw4 < 4096 && bh < 4096 && bw4 * bh <= cap_chunks;
    if (lane == 0) {
        ent[wid] = make_uint2((unsigned)(bw4 & 4095) | ((unsigned)(bh & 4095) << 12) | (fits ? 0x80000000u : 0u), (unsigned)xs4 | ((unsigned)ymin << 16));
        if (fits) atomicMax(&stats[0], bw4 * bh);
        else stats[2 + atomicAdd(&stats[1], 1)] = wid;             // fallback list (order irrelevant)
    }
}

// NPX results of one lane -> NPX adjacent elements, one store
template <typename T> struct E2BStore4;
template <> struct E2BStore4<float> {
    static __device__ __forceinline__ void st(float* p, const float (&r)[2]) { *reinterpret_cast<float2*>(p) = make_float2(r[0], r[1]); }
    static __device__ __forceinline__ void st_nt(float* p, const float (&r)[2])
    {
        typedef float v2f __attribute__((ext_vector_type(2)));
        const v2f v = {r[0], r[1]};
        __builtin_nontemporal_store(v, reinterpret_cast<v2f*>(p));
    }
    static __device__ __forceinline__ void st(float* p, const float (&r)[4]) { *reinterpret_cast<float4*>(p) = make_float4(r[0], r[1], r[2], r[3]); }
    // non-temporal form: the output is never re-read by this kernel (the ERP boxes are), and at 16 panoramas — 327 MB per launch, more than
    // the 256-MB memory-side cache — it is what keeps the launch at the 8-panorama rate (106 -> 64-74 us; 8 panoramas 38.5 -> 36.8).  sc1
    // (write-through, line dropped from the XCD's L2) measured no better than plain stores.
    static __device__ __forceinline__ void st_nt(float* p, const float (&r)[4])
    {
        typedef float v4f __attribute__((ext_vector_type(4)));
        const v4f v = {r[0], r[1], r[2], r[3]};
        __builtin_nontemporal_store(v, reinterpret_cast<v4f*>(p));
    }
};
template <> struct E2BStore4<__half> {
    static __device__ __forceinline__ void st(__half* p, const float (&r)[2])
    { __half2 v = __floats2half2_rn(r[0], r[1]); *reinterpret_cast<unsigned*>(p) = *reinterpret_cast<unsigned*>(&v); }
    static __device__ __forceinline__ void st_nt(__half* p, const float (&r)[2])
    { __half2 v = __floats2half2_rn(r[0], r[1]); __builtin_nontemporal_store(*reinterpret_cast<unsigned*>(&v), reinterpret_cast<unsigned*>(p)); }
    static __device__ __forceinline__ void st(__half* p, const float (&r)[4])
    { __half2 lo = __floats2half2_rn(r[0], r[1]), hi = __floats2half2_rn(r[2], r[3]); uint2 v; v.x = *reinterpret_cast<unsigned*>(&lo); v.y = *reinterpret_cast<unsigned*>(&hi); *reinterpret_cast<uint2*>(p) = v; }
    static __device__ __forceinline__ void st_nt(__half* p, const float (&r)[4])
    {
        __half2 lo = __floats2half2_rn(r[0], r[1]), hi = __floats2half2_rn(r[2], r[3]);
        typedef unsigned v2u __attribute__((ext_vector_type(2)));
        const v2u v = {*reinterpret_cast<unsigned*>(&lo), *reinterpret_cast<unsigned*>(&hi)};
        __builtin_nontemporal_store(v, reinterpret_cast<v2u*>(p));
    }
};

template <typename T, int NBMAX, bool ROWMAP, int NPX = E2B_NPX>
__global__ __launch_bounds__(64, 4) void e2p_box_kernel(E2PArgs a, const uint4* __restrict__ work, int tiles_x, int tiles_pp, unsigned tensor_bytes)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char e2b_smem[];        // the ONLY LDS object of this kernel
    constexpr int EPC = 16 / (int)sizeof(T), TILE_H = 2 * NPX, LPR = E2B_TW / NPX;   // tile rows; lanes per tile row of the direct lane map
    const int lane = threadIdx.x;
    long long tr0 = 0, tr1 = 0;
    if (OMNI_DBG(a, 16)) tr0 = wall_clock64();
    // block -> (tile, plane range): the work table of this (geometry, plane count), e2p_work_table().  x = tile | gather flag << 31,
    // y = first plane | planes << 16, (z, w) = the tile's box entry.  Block b runs on XCD b % 8 and the table keeps the tiles of one
    // ERP region on one XCD (an ERP pixel is sampled by 2.1 patches on average; only tiles on ONE XCD share an L2).
    const uint4 wk = work[blockIdx.x];                             // wave-uniform address: scalar load
    const int np = (int)(wk.y >> 16), p_start = (int)(wk.y & 0xffffu);
    if (np == 0) return;                                           // padding
    const bool fb_block = (wk.x >> 31) != 0;
    const int wid = (int)(wk.x & 0x7fffffffu);
    const uint2 e = make_uint2(wk.z, wk.w);
    const int n = wid / tiles_pp, t = wid - n * tiles_pp;
    const int th0 = (t / tiles_x) * TILE_H, tw0 = (t % tiles_x) * E2B_TW;
    const int W = a.W, H = a.H;
    // lane -> samples.  ROWMAP (2-byte elements): lane computes (row lane/32 + 2k, column lane%32) — a half-wave, the conflict group of
    // ds_read2_b32, reads the taps of 32 consecutive samples of one row — and the results are transposed inside each quad before the
    // store.  Otherwise (4-byte elements) lane computes the 4 adjacent samples (row lane/8, columns 4 (lane%8) + k) it stores; measured
    // per shape: fp32 22 vs 27 us at P = 128 and equal at P = 256 in favour of the direct form, fp16 cfg5 72 vs 126 us in favour of ROWMAP.
    const int w = ROWMAP ? tw0 + (lane & 31) : tw0 + NPX * (lane % LPR), hb = ROWMAP ? th0 + (lane >> 5) : th0 + lane / LPR;
    const int xs4 = (int)(e.y & 0xffff), ymin = (int)(e.y >> 16), bw4 = (int)(e.x & 4095), bh = (int)((e.x >> 12) & 4095);
    const int pitch = bw4 * EPC;

    // ---- the sampling coordinates of my NPX samples go out FIRST (plain loads, hand-counted: the compiler does not see them), the first
    // NB boxes right behind them, and the taps are computed while both travel: a block's set-up was two dependent round trips in front
    // of its first DMA (work entry -> coordinates; 2.3 us of a third-of-a-tile block's 5.6 us), now the boxes ride on the second
    typedef float e2b_v2f __attribute__((ext_vector_type(2)));
    e2b_v2f cxy[NPX];
    const bool tab = a.ixy != nullptr;
#pragma unroll
    for (int k = 0; k < NPX; ++k) {
        const int hh = ROWMAP ? hb + 2 * k : hb, ww = ROWMAP ? w : w + k;
        if (tab) {
            const float2* q = a.ixy + ((size_t)n * a.ph + hh) * a.pw + ww;
            asm volatile("global_load_dwordx2 %0, %1, off" : "=v"(cxy[k]) : "v"(q) : "memory");
        } else {
            float ix, iy;
            e2p_sample_xy(a, n, hh, ww, ix, iy);
            cxy[k].x = ix; cxy[k].y = iy;
        }
    }
    const int plane = a.ph * a.pw;
    const size_t img_plane = (size_t)H * W;
    const e2b_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(a.erp), (short)0, (int)tensor_bytes, 0x00020000);
    const unsigned rowb = (unsigned)W * (unsigned)sizeof(T), planeb = (unsigned)img_plane * (unsigned)sizeof(T);
    const int nchunk = bw4 * bh, njj = (nchunk + 63) >> 6;
    unsigned g[E2B_NJMAX];                                          // byte offset of my chunk of piece q inside an image plane
    auto prologue = [&]<int NJ>(std::integral_constant<int, NJ>) {
        constexpr int NBR = E2B_RING_KB / NJ >= 4 ? 4 : E2B_RING_KB / NJ >= 2 ? 2 : 1;
        constexpr int NB = NBR < NBMAX ? NBR : NBMAX;
        constexpr unsigned slot_bytes = NJ * 1024u;
        const float rbw = __builtin_amdgcn_rcpf((float)bw4);
#pragma unroll
        for (int q = 0; q < NJ; ++q) {
            const int qc = q * 64 + lane;
            const int rr = (int)(((float)qc + 0.5f) * rbw);                    // qc / bw4, exact for qc, bw4 <= 4096
            int gx = xs4 + (qc - rr * bw4) * EPC;
            if (gx >= W) gx -= W;                                                 // the box wraps at the seam
            g[q] = qc < nchunk ? (unsigned)(ymin + rr) * rowb + (unsigned)gx * (unsigned)sizeof(T) : 0x80000000u;   // past the end: zeros
            if (OMNI_DBG(a, 64) && (qc & 63) >= 40) g[q] = 0x80000000u;   // (debug bit 64: 3/8 of every piece's lanes read nothing — the volume packed row spans would move)
        }
        if (!OMNI_DBG(a, 2)) {
#pragma unroll
            for (int d = 0; d < NB; ++d) {
                const unsigned so = OMNI_DBG(a, 32) ? 0u : (unsigned)(p_start + d) * planeb;       // (debug bit 32: every plane reads plane 0 — no input traffic)
#pragma unroll
                for (int q = 0; q < NJ; ++q) e2b_dma16(rs, e2b_smem + (unsigned)d * slot_bytes + q * 1024, g[q], so);
            }
            e2b_wait_vm<NB * NJ>();                                 // the coordinates are older than the NB * NJ pieces: they have landed
        } else e2b_wait_vm<0>();
    };
    if (fb_block) e2b_wait_vm<0>();
    else switch (njj) {
    case 1: prologue(std::integral_constant<int, 1>()); break;
    case 2: prologue(std::integral_constant<int, 2>()); break;
    case 3: prologue(std::integral_constant<int, 3>()); break;
    case 4: prologue(std::integral_constant<int, 4>()); break;
    case 5: prologue(std::integral_constant<int, 5>()); break;
    case 6: prologue(std::integral_constant<int, 6>()); break;
    case 7: prologue(std::integral_constant<int, 7>()); break;
    default: prologue(std::integral_constant<int, 8>()); break;
    }
    // ---- taps of my NPX samples: LDS element offsets inside the box + ATen's four weights
    int r0[NPX], r1[NPX];
    float w00[NPX], w01[NPX], w10[NPX], w11[NPX];
    int gx0[NPX], gy0[NPX], gy1[NPX];                              // (fallback path: absolute taps)
#pragma unroll
    for (int k = 0; k < NPX; ++k) {
        asm volatile("" : "+v"(cxy[k]));                           // (use only behind the wait above)
        const float ix = cxy[k].x, iy = cxy[k].y;
        const float fx = floorf(ix), fy = floorf(iy);
        const int x0 = (int)fx, y0 = (int)fy;
        const float tx = ix - fx, ty = iy - fy, ex = 1.0f - tx, ey = 1.0f - ty;
        w00[k] = ey * ex; w01[k] = ey * tx; w10[k] = ty * ex; w11[k] = ty * tx;
        const int y1 = min(y0 + 1, H - 1);
        gx0[k] = x0; gy0[k] = y0; gy1[k] = y1;
        int c0 = x0 - xs4;                                         // column inside the (seam-wrapping) box
        if (c0 < 0) c0 += W;
        if (x0 + 1 >= W) {
            // ix is clamped to W-1: tx == 0, the +1 column is outside the image and its weights (w01, w11) are exactly 0.  Read the
            // pair (x0-1, x0) instead and move the x0 weights to the pair's SECOND element: e2p_blend then returns
            // 0 + (v00 w00 + v10 w10) with the same roundings, and nothing outside the image row is read.
            c0 -= 1;
            w01[k] = w00[k]; w00[k] = 0.0f; w11[k] = w10[k]; w10[k] = 0.0f;
        }
        r0[k] = (y0 - ymin) * pitch + c0;
        r1[k] = (y1 - ymin) * pitch + c0;
    }
    const size_t out_bstride = (size_t)a.tab.N * a.C * plane;
    // ROWMAP: after the transpose inside each group of NPX lanes, lane l holds row hb + 2 (l % NPX), columns NPX ((l % 32) / NPX) .. + NPX - 1 of the tile
    const int b0 = p_start / a.C, c0p = p_start - b0 * a.C;        // batch item / channel of my first plane
    T* out = (T*)a.pers + (size_t)b0 * out_bstride + ((size_t)n * a.C + c0p) * plane +
             (ROWMAP ? (size_t)(hb + 2 * (lane % NPX)) * a.pw + tw0 + NPX * ((lane & 31) / NPX) : (size_t)hb * a.pw + w);
    const size_t bskip = out_bstride - (size_t)a.C * plane;

    if (fb_block) {
        // ---- direct gathers (same taps, same fma chain) for a tile whose box does not fit a slot: planes [p_start, p_start + np), FBU at a
        // time — their pair loads are all in flight before the first is used (a block of these is a chain of memory round trips and
        // nothing else)
        const T* img = (const T*)a.erp + (size_t)p_start * img_plane;
        T* dstb = out;
        int cc = c0p;
        int g0[NPX], g1[NPX];
#pragma unroll
        for (int k = 0; k < NPX; ++k) {
            const int sh = (gx0[k] + 1 >= W) ? 1 : 0;              // pair moved one column left (see above)
            g0[k] = gy0[k] * W + gx0[k] - sh; g1[k] = gy1[k] * W + gx0[k] - sh;
        }
        constexpr int FBU = sizeof(T) == 4 ? 2 : 3;              // (three fp32 planes: 48 result registers, and the kernel spills at 128)
        for (int p = 0; p < np; p += FBU) {
            float v[FBU][NPX][4];
#pragma unroll
            for (int u = 0; u < FBU; ++u) {
                const T* im = img + (size_t)min(u, np - 1 - p) * img_plane;      // (past the range: the last plane again, result unused)
#pragma unroll
                for (int k = 0; k < NPX; ++k) {
                    Pair<T>::ld(im + g0[k], v[u][k][0], v[u][k][1]);
                    Pair<T>::ld(im + g1[k], v[u][k][2], v[u][k][3]);
                }
            }
#pragma unroll
            for (int u = 0; u < FBU; ++u) {
                if (p + u < np) {
                    float r[NPX];
#pragma unroll
                    for (int k = 0; k < NPX; ++k) r[k] = e2p_blend(v[u][k][0], v[u][k][1], v[u][k][2], v[u][k][3], w00[k], w01[k], w10[k], w11[k]);
                    if (ROWMAP) e2b_transpose(r, lane);
                    E2BStore4<T>::st(dstb, r);
                    dstb += plane;
                    if (++cc == a.C) { cc = 0; dstb += bskip; }
                }
            }
            img += (size_t)FBU * img_plane;
        }
        return;
    }
    // nothing but LDS-DMA pieces and stores below
    if (OMNI_DBG(a, 16)) tr1 = wall_clock64();

    auto run = [&]<int NJ>(std::integral_constant<int, NJ>) {
        // ring of E2B_RING_KB 1-KiB pieces per wave: a box of NJ pieces gets NB = min(NBMAX, largest power of two <= RING / NJ) slots of
        // exactly NJ KiB — small boxes (the common case) keep 4 stages in flight, the rare large ones 2 or 1, in the same LDS footprint
        constexpr int NBR = E2B_RING_KB / NJ >= 4 ? 4 : E2B_RING_KB / NJ >= 2 ? 2 : 1;
        constexpr int NB = NBR < NBMAX ? NBR : NBMAX;
        constexpr unsigned slot_bytes = NJ * 1024u;
        auto issue = [&](int p, int slot) {
            if (OMNI_DBG(a, 2)) return;
            unsigned char* dst = e2b_smem + (unsigned)slot * slot_bytes;
            const unsigned so = OMNI_DBG(a, 32) ? 0u : (unsigned)(p_start + p) * planeb;
#pragma unroll
            for (int q = 0; q < NJ; ++q) e2b_dma16(rs, dst + q * 1024, g[q], so);
        };
        T* dst = out;
        int cc = c0p;
        auto consume = [&](int slot, float (&r)[NPX]) {
            const unsigned char* box = e2b_smem + (unsigned)slot * slot_bytes;
#pragma unroll
            for (int k = 0; k < NPX; ++k) {
                float a0, a1, b0, b1;
                E2BPair<T>::ld(box, r0[k], a0, a1);
                E2BPair<T>::ld(box, r1[k], b0, b1);
                r[k] = e2p_blend(a0, a1, b0, b1, w00[k], w01[k], w10[k], w11[k]);
            }
            if (ROWMAP) e2b_transpose(r, lane);
        };
        auto store = [&](const float (&r)[NPX]) {
            if (OMNI_DBG(a, 128)) { if (lane == 0) E2BStore4<T>::st_nt(dst, r); }      // (debug bit 128: the store instruction with ONE active lane — its acknowledge without its bytes)
            else if (!OMNI_DBG(a, 1)) { if (a.store_mode) E2BStore4<T>::st_nt(dst, r); else E2BStore4<T>::st(dst, r); }
            dst += plane;
            if (++cc == a.C) { cc = 0; dst += bskip; }
        };
        // groups of NB stages, then np % NB single stages (NB <= np: the work table never cuts a range shorter).  vmcnt counts, per stage s of a
        // group (in-order completion; the ONE store of a stage counts like a DMA piece):  first group  (NB-1) NJ + s   |  middle  (NB-1)(1 + NJ)
        //                                             last group   (NB-1) + (NB-1-s) NJ   |  only group  (NB-1-s) NJ + s
        // (round 3: issuing a stage's refill BEFORE its store, so that a write acknowledge has two stage-times instead of one before a
        //  counted wait can stall on it, measured no different: 39.3 vs 38.5 us — the stores are not what the waits wait for)
        const int groups = np / NB;                                 // (the first NB stages went out in the prologue)
        auto group = [&]<int KIND>(std::integral_constant<int, KIND>, int p0) {
            [&]<int... S>(std::integer_sequence<int, S...>) {
                (([&] {
                    constexpr int CNT = KIND == 0 ? (NB - 1) * NJ + S : KIND == 1 ? (NB - 1) * (1 + NJ)
                                      : KIND == 2 ? (NB - 1) + (NB - 1 - S) * NJ : (NB - 1 - S) * NJ + S;
                    e2b_wait_vm<CNT>();
                    float r[NPX];
                    consume(S, r);
                    store(r);
                    if (KIND <= 1) { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); issue(p0 + S + NB, S); }
                }()), ...);
            }(std::make_integer_sequence<int, NB>());
        };
        if (groups == 1) group(std::integral_constant<int, 3>(), 0);
        else {
            group(std::integral_constant<int, 0>(), 0);
            for (int gi = 1; gi + 1 < groups; ++gi) group(std::integral_constant<int, 1>(), gi * NB);
            group(std::integral_constant<int, 2>(), (groups - 1) * NB);
        }
        for (int p = groups * NB; p < np; ++p) {                 // an odd plane count (3 planes of ONE panorama): the rest, one stage at a time
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            issue(p, 0);
            e2b_wait_vm<0>();
            float r[NPX];
            consume(0, r);
            store(r);
        }
    };
    switch (njj) {
    case 1: run(std::integral_constant<int, 1>()); break;
    case 2: run(std::integral_constant<int, 2>()); break;
    case 3: run(std::integral_constant<int, 3>()); break;
    case 4: run(std::integral_constant<int, 4>()); break;
    case 5: run(std::integral_constant<int, 5>()); break;
    case 6: run(std::integral_constant<int, 6>()); break;
    case 7: run(std::integral_constant<int, 7>()); break;
    default: run(std::integral_constant<int, 8>()); break;
    }
    if (OMNI_DBG(a, 16)) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if (lane == 0 && a.trace) {
            long long* t = a.trace + 4 * (size_t)blockIdx.x;
            t[0] = tr0; t[1] = tr1; t[2] = wall_clock64();
            t[3] = (long long)(unsigned)__builtin_amdgcn_s_getreg((31 << 11) | 4) | ((long long)(unsigned)__builtin_amdgcn_s_getreg((31 << 11) | 20) << 32) | ((long long)njj << 40);
        }
    }
}

// ------------------------------------------------------------------ backward
// g_erp[b,c,y,x] = sum over patch samples and their four taps of w_tap * g_pers[b,c,h,w,n]: the transpose of the bilinear
// gather (ATen grid_sampler_2d_backward with bilinear / border / align_corners=True; taps outside the image are dropped).
// One thread per patch sample, all B*C planes; fp32 hardware atomics into a zeroed g_erp (the summation order is not
// deterministic, exactly like the reference's CUDA/HIP grid_sample backward).
__global__ __launch_bounds__(256) void e2p_bwd_kernel(E2PArgs a /* erp = g_erp (out), pers = g_pers (in) */, int n_fastest, int total)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= total) return;
    int n, h, w;
    if (n_fastest) { n = i % a.tab.N; w = (i / a.tab.N) % a.pw; h = i / (a.tab.N * a.pw); }     // [B,C,h,w,N]: coalesced reads
    else           { w = i % a.pw; h = (i / a.pw) % a.ph; n = i / (a.pw * a.ph); }              // [B,N,C,h,w]
    float ix, iy;
    if (a.ixy) { const float2 c = a.ixy[((size_t)n * a.ph + h) * a.pw + w]; ix = c.x; iy = c.y; }
    else e2p_sample_xy(a, n, h, w, ix, iy);
    if (!(ix == ix) || !(iy == iy)) return;                       // q4: an odd x odd patch has a NaN centre sample
    const float fx = floorf(ix), fy = floorf(iy);
    const int x0 = (int)fx, y0 = (int)fy;
    const float tx = ix - fx, ty = iy - fy, ex = 1.0f - tx, ey = 1.0f - ty;
    const bool okx = x0 + 1 < a.W, oky = y0 + 1 < a.H;
    const size_t plane = (size_t)a.H * a.W, pp = (size_t)a.ph * a.pw;
    float* gerp = (float*)a.erp;
    const float* gp = (const float*)a.pers;
    const size_t o00 = (size_t)y0 * a.W + x0;
    for (int b = 0; b < a.B; ++b)
        for (int c = 0; c < a.C; ++c) {
            const size_t src = n_fastest ? ((((size_t)b * a.C + c) * a.ph + h) * a.pw + w) * a.tab.N + n
                                         : (((size_t)b * a.tab.N + n) * a.C + c) * pp + (size_t)h * a.pw + w;
            const float g = gp[src];
            float* e = gerp + ((size_t)b * a.C + c) * plane + o00;
            atomicAdd(e, g * (ey * ex));
            if (okx) atomicAdd(e + 1, g * (ey * tx));
            if (oky) atomicAdd(e + a.W, g * (ty * ex));
            if (okx && oky) atomicAdd(e + a.W + 1, g * (ty * tx));
        }
}

// ------------------------------------------------------------------ reference output [B,C,ph,pw,N]
// A block owns one patch row h and TW = 64 columns for ALL N patches.  Wave v gathers patches
// v, v+4, ... with lane <-> w (good ERP locality), parks the results in an LDS tile laid out
// exactly like the destination ([w][n], N innermost), and the whole block then streams the tile
// out as one contiguous run of TW*N elements: the unfold/reshape of equi2pers_v3.py:112-113
// costs no extra HBM pass and the N-innermost stores stay coalesced.
constexpr int E2P_TW = 64;
constexpr int E2P_CCH = 4;                       // image planes staged per LDS round
constexpr int E2P_MAXPW = (OMNI_MAX_PATCH + 3) / 4;

template <typename T, bool PAIR>
__global__ __launch_bounds__(256) void e2p_reflayout_kernel(E2PArgs a, int tiles_w)
{
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    float* tile = reinterpret_cast<float*>(smem_raw);          // [E2P_CCH][TW*N]
    const int N = a.tab.N;
    const int h = blockIdx.x / tiles_w;
    const int w0 = (blockIdx.x % tiles_w) * E2P_TW;
    const int wv = min(E2P_TW, a.pw - w0);                     // valid columns in this tile
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int run = wv * N;                                    // contiguous elements per plane
    const int npw = (N + 3) >> 2;

    Tap tp[E2P_MAXPW];
#pragma unroll
    for (int k = 0; k < E2P_MAXPW; ++k) {
        const int n = wave + 4 * k;
        if (k < npw && n < N) tp[k] = e2p_tap<PAIR>(a, n, h, min(w0 + lane, a.pw - 1));
    }
    const T* erp = (const T*)a.erp;
    T* out = (T*)a.pers;
    const size_t img_plane = (size_t)a.H * a.W;
    const size_t out_plane = (size_t)a.ph * a.pw * N;
    const size_t out_off = ((size_t)h * a.pw + w0) * N;
    const int planes = a.B * a.C;
    for (int p0 = 0; p0 < planes; p0 += E2P_CCH) {
        const int pc = min(E2P_CCH, planes - p0);
        for (int pp = 0; pp < pc; ++pp) {
            const T* img = erp + (size_t)(p0 + pp) * img_plane;
#pragma unroll
            for (int k = 0; k < E2P_MAXPW; ++k) {
                const int n = wave + 4 * k;
                if (k < npw && n < N && lane < wv)
                    tile[pp * (E2P_TW * N) + lane * N + n] = e2p_fetch<T, PAIR>(img, tp[k]);
            }
        }
        __syncthreads();
        for (int pp = 0; pp < pc; ++pp) {
            T* dst = out + (size_t)(p0 + pp) * out_plane + out_off;
            const float* src = tile + pp * (E2P_TW * N);
            for (int i = threadIdx.x; i < run; i += 256) Store<T>::st(dst + i, src[i]);
        }
        __syncthreads();
    }
}

// ------------------------------------------------------------------ reference output [B,C,ph,pw,N], LDS-staged (round 4)
// The drop-in equi2pers() returns the reference's own layout (equi2pers_v3.py:112-113: N innermost).  e2p_reflayout_kernel above gathers through
// L1 / L2 (72 us at 8 x 18 x 256^2 where the planar box kernel takes 32); planar + a transposing pass is no faster (113 MB more in each
// direction).  Here ONE BLOCK owns a tile POSITION (8 x 32 samples) of ALL N patches: wave w stages the ERP tap boxes of its PPW patches
// (w PPW .. w PPW + PPW - 1) by LDS-DMA exactly as e2p_box_kernel does — same per-geometry box table, same taps, same e2p_blend: same bits —,
// parks its results in an LDS tile laid out like the destination ([row][column][patch]) and after ONE block barrier per plane the whole block
// streams that tile out as 16-byte pieces (8 contiguous runs of 32 N elements).  The boxes of plane p+1 are in flight from the moment plane p's
// taps have been read (one slot per patch: the barrier, the tile write and the stores are what they travel under); the output tile is double
// buffered, so the barrier of plane p also licenses the rewrite of the tile of plane p-1.  Counted waits: a wave's queue holds, in order, the
// pieces of its PPW boxes and the S store instructions of the previous plane — waiting for box j leaves (PPW-1) NJ + S younger operations, NJ
// the wave's pieces per box (the largest of its patches: smaller boxes pad with out-of-range lanes).  A wave with a patch whose box does not
// fit a slot (pole tiles) takes that patch by direct gathers and waits with vmcnt(0) throughout.
template <typename T, int PPW, int S>
__global__ __launch_bounds__(PPW == 1 ? 1024 : 640) void e2p_ref_kernel(E2PArgs a, const uint2* __restrict__ ent, int tiles_x, int tiles_pp, unsigned tensor_bytes,
                                                       int slot_bytes, int planes_per_block)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char e2r_smem[];
    constexpr int EPC = 16 / (int)sizeof(T), NPX = 4, TH = 8, TW = 32;
    // lane -> samples: (row lane / 32 + 2 k, column lane % 32) for every element size — the results go to the [row][column][patch] tile one by
    // one, and 32 consecutive columns of one row are N elements apart there: 16 distinct banks (N = 18).  The box kernel's direct map for 4-byte
    // elements (4 adjacent columns per lane) puts a wave's 64 stores on FOUR banks: 2.4 us per plane, measured.
    constexpr bool ROWMAP = true;
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)), nthreads = (int)blockDim.x;
    const int N = a.tab.N, W = a.W, H = a.H, ph_ = a.ph, pw_ = a.pw;
    const float2* __restrict__ ixy = a.ixy;
    const int t = (int)omni_xcd_remap(blockIdx.x, gridDim.x);      // an XCD owns a band of tile rows: neighbouring boxes of every patch share its L2
    const int th0 = (t / tiles_x) * TH, tw0 = (t % tiles_x) * TW;
    const int planes = a.B * a.C;
    const int p_begin = (int)blockIdx.y * planes_per_block, np = min(planes, p_begin + planes_per_block) - p_begin;
    if (np <= 0) return;
    unsigned char* const ring = e2r_smem + (unsigned)(wave * PPW) * (unsigned)slot_bytes;
    const int out_elems = TH * TW * N;
    T* const otile = reinterpret_cast<T*>(e2r_smem + (unsigned)((nthreads >> 6) * PPW) * (unsigned)slot_bytes);   // [2][TH][TW][N]
    const int w = ROWMAP ? tw0 + (lane & 31) : tw0 + 4 * (lane & 7), hb = ROWMAP ? th0 + (lane >> 5) : th0 + (lane >> 3);

    // (every per-patch array below is indexed by a COMPILE-TIME j: a run-time index would put them into scratch memory)
    auto for_j = [&](auto&& f) { [&]<int... J>(std::integer_sequence<int, J...>) { (f(std::integral_constant<int, J>()), ...); }(std::make_integer_sequence<int, PPW>()); };
    // ---- my patches: box entries, sampling coordinates, taps
    bool valid[PPW], fits[PPW];
    int xs4[PPW], ymin[PPW], bw4[PPW], nchunk[PPW], pn[PPW];
    int r0[PPW][NPX], r1[PPW][NPX], oi[PPW];                      // (oi: tile element of sample 0; sample k is OSTEP elements further)
    int g0[PPW][NPX], g1[PPW][NPX];                                // absolute tap pairs: used by the gather path only (dead in the waves without one)
    const int OSTEP = ROWMAP ? 2 * TW * N : N;
    float w00[PPW][NPX], w01[PPW][NPX], w10[PPW][NPX], w11[PPW][NPX];
    int nj = 1;
    bool sync_mode = false;
    for_j([&](auto jc) {
        constexpr int j = decltype(jc)::value;
        const int n = wave * PPW + j;
        pn[j] = n;
        valid[j] = n < N;
        fits[j] = false; xs4[j] = ymin[j] = 0; bw4[j] = 1; nchunk[j] = 0; oi[j] = 0;
        if (!valid[j]) { sync_mode = sync_mode || PPW > 1; return; }
        const uint2 e = ent[(size_t)n * tiles_pp + t];
        fits[j] = (e.x >> 31) != 0;
        xs4[j] = (int)(e.y & 0xffff); ymin[j] = (int)(e.y >> 16); bw4[j] = (int)(e.x & 4095);
        nchunk[j] = bw4[j] * (int)((e.x >> 12) & 4095);
        if (a.dbg_skip_fb && !fits[j]) { valid[j] = false; return; }      // (timing experiment: option e2p_ref_lds = 2 drops the pole patches — wrong results)
        if (fits[j]) nj = max(nj, (nchunk[j] + 63) >> 6); else sync_mode = true;
        const int pitch = bw4[j] * EPC;
#pragma unroll
        for (int k = 0; k < NPX; ++k) {
            const int hh = ROWMAP ? hb + 2 * k : hb, ww = ROWMAP ? w : w + k;
            const float2 cxy = ixy[((size_t)n * ph_ + hh) * pw_ + ww];   // (the per-geometry coordinate table: the launch requires it)
            const float ix = cxy.x, iy = cxy.y;
            const float fx = floorf(ix), fy = floorf(iy);
            const int x0 = (int)fx, y0 = (int)fy;
            const float tx = ix - fx, ty = iy - fy, ex = 1.0f - tx, ey = 1.0f - ty;
            w00[j][k] = ey * ex; w01[j][k] = ey * tx; w10[j][k] = ty * ex; w11[j][k] = ty * tx;
            const int y1 = min(y0 + 1, H - 1);
            int c0 = x0 - xs4[j];
            if (c0 < 0) c0 += W;
            int sh = 0;
            if (x0 + 1 >= W) {                                     // (see e2p_box_kernel: the pair moved one column left, the x0 weights to its second element)
                c0 -= 1; sh = 1;
                w01[j][k] = w00[j][k]; w00[j][k] = 0.0f; w11[j][k] = w10[j][k]; w10[j][k] = 0.0f;
            }
            r0[j][k] = (y0 - ymin[j]) * pitch + c0;
            r1[j][k] = (y1 - ymin[j]) * pitch + c0;
            g0[j][k] = y0 * W + x0 - sh; g1[j][k] = y1 * W + x0 - sh;
            if (k == 0) oi[j] = ((hh - th0) * TW + (ww - tw0)) * N + n;      // element of the [row][column][patch] tile
        }
    });
    nj = __builtin_amdgcn_readfirstlane(nj);
    const size_t img_plane = (size_t)H * W;
    const e2b_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(a.erp), (short)0, (int)tensor_bytes, 0x00020000);
    const unsigned rowb = (unsigned)W * (unsigned)sizeof(T), planeb = (unsigned)img_plane * (unsigned)sizeof(T);
    // output: plane p of [B,C,ph,pw,N] starts at p * ph * pw * N; row hh of my tile is the contiguous run [th0 + hh][tw0 .. tw0 + 31][0 .. N)
    const int ppr = TW * N * (int)sizeof(T) / 16, total_pieces = TH * ppr;          // 16-byte pieces per tile row / per tile
    const size_t out_plane = (size_t)a.ph * a.pw * N;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");               // every set-up load has landed: only counted operations from here on
#pragma unroll
    for (int j = 0; j < PPW; ++j)
#pragma unroll
        for (int k = 0; k < NPX; ++k) asm volatile("" ::"v"(w00[j][k]), "v"(w11[j][k]), "v"(r0[j][k]));

    auto run = [&]<int NJ, bool SYNC>(std::integral_constant<int, NJ>, std::bool_constant<SYNC>) {
        unsigned g[PPW][NJ];
        for_j([&](auto jc) {
            constexpr int j = decltype(jc)::value;
            const float rbw = __builtin_amdgcn_rcpf((float)bw4[j]);
#pragma unroll
            for (int q = 0; q < NJ; ++q) {
                const int qc = q * 64 + lane;
                const int rr = (int)(((float)qc + 0.5f) * rbw);
                int gx = xs4[j] + (qc - rr * bw4[j]) * EPC;
                if (gx >= W) gx -= W;                                 // the box wraps at the seam
                g[j][q] = (fits[j] && qc < nchunk[j]) ? (unsigned)(ymin[j] + rr) * rowb + (unsigned)gx * (unsigned)sizeof(T) : 0x80000000u;
            }
        });
        // planes in flight per patch: two where two boxes of NJ KiB fit the patch's slot (a plane's period is then half a memory round trip
        // instead of a whole one: with ONE box per patch the refill issued after plane p has a single plane-time to land — measured 2.4 us per plane)
        const int nb = (2 * NJ * 1024 <= slot_bytes) ? 2 : 1;        // (wave-uniform, a property of the launch)
        auto issue = [&]<int j>(std::integral_constant<int, j>, int p) {
            unsigned char* dst = ring + (unsigned)j * (unsigned)slot_bytes + (unsigned)((p & (nb - 1)) * NJ * 1024);
            const unsigned so = (unsigned)(p_begin + p) * planeb;
#pragma unroll
            for (int q = 0; q < NJ; ++q) e2b_dma16(rs, dst + q * 1024, g[j][q], so);
        };
        // NBC: boxes per patch in flight in the STEADY state (0: this plane waits with vmcnt(0) — the first plane, the last NB planes, SYNC waves)
        auto plane = [&]<int NBC>(std::integral_constant<int, NBC>, int p) {
            T* ot = otile + (size_t)(p & 1) * out_elems;
            const bool refill = p + nb < np;
            // ONE wait for all my boxes of this plane: behind the last of them the queue holds the boxes of the NBC - 1 planes ahead and the stores
            // of the NBC planes before this one (a plane's boxes are refilled together, after all of them have been read)
            if constexpr (SYNC || NBC == 0) e2b_wait_vm<0>();
            else e2b_wait_vm<(NBC - 1) * PPW * NJ + NBC * S>();
            float v[PPW][NPX][4];
            for_j([&](auto jc) {
                constexpr int j = decltype(jc)::value;
                if (!valid[j]) return;                              // (wave-uniform)
                if (fits[j]) {
                    const unsigned char* box = ring + (unsigned)j * (unsigned)slot_bytes + (unsigned)((p & (nb - 1)) * NJ * 1024);
#pragma unroll
                    for (int k = 0; k < NPX; ++k) {
                        E2BPair<T>::ld(box, r0[j][k], v[j][k][0], v[j][k][1]);
                        E2BPair<T>::ld(box, r1[j][k], v[j][k][2], v[j][k][3]);
                    }
                } else if constexpr (SYNC) {                        // (a pole tile: direct gathers of this plane's taps)
                    const T* im = (const T*)a.erp + (size_t)(p_begin + p) * img_plane;
#pragma unroll
                    for (int k = 0; k < NPX; ++k) {
                        Pair<T>::ld(im + g0[j][k], v[j][k][0], v[j][k][1]);
                        Pair<T>::ld(im + g1[j][k], v[j][k][2], v[j][k][3]);
                    }
                }
            });
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");       // every box of this plane has been read: its slots are DMA targets again
            if (refill) for_j([&](auto jc) { constexpr int j = decltype(jc)::value; if (valid[j] && fits[j]) issue(jc, p + nb); });
            for_j([&](auto jc) {
                constexpr int j = decltype(jc)::value;
                if (!valid[j]) return;
#pragma unroll
                for (int k = 0; k < NPX; ++k) {
                    float r = 0.0f;
                    if (SYNC || fits[j]) r = e2p_blend(v[j][k][0], v[j][k][1], v[j][k][2], v[j][k][3], w00[j][k], w01[j][k], w10[j][k], w11[j][k]);
                    Store<T>::st(ot + oi[j] + k * OSTEP, r);
                }
            });
            __syncthreads();                                        // the tile of plane p is complete (and the tile of plane p-1 has been read by everybody)
            // the tile leaves as 16-byte pieces; EVERY wave issues exactly S store instructions (a thread past the end repeats the last piece)
            const unsigned char* src = reinterpret_cast<const unsigned char*>(ot);
            unsigned char* dstp = reinterpret_cast<unsigned char*>((T*)a.pers + (size_t)(p_begin + p) * out_plane + ((size_t)th0 * a.pw + tw0) * N);
            const size_t row_stride = (size_t)a.pw * N * sizeof(T);
#pragma unroll
            for (int s2 = 0; s2 < S; ++s2) {
                const int piece = min((int)threadIdx.x + s2 * nthreads, total_pieces - 1);
                const int row = piece / ppr, c16 = piece - row * ppr;
                typedef unsigned v4u __attribute__((ext_vector_type(4)));
                const v4u v = *reinterpret_cast<const v4u*>(src + (size_t)piece * 16);
                __builtin_nontemporal_store(v, reinterpret_cast<v4u*>(dstp + (size_t)row * row_stride + (size_t)c16 * 16));
            }
        };
        for (int d = 0; d < nb && d < np; ++d)
            for_j([&](auto jc) { constexpr int j = decltype(jc)::value; if (valid[j] && fits[j]) issue(jc, d); });
        // plane 0 (no stores in the queue yet) and the last nb planes (no refills) wait for everything; the planes in between with counted waits
        plane(std::integral_constant<int, 0>(), 0);
        int p = 1;
        if (nb == 2) for (; p + 2 < np; ++p) plane(std::integral_constant<int, 2>(), p);
        else         for (; p + 1 < np; ++p) plane(std::integral_constant<int, 1>(), p);
        for (; p < np; ++p) plane(std::integral_constant<int, 0>(), p);
    };
    auto with_nj = [&]<bool SYNC>(std::bool_constant<SYNC> sy) {
        switch (nj) {
        case 1: run(std::integral_constant<int, 1>(), sy); break;
        case 2: run(std::integral_constant<int, 2>(), sy); break;
        case 3: run(std::integral_constant<int, 3>(), sy); break;
        case 4: run(std::integral_constant<int, 4>(), sy); break;
        case 5: run(std::integral_constant<int, 5>(), sy); break;
        default: run(std::integral_constant<int, 6>(), sy); break;
        }
    };
    if (__builtin_amdgcn_readfirstlane((int)sync_mode)) with_nj(std::bool_constant<true>());
    else with_nj(std::bool_constant<false>());
}

// ------------------------------------------------------------------ aux outputs
// xyz[n,:,h,w] = (cos lat sin lon, cos lat cos lon, sin lat) from the UNWRAPPED lon (:13-18,115-118),
// here without further trig:  cos lat * (sin|cos)(l0 + atan2(x,q)) = inv * (..) algebraically.
// uv[b,:,h,a] = strip[h, a*N + b]  (:106-108,120 — the reference reinterprets the [ph, N*pw] strip
// as [ph, pw, N]).
__global__ __launch_bounds__(256) void e2p_aux_kernel(E2PArgs a, float* xyz, float* uv)
{
    const int N = a.tab.N, plane = a.ph * a.pw;
    const int idx = blockIdx.x * 256 + threadIdx.x;
    if (idx >= N * plane) return;
    const int n = idx / plane, e = idx % plane, h = e / a.pw, w = e % a.pw;
    if (xyz) {
        float lon, lat, x, q, t, inv;
        e2p_lonlat(a, n, h, w, lon, lat, x, q, t, inv);
        const float sl = a.tab.slam[n], cl = a.tab.clam[n];
        const bool q4 = lat != lat;                  // quirk q4: NaN latitude poisons the whole ray
        xyz[((size_t)n * 3 + 0) * plane + e] = q4 ? lat : inv * (sl * q + cl * x);
        xyz[((size_t)n * 3 + 1) * plane + e] = q4 ? lat : inv * (cl * q - sl * x);
        xyz[((size_t)n * 3 + 2) * plane + e] = q4 ? lat : fminf(1.0f, fmaxf(-1.0f, t * inv));
    }
    if (uv) {
        const int s = w * N + n;                     // here (n, w) play the roles (b, a)
        const int n2 = s / a.pw, w2 = s % a.pw;
        float lon, lat, x, q, t, inv, u, v;
        e2p_lonlat(a, n2, h, w2, lon, lat, x, q, t, inv);
        e2p_uv(lon, lat, u, v);
        uv[((size_t)n * 2 + 0) * plane + e] = u;
        uv[((size_t)n * 2 + 1) * plane + e] = v;
    }
}

void fill_args(E2PArgs& a, const omni_geometry* g, const void* erp, void* pers, int B, int C)
{
    a.erp = erp; a.pers = pers; a.B = B; a.C = C; a.H = g->H; a.W = g->W; a.ph = g->ph; a.pw = g->pw;
    a.fovx = g->fov_w / 360.0f; a.fovy = g->fov_h / 180.0f;
    a.stepx = g->pw > 1 ? 1.0f / (float)(g->pw - 1) : 0.0f;
    a.stepy = g->ph > 1 ? 1.0f / (float)(g->ph - 1) : 0.0f;
    a.sx_scale = (float)(g->W - 1) / 2.0f; a.sy_scale = (float)(g->H - 1) / 2.0f;
    a.tab = g->e2p;
    a.ixy = g->e2p_ixy;
    a.dbg = 0; a.trace = nullptr;
    a.store_mode = omni_options().e2p_store;
#ifdef OMNI_DEBUG_BUILD
    a.dbg_skip_fb = omni_options().e2p_ref_lds == 2;     // (a RESULT-changing timing experiment: the debug build only, like every OMNI_*_DBG bit)
#else
    a.dbg_skip_fb = 0;
#endif
#ifdef OMNI_DEBUG_BUILD
    a.dbg = omni_debug_bits("OMNI_E2P_DBG");
    a.trace = omni_debug_trace_buf();
#endif
}

}  // namespace

int omni_e2p_build_tileflags(omni_geometry* g, hipStream_t stream)
{
    E2PArgs a; fill_args(a, g, nullptr, nullptr, 1, 1);
    // sampling-coordinate table (8 bytes per patch sample: 9.4 MB at 18 x 256^2), read once per launch instead of two
    // transcendentals per sample and tile
    const long long total = (long long)g->N * g->ph * g->pw;
    if (!g->e2p_ixy && total < (1ll << 28) && !omni_options().e2p_notab) {
        OMNI_HIP(hipMalloc((void**)&g->e2p_ixy, sizeof(float2) * (size_t)total));
        hipLaunchKernelGGL(e2p_ixy_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, stream, a, g->e2p_ixy, (int)total);
        OMNI_HIP(hipGetLastError());
        a.ixy = g->e2p_ixy;
    }
    // tiles whose ERP footprint does not fit the LDS box are listed once per geometry and take the gather fallback
    std::vector<int> list;
    int ts = 32;
    for (;;) {
        const int tx = (g->pw + ts - 1) / ts, ty = (g->ph + ts - 1) / ts;
        const int nt = g->N * tx * ty;
        unsigned char* dflags = nullptr;
        OMNI_HIP(hipMalloc((void**)&dflags, nt));
        if (ts == 32) hipLaunchKernelGGL(e2p_lds_kernel<32>, dim3(nt), dim3(256), 0, stream, a, tx, tx * ty, nt, (const int*)nullptr, dflags);
        else          hipLaunchKernelGGL(e2p_lds_kernel<16>, dim3(nt), dim3(256), 0, stream, a, tx, tx * ty, nt, (const int*)nullptr, dflags);
        OMNI_HIP(hipGetLastError());
        std::vector<unsigned char> hf(nt);
        OMNI_HIP(hipMemcpyAsync(hf.data(), dflags, nt, hipMemcpyDeviceToHost, stream));
        OMNI_HIP(hipStreamSynchronize(stream));
        (void)hipFree(dflags);
        list.clear();
        for (int i = 0; i < nt; ++i) if (hf[i]) list.push_back(i);
        if (omni_options().e2p_verbose) fprintf(stderr, "[omni] equi2pers %dx%d patches on %dx%d, %dx%d tiles: %d of %d take the gather fallback\n",
                                                g->ph, g->pw, g->H, g->W, ts, ts, (int)list.size(), nt);
        // (16x16 tiles — OMNI_E2P_TS=16 — cut the fallback count 3-5x where footprints are large (P = 128 at 512x1024, nrows = 6)
        //  but amortise the per-tile prologue over a quarter of the samples: measured equal or slower, so not selected automatically)
        break;
    }
    g->e2p_ts = ts;
    g->e2p_nfb = (int)list.size();
    if (!list.empty()) {
        OMNI_HIP(hipMalloc((void**)&g->e2p_fb_tiles, sizeof(int) * list.size()));
        OMNI_HIP(hipMemcpy(g->e2p_fb_tiles, list.data(), sizeof(int) * list.size(), hipMemcpyHostToDevice));
    }
    return OMNI_OK;
}

// Per-tile tap boxes of e2p_box_kernel, one table per element size (tile shape and 16-byte chunk alignment differ).  One-time setup.
int omni_e2p_build_boxes(omni_geometry* g, hipStream_t stream)
{
    E2PArgs a; fill_args(a, g, nullptr, nullptr, 1, 1);
    int cap_kb = omni_options().e2p_slot_kb;
    if (cap_kb < 1) cap_kb = 1;
    if (cap_kb > E2B_NJMAX) cap_kb = E2B_NJMAX;
    // tile height: 8 x 32 samples (option e2p_tile_h: 4 = 4 x 32 tiles, 2 = 4 x 32 where more than 1 tile in 8 of the 8-row tiling would take the gather
    // path — a sample spans several ERP pixels: 128^2 patches on 512 x 1024, 256^2 on 1024 x 2048, 512^2 on 2048 x 4096)
    auto build_one = [&](int e, int th) -> int {
        auto& tt = g->e2p_boxes[e];
        if (tt.ent) (void)hipFree(tt.ent);
        if (tt.fb) (void)hipFree(tt.fb);
        if (tt.order) (void)hipFree(tt.order);
        tt.ent = nullptr; tt.fb = nullptr; tt.order = nullptr; tt.norder = 0; tt.nfb = 0; tt.h_fb.clear();
        tt.tw = E2B_TW; tt.th = th;
        tt.ok = 0;
        if (g->pw % tt.tw != 0 || g->ph % tt.th != 0 || g->W < 2) return OMNI_OK;      // whole tiles only (16-byte stores, static store count per stage)
        tt.tx = g->pw / tt.tw; tt.ty = g->ph / tt.th;
        const long long ntiles = (long long)g->N * tt.tx * tt.ty;
        if (ntiles >= (1ll << 24)) return OMNI_OK;
        const int epc = e ? 8 : 4;
        int* dstats = nullptr;
        OMNI_HIP(hipMalloc((void**)&dstats, sizeof(int) * (size_t)(2 + ntiles)));
        if (hipMalloc((void**)&tt.ent, sizeof(uint2) * (size_t)ntiles) != hipSuccess) { (void)hipFree(dstats); OMNI_FAIL(OMNI_ERR_HIP, "omni_e2p_build_boxes: out of memory"); }
        (void)hipMemsetAsync(dstats, 0, 2 * sizeof(int), stream);
        const unsigned nb = (unsigned)((ntiles + 3) / 4);
        if (tt.th == 8) hipLaunchKernelGGL(e2b_tiles_kernel<4>, dim3(nb), dim3(256), 0, stream, a, tt.ent, tt.tx, tt.tx * tt.ty, (int)ntiles, epc, cap_kb * 64, e /* odd pitch: fp16 */, dstats);
        else            hipLaunchKernelGGL(e2b_tiles_kernel<2>, dim3(nb), dim3(256), 0, stream, a, tt.ent, tt.tx, tt.tx * tt.ty, (int)ntiles, epc, cap_kb * 64, e, dstats);
        std::vector<int> hs((size_t)(2 + ntiles));
        if (hipGetLastError() != hipSuccess || hipMemcpyAsync(hs.data(), dstats, sizeof(int) * hs.size(), hipMemcpyDeviceToHost, stream) != hipSuccess ||
            hipStreamSynchronize(stream) != hipSuccess) { (void)hipFree(dstats); OMNI_FAIL(OMNI_ERR_HIP, "omni_e2p_build_boxes: kernel failed"); }
        (void)hipFree(dstats);
        tt.max_chunks = hs[0]; tt.nfb = hs[1];
        {
            // LDS-path tiles grouped by the ERP REGION of their box centre, one region per XCD: 4 longitude sectors x 2 hemispheres
            // (a box is ~40-100 x 10 pixels: few boxes straddle the borders of a 256 x H/2 region, whereas with 8 longitude strips
            // of W/8 columns every second box did and was fetched by two XCDs).  Inside a region by latitude band, then longitude:
            // tiles that run at the same time read neighbouring boxes.
            std::vector<uint2> he((size_t)ntiles);
            OMNI_HIP(hipMemcpy(he.data(), tt.ent, sizeof(uint2) * (size_t)ntiles, hipMemcpyDeviceToHost));
            // (round 3, option e2p_region = 1: 8 LATITUDE BANDS of equal estimated cost instead.  A pole tile reads a few ERP rows over hundreds
            //  of columns; with the tiles of a cap spread over four sector XCDs — and the gather tiles over all eight — every XCD fetched the
            //  polar rows of every plane: FETCH_SIZE 100 MB for the 50-MB input, 91 MB with half as many gather tiles.  A band keeps a cap on one XCD.)
            std::vector<std::vector<std::pair<unsigned, int>>> sec(8);
            std::vector<int> region((size_t)ntiles, 0);
            struct TI { int wid, xc, yc, ymin; bool fit; };
            std::vector<TI> all((size_t)ntiles);
            for (int i = 0; i < (int)ntiles; ++i) {
                const int xs4 = (int)(he[i].y & 0xffff), ymin = (int)(he[i].y >> 16), bw = (int)(he[i].x & 4095) * epc, bh = (int)((he[i].x >> 12) & 4095);
                int xc = xs4 + bw / 2; if (xc >= g->W) xc -= g->W;
                all[i] = {i, xc, ymin + bh / 2, ymin, (he[i].x >> 31) != 0};
                region[i] = ((int)((long long)xc * 4 / g->W) & 3) + 4 * (all[i].yc * 2 >= g->H ? 1 : 0);
            }
            if (omni_options().e2p_region == 1) {
                std::vector<TI> srt = all;
                std::sort(srt.begin(), srt.end(), [](const TI& p, const TI& q) { return p.yc != q.yc ? p.yc < q.yc : p.xc < q.xc; });
                auto cost = [](const TI& t) { return t.fit ? 166ll : 430ll; };       // a streaming tile vs a gather tile (8 blocks of 3 planes), 0.1 us
                long long total = 0, run = 0;
                for (auto& t : srt) total += cost(t);
                for (auto& t : srt) { region[t.wid] = (int)std::min<long long>(7, run * 8 / std::max<long long>(1, total)); run += cost(t); }
            }
            for (int i = 0; i < (int)ntiles; ++i)
                if (all[i].fit) sec[region[i]].push_back({((unsigned)(all[i].ymin / 8) << 16) | (unsigned)all[i].xc, i});
            tt.h_region = region;
            size_t mx = 0;
            for (auto& v : sec) { std::sort(v.begin(), v.end()); mx = v.size() > mx ? v.size() : mx; }
            std::vector<int> ord(mx * 8, -1);
            for (int x = 0; x < 8; ++x) for (size_t i = 0; i < sec[x].size(); ++i) ord[i * 8 + x] = sec[x][i].second;
            tt.norder = (int)ord.size();
            tt.h_ent = he; tt.h_order = ord;
            if (tt.norder > 0) {
                OMNI_HIP(hipMalloc((void**)&tt.order, sizeof(int) * ord.size()));
                OMNI_HIP(hipMemcpy(tt.order, ord.data(), sizeof(int) * ord.size(), hipMemcpyHostToDevice));
            }
        }
        tt.h_fb.assign(hs.begin() + 2, hs.begin() + 2 + tt.nfb);
        if (tt.nfb > 0) {
            OMNI_HIP(hipMalloc((void**)&tt.fb, sizeof(int) * (size_t)tt.nfb));
            OMNI_HIP(hipMemcpy(tt.fb, hs.data() + 2, sizeof(int) * (size_t)tt.nfb, hipMemcpyHostToDevice));
        }
        tt.ok = (tt.max_chunks > 0 || tt.nfb > 0) ? 1 : 0;
        if (omni_options().e2p_verbose)
            fprintf(stderr, "[omni] equi2pers %dx%d patches on %dx%d, %d-byte elements, %dx%d sample tiles: largest staged tap box %d chunks, "
                            "%d of %lld tiles take the gather path (box > %d KiB)\n", g->ph, g->pw, g->H, g->W, 16 / epc, tt.th, tt.tw, tt.max_chunks,
                    tt.nfb, ntiles, cap_kb);
        return OMNI_OK;
    };
    for (int e = 0; e < 2; ++e) {
        const int opt = omni_options().e2p_tile_h;
        int rc = build_one(e, opt == 4 ? 4 : 8);
        if (rc != OMNI_OK) return rc;
        auto& tt = g->e2p_boxes[e];
        // (round 4: 4 x 32 tiles where more than 1 tile in 8 would gather — option e2p_tile_h = 2 — measured: 18 x 128^2 patches at 8 panoramas
        //  24.4 -> 20.9 us, but every single-panorama shape LOSES (cfg 3 26.5 -> 28.8 us, cfg 5 fp16 72.9 -> 95.9: twice the blocks, each with its
        //  set-up, and only 1-3 planes to amortise it over; the gather share only falls from 37 % to 21 %: the boxes are WIDE, not tall) — not the default)
        if (opt == 2 && tt.ok && (long long)tt.nfb * 8 > (long long)g->N * tt.tx * tt.ty && g->ph % 4 == 0) {
            rc = build_one(e, 4);
            if (rc != OMNI_OK) return rc;
        }
    }
    return OMNI_OK;
}

namespace {
// ---- work table of e2p_box_kernel for one plane count: which (tile, plane range) each block processes.
// A block's life is set-up (~2.6 us: table entry -> sampling coordinates -> taps) + np stages; with ONE block per tile, 4372 streaming blocks of
// ~16 us each on the 3072 wave slots of the chip (12 per CU: the LDS ring) are 1.42 rounds — per-block time stamps (tools/trace_resample.py,
// profiles/r03a_trace.txt) show the chip full only from 8 to 24 us of a 36-us launch: the gather blocks of the pole tiles took 60 % of the slots
// for the first 5 us, and the second round ran at 40 % occupancy.  List scheduling, longest first: every XCD gets as many whole tiles as it has
// slots, the remaining tiles are cut into `split` plane ranges (short blocks that fill the slots the long ones free, and end together), and the
// gather blocks — the shortest — come last.  Block b runs on XCD b % 8: column x of the table holds tiles of ERP region x only.
struct E2PWork { uint4* dev = nullptr; int nblocks = 0; };

// LDS one wave of e2p_box_kernel really uses for a geometry whose largest staged box is `max_chunks` 16-byte chunks: NB(NJ) slots of NJ KiB for the
// largest NJ (the ring is addressed slot by slot, so smaller boxes use less).  cfg 1 (boxes to 6 KiB): 12 KiB; cfg 3 / cfg 5 (1-2 KiB): 4 KiB — LDS
// then admits far more waves than the registers (16 per CU)
static size_t e2b_lds_bytes(int max_chunks, int nbmax)
{
    const int njmax = std::max(1, (max_chunks + 63) / 64);
    int kb = 1;
    for (int nj = 1; nj <= std::min(njmax, E2B_NJMAX); ++nj) {
        const int nbr = E2B_RING_KB / nj >= 4 ? 4 : E2B_RING_KB / nj >= 2 ? 2 : 1;
        kb = std::max(kb, std::min(nbr, nbmax) * nj);
    }
    return (size_t)std::max(kb, njmax) * 1024;
}

template <int E>
int e2p_work_table(const omni_geometry* gc, int planes, int nbmax, hipStream_t stream, E2PWork& out)
{
    omni_geometry* g = const_cast<omni_geometry*>(gc);             // (the cache is a mutable part of the handle)
    auto& tt = g->e2p_boxes[E];
    const OmniOptions& o = omni_options();
    // wave slots per CU the schedule plans for: what the kernel's LDS footprint admits (e2b_lds_bytes: 12 KiB at cfg 1 -> 12 or 13), at most the 16
    // that 108 registers per wave admit
    const int slots_lds = (int)std::min<size_t>(16, (160 * 1024) / std::max<size_t>(1024, e2b_lds_bytes(tt.max_chunks, nbmax)));
    const int slots_cu = o.e2p_slots > 0 ? o.e2p_slots : std::min(slots_lds, 12) == 12 ? 12 : slots_lds, split = o.e2p_split > 0 ? o.e2p_split : 3,
              fbp = o.e2p_fb_planes > 0 && o.e2p_fb_planes < 90 ? o.e2p_fb_planes : std::min(tt.norder < slots_cu * omni_num_cus() ? 6 : 12, planes);   // (shorter where the launch is under one round: P = 128)
    // key = everything the table depends on: the plane count and the option set (NOT the channel count: the table holds plane ranges, the
    // kernel splits a plane index into (batch item, channel) itself — B * 3 and 3 B * 1 planes share one table).  Every field whole (ADVICE r3:
    // `e2p_fb_pos & 3` made 4 an alias of 0).
    const long long key = ((long long)planes << 40) | ((long long)(slots_cu & 0xff) << 32) | ((long long)(split & 0xff) << 24) | ((long long)(fbp & 0xff) << 16) |
                          ((long long)(nbmax & 0xf) << 12) | ((long long)(o.e2p_fb_planes >= 98 ? o.e2p_fb_planes - 97 : 0) << 10) | ((long long)(o.e2p_fb_pos & 7) << 7) |
                          ((long long)((o.e2p_full + 1) & 0x7f));
    std::lock_guard<std::mutex> lk(g->work_mu);
    for (auto& w : tt.work) if (w.key == key) { out.dev = w.dev; out.nblocks = w.nblocks; return OMNI_OK; }
    {   // a new plane count under capture cannot be served (an allocation and a synchronous copy would invalidate the capture): like a new
        // geometry, it must have been run once before the capture starts
        hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
        if (stream && hipStreamIsCapturing(stream, &cs) == hipSuccess && cs == hipStreamCaptureStatusActive)
            OMNI_FAIL(OMNI_ERR_UNSUPPORTED, "omni_equi2pers: first use of a plane count (B * C) on this geometry while its stream is being captured (run the shape once before capturing)");
    }
    const int slots_xcd = slots_cu * (omni_num_cus() / 8);
    std::vector<std::vector<uint4>> col(8);
    auto seg = [&](int wid, bool fb, int p0, int np) { return make_uint4((unsigned)wid | (fb ? 0x80000000u : 0u), (unsigned)p0 | ((unsigned)np << 16), tt.h_ent[wid].x, tt.h_ent[wid].y); };
    // streaming tiles, in the region order of the geometry (order[k]: XCD k % 8)
    std::vector<std::vector<int>> tiles(8);
    for (int k = 0; k < tt.norder; ++k) if (tt.h_order[k] >= 0) tiles[k % 8].push_back(tt.h_order[k]);
    const int parts = std::max(1, std::min(split, planes / std::max(1, nbmax)));
    for (int x = 0; x < 8; ++x) {
        const int nt = (int)tiles[x].size();
        // whole tiles while they fill the slots exactly once; the rest in `parts` ranges (all of one range first: blocks that start together run the same planes)
        int nfb_x = 0;
        for (int f = 0; f < tt.nfb; ++f) if ((tt.h_region[tt.h_fb[f]] & 7) == x) nfb_x += (planes + fbp - 1) / fbp;
        const int first = o.e2p_fb_pos == 0 ? std::max(0, slots_xcd - nfb_x) : slots_xcd;   // slots left for whole tiles in round one
        const int nfull = nt <= first ? nt : (o.e2p_full >= 0 ? std::min(nt, first * o.e2p_full / 100) : first);
        auto gathers = [&]() {
            if (o.e2p_fb_planes == 99) return;                         // (tuning: no gather blocks at all — results wrong, timing of the streaming part)
            for (int f = 0; f < tt.nfb; ++f)
                if ((tt.h_region[tt.h_fb[f]] & 7) == x)
                    for (int p0 = 0; p0 < planes; p0 += fbp) col[x].push_back(seg(tt.h_fb[f], true, p0, std::min(fbp, planes - p0)));
        };
        // (the gather blocks go FIRST, 12 planes each: latency-bound blocks about as long as a whole streaming tile, on 1 slot in 13 — as a tail
        //  of 3-plane blocks they cost 5 us: 36.4 -> 33.4 us, 16 panoramas 69 -> 64 us)
        const int fb_pos = o.e2p_fb_pos == 0 ? 3 : o.e2p_fb_pos;   // 3 = first (default): the longest blocks of the launch, beside the whole tiles
        if (fb_pos == 3) gathers();
        if (o.e2p_fb_planes == 98) { gathers(); continue; }            // (tuning: ONLY the gather blocks)
        for (int i = 0; i < nfull; ++i) col[x].push_back(seg(tiles[x][i], false, 0, planes));
        if (fb_pos == 1) gathers();
        for (int q = 0; q < parts; ++q) {
            const int p0 = (int)((long long)planes * q / parts), p1 = (int)((long long)planes * (q + 1) / parts);
            for (int i = nfull; i < nt; ++i) col[x].push_back(seg(tiles[x][i], false, p0, p1 - p0));
            if (fb_pos == 2 && q == 0) gathers();
        }
        if (fb_pos == 4) gathers();
    }
    // (gather tiles: ranges of fbp planes, spread over the XCDs — behind the whole tiles or last, see above)
    size_t mx = 0;
    for (auto& c : col) mx = std::max(mx, c.size());
    std::vector<uint4> tab(mx * 8, make_uint4(0u, 0u, 0u, 0u));
    for (int x = 0; x < 8; ++x) for (size_t i = 0; i < col[x].size(); ++i) tab[i * 8 + x] = col[x][i];
    if (tab.empty()) OMNI_FAIL(OMNI_ERR_INVALID, "omni_equi2pers: empty work table");
    if (tt.work.size() >= 256) OMNI_FAIL(OMNI_ERR_UNSUPPORTED, "omni_equi2pers: more than 256 distinct plane counts / option sets on one geometry handle (omni_geometry_cache_clear() drops them)");
    uint4* dev = nullptr;
    OMNI_HIP(hipMalloc((void**)&dev, sizeof(uint4) * tab.size()));
    if (hipMemcpy(dev, tab.data(), sizeof(uint4) * tab.size(), hipMemcpyHostToDevice) != hipSuccess) { (void)hipFree(dev); OMNI_FAIL(OMNI_ERR_HIP, "omni_equi2pers: work table upload"); }
    // a table lives as long as its geometry handle (omni_geometry.hip frees them with it; a handle a hipGraph holds is pinned and never
    // destroyed): a launch in flight on another stream, or a captured graph, may hold the pointer — never freed here (the FIFO
    // that was here freed tables under running kernels).  A table is 16 B per block, ~100 KB; one per plane count seen.
    tt.work.push_back({key, dev, (int)tab.size()});
    out.dev = dev; out.nblocks = (int)tab.size();
    return OMNI_OK;
}

template <typename T, int NBMAX>
int launch_e2b_nb(const E2PArgs& a, const omni_geometry* g, int B, int C, size_t tensor_bytes, hipStream_t stream)
{
    constexpr int E = sizeof(T) == 2 ? 1 : 0;
    const auto& tt = g->e2p_boxes[E];
    E2PWork wk;
    int rc = e2p_work_table<E>(g, B * C, NBMAX, stream, wk);
    if (rc != OMNI_OK) return rc;
    const int njmax = (tt.max_chunks + 63) / 64;
    const size_t lds = e2b_lds_bytes(tt.max_chunks, NBMAX);
    (void)njmax;
    if (tt.th == 8) hipLaunchKernelGGL((e2p_box_kernel<T, NBMAX, sizeof(T) == 2, 4>), dim3(wk.nblocks), dim3(64), lds, stream, a, (const uint4*)wk.dev, tt.tx, tt.tx * tt.ty,
                                       (unsigned)tensor_bytes);
    else            hipLaunchKernelGGL((e2p_box_kernel<T, NBMAX, sizeof(T) == 2, 2>), dim3(wk.nblocks), dim3(64), lds, stream, a, (const uint4*)wk.dev, tt.tx, tt.tx * tt.ty,
                                       (unsigned)tensor_bytes);
    OMNI_HIP(hipGetLastError());
    return OMNI_OK;
}

template <typename T>
int launch_e2b(const E2PArgs& a, const omni_geometry* g, int B, int C, size_t tensor_bytes, hipStream_t stream)
{
    // stages in flight: 1, 2 or 4 (option e2p_nbuf), at most the plane count (the stage loop runs in groups of NB, the remainder singly)
    const int planes = B * C;
    if (planes >= 65536) OMNI_FAIL(OMNI_ERR_UNSUPPORTED, "omni_equi2pers: B * C must be < 65536");
    int nb = omni_options().e2p_nbuf;
    if (nb <= 0) nb = 2;                               // (4 stages in flight measured slower: 41.5 vs 38.3 us at B = 8, 18 x 256^2)
    if (nb >= 4 && planes >= 4) return launch_e2b_nb<T, 4>(a, g, B, C, tensor_bytes, stream);
    if (nb >= 2 && planes >= 2) return launch_e2b_nb<T, 2>(a, g, B, C, tensor_bytes, stream);
    return launch_e2b_nb<T, 1>(a, g, B, C, tensor_bytes, stream);
}

template <typename T>
int launch_e2p(const omni_geometry* g, const void* erp, void* pers, int B, int C, int layout, hipStream_t stream)
{
    E2PArgs a; fill_args(a, g, erp, pers, B, C);
    const int plane = g->ph * g->pw, N = g->N;
    const bool pair = g->W >= 2;
    if (layout == OMNI_LAYOUT_BNCHW) {
        int spt = 2, unr = 3;
        const auto& bt = g->e2p_boxes[sizeof(T) == 2 ? 1 : 0];
        const long long tensor_bytes = (long long)B * C * g->H * g->W * (long long)sizeof(T);
        if (bt.ok && !omni_options().e2p_gather && tensor_bytes < (1ll << 31) && (uintptr_t)erp % 16 == 0 && (uintptr_t)pers % 16 == 0)
            return launch_e2b<T>(a, g, B, C, (size_t)tensor_bytes, stream);
        if (sizeof(T) == 4 && !omni_options().e2p_gather && g->W >= 2) {
            const int ts = g->e2p_ts;
            const int tx = (g->pw + ts - 1) / ts, ty = (g->ph + ts - 1) / ts;
            const int nt = N * tx * ty;
            int psplit = 1;                                      // plane ranges (tuning hook; splitting repeats the per-tile prologue)
            if (ts == 32) hipLaunchKernelGGL(e2p_lds_kernel<32>, dim3(nt + g->e2p_nfb * B, psplit), dim3(256), 0, stream, a, tx, tx * ty, nt,
                                             (const int*)g->e2p_fb_tiles, (unsigned char*)nullptr);
            else          hipLaunchKernelGGL(e2p_lds_kernel<16>, dim3(nt + g->e2p_nfb * B, psplit), dim3(256), 0, stream, a, tx, tx * ty, nt,
                                             (const int*)g->e2p_fb_tiles, (unsigned char*)nullptr);
            OMNI_HIP(hipGetLastError());
            return OMNI_OK;
        }
        const int per_block = 256 * spt;
        const int bpp = (plane + per_block - 1) / per_block;
        const int nblocks = N * bpp;
#define E2P_LAUNCH(S, U)                                                                                      \
        do { if (pair) hipLaunchKernelGGL((e2p_planar_kernel<T, true, S, U>), dim3(nblocks), dim3(256), 0, stream, a, bpp, nblocks); \
             else      hipLaunchKernelGGL((e2p_planar_kernel<T, false, S, U>), dim3(nblocks), dim3(256), 0, stream, a, bpp, nblocks); } while (0)
        if (spt == 4 && unr == 1) E2P_LAUNCH(4, 1);
        else if (spt == 4 && unr == 3) E2P_LAUNCH(4, 3);
        else if (spt == 2 && unr == 1) E2P_LAUNCH(2, 1);
        else if (spt == 2 && unr == 3) E2P_LAUNCH(2, 3);
        else if (spt == 1 && unr == 3) E2P_LAUNCH(1, 3);
        else if (spt == 1 && unr == 6) E2P_LAUNCH(1, 6);
        else if (spt == 2 && unr == 6) E2P_LAUNCH(2, 6);
        else OMNI_FAIL(OMNI_ERR_INVALID, "bad OMNI_E2P_VAR");
#undef E2P_LAUNCH
    } else if (layout == OMNI_LAYOUT_BCHWN) {
        // the LDS-staged form (e2p_ref_kernel) where its block fits the CU: one wave per PPW patches, a slot per patch, two output tiles
        {
            const auto& bt = g->e2p_boxes[sizeof(T) == 2 ? 1 : 0];
            const long long tensor_bytes = (long long)B * C * g->H * g->W * (long long)sizeof(T);
            const int ppw = N <= 16 ? 1 : 2, nwv = (N + ppw - 1) / ppw;                          // (N <= 20: nrows 3 and 4; larger sets do not fit a CU's LDS)
            const int slot_bytes = std::max(1, (bt.max_chunks + 63) / 64) * 1024;
            const size_t lds = (size_t)nwv * ppw * slot_bytes + 2 * (size_t)8 * 32 * N * sizeof(T);
            const int total_pieces = 8 * 32 * N * (int)sizeof(T) / 16, sst = (total_pieces + nwv * 64 - 1) / (nwv * 64);
            if (bt.ok && bt.th == 8 && omni_options().e2p_ref_lds && !omni_options().e2p_gather && tensor_bytes < (1ll << 31) && (uintptr_t)erp % 16 == 0 &&
                (uintptr_t)pers % 16 == 0 && lds <= 160 * 1024 && sst <= 3 && g->e2p_ixy && nwv <= (ppw == 1 ? 16 : 10)) {
                // one block per tile position and plane range: all planes in one block where the positions alone fill the chip (18 x 256^2: 256),
                // else plane ranges of >= 6 planes until there are about as many blocks as CUs (18 x 128^2: 64 positions)
                const int tiles = bt.tx * bt.ty, planes = B * C;
                int py = std::max(1, std::min(planes / 6, (omni_num_cus() + tiles - 1) / tiles));
                const int ppb = (planes + py - 1) / py;
                py = (planes + ppb - 1) / ppb;
                const dim3 grid((unsigned)tiles, (unsigned)py), block((unsigned)(nwv * 64));
                auto go = [&](auto kern) -> int {
                    if (lds > 64 * 1024) {                              // (more than 64 KiB of dynamic LDS must be asked for, once per kernel and device)
                        static std::mutex mu;
                        static std::vector<std::pair<const void*, int>> done;
                        int dev = 0;
                        OMNI_HIP(hipGetDevice(&dev));
                        std::lock_guard<std::mutex> lk(mu);
                        const std::pair<const void*, int> key(reinterpret_cast<const void*>(kern), dev);
                        if (std::find(done.begin(), done.end(), key) == done.end()) {
                            OMNI_HIP(hipFuncSetAttribute(key.first, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
                            done.push_back(key);
                        }
                    }
                    hipLaunchKernelGGL(kern, grid, block, lds, stream, a, (const uint2*)bt.ent, bt.tx, bt.tx * bt.ty, (unsigned)tensor_bytes, slot_bytes, ppb);
                    OMNI_HIP(hipGetLastError());
                    return OMNI_OK;
                };
#define E2R(P, S_) go(e2p_ref_kernel<T, P, S_>)
                if (ppw == 1) return sst == 1 ? E2R(1, 1) : sst == 2 ? E2R(1, 2) : E2R(1, 3);
                return sst == 1 ? E2R(2, 1) : sst == 2 ? E2R(2, 2) : E2R(2, 3);
#undef E2R
            }
        }
        const int tiles_w = (g->pw + E2P_TW - 1) / E2P_TW;
        const size_t lds = sizeof(float) * E2P_CCH * E2P_TW * N;
        if (pair) hipLaunchKernelGGL((e2p_reflayout_kernel<T, true>), dim3(g->ph * tiles_w), dim3(256), lds, stream, a, tiles_w);
        else      hipLaunchKernelGGL((e2p_reflayout_kernel<T, false>), dim3(g->ph * tiles_w), dim3(256), lds, stream, a, tiles_w);
    } else {
        OMNI_FAIL(OMNI_ERR_UNSUPPORTED, "omni_equi2pers: layout must be OMNI_LAYOUT_BCHWN or OMNI_LAYOUT_BNCHW");
    }
    OMNI_HIP(hipGetLastError());
    return OMNI_OK;
}
}  // namespace

extern "C" int omni_equi2pers_g(const omni_geometry_t* g, const void* erp, void* pers, int dtype,
                                int B, int C, int layout, omni_stream_t stream)
{
    if (!g) OMNI_FAIL(OMNI_ERR_INVALID, "omni_equi2pers: null geometry");
    if (B < 0 || C < 0) OMNI_FAIL(OMNI_ERR_INVALID, "omni_equi2pers: negative batch/channels");
    if (B == 0 || C == 0) return OMNI_OK;                         // empty input: nothing to do
    if (g->H < 1 || g->W < 1) OMNI_FAIL(OMNI_ERR_INVALID, "omni_equi2pers: empty ERP image");
    if (!erp || !pers) OMNI_FAIL(OMNI_ERR_INVALID, "omni_equi2pers: null device pointer");
    if ((size_t)g->H * g->W >= (1u << 31)) OMNI_FAIL(OMNI_ERR_UNSUPPORTED, "omni_equi2pers: ERP plane too large");
    if (dtype == OMNI_F32) return launch_e2p<float>(g, erp, pers, B, C, layout, (hipStream_t)stream);
    if (dtype == OMNI_F16) return launch_e2p<__half>(g, erp, pers, B, C, layout, (hipStream_t)stream);
    OMNI_FAIL(OMNI_ERR_INVALID, "omni_equi2pers: dtype must be OMNI_F32 or OMNI_F16");
}

extern "C" int omni_equi2pers(const void* erp, void* pers, int dtype, int B, int C, int H, int W,
                              int ph, int pw, int nrows, float fov_h, float fov_w, int layout,
                              omni_stream_t stream)
{
    const omni_geometry* g = nullptr;
    // the sampler does not need the ERP-side tables; key them by (H,W) anyway so one handle serves both operators
    int rc = omni_geometry_lookup(&g, nrows, fov_h, fov_w, ph, pw, H, W, (hipStream_t)stream);
    if (rc != OMNI_OK) return rc;
    return omni_equi2pers_g(g, erp, pers, dtype, B, C, layout, stream);
}

extern "C" int omni_equi2pers_aux(float* xyz, float* uv, int ph, int pw, int nrows, float fov_h, float fov_w,
                                  omni_stream_t stream)
{
    const omni_geometry* g = nullptr;
    int rc = omni_geometry_lookup(&g, nrows, fov_h, fov_w, ph, pw, 0, 0, (hipStream_t)stream);
    if (rc != OMNI_OK) return rc;
    if (!xyz && !uv) return OMNI_OK;
    E2PArgs a; fill_args(a, g, nullptr, nullptr, 0, 0);
    const int total = g->N * ph * pw;
    hipLaunchKernelGGL(e2p_aux_kernel, dim3((total + 255) / 256), dim3(256), 0, (hipStream_t)stream, a, xyz, uv);
    OMNI_HIP(hipGetLastError());
    return OMNI_OK;
}

// Vector-Jacobian product of equi2pers w.r.t. the ERP image (the operator is linear in it): grad_pers in the layout of the
// forward's output, grad_erp [B,C,H,W] is overwritten.  fp32 only.  Replaces what autograd derives from F.grid_sample
// (equi2pers_v3.py:111) in the reference's training scripts (train_erp_depth.py:255-300).
// ---- backward by gathers (no global atomics, nothing to zero): the mirror image of p2e_bwd_gather_kernel (omni_pers2equi.hip).  ERP tiles
// are disjoint: one wave owns a 4 x 32 ERP tile, walks — per patch — the box of the samples whose bilinear taps can touch it (a constant
// of the geometry, from the same coordinate table and tap arithmetic: exact superset), adds the taps that land inside its tile into an
// LDS accumulator and writes the tile once.  Taps as in e2p_bwd_kernel (= what autograd derives from F.grid_sample, border padding).
namespace {
constexpr int E2G_TH = 4, E2G_TW = 32;

__global__ __launch_bounds__(256) void e2p_bwd_box_kernel(E2PArgs a, int* __restrict__ boxes, int gtx, int total)
{
    const int s = blockIdx.x * 256 + threadIdx.x;
    if (s >= total) return;
    const int w = s % a.pw, h = (s / a.pw) % a.ph, n = s / (a.pw * a.ph);
    const float2 c = a.ixy[s];
    if (!(c.x == c.x) || !(c.y == c.y)) return;
    const int x0 = (int)floorf(c.x), y0 = (int)floorf(c.y);
    const int x1 = x0 + 1 < a.W ? x0 + 1 : x0, y1 = y0 + 1 < a.H ? y0 + 1 : y0;
    const int xs[2] = {x0, x1}, ys[2] = {y0, y1};
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        if (k == 1 && x1 == x0) continue;
        if (k == 2 && y1 == y0) continue;
        if (k == 3 && (x1 == x0 || y1 == y0)) continue;
        int* b = boxes + 4 * ((size_t)((ys[k >> 1] / E2G_TH) * gtx + xs[k & 1] / E2G_TW) * a.tab.N + n);
        atomicMin(b + 0, h); atomicMax(b + 1, h); atomicMin(b + 2, w); atomicMax(b + 3, w);
    }
}

// The transpose as a sparse matrix (omni_spgather.h): every tap of every patch sample is one entry (source = the sample, packed
// patch << 24 | h * pw + w; weight = the bilinear weight) of the row of the ERP pixel it reads.  Taps as in e2p_bwd_kernel.
__global__ __launch_bounds__(256) void e2p_sp_walk_kernel(E2PArgs a, int total, SpEmit b)
{
    const int s = blockIdx.x * 256 + threadIdx.x;
    if (s >= total) return;
    const int pp = a.ph * a.pw, n = s / pp;
    const unsigned src = ((unsigned)n << 24) | (unsigned)(s - n * pp);
    const float2 c = a.ixy[s];
    if (!(c.x == c.x) || !(c.y == c.y)) return;                   // q4: an odd x odd patch has a NaN centre sample
    const float fx = floorf(c.x), fy = floorf(c.y);
    const int x0 = (int)fx, y0 = (int)fy;
    const float tx = c.x - fx, ty = c.y - fy, ex = 1.0f - tx, ey = 1.0f - ty;
    const bool okx = x0 + 1 < a.W, oky = y0 + 1 < a.H;
    const int row = y0 * a.W + x0;
    sp_emit(b, row, src, ey * ex);
    if (okx) sp_emit(b, row + 1, src, ey * tx);
    if (oky) sp_emit(b, row + a.W, src, ty * ex);
    if (okx && oky) sp_emit(b, row + a.W + 1, src, ty * tx);
}

template <int PL, int NT>
__global__ __launch_bounds__(NT) void e2p_bwd_gather_kernel(E2PArgs a /* erp = g_erp (out), pers = g_pers (in) */, const int4* __restrict__ boxes,
                                                            const int* __restrict__ ids, int gtx, int planes, int n_fastest)
{
    __shared__ float acc[PL][E2G_TH * E2G_TW];
    const int lane = threadIdx.x;
    const int id = ids[blockIdx.x], p0 = blockIdx.y * PL;
    const int ty0 = (id / gtx) * E2G_TH, tx0 = (id % gtx) * E2G_TW;
#pragma unroll
    for (int p = 0; p < PL; ++p)
        for (int e = lane; e < E2G_TH * E2G_TW; e += NT) acc[p][e] = 0.0f;
    if (NT > 64) __syncthreads();
    const float* gp = (const float*)a.pers;
    const size_t pp = (size_t)a.ph * a.pw;
    for (int n = 0; n < a.tab.N; ++n) {
        const int4 box = boxes[(size_t)id * a.tab.N + n];          // sample rows min, max, columns min, max
        if (box.x > box.y) continue;                               // (wave-uniform)
        const int bw = box.w - box.z + 1, npx = bw * (box.y - box.x + 1);
        const float rbw = 1.0f / (float)bw;
        for (int base = 0; base < npx; base += NT) {
            const int idx = base + lane;
            if (idx >= npx) continue;
            int dy = (int)(((float)idx + 0.5f) * rbw);
            int dxi = idx - dy * bw;
            if (dxi < 0) { --dy; dxi += bw; } else if (dxi >= bw) { ++dy; dxi -= bw; }
            const int h = box.x + dy, w = box.z + dxi;
            const float2 c = a.ixy[((size_t)n * a.ph + h) * a.pw + w];
            if (!(c.x == c.x) || !(c.y == c.y)) continue;
            const float fx = floorf(c.x), fy = floorf(c.y);
            const int x0 = (int)fx, y0 = (int)fy;
            const float tx = c.x - fx, ty = c.y - fy, ex = 1.0f - tx, ey = 1.0f - ty;
            const bool okx = x0 + 1 < a.W, oky = y0 + 1 < a.H;
            const int xa = x0 - tx0, xb = xa + 1, ya = y0 - ty0, yb = ya + 1;
            const bool xa_in = (unsigned)xa < (unsigned)E2G_TW, xb_in = okx && (unsigned)xb < (unsigned)E2G_TW;
            const bool ya_in = (unsigned)ya < (unsigned)E2G_TH, yb_in = oky && (unsigned)yb < (unsigned)E2G_TH;
            const float w00 = (ya_in && xa_in) ? ey * ex : 0.0f, w01 = (ya_in && xb_in) ? ey * tx : 0.0f;
            const float w10 = (yb_in && xa_in) ? ty * ex : 0.0f, w11 = (yb_in && xb_in) ? ty * tx : 0.0f;
            if (!((ya_in || yb_in) && (xa_in || xb_in))) continue;
#pragma unroll
            for (int p = 0; p < PL; ++p) {
                if (p0 + p >= planes) break;
                const int b = (p0 + p) / a.C, ch = (p0 + p) % a.C;
                const size_t src = n_fastest ? ((((size_t)b * a.C + ch) * a.ph + h) * a.pw + w) * a.tab.N + n
                                             : (((size_t)b * a.tab.N + n) * a.C + ch) * pp + (size_t)h * a.pw + w;
                const float g = gp[src];
                // (a zero weight of a tap INSIDE the tile must still be added as 0 x g only if g is finite: skip instead, like a tap outside)
                if (ya_in && xa_in) atomicAdd(&acc[p][ya * E2G_TW + xa], g * w00);
                if (ya_in && xb_in) atomicAdd(&acc[p][ya * E2G_TW + xb], g * w01);
                if (yb_in && xa_in) atomicAdd(&acc[p][yb * E2G_TW + xa], g * w10);
                if (yb_in && xb_in) atomicAdd(&acc[p][yb * E2G_TW + xb], g * w11);
            }
        }
    }
    __syncthreads();
    float* gerp = (float*)const_cast<void*>(a.erp);
    const size_t plane = (size_t)a.H * a.W;
#pragma unroll
    for (int p = 0; p < PL; ++p) {
        if (p0 + p >= planes) break;
        for (int e = lane; e < E2G_TH * E2G_TW; e += NT) {
            const int y = ty0 + e / E2G_TW, x = tx0 + e % E2G_TW;
            if (y < a.H && x < a.W) gerp[(size_t)(p0 + p) * plane + (size_t)y * a.W + x] = acc[p][e];
        }
    }
}
}  // namespace

int omni_e2p_build_bwd(omni_geometry* g, hipStream_t stream)
{
    if (!g->e2p_ixy) return OMNI_OK;                               // no coordinate table: the scatter kernels serve this geometry
    E2PArgs a; fill_args(a, g, nullptr, nullptr, 1, 1);
    g->e2p_gtx = (g->W + E2G_TW - 1) / E2G_TW; g->e2p_gty = (g->H + E2G_TH - 1) / E2G_TH;
    const size_t ntiles = (size_t)g->e2p_gtx * g->e2p_gty, nbox = ntiles * g->N;
    const long long total = (long long)g->N * g->ph * g->pw;
    if (ntiles == 0 || nbox >= (1u << 28) || total >= (1ll << 31)) return OMNI_OK;
    OMNI_HIP(hipMalloc((void**)&g->e2p_bwd_box, sizeof(int4) * nbox));
    std::vector<int4> hb(nbox, make_int4(0x7fffffff, -0x7fffffff, 0x7fffffff, -0x7fffffff));
    OMNI_HIP(hipMemcpy(g->e2p_bwd_box, hb.data(), sizeof(int4) * nbox, hipMemcpyHostToDevice));
    hipLaunchKernelGGL(e2p_bwd_box_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, stream, a, (int*)g->e2p_bwd_box, g->e2p_gtx, (int)total);
    OMNI_HIP(hipGetLastError());
    OMNI_HIP(hipStreamSynchronize(stream));
    OMNI_HIP(hipMemcpy(hb.data(), g->e2p_bwd_box, sizeof(int4) * nbox, hipMemcpyDeviceToHost));
    std::vector<int> small, big;
    long long ps = 0, pb = 0, mx = 0;
    for (size_t t = 0; t < ntiles; ++t) {
        long long npx = 0;
        for (int n = 0; n < g->N; ++n) {
            const int4 b = hb[t * g->N + n];
            if (b.x <= b.y) npx += (long long)(b.y - b.x + 1) * (b.w - b.z + 1);
        }
        (npx <= 4096 ? small : big).push_back((int)t);
        (npx <= 4096 ? ps : pb) += npx; mx = npx > mx ? npx : mx;
    }
    if (omni_options().e2p_verbose)
        fprintf(stderr, "[omni] equi2pers backward boxes (%dx%d ERP, %d patches %dx%d): %zu tiles, %zu big; box samples small %lld big %lld, largest %lld\n",
                g->H, g->W, g->N, g->ph, g->pw, ntiles, big.size(), ps, pb, mx);
    g->e2p_bwd_nsmall = (int)small.size(); g->e2p_bwd_nbig = (int)big.size();
    small.insert(small.end(), big.begin(), big.end());
    OMNI_HIP(hipMalloc((void**)&g->e2p_bwd_ids, sizeof(int) * ntiles));
    OMNI_HIP(hipMemcpy(g->e2p_bwd_ids, small.data(), sizeof(int) * ntiles, hipMemcpyHostToDevice));
    g->e2p_bwd_ok = 1;
    // the sparse-matrix form (the default): rows = ERP pixels, sources = patch samples (patch in the high 8 bits, sample in the low 24)
    if ((long long)g->H * g->W < (1ll << 31) && (long long)g->ph * g->pw <= (1ll << 24) && g->N < 256) {
        SpBuilder sb;
        int rc = sb.begin(&g->e2p_sp, g->H * g->W, stream);
        if (rc != OMNI_OK) return rc;
        hipLaunchKernelGGL(e2p_sp_walk_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, stream, a, (int)total, sb.emit(0));
        OMNI_HIP(hipGetLastError());
        OMNI_HIP(hipStreamSynchronize(stream));
        bool fits = false;
        rc = sb.layout((size_t)omni_options().bwd_table_mb << 20, &fits, stream);
        if (rc != OMNI_OK) return rc;
        if (fits) {
            hipLaunchKernelGGL(e2p_sp_walk_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, stream, a, (int)total, sb.emit(1));
            OMNI_HIP(hipGetLastError());
            OMNI_HIP(hipStreamSynchronize(stream));
            rc = sb.finish(stream);
            if (rc != OMNI_OK) return rc;
        } else omni_sp_free(g->e2p_sp);
        if (omni_options().e2p_verbose)
            fprintf(stderr, "[omni] equi2pers backward as a sparse matrix: %d rows, %lld entries (%lld with padding) + %d long rows with %lld entries%s\n",
                    g->e2p_sp.nrows, g->e2p_sp.nent, g->e2p_sp.npadded, g->e2p_sp.nlong, g->e2p_sp.nlong_ent, fits ? "" : " -> over the table budget, not kept");
    }
    return OMNI_OK;
}

extern "C" int omni_equi2pers_bwd(const void* grad_pers, void* grad_erp, int dtype, int B, int C, int H, int W,
                                  int ph, int pw, int nrows, float fov_h, float fov_w, int layout, omni_stream_t stream)
{
    if (dtype != OMNI_F32) OMNI_FAIL(OMNI_ERR_UNSUPPORTED, "omni_equi2pers_bwd: fp32 only");
    if (layout != OMNI_LAYOUT_BCHWN && layout != OMNI_LAYOUT_BNCHW) OMNI_FAIL(OMNI_ERR_INVALID, "omni_equi2pers_bwd: layout must be BCHWN or BNCHW");
    const omni_geometry* g = nullptr;
    int rc = omni_geometry_lookup(&g, nrows, fov_h, fov_w, ph, pw, H, W, (hipStream_t)stream);
    if (rc != OMNI_OK) return rc;
    if (B < 0 || C < 0) OMNI_FAIL(OMNI_ERR_INVALID, "omni_equi2pers_bwd: negative batch/channels");
    if (B == 0 || C == 0) return OMNI_OK;
    if (!grad_pers || !grad_erp) OMNI_FAIL(OMNI_ERR_INVALID, "omni_equi2pers_bwd: null device pointer");
    E2PArgs a; fill_args(a, g, grad_erp, const_cast<void*>(grad_pers), B, C);
    {   // first backward of this geometry: build its tables (synchronises the stream once)
        omni_geometry* gm = const_cast<omni_geometry*>(g);
        std::lock_guard<std::mutex> lk(gm->bwd_mu);
        if (!gm->e2p_bwd_tried) {
            gm->e2p_bwd_tried = 1;
            rc = omni_e2p_build_bwd(gm, (hipStream_t)stream);
            if (rc != OMNI_OK) return rc;
        }
    }
    const int mode = omni_options().e2p_bwd_simple;
    if (g->e2p_sp.ok && (mode == 0 || mode == 4)) {               // the sparse-matrix gather: no atomics, nothing to zero
        SpApply s;
        const long long pp = (long long)ph * pw;
        s.src = (const float*)grad_pers; s.dst = (float*)grad_erp; s.C = C; s.planes = B * C;
        if (layout == OMNI_LAYOUT_BNCHW) { s.s_sB = (long long)g->N * C * pp; s.s_sC = pp; s.s_hi = (int)(C * pp); s.s_lo = 1; }
        else                             { s.s_sB = (long long)C * pp * g->N; s.s_sC = pp * g->N; s.s_hi = 1; s.s_lo = g->N; }
        s.d_sB = (long long)C * H * W; s.d_sC = (long long)H * W; s.rdiv = 0x7fffffff; s.d_hi = 0; s.d_lo = 1;
        s.PT = (B * C + 3) / 4 * 4; s.nhi = g->N; s.nlo = (int)pp; s.hi_fastest = layout == OMNI_LAYOUT_BCHWN; s.chunk = 16;
        if ((long long)g->N * C * pp < (1ll << 31)) {
            float* ws = nullptr;
            if (omni_options().bwd_wide) {
                rc = omni_bwd_workspace(const_cast<omni_geometry*>(g), (hipStream_t)stream, (size_t)g->N * pp * s.PT * sizeof(float), &ws);
                if (rc != OMNI_OK) return rc;
            }
            return sp_apply(g->e2p_sp, s, (hipStream_t)stream, ws);
        }
    }
    // without the table, mode 0: whichever is faster for the layout — measured at B = 8, cfg 1: planar 0.74 ms (LDS boxes + coalesced global atomics) vs
    // 0.88 ms (gathers); reference layout 0.88 ms (gathers) vs 3.17 ms (plain scatter).  3 forces the gathers, 1 the plain scatter, 2 the LDS boxes.
    const bool planar_boxes = layout == OMNI_LAYOUT_BNCHW && g->W >= 2;
    if (g->e2p_bwd_ok && (mode == 3 || (mode == 0 && !planar_boxes))) {
        constexpr int PL = 4;
        const int groups = (B * C + PL - 1) / PL, nf = layout == OMNI_LAYOUT_BCHWN ? 1 : 0;
        if (g->e2p_bwd_nbig)
            hipLaunchKernelGGL((e2p_bwd_gather_kernel<PL, 1024>), dim3(g->e2p_bwd_nbig, groups), dim3(1024), 0, (hipStream_t)stream, a,
                               (const int4*)g->e2p_bwd_box, (const int*)g->e2p_bwd_ids + g->e2p_bwd_nsmall, g->e2p_gtx, B * C, nf);
        if (g->e2p_bwd_nsmall)
            hipLaunchKernelGGL((e2p_bwd_gather_kernel<PL, 64>), dim3(g->e2p_bwd_nsmall, groups), dim3(64), 0, (hipStream_t)stream, a,
                               (const int4*)g->e2p_bwd_box, (const int*)g->e2p_bwd_ids, g->e2p_gtx, B * C, nf);
        OMNI_HIP(hipGetLastError());
        return OMNI_OK;
    }
    OMNI_HIP(hipMemsetAsync(grad_erp, 0, (size_t)B * C * H * W * sizeof(float), (hipStream_t)stream));
    const long long total = (long long)g->N * ph * pw;
    if (total >= (1ll << 31)) OMNI_FAIL(OMNI_ERR_UNSUPPORTED, "omni_equi2pers_bwd: too many patch samples");
    if (planar_boxes && mode != 1) {
        // planar layout: the transposed LDS-box kernel (same tiling and fallback list as the forward)
        const int ts = g->e2p_ts, tx = (g->pw + ts - 1) / ts, ty = (g->ph + ts - 1) / ts, nt = g->N * tx * ty;
        if (ts == 32) hipLaunchKernelGGL((e2p_lds_kernel<32, true>), dim3(nt + g->e2p_nfb * B), dim3(256), 0, (hipStream_t)stream, a, tx, tx * ty, nt,
                                         (const int*)g->e2p_fb_tiles, (unsigned char*)nullptr);
        else          hipLaunchKernelGGL((e2p_lds_kernel<16, true>), dim3(nt + g->e2p_nfb * B), dim3(256), 0, (hipStream_t)stream, a, tx, tx * ty, nt,
                                         (const int*)g->e2p_fb_tiles, (unsigned char*)nullptr);
        OMNI_HIP(hipGetLastError());
        return OMNI_OK;
    }
    hipLaunchKernelGGL(e2p_bwd_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, a,
                       layout == OMNI_LAYOUT_BCHWN ? 1 : 0, (int)total);
    OMNI_HIP(hipGetLastError());
    return OMNI_OK;
}
