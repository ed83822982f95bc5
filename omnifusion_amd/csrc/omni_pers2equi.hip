// omni_pers2equi.hip — N tangent-plane patches -> one ERP map (gfx950).
//
// Replaces /root/reference/equi_pers/pers2equi_v3.py:16-198:
//   K6  [N,H,W] int64 tap tables + [N,H,W,4] weights, torch.save / torch.load of a 0.5 GB file
//       per call (:29-167)                          -> evaluated in-kernel per (pixel, patch)
//   K7  4 dense advanced-index gathers [B,C,N,H,W] (:174-177) -> gathers over COVERING patches only
//   K8  mask multiply, permutes (:179-187)          -> fused
//   K9  threshold 1e-5, L1-normalise over N*4 (:189-193) -> running sum of weights
//   K10 weighted sum (:194-196)                     -> fused
//   K11 confidence fusion, spherical_model.py:307-311 -> p2e kernel with CONF = true
//
// One thread owns one ERP pixel and all B*C planes of it; a wave owns 64 consecutive pixels
// of one ERP row.  Which patches can cover a 64-pixel tile is a constant of the geometry: a
// 64-bit mask per tile (H * ceil(W/64) words, built once by p2e_candidates_kernel with the SAME
// device function the blend uses, so the mask is an exact superset by construction).  The blend
// loops over the set bits only (wave-uniform scalar loop; mean 2.1 of 18 patches for nrows=4,
// 4.8 of 46 for nrows=6) instead of the reference's dense N.
//
// Per (pixel, patch) arithmetic follows pers2equi_v3.py:112-152 exactly in structure —
// including the quirks that define parity (X scaled by height and compared with
// width, clamped x1/y1 in the weights, strict 0<X<P mask, threshold 1e-5, L1 over all taps) —
// but takes sin/cos of lat_i, lon_j and the centre angles from tables evaluated in double on the
// host:  cos(lon-l0) = cos lon cos l0 + sin lon sin l0  etc.  No transcendental in the kernel.
//
// HBM-bound: algorithmic bytes B*C*(ph*pw*N + H*W)*sizeof(T); tables read: 8 B per tile.
#include <stdio.h>
#include <string.h>
#include <utility>
#include <vector>
#include <algorithm>
#include "omni_internal.h"
#include "omni_spgather.h"

namespace {

struct P2EArgs {
    const void* pers; const void* pers2;      // pers2: confidence tensor for the fused K11 blend
    void* erp;
    const float2* row_trig; const float2* col_trig;
    const unsigned long long* cand;
    int B, C, H, W, ph, pw, ntx;
    long long sB, sC, sN, sY, sX;              // element strides of the patch tensor
    float kx, ky;                              // 1/(FOVx*PI), 1/(FOVy*PI_2)   (:115-116)
    float half_h, half_w;                      // 0.5*height, 0.5*width        (:122-123)
    int store_nt;                              // 1: non-temporal ERP stores (option p2e_store)
    int dbg;                                   // debug build only (OMNI_P2E_DBG ablation bits): 1 no tap geometry, 2 no LDS tap reads, 4 no DMA, 8 no stores
    long long* trace;                          // debug build, bit 16: per-block time stamps (omni_debug_set_trace)
    PatchTab tab;
};

struct Taps { int x0, x1, y0, y1; float wa, wb, wc, wd; };


// pers2equi_v3.py:112-152 + :191 for one (pixel, patch).  Returns the validity mask.
// Split in two so that the pixels of one ERP column (same lon) share cos/sin(lon - l0); every kernel (candidate masks, tile
// boxes, gather blend, LDS blend) goes through these SAME two functions, so all of them see the same bits.
__device__ __forceinline__ void p2e_lon(const P2EArgs& a, int n, float slon, float clon, float& cd, float& sd)
{
    const float sl0 = a.tab.slam[n], cl0 = a.tab.clam[n];
    cd = clon * cl0 + slon * sl0;                                   // cos(lon - l0)
    sd = slon * cl0 - clon * sl0;                                   // sin(lon - l0)
}
// the float part of the taps (everything but the integer conversions): the four weights and the tap coordinates as floats
struct TapsF { float x0f, x1f, y0f, y1f; float wa, wb, wc, wd; };
__device__ __forceinline__ bool p2e_taps_f(const P2EArgs& a, float sp, float cp, float slat, float clat, float cd, float sd, TapsF& t)
{
    const float cos_c = sp * slat + cp * clat * cd;                 // :112
    // :113-114 divide twice by cos_c; one reciprocal and two products differ from that by <= 2 ulp of X, Y (a
    // validity / floor predicate can flip only where the reference's own coordinate is within round-off of the step)
    float rc = __builtin_amdgcn_rcpf(cos_c);                       // 1 ulp ...
    rc = fmaf(fmaf(-cos_c, rc, 1.0f), rc, rc);                      // ... + one Newton step: ~0.5 ulp (an IEEE division costs 11 instructions)
    float nx = (clat * sd) * rc;                                    // :113
    float ny = (cp * slat - sp * clat * cd) * rc;                   // :114
    nx = nx * a.kx;                                                 // :115
    ny = ny * a.ky;                                                 // :116
    const float X = (nx + 1.0f) * a.half_h;                         // :122 (sic)
    const float Y = (ny + 1.0f) * a.half_w;                         // :123 (sic)
    const float fw = (float)a.pw, fh = (float)a.ph;
    const bool valid = (X < fw) && (X > 0.0f) && (Y < fh) && (Y > 0.0f) && (cos_c > 0.0f);   // :118,126-127
    const float fx = floorf(X), fy = floorf(Y);                     // :129-132
    // :134-137 clamp x0, x1, y0, y1 to [0, P-1].  A VALID pixel has 0 < X < P, so floor(X) is already in range and only the +1
    // taps can leave it (at the far edge); for an invalid pixel every weight is zeroed below and no kernel uses its indices.
    const float x0f = fx, x1f = fminf(fx + 1.0f, fw - 1.0f);
    const float y0f = fy, y1f = fminf(fy + 1.0f, fh - 1.0f);
    // :144-147 multiply by mask: the mask goes onto the two x factors (two selects instead of four; for a valid pixel the products are the
    // reference's, for an invalid one they are 0, -0 or — where Y is not finite — NaN, and the threshold below turns all three into 0)
    const float hx1 = valid ? x1f - X : 0.0f, hx0 = valid ? X - x0f : 0.0f;
    const float wa = hx1 * (y1f - Y);                               // :139  tap (y0,x0)
    const float wb = hx1 * (Y - y0f);                               // :140  tap (y1,x0)
    const float wc = hx0 * (y1f - Y);                               // :141  tap (y0,x1)
    const float wd = hx0 * (Y - y0f);                               // :142  tap (y1,x1)
    // :191 zero everything <= 1e-5 (a NaN compares false)
    t.wa = wa > 1e-5f ? wa : 0.0f;
    t.wb = wb > 1e-5f ? wb : 0.0f;
    t.wc = wc > 1e-5f ? wc : 0.0f;
    t.wd = wd > 1e-5f ? wd : 0.0f;
    // Right patch edge (x1 == x0 == pw-1, X in [pw-1, pw)): the x0 taps carry the factor (x1 - X) <= 0, so wa and wb are already
    // exactly 0 — except in the corner cell, where y is clamped too and wa = (x1-X)(y1-Y) > 0.  There all four taps are the same
    // pixel; its weight is moved to the (y1, x1) tap (v*wa + v*wd -> v*(wa + wd): one rounding), so that EVERY kernel may assume
    // "x1 == x0  =>  wa == wb == 0" and read the tap pair one column to the left without a select.
    const bool xedge = x1f == x0f;
    t.wd = xedge ? t.wd + t.wa : t.wd;
    t.wa = xedge ? 0.0f : t.wa;
    t.x0f = x0f; t.x1f = x1f; t.y0f = y0f; t.y1f = y1f;
    return valid;
}
__device__ __forceinline__ bool p2e_taps_core(const P2EArgs& a, float sp, float cp, float slat, float clat, float cd, float sd, Taps& t)
{
    TapsF f;
    const bool valid = p2e_taps_f(a, sp, cp, slat, clat, cd, sd, f);
    t.wa = f.wa; t.wb = f.wb; t.wc = f.wc; t.wd = f.wd;
    t.x0 = (int)f.x0f; t.x1 = (int)f.x1f; t.y0 = (int)f.y0f; t.y1 = (int)f.y1f;
    return valid;
}
// The taps as the LDS kernels use them: element offsets of the two tap ROW pairs inside a box whose origin is (xa, ymin) and whose rows are `pitch`
// elements apart — the pair (x1 - 1, x1) of rows y0 and y1 (at the right patch edge, x1 == x0, wa == wb == 0 and the pair's second element is the x1
// tap: no select) — and the weights; a pixel the patch does not cover (all weights 0) reads the box origin.  Returns the weight sum.
// (xo = x0 - (xa + 1 - (x1 - x0)) = x1 - xa - 1; y1 - y0 is 0 or 1: one 24-bit multiply-add and one select instead of two 32-bit multiplies.)
__device__ __forceinline__ float p2e_taps_box(const P2EArgs& a, float sp, float cp, float slat, float clat, float cd, float sd, int xa1, int ymin, int pitch,
                                              int& r0, int& r1, float& wa, float& wb, float& wc, float& wd)
{
    TapsF f;
    p2e_taps_f(a, sp, cp, slat, clat, cd, sd, f);
    const float wsum = (f.wa + f.wb) + (f.wc + f.wd);               // all >= 0 after the threshold
    const bool used = wsum > 0.0f;
    const int o0 = __mul24((int)f.y0f - ymin, pitch) + ((int)f.x1f - xa1);
    r0 = used ? o0 : 0;
    r1 = used ? o0 + (f.y1f != f.y0f ? pitch : 0) : 0;
    wa = f.wa; wb = f.wb; wc = f.wc; wd = f.wd;
    return wsum;
}
__device__ __forceinline__ bool p2e_taps_cs(const P2EArgs& a, int n, float slat, float clat, float cd, float sd, Taps& t)
{
    return p2e_taps_core(a, a.tab.sphi[n], a.tab.cphi[n], slat, clat, cd, sd, t);
}
__device__ __forceinline__ bool p2e_taps(const P2EArgs& a, int n, float slat, float clat, float slon, float clon, Taps& t)
{
    float cd, sd;
    p2e_lon(a, n, slon, clon, cd, sd);
    return p2e_taps_cs(a, n, slat, clat, cd, sd, t);
}

// One wave per 64-pixel tile: bit n of cand[row][tile] = any lane valid for patch n.
__global__ __launch_bounds__(256) void p2e_candidates_kernel(P2EArgs a, unsigned long long* cand)
{
    const int wave = (blockIdx.x * 256 + threadIdx.x) >> 6, lane = threadIdx.x & 63;
    if (wave >= a.H * a.ntx) return;
    const int i = wave / a.ntx, j = (wave % a.ntx) * 64 + lane;
    const float2 rt = a.row_trig[i];
    const float2 ct = a.col_trig[min(j, a.W - 1)];
    unsigned long long m = 0;
    for (int n = 0; n < a.tab.N; ++n) {
        Taps t;
        const bool v = p2e_taps(a, n, rt.x, rt.y, ct.x, ct.y, t) && (j < a.W);
        if (__ballot(v) != 0ull) m |= (1ull << n);
    }
    if (lane == 0) cand[wave] = m;
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

// four taps of one plane.  XS1: the patch rows are unit-stride in x (planar layout), so the two
// x-taps of a row come from ONE 8-byte load of (xb, xb+1), xb = min(x0, pw-2); when x0 == pw-1 the
// reference's clamped x1 equals x0 and both taps read element xb+1 (sel = 1).
template <typename T, bool XS1>
__device__ __forceinline__ float p2e_fetch(const T* __restrict__ q, unsigned o0, unsigned o1, unsigned dx, int sel,
                                           const Taps& t)
{
    float va, vb, vc, vd;
    if (XS1) {
        float ax, ay, bx, by;
        Pair<T>::ld(q + o0, ax, ay);
        Pair<T>::ld(q + o1, bx, by);
        va = sel ? ay : ax; vc = ay; vb = sel ? by : bx; vd = by;
    } else {
        va = Store<T>::ld(q + o0); vc = Store<T>::ld(q + o0 + dx);
        vb = Store<T>::ld(q + o1); vd = Store<T>::ld(q + o1 + dx);
    }
    return fmaf(vd, t.wd, fmaf(vc, t.wc, fmaf(vb, t.wb, va * t.wa)));
}

// Blend.  PL = planes accumulated per pass (B*C are walked in chunks of PL).
template <typename T, int PL, bool CONF, bool XS1>
__global__ __launch_bounds__(256) void p2e_kernel(P2EArgs a, int tiles_per_row4, int nblocks)
{
    // block = 4 waves = 4 consecutive rows x 64 columns (vertical neighbours share gather lines in L1)
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    int r4, tx;
    if (!omni_xcd_rows(blockIdx.x, tiles_per_row4, a.ntx, r4, tx)) return;        // tiles_per_row4 = groups of 4 rows
    const int i = __builtin_amdgcn_readfirstlane(r4 * 4 + wave);                   // wave-uniform row
    const int j = tx * 64 + lane;
    if (i >= a.H) return;
    const bool inside = j < a.W;
    const float2 rt = a.row_trig[i];
    const float2 ct = a.col_trig[inside ? j : a.W - 1];
    const unsigned long long cm_ = a.cand[(size_t)i * a.ntx + tx];
    // keep the candidate mask in SGPRs: the patch loop below is then a scalar loop and the
    // per-patch constants come in through scalar loads from the kernarg segment
    // (readfirstlane returns a signed int: go through unsigned or bit 31 sign-extends into bits 32..63)
    const unsigned cm_hi = (unsigned)__builtin_amdgcn_readfirstlane((int)(cm_ >> 32));
    const unsigned cm_lo = (unsigned)__builtin_amdgcn_readfirstlane((int)(cm_ & 0xffffffffull));
    const unsigned long long cmask = ((unsigned long long)cm_hi << 32) | (unsigned long long)cm_lo;
    const T* pers = (const T*)a.pers;
    const T* pers2 = (const T*)a.pers2;
    const int planes = CONF ? a.B : a.B * a.C;
    const size_t erp_plane = (size_t)a.H * a.W;
    const size_t pix = (size_t)i * a.W + j;
    const unsigned sX = (unsigned)a.sX, sY = (unsigned)a.sY, sN = (unsigned)a.sN;   // < 2^31 (host-checked)

    for (int p0 = 0; p0 < planes; p0 += PL) {
        float acc[PL], acc2[CONF ? PL : 1];
#pragma unroll
        for (int k = 0; k < PL; ++k) acc[k] = 0.0f;
        if (CONF) {
#pragma unroll
            for (int k = 0; k < PL; ++k) acc2[k] = 0.0f;
        }
        float l1 = 0.0f;
        unsigned long long m = cmask;
        // wave-uniform loop over candidate patches, TWO per trip: both patches' tap loads are in flight before the first
        // is consumed (the blend is a chain of dependent memory round trips: candidate mask -> patch constants -> taps)
        while (m) {
            const int n0 = __builtin_ctzll(m);
            m &= m - 1;
            const bool two = m != 0ull;
            const int n1 = two ? __builtin_ctzll(m) : n0;
            if (two) m &= m - 1;
            Taps t[2];
            p2e_taps(a, n0, rt.x, rt.y, ct.x, ct.y, t[0]);
            p2e_taps(a, n1, rt.x, rt.y, ct.x, ct.y, t[1]);
            if (!two) { t[1].wa = 0.0f; t[1].wb = 0.0f; t[1].wc = 0.0f; t[1].wd = 0.0f; }
            unsigned o0[2], o1[2], dx[2]; int sel[2]; bool any[2];
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                const int n = u ? n1 : n0;
                const float wsum = (t[u].wa + t[u].wb) + (t[u].wc + t[u].wd);   // all >= 0 after the threshold
                l1 += wsum;
                any[u] = inside && wsum > 0.0f;
                dx[u] = 0; sel[u] = 0;
                if (XS1) {
                    const int xb = min(t[u].x0, a.pw - 2);
                    sel[u] = t[u].x0 - xb;
                    o0[u] = (unsigned)n * sN + (unsigned)t[u].y0 * sY + (unsigned)xb;
                    o1[u] = (unsigned)n * sN + (unsigned)t[u].y1 * sY + (unsigned)xb;
                } else {
                    o0[u] = (unsigned)n * sN + (unsigned)t[u].y0 * sY + (unsigned)t[u].x0 * sX;
                    o1[u] = (unsigned)n * sN + (unsigned)t[u].y1 * sY + (unsigned)t[u].x0 * sX;
                    dx[u] = (unsigned)(t[u].x1 - t[u].x0) * sX;
                }
            }
#pragma unroll
            for (int k = 0; k < PL; ++k) {
                const int p = p0 + k;
                if (p < planes) {
                    const size_t base = CONF ? (size_t)p * (size_t)a.sB
                                             : (size_t)(p / a.C) * (size_t)a.sB + (size_t)(p % a.C) * (size_t)a.sC;
#pragma unroll
                    for (int u = 0; u < 2; ++u)
                        if (any[u]) {
                            acc[k] += p2e_fetch<T, XS1>(pers + base, o0[u], o1[u], dx[u], sel[u], t[u]);
                            if (CONF) acc2[k] += p2e_fetch<T, XS1>(pers2 + base, o0[u], o1[u], dx[u], sel[u], t[u]);
                        }
                }
            }
        }
        if (inside) {
            const float rden = 1.0f / fmaxf(l1, 1e-12f);           // F.normalize(p=1, eps=1e-12), :192
#pragma unroll
            for (int k = 0; k < PL; ++k) {
                const int p = p0 + k;
                if (p < planes) {
                    if (CONF) {
                        const float pr = acc[k] * rden, cf = acc2[k] * rden;
                        const float z = (cf <= 1e-8f) ? 1.0f : 0.0f;                // spherical_model.py:310
                        reinterpret_cast<float*>(a.erp)[(size_t)p * erp_plane + pix] = pr / (cf + 1e-8f * z);   // :311
                    } else {
                        Store<T>::st(reinterpret_cast<T*>(a.erp) + (size_t)p * erp_plane + pix, acc[k] * rden);
                    }
                }
            }
        }
    }
}


// ------------------------------------------------------------------ LDS-staged blend (planar layout)
// The gather kernel above is bound by the vector L1 (a 64-lane gather of 8-byte pairs costs ~20 tag look-ups for 0.5 KB of
// useful data, profiles/r01f_resample_pmc.txt), not by HBM.  Here ONE WAVE owns an ERP tile of P2E_TH x P2E_TW pixels (4 per
// lane) and, per covering patch and image plane, streams the bounding box of the tile's bilinear taps inside the patch into
// LDS with 16-byte LDS-DMA pieces (global_load_lds: 64 useful bytes per L1 access, no VGPR staging, no ds_write) and takes the
// taps from LDS.  Boxes are constants of the geometry: a per-tile table (patch, origin, size) is built once per handle by the
// SAME tap function the blend uses (exact superset by construction), so the blend itself has no reduction, no barrier and no
// block-level synchronisation at all — a wave orders its own DMA -> ds_read hand-off with counted `s_waitcnt vmcnt(N)`, and
// keeps `nbuf` boxes in flight (ring of equal slots, sized for the largest box of the geometry).  Arithmetic, summation order
// and therefore every output bit are those of the gather kernel (tests compare the two with torch.equal).
constexpr int P2E_TH = 4, P2E_TW = 32;          // ERP tile of one wave: NPX = TH/2 pixels per lane (lane -> column lane%32, rows lane/32 + 2k)
constexpr int P2E_NPX = P2E_TH / 2;
constexpr int P2E_MAXC = 12;                    // table entries (covering patches) per tile
// The ORDERED table the kernel reads: per block slot P2E_REC records of 32 bytes — {tile id | -1, covering patches, 0...}, then per covering
// patch {entry x, entry y, sin l0, cos l0 | sin p1, cos p1, 0, 0} (the patch constants ride with the entry: one scalar load per patch, issued
// one patch AHEAD, instead of a table entry and then four dependent loads from the argument segment in front of every patch), one spare.
constexpr int P2E_REC = P2E_MAXC + 2;
constexpr int P2E_NJMAX = 8;                    // 1-KiB DMA pieces per box at most: boxes up to 8 KiB
constexpr int P2E_MAX_CHUNKS = 64 * P2E_NJMAX;

// entry: x = n | bw4 << 6 | bh << 16 | (entry 0 only) count << 26 (bw4 = 16-byte chunks per box row, bh = box rows, both <= 512;
// count = covering patches of the tile), y = xa | ymin << 16
template <int TH>                                                  // tile height: P2E_TH (every LDS kernel) or 8 (the one-plane walk kernel, round 5)
__global__ __launch_bounds__(256) void p2e_tiles_kernel(P2EArgs a, uint2* __restrict__ ent, int tiles_x, int ntiles, int epc,
                                                        int* __restrict__ stats)
{
    const int wid = (int)((blockIdx.x * 256 + threadIdx.x) >> 6), lane = threadIdx.x & 63;
    if (wid >= ntiles) return;
    const int ti = wid / tiles_x, tj = wid - ti * tiles_x;
    const int col = lane & 31, rsub = lane >> 5;
    const int j = tj * P2E_TW + col;
    const bool jin = j < a.W;
    const float2 ct = a.col_trig[jin ? j : a.W - 1];
    int cnt = 0, maxch = 0, sumch = 0;
    for (int n = 0; n < a.tab.N; ++n) {
        int xmin = 0x7fffffff, xmax = -1, ymin = 0x7fffffff, ymax = -1;
#pragma unroll
        for (int k = 0; k < TH / 2; ++k) {
            const int i = ti * TH + rsub + 2 * k;
            const bool iin = i < a.H;
            const float2 rt = a.row_trig[iin ? i : a.H - 1];
            Taps t;
            p2e_taps(a, n, rt.x, rt.y, ct.x, ct.y, t);
            const float wsum = (t.wa + t.wb) + (t.wc + t.wd);
            if (jin && iin && wsum > 0.0f) {
                xmin = min(xmin, t.x0); xmax = max(xmax, t.x1); ymin = min(ymin, t.y0); ymax = max(ymax, t.y1);
            }
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            xmin = min(xmin, __shfl_xor(xmin, o)); xmax = max(xmax, __shfl_xor(xmax, o));
            ymin = min(ymin, __shfl_xor(ymin, o)); ymax = max(ymax, __shfl_xor(ymax, o));
        }
        if (xmax < 0) continue;                                   // wave-uniform: patch n covers no pixel of this tile
        int xa = xmin / epc * epc;
        int bw4 = (xmax / epc * epc + epc - xa) / epc;
        const int bh = ymax - ymin + 1;
        const bool fits = bw4 < 1024 && bh < 1024 && xa < 65536 && ymin < 65536 && bw4 * bh <= P2E_MAX_CHUNKS;
        maxch = max(maxch, fits ? bw4 * bh : P2E_MAX_CHUNKS + 1);
        sumch += fits ? bw4 * bh : (1 << 20);
        if (lane == 0 && cnt < P2E_MAXC && fits)
            ent[(size_t)wid * P2E_MAXC + cnt] = make_uint2((unsigned)n | ((unsigned)bw4 << 6) | ((unsigned)bh << 16),
                                                           (unsigned)xa | ((unsigned)ymin << 16));
        ++cnt;
    }
    if (lane == 0) {
        for (int c = cnt; c < P2E_MAXC; ++c) ent[(size_t)wid * P2E_MAXC + c] = make_uint2(0u, 0u);
        if (cnt <= P2E_MAXC) ent[(size_t)wid * P2E_MAXC].x |= (unsigned)cnt << 26;
        atomicMax(&stats[0], maxch); atomicMax(&stats[1], cnt); atomicMax(&stats[2], sumch);
    }
}

// column origin of a tap pair: the box origin, shifted so that an x1 == x0 tap (right patch edge) becomes the pair's second element
__device__ __forceinline__ int xa_adj(int x0, int x1, int xa) { return xa + 1 - (x1 - x0); }

template <typename T> struct LdsPair;
template <> struct LdsPair<float> {
    static __device__ __forceinline__ void ld(const unsigned char* b, int o, float& x, float& y)
    { const float* p = reinterpret_cast<const float*>(b) + o; x = p[0]; y = p[1]; }              // one ds_read2_b32
};
template <> struct LdsPair<__half> {
    // halfs o, o+1: one ds_read2_b32 of the two 32-bit words around them + a byte-align (no 16-bit LDS reads, which cost a full
    // LDS instruction each)
    static __device__ __forceinline__ void ld(const unsigned char* b, int o, float& x, float& y)
    {
        const unsigned* p = reinterpret_cast<const unsigned*>(b) + (o >> 1);
        const unsigned w0 = p[0], w1 = p[1];
        const unsigned v = (o & 1) ? __builtin_amdgcn_alignbyte(w1, w0, 2u) : w0;
        const __half2 h = *reinterpret_cast<const __half2*>(&v);
        x = __low2float(h); y = __high2float(h);
    }
};

typedef __amdgpu_buffer_rsrc_t p2e_rsrc_t;
typedef __attribute__((address_space(3))) void* p2e_lptr_t;
__device__ __forceinline__ p2e_rsrc_t p2e_make_rsrc(const void* p, unsigned bytes)
{
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p), (short)0, (int)bytes, 0x00020000);
}
// one LDS-DMA instruction: lane l's 16 bytes at buffer offset voff + soff (soff wave-uniform) land at lds + 16 l; an offset
// outside the buffer deposits zeros without touching memory (used for the padding lanes of a box's last piece).
// (A plain function: the host pass does not accept this builtin inside a kernel template's body.)
__device__ __forceinline__ void p2e_dma16(p2e_rsrc_t rs, unsigned char* lds, unsigned voff, unsigned soff)
{
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (p2e_lptr_t)lds, 16, (int)voff, (int)soff, 0, 0);
}
template <int N> __device__ __forceinline__ void p2e_wait_vm()
{
    static_assert(N >= 0 && N <= 63, "vmcnt is a 6-bit counter");
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

// ONE WAVE (= one 64-thread block) per (ERP tile, group of PL planes); nothing in here synchronises with another wave.  The
// covering patches of the tile are walked one after the other; per patch: (1) the first NB boxes are put in flight, (2) the tap
// geometry of my NPX pixels is evaluated while they travel, (3) the PL stages run.  A stage = (patch, plane p): the box of the
// patch in plane p, NJ = ceil(chunks / 64) DMA pieces of 1 KiB, NB ring slots (stages in flight; PL % NB == 0 so that a stage's
// slot is a compile-time constant).  The stage loop is instantiated per NJ (1..8), which makes every s_waitcnt count and every
// DMA piece loop a compile-time constant: waiting for stage p leaves min(NB-1, PL-1-p) later stages = that many x NJ pieces in
// flight.  The tile is small (4 x 32 pixels, 2 per lane) so that a launch of BASELINE size still has >= 16 waves per CU to
// hide each other's latencies: with one big tile per wave the launch is a handful of long dependent instruction chains per SIMD
// (measured: 8 x 32 tiles, 24 us at B = 8 whatever the ring depth).
// ring of P2E_RING_KB 1-KiB pieces per wave: a stage of `pieces` pieces (NJ, x2 for the fused confidence blend) gets the largest power
// of two <= RING / pieces slots (at most nbmax, at most PL — PL % NB == 0 keeps a stage's slot a compile-time constant), each of
// exactly `pieces` KiB: small boxes (the common case) keep 4 stages in flight, the rare large ones 2 or 1, in one LDS footprint
constexpr int P2E_RING_KB = 10;
constexpr int p2e_slots(int pieces, int nbmax, int pl)
{
    int nb = P2E_RING_KB / pieces >= 4 ? 4 : P2E_RING_KB / pieces >= 2 ? 2 : 1;
    if (nb > nbmax) nb = nbmax;
    while (nb > pl) nb >>= 1;
    return nb;
}

template <typename T, int PL, bool CONF, int NBMAX>
__global__ __launch_bounds__(64, 4) void p2e_lds_kernel(P2EArgs a, const uint2* __restrict__ tiles, int tiles_x, int tiles_y,
                                                        unsigned tensor_bytes, int p_first)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char p2e_smem[];        // the ONLY LDS object of this kernel
    constexpr int EPC = 16 / (int)sizeof(T), M = CONF ? 2 : 1, NPX = P2E_NPX;
    const int lane = threadIdx.x;
    long long tr0 = 0, tr1 = 0;
    if (OMNI_DBG(a, 16)) tr0 = wall_clock64();
    // block -> tile: the ordered table of the geometry (omni_p2e_build_tiles: whole bands of tile rows per XCD, and inside an XCD the
    // tiles dealt so that every CU gets the same amount of work; slot blockIdx.x holds the tile's entries and its id)
    const uint4* __restrict__ sl = reinterpret_cast<const uint4*>(tiles) + (size_t)blockIdx.x * (2 * P2E_REC);
    const uint4 hd = sl[0];
    const int wid = (int)hd.x;
    if (wid < 0) return;                                                            // padding slot (block-uniform)
    const int ti = wid / tiles_x, tj = wid - ti * tiles_x;
    const int col = lane & 31, rsub = lane >> 5;
    const int j = tj * P2E_TW + col;
    const bool jin = j < a.W;
    const int p_begin = p_first + (int)blockIdx.y * PL;
    unsigned char* const ring = p2e_smem;
    const p2e_rsrc_t rs1 = p2e_make_rsrc(a.pers, tensor_bytes);
    const p2e_rsrc_t rs2 = p2e_make_rsrc(CONF ? a.pers2 : a.pers, tensor_bytes);
    const unsigned sYb = (unsigned)a.sY * (unsigned)sizeof(T), sNb = (unsigned)a.sN * (unsigned)sizeof(T);

    // ---- the trig of my pixels (vector loads) and the tile's patch list (wave-uniform address: scalar loads)
    const float2 ct = a.col_trig[jin ? j : a.W - 1];
    float2 rt[NPX];
#pragma unroll
    for (int k = 0; k < NPX; ++k) rt[k] = a.row_trig[min(ti * P2E_TH + rsub + 2 * k, a.H - 1)];
    const int ncand = (int)hd.y;
    uint4 ea = sl[2], eb = sl[3];                                  // record of the first patch
    unsigned poff_l;                                               // lane p: byte offset of plane p_begin + p (< 2^31, host-checked)
    {
        const int p = p_begin + min(lane, PL - 1);
        const unsigned e = CONF ? (unsigned)p * (unsigned)a.sB : (unsigned)(p / a.C) * (unsigned)a.sB + (unsigned)(p % a.C) * (unsigned)a.sC;
        poff_l = e * (unsigned)sizeof(T);
    }
    // all ordinary vector loads are consumed HERE, before the first LDS-DMA is issued (a later first use would make the compiler
    // drain the DMA queue with vmcnt(0))
    asm volatile("" ::"v"(ct.x), "v"(ct.y));
#pragma unroll
    for (int k = 0; k < NPX; ++k) asm volatile("" ::"v"(rt[k].x), "v"(rt[k].y));

    if (OMNI_DBG(a, 16)) tr1 = wall_clock64();
    float acc[PL][NPX], acc2[CONF ? PL : 1][NPX], l1[NPX];
#pragma unroll
    for (int k = 0; k < NPX; ++k) {
        l1[k] = 0.0f;
#pragma unroll
        for (int p = 0; p < PL; ++p) { acc[p][k] = 0.0f; if constexpr (CONF) acc2[p][k] = 0.0f; }
    }

    for (int c = 0; c < ncand; ++c) {
        const uint4 na = sl[2 * (c + 2)], nb4 = sl[2 * (c + 2) + 1];   // the NEXT patch's record travels under this patch's stages
        const unsigned e0 = ea.x, e1 = ea.y;
        const float sl0 = __uint_as_float(ea.z), cl0 = __uint_as_float(ea.w), sp = __uint_as_float(eb.x), cp = __uint_as_float(eb.y);
        const int n = e0 & 63, bw4 = (e0 >> 6) & 1023, xa = e1 & 0xffff, ymin = e1 >> 16;
        const int nchunk = bw4 * (int)((e0 >> 16) & 1023), njj = (nchunk + 63) >> 6;
        const unsigned base = (unsigned)n * sNb + (unsigned)ymin * sYb + (unsigned)xa * (unsigned)sizeof(T);
        unsigned g[P2E_NJMAX];                                      // byte offset of my chunk of piece q, from the box origin
        // ---- (1) fill the ring: stages 0..NB-1
        auto prologue = [&]<int NJ>(std::integral_constant<int, NJ>) {
            constexpr int NB = p2e_slots(NJ * M, NBMAX, PL);
            constexpr unsigned slot_bytes = NJ * M * 1024u;
            const float rbw = __builtin_amdgcn_rcpf((float)bw4);
#pragma unroll
            for (int q = 0; q < NJ; ++q) {
                const int qc = q * 64 + lane;
                const int rr = (int)(((float)qc + 0.5f) * rbw);                        // qc / bw4, exact for qc, bw4 <= 1024
                g[q] = qc < nchunk ? (unsigned)rr * sYb + (unsigned)(qc - rr * bw4) * 16u : 0x80000000u;   // past the end: zeros
            }
            if (OMNI_DBG(a, 4)) return;
#pragma unroll
            for (int d = 0; d < NB; ++d) {
                const unsigned so = (unsigned)__builtin_amdgcn_readlane((int)poff_l, d) + base;
                unsigned char* dst = ring + (unsigned)d * slot_bytes;
#pragma unroll
                for (int q = 0; q < NJ; ++q) {
                    p2e_dma16(rs1, dst + q * 1024, g[q], so);
                    if (CONF) p2e_dma16(rs2, dst + NJ * 1024u + q * 1024, g[q], so);
                }
            }
        };
        switch (njj) {
        case 1: prologue(std::integral_constant<int, 1>()); break;
        case 2: prologue(std::integral_constant<int, 2>()); break;
        case 3: prologue(std::integral_constant<int, 3>()); break;
        case 4: prologue(std::integral_constant<int, 4>()); break;
        case 5: prologue(std::integral_constant<int, 5>()); break;
        case 6: prologue(std::integral_constant<int, 6>()); break;
        case 7: prologue(std::integral_constant<int, 7>()); break;
        default: prologue(std::integral_constant<int, 8>()); break;
        }
        // ---- (2) taps of patch n for my pixels, as LDS element offsets inside the box (the DMA latency overlaps this)
        const int pitch = bw4 * EPC;
        int r0[NPX], r1[NPX];
        float wa[NPX], wb[NPX], wc[NPX], wd[NPX];
        const float cd = ct.y * cl0 + ct.x * sl0;                   // p2e_lon with the record's constants (ct = (sin lon, cos lon)): cos(lon - l0)
        const float sd = ct.x * cl0 - ct.y * sl0;                   //                                                          sin(lon - l0)
#pragma unroll
        for (int k = 0; k < NPX; ++k) {
            // the pair (x0, x0+1) of both tap rows; at the right patch edge (x1 == x0: wa == wb == 0, see p2e_taps_f) the pair is
            // moved one column left so that its SECOND element is the x1 tap.  Pixels the patch does not cover have all weights 0
            // and read the box origin (p2e_taps_box).
            if (OMNI_DBG(a, 1)) { r0[k] = r1[k] = 1; wa[k] = wb[k] = wc[k] = wd[k] = 0.25f * rt[k].x; l1[k] += rt[k].x; continue; }
            l1[k] += p2e_taps_box(a, sp, cp, rt[k].x, rt[k].y, cd, sd, xa + 1, ymin, pitch, r0[k], r1[k], wa[k], wb[k], wc[k], wd[k]);
        }
        // ---- (3) the PL stages, instantiated on the number of DMA pieces per box
        auto stages = [&]<int NJ>(std::integral_constant<int, NJ>) {
            constexpr int NB = p2e_slots(NJ * M, NBMAX, PL);
            constexpr unsigned slot_bytes = NJ * M * 1024u;
            auto stage = [&]<int P>(std::integral_constant<int, P>) {
                constexpr int SLOT = P % NB;
                constexpr int K1 = (NB - 1 < PL - 1 - P) ? NB - 1 : PL - 1 - P;
                p2e_wait_vm<K1 * NJ * M>();
                const unsigned char* box = ring + (unsigned)SLOT * slot_bytes;
#pragma unroll
                for (int k = 0; k < NPX; ++k) {
                    if (OMNI_DBG(a, 2)) { acc[P][k] += wa[k] * (float)P; continue; }
                    float ax, ay, bx, by;
                    LdsPair<T>::ld(box, r0[k], ax, ay);
                    LdsPair<T>::ld(box, r1[k], bx, by);
                    // taps (y0,x0) (y1,x0) (y0,x1) (y1,x1) in the gather kernel's order of operations
                    acc[P][k] += fmaf(by, wd[k], fmaf(ay, wc[k], fmaf(bx, wb[k], ax * wa[k])));
                    if constexpr (CONF) {
                        const unsigned char* box2 = box + NJ * 1024u;
                        LdsPair<T>::ld(box2, r0[k], ax, ay);
                        LdsPair<T>::ld(box2, r1[k], bx, by);
                        acc2[P][k] += fmaf(by, wd[k], fmaf(ay, wc[k], fmaf(bx, wb[k], ax * wa[k])));
                    }
                }
                if constexpr (P + NB < PL) if (!OMNI_DBG(a, 4)) {   // refill the slot just read (its ds_reads must have returned first)
                    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                    const unsigned so = (unsigned)__builtin_amdgcn_readlane((int)poff_l, P + NB) + base;
                    unsigned char* dst = ring + (unsigned)SLOT * slot_bytes;
#pragma unroll
                    for (int q = 0; q < NJ; ++q) {
                        p2e_dma16(rs1, dst + q * 1024, g[q], so);
                        if (CONF) p2e_dma16(rs2, dst + NJ * 1024u + q * 1024, g[q], so);
                    }
                }
            };
            [&]<int... P>(std::integer_sequence<int, P...>) { (stage(std::integral_constant<int, P>()), ...); }(std::make_integer_sequence<int, PL>());
        };
        switch (njj) {
        case 1: stages(std::integral_constant<int, 1>()); break;
        case 2: stages(std::integral_constant<int, 2>()); break;
        case 3: stages(std::integral_constant<int, 3>()); break;
        case 4: stages(std::integral_constant<int, 4>()); break;
        case 5: stages(std::integral_constant<int, 5>()); break;
        case 6: stages(std::integral_constant<int, 6>()); break;
        case 7: stages(std::integral_constant<int, 7>()); break;
        default: stages(std::integral_constant<int, 8>()); break;
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");         // the ring is a DMA target again for the next patch
        ea = na; eb = nb4;
    }
    // ---- normalise and store (pers2equi_v3.py:192-196; K11: spherical_model.py:310-311)
    const size_t erp_plane = (size_t)a.H * a.W;
#pragma unroll
    for (int k = 0; k < NPX; ++k) {
        const int i = ti * P2E_TH + rsub + 2 * k;
        if (!(jin && i < a.H)) continue;
        if (OMNI_DBG(a, 8) && l1[k] != -1.0f) continue;
        const float rden = 1.0f / fmaxf(l1[k], 1e-12f);
        const size_t pix = (size_t)i * a.W + j;
#pragma unroll
        for (int p = 0; p < PL; ++p) {
            const size_t o = (size_t)(p_begin + p) * erp_plane + pix;
            if constexpr (CONF) {
                const float pr = acc[p][k] * rden, cf = acc2[p][k] * rden;
                const float z = (cf <= 1e-8f) ? 1.0f : 0.0f;
                const float r = pr / (cf + 1e-8f * z);
                if (a.store_nt) Store<float>::st_nt(reinterpret_cast<float*>(a.erp) + o, r); else reinterpret_cast<float*>(a.erp)[o] = r;
            } else {
                if (a.store_nt) Store<T>::st_nt(reinterpret_cast<T*>(a.erp) + o, acc[p][k] * rden);
                else Store<T>::st(reinterpret_cast<T*>(a.erp) + o, acc[p][k] * rden);
            }
        }
    }
    if (OMNI_DBG(a, 16)) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if (lane == 0 && a.trace) {
            long long* t = a.trace + 4 * ((size_t)blockIdx.y * gridDim.x + blockIdx.x);
            t[0] = tr0; t[1] = tr1; t[2] = wall_clock64();
            t[3] = (long long)(unsigned)__builtin_amdgcn_s_getreg((31 << 11) | 4) | ((long long)(unsigned)__builtin_amdgcn_s_getreg((31 << 11) | 20) << 32) | ((long long)ncand << 40);
        }
    }
}

// ------------------------------------------------------------------ flat-pipeline blend (round 4: default for the planar layout)
// What bounds p2e_lds_kernel is not bytes but its waves' dependent CHAINS: a launch of BASELINE size has one wave per tile, every wave resident
// at once, and ends with the longest chain — a 4-patch tile, 3.3 us + 2.9 us per patch (profiles/r03b_resample_sweeps.txt, r03_p2e_shed).  Its ISA
// shows why a patch costs 2.9 us: per stage [wait DMA -> 2 ds_read2 -> wait -> 4 fma -> 2 ds_read2 -> wait -> 4 fma -> wait -> refill], i.e. two
// exposed LDS round trips per plane; a drained ring at every patch boundary (the next patch's first boxes wait for its table record, then a full
// memory round trip); and in front of everything two DEPENDENT cold misses (slot record -> tile id -> trig tables).  This kernel keeps the tile,
// the taps, the boxes and every output bit of p2e_lds_kernel and changes the schedule:
//   * ONE stage stream per tile: stage s = (patch s / PL, plane s % PL), NB ring slots, refill distance NB — ACROSS patch boundaries: while the
//     last stages of patch c are consumed, the first stages of patch c+1 are already in flight (its DMA offsets are computed one patch ahead), so
//     the tap geometry of c+1 is evaluated under its own boxes' flight time;
//   * stages are consumed in PAIRS where 4 slots fit the ring (U = 2): one counted wait, all ds_read2 of both planes issued back to back, one LDS
//     wait, both refills, then the 16 fmas — one exposed LDS round trip per two planes instead of four;
//   * the number of 1-KiB pieces per stage is a constant of the TILE (the largest box of its patches; smaller boxes pad with out-of-range lanes,
//     which deposit zeros without touching memory): every s_waitcnt count of the flat stream is a compile-time constant of (NJ, last patch or not);
//   * the slot record carries the tile's trig (32 column + 4 row pairs, 288 B): header, first patch record and trig leave in parallel, ONE cold
//     round trip in front of the first DMA.
constexpr int P2W_NJMAX = 6;                                   // largest box (KiB) the walk kernel is instantiated for
constexpr int P2W_SLOT = 768;                                  // bytes per block slot: header 32 | 12 patch records x 32 | trig 288 | pad
constexpr int P2W_OFF_PATCH = 32, P2W_OFF_TRIG = 32 + 32 * P2E_MAXC;
static_assert(P2W_OFF_TRIG + 8 * (P2E_TW + 8) <= P2W_SLOT, "slot layout (8-row tiles included)");
constexpr int p2w_nb(int pieces, int pl)
{
    int nb = P2E_RING_KB / pieces >= 4 ? 4 : P2E_RING_KB / pieces >= 2 ? 2 : 1;
    while (nb > pl) nb >>= 1;
    return nb;
}
constexpr int p2w_u(int nb, int pl) { return nb == 4 ? 2 : (nb == 2 && pl == 2) ? 2 : 1; }

constexpr int P2W_WPB = 1;                                     // waves per block: independent waves (no barrier, each its own tile and ring) — 4x fewer workgroups to dispatch
// TH: rows of the ERP tile a wave owns (TH / 2 pixels per lane).  8 for ONE plane per wave (round 5): the per-(wave, patch) set-up — records, box
// parameters, the stage switch — is paid half as often (cfg 5 fp32 96 -> 88 us, cfg 3 27.3 -> 26.3, fp16 unchanged: profiles/r05f_p2e_tile8.txt).  Same taps, same candidate order, same blend
// expression per pixel (a patch that does not cover a pixel adds an exact 0): the bits do not depend on the tile.
template <typename T, int PL, bool CONF, int TH = P2E_TH>
__global__ __launch_bounds__(64 * P2W_WPB, TH > P2E_TH ? (CONF ? 4 : 5) : PL == 1 ? (CONF ? 6 : 8) : PL == 2 ? (CONF ? 5 : 6) : 4) void p2e_walk_kernel(P2EArgs a, const unsigned char* __restrict__ table, int tiles_x, unsigned tensor_bytes, int p_first, unsigned ring_bytes)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char p2w_smem_all[];    // the ONLY LDS object of this kernel: one ring per wave
    constexpr int EPC = 16 / (int)sizeof(T), M = CONF ? 2 : 1, NPX = TH / 2;
    const int lane = threadIdx.x & 63, wv = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    unsigned char* const p2w_smem = p2w_smem_all + (unsigned)wv * ring_bytes;
    const unsigned char* __restrict__ slot = table + ((size_t)blockIdx.x * P2W_WPB + (size_t)wv) * P2W_SLOT;
    const uint4* __restrict__ sl = reinterpret_cast<const uint4*>(slot);            // wave-uniform addresses: scalar loads
    const uint4 hd = sl[0];
    // my pixels' trig: an address that does not depend on the header (issued beside it)
    const int col = lane & 31, rsub = lane >> 5;
    const float2* __restrict__ tg = reinterpret_cast<const float2*>(slot + P2W_OFF_TRIG);
    const float2 ct = tg[col];
    float2 rt[NPX];
#pragma unroll
    for (int k = 0; k < NPX; ++k) rt[k] = tg[P2E_TW + rsub + 2 * k];
    uint4 ea = sl[2], eb = sl[3];                                  // record of the first patch
    const int wid = (int)hd.x;
    if (wid < 0) return;                                           // padding slot (block-uniform)
    if (OMNI_DBG(a, 32)) return;                                   // (debug: the launch floor — blocks that read their header and end)
    const int ncand = (int)hd.y, njt = (int)hd.z;
    const int ti = wid / tiles_x, tj = wid - ti * tiles_x;
    const int j = tj * P2E_TW + col;
    const bool jin = j < a.W;
    const int p_begin = p_first + (int)blockIdx.y * PL;
    unsigned char* const ring = p2w_smem;
    const p2e_rsrc_t rs1 = p2e_make_rsrc(a.pers, tensor_bytes);
    const p2e_rsrc_t rs2 = p2e_make_rsrc(CONF ? a.pers2 : a.pers, tensor_bytes);
    const unsigned sYb = (unsigned)a.sY * (unsigned)sizeof(T), sNb = (unsigned)a.sN * (unsigned)sizeof(T);
    unsigned poff_l;                                               // lane p: byte offset of plane p_begin + p (< 2^31, host-checked)
    {
        const int p = p_begin + min(lane, PL - 1);
        const unsigned e = CONF ? (unsigned)p * (unsigned)a.sB : (unsigned)(p / a.C) * (unsigned)a.sB + (unsigned)(p % a.C) * (unsigned)a.sC;
        poff_l = e * (unsigned)sizeof(T);
    }
    // all ordinary vector loads are consumed HERE, before the first LDS-DMA is issued (a later first use would make the compiler drain the DMA
    // queue with vmcnt(0))
    asm volatile("" ::"v"(ct.x), "v"(ct.y));
#pragma unroll
    for (int k = 0; k < NPX; ++k) asm volatile("" ::"v"(rt[k].x), "v"(rt[k].y));

    float acc[PL][NPX], acc2[CONF ? PL : 1][NPX], l1[NPX];
#pragma unroll
    for (int k = 0; k < NPX; ++k) {
        l1[k] = 0.0f;
#pragma unroll
        for (int p = 0; p < PL; ++p) { acc[p][k] = 0.0f; if constexpr (CONF) acc2[p][k] = 0.0f; }
    }
    // taps of the current patch for my pixels: LDS element offsets inside the box + the four weights (p2e_taps_core: the bits of every kernel)
    int r0[NPX], r1[NPX];
    float wa[NPX], wb[NPX], wc[NPX], wd[NPX];
    auto taps = [&](const uint4& ra, const uint4& rb) {
        const unsigned e0 = ra.x, e1 = ra.y;
        const float sl0 = __uint_as_float(ra.z), cl0 = __uint_as_float(ra.w), sp = __uint_as_float(rb.x), cp = __uint_as_float(rb.y);
        const int bw4 = (e0 >> 6) & 1023, xa = e1 & 0xffff, ymin = e1 >> 16, pitch = bw4 * EPC;
        const float cd = ct.y * cl0 + ct.x * sl0;                   // p2e_lon with the record's constants: cos(lon - l0)
        const float sd = ct.x * cl0 - ct.y * sl0;                   //                                      sin(lon - l0)
#pragma unroll
        for (int k = 0; k < NPX; ++k) {
            if (OMNI_DBG(a, 1)) { r0[k] = r1[k] = 1; wa[k] = wb[k] = wc[k] = wd[k] = 0.25f * rt[k].x; l1[k] += rt[k].x; (void)cd; (void)sd; continue; }
            l1[k] += p2e_taps_box(a, sp, cp, rt[k].x, rt[k].y, cd, sd, xa + 1, ymin, pitch, r0[k], r1[k], wa[k], wb[k], wc[k], wd[k]);   // (see p2e_lds_kernel)
        }
    };

    // byte offset of my chunk of piece q from the box origin + the box origin (wave-uniform) of the patch the NEXT refill reads: the current
    // patch for the iterations whose refill stages lie inside it, the following patch from iteration JS on (ONE set of registers: U | NB | PL,
    // so an iteration refills entirely from one patch).  Only the part of the program that depends on NJ — the DMA pieces, the stage loop and its
    // counted waits — is instantiated per NJ (the tap geometry, ~350 instructions, exists twice in the kernel, not twice per NJ: 45 -> ~25 KB of
    // code; the two CUs that share an instruction cache run tiles of every NJ at once).
    unsigned gc[P2W_NJMAX], base_c = 0;
    auto run = [&]<int NJ, int PHASE>(std::integral_constant<int, NJ>, std::integral_constant<int, PHASE>, const uint4& nrec) {
        constexpr int NB = p2w_nb(NJ * M, PL), U = p2w_u(NB, PL), ITERS = PL / U;
        constexpr unsigned slot_bytes = NJ * M * 1024u;
        static_assert(PL % NB == 0 && NB % U == 0, "a stage's ring slot must be a compile-time constant");
        constexpr int JS = (PL - NB) / U;                          // first iteration whose refills belong to the next patch
        auto dma_params = [&](const uint4& ra, unsigned (&g)[P2W_NJMAX], unsigned& base) {
            const unsigned e0 = ra.x, e1 = ra.y;
            const int n = e0 & 63, bw4 = (e0 >> 6) & 1023, xa = e1 & 0xffff, ymin = e1 >> 16;
            const int nchunk = bw4 * (int)((e0 >> 16) & 1023);
            base = (unsigned)n * sNb + (unsigned)ymin * sYb + (unsigned)xa * (unsigned)sizeof(T);
            const float rbw = __builtin_amdgcn_rcpf((float)bw4);
#pragma unroll
            for (int q = 0; q < NJ; ++q) {
                const int qc = q * 64 + lane;
                const int rr = (int)(((float)qc + 0.5f) * rbw);                        // qc / bw4, exact for qc, bw4 <= 1024
                g[q] = qc < nchunk ? (unsigned)__umul24((unsigned)rr, sYb) + (unsigned)(qc - __mul24(rr, bw4)) * 16u : 0x80000000u;   // past the end (of THIS patch's box): zeros  (row pitch < 2^24 bytes: host-checked)
            }
        };
        auto issue = [&](const unsigned (&g)[P2W_NJMAX], unsigned base, int plane, int slot_i) {
            if (OMNI_DBG(a, 4)) return;
            const unsigned so = (unsigned)__builtin_amdgcn_readlane((int)poff_l, plane) + base;
            unsigned char* dst = ring + (unsigned)slot_i * slot_bytes;
#pragma unroll
            for (int q = 0; q < NJ; ++q) {
                p2e_dma16(rs1, dst + q * 1024, g[q], so);
                if (CONF) p2e_dma16(rs2, dst + NJ * 1024u + q * 1024, g[q], so);
            }
        };
        // one patch of the stream: ITERS iterations of U stages.  LAST: no patch follows (nothing to refill from beyond plane PL - 1).
        auto body = [&]<bool LAST>(std::bool_constant<LAST>, const uint4& nrec) {
            [&]<int... J>(std::integer_sequence<int, J...>) {
                (([&] {
                    constexpr int first = J * U;
                    if constexpr (!LAST && J == JS) dma_params(nrec, gc, base_c);     // from here on the refills read the next patch
                    // stages younger than the ones consumed here and already in flight: first + U .. first + NB - 1 of the flat stream
                    constexpr int ahead = NB - U;
                    constexpr int left = PL - first - U;            // ... of which this many belong to THIS patch
                    constexpr int younger = LAST ? (left < ahead ? (left > 0 ? left : 0) : ahead) : ahead;
                    p2e_wait_vm<younger * NJ * M>();
                    float v[U][M][NPX][4];
#pragma unroll
                    for (int u = 0; u < U; ++u) {
                        const unsigned char* box = ring + (unsigned)((first + u) % NB) * slot_bytes;
#pragma unroll
                        for (int m = 0; m < M; ++m)
#pragma unroll
                            for (int k = 0; k < NPX; ++k) {
                                if (OMNI_DBG(a, 2)) { v[u][m][k][0] = v[u][m][k][1] = v[u][m][k][2] = v[u][m][k][3] = wa[k]; continue; }
                                LdsPair<T>::ld(box + m * NJ * 1024u, r0[k], v[u][m][k][0], v[u][m][k][1]);
                                LdsPair<T>::ld(box + m * NJ * 1024u, r1[k], v[u][m][k][2], v[u][m][k][3]);
                            }
                    }
                    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // every read of these slots has returned: they are DMA targets again
#pragma unroll
                    for (int u = 0; u < U; ++u) {
                        const int t = first + u + NB;
                        if (t < PL) issue(gc, base_c, t, (first + u) % NB);
                        else if (!LAST) issue(gc, base_c, t - PL, (first + u) % NB);
                    }
#pragma unroll
                    for (int u = 0; u < U; ++u)
#pragma unroll
                        for (int k = 0; k < NPX; ++k) {
                            // taps (y0,x0) (y1,x0) (y0,x1) (y1,x1) in the gather kernel's order of operations
                            acc[first + u][k] += fmaf(v[u][0][k][3], wd[k], fmaf(v[u][0][k][1], wc[k], fmaf(v[u][0][k][2], wb[k], v[u][0][k][0] * wa[k])));
                            if constexpr (CONF)
                                acc2[first + u][k] += fmaf(v[u][1][k][3], wd[k], fmaf(v[u][1][k][1], wc[k], fmaf(v[u][1][k][2], wb[k], v[u][1][k][0] * wa[k])));
                            // the blend happens HERE (volatile statements keep their order): left to itself the scheduler defers the fmas of every
                            // iteration to the end of the patch and keeps all their operands alive — 64 registers, scratch spills
                            asm volatile("" : "+v"(acc[first + u][k]));
                            if constexpr (CONF) asm volatile("" : "+v"(acc2[first + u][k]));
                        }
                }()), ...);
            }(std::make_integer_sequence<int, ITERS>());
        };
        if constexpr (PHASE == 0) {                               // head of the stream: the first NB stages of patch 0 go out
            dma_params(nrec, gc, base_c);
#pragma unroll
            for (int d = 0; d < NB; ++d) issue(gc, base_c, d, d);
        } else if constexpr (PHASE == 1) body(std::bool_constant<false>(), nrec);
        else body(std::bool_constant<true>(), nrec);
    };
    // (boxes of up to P2W_NJMAX KiB: every BASELINE shape is 1-3; a geometry with larger boxes — patches much finer than the ERP — stays on p2e_lds_kernel)
    auto with_nj = [&]<int PHASE>(std::integral_constant<int, PHASE> ph, const uint4& nrec) {
        switch (njt) {
        case 1: run(std::integral_constant<int, 1>(), ph, nrec); break;
        case 2: run(std::integral_constant<int, 2>(), ph, nrec); break;
        case 3: run(std::integral_constant<int, 3>(), ph, nrec); break;
        case 4: run(std::integral_constant<int, 4>(), ph, nrec); break;
        case 5: run(std::integral_constant<int, 5>(), ph, nrec); break;
        default: run(std::integral_constant<int, P2W_NJMAX>(), ph, nrec); break;
        }
    };
    with_nj(std::integral_constant<int, 0>(), ea);
    taps(ea, eb);                                                  // ... evaluated while the first boxes travel
    for (int c = 0; c < ncand; ++c) {
        const uint4 na = sl[2 * (c + 2)], nb4 = sl[2 * (c + 2) + 1];        // record of patch c + 1 (a zero record past the list)
        if (c + 1 < ncand) {
            with_nj(std::integral_constant<int, 1>(), na);
            taps(na, nb4);                                          // ... under the flight time of its own first boxes
        } else {
            with_nj(std::integral_constant<int, 2>(), na);
        }
    }
    // ---- normalise and store (pers2equi_v3.py:192-196; K11: spherical_model.py:310-311)
    const size_t erp_plane = (size_t)a.H * a.W;
#pragma unroll
    for (int k = 0; k < NPX; ++k) {
        const int i = ti * TH + rsub + 2 * k;
        if (!(jin && i < a.H)) continue;
        if (OMNI_DBG(a, 8) && l1[k] != -1.0f) continue;
        const float rden = 1.0f / fmaxf(l1[k], 1e-12f);
        const size_t pix = (size_t)i * a.W + j;
#pragma unroll
        for (int p = 0; p < PL; ++p) {
            const size_t o = (size_t)(p_begin + p) * erp_plane + pix;
            if constexpr (CONF) {
                const float pr = acc[p][k] * rden, cf = acc2[p][k] * rden;
                const float z = (cf <= 1e-8f) ? 1.0f : 0.0f;
                const float r = pr / (cf + 1e-8f * z);
                if (a.store_nt) Store<float>::st_nt(reinterpret_cast<float*>(a.erp) + o, r); else reinterpret_cast<float*>(a.erp)[o] = r;
            } else {
                if (a.store_nt) Store<T>::st_nt(reinterpret_cast<T*>(a.erp) + o, acc[p][k] * rden);
                else Store<T>::st(reinterpret_cast<T*>(a.erp) + o, acc[p][k] * rden);
            }
        }
    }
}

// ------------------------------------------------------------------ (round 5: ONE plane per wave with D = 2 / 3 patches in flight, measured and dropped)
// p2e_walk1_kernel kept the boxes of D consecutive patches of a single-plane tile in flight (the flat pipeline above holds ONE stage when a stage is a
// whole patch).  Same bits; BASELINE cfg 5 fp16 108.8 / 115.3 / 106.2 us and cfg 3 27.1 / 27.5 / 29.9 us for D = 1 / 2 / 3, same box
// (profiles/r05c_p2e_single_plane.txt): the one-plane blend is not waiting for its boxes.  Neither is it short of issue slots: 24 % fewer VALU
// instructions in the tap section (p2e_taps_box: validity on two factors, 24-bit index arithmetic — kept) changed nothing either.  With parts switched
// off (debug build): no tap geometry -47 us, no LDS reads -14, no DMA -1.5, no stores -6, everything off 54 us of 113: half of the kernel is the per-(wave,
// patch) skeleton — records, box parameters, the stage switch.  MORE PIXELS PER WAVE (8-row tiles, TH = 8 above: half as many (wave, patch) pairs) bought
// 8 % at fp32 (96 -> 88 us), 3 % at BASELINE cfg 3 and nothing at fp16: with 4.8 covering patches per ERP pixel at nrows 6 (2.1 at nrows 4; the oracle's cover count) the kernel is bound by the
// instructions it spends per (pixel, patch), which no tile shape changes.
// ------------------------------------------------------------------ (round 3: a plane-major form of this kernel, measured and dropped)
// p2e_tile_kernel swapped the loops: set-up once per tile (all table entries and patch constants in one batch of scalar loads, the taps of
// EVERY covering patch in registers, 6 per pixel and patch), then ONE pipeline of PL stages per tile, a stage = the boxes of all covering
// patches of one plane packed back to back into one ring slot (13.3 instead of 16.5 MB of LDS-DMA per plane at cfg 1), the blend of a pixel
// a loop over patches inside the stage.  Same bits.  B = 8, 18 x 256^2: 19.3 us against 18.4 us for the patch-by-patch kernel above (17.2 vs
// 15.9 with the balanced block order), 46.9 vs 33.3 us at nrows = 6: its per-block phases (tools/trace_resample.py) were loads 1.3 us, DMA
// addresses + ring fill 1.4, taps 1.4, the 8 stages 4.8, stores 0.8 for a 2-patch tile and still grew by 2.5 us per patch — a block's time is
// the WORK per patch (bytes, taps, geometry) divided by a throughput 16 waves per CU share, not the number of dependent pipelines.
// What the same traces did show: the CU on a patch seam had 56 patch-tiles to the median CU's 34 — see the block order in omni_p2e_build_tiles.
// ------------------------------------------------------------------ backward
// g_pers[b,c,y,x,n] = sum over the ERP pixels (i,j) whose tap of patch n is (y,x) of w~ * g_erp[b,c,i,j], w~ the thresholded,
// L1-normalised weights of the forward (the operator is linear in the patches; the weights do not depend on them).
// One thread per ERP pixel, two passes over the candidate patches (normaliser, then scatter); fp32 hardware atomics into a
// zeroed g_pers.
__global__ __launch_bounds__(256) void p2e_bwd_kernel(P2EArgs a /* erp = g_erp (in), pers = g_pers (out) */, int nblocks)
{
    const unsigned lb = omni_xcd_remap(blockIdx.x, nblocks);
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int tx = lb % a.ntx;
    const int i = __builtin_amdgcn_readfirstlane((int)(lb / a.ntx) * 4 + wave);
    const int j = tx * 64 + lane;
    if (i >= a.H) return;
    const bool inside = j < a.W;
    const float2 rt = a.row_trig[i];
    const float2 ct = a.col_trig[inside ? j : a.W - 1];
    const unsigned long long cm_ = a.cand[(size_t)i * a.ntx + tx];
    const unsigned cm_hi = (unsigned)__builtin_amdgcn_readfirstlane((int)(cm_ >> 32));
    const unsigned cm_lo = (unsigned)__builtin_amdgcn_readfirstlane((int)(cm_ & 0xffffffffull));
    const unsigned long long cmask = ((unsigned long long)cm_hi << 32) | (unsigned long long)cm_lo;
    float l1 = 0.0f;
    for (unsigned long long m = cmask; m;) {
        const int n = __builtin_ctzll(m); m &= m - 1;
        Taps t; p2e_taps(a, n, rt.x, rt.y, ct.x, ct.y, t);
        l1 += (t.wa + t.wb) + (t.wc + t.wd);
    }
    if (!inside) return;
    const float rden = 1.0f / fmaxf(l1, 1e-12f);
    const float* gerp = (const float*)a.erp;
    float* gp = (float*)const_cast<void*>(a.pers);
    const size_t erp_plane = (size_t)a.H * a.W, pix = (size_t)i * a.W + j;
    for (unsigned long long m = cmask; m;) {
        const int n = __builtin_ctzll(m); m &= m - 1;
        Taps t; p2e_taps(a, n, rt.x, rt.y, ct.x, ct.y, t);
        if (!((t.wa + t.wb) + (t.wc + t.wd) > 0.0f)) continue;
        const size_t oa = (size_t)n * a.sN + (size_t)t.y0 * a.sY + (size_t)t.x0 * a.sX, ob = (size_t)n * a.sN + (size_t)t.y1 * a.sY + (size_t)t.x0 * a.sX;
        const size_t oc = (size_t)n * a.sN + (size_t)t.y0 * a.sY + (size_t)t.x1 * a.sX, od = (size_t)n * a.sN + (size_t)t.y1 * a.sY + (size_t)t.x1 * a.sX;
        for (int b = 0; b < a.B; ++b)
            for (int c = 0; c < a.C; ++c) {
                const float g = gerp[((size_t)b * a.C + c) * erp_plane + pix] * rden;
                float* q = gp + (size_t)b * a.sB + (size_t)c * a.sC;
                if (t.wa != 0.0f) atomicAdd(q + oa, g * t.wa);
                if (t.wb != 0.0f) atomicAdd(q + ob, g * t.wb);
                if (t.wc != 0.0f) atomicAdd(q + oc, g * t.wc);
                if (t.wd != 0.0f) atomicAdd(q + od, g * t.wd);
            }
    }
}

int fill_args(P2EArgs& a, const omni_geometry* g, const void* pers, const void* pers2, void* erp,
              int B, int C, int layout)
{
    a.pers = pers; a.pers2 = pers2; a.erp = erp;
    a.row_trig = g->row_trig; a.col_trig = g->col_trig; a.cand = g->cand;
    a.B = B; a.C = C; a.H = g->H; a.W = g->W; a.ph = g->ph; a.pw = g->pw; a.ntx = g->ntx;
    const long long N = g->N, ph = g->ph, pw = g->pw;
    if (layout == OMNI_LAYOUT_BCHWN)      { a.sX = N; a.sY = pw * N; a.sN = 1; a.sC = ph * pw * N; a.sB = C * a.sC; }
    else if (layout == OMNI_LAYOUT_BNCHW) { a.sX = 1; a.sY = pw; a.sC = ph * pw; a.sN = C * a.sC; a.sB = N * a.sN; }
    else if (layout == OMNI_LAYOUT_BNHWC) { a.sC = 1; a.sX = C; a.sY = pw * C; a.sN = ph * a.sY; a.sB = N * a.sN; }
    else OMNI_FAIL(OMNI_ERR_INVALID, "omni_pers2equi: unknown layout");
    const float PIf = (float)M_PI, PI2f = (float)(M_PI * 0.5);
    // the reference divides twice in fp32 (new_x / FOV[0] / PI); a reciprocal product differs by <= 1.5 ulp
    a.kx = (float)(1.0 / ((double)(g->fov_w / 360.0f) * (double)PIf));
    a.ky = (float)(1.0 / ((double)(g->fov_h / 180.0f) * (double)PI2f));
    a.half_h = 0.5f * (float)g->ph; a.half_w = 0.5f * (float)g->pw;
    a.tab = g->p2e;
    a.store_nt = omni_options().p2e_store ? 1 : 0;
    a.dbg = 0; a.trace = nullptr;
#ifdef OMNI_DEBUG_BUILD
    a.dbg = omni_debug_bits("OMNI_P2E_DBG");
    a.trace = omni_debug_trace_buf();
#endif
    return OMNI_OK;
}

template <typename T, bool CONF, bool XS1>
void launch_p2e_pl(const P2EArgs& a, int planes, int rows4, int nblocks, hipStream_t stream)
{
    if (planes <= 1)      hipLaunchKernelGGL((p2e_kernel<T, 1, CONF, XS1>), dim3(nblocks), dim3(256), 0, stream, a, rows4, nblocks);
    else if (planes <= 2) hipLaunchKernelGGL((p2e_kernel<T, 2, CONF, XS1>), dim3(nblocks), dim3(256), 0, stream, a, rows4, nblocks);
    else if (planes <= 4) hipLaunchKernelGGL((p2e_kernel<T, 4, CONF, XS1>), dim3(nblocks), dim3(256), 0, stream, a, rows4, nblocks);
    else                  hipLaunchKernelGGL((p2e_kernel<T, 8, CONF, XS1>), dim3(nblocks), dim3(256), 0, stream, a, rows4, nblocks);
}

// ---- LDS path: launch geometry.  One wave per (tile, group of PL planes); LDS per wave = the ring (or the largest stage of the
// geometry if that is larger); option "p2e_nbuf" caps the stages in flight (tuning).
template <typename T, int PL, bool CONF, int NBMAX>
int launch_p2e_lds_nb(const P2EArgs& a, const omni_geometry* g, int p_first, int planes, size_t tensor_bytes, hipStream_t stream)
{
    const auto& tt = g->p2e_tiles[sizeof(T) == 2 ? 1 : 0];
    const int stage_kb = (tt.max_chunks + 63) / 64 * (CONF ? 2 : 1);
    const size_t lds = std::max((size_t)(stage_kb > P2E_RING_KB ? stage_kb : P2E_RING_KB), (size_t)std::max(0, omni_options().p2e_lds_kb)) * 1024;
    hipLaunchKernelGGL((p2e_lds_kernel<T, PL, CONF, NBMAX>), dim3(tt.nslots, planes / PL), dim3(64), lds, stream, a,
                       (const uint2*)tt.ord, g->p2e_tx, g->p2e_ty, (unsigned)tensor_bytes, p_first);
    OMNI_HIP(hipGetLastError());
    return OMNI_OK;
}

template <typename T, int PL, bool CONF>
int launch_p2e_walk_pl(const P2EArgs& a, const omni_geometry* g, int p_first, int planes, size_t tensor_bytes, hipStream_t stream)
{
    if constexpr (PL == 1) {
        // ONE plane per wave: the 8-row tile set where the geometry has one whose boxes fit (option p2e_tile8: 1 (default) | 0: 4-row tiles)
        const auto& t8 = g->p2e_tiles[2 + (sizeof(T) == 2 ? 1 : 0)];
        if (omni_options().p2e_tile8 && t8.walk && t8.max_chunks <= 64 * P2W_NJMAX && P2W_WPB == 1) {
            const size_t lds8 = (size_t)((t8.max_chunks + 63) / 64 * (CONF ? 2 : 1)) * 1024;
            hipLaunchKernelGGL((p2e_walk_kernel<T, 1, CONF, 8>), dim3(t8.nslots_walk, planes), dim3(64), lds8, stream, a,
                               (const unsigned char*)t8.walk, g->p2e_tx, (unsigned)tensor_bytes, p_first, (unsigned)lds8);
            OMNI_HIP(hipGetLastError());
            return OMNI_OK;
        }
    }
    const auto& tt = g->p2e_tiles[sizeof(T) == 2 ? 1 : 0];
    const int stage_kb = (tt.max_chunks + 63) / 64 * (CONF ? 2 : 1);
    // (ONE plane per wave keeps one stage in flight whatever the ring: its LDS is the largest stage of the geometry, 1-2 KiB — up to 32 waves per CU)
    const size_t lds = (size_t)(PL == 1 ? stage_kb : (stage_kb > P2E_RING_KB ? stage_kb : P2E_RING_KB)) * 1024;
    static_assert(256 % P2W_WPB == 0, "nslots is a multiple of 256");
    hipLaunchKernelGGL((p2e_walk_kernel<T, PL, CONF>), dim3(tt.nslots_walk / P2W_WPB, planes / PL), dim3(64 * P2W_WPB), lds * P2W_WPB, stream, a,
                       (const unsigned char*)tt.walk, g->p2e_tx, (unsigned)tensor_bytes, p_first, (unsigned)lds);
    OMNI_HIP(hipGetLastError());
    return OMNI_OK;
}

// the forms of the walk kernel that fit 128 registers WITHOUT scratch (a scratch access inside the counted DMA pipeline would shift every
// s_waitcnt; omnifusion_amd/isa.py fails the build otherwise): 2-byte elements need extra registers for the byte-aligns of their tap pairs
template <typename T, int PL, bool CONF> constexpr bool p2w_fits() { return true; }

template <typename T, int PL, bool CONF>
int launch_p2e_lds_pl(const P2EArgs& a, const omni_geometry* g, int p_first, int planes, size_t tensor_bytes, hipStream_t stream)
{
    // option p2e_walk: 1 (default) the flat-pipeline kernel for waves of ONE or TWO planes — a lone panorama's depth map, BASELINE cfg 3 / cfg 5: 30.9 ->
    // 25.7 us, 110 -> 100 us (fp16), 105 -> 88 us — where its small LDS footprint (one stage) admits 24-32 waves per CU; with 4 / 8 planes per wave the
    // two kernels measure equal at 8 panoramas and p2e_lds_kernel 3 % ahead at 16 (profiles/r04_p2e_walk.txt) | 2: always | 0: never
    if constexpr (p2w_fits<T, PL, CONF>())
    if ((omni_options().p2e_walk == 2 || (omni_options().p2e_walk == 1 && PL <= 2)) && g->p2e_tiles[sizeof(T) == 2 ? 1 : 0].walk && g->p2e_tiles[sizeof(T) == 2 ? 1 : 0].max_chunks <= 64 * P2W_NJMAX &&
        (unsigned long long)a.sY * sizeof(T) < (1ull << 24))          // (its DMA offsets use the full-rate 24-bit multiply on the row pitch)
        return launch_p2e_walk_pl<T, PL, CONF>(a, g, p_first, planes, tensor_bytes, stream);
    int nb = omni_options().p2e_nbuf;
    if (nb <= 0) nb = 2;   // measured: 2 stages x 16 waves/CU beat 4 stages (18.4 vs 19.3 us at B=8 18x256^2)
    if (nb >= 4) return launch_p2e_lds_nb<T, PL, CONF, 4>(a, g, p_first, planes, tensor_bytes, stream);
    if (nb >= 2) return launch_p2e_lds_nb<T, PL, CONF, 2>(a, g, p_first, planes, tensor_bytes, stream);
    return launch_p2e_lds_nb<T, PL, CONF, 1>(a, g, p_first, planes, tensor_bytes, stream);
}

template <typename T, bool CONF>
int launch_p2e_lds(const P2EArgs& a, const omni_geometry* g, int planes, size_t tensor_bytes, hipStream_t stream)
{
    // planes per wave: 8, 4, 2 or 1 (the tap geometry is evaluated once per wave and patch and amortised over them); the groups go to
    // blockIdx.y.  A plane count that is not a multiple of 8 runs as up to four launches over consecutive plane ranges (7 = 4 + 2 + 1)
    // instead of falling to one plane per wave, where the geometry is as much work as the blend.
    // (CONF, 8 planes: 2 x 16 accumulators spill at 128 VGPRs, and a scratch access inside the DMA pipeline would shift every hand-counted
    //  s_waitcnt vmcnt(N) — that form is not instantiated at all; omnifusion_amd/isa.py checks that no counted-wait kernel has scratch)
    constexpr int CAPMAX = CONF ? 4 : 8;
    const int cap = std::min(CAPMAX, omni_options().p2e_planes > 0 ? omni_options().p2e_planes : CAPMAX);
    int p = 0;
    while (p < planes) {
        const int left = planes - p;
        int rc = OMNI_OK;
        if (left >= 8 && cap >= 8)      { const int n = left / 8 * 8; if constexpr (!CONF) rc = launch_p2e_lds_pl<T, 8, CONF>(a, g, p, n, tensor_bytes, stream); p += n; }
        else if (left >= 4 && cap >= 4) { const int n = left / 4 * 4; rc = launch_p2e_lds_pl<T, 4, CONF>(a, g, p, n, tensor_bytes, stream); p += n; }
        else if (left >= 2 && cap >= 2) { const int n = left / 2 * 2; rc = launch_p2e_lds_pl<T, 2, CONF>(a, g, p, n, tensor_bytes, stream); p += n; }
        else                            { rc = launch_p2e_lds_pl<T, 1, CONF>(a, g, p, left, tensor_bytes, stream); p += left; }
        if (rc != OMNI_OK) return rc;
    }
    return OMNI_OK;
}

template <typename T, bool CONF>
int launch_p2e(const omni_geometry* g, const void* pers, const void* pers2, void* erp, int B, int C,
               int layout, hipStream_t stream)
{
    P2EArgs a;
    int rc = fill_args(a, g, pers, pers2, erp, B, C, layout);
    if (rc != OMNI_OK) return rc;
    if ((long long)g->N * C * g->ph * g->pw >= (1ll << 31))
        OMNI_FAIL(OMNI_ERR_UNSUPPORTED, "omni_pers2equi: one batch item of the patch tensor must hold < 2^31 elements");
    const int planes = CONF ? B : B * C;
    // LDS-staged path: planar patches whose rows are whole 16-byte chunks, boxes that fit a slot, 32-bit element offsets
    const auto& tt = g->p2e_tiles[sizeof(T) == 2 ? 1 : 0];
    constexpr int EPC = 16 / (int)sizeof(T);
    const bool aligned = ((uintptr_t)pers % 16 == 0) && (!pers2 || (uintptr_t)pers2 % 16 == 0) &&
                         g->pw % EPC == 0 && (g->ph * g->pw) % EPC == 0;
    const long long tensor_bytes = (long long)B * g->N * C * g->ph * g->pw * (long long)sizeof(T);
    // (ONE plane — a lone panorama's depth map, BASELINE cfg 3 / cfg 5 — has nothing to amortise the boxes over: the launch is bound by the tap
    //  geometry of its (pixel, patch) pairs, ~100 vector instructions each, and the direct gathers are the shorter program: cfg 3 31.2 vs 33.6 us,
    //  cfg 5 fp16 108 vs 123 us; at 512x1024 the two are equal)
    // (round 4: with the flat-pipeline kernel the LDS path wins for one plane too; the gathers remain for option p2e_walk = 0)
    const bool single = planes == 1 && (long long)g->H * g->W >= (1ll << 20) && omni_options().p2e_gather != 2 &&
                        !(omni_options().p2e_walk && tt.walk && tt.max_chunks <= 64 * P2W_NJMAX);
    if (a.sX == 1 && tt.ok && aligned && omni_options().p2e_gather != 1 && !single && tensor_bytes < (1ll << 31))   // 32-bit buffer offsets
        return launch_p2e_lds<T, CONF>(a, g, planes, (size_t)tensor_bytes, stream);
    const int rows4 = (g->H + 3) / 4;
    const int nblocks = omni_xcd_rows_grid(rows4, g->ntx);
    if (a.sX == 1 && g->pw >= 2) launch_p2e_pl<T, CONF, true>(a, planes, rows4, nblocks, stream);
    else                         launch_p2e_pl<T, CONF, false>(a, planes, rows4, nblocks, stream);
    OMNI_HIP(hipGetLastError());
    return OMNI_OK;
}

int check_common(const omni_geometry* g, int B, int C, const char* who)
{
    if (!g) OMNI_FAIL(OMNI_ERR_INVALID, std::string(who) + ": null geometry");
    if (B < 0 || C < 0) OMNI_FAIL(OMNI_ERR_INVALID, std::string(who) + ": negative batch/channels");
    if (g->H < 1 || g->W < 1) OMNI_FAIL(OMNI_ERR_INVALID, std::string(who) + ": empty ERP size");
    return OMNI_OK;
}
}  // namespace

int omni_p2e_build_candidates(omni_geometry* g, hipStream_t stream)
{
    P2EArgs a;
    int rc = fill_args(a, g, nullptr, nullptr, nullptr, 0, 1, OMNI_LAYOUT_BNCHW);
    if (rc != OMNI_OK) return rc;
    const int waves = g->H * g->ntx;
    hipLaunchKernelGGL(p2e_candidates_kernel, dim3((waves + 3) / 4), dim3(256), 0, stream, a, g->cand);
    OMNI_HIP(hipGetLastError());
    // one-time setup: make the table visible to every stream that may use this handle later
    OMNI_HIP(hipStreamSynchronize(stream));
    return OMNI_OK;
}

// Per-tile box tables of the LDS path (one per element size: the 16-byte chunk alignment differs).  One-time setup.
int omni_p2e_build_tiles(omni_geometry* g, hipStream_t stream)
{
    P2EArgs a;
    int rc = fill_args(a, g, nullptr, nullptr, nullptr, 0, 1, OMNI_LAYOUT_BNCHW);
    if (rc != OMNI_OK) return rc;
    g->p2e_tx = (g->W + P2E_TW - 1) / P2E_TW; g->p2e_ty = (g->H + P2E_TH - 1) / P2E_TH;
    if ((long long)g->p2e_tx * g->p2e_ty >= (1ll << 28)) return OMNI_OK;                     // absurd sizes: gather path only
    int* dstats = nullptr;
    OMNI_HIP(hipMalloc((void**)&dstats, 3 * sizeof(int)));
    // sets 0 / 1: P2E_TH-row tiles (4- / 2-byte elements), what every LDS kernel reads; sets 2 / 3: 8-row tiles for the one-plane walk kernel (its slot
    // table only; built when the 4-row set of the element size exists)
    for (int e = 0; e < 4; ++e) {
        auto& tt = g->p2e_tiles[e];
        const int epc = (e & 1) ? 8 : 4, TH = e < 2 ? P2E_TH : 8;
        if (e >= 2 && (!g->p2e_tiles[e - 2].ok || !omni_options().p2e_tile8)) continue;
        const int ty_set = (g->H + TH - 1) / TH;
        const long long ntiles = (long long)g->p2e_tx * ty_set;
        if (hipMalloc((void**)&tt.ent, sizeof(uint2) * (size_t)ntiles * P2E_MAXC) != hipSuccess) { (void)hipFree(dstats); OMNI_FAIL(OMNI_ERR_HIP, "omni_p2e_build_tiles: out of memory"); }
        if (hipMemsetAsync(dstats, 0, 3 * sizeof(int), stream) != hipSuccess) { (void)hipFree(dstats); OMNI_FAIL(OMNI_ERR_HIP, "omni_p2e_build_tiles: memset"); }
        if (TH == 8) hipLaunchKernelGGL(p2e_tiles_kernel<8>, dim3((unsigned)((ntiles + 3) / 4)), dim3(256), 0, stream, a, tt.ent, g->p2e_tx, (int)ntiles, epc, dstats);
        else         hipLaunchKernelGGL(p2e_tiles_kernel<P2E_TH>, dim3((unsigned)((ntiles + 3) / 4)), dim3(256), 0, stream, a, tt.ent, g->p2e_tx, (int)ntiles, epc, dstats);
        int hs[3] = {0, 0, 0};
        if (hipGetLastError() != hipSuccess || hipMemcpyAsync(hs, dstats, sizeof(hs), hipMemcpyDeviceToHost, stream) != hipSuccess ||
            hipStreamSynchronize(stream) != hipSuccess) { (void)hipFree(dstats); OMNI_FAIL(OMNI_ERR_HIP, "omni_p2e_build_tiles: kernel failed"); }
        tt.max_chunks = hs[0]; tt.max_cand = hs[1];
        tt.ok = (hs[0] <= P2E_MAX_CHUNKS && hs[1] <= P2E_MAXC && g->pw % epc == 0) ? 1 : 0;
        tt.sum_chunks = hs[2];
        if (tt.ok) {
            // ---- block order.  A block's duration grows with the number of covering patches of its tile (3.3 us + 2.9 us per patch at
            // cfg 1, tools/trace_resample.py), all blocks of a BASELINE-size launch are resident at once, and the dispatcher deals an XCD's
            // blocks to its 32 CUs round-robin (block b -> XCD b % 8, CU (b / 8) % 32 of it: tools/trace_cu.py) — with 32 tiles per ERP
            // row every CU got ONE column strip of the image and the CU on a patch seam 56 patch-tiles where the median CU has 34; the
            // launch ended when that CU did (17.4 us for blocks of 9.8 us on average).  So: whole bands of tile rows per XCD as before
            // (vertical neighbours share their boxes in one L2), bands dealt to the XCDs by cost (heaviest with lightest), and inside an
            // XCD the tiles sorted by cost and dealt to the 32 round-robin positions in snake order.  Pure speed: any order is correct.
            std::vector<uint2> he((size_t)ntiles * P2E_MAXC);
            if (hipMemcpy(he.data(), tt.ent, sizeof(uint2) * he.size(), hipMemcpyDeviceToHost) != hipSuccess) { (void)hipFree(dstats); OMNI_FAIL(OMNI_ERR_HIP, "omni_p2e_build_tiles: copy"); }
            const int tx = g->p2e_tx, ty = ty_set, band = omni_options().p2e_band > 0 ? omni_options().p2e_band : std::max(1, ty / 8), nbands = (ty + band - 1) / band;   // (one contiguous range of tile rows per XCD: 15.8 us, FETCH 52 MB; 8-row bands 16.0, 4-row 17.2 / 61 MB, 2-row 19.6 / 85 MB)
            auto cost = [&](int wid) { return 2 + (int)(he[(size_t)wid * P2E_MAXC].x >> 26); };
            std::vector<std::pair<long long, int>> bc(nbands);
            for (int b = 0; b < nbands; ++b) {
                long long c = 0;
                for (int r = b * band; r < std::min(ty, (b + 1) * band); ++r) for (int x = 0; x < tx; ++x) c += cost(r * tx + x);
                bc[b] = {-c, b};
            }
            std::sort(bc.begin(), bc.end());
            std::vector<std::vector<int>> per(8);
            for (int k = 0; k < nbands; ++k) {
                const int r = k / 8, i = k % 8, xcd = (r & 1) ? 7 - i : i, b = bc[k].second;
                for (int row = b * band; row < std::min(ty, (b + 1) * band); ++row) for (int x = 0; x < tx; ++x) per[xcd].push_back(row * tx + x);
            }
            size_t mx = 0;
            for (auto& v : per) {
                std::stable_sort(v.begin(), v.end(), [&](int p, int q) { return cost(p) > cost(q); });
                mx = std::max(mx, v.size());
            }
            const size_t rounds = (mx + 31) / 32;
            tt.nslots = (int)(rounds * 32 * 8);
            std::vector<uint2> ord((size_t)tt.nslots * 4 * P2E_REC, make_uint2(0u, 0u));       // (a 32-byte record = 4 uint2)
            for (int s2 = 0; s2 < tt.nslots; ++s2) ord[(size_t)s2 * 4 * P2E_REC].x = 0xffffffffu;
            auto fbits = [](float f) { unsigned u; memcpy(&u, &f, 4); return u; };
            for (int xcd = 0; xcd < 8; ++xcd)
                for (size_t k = 0; k < per[xcd].size(); ++k) {
                    const size_t r = k / 32, i = k % 32, pos = r * 32 + ((r & 1) ? 31 - i : i);
                    const size_t slot = pos * 8 + (size_t)xcd;
                    const int wid = per[xcd][k];
                    uint2* rec = ord.data() + slot * 4 * P2E_REC;
                    const int cnt = (int)(he[(size_t)wid * P2E_MAXC].x >> 26);
                    rec[0] = make_uint2((unsigned)wid, (unsigned)cnt);
                    for (int c = 0; c < P2E_MAXC; ++c) {
                        const uint2 e2 = he[(size_t)wid * P2E_MAXC + c];
                        const int n = (int)(e2.x & 63u);
                        rec[4 * (c + 1) + 0] = e2;
                        rec[4 * (c + 1) + 1] = make_uint2(fbits(g->p2e.slam[n]), fbits(g->p2e.clam[n]));
                        rec[4 * (c + 1) + 2] = make_uint2(fbits(g->p2e.sphi[n]), fbits(g->p2e.cphi[n]));
                    }
                }
            if (e < 2 && (hipMalloc((void**)&tt.ord, sizeof(uint2) * ord.size()) != hipSuccess ||
                hipMemcpy(tt.ord, ord.data(), sizeof(uint2) * ord.size(), hipMemcpyHostToDevice) != hipSuccess)) { (void)hipFree(dstats); OMNI_FAIL(OMNI_ERR_HIP, "omni_p2e_build_tiles: order table"); }
            // ---- the same slots for p2e_walk_kernel: header {tile | -1, covering patches, pieces per stage (tile-uniform)}, the patch records, the tile's trig
            {
                std::vector<float2> hrow((size_t)g->H), hcol((size_t)g->W);
                if (hipMemcpy(hrow.data(), g->row_trig, sizeof(float2) * hrow.size(), hipMemcpyDeviceToHost) != hipSuccess ||
                    hipMemcpy(hcol.data(), g->col_trig, sizeof(float2) * hcol.size(), hipMemcpyDeviceToHost) != hipSuccess) { (void)hipFree(dstats); OMNI_FAIL(OMNI_ERR_HIP, "omni_p2e_build_tiles: trig copy"); }
                // block b of the walk kernel = P2W_WPB waves = the slots WPB b .. WPB b + WPB - 1, all on CU (b / 8) % 32 of XCD b % 8: the tile that
                // the one-wave-per-block order gives to (XCD x, CU c, round r) keeps its CU — slot WPB (((r / WPB) 32 + c) 8 + x) + r % WPB
                const size_t rounds_w = (rounds + P2W_WPB - 1) / P2W_WPB * P2W_WPB;
                tt.nslots_walk = (int)(rounds_w * 32 * 8);
                std::vector<unsigned char> wt((size_t)tt.nslots_walk * P2W_SLOT, 0);
                for (int s2 = 0; s2 < tt.nslots_walk; ++s2) { const unsigned m1 = 0xffffffffu; memcpy(wt.data() + (size_t)s2 * P2W_SLOT, &m1, 4); }
                int hist[P2E_NJMAX + 1] = {0};
                for (int s2 = 0; s2 < tt.nslots; ++s2) {
                    const uint2* rec = ord.data() + (size_t)s2 * 4 * P2E_REC;
                    const size_t x8 = (size_t)s2 % 8, pos = (size_t)s2 / 8, rr = pos / 32, cu = pos % 32;
                    unsigned char* dst = wt.data() + ((((rr / P2W_WPB) * 32 + cu) * 8 + x8) * P2W_WPB + rr % P2W_WPB) * P2W_SLOT;
                    const int wid = (int)rec[0].x, cnt = (int)rec[0].y;
                    int njt = 1;
                    if (wid >= 0)
                        for (int c = 0; c < cnt && c < P2E_MAXC; ++c) {
                            const unsigned e0 = rec[4 * (c + 1)].x;
                            const int nchunk = (int)((e0 >> 6) & 1023) * (int)((e0 >> 16) & 1023);
                            njt = std::max(njt, (nchunk + 63) / 64);
                        }
                    if (wid >= 0) ++hist[std::min(njt, P2E_NJMAX)];
                    const unsigned hdr[8] = {(unsigned)wid, (unsigned)cnt, (unsigned)njt, 0u, 0u, 0u, 0u, 0u};
                    memcpy(dst, hdr, 32);
                    memcpy(dst + P2W_OFF_PATCH, rec + 4, 32 * P2E_MAXC);
                    if (wid >= 0) {
                        const int ti = wid / tx, tj = wid - ti * tx;
                        float2* tg = reinterpret_cast<float2*>(dst + P2W_OFF_TRIG);
                        for (int cc = 0; cc < P2E_TW; ++cc) tg[cc] = hcol[(size_t)std::min(tj * P2E_TW + cc, g->W - 1)];
                        for (int r = 0; r < TH; ++r) tg[P2E_TW + r] = hrow[(size_t)std::min(ti * TH + r, g->H - 1)];
                    }
                }
                if (omni_options().e2p_verbose)
                    fprintf(stderr, "[omni] pers2equi %dx%d <- %dx%d, %d-byte elements: tiles by KiB pieces per stage (largest box of the tile): 1:%d 2:%d 3:%d 4:%d 5:%d 6:%d 7:%d 8:%d\n",
                            g->H, g->W, g->ph, g->pw, 16 / epc, hist[1], hist[2], hist[3], hist[4], hist[5], hist[6], hist[7], hist[8]);
                if (hipMalloc((void**)&tt.walk, wt.size()) != hipSuccess ||
                    hipMemcpy(tt.walk, wt.data(), wt.size(), hipMemcpyHostToDevice) != hipSuccess) { (void)hipFree(dstats); OMNI_FAIL(OMNI_ERR_HIP, "omni_p2e_build_tiles: walk table"); }
            }
        }
        (void)hipFree(tt.ent); tt.ent = nullptr;                     // the kernels read only the ordered table (tt.ord)
        if (omni_options().e2p_verbose)
            fprintf(stderr, "[omni] pers2equi %dx%d <- %d patches %dx%d, %d-byte elements: largest tap box %d chunks, <= %d patches and <= %d chunks per %dx%d tile -> %s\n",
                    g->H, g->W, g->N, g->ph, g->pw, 16 / epc, hs[0], hs[1], hs[2], TH, P2E_TW, tt.ok ? "LDS path" : "gather path");
    }
    (void)hipFree(dstats);
    return OMNI_OK;
}

// ------------------------------------------------------------------ reference layout -> planar
// [B,C,ph,pw,N] (N innermost: the reference's stack(dim=-1) / unfold layout) -> [B,N,C,ph,pw].  A blend over the N-innermost tensor
// touches every cache line of a patch region once per covering patch (18 patches share each line): 94.6 us for 8 x 18 x 256^2 where the
// LDS-staged kernel on planar patches takes 18.5.  Converting first costs one coalesced pass (256 pixels x N elements per block in,
// N runs of 256 elements out, transposed in LDS with an odd pitch): the drop-in pers2equi() runs both.
namespace {
template <typename T>
__global__ __launch_bounds__(256) void patches_to_planar_kernel(const T* __restrict__ src, T* __restrict__ dst, int C, int N, int pp, int tiles)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char p2p_raw[];
    T* tile = reinterpret_cast<T*>(p2p_raw);                     // [256][N + 1]
    const int bc = blockIdx.x / tiles, p0 = (blockIdx.x % tiles) * 256;
    const int b = bc / C, c = bc % C;
    const int npx = min(256, pp - p0), run = npx * N, pitch = N + 1;
    const T* s = src + ((size_t)bc * pp + p0) * N;
    for (int i = threadIdx.x; i < run; i += 256) tile[(i / N) * pitch + i % N] = s[i];
    __syncthreads();
    if ((int)threadIdx.x < npx)
        for (int n = 0; n < N; ++n)
            dst[(((size_t)b * N + n) * C + c) * pp + p0 + threadIdx.x] = tile[threadIdx.x * pitch + n];
}
}  // namespace

extern "C" int omni_patches_to_planar(const void* src, void* dst, int dtype, int B, int C, int ph, int pw, int N, omni_stream_t stream)
{
    if (B < 0 || C < 0 || ph <= 0 || pw <= 0 || N <= 0 || N > 64) OMNI_FAIL(OMNI_ERR_INVALID, "omni_patches_to_planar: bad shape (1..64 patches)");
    if (B == 0 || C == 0) return OMNI_OK;
    if (!src || !dst) OMNI_FAIL(OMNI_ERR_INVALID, "omni_patches_to_planar: null device pointer");
    const int pp = ph * pw, tiles = (pp + 255) / 256;
    if ((long long)B * C * tiles >= (1ll << 31)) OMNI_FAIL(OMNI_ERR_UNSUPPORTED, "omni_patches_to_planar: too many tiles");
    const dim3 grid((unsigned)(B * C * tiles));
    if (dtype == OMNI_F32)      hipLaunchKernelGGL(patches_to_planar_kernel<float>, grid, dim3(256), 256 * (N + 1) * 4, (hipStream_t)stream, (const float*)src, (float*)dst, C, N, pp, tiles);
    else if (dtype == OMNI_F16) hipLaunchKernelGGL(patches_to_planar_kernel<__half>, grid, dim3(256), 256 * (N + 1) * 2, (hipStream_t)stream, (const __half*)src, (__half*)dst, C, N, pp, tiles);
    else OMNI_FAIL(OMNI_ERR_INVALID, "omni_patches_to_planar: dtype must be OMNI_F32 or OMNI_F16");
    OMNI_HIP(hipGetLastError());
    return OMNI_OK;
}

extern "C" int omni_pers2equi_g(const omni_geometry_t* g, const void* pers, void* erp, int dtype, int B, int C,
                                int layout, omni_stream_t stream)
{
    int rc = check_common(g, B, C, "omni_pers2equi");
    if (rc != OMNI_OK) return rc;
    if (B == 0 || C == 0) return OMNI_OK;
    if (!pers || !erp) OMNI_FAIL(OMNI_ERR_INVALID, "omni_pers2equi: null device pointer");
    if (dtype == OMNI_F32) return launch_p2e<float, false>(g, pers, nullptr, erp, B, C, layout, (hipStream_t)stream);
    if (dtype == OMNI_F16) return launch_p2e<__half, false>(g, pers, nullptr, erp, B, C, layout, (hipStream_t)stream);
    OMNI_FAIL(OMNI_ERR_INVALID, "omni_pers2equi: dtype must be OMNI_F32 or OMNI_F16");
}

extern "C" int omni_pers2equi(const void* pers, void* erp, int dtype, int B, int C, int ph, int pw,
                              int H, int W, int nrows, float fov_h, float fov_w, int layout,
                              omni_stream_t stream)
{
    const omni_geometry* g = nullptr;
    int rc = omni_geometry_lookup(&g, nrows, fov_h, fov_w, ph, pw, H, W, (hipStream_t)stream);
    if (rc != OMNI_OK) return rc;
    return omni_pers2equi_g(g, pers, erp, dtype, B, C, layout, stream);
}

extern "C" int omni_pers2equi_conf(const void* pred_w, const void* conf, float* out, int dtype, int B,
                                   int ph, int pw, int H, int W, int nrows, float fov_h, float fov_w,
                                   int layout, omni_stream_t stream)
{
    const omni_geometry* g = nullptr;
    int rc = omni_geometry_lookup(&g, nrows, fov_h, fov_w, ph, pw, H, W, (hipStream_t)stream);
    if (rc != OMNI_OK) return rc;
    rc = check_common(g, B, 1, "omni_pers2equi_conf");
    if (rc != OMNI_OK) return rc;
    if (B == 0) return OMNI_OK;
    if (!pred_w || !conf || !out) OMNI_FAIL(OMNI_ERR_INVALID, "omni_pers2equi_conf: null device pointer");
    if (dtype == OMNI_F32) return launch_p2e<float, true>(g, pred_w, conf, out, B, 1, layout, (hipStream_t)stream);
    if (dtype == OMNI_F16) return launch_p2e<__half, true>(g, pred_w, conf, out, B, 1, layout, (hipStream_t)stream);
    OMNI_FAIL(OMNI_ERR_INVALID, "omni_pers2equi_conf: dtype must be OMNI_F32 or OMNI_F16");
}

// Vector-Jacobian product of pers2equi w.r.t. the patches: grad_erp [B,C,H,W] -> grad_pers in the layout of the forward's
// input (overwritten).  fp32 only.  Replaces what autograd derives from the advanced-indexing gathers of
// pers2equi_v3.py:174-196 in the reference's training scripts.
namespace {
// ---- backward by gathers (no global atomics, nothing to zero).  The scatter kernel above issues 4 global atomics per (ERP pixel, covering
// patch, plane): 35 M of them at B = 8, 18 x 256^2 — 1.28 ms, bound by the L2 atomic rate.  Transposed, every PATCH pixel is the sum over the ERP
// pixels whose bilinear taps touch it, and patch tiles are disjoint: one wave owns a 4 x 32 tile of one patch, walks the ERP box of the
// pixels that can touch it (a constant of the geometry, built once with the SAME tap function — exact superset), evaluates their taps
// for this patch, and accumulates the ones that fall into its tile in LDS (ds_add_f32: order within the wave's own instruction stream);
// then it writes the tile once, coalesced.  An ERP pixel is visited by every tile its taps touch (1-4 per covering patch), so the tap
// geometry is evaluated ~2.5x as often as in the forward; the L1 normaliser of a pixel (all covering patches) is a table.
constexpr int P2B_TH = 4, P2B_TW = 32;

__device__ __forceinline__ int p2b_centre_col(const P2EArgs& a, int n)
{
    return (int)((a.tab.lam0[n] + 3.14159265358979f) * (0.5f / 3.14159265358979f) * (float)(a.W - 1) + 0.5f);
}
__device__ __forceinline__ int p2b_wrap(int dx, int W)              // column difference into [-W/2, W - W/2)
{
    const int h = W >> 1;
    dx = dx >= W - h ? dx - W : dx;
    return dx < -h ? dx + W : dx;
}

// one wave per 64 ERP pixels of one row: the L1 normaliser of every pixel and, per (patch, tile), the box of the pixels touching it
__global__ __launch_bounds__(256) void p2e_bwd_box_kernel(P2EArgs a, int* __restrict__ boxes, float* __restrict__ rden, int btx, int bty)
{
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int tx = blockIdx.x % a.ntx;
    const int i = __builtin_amdgcn_readfirstlane((int)(blockIdx.x / a.ntx) * 4 + wave);
    const int j = tx * 64 + lane;
    if (i >= a.H) return;
    const bool inside = j < a.W;
    const float2 rt = a.row_trig[i];
    const float2 ct = a.col_trig[inside ? j : a.W - 1];
    const unsigned long long cm_ = a.cand[(size_t)i * a.ntx + tx];
    const unsigned cm_hi = (unsigned)__builtin_amdgcn_readfirstlane((int)(cm_ >> 32));
    const unsigned cm_lo = (unsigned)__builtin_amdgcn_readfirstlane((int)(cm_ & 0xffffffffull));
    const unsigned long long cmask = ((unsigned long long)cm_hi << 32) | (unsigned long long)cm_lo;
    float l1 = 0.0f;
    for (unsigned long long m = cmask; m;) {
        const int n = __builtin_ctzll(m); m &= m - 1;
        Taps t; p2e_taps(a, n, rt.x, rt.y, ct.x, ct.y, t);
        const float wsum = (t.wa + t.wb) + (t.wc + t.wd);
        l1 += wsum;
        if (!(inside && wsum > 0.0f)) continue;
        const int dx = p2b_wrap(j - p2b_centre_col(a, n), a.W);
        const int xs[2] = {t.x0, t.x1}, ys[2] = {t.y0, t.y1};
        const float w[4] = {t.wa, t.wb, t.wc, t.wd};               // (y0,x0) (y1,x0) (y0,x1) (y1,x1)
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            if (w[k] == 0.0f) continue;
            const int id = (n * bty + ys[k & 1] / P2B_TH) * btx + xs[k >> 1] / P2B_TW;
            atomicMin(boxes + 4 * id + 0, dx); atomicMax(boxes + 4 * id + 1, dx);
            atomicMin(boxes + 4 * id + 2, i);  atomicMax(boxes + 4 * id + 3, i);
        }
    }
    if (inside) rden[(size_t)i * a.W + j] = 1.0f / fmaxf(l1, 1e-12f);
}

// planar [planes][N][pp] -> the reference's [planes][pp][N] (N innermost), 64 samples of all N patches per block through LDS: coalesced
// reads (N runs of 256 bytes) and one contiguous run of 64 N floats out.  (Writing N-innermost straight from the gather kernel puts 4 bytes
// into every 4 N: 260 MB of write traffic for 38 MB at 18 x 256^2.)
__global__ __launch_bounds__(256) void p2e_nlast_kernel(const float* __restrict__ src, float* __restrict__ dst, int N, int pp, int C)
{
    extern __shared__ float nl_tile[];                            // [64][N | 1]
    const int NP = N | 1, t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const int p = blockIdx.y, b = p / C, c = p - b * C, s0 = blockIdx.x * 64, ns = min(64, pp - s0);
    const float* sp = src + ((size_t)b * N * C + c) * pp + s0;    // + n * C * pp
    for (int n = wave; n < N; n += 4)
        if (lane < ns) nl_tile[lane * NP + n] = sp[(size_t)n * C * pp + lane];
    __syncthreads();
    float* dp = dst + ((size_t)p * pp + s0) * N;
    for (int i = t; i < ns * N; i += 256) { const int px = i / N, n = i - px * N; dp[i] = nl_tile[px * NP + n]; }
}

// The transpose as a sparse matrix (omni_spgather.h): every (ERP pixel, covering patch, tap with a non-zero weight) is one entry
// (source = the pixel, weight = w_tap / l1) of the row of the patch pixel the tap reads.  Same traversal and tap function as above.
__global__ __launch_bounds__(256) void p2e_sp_walk_kernel(P2EArgs a, const float* __restrict__ rden, SpEmit b)
{
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int tx = blockIdx.x % a.ntx;
    const int i = __builtin_amdgcn_readfirstlane((int)(blockIdx.x / a.ntx) * 4 + wave);
    const int j = tx * 64 + lane;
    if (i >= a.H || j >= a.W) return;
    const float2 rt = a.row_trig[i], ct = a.col_trig[j];
    const size_t pix = (size_t)i * a.W + j;
    const float r = rden[pix];
    const unsigned long long cm_ = a.cand[(size_t)i * a.ntx + tx];
    const unsigned cm_hi = (unsigned)__builtin_amdgcn_readfirstlane((int)(cm_ >> 32));
    const unsigned cm_lo = (unsigned)__builtin_amdgcn_readfirstlane((int)(cm_ & 0xffffffffull));
    for (unsigned long long m = ((unsigned long long)cm_hi << 32) | (unsigned long long)cm_lo; m;) {
        const int n = __builtin_ctzll(m); m &= m - 1;
        Taps t; p2e_taps(a, n, rt.x, rt.y, ct.x, ct.y, t);
        const int xs[2] = {t.x0, t.x1}, ys[2] = {t.y0, t.y1};
        const float w[4] = {t.wa, t.wb, t.wc, t.wd};               // (y0,x0) (y1,x0) (y0,x1) (y1,x1)
#pragma unroll
        for (int k = 0; k < 4; ++k)
            if (w[k] != 0.0f) sp_emit(b, (n * a.ph + ys[k & 1]) * a.pw + xs[k >> 1], (unsigned)pix, w[k] * r);
    }
}

// NT threads per tile: 64 for the ordinary tiles, 1024 for the few polar ones whose box is whole ERP rows (tens of thousands of pixels)
template <int PL, int NT>
__global__ __launch_bounds__(NT) void p2e_bwd_gather_kernel(P2EArgs a /* erp = g_erp (in), pers = g_pers (out) */, const int4* __restrict__ boxes,
                                                            const float* __restrict__ rden, const int* __restrict__ ids, int btx, int bty, int planes)
{
    __shared__ float acc[PL][P2B_TH * P2B_TW];
    const int lane = threadIdx.x;
    const int id = ids[blockIdx.x], p0 = blockIdx.y * PL;
    const int n = id / (btx * bty), tt = id - n * (btx * bty);
    const int ty0 = (tt / btx) * P2B_TH, tx0 = (tt % btx) * P2B_TW;
#pragma unroll
    for (int p = 0; p < PL; ++p)
        for (int e = lane; e < P2B_TH * P2B_TW; e += NT) acc[p][e] = 0.0f;
    if (NT > 64) __syncthreads();
    const int4 box = boxes[id];                                    // dx min, dx max, row min, row max
    const float* gerp = (const float*)a.erp;
    const size_t erp_plane = (size_t)a.H * a.W;
    if (box.x <= box.y) {
        const int bw = box.y - box.x + 1, npx = bw * (box.w - box.z + 1);
        const int xc = p2b_centre_col(a, n);
        const float rbw = 1.0f / (float)bw;
        for (int base = 0; base < npx; base += NT) {
            const int idx = base + lane;
            const bool in = idx < npx;
            int dy = (int)(((float)idx + 0.5f) * rbw);               // idx / bw (exact for the sizes here, fixed up below)
            int dxi = idx - dy * bw;
            if (dxi < 0) { --dy; dxi += bw; } else if (dxi >= bw) { ++dy; dxi -= bw; }
            const int i = in ? box.z + dy : box.z;
            int j = xc + box.x + (in ? dxi : 0);
            j = j < 0 ? j + a.W : (j >= a.W ? j - a.W : j);
            const float2 rt = a.row_trig[i], ct = a.col_trig[j];
            Taps t; p2e_taps(a, n, rt.x, rt.y, ct.x, ct.y, t);
            const size_t pix = (size_t)i * a.W + j;
            const float r = in ? rden[pix] : 0.0f;
            // tile-relative tap positions; a tap outside my tile belongs to a neighbouring wave
            const int ya = t.y0 - ty0, yb = t.y1 - ty0, xa = t.x0 - tx0, xb = t.x1 - tx0;
            const bool ya_in = (unsigned)ya < (unsigned)P2B_TH, yb_in = (unsigned)yb < (unsigned)P2B_TH;
            const bool xa_in = (unsigned)xa < (unsigned)P2B_TW, xb_in = (unsigned)xb < (unsigned)P2B_TW;
            const float wa = (ya_in && xa_in) ? t.wa * r : 0.0f, wb = (yb_in && xa_in) ? t.wb * r : 0.0f;
            const float wc = (ya_in && xb_in) ? t.wc * r : 0.0f, wd = (yb_in && xb_in) ? t.wd * r : 0.0f;
            if (wa == 0.0f && wb == 0.0f && wc == 0.0f && wd == 0.0f) continue;
#pragma unroll
            for (int p = 0; p < PL; ++p) {
                if (p0 + p >= planes) break;
                const float g = gerp[(size_t)(p0 + p) * erp_plane + pix];
                if (wa != 0.0f) atomicAdd(&acc[p][ya * P2B_TW + xa], g * wa);
                if (wb != 0.0f) atomicAdd(&acc[p][yb * P2B_TW + xa], g * wb);
                if (wc != 0.0f) atomicAdd(&acc[p][ya * P2B_TW + xb], g * wc);
                if (wd != 0.0f) atomicAdd(&acc[p][yb * P2B_TW + xb], g * wd);
            }
        }
    }
    __syncthreads();                                               // the LDS adds of every wave are done
    float* gp = (float*)const_cast<void*>(a.pers);
#pragma unroll
    for (int p = 0; p < PL; ++p) {
        if (p0 + p >= planes) break;
        const size_t pb = (size_t)((p0 + p) / a.C) * a.sB + (size_t)((p0 + p) % a.C) * a.sC + (size_t)n * a.sN;
        for (int e = lane; e < P2B_TH * P2B_TW; e += NT) {
            const int y = ty0 + e / P2B_TW, x = tx0 + e % P2B_TW;
            if (y < a.ph && x < a.pw) gp[pb + (size_t)y * a.sY + (size_t)x * a.sX] = acc[p][e];
        }
    }
}
}  // namespace

int omni_p2e_build_bwd(omni_geometry* g, hipStream_t stream)
{
    P2EArgs a;
    int rc = fill_args(a, g, nullptr, nullptr, nullptr, 1, 1, OMNI_LAYOUT_BNCHW);
    if (rc != OMNI_OK) return rc;
    g->p2e_btx = (g->pw + P2B_TW - 1) / P2B_TW; g->p2e_bty = (g->ph + P2B_TH - 1) / P2B_TH;
    const size_t ntiles = (size_t)g->N * g->p2e_btx * g->p2e_bty;
    if (ntiles == 0 || ntiles >= (1u << 30)) return OMNI_OK;       // no table: the scatter kernel serves this geometry
    OMNI_HIP(hipMalloc((void**)&g->p2e_bwd_box, sizeof(int4) * ntiles));
    OMNI_HIP(hipMalloc((void**)&g->p2e_rden, sizeof(float) * (size_t)g->H * g->W));
    std::vector<int4> init(ntiles, make_int4(0x7fffffff, -0x7fffffff, 0x7fffffff, -0x7fffffff));
    OMNI_HIP(hipMemcpy(g->p2e_bwd_box, init.data(), sizeof(int4) * ntiles, hipMemcpyHostToDevice));
    const int rows4 = (g->H + 3) / 4;
    hipLaunchKernelGGL(p2e_bwd_box_kernel, dim3(rows4 * g->ntx), dim3(256), 0, stream, a, (int*)g->p2e_bwd_box, g->p2e_rden, g->p2e_btx, g->p2e_bty);
    OMNI_HIP(hipGetLastError());
    OMNI_HIP(hipStreamSynchronize(stream));
    std::vector<int4> hb(ntiles);
    OMNI_HIP(hipMemcpy(hb.data(), g->p2e_bwd_box, sizeof(int4) * ntiles, hipMemcpyDeviceToHost));
    std::vector<int> small, big;
    for (size_t t = 0; t < ntiles; ++t) {
        const long long npx = hb[t].x <= hb[t].y ? (long long)(hb[t].y - hb[t].x + 1) * (hb[t].w - hb[t].z + 1) : 0;
        (npx <= 2048 ? small : big).push_back((int)t);
    }
    g->p2e_bwd_nsmall = (int)small.size(); g->p2e_bwd_nbig = (int)big.size();
    if (omni_options().e2p_verbose) {
        long long ps = 0, pb = 0, mx = 0;
        for (size_t t = 0; t < ntiles; ++t) {
            const long long npx = hb[t].x <= hb[t].y ? (long long)(hb[t].y - hb[t].x + 1) * (hb[t].w - hb[t].z + 1) : 0;
            (npx <= 2048 ? ps : pb) += npx; mx = npx > mx ? npx : mx;
        }
        fprintf(stderr, "[omni] pers2equi backward boxes (%dx%d ERP, %dx%d patches): %zu tiles, %d big; box pixels small %lld big %lld, largest %lld\n",
                g->H, g->W, g->ph, g->pw, ntiles, g->p2e_bwd_nbig, ps, pb, mx);
    }
    small.insert(small.end(), big.begin(), big.end());
    OMNI_HIP(hipMalloc((void**)&g->p2e_bwd_ids, sizeof(int) * ntiles));
    OMNI_HIP(hipMemcpy(g->p2e_bwd_ids, small.data(), sizeof(int) * ntiles, hipMemcpyHostToDevice));
    g->p2e_bwd_ok = 1;
    // the sparse-matrix form (the default): rows = patch pixels.  (ERP pixel indices must fit the 24-bit source field.)
    const long long nrows = (long long)g->N * g->ph * g->pw;
    if (nrows < (1ll << 31) && (long long)g->H * g->W <= (1ll << 24)) {
        SpBuilder sb;
        rc = sb.begin(&g->p2e_sp, (int)nrows, stream);
        if (rc != OMNI_OK) return rc;
        hipLaunchKernelGGL(p2e_sp_walk_kernel, dim3(rows4 * g->ntx), dim3(256), 0, stream, a, (const float*)g->p2e_rden, sb.emit(0));
        OMNI_HIP(hipGetLastError());
        OMNI_HIP(hipStreamSynchronize(stream));
        bool fits = false;
        rc = sb.layout((size_t)omni_options().bwd_table_mb << 20, &fits, stream);
        if (rc != OMNI_OK) return rc;
        if (fits) {
            hipLaunchKernelGGL(p2e_sp_walk_kernel, dim3(rows4 * g->ntx), dim3(256), 0, stream, a, (const float*)g->p2e_rden, sb.emit(1));
            OMNI_HIP(hipGetLastError());
            OMNI_HIP(hipStreamSynchronize(stream));
            rc = sb.finish(stream);
            if (rc != OMNI_OK) return rc;
        } else omni_sp_free(g->p2e_sp);
        if (omni_options().e2p_verbose)
            fprintf(stderr, "[omni] pers2equi backward as a sparse matrix: %d rows, %lld entries (%lld with padding) + %d long rows with %lld entries%s\n",
                    g->p2e_sp.nrows, g->p2e_sp.nent, g->p2e_sp.npadded, g->p2e_sp.nlong, g->p2e_sp.nlong_ent, fits ? "" : " -> over the table budget, not kept");
    }
    return OMNI_OK;
}

extern "C" int omni_pers2equi_bwd(const void* grad_erp, void* grad_pers, int dtype, int B, int C, int ph, int pw,
                                  int H, int W, int nrows, float fov_h, float fov_w, int layout, omni_stream_t stream)
{
    if (dtype != OMNI_F32) OMNI_FAIL(OMNI_ERR_UNSUPPORTED, "omni_pers2equi_bwd: fp32 only");
    const omni_geometry* g = nullptr;
    int rc = omni_geometry_lookup(&g, nrows, fov_h, fov_w, ph, pw, H, W, (hipStream_t)stream);
    if (rc != OMNI_OK) return rc;
    rc = check_common(g, B, C, "omni_pers2equi_bwd");
    if (rc != OMNI_OK) return rc;
    if (B == 0 || C == 0) return OMNI_OK;
    if (!grad_erp || !grad_pers) OMNI_FAIL(OMNI_ERR_INVALID, "omni_pers2equi_bwd: null device pointer");
    P2EArgs a;
    rc = fill_args(a, g, grad_pers, nullptr, const_cast<void*>(grad_erp), B, C, layout);
    if (rc != OMNI_OK) return rc;
    {   // first backward of this geometry: build its tables (synchronises the stream once)
        omni_geometry* gm = const_cast<omni_geometry*>(g);
        std::lock_guard<std::mutex> lk(gm->bwd_mu);
        if (!gm->p2e_bwd_tried) {
            gm->p2e_bwd_tried = 1;
            rc = omni_p2e_build_bwd(gm, (hipStream_t)stream);
            if (rc != OMNI_OK) return rc;
        }
    }
    if (g->p2e_sp.ok && omni_options().p2e_bwd_simple == 0 && a.sY == (long long)pw * a.sX) {
        SpApply s;
        s.src = (const float*)grad_erp; s.dst = (float*)grad_pers; s.C = C; s.planes = B * C;
        s.s_sB = (long long)C * H * W; s.s_sC = (long long)H * W; s.s_hi = 0; s.s_lo = 1;
        s.d_sB = a.sB; s.d_sC = a.sC; s.rdiv = ph * pw; s.d_hi = a.sN; s.d_lo = (int)a.sX;
        s.PT = (B * C + 3) / 4 * 4; s.nhi = 1; s.nlo = H * W; s.hi_fastest = 0; s.chunk = 16;
        float* ws = nullptr;
        if (omni_options().bwd_wide) {
            // reference layout: the gathers write the planar form into the scratch, p2e_nlast_kernel turns it N-innermost
            const size_t n1 = (size_t)H * W * s.PT, n2 = layout == OMNI_LAYOUT_BCHWN ? (size_t)B * C * g->N * ph * pw : 0;
            rc = omni_bwd_workspace(const_cast<omni_geometry*>(g), (hipStream_t)stream, (n1 + n2) * sizeof(float), &ws);
            if (rc != OMNI_OK) return rc;
            if (n2) {
                const long long pp = (long long)ph * pw;
                s.dst = ws + n1; s.d_sB = (long long)g->N * C * pp; s.d_sC = pp; s.d_hi = C * pp; s.d_lo = 1;
                rc = sp_apply(g->p2e_sp, s, (hipStream_t)stream, ws);
                if (rc != OMNI_OK) return rc;
                hipLaunchKernelGGL(p2e_nlast_kernel, dim3((unsigned)((pp + 63) / 64), (unsigned)(B * C)), dim3(256), sizeof(float) * 64 * (g->N | 1), (hipStream_t)stream,
                                   (const float*)(ws + n1), (float*)grad_pers, g->N, (int)pp, C);
                OMNI_HIP(hipGetLastError());
                return OMNI_OK;
            }
        }
        return sp_apply(g->p2e_sp, s, (hipStream_t)stream, ws);
    }
    if (g->p2e_bwd_ok && omni_options().p2e_bwd_simple != 1) {
        constexpr int PL = 4;
        const int groups = (B * C + PL - 1) / PL;
        if (g->p2e_bwd_nbig)                                      // first: they are the long ones
            hipLaunchKernelGGL((p2e_bwd_gather_kernel<PL, 1024>), dim3(g->p2e_bwd_nbig, groups), dim3(1024), 0, (hipStream_t)stream, a,
                               (const int4*)g->p2e_bwd_box, (const float*)g->p2e_rden, (const int*)g->p2e_bwd_ids + g->p2e_bwd_nsmall,
                               g->p2e_btx, g->p2e_bty, B * C);
        if (g->p2e_bwd_nsmall)
            hipLaunchKernelGGL((p2e_bwd_gather_kernel<PL, 64>), dim3(g->p2e_bwd_nsmall, groups), dim3(64), 0, (hipStream_t)stream, a,
                               (const int4*)g->p2e_bwd_box, (const float*)g->p2e_rden, (const int*)g->p2e_bwd_ids, g->p2e_btx, g->p2e_bty, B * C);
        OMNI_HIP(hipGetLastError());
        return OMNI_OK;
    }
    OMNI_HIP(hipMemsetAsync(grad_pers, 0, (size_t)B * C * g->N * ph * pw * sizeof(float), (hipStream_t)stream));
    const int rows4 = (g->H + 3) / 4, nblocks = rows4 * g->ntx;
    hipLaunchKernelGGL(p2e_bwd_kernel, dim3(nblocks), dim3(256), 0, (hipStream_t)stream, a, nblocks);
    OMNI_HIP(hipGetLastError());
    return OMNI_OK;
}
