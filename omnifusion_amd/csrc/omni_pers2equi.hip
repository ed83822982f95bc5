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
#include "omni_internal.h"

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
    PatchTab tab;
};

struct Taps { int x0, x1, y0, y1; float wa, wb, wc, wd; };


// pers2equi_v3.py:112-152 + :191 for one (pixel, patch).  Returns the validity mask.
__device__ __forceinline__ bool p2e_taps(const P2EArgs& a, int n, float slat, float clat, float slon, float clon, Taps& t)
{
    const float sl0 = a.tab.slam[n], cl0 = a.tab.clam[n], sp = a.tab.sphi[n], cp = a.tab.cphi[n];
    const float cd = clon * cl0 + slon * sl0;                       // cos(lon - l0)
    const float sd = slon * cl0 - clon * sl0;                       // sin(lon - l0)
    const float cos_c = sp * slat + cp * clat * cd;                 // :112
    float nx = (clat * sd) / cos_c;                                 // :113
    float ny = (cp * slat - sp * clat * cd) / cos_c;                // :114
    nx = nx * a.kx;                                                 // :115
    ny = ny * a.ky;                                                 // :116
    const float X = (nx + 1.0f) * a.half_h;                         // :122 (sic)
    const float Y = (ny + 1.0f) * a.half_w;                         // :123 (sic)
    const float fw = (float)a.pw, fh = (float)a.ph;
    const bool valid = (X < fw) && (X > 0.0f) && (Y < fh) && (Y > 0.0f) && (cos_c > 0.0f);   // :118,126-127
    const float fx = floorf(X), fy = floorf(Y);                     // :129-132
    const float x0f = fminf(fmaxf(fx, 0.0f), fw - 1.0f), x1f = fminf(fmaxf(fx + 1.0f, 0.0f), fw - 1.0f);   // :134-137
    const float y0f = fminf(fmaxf(fy, 0.0f), fh - 1.0f), y1f = fminf(fmaxf(fy + 1.0f, 0.0f), fh - 1.0f);
    float wa = (x1f - X) * (y1f - Y);                               // :139  tap (y0,x0)
    float wb = (x1f - X) * (Y - y0f);                               // :140  tap (y1,x0)
    float wc = (X - x0f) * (y1f - Y);                               // :141  tap (y0,x1)
    float wd = (X - x0f) * (Y - y0f);                               // :142  tap (y1,x1)
    // :144-147 multiply by mask, :191 zero everything <= 1e-5
    t.wa = (valid && wa > 1e-5f) ? wa : 0.0f;
    t.wb = (valid && wb > 1e-5f) ? wb : 0.0f;
    t.wc = (valid && wc > 1e-5f) ? wc : 0.0f;
    t.wd = (valid && wd > 1e-5f) ? wd : 0.0f;
    t.x0 = (int)x0f; t.x1 = (int)x1f; t.y0 = (int)y0f; t.y1 = (int)y1f;
    return valid;
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
    const unsigned lb = omni_xcd_remap(blockIdx.x, nblocks);
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int tx = lb % a.ntx;
    const int i = __builtin_amdgcn_readfirstlane((int)(lb / a.ntx) * 4 + wave);   // wave-uniform row
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

template <typename T, bool CONF>
int launch_p2e(const omni_geometry* g, const void* pers, const void* pers2, void* erp, int B, int C,
               int layout, hipStream_t stream)
{
    P2EArgs a;
    int rc = fill_args(a, g, pers, pers2, erp, B, C, layout);
    if (rc != OMNI_OK) return rc;
    if ((long long)g->N * C * g->ph * g->pw >= (1ll << 31))
        OMNI_FAIL(OMNI_ERR_UNSUPPORTED, "omni_pers2equi: one batch item of the patch tensor must hold < 2^31 elements");
    const int rows4 = (g->H + 3) / 4;
    const int nblocks = rows4 * g->ntx;
    const int planes = CONF ? B : B * C;
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
    OMNI_HIP(hipMemsetAsync(grad_pers, 0, (size_t)B * C * g->N * ph * pw * sizeof(float), (hipStream_t)stream));
    const int rows4 = (g->H + 3) / 4, nblocks = rows4 * g->ntx;
    hipLaunchKernelGGL(p2e_bwd_kernel, dim3(nblocks), dim3(256), 0, (hipStream_t)stream, a, nblocks);
    OMNI_HIP(hipGetLastError());
    return OMNI_OK;
}
