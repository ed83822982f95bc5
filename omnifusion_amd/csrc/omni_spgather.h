// omni_spgather.h — the backward of equi2pers / pers2equi as a CONSTANT SPARSE MATRIX applied by gathers (gfx950).
//
// Both operators are linear maps whose coefficients depend on the geometry only (bilinear tap weights; for pers2equi also the L1
// normaliser of the ERP pixel), so their transposes are sparse matrices that can be written down once per geometry handle:
//   pers2equi^T:  g_pers[n, y, x]  = sum over the ERP pixels whose taps in patch n touch (y, x) of (w_tap / l1) * g_erp[pixel]
//                 (what autograd derives from the advanced-indexing gathers of pers2equi_v3.py:174-196)
//   equi2pers^T:  g_erp[i, j]      = sum over the patch samples whose taps touch (i, j) of w_tap * g_pers[n, h, w]
//                 (ATen grid_sampler_2d_backward, bilinear / border / align_corners=True, equi2pers_v3.py:111)
// ~3.5 and ~9 entries per output element.  The kernels of rounds 2-3 re-derived the taps on every call and reduced them through LDS or global
// atomics (0.32 / 0.74 ms at B = 8, 512 x 1024, 18 x 256^2: 0.17 / 0.22 TB/s); here a call is one pass over the 8-byte entries (coalesced: the
// sliced-ELL layout puts entry k of 64 consecutive rows side by side), one gather per entry and plane from a source that sits in L2, one
// coalesced store per output element — no atomics anywhere, and the summation order is a constant of the geometry (entries sorted by source).
//
// Included by omni_pers2equi.hip and omni_equi2pers.hip (each walks its own taps to emit the entries; everything else is shared).
#pragma once
#include <algorithm>
#include <vector>
#include "omni_internal.h"

namespace {

// how one application addresses its operands: source element of entry e (hi 8 bits | lo 24 bits of e.x) and plane p, destination of row r
struct SpApply {
    const uint2* ent; const int* slice_off; const int* cnt; int nrows, nslices;
    const uint2* long_ent; const int* long_off; const int* long_row; int nlong;
    const float* src; float* dst;
    int C, planes;                        // plane p = (batch p / C, channel p % C)
    long long s_sB, s_sC, d_sB, d_sC;     // element strides of batch / channel in source and destination
    int s_hi, s_lo;                       // source offset inside a plane: (e.x >> 24) * s_hi + (e.x & 0xffffff) * s_lo
    int rdiv; long long d_hi; int d_lo;   // destination offset of row r inside a plane: (r / rdiv) * d_hi + (r % rdiv) * d_lo
    // the plane-interleaved copy of the source (sp_interleave_kernel): record (hi, lo) = PT consecutive floats, one per plane (padded to a multiple of 4)
    int chunk;                            // blocks (of 4 slices) per XCD chunk
    const float* ws; int PT; int nhi, nlo, hi_fastest;   // record index = hi_fastest ? lo * nhi + hi : hi * nlo + lo  (the order the source itself is contiguous in)
};

// One wave per slice of 64 rows, PL planes in registers; the first blocks of the grid take four long rows each (longest first), one per wave (fixed
// partition of a row's entries over the 64 lanes and a fixed shuffle tree: deterministic).  A slice's entries are consumed four at a time: the four table loads,
// then their 4 x PL gathers, are all in flight together (a chain of dependent round trips otherwise: 133 -> 111 us for pers2equi^T);
// a padding slot gathers element 0 (one hot line) and contributes nothing: its VALUE is masked, not its weight, so a non-finite gradient at
// element 0 stays where it is (test_backward_keeps_non_finite_gradients_local).  Measured alternatives: the masked lanes sitting the entry out
// (a branch per entry: the loads serialise, 127 -> 177 us); padding that repeats the row's own first source with weight 0, nothing to mask
// (127 -> 147 us: four in ten slots are padding and then fetch real, scattered lines).
template <int PL>
__global__ __launch_bounds__(256) void sp_gather_kernel(SpApply s, int nlong_blocks, int nslice_blocks)
{
    const int lane = threadIdx.x & 63, p0 = blockIdx.y * PL;
    const float* sp[PL];
#pragma unroll
    for (int p = 0; p < PL; ++p) {
        const int pl = min(p0 + p, s.planes - 1);
        sp[p] = s.src + (size_t)(pl / s.C) * s.s_sB + (size_t)(pl % s.C) * s.s_sC;
    }
    float acc[PL];
#pragma unroll
    for (int p = 0; p < PL; ++p) acc[p] = 0.0f;

    if ((int)blockIdx.x < nlong_blocks) {                          // ---- long rows (first in the grid, longest first): one per wave, entries strided over its lanes, fixed shuffle tree
        const int lr = blockIdx.x * 4 + (threadIdx.x >> 6);
        if (lr >= s.nlong) return;
        const int row = s.long_row[lr], o0 = s.long_off[lr], o1 = s.long_off[lr + 1];
#pragma unroll 4
        for (int i = o0 + lane; i < o1; i += 64) {
            const uint2 en = s.long_ent[i];
            const int off = (int)(en.x >> 24) * s.s_hi + (int)(en.x & 0xffffffu) * s.s_lo;
            const float w = __uint_as_float(en.y);
#pragma unroll
            for (int p = 0; p < PL; ++p) acc[p] = fmaf(sp[p][off], w, acc[p]);
        }
        const size_t doff = (size_t)(row / s.rdiv) * s.d_hi + (size_t)(row % s.rdiv) * s.d_lo;
#pragma unroll
        for (int p = 0; p < PL; ++p) {
            float v = acc[p];
#pragma unroll
            for (int d = 32; d >= 1; d >>= 1) v += __shfl_xor(v, d, 64);
            if (lane == 0 && p0 + p < s.planes) s.dst[(size_t)((p0 + p) / s.C) * s.d_sB + (size_t)((p0 + p) % s.C) * s.d_sC + doff] = v;
        }
        return;
    }

    // (hardware block b runs on XCD b % 8: every XCD gets one contiguous range of slices, so neighbouring rows — which gather the same
    //  source lines — share one L2; in chunks of s.chunk blocks dealt round-robin, because the rows near a pole are the expensive ones)
    unsigned lb = blockIdx.x - nlong_blocks;                      // (nlong_blocks is a multiple of 8: logical block lb still runs on XCD lb % 8)
    {
        const unsigned ch = (unsigned)s.chunk, span = 8u * ch, full = (unsigned)nslice_blocks / span * span;
        if (lb < full) { const unsigned x = lb & 7u, q = lb >> 3; lb = ((q / ch) * 8u + x) * ch + q % ch; }
    }
    const int slice = __builtin_amdgcn_readfirstlane((int)(lb * 4 + (threadIdx.x >> 6)));
    if (slice >= s.nslices) return;
    const int row = slice * 64 + lane;
    const int o0 = s.slice_off[slice], K = s.slice_off[slice + 1] - o0;
    const int nk = row < s.nrows ? s.cnt[row] : -1;
    const uint2* e = s.ent + (size_t)o0 * 64 + lane;
    constexpr int U = 4;
    for (int k0 = 0; k0 < K; k0 += U) {
        uint2 en[U];
#pragma unroll
        for (int u = 0; u < U; ++u) en[u] = k0 + u < K ? e[(size_t)(k0 + u) * 64] : make_uint2(0u, 0u);     // (K is wave-uniform)
        float v[U][PL];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int off = (int)(en[u].x >> 24) * s.s_hi + (int)(en[u].x & 0xffffffu) * s.s_lo;
#pragma unroll
            for (int p = 0; p < PL; ++p) v[u][p] = sp[p][off];
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const bool real = k0 + u < nk;
            const float w = __uint_as_float(en[u].y);
#pragma unroll
            for (int p = 0; p < PL; ++p) acc[p] = fmaf(real ? v[u][p] : 0.0f, w, acc[p]);
        }
    }
    if (nk < 0) return;                                           // past the end, or a long row
    const size_t doff = (size_t)(row / s.rdiv) * s.d_hi + (size_t)(row % s.rdiv) * s.d_lo;
#pragma unroll
    for (int p = 0; p < PL; ++p)
        if (p0 + p < s.planes) s.dst[(size_t)((p0 + p) / s.C) * s.d_sB + (size_t)((p0 + p) % s.C) * s.d_sC + doff] = acc[p];
}

// ---- the same through a plane-interleaved copy of the source.  A 4-byte gather costs the texture path one tag look-up per lane quad and
// line whatever it returns (measured: 28 L1 accesses per wave-level gather, the L1 busy 60 % of the 90 us of pers2equi^T): with the PT planes of
// a source element side by side, ONE 16-byte gather per entry and four planes replaces four.  The copy is one coalesced pass (LDS transposition).
constexpr int SP_ICH = 24;                                         // planes per block of the interleave kernel
__global__ __launch_bounds__(256) void sp_interleave_kernel(SpApply s, float* __restrict__ ws, int R)
{
    __shared__ float tile[256 * (SP_ICH + 1)];
    constexpr int PP = SP_ICH + 1;
    const int t = threadIdx.x, rec0 = blockIdx.x * 256, rec = rec0 + t, c0 = blockIdx.y * SP_ICH;
    const int wc = min(SP_ICH, s.PT - c0), nrec = min(256, R - rec0);
    if (rec < R) {
        int hi, lo;
        if (s.hi_fastest) { lo = rec / s.nhi; hi = rec - lo * s.nhi; } else { hi = rec / s.nlo; lo = rec - hi * s.nlo; }
        const size_t off = (size_t)hi * s.s_hi + (size_t)lo * s.s_lo;
        for (int p4 = 0; p4 < wc; p4 += 4) {                      // (four independent loads in flight, then their LDS writes)
            float v[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int pl = c0 + p4 + u;
                v[u] = pl < s.planes ? s.src[(size_t)(pl / s.C) * s.s_sB + (size_t)(pl % s.C) * s.s_sC + off] : 0.0f;
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) tile[t * PP + p4 + u] = v[u];
        }
    }
    __syncthreads();
    const int q = wc >> 2;                                         // 16-byte pieces per record
    for (int i = t; i < nrec * q; i += 256) {
        const int r = i / q, p = (i - r * q) * 4;
        const float* tp = tile + r * PP + p;
        *reinterpret_cast<float4*>(ws + (size_t)(rec0 + r) * s.PT + c0 + p) = make_float4(tp[0], tp[1], tp[2], tp[3]);
    }
}

template <int PG>                                                  // planes per pass (a multiple of 4)
__global__ __launch_bounds__(256) void sp_gather_wide_kernel(SpApply s, int nlong_blocks, int nslice_blocks)
{
    constexpr int Q = PG / 4;
    const int lane = threadIdx.x & 63, p0 = blockIdx.y * PG;
    const int r_hi = s.hi_fastest ? 1 : s.nlo, r_lo = s.hi_fastest ? s.nhi : 1;
    const float* wsp = s.ws + p0;
    float acc[PG];
#pragma unroll
    for (int p = 0; p < PG; ++p) acc[p] = 0.0f;
    auto rec_ptr = [&](unsigned src) {
        return reinterpret_cast<const float4*>(wsp + (size_t)((int)(src >> 24) * r_hi + (int)(src & 0xffffffu) * r_lo) * s.PT);
    };

    if ((int)blockIdx.x < nlong_blocks) {                          // ---- long rows: one per wave
        const int lr = blockIdx.x * 4 + (threadIdx.x >> 6);
        if (lr >= s.nlong) return;
        const int row = s.long_row[lr], o0 = s.long_off[lr], o1 = s.long_off[lr + 1];
#pragma unroll 4
        for (int i = o0 + lane; i < o1; i += 64) {
            const uint2 en = s.long_ent[i];
            const float4* rp = rec_ptr(en.x);
            const float w = __uint_as_float(en.y);
#pragma unroll
            for (int q = 0; q < Q; ++q) {
                const float4 v = rp[q];
                acc[4 * q] = fmaf(v.x, w, acc[4 * q]); acc[4 * q + 1] = fmaf(v.y, w, acc[4 * q + 1]);
                acc[4 * q + 2] = fmaf(v.z, w, acc[4 * q + 2]); acc[4 * q + 3] = fmaf(v.w, w, acc[4 * q + 3]);
            }
        }
        const size_t doff = (size_t)(row / s.rdiv) * s.d_hi + (size_t)(row % s.rdiv) * s.d_lo;
#pragma unroll
        for (int p = 0; p < PG; ++p) {
            float v = acc[p];
#pragma unroll
            for (int d = 32; d >= 1; d >>= 1) v += __shfl_xor(v, d, 64);
            if (lane == 0 && p0 + p < s.planes) s.dst[(size_t)((p0 + p) / s.C) * s.d_sB + (size_t)((p0 + p) % s.C) * s.d_sC + doff] = v;
        }
        return;
    }

    unsigned lb = blockIdx.x - nlong_blocks;                      // (XCD map as in sp_gather_kernel)
    {
        const unsigned ch = (unsigned)s.chunk, span = 8u * ch, full = (unsigned)nslice_blocks / span * span;
        if (lb < full) { const unsigned x = lb & 7u, q = lb >> 3; lb = ((q / ch) * 8u + x) * ch + q % ch; }
    }
    const int slice = __builtin_amdgcn_readfirstlane((int)(lb * 4 + (threadIdx.x >> 6)));
    if (slice >= s.nslices) return;
    const int row = slice * 64 + lane;
    const int o0 = s.slice_off[slice], K = s.slice_off[slice + 1] - o0;
    const int nk = row < s.nrows ? s.cnt[row] : -1;
    const uint2* e = s.ent + (size_t)o0 * 64 + lane;
    constexpr int U = PG >= 16 ? 2 : 4;                          // entries in flight together (registers: U * PG / 4 float4; 8 for PG = 8 measured: no change)
    for (int k0 = 0; k0 < K; k0 += U) {
        uint2 en[U];
#pragma unroll
        for (int u = 0; u < U; ++u) en[u] = k0 + u < K ? e[(size_t)(k0 + u) * 64] : make_uint2(0u, 0u);     // (K is wave-uniform)
        float4 v[U][Q];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const float4* rp = rec_ptr(en[u].x);
#pragma unroll
            for (int q = 0; q < Q; ++q) v[u][q] = rp[q];
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const bool real = k0 + u < nk;
            const float w = __uint_as_float(en[u].y);
#pragma unroll
            for (int q = 0; q < Q; ++q) {
                acc[4 * q]     = fmaf(real ? v[u][q].x : 0.0f, w, acc[4 * q]);
                acc[4 * q + 1] = fmaf(real ? v[u][q].y : 0.0f, w, acc[4 * q + 1]);
                acc[4 * q + 2] = fmaf(real ? v[u][q].z : 0.0f, w, acc[4 * q + 2]);
                acc[4 * q + 3] = fmaf(real ? v[u][q].w : 0.0f, w, acc[4 * q + 3]);
            }
        }
    }
    if (nk < 0) return;                                           // past the end, or a long row
    const size_t doff = (size_t)(row / s.rdiv) * s.d_hi + (size_t)(row % s.rdiv) * s.d_lo;
#pragma unroll
    for (int p = 0; p < PG; ++p)
        if (p0 + p < s.planes) s.dst[(size_t)((p0 + p) / s.C) * s.d_sB + (size_t)((p0 + p) % s.C) * s.d_sC + doff] = acc[p];
}

// ---- building a table.  The operator's own file walks its taps twice with sp_emit: pass 0 counts the entries of every row, pass 1 deposits
// them (slot = the row's running cursor); sp_sort_kernel then orders every row by source index so that the result does not depend on the
// order the atomics of pass 1 happened to take.
struct SpEmit {
    int pass;                 // 0: count, 1: fill
    int* cnt;                 // pass 0: entries per row; pass 1: the cursor (zeroed again)
    const int* rowpos;        // pass 1: >= 0: index of the row's entry 0 in `ent` (stride 64); < 0: -(1 + index in long_ent) (stride 1)
    uint2* ent; uint2* long_ent;
};
__device__ __forceinline__ void sp_emit(const SpEmit& b, int row, unsigned src, float w)
{
    if (b.pass == 0) { atomicAdd(b.cnt + row, 1); return; }
    const int slot = atomicAdd(b.cnt + row, 1), rp = b.rowpos[row];
    const uint2 e = make_uint2(src, __float_as_uint(w));
    if (rp >= 0) b.ent[(size_t)rp + (size_t)slot * 64] = e;
    else         b.long_ent[(size_t)(-1 - rp) + slot] = e;
}
__device__ __forceinline__ bool sp_after(uint2 a, uint2 b) { return a.x > b.x || (a.x == b.x && a.y > b.y); }

__global__ __launch_bounds__(256) void sp_sort_kernel(uint2* ent, const int* __restrict__ slice_off, const int* __restrict__ cnt, int nrows)
{
    const int row = blockIdx.x * 256 + threadIdx.x;
    if (row >= nrows) return;
    const int n = cnt[row];
    uint2* e = ent + (size_t)slice_off[row >> 6] * 64 + (row & 63);
    for (int i = 1; i < n; ++i) {                                  // insertion sort, n <= the long-row threshold
        const uint2 key = e[(size_t)i * 64];
        int j = i - 1;
        while (j >= 0 && sp_after(e[(size_t)j * 64], key)) { e[(size_t)(j + 1) * 64] = e[(size_t)j * 64]; --j; }
        e[(size_t)(j + 1) * 64] = key;
    }
}

// Host side of the build.  layout(): from the device counts of pass 0 to the allocated (zeroed) table and the rowpos array of pass 1;
// finish(): sorts.  `budget` caps the table size in bytes (a geometry past it keeps the older kernels).
struct SpBuilder {
    OmniSpTable* t; int* d_cnt = nullptr; int* d_rowpos = nullptr; std::vector<int> h_cnt;
    ~SpBuilder() { if (d_cnt) (void)hipFree(d_cnt); if (d_rowpos) (void)hipFree(d_rowpos); }
    // (every memset below is issued ON the build's stream: a null-stream memset is not ordered against a non-blocking stream's kernels)
    int begin(OmniSpTable* table, int nrows, hipStream_t stream)
    {
        t = table; t->nrows = nrows; t->nslices = (nrows + 63) / 64;
        OMNI_HIP(hipMalloc((void**)&d_cnt, sizeof(int) * (size_t)nrows));
        OMNI_HIP(hipMemsetAsync(d_cnt, 0, sizeof(int) * (size_t)nrows, stream));
        return OMNI_OK;
    }
    SpEmit emit(int pass) const { SpEmit e; e.pass = pass; e.cnt = d_cnt; e.rowpos = d_rowpos; e.ent = t->ent; e.long_ent = t->long_ent; return e; }
    // after pass 0 (stream synchronised by the caller): returns OMNI_OK with t->ok == 0 when the table would exceed `budget`
    int layout(size_t budget, bool* fits, hipStream_t stream)
    {
        const int nrows = t->nrows, ns = t->nslices;
        const int lmax = omni_options().bwd_lmax > 0 ? omni_options().bwd_lmax : OMNI_SP_LMAX;
        h_cnt.resize(nrows);
        OMNI_HIP(hipMemcpy(h_cnt.data(), d_cnt, sizeof(int) * (size_t)nrows, hipMemcpyDeviceToHost));
        std::vector<int> so(ns + 1, 0), rowpos(nrows), lrow, loff(1, 0), tcnt(nrows);
        long long nent = 0, nl = 0;
        for (int s = 0; s < ns; ++s) {
            int K = 0;
            for (int r = s * 64; r < std::min(nrows, s * 64 + 64); ++r) if (h_cnt[r] <= lmax) K = std::max(K, h_cnt[r]);
            so[s + 1] = so[s] + K;
            if ((long long)so[s + 1] * 64 >= (1ll << 31)) { *fits = false; return OMNI_OK; }
        }
        for (int r = 0; r < nrows; ++r) {
            if (h_cnt[r] <= lmax) { rowpos[r] = so[r >> 6] * 64 + (r & 63); tcnt[r] = h_cnt[r]; nent += h_cnt[r]; }
            else {
                if (nl + h_cnt[r] >= (1ll << 31) - 1) { *fits = false; return OMNI_OK; }
                rowpos[r] = -1 - (int)nl; tcnt[r] = -1; lrow.push_back(r); nl += h_cnt[r]; loff.push_back((int)nl);
            }
        }
        if (omni_options().e2p_verbose) {
            long long hist[8] = {0, 0, 0, 0, 0, 0, 0, 0};          // rows by entry count: <= 4, 8, 16, 24, 32, 48, 64, more
            const int edge[7] = {4, 8, 16, 24, 32, 48, 64};
            for (int r = 0; r < nrows; ++r) { int b = 0; while (b < 7 && h_cnt[r] > edge[b]) ++b; ++hist[b]; }
            fprintf(stderr, "[omni] sparse rows by entry count (<=4 <=8 <=16 <=24 <=32 <=48 <=64 more): %lld %lld %lld %lld %lld %lld %lld %lld\n",
                    hist[0], hist[1], hist[2], hist[3], hist[4], hist[5], hist[6], hist[7]);
        }
        if (!lrow.empty()) {                                       // longest first (they start first: the launch ends with the slices, not with one wave's 2000 entries)
            std::vector<int> order(lrow.size());
            for (size_t i = 0; i < order.size(); ++i) order[i] = (int)i;
            std::stable_sort(order.begin(), order.end(), [&](int a, int b) { return h_cnt[lrow[a]] > h_cnt[lrow[b]]; });
            std::vector<int> lrow2(lrow.size()), loff2(1, 0);
            long long pos = 0;
            for (size_t i = 0; i < order.size(); ++i) {
                const int r = lrow[order[i]];
                lrow2[i] = r; rowpos[r] = -1 - (int)pos; pos += h_cnt[r]; loff2.push_back((int)pos);
            }
            lrow.swap(lrow2); loff.swap(loff2);
        }
        t->nent = nent; t->npadded = (long long)so[ns] * 64; t->nlong = (int)lrow.size(); t->nlong_ent = nl;
        const size_t bytes = (size_t)(t->npadded + nl) * sizeof(uint2) + (size_t)nrows * 4;
        *fits = bytes <= budget;
        if (!*fits) return OMNI_OK;
        OMNI_HIP(hipMalloc((void**)&t->ent, sizeof(uint2) * (size_t)std::max<long long>(t->npadded, 1)));
        OMNI_HIP(hipMemsetAsync(t->ent, 0, sizeof(uint2) * (size_t)std::max<long long>(t->npadded, 1), stream));
        OMNI_HIP(hipMalloc((void**)&t->long_ent, sizeof(uint2) * (size_t)std::max<long long>(nl, 1)));
        OMNI_HIP(hipMalloc((void**)&t->slice_off, sizeof(int) * (size_t)(ns + 1)));
        OMNI_HIP(hipMalloc((void**)&t->cnt, sizeof(int) * (size_t)nrows));
        OMNI_HIP(hipMalloc((void**)&t->long_off, sizeof(int) * loff.size()));
        OMNI_HIP(hipMalloc((void**)&t->long_row, sizeof(int) * std::max<size_t>(lrow.size(), 1)));
        OMNI_HIP(hipMalloc((void**)&d_rowpos, sizeof(int) * (size_t)nrows));
        OMNI_HIP(hipMemcpy(t->slice_off, so.data(), sizeof(int) * (size_t)(ns + 1), hipMemcpyHostToDevice));
        OMNI_HIP(hipMemcpy(t->cnt, tcnt.data(), sizeof(int) * (size_t)nrows, hipMemcpyHostToDevice));
        OMNI_HIP(hipMemcpy(t->long_off, loff.data(), sizeof(int) * loff.size(), hipMemcpyHostToDevice));
        if (!lrow.empty()) OMNI_HIP(hipMemcpy(t->long_row, lrow.data(), sizeof(int) * lrow.size(), hipMemcpyHostToDevice));
        OMNI_HIP(hipMemcpy(d_rowpos, rowpos.data(), sizeof(int) * (size_t)nrows, hipMemcpyHostToDevice));
        OMNI_HIP(hipMemsetAsync(d_cnt, 0, sizeof(int) * (size_t)nrows, stream));   // the cursors of pass 1
        h_loff.swap(loff);
        return OMNI_OK;
    }
    std::vector<int> h_loff;
    // after pass 1 (stream synchronised by the caller)
    int finish(hipStream_t stream)
    {
        hipLaunchKernelGGL(sp_sort_kernel, dim3((unsigned)((t->nrows + 255) / 256)), dim3(256), 0, stream, t->ent, (const int*)t->slice_off, (const int*)t->cnt, t->nrows);
        OMNI_HIP(hipGetLastError());
        if (t->nlong_ent) {                                        // the few long rows: sorted on the host
            std::vector<uint2> le((size_t)t->nlong_ent);
            OMNI_HIP(hipMemcpy(le.data(), t->long_ent, sizeof(uint2) * le.size(), hipMemcpyDeviceToHost));
            for (int i = 0; i < t->nlong; ++i)
                std::sort(le.begin() + h_loff[i], le.begin() + h_loff[i + 1], [](const uint2& a, const uint2& b) { return a.x < b.x || (a.x == b.x && a.y < b.y); });
            OMNI_HIP(hipMemcpy(t->long_ent, le.data(), sizeof(uint2) * le.size(), hipMemcpyHostToDevice));
        }
        OMNI_HIP(hipStreamSynchronize(stream));
        t->ok = 1;
        return OMNI_OK;
    }
};

// launch.  ws != null: through the plane-interleaved copy (ws holds nhi * nlo records of s.PT floats); else 4-byte gathers from the source
// itself, planes in groups of 8 (12 where the plane count is a multiple of 12).  The long rows ride in the same grid.
inline int sp_apply(const OmniSpTable& t, SpApply s, hipStream_t stream, float* ws)
{
    s.ent = t.ent; s.slice_off = t.slice_off; s.cnt = t.cnt; s.nrows = t.nrows; s.nslices = t.nslices;
    s.long_ent = t.long_ent; s.long_off = t.long_off; s.long_row = t.long_row; s.nlong = t.nlong;
    const int nb = (t.nslices + 3) / 4, nlb = ((t.nlong + 3) / 4 + 7) / 8 * 8;
    if (omni_options().bwd_chunk > 0) s.chunk = omni_options().bwd_chunk;
    if (ws) {
        const int R = s.nhi * s.nlo;
        s.ws = ws;
        hipLaunchKernelGGL(sp_interleave_kernel, dim3((unsigned)((R + 255) / 256), (unsigned)((s.PT + SP_ICH - 1) / SP_ICH)), dim3(256), 0, stream, s, ws, R);
        if (s.PT % 24 == 0)     hipLaunchKernelGGL(sp_gather_wide_kernel<24>, dim3((unsigned)(nb + nlb), (unsigned)(s.PT / 24)), dim3(256), 0, stream, s, nlb, nb);
        else if (s.PT % 16 == 0) hipLaunchKernelGGL(sp_gather_wide_kernel<16>, dim3((unsigned)(nb + nlb), (unsigned)(s.PT / 16)), dim3(256), 0, stream, s, nlb, nb);
        else if (s.PT % 12 == 0) hipLaunchKernelGGL(sp_gather_wide_kernel<12>, dim3((unsigned)(nb + nlb), (unsigned)(s.PT / 12)), dim3(256), 0, stream, s, nlb, nb);
        else if (s.PT % 8 == 0) hipLaunchKernelGGL(sp_gather_wide_kernel<8>, dim3((unsigned)(nb + nlb), (unsigned)(s.PT / 8)), dim3(256), 0, stream, s, nlb, nb);
        else                    hipLaunchKernelGGL(sp_gather_wide_kernel<4>, dim3((unsigned)(nb + nlb), (unsigned)(s.PT / 4)), dim3(256), 0, stream, s, nlb, nb);
        OMNI_HIP(hipGetLastError());
        return OMNI_OK;
    }
    s.ws = nullptr;
    if (s.planes > 8 && s.planes % 12 == 0)
        hipLaunchKernelGGL(sp_gather_kernel<12>, dim3((unsigned)(nb + nlb), (unsigned)(s.planes / 12)), dim3(256), 0, stream, s, nlb, nb);
    else
        hipLaunchKernelGGL(sp_gather_kernel<8>, dim3((unsigned)(nb + nlb), (unsigned)((s.planes + 7) / 8)), dim3(256), 0, stream, s, nlb, nb);
    OMNI_HIP(hipGetLastError());
    return OMNI_OK;
}

}  // namespace
