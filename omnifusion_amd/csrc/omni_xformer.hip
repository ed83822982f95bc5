// omni_xformer.hip — the whole Transformer_cascade (model/spherical_model.py:169-187; Transformer_Block / Attention / Mlp, model/blocks.py:14-89)
// of a (small) batch in ONE launch: 6 x [x += proj(softmax(q k^T 128^-1/2) v); x += fc2(GELU(fc1(LN x)))] + encoder_norm.
//
// Why: the transformer is 1 % of the network's flops and 12 % of a forward's time (8 panoramas, one lane: 48 launches, 364 us; a lone panorama:
// 36 launches, 250 us of its 0.97 ms).  Its GEMMs have 18 .. 144 rows: every launch is a stream of weights (1-4 MB) behind a dependency, and what
// a launch costs is not the kernel boundary (< 1 us in stream order, profiles/r03i_grid_barrier.txt) but the cold start of that stream: the weights of
// GEMM k+1 cannot travel before GEMM k has ended, although they do not depend on it.  Here ONE cooperative grid of XF_GRID blocks walks the phases
//     [LN1 + qkv] | attention | proj (+ residual) | [LN2 + fc1 + GELU] | fc2 (+ residual)          x 6, then encoder_norm,
// separated by device-wide barriers (one agent-scope arrival counter; release / acquire so that the activations cross the XCDs), and every block
// fetches the weight fragments of its NEXT phase BEFORE it arrives at the barrier: they travel while the block waits.
//
// A phase is the arithmetic of the kernels it replaces, element for element — gemm_rows_ln_sh_kernel / gemm_rows_sh_kernel (a block owns 32 output
// channels, its 8 waves split K, fragments straight into registers from the fragment-ordered weights of omni_gemm_rows_pack, the 8 partial tiles meet
// in LDS in wave order), layernorm512_kernel, attention_kernel — applied to row tiles of 32 tokens: the bits of a LONE panorama's forward (which
// runs exactly those kernels), for every panorama of the batch, whatever the batch.
//
// Co-residency: the barrier spins, so all XF_GRID = 64 blocks must be resident together: 512 threads and 100 KiB of LDS each — one per CU, 64 of 256 CUs;
// blocks that find no room wait for other kernels' blocks to END (those never wait for us), so several such grids (pipelined forwards, lanes) cannot
// lock each other out as long as 64 x (forwards in flight) <= 256.  The spin is bounded all the same: a block that waits ~1 s raises a sticky flag
// (omni_transformer_status) and goes on — wrong numbers that are reported, not a hung GPU.
#include <string.h>
#include "omni_internal.h"
#include "omni_sh.h"

namespace {
typedef float f16v __attribute__((ext_vector_type(16)));
typedef float f4v __attribute__((ext_vector_type(4)));
typedef _Float16 h8v __attribute__((ext_vector_type(8)));

constexpr int XF_GRID = 64, XF_THREADS = 512, XF_NW = 8, XF_PITCH = 36, XF_LAYERS = 6;
constexpr int XF_XS_BYTES = 32 * 2048, XF_RED_BYTES = XF_NW * 32 * XF_PITCH * 4;
constexpr int XF_LDS = XF_XS_BYTES + XF_RED_BYTES;                   // 100 KiB; the attention phase's K / V / P tiles (68 KiB) alias it
constexpr unsigned XF_SPIN_LIMIT = 1u << 22;

struct XfLayer {
    const float *ln1g, *ln1b, *ln2g, *ln2b, *bproj, *bfc1, *bfc2;
    const unsigned char *wqkv, *wproj, *wfc1, *wfc2;                 // fragment order (omni_gemm_rows_pack)
};
struct XfArgs {
    XfLayer L[XF_LAYERS];
    const float *encg, *encb;
    float* tok;                                                      // [M, 512] fp32, updated in place
    float* qkv;                                                      // [M, 1536] fp32 scratch
    unsigned char* att;                                              // [M, 512] SH scratch
    unsigned char* hid;                                              // [M, 2048] SH scratch
    float* out;                                                      // [M, 512] fp32: encoder_norm(tok)
    unsigned* sync;                                                  // [0] arrivals, [1] exits: zero between launches
    long long* trace;                                                // tools only (omni_debug_set_trace): wall_clock64 of block 0 before / after every barrier
    int M, N, B;
};

__device__ unsigned xf_timeout_flag = 0;

// device-wide barrier number k (1-based) of this launch: everything this block wrote is visible to every block that leaves it
__device__ __forceinline__ void xf_barrier(unsigned* sync, unsigned k, long long* trace)
{
    __syncthreads();
    if (trace && threadIdx.x == 0 && blockIdx.x == 0) trace[2 * k - 1] = wall_clock64();
    if (threadIdx.x == 0) {
        __hip_atomic_fetch_add(sync, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);      // (release: this XCD's dirty lines are written back first)
        const unsigned target = k * (unsigned)XF_GRID;
        unsigned spins = 0;
        while (__hip_atomic_load(sync, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) {   // (relaxed polls: no cache maintenance while waiting)
            __builtin_amdgcn_s_sleep(2);
            if (++spins > XF_SPIN_LIMIT) { atomicOr(&xf_timeout_flag, 1u); break; }
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");      // ONE acquire: stale lines of this CU's L1 / this XCD's L2 are dropped
        if (trace && blockIdx.x == 0) trace[2 * k] = wall_clock64();
    }
    __syncthreads();
}

// the block's weight fragments of one phase: wave w owns K-steps w * KPW .. + KPW - 1 of column tile `tile`: [K-step][hi kc0, hi kc1, lo kc0, lo kc1]
template <int KPW>
__device__ __forceinline__ void xf_load_w(const unsigned char* __restrict__ wt, int tile, int ksteps, int wave, int lane, h8v (&W)[KPW][4])
{
    const unsigned char* wp = wt + ((size_t)tile * ksteps + (size_t)wave * KPW) * 4096 + lane * 16;
#pragma unroll
    for (int i = 0; i < KPW; ++i)
#pragma unroll
        for (int f = 0; f < 4; ++f) W[i][f] = *reinterpret_cast<const h8v*>(wp + i * 4096 + f * 1024);
}

// LayerNorm(512) of rows row0 .. row0 + 31 (those < M) into xs as split-half rows [32][16 groups][hi32|lo32]: layernorm512_kernel<true>'s expression
__device__ __forceinline__ void xf_ln_tile(const float* __restrict__ tok, const float* __restrict__ lg, const float* __restrict__ lb, float eps,
                                           int row0, int M, unsigned char* xs, int wave, int lane)
{
    for (int row = wave; row < 32 && row0 + row < M; row += XF_NW) {
        const float* p = tok + (size_t)(row0 + row) * 512;
        f4v v0 = *reinterpret_cast<const f4v*>(p + lane * 4), v1 = *reinterpret_cast<const f4v*>(p + 256 + lane * 4);
        float s = (v0.x + v0.y) + (v0.z + v0.w) + (v1.x + v1.y) + (v1.z + v1.w);
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
        const float mean = s * (1.0f / 512.0f);
        v0 -= mean; v1 -= mean;
        float q = (v0.x * v0.x + v0.y * v0.y) + (v0.z * v0.z + v0.w * v0.w) + (v1.x * v1.x + v1.y * v1.y) + (v1.z * v1.z + v1.w * v1.w);
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) q += __shfl_xor(q, o);
        const float rstd = 1.0f / sqrtf(q * (1.0f / 512.0f) + eps);
        const f4v g0 = *reinterpret_cast<const f4v*>(lg + lane * 4), g1 = *reinterpret_cast<const f4v*>(lg + 256 + lane * 4);
        const f4v b0 = *reinterpret_cast<const f4v*>(lb + lane * 4), b1 = *reinterpret_cast<const f4v*>(lb + 256 + lane * 4);
        act_store4<true>(xs, (size_t)row * 512 + lane * 4, v0 * rstd * g0 + b0);
        act_store4<true>(xs, (size_t)row * 512 + 256 + lane * 4, v1 * rstd * g1 + b1);
    }
}

enum { XF_EPI_QKV = 0, XF_EPI_RES = 1, XF_EPI_GELU_SH = 2 };

// One row tile of a GEMM phase: D[32 channels of `tile`][32 rows] = W . x^T over this wave's K slice, the 8 partial tiles summed in wave order, epilogue.
// x: split-half rows with `ksteps` groups of 128 bytes per row (LDS or global), row r of the tile at xrows + r * ksteps * 128.
template <int KPW, int EPI>
__device__ __forceinline__ void xf_gemm_tile(const h8v (&W)[KPW][4], const unsigned char* xrows, int ksteps, int nlive, int row0, int tile, int N,
                                             const float* __restrict__ bias, float* tok, void* dst, float* red, int wave, int lane)
{
    const int r = lane & 31, h = lane >> 5, t = threadIdx.x;
    const bool live = r < nlive;
    const unsigned char* xp = xrows + ((size_t)r * ksteps + (size_t)wave * KPW) * 128 + h * 32;
    f16v acc = (f16v)(0.0f), acc1 = (f16v)(0.0f);
#pragma unroll
    for (int i = 0; i < KPW; ++i)
#pragma unroll
        for (int kc = 0; kc < 2; ++kc) {
            const h8v xh = live ? *reinterpret_cast<const h8v*>(xp + i * 128 + kc * 16) : (h8v)(_Float16)0.0f;
            const h8v xl = live ? *reinterpret_cast<const h8v*>(xp + i * 128 + 64 + kc * 16) : (h8v)(_Float16)0.0f;
            acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(W[i][kc], xh, acc, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(W[i][2 + kc], xh, acc1, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(W[i][kc], xl, acc1, 0, 0, 0);
        }
    // D = W x tokens: column (lane & 31) = token, row (reg & 3) + 8 (reg >> 2) + 4 (lane >> 5) = channel
    float (*rd)[32][XF_PITCH] = reinterpret_cast<float (*)[32][XF_PITCH]>(red);
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        f4v v;
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = fmaf(acc1[4 * q + e], 4.8828125e-4f, acc[4 * q + e]);
        *reinterpret_cast<f4v*>(&rd[wave][r][8 * q + 4 * h]) = v;
    }
    __syncthreads();
    const int tk = t >> 3, c4 = (t & 7) * 4;
    if (t < 256 && tk < nlive) {
        f4v v = *reinterpret_cast<const f4v*>(&rd[0][tk][c4]);
#pragma unroll
        for (int w = 1; w < XF_NW; ++w) v += *reinterpret_cast<const f4v*>(&rd[w][tk][c4]);
        const int col = tile * 32 + c4;
        const size_t o = (size_t)(row0 + tk) * N + col;
        if (EPI != XF_EPI_QKV) v += *reinterpret_cast<const f4v*>(bias + col);
        if (EPI == XF_EPI_RES) v += *reinterpret_cast<const f4v*>(tok + o);
        if (EPI == XF_EPI_GELU_SH) {
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = 0.5f * v[e] * (1.0f + erff(v[e] * 0.70710678118654752440f));
            act_store4<true>(dst, o, v);
        } else act_store4<false>(dst, o, v);
    }
    __syncthreads();                                              // `red` (and, for the LN phases, xs) are free again
}

// softmax(q k^T * scale) v of (panorama b, head hd), N <= 64 tokens, head dim 128: attention_kernel<true>'s arithmetic, a query per wave and trip
__device__ __forceinline__ void xf_attention(const float* __restrict__ qkv, unsigned char* __restrict__ att, int b, int hd, int N, unsigned char* smem, int wave, int lane)
{
    float (*ks)[129] = reinterpret_cast<float (*)[129]>(smem);
    float (*vs)[128] = reinterpret_cast<float (*)[128]>(smem + 64 * 129 * 4 + 64);          // (16-byte aligned: 33 024 + 64)
    float (*ps)[64] = reinterpret_cast<float (*)[64]>(smem + 64 * 129 * 4 + 64 + 64 * 128 * 4);
    const float* q = qkv; const float* kv = qkv + 512;
    for (int i = threadIdx.x; i < N * 32; i += XF_THREADS) {
        const int j = i >> 5, d = (i & 31) * 4;
        const float* row = kv + (size_t)(b * N + j) * 1536 + hd * 128 + d;
        const f4v kk = *reinterpret_cast<const f4v*>(row), vv = *reinterpret_cast<const f4v*>(row + 512);
        ks[j][d] = kk.x; ks[j][d + 1] = kk.y; ks[j][d + 2] = kk.z; ks[j][d + 3] = kk.w;
        *reinterpret_cast<f4v*>(&vs[j][d]) = vv;
    }
    __syncthreads();
    const float scale = 0.08838834764831845f;                     // 128^-1/2
    for (int i = wave; i < N; i += XF_NW) {
        const float* qi = q + (size_t)(b * N + i) * 1536 + hd * 128;
        float s = -INFINITY;
        if (lane < N) {
            float acc = 0.0f;
            for (int d = 0; d < 128; ++d) acc = fmaf(qi[d], ks[lane][d], acc);
            s = acc * scale;
        }
        float mx = s;
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o));
        const float e = (lane < N) ? expf(s - mx) : 0.0f;
        float sum = e;
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) sum += __shfl_xor(sum, o);
        ps[wave][lane] = e / sum;
        __builtin_amdgcn_wave_barrier();
        float o0 = 0.0f, o1 = 0.0f;
        for (int j = 0; j < N; ++j) { const float pj = ps[wave][j]; o0 = fmaf(pj, vs[j][lane], o0); o1 = fmaf(pj, vs[j][lane + 64], o1); }
        unsigned char* orow = att + (size_t)(b * N + i) * 2048 + hd * 512;
        const float ov[2] = {o0, o1};
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const int c = lane + 64 * u;
            const float x = ov[u];
            const _Float16 hi = (fabsf(x) < 6.103515625e-05f) ? (_Float16)0.0f : (_Float16)x;
            _Float16* gp = reinterpret_cast<_Float16*>(orow + (c >> 5) * 128) + (c & 31);
            gp[0] = hi; gp[32] = (_Float16)((x - (float)hi) * 2048.0f);
        }
        __builtin_amdgcn_wave_barrier();
    }
    __syncthreads();                                              // the K / V tiles are free again
}

__global__ __launch_bounds__(XF_THREADS) void transformer_coop_kernel(XfArgs a)
{
    __shared__ __attribute__((aligned(1024))) unsigned char smem[XF_LDS];
    unsigned char* xs = smem;
    float* red = reinterpret_cast<float*>(smem + XF_XS_BYTES);
    const int t = threadIdx.x, lane = t & 63, wave = __builtin_amdgcn_readfirstlane(t >> 6);
    const int blk = blockIdx.x, M = a.M;
    const int rtiles = (M + 31) / 32;
    // column tile of this block per phase (-1: none): 48 / 16 / 64 / 16 tiles of 32 channels, spread over the grid
    const int t_qkv = blk < 48 ? blk : -1;
    const int t_proj = (blk & 3) == 0 ? blk >> 2 : -1;
    const int t_fc1 = blk;
    const int t_fc2 = (blk & 3) == 2 ? blk >> 2 : -1;            // (other blocks than the proj tiles: their weights travel during the fc1 phase)
    unsigned bar = 0;
    if (a.trace && blk == 0 && t == 0) a.trace[0] = wall_clock64();

    h8v Wa[2][4], Wb[8][4];
    if (t_qkv >= 0) xf_load_w<2>(a.L[0].wqkv, t_qkv, 16, wave, lane, Wa);
#pragma unroll 1
    for (int l = 0; l < XF_LAYERS; ++l) {
        const XfLayer& L = a.L[l];
        // ---- LN1 + qkv (no bias)
        if (t_qkv >= 0)
            for (int rt = 0; rt < rtiles; ++rt) {
                xf_ln_tile(a.tok, L.ln1g, L.ln1b, 1e-5f, rt * 32, M, xs, wave, lane);
                __syncthreads();
                xf_gemm_tile<2, XF_EPI_QKV>(Wa, xs, 16, min(32, M - rt * 32), rt * 32, t_qkv, 1536, nullptr, nullptr, a.qkv, red, wave, lane);
            }
        if (t_proj >= 0) xf_load_w<2>(L.wproj, t_proj, 16, wave, lane, Wa);          // travels across the next two barriers
        xf_barrier(a.sync, ++bar, a.trace);
        // ---- attention: (panorama, head) pairs over the blocks
        for (int u = blk; u < a.B * 4; u += XF_GRID) xf_attention(a.qkv, a.att, u >> 2, u & 3, a.N, smem, wave, lane);
        xf_barrier(a.sync, ++bar, a.trace);
        // ---- proj + bias + residual
        if (t_proj >= 0)
            for (int rt = 0; rt < rtiles; ++rt)
                xf_gemm_tile<2, XF_EPI_RES>(Wa, a.att + (size_t)rt * 32 * 2048, 16, min(32, M - rt * 32), rt * 32, t_proj, 512, L.bproj, a.tok, a.tok, red, wave, lane);
        xf_load_w<2>(L.wfc1, t_fc1, 16, wave, lane, Wa);
        xf_barrier(a.sync, ++bar, a.trace);
        // ---- LN2 + fc1 + bias + GELU -> split-half hidden rows
        for (int rt = 0; rt < rtiles; ++rt) {
            xf_ln_tile(a.tok, L.ln2g, L.ln2b, 1e-5f, rt * 32, M, xs, wave, lane);
            __syncthreads();
            xf_gemm_tile<2, XF_EPI_GELU_SH>(Wa, xs, 16, min(32, M - rt * 32), rt * 32, t_fc1, 2048, L.bfc1, nullptr, a.hid, red, wave, lane);
        }
        if (t_fc2 >= 0) xf_load_w<8>(L.wfc2, t_fc2, 64, wave, lane, Wb);
        xf_barrier(a.sync, ++bar, a.trace);
        // ---- fc2 + bias + residual
        if (t_fc2 >= 0)
            for (int rt = 0; rt < rtiles; ++rt)
                xf_gemm_tile<8, XF_EPI_RES>(Wb, a.hid + (size_t)rt * 32 * 8192, 64, min(32, M - rt * 32), rt * 32, t_fc2, 512, L.bfc2, a.tok, a.tok, red, wave, lane);
        if (l + 1 < XF_LAYERS && t_qkv >= 0) xf_load_w<2>(a.L[l + 1].wqkv, t_qkv, 16, wave, lane, Wa);
        xf_barrier(a.sync, ++bar, a.trace);
    }
    // ---- encoder_norm (eps 1e-6), a row per wave: layernorm512_kernel<false>
    for (int row = blk * XF_NW + wave; row < M; row += XF_GRID * XF_NW) {
        const float* p = a.tok + (size_t)row * 512;
        f4v v0 = *reinterpret_cast<const f4v*>(p + lane * 4), v1 = *reinterpret_cast<const f4v*>(p + 256 + lane * 4);
        float s = (v0.x + v0.y) + (v0.z + v0.w) + (v1.x + v1.y) + (v1.z + v1.w);
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
        const float mean = s * (1.0f / 512.0f);
        v0 -= mean; v1 -= mean;
        float q = (v0.x * v0.x + v0.y * v0.y) + (v0.z * v0.z + v0.w * v0.w) + (v1.x * v1.x + v1.y * v1.y) + (v1.z * v1.z + v1.w * v1.w);
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) q += __shfl_xor(q, o);
        const float rstd = 1.0f / sqrtf(q * (1.0f / 512.0f) + 1e-6f);
        const f4v g0 = *reinterpret_cast<const f4v*>(a.encg + lane * 4), g1 = *reinterpret_cast<const f4v*>(a.encg + 256 + lane * 4);
        const f4v b0 = *reinterpret_cast<const f4v*>(a.encb + lane * 4), b1 = *reinterpret_cast<const f4v*>(a.encb + 256 + lane * 4);
        act_store4<false>(a.out, (size_t)row * 512 + lane * 4, v0 * rstd * g0 + b0);
        act_store4<false>(a.out, (size_t)row * 512 + 256 + lane * 4, v1 * rstd * g1 + b1);
    }
    if (a.trace && blk == 0 && t == 0) a.trace[2 * bar + 1] = wall_clock64();
    // ---- the counters are clean for the next launch: the last block to leave (every block has passed the last barrier by then) resets them
    __syncthreads();
    if (t == 0) {
        const unsigned gone = __hip_atomic_fetch_add(a.sync + 1, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (gone == (unsigned)XF_GRID - 1u) {
            __hip_atomic_store(a.sync, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(a.sync + 1, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
}
}  // namespace

OMNI_SH_OVERFLOW_ACCESSOR(omni_sh_overflow_xformer)            // this translation unit's copy of the sticky range flag (act_store4<true>)

extern "C" size_t omni_transformer_scratch_bytes(int M)
{
    return M > 0 ? (size_t)M * (1536 + 512 + 2048) * sizeof(float) : 0;
}

// One launch for Transformer_cascade (see the file header).  tok fp32 [B*N, 512] (token + pos_emb; overwritten), layers: 6 records of DEVICE pointers
// (the array itself in host memory) — LayerNorm weights / biases fp32 [512], proj / fc1 / fc2 biases fp32, the four matrices in the fragment order of
// omni_gemm_rows_pack (qkv = cat(q.weight, kv.weight) [1536, 512], proj [512, 512], fc1 [2048, 512], fc2 [512, 2048]) —, enc_w / enc_b: encoder_norm;
// out fp32 [B*N, 512]; scratch: omni_transformer_scratch_bytes(B*N) bytes; sync: TWO zero-initialised 32-bit counters owned by the caller's execution
// context (one per stream; the kernel leaves them zero).  N <= 64 tokens per panorama.  The bits of omni_gemm_rows_ln_sh_f16x3 / omni_attention_qkv_sh /
// omni_gemm_rows_sh_f16x3 / omni_layernorm512_f32 applied layer by layer (what a lone panorama's forward runs).
extern "C" int omni_transformer_sh_f16x3(float* tok, const omni_xf_layer* layers, const float* enc_w, const float* enc_b, float* out,
                                         void* scratch, size_t scratch_bytes, unsigned* sync, int B, int N, omni_stream_t stream)
{
    if (!tok || !layers || !enc_w || !enc_b || !out || !scratch || !sync) OMNI_FAIL(OMNI_ERR_INVALID, "omni_transformer_sh: null pointer");
    if (B <= 0 || N <= 0 || N > 64) OMNI_FAIL(OMNI_ERR_UNSUPPORTED, "omni_transformer_sh: 1..64 tokens per panorama");
    const long long M = (long long)B * N;
    if (M > 4096) OMNI_FAIL(OMNI_ERR_UNSUPPORTED, "omni_transformer_sh: at most 4096 token rows per launch (larger batches: the per-operator path)");
    if (scratch_bytes < omni_transformer_scratch_bytes((int)M)) OMNI_FAIL(OMNI_ERR_INVALID, "omni_transformer_sh: scratch too small (omni_transformer_scratch_bytes)");
    if (omni_num_cus() < XF_GRID) OMNI_FAIL(OMNI_ERR_UNSUPPORTED, "omni_transformer_sh: the cooperative grid needs 64 compute units");
    XfArgs a;
    static_assert(sizeof(XfLayer) == sizeof(omni_xf_layer), "the C-ABI record is the kernel's");
    for (int l = 0; l < XF_LAYERS; ++l) {
        const omni_xf_layer& s = layers[l];
        if (!s.ln1_w || !s.ln1_b || !s.ln2_w || !s.ln2_b || !s.proj_b || !s.fc1_b || !s.fc2_b || !s.qkv_w16r || !s.proj_w16r || !s.fc1_w16r || !s.fc2_w16r)
            OMNI_FAIL(OMNI_ERR_INVALID, "omni_transformer_sh: null pointer in a layer record");
        memcpy(&a.L[l], &s, sizeof(XfLayer));
    }
#ifdef OMNI_DEBUG_BUILD
    a.trace = omni_debug_trace_buf();
#else
    a.trace = nullptr;
#endif
    a.encg = enc_w; a.encb = enc_b; a.tok = tok; a.out = out; a.sync = sync; a.M = (int)M; a.N = N; a.B = B;
    a.qkv = (float*)scratch;
    a.att = (unsigned char*)scratch + (size_t)M * 1536 * 4;
    a.hid = a.att + (size_t)M * 512 * 4;
    hipLaunchKernelGGL(transformer_coop_kernel, dim3(XF_GRID), dim3(XF_THREADS), 0, (hipStream_t)stream, a);
    OMNI_HIP(hipGetLastError());
    return OMNI_OK;
}

// *timeouts = 1 if a block of the cooperative transformer ever gave up waiting at a device-wide barrier (its grid was not co-resident for ~1 s: results of
// that launch are wrong).  Synchronises: diagnostic only.
extern "C" int omni_transformer_status(int* timeouts, int reset)
{
    if (!timeouts) OMNI_FAIL(OMNI_ERR_INVALID, "omni_transformer_status: null output");
    OMNI_HIP(hipDeviceSynchronize());
    unsigned v = 0;
    OMNI_HIP(hipMemcpyFromSymbol(&v, HIP_SYMBOL(xf_timeout_flag), sizeof(v)));
    if (reset && v) { const unsigned z = 0; OMNI_HIP(hipMemcpyToSymbol(HIP_SYMBOL(xf_timeout_flag), &z, sizeof(z))); }
    *timeouts = v ? 1 : 0;
    return OMNI_OK;
}
