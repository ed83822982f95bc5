// omni_net.hip — the non-GEMM kernels of the network (gfx950), all NHWC fp32 over M = B*N patches.
//
//   stem            conv 7x7 s2 p3 3->64 + folded BN + ReLU          model/spherical_model.py:254 (conv1,bn1,relu)
//   maxpool         3x3 s2 p1                                         :255  F.max_pool3d((3,3,1),(2,2,1))
//   upsample        bilinear, align_corners=False                     :271,279,286,293,300 (q9)
//   add_hw / add_b  broadcast adds (token bias :267-268, point_feat :258)
//   token_pack      `down` output -> [B,N,512] tokens + pos_emb       :263-264, :181
//   layernorm       nn.LayerNorm(512)                                 model/blocks.py:74,81; spherical_model.py:173
//   attention       softmax(q k^T * 128^-1/2) v over the N tokens     model/blocks.py:50-62
//   heads           pred (ReLU) / weight_pred (sigmoid) 3x3 32->1     :223-224,304-307
//   mlp_points      1x1 conv 3->16->64 (+BN+ReLU) of xyz (* depth)    spherical_model_iterative.py:290-305,319,387-393
#include "omni_internal.h"
#include "omni_sh.h"

namespace {
typedef float f4v __attribute__((ext_vector_type(4)));

// ------------------------------------------------------------------ stem
// block: 8x8 output pixels x 64 channels.  Input patch (21x21x3, zero padded) and the whole folded
// filter bank [147][64] live in LDS; thread = (pixel t&63, 16 channels t>>6): the four 16-byte weight
// reads per tap are wave-uniform addresses (LDS broadcast).
template <bool SH>
__global__ __launch_bounds__(256) void stem_kernel(const float* __restrict__ src, const float* __restrict__ wt,
                                                   const float* __restrict__ bias, void* __restrict__ dst,
                                                   int M, int P, int Po)
{
    __shared__ __attribute__((aligned(16))) float wl[147 * 64];
    __shared__ float in[3][21][22];
    const int t = threadIdx.x;
    const int tiles = Po / 8;
    const int m = blockIdx.x / (tiles * tiles), tt = blockIdx.x % (tiles * tiles);
    const int oy0 = (tt / tiles) * 8, ox0 = (tt % tiles) * 8;
    for (int i = t; i < 147 * 64 / 4; i += 256) reinterpret_cast<f4v*>(wl)[i] = reinterpret_cast<const f4v*>(wt)[i];
    const int iy0 = oy0 * 2 - 3, ix0 = ox0 * 2 - 3;
    for (int i = t; i < 3 * 21 * 21; i += 256) {
        const int c = i / 441, r = (i % 441) / 21, q = i % 21;
        const int iy = iy0 + r, ix = ix0 + q;
        float v = 0.0f;
        if ((unsigned)iy < (unsigned)P && (unsigned)ix < (unsigned)P) v = src[((size_t)m * 3 + c) * P * P + (size_t)iy * P + ix];
        in[c][r][q] = v;
    }
    __syncthreads();
    const int p = t & 63, cg = (t >> 6) * 16;
    const int py = p >> 3, px = p & 7;
    float acc[16];
#pragma unroll
    for (int j = 0; j < 16; ++j) acc[j] = bias[cg + j];
    for (int ky = 0; ky < 7; ++ky)
        for (int kx = 0; kx < 7; ++kx)
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                const float x = in[c][py * 2 + ky][px * 2 + kx];
                const float* w = wl + ((ky * 7 + kx) * 3 + c) * 64 + cg;
#pragma unroll
                for (int j4 = 0; j4 < 4; ++j4) {
                    const f4v wv = *reinterpret_cast<const f4v*>(w + 4 * j4);
                    acc[4 * j4 + 0] = fmaf(x, wv.x, acc[4 * j4 + 0]); acc[4 * j4 + 1] = fmaf(x, wv.y, acc[4 * j4 + 1]);
                    acc[4 * j4 + 2] = fmaf(x, wv.z, acc[4 * j4 + 2]); acc[4 * j4 + 3] = fmaf(x, wv.w, acc[4 * j4 + 3]);
                }
            }
    const size_t o = (((size_t)m * Po + oy0 + py) * Po + ox0 + px) * 64 + cg;
#pragma unroll
    for (int j4 = 0; j4 < 4; ++j4) {
        f4v v;
        v.x = fmaxf(acc[4 * j4], 0.f); v.y = fmaxf(acc[4 * j4 + 1], 0.f); v.z = fmaxf(acc[4 * j4 + 2], 0.f); v.w = fmaxf(acc[4 * j4 + 3], 0.f);
        act_store4<SH>(dst, o + 4 * j4, v);
    }
}

// ------------------------------------------------------------------ maxpool 3x3 s2 p1 (NHWC, 4 channels per thread)
template <bool SH>
__global__ __launch_bounds__(256) void maxpool_kernel(const void* __restrict__ src, void* __restrict__ dst,
                                                      int M, int H, int W, int C, int Ho, int Wo)
{
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    const int c4 = C / 4;
    const size_t total = (size_t)M * Ho * Wo * c4;
    if (i >= total) return;
    const int c = (int)(i % c4) * 4;
    size_t r = i / c4;
    const int ox = (int)(r % Wo); r /= Wo;
    const int oy = (int)(r % Ho); const int m = (int)(r / Ho);
    f4v best = (f4v)(-INFINITY);
#pragma unroll
    for (int ky = 0; ky < 3; ++ky)
#pragma unroll
        for (int kx = 0; kx < 3; ++kx) {
            const int iy = oy * 2 - 1 + ky, ix = ox * 2 - 1 + kx;
            if ((unsigned)iy < (unsigned)H && (unsigned)ix < (unsigned)W) {
                const f4v v = act_load4<SH>(src, (((size_t)m * H + iy) * W + ix) * C + c);
                best.x = fmaxf(best.x, v.x); best.y = fmaxf(best.y, v.y); best.z = fmaxf(best.z, v.z); best.w = fmaxf(best.w, v.w);
            }
        }
    act_store4<SH>(dst, (((size_t)m * Ho + oy) * Wo + ox) * C + c, best);
}

// ------------------------------------------------------------------ bilinear upsample, align_corners=False (ATen upsample_bilinear2d)
template <bool SH>
__global__ __launch_bounds__(256) void upsample_kernel(const void* __restrict__ src, void* __restrict__ dst,
                                                       int M, int H, int W, int C, int Ho, int Wo, float sy, float sx)
{
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    const int c4 = C / 4;
    const size_t total = (size_t)M * Ho * Wo * c4;
    if (i >= total) return;
    const int c = (int)(i % c4) * 4;
    size_t r = i / c4;
    const int ox = (int)(r % Wo); r /= Wo;
    const int oy = (int)(r % Ho); const int m = (int)(r / Ho);
    // area_pixel_compute_source_index: scale*(dst+0.5)-0.5, clamped below at 0
    const float fy = fmaxf(sy * ((float)oy + 0.5f) - 0.5f, 0.0f), fx = fmaxf(sx * ((float)ox + 0.5f) - 0.5f, 0.0f);
    const int y0 = (int)fy, x0 = (int)fx;
    const int y1 = y0 + (y0 < H - 1 ? 1 : 0), x1 = x0 + (x0 < W - 1 ? 1 : 0);
    const float ly = fy - (float)y0, lx = fx - (float)x0, hy = 1.0f - ly, hx = 1.0f - lx;
    const size_t b = (size_t)m * H * W * C + c;
    const f4v v00 = act_load4<SH>(src, b + ((size_t)y0 * W + x0) * C);
    const f4v v01 = act_load4<SH>(src, b + ((size_t)y0 * W + x1) * C);
    const f4v v10 = act_load4<SH>(src, b + ((size_t)y1 * W + x0) * C);
    const f4v v11 = act_load4<SH>(src, b + ((size_t)y1 * W + x1) * C);
    f4v o;
    o.x = hy * (hx * v00.x + lx * v01.x) + ly * (hx * v10.x + lx * v11.x);
    o.y = hy * (hx * v00.y + lx * v01.y) + ly * (hx * v10.y + lx * v11.y);
    o.z = hy * (hx * v00.z + lx * v01.z) + ly * (hx * v10.z + lx * v11.z);
    o.w = hy * (hx * v00.w + lx * v01.w) + ly * (hx * v10.w + lx * v11.w);
    act_store4<SH>(dst, (((size_t)m * Ho + oy) * Wo + ox) * C + c, o);
}

// SH -> SH, eight channels per thread (16-byte hi and lo pieces): half the instructions per byte of the generic form
typedef _Float16 omni_h8v __attribute__((ext_vector_type(8)));
__global__ __launch_bounds__(256) void upsample_sh8_kernel(const unsigned char* __restrict__ src, unsigned char* __restrict__ dst,
                                                           int M, int H, int W, int C, int Ho, int Wo, float sy, float sx)
{
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    const int c8 = C / 8;
    const size_t total = (size_t)M * Ho * Wo * c8;
    if (i >= total) return;
    const int c = (int)(i % c8) * 8;
    size_t r = i / c8;
    const int ox = (int)(r % Wo); r /= Wo;
    const int oy = (int)(r % Ho); const int m = (int)(r / Ho);
    const float fy = fmaxf(sy * ((float)oy + 0.5f) - 0.5f, 0.0f), fx = fmaxf(sx * ((float)ox + 0.5f) - 0.5f, 0.0f);
    const int y0 = (int)fy, x0 = (int)fx;
    const int y1 = y0 + (y0 < H - 1 ? 1 : 0), x1 = x0 + (x0 < W - 1 ? 1 : 0);
    const float ly = fy - (float)y0, lx = fx - (float)x0, hy = 1.0f - ly, hx = 1.0f - lx;
    const size_t coff = (size_t)(c >> 5) * 128 + (c & 31) * 2;               // byte offset of the hi halfs inside a pixel
    const size_t pp = (size_t)C * 4;                                         // pixel pitch in bytes
    const unsigned char* b = src + (size_t)m * H * W * pp + coff;
    const unsigned char* p00 = b + ((size_t)y0 * W + x0) * pp; const unsigned char* p01 = b + ((size_t)y0 * W + x1) * pp;
    const unsigned char* p10 = b + ((size_t)y1 * W + x0) * pp; const unsigned char* p11 = b + ((size_t)y1 * W + x1) * pp;
    const omni_h8v h00 = *reinterpret_cast<const omni_h8v*>(p00), l00 = *reinterpret_cast<const omni_h8v*>(p00 + 64);
    const omni_h8v h01 = *reinterpret_cast<const omni_h8v*>(p01), l01 = *reinterpret_cast<const omni_h8v*>(p01 + 64);
    const omni_h8v h10 = *reinterpret_cast<const omni_h8v*>(p10), l10 = *reinterpret_cast<const omni_h8v*>(p10 + 64);
    const omni_h8v h11 = *reinterpret_cast<const omni_h8v*>(p11), l11 = *reinterpret_cast<const omni_h8v*>(p11 + 64);
    omni_h8v oh, ol;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        const float v00 = fmaf((float)l00[e], 4.8828125e-4f, (float)h00[e]), v01 = fmaf((float)l01[e], 4.8828125e-4f, (float)h01[e]);
        const float v10 = fmaf((float)l10[e], 4.8828125e-4f, (float)h10[e]), v11 = fmaf((float)l11[e], 4.8828125e-4f, (float)h11[e]);
        const float o = hy * (hx * v00 + lx * v01) + ly * (hx * v10 + lx * v11);
        const _Float16 h = (fabsf(o) < 6.103515625e-05f) ? (_Float16)0.0f : (_Float16)o;
        oh[e] = h; ol[e] = (_Float16)((o - (float)h) * 2048.0f);
    }
    unsigned char* d = dst + (((size_t)m * Ho + oy) * Wo + ox) * pp + coff;
    *reinterpret_cast<omni_h8v*>(d) = oh; *reinterpret_cast<omni_h8v*>(d + 64) = ol;
}

// x[m][hw][c] += y[m][c]            (token bias on layer4)
__global__ __launch_bounds__(256) void add_hw_kernel(float* __restrict__ x, const float* __restrict__ y, size_t total, int HW, int C)
{
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= total) return;
    const int c = (int)(i % C);
    const size_t m = i / ((size_t)HW * C);
    x[i] += y[m * C + c];
}
// x[i] += y[i % period]            (point_feat [N,h,w,64] broadcast over the batch, or a full-size add)
__global__ __launch_bounds__(256) void add_period_kernel(float* __restrict__ x, const float* __restrict__ y, size_t total, size_t period)
{
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= total) return;
    x[i] += y[i % period];
}

// d[m][hw][c32] -> tok[m][c*HW + hw] + pos[(m % N)][.]      (reshape(bs,-1,N).transpose(1,2) of :264, + pos_emb :181)
// the same two adds on an SH tensor x (y stays fp32), four channels per thread
__global__ __launch_bounds__(256) void add_hw_sh_kernel(void* __restrict__ x, const float* __restrict__ y, size_t total4, int HW, int C)
{
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= total4) return;
    const size_t e = i * 4;
    const int c = (int)(e % C);
    const size_t m = e / ((size_t)HW * C);
    act_store4<true>(x, e, act_load4<true>(x, e) + *reinterpret_cast<const f4v*>(y + m * C + c));
}
__global__ __launch_bounds__(256) void add_period_sh_kernel(void* __restrict__ x, const float* __restrict__ y, size_t total4, size_t period)
{
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= total4) return;
    const size_t e = i * 4;
    act_store4<true>(x, e, act_load4<true>(x, e) + *reinterpret_cast<const f4v*>(y + e % period));
}
__global__ __launch_bounds__(256) void token_pack_kernel(const float* __restrict__ d, const float* __restrict__ pos,
                                                         float* __restrict__ tok, int M, int N, int HW, int C)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    const int D = HW * C;
    if (i >= M * D) return;
    const int m = i / D, j = i % D, c = j / HW, hw = j % HW;
    tok[i] = d[((size_t)m * HW + hw) * C + c] + pos[(m % N) * D + j];
}

// one wave per row of 512
template <bool SH>
__global__ __launch_bounds__(256) void layernorm512_kernel(const float* __restrict__ x, const float* __restrict__ g,
                                                           const float* __restrict__ b, void* __restrict__ y, int rows, float eps)
{
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (row >= rows) return;
    const float* p = x + (size_t)row * 512;
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
    const f4v g0 = *reinterpret_cast<const f4v*>(g + lane * 4), g1 = *reinterpret_cast<const f4v*>(g + 256 + lane * 4);
    const f4v b0 = *reinterpret_cast<const f4v*>(b + lane * 4), b1 = *reinterpret_cast<const f4v*>(b + 256 + lane * 4);
    const size_t o = (size_t)row * 512;
    act_store4<SH>(y, o + lane * 4, v0 * rstd * g0 + b0);
    act_store4<SH>(y, o + 256 + lane * 4, v1 * rstd * g1 + b1);
}

// attention core: block = (batch item, head); N <= 64 tokens, head dim 128.
// q rows have pitch qs floats, k|v rows pitch kvs (k at column 0, v at column vo); SH: the result is a split-half tensor
template <bool SH>
__global__ __launch_bounds__(256) void attention_kernel(const float* __restrict__ q, const float* __restrict__ kv,
                                                        void* __restrict__ out, int N, float scale, int qs, int kvs, int vo)
{
    __shared__ float ks[64][129];
    __shared__ __attribute__((aligned(16))) float vs[64][128];
    __shared__ float ps[4][64];
    const int b = blockIdx.x >> 2, h = blockIdx.x & 3;
    const int t = threadIdx.x, wave = t >> 6, lane = t & 63;
    for (int i = t; i < N * 32; i += 256) {                       // 16-byte loads of K and V rows
        const int j = i >> 5, d = (i & 31) * 4;
        const float* row = kv + (size_t)(b * N + j) * kvs + h * 128 + d;
        const f4v kk = *reinterpret_cast<const f4v*>(row), vv = *reinterpret_cast<const f4v*>(row + vo);
        ks[j][d] = kk.x; ks[j][d + 1] = kk.y; ks[j][d + 2] = kk.z; ks[j][d + 3] = kk.w;
        *reinterpret_cast<f4v*>(&vs[j][d]) = vv;
    }
    __syncthreads();
    // blockIdx.y = group of TPB query tokens (one per wave and trip): the few (panorama, head) pairs alone would leave
    // most CUs idle and every block walking all N tokens serially
    constexpr int TPB = 4;
    for (int i = blockIdx.y * TPB + wave; i < min(N, (int)(blockIdx.y + 1) * TPB); i += 4) {
        const float* qi = q + (size_t)(b * N + i) * qs + h * 128;
        float s = -INFINITY;
        if (lane < N) {
            float acc = 0.0f;
            for (int d = 0; d < 128; ++d) acc = fmaf(qi[d], ks[lane][d], acc);
            s = acc * scale;
        }
        float mx = s;
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o));
        float e = (lane < N) ? expf(s - mx) : 0.0f;
        float sum = e;
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) sum += __shfl_xor(sum, o);
        ps[wave][lane] = e / sum;
        __builtin_amdgcn_wave_barrier();
        float o0 = 0.0f, o1 = 0.0f;
        for (int j = 0; j < N; ++j) { const float pj = ps[wave][j]; o0 = fmaf(pj, vs[j][lane], o0); o1 = fmaf(pj, vs[j][lane + 64], o1); }
        if (SH) {                                                 // channel c of a row: group c>>5, hi half at 2*(c&31), lo 64 bytes on
            unsigned char* orow = (unsigned char*)out + (size_t)(b * N + i) * 2048 + h * 512;
            const float ov[2] = {o0, o1};
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                const int c = lane + 64 * u;
                const float x = ov[u];
                const _Float16 hi = (fabsf(x) < 6.103515625e-05f) ? (_Float16)0.0f : (_Float16)x;
                _Float16* gp = reinterpret_cast<_Float16*>(orow + (c >> 5) * 128) + (c & 31);
                gp[0] = hi; gp[32] = (_Float16)((x - (float)hi) * 2048.0f);
            }
        } else {
            float* oi = (float*)out + (size_t)(b * N + i) * 512 + h * 128;
            oi[lane] = o0; oi[lane + 64] = o1;
        }
        __builtin_amdgcn_wave_barrier();
    }
}

// heads: x [M,P,P,32] -> a = relu(conv_pred(x)) * (conf ? sigmoid(conv_w(x)) : 1), c = sigmoid(conv_w(x)); planar [M,P,P].
// Block = one 16x16 pixel tile of one patch: the 18x18x32 halo tile is staged in LDS once (pixel pitch 36 floats:
// conflict-free ds_read_b128 across pixels) instead of being re-fetched by all nine taps; the two filters are read
// through the scalar cache (wave-uniform indices).
constexpr int HT = 16, HPITCH = 36;
__global__ __launch_bounds__(256) void heads_kernel(const float* __restrict__ x, const float* __restrict__ w /*[2][9][32]*/,
                                                    float bp, float bw, float* __restrict__ outa, float* __restrict__ outc,
                                                    int M, int P, int conf, int ntiles)
{
    // persistent blocks: the next tile's 18x18x32 halo is in flight in registers (11 x 16 bytes per thread) while the
    // current tile's 2 x 288 multiply-adds per pixel run out of LDS
    __shared__ __attribute__((aligned(16))) float tile[(HT + 2) * (HT + 2) * HPITCH];
    constexpr int NLD = ((HT + 2) * (HT + 2) * 8 + 255) / 256;
    const int tiles = (P + HT - 1) / HT;
    f4v pre[NLD];
    auto fetch = [&](int tid) {
        const int m = tid / (tiles * tiles), tt = tid % (tiles * tiles);
        const int y0 = (tt / tiles) * HT - 1, x0 = (tt % tiles) * HT - 1;
#pragma unroll
        for (int k = 0; k < NLD; ++k) {
            const int i = threadIdx.x + 256 * k;
            const int px = i >> 3, q = i & 7;
            const int iy = y0 + px / (HT + 2), ix = x0 + px % (HT + 2);
            pre[k] = (f4v)(0.0f);
            if (i < (HT + 2) * (HT + 2) * 8 && (unsigned)iy < (unsigned)P && (unsigned)ix < (unsigned)P)
                pre[k] = *reinterpret_cast<const f4v*>(x + (((size_t)m * P + iy) * P + ix) * 32 + q * 4);
        }
    };
    const int ly = threadIdx.x >> 4, lx = threadIdx.x & 15;
    int tid = blockIdx.x;
    if (tid < ntiles) fetch(tid);
    for (; tid < ntiles; tid += gridDim.x) {
        __syncthreads();                                          // the previous tile's taps are done
#pragma unroll
        for (int k = 0; k < NLD; ++k) {
            const int i = threadIdx.x + 256 * k;
            if (i < (HT + 2) * (HT + 2) * 8) *reinterpret_cast<f4v*>(tile + (i >> 3) * HPITCH + (i & 7) * 4) = pre[k];
        }
        __syncthreads();
        if (tid + (int)gridDim.x < ntiles) fetch(tid + gridDim.x);
        float ap = bp, aw = bw;
#pragma unroll
        for (int ky = 0; ky < 3; ++ky)
#pragma unroll
            for (int kx = 0; kx < 3; ++kx) {
                const float* px = tile + ((ly + ky) * (HT + 2) + lx + kx) * HPITCH;
                const float* w0 = w + (ky * 3 + kx) * 32; const float* w1 = w0 + 288;
#pragma unroll
                for (int c = 0; c < 32; c += 4) {
                    const f4v v = *reinterpret_cast<const f4v*>(px + c);
                    ap = fmaf(v.x, w0[c], ap); ap = fmaf(v.y, w0[c + 1], ap); ap = fmaf(v.z, w0[c + 2], ap); ap = fmaf(v.w, w0[c + 3], ap);
                    aw = fmaf(v.x, w1[c], aw); aw = fmaf(v.y, w1[c + 1], aw); aw = fmaf(v.z, w1[c + 2], aw); aw = fmaf(v.w, w1[c + 3], aw);
                }
            }
        const int m = tid / (tiles * tiles), tt = tid % (tiles * tiles);
        const int oy = (tt / tiles) * HT + ly, ox = (tt % tiles) * HT + lx;
        if (oy < P && ox < P) {
            const size_t i = ((size_t)m * P + oy) * P + ox;
            const float pr = fmaxf(ap, 0.0f), cf = 1.0f / (1.0f + expf(-aw));
            outa[i] = conf ? pr * cf : pr;
            if (outc) outc[i] = cf;
        }
    }
}

// mlp_points: rows = Mo*HW; in = xyz[(m % N)][c][hw] * (depth ? depth[m][hw] : 1)
__global__ __launch_bounds__(256) void mlp_points_kernel(const float* __restrict__ xyz, const float* __restrict__ depth,
                                                         const float* __restrict__ w1 /*[16][3]*/, const float* __restrict__ b1,
                                                         const float* __restrict__ w2 /*[64][16]*/, const float* __restrict__ b2,
                                                         float* __restrict__ out, int Mo, int N, int HW)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= Mo * HW) return;
    const int m = i / HW, hw = i % HW, n = m % N;
    const float d = depth ? depth[i] : 1.0f;
    const float x0 = xyz[((size_t)n * 3 + 0) * HW + hw] * d, x1 = xyz[((size_t)n * 3 + 1) * HW + hw] * d, x2 = xyz[((size_t)n * 3 + 2) * HW + hw] * d;
    float hid[16];
#pragma unroll
    for (int j = 0; j < 16; ++j) hid[j] = fmaxf(fmaf(x2, w1[j * 3 + 2], fmaf(x1, w1[j * 3 + 1], fmaf(x0, w1[j * 3], b1[j]))), 0.0f);
    float* o = out + (size_t)i * 64;
    for (int j = 0; j < 64; j += 4) {
        float r[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            float acc = b2[j + e];
#pragma unroll
            for (int k = 0; k < 16; ++k) acc = fmaf(hid[k], w2[(j + e) * 16 + k], acc);
            r[e] = fmaxf(acc, 0.0f);
        }
        f4v v; v.x = r[0]; v.y = r[1]; v.z = r[2]; v.w = r[3];
        *reinterpret_cast<f4v*>(o + j) = v;
    }
}

inline int nblk(size_t n) { return (int)((n + 255) / 256); }
}  // namespace

#define S_ (hipStream_t)stream
OMNI_SH_OVERFLOW_ACCESSOR(omni_sh_overflow_net)

extern "C" {

int omni_stem_f32(const float* src, const float* wt, const float* bias, float* dst, int M, int P, omni_stream_t stream)
{
    if (P % 16) OMNI_FAIL(OMNI_ERR_INVALID, "omni_stem: patch size must be a multiple of 16");
    const int Po = P / 2;
    hipLaunchKernelGGL(stem_kernel<false>, dim3(M * (Po / 8) * (Po / 8)), dim3(256), 0, S_, src, wt, bias, (void*)dst, M, P, Po);
    OMNI_HIP(hipGetLastError()); return OMNI_OK;
}
int omni_maxpool3x3s2_f32(const float* src, float* dst, int M, int H, int W, int C, omni_stream_t stream)
{
    const int Ho = (H + 2 - 3) / 2 + 1, Wo = (W + 2 - 3) / 2 + 1;
    hipLaunchKernelGGL(maxpool_kernel<false>, dim3(nblk((size_t)M * Ho * Wo * C / 4)), dim3(256), 0, S_, (const void*)src, (void*)dst, M, H, W, C, Ho, Wo);
    OMNI_HIP(hipGetLastError()); return OMNI_OK;
}
int omni_upsample_bilinear_f32(const float* src, float* dst, int M, int H, int W, int C, int Ho, int Wo, omni_stream_t stream)
{
    hipLaunchKernelGGL(upsample_kernel<false>, dim3(nblk((size_t)M * Ho * Wo * C / 4)), dim3(256), 0, S_, (const void*)src, (void*)dst, M, H, W, C, Ho, Wo,
                       (float)H / (float)Ho, (float)W / (float)Wo);
    OMNI_HIP(hipGetLastError()); return OMNI_OK;
}
int omni_add_hw_f32(float* x, const float* y, int M, int HW, int C, omni_stream_t stream)
{
    const size_t total = (size_t)M * HW * C;
    hipLaunchKernelGGL(add_hw_kernel, dim3(nblk(total)), dim3(256), 0, S_, x, y, total, HW, C);
    OMNI_HIP(hipGetLastError()); return OMNI_OK;
}
int omni_add_period_f32(float* x, const float* y, size_t total, size_t period, omni_stream_t stream)
{
    hipLaunchKernelGGL(add_period_kernel, dim3(nblk(total)), dim3(256), 0, S_, x, y, total, period);
    OMNI_HIP(hipGetLastError()); return OMNI_OK;
}
// ---- the same operators on split-half (SH) activations (omni_sh.h); C % 32 == 0
int omni_stem_sh(const float* src, const float* wt, const float* bias, void* dst, int M, int P, omni_stream_t stream)
{
    if (P % 16) OMNI_FAIL(OMNI_ERR_INVALID, "omni_stem: patch size must be a multiple of 16");
    const int Po = P / 2;
    hipLaunchKernelGGL(stem_kernel<true>, dim3(M * (Po / 8) * (Po / 8)), dim3(256), 0, S_, src, wt, bias, dst, M, P, Po);
    OMNI_HIP(hipGetLastError()); return OMNI_OK;
}
int omni_maxpool3x3s2_sh(const void* src, void* dst, int M, int H, int W, int C, omni_stream_t stream)
{
    if (C % 32) OMNI_FAIL(OMNI_ERR_INVALID, "SH tensors need C % 32 == 0");
    const int Ho = (H + 2 - 3) / 2 + 1, Wo = (W + 2 - 3) / 2 + 1;
    hipLaunchKernelGGL(maxpool_kernel<true>, dim3(nblk((size_t)M * Ho * Wo * C / 4)), dim3(256), 0, S_, src, dst, M, H, W, C, Ho, Wo);
    OMNI_HIP(hipGetLastError()); return OMNI_OK;
}
int omni_upsample_bilinear_sh(const void* src, void* dst, int M, int H, int W, int C, int Ho, int Wo, omni_stream_t stream)
{
    if (C % 32) OMNI_FAIL(OMNI_ERR_INVALID, "SH tensors need C % 32 == 0");
    hipLaunchKernelGGL(upsample_sh8_kernel, dim3(nblk((size_t)M * Ho * Wo * C / 8)), dim3(256), 0, S_, (const unsigned char*)src,
                       (unsigned char*)dst, M, H, W, C, Ho, Wo, (float)H / (float)Ho, (float)W / (float)Wo);
    OMNI_HIP(hipGetLastError()); return OMNI_OK;
}
int omni_add_hw_sh(void* x, const float* y, int M, int HW, int C, omni_stream_t stream)
{
    if (C % 32) OMNI_FAIL(OMNI_ERR_INVALID, "SH tensors need C % 32 == 0");
    const size_t total4 = (size_t)M * HW * C / 4;
    hipLaunchKernelGGL(add_hw_sh_kernel, dim3(nblk(total4)), dim3(256), 0, S_, x, y, total4, HW, C);
    OMNI_HIP(hipGetLastError()); return OMNI_OK;
}
int omni_add_period_sh(void* x, const float* y, size_t total, size_t period, omni_stream_t stream)
{
    if (total % 32 || period % 32) OMNI_FAIL(OMNI_ERR_INVALID, "SH tensors need C % 32 == 0");
    hipLaunchKernelGGL(add_period_sh_kernel, dim3(nblk(total / 4)), dim3(256), 0, S_, x, y, total / 4, period);
    OMNI_HIP(hipGetLastError()); return OMNI_OK;
}
int omni_layernorm512_sh(const float* x, const float* g, const float* b, void* y, int rows, float eps, omni_stream_t stream)
{
    hipLaunchKernelGGL(layernorm512_kernel<true>, dim3((rows + 3) / 4), dim3(256), 0, S_, x, g, b, y, rows, eps);
    OMNI_HIP(hipGetLastError()); return OMNI_OK;
}
int omni_attention_qkv_sh(const float* qkv, void* out, int B, int N, omni_stream_t stream)
{
    if (N > 64) OMNI_FAIL(OMNI_ERR_UNSUPPORTED, "omni_attention: at most 64 tokens");
    hipLaunchKernelGGL(attention_kernel<true>, dim3(B * 4, (N + 3) / 4), dim3(256), 0, S_, qkv, qkv + 512, out, N, 0.08838834764831845f, 1536, 1536, 512);
    OMNI_HIP(hipGetLastError()); return OMNI_OK;
}
int omni_token_pack_f32(const float* d, const float* pos, float* tok, int M, int N, int HW, int C, omni_stream_t stream)
{
    hipLaunchKernelGGL(token_pack_kernel, dim3(nblk((size_t)M * HW * C)), dim3(256), 0, S_, d, pos, tok, M, N, HW, C);
    OMNI_HIP(hipGetLastError()); return OMNI_OK;
}
int omni_layernorm512_f32(const float* x, const float* g, const float* b, float* y, int rows, float eps, omni_stream_t stream)
{
    hipLaunchKernelGGL(layernorm512_kernel<false>, dim3((rows + 3) / 4), dim3(256), 0, S_, x, g, b, (void*)y, rows, eps);
    OMNI_HIP(hipGetLastError()); return OMNI_OK;
}
int omni_attention_f32(const float* q, const float* kv, float* out, int B, int N, omni_stream_t stream)
{
    if (N > 64) OMNI_FAIL(OMNI_ERR_UNSUPPORTED, "omni_attention: at most 64 tokens");
    hipLaunchKernelGGL(attention_kernel<false>, dim3(B * 4, (N + 3) / 4), dim3(256), 0, S_, q, kv, (void*)out, N, 0.08838834764831845f /* 128^-1/2 */, 512, 1024, 512);
    OMNI_HIP(hipGetLastError()); return OMNI_OK;
}
int omni_heads_f32(const float* x, const float* w, float bias_pred, float bias_weight, float* out_a, float* out_c,
                   int M, int P, int confidence, omni_stream_t stream)
{
    const int ht = (P + HT - 1) / HT;
    const int ntiles = M * ht * ht;
    hipLaunchKernelGGL(heads_kernel, dim3(ntiles < 768 ? ntiles : 768), dim3(256), 0, S_, x, w, bias_pred, bias_weight, out_a, out_c, M, P, confidence, ntiles);
    OMNI_HIP(hipGetLastError()); return OMNI_OK;
}
int omni_mlp_points_f32(const float* xyz, const float* depth, const float* w1, const float* b1, const float* w2,
                        const float* b2, float* out, int Mo, int N, int HW, omni_stream_t stream)
{
    hipLaunchKernelGGL(mlp_points_kernel, dim3(nblk((size_t)Mo * HW)), dim3(256), 0, S_, xyz, depth, w1, b1, w2, b2, out, Mo, N, HW);
    OMNI_HIP(hipGetLastError()); return OMNI_OK;
}
}
