// omni_conv.hip — implicit-GEMM convolution / GEMM on the fp32 matrix cores (gfx950).
//
// Replaces the reference's Conv3d(k,k,1)+BatchNorm3d(+ReLU)(+residual) stacks
// (model/spherical_model.py:122-167 encoder, :29-37,214-222 decoder, :211 `down`) and every
// nn.Linear of the transformer (model/blocks.py:19-21,43-46).  Mathematically the reference's
// Conv3d over [B,C,P,P,N] is a 2-D convolution over the B*N patches; here activations are NHWC
// fp32 [M = B*N, H, W, C] and eval-mode BatchNorm is folded into the weights at load time.
//
// Precision: the 1e-3 abs gate on depth (values up to ~8 m through ~50 layers) needs more than
// 11 mantissa bits per stored activation: an fp16-storage emulation of the whole network is off by
// 1.8e-2 max (oracle/model_ref.py, fp16=True), so this path uses the EXACT fp32 MFMA
// v_mfma_f32_32x32x2_f32 (157 TFLOP/s dense peak, bit-for-bit a k-ordered fmaf chain).
//
// F16X3 = true selects the fp16 matrix cores at fp32-class accuracy instead: every operand x is split into a pair of
// halfs x = hi + lo * 2^-11 (hi = fp16(x), lo = fp16((x - hi) * 2^11): 22-23 significant bits, the 2^11 scale keeps `lo`
// out of the fp16 subnormal range) and a product block is THREE v_mfma_f32_32x32x16_f16 with fp32 accumulation,
//        A.B = A_hi.B_hi + 2^-11 (A_hi.B_lo + A_lo.B_hi)          [A_lo.B_lo * 2^-22 dropped]
// i.e. 3/16 of the fp32-MFMA time.  Activations stay fp32 NHWC in HBM (so every other kernel is untouched): the A tile
// is split while it is parked in LDS (2 cvt_pk + 4 sub/mul per 4 channels); the weights are split once at load time
// ([Cout][K/32][hi32|lo32] halfs).  Max error 1.6e-6..3.7e-6 against float64 on every conv shape of the network.
//
// GEMM view: D[r][j] = sum_k A[r][k] * Wt[j][k],  r = output pixel (m, oy, ox), j = output channel,
// k = (tap, input channel).  A is gathered on the fly (zero outside the image), channels of a tap
// are contiguous in NHWC so every A fetch is a 16-byte load; an optional second source supplies
// channels [C1, C1+C2) (the decoder's torch.cat skip connections, :275,282,289,296, without the copy).
// Block = 256 threads = 4 waves; LDS tiles A[BM][32], B[BN][32] (row pitch 36 floats: conflict-free
// ds_read_b128); the next K-step's tiles are in flight in registers while the current one is on the
// matrix cores.  Lane l feeds MFMA with A[row l&31][k = 4*(l>>5) .. +3] from ONE ds_read_b128: the
// four k-pairs {0,4},{1,5},{2,6},{3,7} of an 8-wide k group go to four back-to-back MFMAs.
// Epilogue (fused): + bias[j], + residual[r][j], ReLU / GELU, row-major NHWC store (a lane group
// writes 32 consecutive channels = 128 B).
#include <stdlib.h>
#include "omni_internal.h"

namespace {

typedef float f16v __attribute__((ext_vector_type(16)));
typedef float f4v __attribute__((ext_vector_type(4)));
typedef _Float16 h8v __attribute__((ext_vector_type(8)));
typedef _Float16 h4v __attribute__((ext_vector_type(4)));

// raw buffer descriptor (V#) over [p, p+bytes): stride 0, num_records = bytes; gfx9-family flags dword
typedef __amdgpu_buffer_rsrc_t rsrc_t;
__device__ __forceinline__ rsrc_t make_rsrc(const void* p, size_t bytes)
{
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p), (short)0,
                                             (int)(unsigned)(bytes > 0xffffffffull ? 0xffffffffull : bytes), 0x00020000);
}

struct ConvArgs {
    const float* src1; const float* src2;   // NHWC inputs (src2 may be null)
    const float* wt;                         // [Cout][KH*KW*(C1+C2)], BN folded
    const float* bias;                       // [Cout] or null
    const float* res;                        // residual, same shape as dst, or null
    float* dst;                              // [M, Ho, Wo, Cout]
    int M, H, W, C1, C2, Ho, Wo, Cout;
    int KH, KW, stride, pad, act;
    int rows;                                // M*Ho*Wo
    int f16x3;                               // 1: wt is the pre-split half format, products on the fp16 matrix cores
    int splitk;                              // >1: blockIdx.y owns a K range and writes raw partial sums to ws[y]
    float* ws;                               // [splitk][rows][Cout]
    int dbg;                                 // debug build only (OMNI_CONV_DBG ablation bits): 1 no MFMA, 2 no refetch, 4 no stash
};

constexpr int BK = 32;
constexpr int LDP = BK + 4;                  // LDS row pitch in floats
constexpr int HP = BK + 8;                   // LDS row pitch in halfs (F16X3 images): 80 B, conflict-free ds_read_b128

// __launch_bounds__(256, 5): at most 96 registers (VGPR+AGPR) so that five blocks share a CU (5 x 27.6 KB LDS fits):
// the 1152-block layers (layer1, de_conv2_x at B=8) then run in ONE round of 4.5 waves per SIMD instead of 4 + a tail.
template <int BM, int BN, int WM, int WN, bool F16X3>
__global__ __launch_bounds__(256, (F16X3 ? 3 : 5)) void conv_igemm_f32_kernel(ConvArgs a)
{
    constexpr int TM = BM / WM / 32, TN = BN / WN / 32;        // 32x32 MFMA tiles per wave
    constexpr int APASS = BM / 32, BPASS = BN / 32;            // float4 loads per thread per K-step
    constexpr int LDS_BYTES = F16X3 ? 2 * (BM + BN) * HP * 2 : (BM + BN) * LDP * 4;
    __shared__ __attribute__((aligned(16))) unsigned char lds_raw[LDS_BYTES];
    float* As = reinterpret_cast<float*>(lds_raw);
    float* Bs = As + BM * LDP;
    _Float16* Ah = reinterpret_cast<_Float16*>(lds_raw);         // F16X3 images: A_hi, A_lo, B_hi, B_lo
    _Float16* Al = Ah + BM * HP;
    _Float16* Bh = Al + BM * HP;
    _Float16* Bl = Bh + BN * HP;

    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const int wm = wave / WN, wn = wave % WN;
    const int ntn = a.Cout / BN;
    const int tile_m = blockIdx.x / ntn, tile_n = blockIdx.x % ntn;
    const int row0 = tile_m * BM, col0 = tile_n * BN;
    const int Cin = a.C1 + a.C2;
    const int cchunks = Cin / BK;
    const int ksteps = a.KH * a.KW * cchunks;
    const int Kfull = a.KH * a.KW * Cin;

    // ---- loader geometry: thread -> (row lr + 32*i, k quad kq).  All addressing is 32-bit: raw buffer loads (SGPR
    // descriptor + VGPR byte offset); a tap outside the image gets an out-of-range offset, for which the buffer unit
    // returns zeros (the conv's zero padding) without a branch; per load the VALU cost is a bit test, an add, a select.
    const int lr = t >> 3, kq = (t & 7) * 4;
    int pix[APASS];                                              // pixel index of the (virtual, possibly padded) window origin
    unsigned vmask[APASS];                                       // bit (ky*KW+kx): tap inside the image
#pragma unroll
    for (int i = 0; i < APASS; ++i) {
        const int r = row0 + lr + 32 * i;
        vmask[i] = 0; pix[i] = 0;
        if (r < a.rows) {
            const int hw = a.Ho * a.Wo;
            const int m = r / hw, rem = r - m * hw;
            const int oy = (rem / a.Wo) * a.stride - a.pad, ox = (rem % a.Wo) * a.stride - a.pad;
            pix[i] = (m * a.H + oy) * a.W + ox;
            unsigned vm = 0;
#pragma unroll
            for (int ky = 0; ky < 3; ++ky)
#pragma unroll
                for (int kx = 0; kx < 3; ++kx)
                    if (ky < a.KH && kx < a.KW && (unsigned)(oy + ky) < (unsigned)a.H && (unsigned)(ox + kx) < (unsigned)a.W)
                        vm |= 1u << (ky * a.KW + kx);
            vmask[i] = vm;
        }
    }
    const rsrc_t rs1 = make_rsrc(a.src1, (size_t)a.M * a.H * a.W * a.C1 * 4);
    const rsrc_t rs2 = make_rsrc(a.src2 ? a.src2 : a.src1, a.src2 ? (size_t)a.M * a.H * a.W * a.C2 * 4 : 0);
    // weights: fp32 [Cout][K] (16-byte piece = 4 k)  |  F16X3: halfs [Cout][K/32][hi32|lo32] (piece t&7: 0-3 hi, 4-7 lo)
    const rsrc_t rsw = make_rsrc(a.wt, (size_t)a.Cout * Kfull * 4);
    int wbase[BPASS];
#pragma unroll
    for (int i = 0; i < BPASS; ++i) wbase[i] = F16X3 ? ((col0 + lr + 32 * i) * ksteps * 64 + (t & 7) * 8) * 2
                                                      : ((col0 + lr + 32 * i) * Kfull + kq) * 4;

    f4v ra[APASS], rb[BPASS];
    int f_tap = 0, f_c = 0, f_ky = 0, f_kx = 0;                  // (tap, channel chunk) of the NEXT fetch
    auto seek = [&](int ks) { f_tap = ks / cchunks; f_c = (ks - f_tap * cchunks) * BK; f_ky = f_tap / a.KW; f_kx = f_tap - f_ky * a.KW; };
    auto fetch = [&](int ks) {
        const bool first = f_c < a.C1;
        const int cs = first ? a.C1 : a.C2;                                        // scalar selects: no per-lane arrays
        const int soff = ((f_ky * a.W + f_kx) * cs + (first ? f_c : f_c - a.C1) + kq) * 4;
#pragma unroll
        for (int i = 0; i < APASS; ++i) {
            // (the range check of a raw buffer load sees the VGPR offset only, and the origin of a padded window may
            //  lie before the tensor: the tap term is folded into the VGPR offset)
            const int off = ((vmask[i] >> f_tap) & 1u) ? pix[i] * cs * 4 + soff : (int)0x80000000;
            ra[i] = first ? __builtin_bit_cast(f4v, __builtin_amdgcn_raw_buffer_load_b128(rs1, off, 0, 0))
                          : __builtin_bit_cast(f4v, __builtin_amdgcn_raw_buffer_load_b128(rs2, off, 0, 0));
        }
#pragma unroll
        for (int i = 0; i < BPASS; ++i) rb[i] = __builtin_bit_cast(f4v, __builtin_amdgcn_raw_buffer_load_b128(rsw, wbase[i], ks * (BK * 4), 0));
        f_c += BK;
        if (f_c == Cin) { f_c = 0; ++f_tap; if (++f_kx == a.KW) { f_kx = 0; ++f_ky; } }
    };
    auto stash = [&]() {
        if (F16X3) {
#pragma unroll
            for (int i = 0; i < APASS; ++i) {                    // split 4 channels: hi = fp16(x) (0 below the normal range), lo = fp16((x-hi)*2^11)
                h4v hi, lo;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const float x = ra[i][e];
                    const _Float16 h = (fabsf(x) < 6.103515625e-05f) ? (_Float16)0.0f : (_Float16)x;
                    hi[e] = h; lo[e] = (_Float16)((x - (float)h) * 2048.0f);
                }
                *reinterpret_cast<h4v*>(Ah + (lr + 32 * i) * HP + kq) = hi;
                *reinterpret_cast<h4v*>(Al + (lr + 32 * i) * HP + kq) = lo;
            }
            _Float16* bd = ((t & 7) >= 4 ? Bl : Bh) + lr * HP + (t & 3) * 8;
#pragma unroll
            for (int i = 0; i < BPASS; ++i) *reinterpret_cast<f4v*>(bd + 32 * i * HP) = rb[i];
        } else {
#pragma unroll
            for (int i = 0; i < APASS; ++i) *reinterpret_cast<f4v*>(As + (lr + 32 * i) * LDP + kq) = ra[i];
#pragma unroll
            for (int i = 0; i < BPASS; ++i) *reinterpret_cast<f4v*>(Bs + (lr + 32 * i) * LDP + kq) = rb[i];
        }
    };

    f16v acc[TM][TN], acc1[F16X3 ? TM : 1][F16X3 ? TN : 1];      // F16X3: acc = hi.hi, acc1 = hi.lo + lo.hi (scaled by 2^11)
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) { acc[i][j] = (f16v)(0.0f); if (F16X3) acc1[i][j] = (f16v)(0.0f); }

    const int frow = lane & 31, fk = (lane >> 5) * 4;
    const float* Aw = As + (wm * TM * 32 + frow) * LDP + fk;
    const float* Bw = Bs + (wn * TN * 32 + frow) * LDP + fk;

    int ks_begin = 0, ks_end = ksteps;
    if (a.splitk > 1) {
        const int per = (ksteps + a.splitk - 1) / a.splitk;
        ks_begin = blockIdx.y * per; ks_end = min(ksteps, ks_begin + per);
    }
    seek(ks_begin);
    if (ks_begin < ks_end) fetch(ks_begin);
    for (int ks = ks_begin; ks < ks_end; ++ks) {
        __syncthreads();                     // previous step's fragment reads are done
        if (!OMNI_DBG(a, 4)) stash();
        __syncthreads();
        if (ks + 1 < ks_end && !OMNI_DBG(a, 2)) fetch(ks + 1);  // next tiles in flight while the matrix cores work
        if (OMNI_DBG(a, 1)) continue;
        if (F16X3) {
            const int foff = (lane & 31) * HP + (lane >> 5) * 8;   // fragment: row lane&31, k = 8*(lane>>5) .. +7 of a 16-wide chunk
#pragma unroll
            for (int kc = 0; kc < 2; ++kc) {
                h8v ah[TM], al[TM], bh[TN], bl[TN];
#pragma unroll
                for (int i = 0; i < TM; ++i) {
                    ah[i] = *reinterpret_cast<const h8v*>(Ah + ((wm * TM + i) * 32) * HP + foff + kc * 16);
                    al[i] = *reinterpret_cast<const h8v*>(Al + ((wm * TM + i) * 32) * HP + foff + kc * 16);
                }
#pragma unroll
                for (int j = 0; j < TN; ++j) {
                    bh[j] = *reinterpret_cast<const h8v*>(Bh + ((wn * TN + j) * 32) * HP + foff + kc * 16);
                    bl[j] = *reinterpret_cast<const h8v*>(Bl + ((wn * TN + j) * 32) * HP + foff + kc * 16);
                }
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int j = 0; j < TN; ++j) {
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[i], bh[j], acc[i][j], 0, 0, 0);
                        acc1[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[i], bl[j], acc1[i][j], 0, 0, 0);
                        acc1[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al[i], bh[j], acc1[i][j], 0, 0, 0);
                    }
            }
        } else {
#pragma unroll
        for (int kg = 0; kg < BK / 8; ++kg) {
            f4v fa[TM], fb[TN];
#pragma unroll
            for (int i = 0; i < TM; ++i) fa[i] = *reinterpret_cast<const f4v*>(Aw + i * 32 * LDP + kg * 8);
#pragma unroll
            for (int j = 0; j < TN; ++j) fb[j] = *reinterpret_cast<const f4v*>(Bw + j * 32 * LDP + kg * 8);
#pragma unroll
            for (int e = 0; e < 4; ++e)
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int j = 0; j < TN; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[i][e], fb[j][e], acc[i][j], 0, 0, 0);
        }
        }
    }

    // ---- epilogue: D layout col = lane&31, row = (reg&3) + 8*(reg>>2) + 4*(lane>>5)
#pragma unroll
    for (int j = 0; j < TN; ++j) {
        const int col = col0 + (wn * TN + j) * 32 + (lane & 31);
        const float bj = a.bias ? a.bias[col] : 0.0f;
#pragma unroll
        for (int i = 0; i < TM; ++i) {
#pragma unroll
            for (int reg = 0; reg < 16; ++reg) {
                const int r = row0 + (wm * TM + i) * 32 + (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5);
                const float d = F16X3 ? fmaf(acc1[i][j][reg], 4.8828125e-4f, acc[i][j][reg]) : acc[i][j][reg];
                if (r < a.rows && a.splitk > 1) {
                    a.ws[((size_t)blockIdx.y * a.rows + r) * a.Cout + col] = d;
                } else if (r < a.rows) {
                    float v = d + bj;
                    const size_t o = (size_t)r * a.Cout + col;
                    if (a.res) v += a.res[o];
                    if (a.act == OMNI_ACT_RELU) v = fmaxf(v, 0.0f);
                    else if (a.act == OMNI_ACT_GELU) v = 0.5f * v * (1.0f + erff(v * 0.70710678118654752440f));
                    a.dst[o] = v;
                }
            }
        }
    }
}

// ------------------------------------------------------------------ 3x3 stride-1 convolution with halo-tile reuse (f16x3)
// The implicit GEMM above re-fetches every input pixel group once per tap: 9x the input through L2->L1, which is what
// bounds the wide, shallow layers once the matrix-core time is cut by f16x3 (de_conv4_0: 2.7 GB of A traffic for a
// 302 MB input).  Here a block owns a 4x32 pixel tile of ONE image and all BN output channels; per 32-channel group it
// parks the 6x34 halo patch in LDS ONCE (split into hi/lo halfs on the way) and serves the nine taps from it: wave w
// owns image row y0+w (32 pixels = one MFMA row tile), its A fragment for tap (ky,kx) is the same LDS image shifted by
// ky rows and kx pixels.  Weights arrive one kernel row (3 taps) at a time.  Barriers per channel group: 7 instead of 18;
// A traffic 1.6x the input instead of 9x.  Requires W % 32 == 0, H % 4 == 0 (the 32^2 / 64^2 / 128^2 layers).
constexpr int HT_H = 4, HT_W = 32, HPX = (HT_H + 2) * (HT_W + 2);     // 204 halo pixels

template <int BN>
__global__ __launch_bounds__(256, 2) void conv3x3_halo_f16x3_kernel(ConvArgs a)
{
    constexpr int TN = BN / 32;
    __shared__ __attribute__((aligned(16))) _Float16 lds[2 * HPX * HP + 2 * 3 * BN * HP];
    _Float16* Ah = lds;
    _Float16* Al = Ah + HPX * HP;
    _Float16* Bh = Al + HPX * HP;
    _Float16* Bl = Bh + 3 * BN * HP;

    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const int ntn = a.Cout / BN, tw = a.W / HT_W, th = a.H / HT_H;
    int bid = blockIdx.x;
    const int tile_n = bid % ntn; bid /= ntn;
    const int tx = bid % tw; bid /= tw;
    const int ty = bid % th; const int m = bid / th;
    const int y0 = ty * HT_H, x0 = tx * HT_W, col0 = tile_n * BN;
    const int G1 = a.C1 >> 5, G = (a.C1 + a.C2) >> 5, ksteps = 9 * G;

    // A loader: piece q of the halo = (pixel q>>3, 4 channels (q&7)*4); 1632 pieces over 256 threads
    constexpr int AP = (HPX * 8 + 255) / 256;
    int apix[AP]; bool aok[AP];
#pragma unroll
    for (int i = 0; i < AP; ++i) {
        const int q = t + 256 * i, px = q >> 3;
        const int hy = px / (HT_W + 2), hx = px - hy * (HT_W + 2);
        const int iy = y0 - 1 + hy, ix = x0 - 1 + hx;
        aok[i] = (px < HPX) && (unsigned)iy < (unsigned)a.H && (unsigned)ix < (unsigned)a.W;
        apix[i] = aok[i] ? (m * a.H + iy) * a.W + ix : 0;
    }
    const rsrc_t rs1 = make_rsrc(a.src1, (size_t)a.M * a.H * a.W * a.C1 * 4);
    const rsrc_t rs2 = make_rsrc(a.src2 ? a.src2 : a.src1, a.src2 ? (size_t)a.M * a.H * a.W * a.C2 * 4 : 0);
    const rsrc_t rsw = make_rsrc(a.wt, (size_t)a.Cout * ksteps * 128);
    // B loader: 3 taps x BN rows x 8 pieces (0-3 hi, 4-7 lo)
    constexpr int BP = 3 * BN * 8 / 256;
    f16v acc[TN], acc1[TN];
#pragma unroll
    for (int j = 0; j < TN; ++j) { acc[j] = (f16v)(0.0f); acc1[j] = (f16v)(0.0f); }
    const int frag = (lane >> 5) * 8;

    for (int g = 0; g < G; ++g) {
        const bool first = g < G1;
        const int cs = first ? a.C1 : a.C2, cg = (first ? g : g - G1) * 32 + (t & 7) * 4;
        f4v ra[AP];
#pragma unroll
        for (int i = 0; i < AP; ++i) {
            const int off = aok[i] ? (apix[i] * cs + cg) * 4 : (int)0x80000000;
            ra[i] = first ? __builtin_bit_cast(f4v, __builtin_amdgcn_raw_buffer_load_b128(rs1, off, 0, 0))
                          : __builtin_bit_cast(f4v, __builtin_amdgcn_raw_buffer_load_b128(rs2, off, 0, 0));
        }
        __syncthreads();                                          // previous group's fragment reads are done
#pragma unroll
        for (int i = 0; i < AP; ++i) {
            const int q = t + 256 * i;
            if (q < HPX * 8) {
                h4v hi, lo;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const float x = ra[i][e];
                    const _Float16 h = (fabsf(x) < 6.103515625e-05f) ? (_Float16)0.0f : (_Float16)x;
                    hi[e] = h; lo[e] = (_Float16)((x - (float)h) * 2048.0f);
                }
                *reinterpret_cast<h4v*>(Ah + (q >> 3) * HP + (q & 7) * 4) = hi;
                *reinterpret_cast<h4v*>(Al + (q >> 3) * HP + (q & 7) * 4) = lo;
            }
        }
        for (int ky = 0; ky < 3; ++ky) {
            f4v rb[BP];
#pragma unroll
            for (int i = 0; i < BP; ++i) {
                const int q = t + 256 * i, pc = q & 7, row = q >> 3;            // row = kx*BN + cout
                const int kx = row / BN, co = row - kx * BN;
                const int off = (((col0 + co) * ksteps + (ky * 3 + kx) * G + g) * 64 + pc * 8) * 2;
                rb[i] = __builtin_bit_cast(f4v, __builtin_amdgcn_raw_buffer_load_b128(rsw, off, 0, 0));
            }
            if (ky) __syncthreads();                              // previous kernel row's B fragments are consumed
#pragma unroll
            for (int i = 0; i < BP; ++i) {
                const int q = t + 256 * i, pc = q & 7, row = q >> 3;
                *reinterpret_cast<f4v*>((pc >= 4 ? Bl : Bh) + row * HP + (pc & 3) * 8) = rb[i];
            }
            __syncthreads();
#pragma unroll
            for (int kx = 0; kx < 3; ++kx) {
                const int apos = ((wave + ky) * (HT_W + 2) + (lane & 31) + kx) * HP + frag;
#pragma unroll
                for (int kc = 0; kc < 2; ++kc) {
                    const h8v ah = *reinterpret_cast<const h8v*>(Ah + apos + kc * 16);
                    const h8v al = *reinterpret_cast<const h8v*>(Al + apos + kc * 16);
#pragma unroll
                    for (int j = 0; j < TN; ++j) {
                        const int bpos = (kx * BN + j * 32 + (lane & 31)) * HP + frag + kc * 16;
                        const h8v bh = *reinterpret_cast<const h8v*>(Bh + bpos);
                        const h8v bl = *reinterpret_cast<const h8v*>(Bl + bpos);
                        acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bh, acc[j], 0, 0, 0);
                        acc1[j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bl, acc1[j], 0, 0, 0);
                        acc1[j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al, bh, acc1[j], 0, 0, 0);
                    }
                }
            }
        }
    }
    // epilogue: wave owns image row y0+wave; D layout col = lane&31, pixel = (reg&3) + 8*(reg>>2) + 4*(lane>>5)
    const size_t rowbase = ((size_t)(m * a.H + y0 + wave) * a.W + x0) * a.Cout;
#pragma unroll
    for (int j = 0; j < TN; ++j) {
        const int col = col0 + j * 32 + (lane & 31);
        const float bj = a.bias ? a.bias[col] : 0.0f;
#pragma unroll
        for (int reg = 0; reg < 16; ++reg) {
            const int px = (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5);
            const size_t o = rowbase + (size_t)px * a.Cout + col;
            float v = fmaf(acc1[j][reg], 4.8828125e-4f, acc[j][reg]) + bj;
            if (a.res) v += a.res[o];
            if (a.act == OMNI_ACT_RELU) v = fmaxf(v, 0.0f);
            else if (a.act == OMNI_ACT_GELU) v = 0.5f * v * (1.0f + erff(v * 0.70710678118654752440f));
            a.dst[o] = v;
        }
    }
}

template <int BM, int BN, int WM, int WN>
void launch_cfg(const ConvArgs& a, hipStream_t s)
{
    const int grid = ((a.rows + BM - 1) / BM) * (a.Cout / BN);
    if (a.f16x3) hipLaunchKernelGGL((conv_igemm_f32_kernel<BM, BN, WM, WN, true>), dim3(grid, a.splitk > 1 ? a.splitk : 1), dim3(256), 0, s, a);
    else         hipLaunchKernelGGL((conv_igemm_f32_kernel<BM, BN, WM, WN, false>), dim3(grid, a.splitk > 1 ? a.splitk : 1), dim3(256), 0, s, a);
}

// dst = act(sum_s ws[s] + bias + res): the deterministic second pass of a split-K launch (4 channels per thread)
__global__ __launch_bounds__(256) void splitk_reduce_kernel(const float* __restrict__ ws, const float* __restrict__ bias,
                                                            const float* __restrict__ res, float* __restrict__ dst,
                                                            size_t n4, int Cout, int splitk, size_t slab, int act)
{
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n4) return;
    const size_t o = i * 4;
    f4v v = *reinterpret_cast<const f4v*>(ws + o);
    for (int s = 1; s < splitk; ++s) v += *reinterpret_cast<const f4v*>(ws + (size_t)s * slab + o);
    if (bias) v += *reinterpret_cast<const f4v*>(bias + (o % Cout));
    if (res) v += *reinterpret_cast<const f4v*>(res + o);
    if (act == OMNI_ACT_RELU) { v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f); }
    else if (act == OMNI_ACT_GELU) {
        v.x = 0.5f * v.x * (1.0f + erff(v.x * 0.70710678118654752440f)); v.y = 0.5f * v.y * (1.0f + erff(v.y * 0.70710678118654752440f));
        v.z = 0.5f * v.z * (1.0f + erff(v.z * 0.70710678118654752440f)); v.w = 0.5f * v.w * (1.0f + erff(v.w * 0.70710678118654752440f));
    }
    *reinterpret_cast<f4v*>(dst + o) = v;
}

// split factor for a problem: aim at >= 512 blocks when the plain launch leaves most of the 256 CUs idle
int plan_splitk(long long rows, int Cout, int ksteps)
{
    long long blocks;
    if (Cout % 64 != 0) blocks = ((rows + 127) / 128) * (Cout / 32);
    else {
        const long long b128 = ((rows + 127) / 128) * (Cout / 64);
        blocks = b128 >= 512 ? b128 : ((rows + 63) / 64) * (Cout / 64);
    }
    if (blocks >= 384 || ksteps < 8) return 1;
    long long s = (640 + blocks - 1) / blocks;
    if (s > ksteps / 4) s = ksteps / 4;
    if (s > 32) s = 32;
    if (const int cap = omni_options().splitk_max; cap > 0 && s > cap) s = cap;                                 // tuning hook
    return s < 2 ? 1 : (int)s;
}
}  // namespace

// Split factor the library would pick for `rows` output pixels.  Callers that need results that do not depend on the
// batch size bit-for-bit (image-sharded multi-GPU runs) plan with a NOMINAL row count and pass the factor explicitly.
extern "C" int omni_conv2d_splitk_plan(long long rows, int Cout, int ksteps) { return plan_splitk(rows, Cout, ksteps); }

// out[M,Ho,Wo,Cout] = act(conv(src1 ++ src2, wt) + bias + res).  wt is [Cout][KH*KW*(C1+C2)] with k ordered
// (ky, kx, c).  Requirements: C1 % 32 == 0, C2 % 32 == 0, Cout % 32 == 0.  A plain GEMM is the case
// H = W = KH = KW = 1 (rows = M).  With a workspace of omni_conv2d_ws_bytes(...) bytes, problems that would occupy
// only a fraction of the 256 CUs (layer4, the decoder's first stage, every transformer GEMM at M = B*N rows) are
// split along K over blockIdx.y (`splitk` ranges, workspace ws of splitk*rows*Cout floats) and summed by a second,
// deterministic pass.  splitk <= 1: plain launch.
static int conv2d_impl(const float* src1, const float* src2, const void* wt_any, int f16x3, const float* bias,
                       const float* res, float* dst, int M, int H, int W, int C1, int C2, int Cout,
                       int KH, int KW, int stride, int pad, int act, int splitk, float* ws, size_t ws_bytes,
                       omni_stream_t stream)
{
    const float* wt = (const float*)wt_any;
    if (!src1 || !wt || !dst) OMNI_FAIL(OMNI_ERR_INVALID, "omni_conv2d: null pointer");
    if (C1 <= 0 || C1 % BK || C2 < 0 || C2 % BK || Cout <= 0 || Cout % 32 || (C2 > 0 && !src2))
        OMNI_FAIL(OMNI_ERR_INVALID, "omni_conv2d: channels must be multiples of 32");
    if (M <= 0 || H <= 0 || W <= 0 || KH <= 0 || KW <= 0 || KH > 3 || KW > 3 || stride <= 0 || pad < 0)
        OMNI_FAIL(OMNI_ERR_INVALID, "omni_conv2d: bad shape (kernels up to 3x3)");
    ConvArgs a;
    a.src1 = src1; a.src2 = src2; a.wt = wt; a.bias = bias; a.res = res; a.dst = dst;
    a.M = M; a.H = H; a.W = W; a.C1 = C1; a.C2 = C2; a.Cout = Cout;
    a.KH = KH; a.KW = KW; a.stride = stride; a.pad = pad; a.act = act; a.f16x3 = f16x3;
    a.Ho = (H + 2 * pad - KH) / stride + 1; a.Wo = (W + 2 * pad - KW) / stride + 1;
    const long long rows = (long long)M * a.Ho * a.Wo;
    if (rows <= 0 || rows >= (1ll << 31) || (long long)M * H * W >= (1ll << 31))
        OMNI_FAIL(OMNI_ERR_UNSUPPORTED, "omni_conv2d: too many pixels for 32-bit row indices");
    a.rows = (int)rows;
    if ((long long)M * H * W * (C1 > C2 ? C1 : C2) * 4 >= (1ll << 31) || (long long)Cout * KH * KW * (C1 + C2) * 4 >= (1ll << 31))
        OMNI_FAIL(OMNI_ERR_UNSUPPORTED, "omni_conv2d: an operand of 2 GiB or more (32-bit buffer offsets)");
    hipStream_t s = (hipStream_t)stream;
    int S = splitk;
    if (S > KH * KW * ((C1 + C2) / BK)) S = KH * KW * ((C1 + C2) / BK);
    if (S > 1 && (!ws || ws_bytes < (size_t)S * rows * Cout * sizeof(float)))
        OMNI_FAIL(OMNI_ERR_INVALID, "omni_conv2d: split-K workspace too small");
    a.splitk = S; a.ws = ws;
    a.dbg = 0;
#ifdef OMNI_DEBUG_BUILD
    a.dbg = omni_debug_bits("OMNI_CONV_DBG");
#endif
    if (f16x3 && S <= 1 && KH == 3 && KW == 3 && stride == 1 && pad == 1 && W % HT_W == 0 && H % HT_H == 0 && !omni_options().conv_nohalo) {
        const int grid = M * (H / HT_H) * (W / HT_W);
        if (Cout % 64 == 0) hipLaunchKernelGGL((conv3x3_halo_f16x3_kernel<64>), dim3(grid * (Cout / 64)), dim3(256), 0, s, a);
        else                hipLaunchKernelGGL((conv3x3_halo_f16x3_kernel<32>), dim3(grid * (Cout / 32)), dim3(256), 0, s, a);
        OMNI_HIP(hipGetLastError());
        return OMNI_OK;
    }
    // tile choice: the matrix pipes are shared per SIMD, so what matters is how evenly the waves divide over the 1024
    // SIMDs: w = blocks*4/1024 waves per SIMD runs at w/ceil(w) of the saturated rate (five blocks fit a CU).  The 128-row
    // tile re-reads the weights half as often, hence the small bonus.
    auto quant = [](double w) { return w / (w <= 5.0 ? (double)(long long)(w + 0.999999) : (double)(long long)(w / 5.0 + 0.999999) * 5.0); };
    const int Sx = S > 1 ? S : 1;
    if (Cout % 64 != 0) launch_cfg<128, 32, 4, 1>(a, s);
    else {
        const double w128 = (double)(((rows + 127) / 128) * (Cout / 64)) * Sx * 4.0 / 1024.0;
        const double w64 = (double)(((rows + 63) / 64) * (Cout / 64)) * Sx * 4.0 / 1024.0;
        // (f16x3: the 64-row tile keeps five blocks per CU (88 registers) and measured 1-2 % ahead of every mix)
        const bool big = a.f16x3 ? false : quant(w128) * 1.06 >= quant(w64);
        if (big) launch_cfg<128, 64, 4, 1>(a, s);
        else     launch_cfg<64, 64, 2, 2>(a, s);
    }
    OMNI_HIP(hipGetLastError());
    if (S > 1) {
        const size_t n4 = (size_t)rows * Cout / 4;
        hipLaunchKernelGGL(splitk_reduce_kernel, dim3((unsigned)((n4 + 255) / 256)), dim3(256), 0, s, (const float*)ws, bias, res, dst,
                           n4, Cout, S, (size_t)rows * Cout, act);
        OMNI_HIP(hipGetLastError());
    }
    return OMNI_OK;
}

extern "C" int omni_conv2d_nhwc_f32_ws(const float* src1, const float* src2, const float* wt, const float* bias,
                                       const float* res, float* dst, int M, int H, int W, int C1, int C2, int Cout,
                                       int KH, int KW, int stride, int pad, int act, int splitk, float* ws, size_t ws_bytes,
                                       omni_stream_t stream)
{
    return conv2d_impl(src1, src2, wt, 0, bias, res, dst, M, H, W, C1, C2, Cout, KH, KW, stride, pad, act, splitk, ws, ws_bytes, stream);
}

// Same operator with the products on the fp16 matrix cores ("f16x3", see the file header).  wt16: the weights split into
// hi/lo halfs, layout [Cout][KH*KW*(C1+C2)/32][hi32|lo32]; everything else (fp32 NHWC activations, bias, residual,
// split-K) exactly as omni_conv2d_nhwc_f32_ws.
extern "C" int omni_conv2d_nhwc_f16x3_ws(const float* src1, const float* src2, const void* wt16, const float* bias,
                                         const float* res, float* dst, int M, int H, int W, int C1, int C2, int Cout,
                                         int KH, int KW, int stride, int pad, int act, int splitk, float* ws, size_t ws_bytes,
                                         omni_stream_t stream)
{
    return conv2d_impl(src1, src2, wt16, 1, bias, res, dst, M, H, W, C1, C2, Cout, KH, KW, stride, pad, act, splitk, ws, ws_bytes, stream);
}

extern "C" int omni_conv2d_nhwc_f32(const float* src1, const float* src2, const float* wt, const float* bias,
                                    const float* res, float* dst, int M, int H, int W, int C1, int C2, int Cout,
                                    int KH, int KW, int stride, int pad, int act, omni_stream_t stream)
{
    return omni_conv2d_nhwc_f32_ws(src1, src2, wt, bias, res, dst, M, H, W, C1, C2, Cout, KH, KW, stride, pad, act,
                                   1, nullptr, 0, stream);
}
