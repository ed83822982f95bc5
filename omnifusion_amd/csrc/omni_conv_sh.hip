// omni_conv_sh.hip — the convolution of the network on the fp16 matrix cores with SPLIT-HALF activations (gfx950).
//
// Same operator as omni_conv.hip (reference: Conv3d(k,k,1)+BatchNorm3d(+ReLU)(+residual), model/spherical_model.py:
// 122-167 encoder, :29-37,214-222 decoder) and the same "f16x3" arithmetic (x = hi + lo*2^-11, three
// v_mfma_f32_32x32x16_f16 per product block, fp32 accumulation), but the activations travel between layers ALREADY
// split: the "SH" layout stores, per pixel and per group of 32 channels, 32 hi halfs followed by 32 lo halfs (128 bytes:
// the footprint of 32 floats, and the very row format of the pre-split weights).  The split is done once, by the
// producer's epilogue, instead of once per (tap, output-channel tile) by every consumer — in the fp32-activation kernel
// that VALU work cost as much issue time as the matrix instructions (ablation: 41 us of an 82 us layer3 convolution).
//
// With both operands in their final bit pattern the tiles go HBM/L2 -> LDS by LDS-DMA (buffer_load_dwordx4 ... lds):
// no staging registers, no conversion, no ds_write.  A block keeps NST stages of (A: BM pixels x 128 B, B: BN output
// channels x 128 B) in flight; per K-step (one tap x 32 channels) there is ONE barrier:
//      s_waitcnt vmcnt((NST-2)*LPS)   my pieces of stage k have landed       (LPS = DMA instructions per wave and stage)
//      s_barrier                      ... everybody's have, and everybody is done reading stage k-1
//      issue DMA for stage k+NST-1    into the slot stage k-1 occupied
//      8 x ds_read_b128 + 6 x MFMA per 32x32 tile pair on stage k
// Out-of-image taps and rows past the end need no branch: their buffer offset is out of range and the DMA deposits zeros
// (checked on hardware: tools/dbg_dma.py).
//
// LDS image: a DMA instruction deposits its 64 lanes' 16-byte pieces back to back, so rows are 128 B with no padding;
// bank conflicts are avoided by permuting the 16 pieces of each 256-B row pair with the row-pair index (g' = g ^ (d & 15)):
// the lane that owns LDS slot g' of pair d FETCHES piece g' ^ (d & 15) and the fragment reads apply the same involution.
// Every ds_read_b128 lane group then touches 16 distinct 16-byte bank groups.
//
// The matrix instruction is fed weights as its row operand and pixels as its column operand, so a lane ends up with FOUR
// CONSECUTIVE channels of ONE pixel per register quad: bias / residual / output move as 8-byte (SH) or 16-byte (fp32)
// pieces instead of scalars.
#include <stdlib.h>
#include <algorithm>
#include <type_traits>
#include "omni_internal.h"
#include "omni_sh.h"

// Compile-time ablations for tools/convabl.sh (a library variant per value; the product is built with 0): 4 no epilogue | 16, 32, 64 drop the
// weight-lo / activation-lo / hi.hi product | 128 no operand DMA | 256 no block barrier in the K loop | 512 no fragment reads | conv3x3_up2_g1_kernel: 1024 no
// halo arithmetic, 2048 no stores, 4096 no pixel loads, 8192 four accumulators, 16384 no heads part, 32768 time stamps of block 0 (tools/g1_stamps.py; conv3x3_halo_sh_kernel: tools/halo_stamps.py; conv_sh_kernel: tools/tile_stamps.py).  (The debug
// build's RUN-time bits put branches around the matrix instructions and run 2-5x slower than the product: useless for timing.)
#ifndef OMNI_CONV_ABL
#define OMNI_CONV_ABL 0
#endif
#define OMNI_ABL(bit) ((OMNI_CONV_ABL & (bit)) != 0)
#ifndef OMNI_PP_PRIO
#define OMNI_PP_PRIO 0                                         // conv_sh_kernel<.., PP>: s_setprio 1 around a phase's matrix instructions
#endif
#ifndef OMNI_G1_PW
#define OMNI_G1_PW 4                                           // producer waves of conv3x3_up2_g1_kernel<HEADS> (8: measured equal)
#endif

namespace {

typedef float f16v __attribute__((ext_vector_type(16)));
typedef float f4v __attribute__((ext_vector_type(4)));
typedef _Float16 h8v __attribute__((ext_vector_type(8)));
typedef _Float16 h4v __attribute__((ext_vector_type(4)));
typedef _Float16 h2v __attribute__((ext_vector_type(2)));
typedef __attribute__((address_space(3))) void* lptr_t;
typedef __amdgpu_buffer_rsrc_t rsrc_t;

__device__ __forceinline__ rsrc_t make_rsrc(const void* p, size_t bytes)
{
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p), (short)0,
                                             (int)(unsigned)(bytes > 0xffffffffull ? 0xffffffffull : bytes), 0x00020000);
}

// one LDS-DMA instruction: lane l's 16 bytes at buffer offset voff (+ soff, wave-uniform) land at lds + 16 l; offsets
// outside the buffer deposit zeros.  (A plain function: the builtin is not accepted inside a kernel template's body by
// the host pass, which then silently drops the kernel's launch stub.)
__device__ __forceinline__ void dma16(rsrc_t rs, unsigned char* lds, int voff, int soff)
{
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lptr_t)lds, 16, voff, soff, 0, 0);
}

// s_waitcnt vmcnt(N) with a compile-time count (the LDS-DMA pieces still allowed in flight)
template <int N> __device__ __forceinline__ void wait_vm()
{
    static_assert(N >= 0 && N <= 63, "vmcnt is a 6-bit counter");
#define OMNI_VM(K) else if constexpr (N == K) asm volatile("s_waitcnt vmcnt(" #K ")" ::: "memory");
    if constexpr (N < 0) {}
    OMNI_VM(0) OMNI_VM(1) OMNI_VM(2) OMNI_VM(3) OMNI_VM(4) OMNI_VM(5) OMNI_VM(6) OMNI_VM(7)
    OMNI_VM(8) OMNI_VM(9) OMNI_VM(10) OMNI_VM(11) OMNI_VM(12) OMNI_VM(13) OMNI_VM(14) OMNI_VM(15)
    OMNI_VM(16) OMNI_VM(17) OMNI_VM(18) OMNI_VM(19) OMNI_VM(20) OMNI_VM(21) OMNI_VM(22) OMNI_VM(23)
    OMNI_VM(24) OMNI_VM(25) OMNI_VM(26) OMNI_VM(27) OMNI_VM(28) OMNI_VM(29) OMNI_VM(30) OMNI_VM(31)
    OMNI_VM(32) OMNI_VM(33) OMNI_VM(34) OMNI_VM(35) OMNI_VM(36) OMNI_VM(37) OMNI_VM(38) OMNI_VM(39)
    OMNI_VM(40) OMNI_VM(41) OMNI_VM(42) OMNI_VM(43) OMNI_VM(44) OMNI_VM(45) OMNI_VM(46) OMNI_VM(47)
    OMNI_VM(48) OMNI_VM(49) OMNI_VM(50) OMNI_VM(51) OMNI_VM(52) OMNI_VM(53) OMNI_VM(54) OMNI_VM(55)
    OMNI_VM(56) OMNI_VM(57) OMNI_VM(58) OMNI_VM(59) OMNI_VM(60) OMNI_VM(61) OMNI_VM(62) OMNI_VM(63)
#undef OMNI_VM
}

// Every LDS read this wave has issued has returned.  REQUIRED in front of a barrier that licenses another wave to overwrite the
// buffer those reads came from: hipcc sinks the MFMAs that consume a stage's last fragments (and the s_waitcnt lgkmcnt that guards
// them) BELOW the following s_barrier, so without this wait a wave can pass the barrier with ds_reads still queued and a faster
// wave's LDS-DMA for the next stage then lands in the buffer first.  Measured: one wrong output row in 1 of 600 forwards at 8
// panoramas on two streams (1 of 30 at 16) with the 4-wave halo kernel, none in 3000 with the wait (tools/lanes_trace.py).
__device__ __forceinline__ void wait_lds_reads() { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); }

struct ShConvArgs {
    const void* src1; const void* src2;      // SH activations [M,H,W,C1], [M,H,W,C2] (src2 may be null)
    const void* wt;                          // halfs [Cout][KH*KW*(C1+C2)/32][hi32|lo32], BN folded
    const float* bias;                       // [Cout] or null
    const void* res;                         // residual (SH), same shape as dst, or null
    void* dst;                               // [M,Ho,Wo,Cout]: SH (dst_sh) or fp32 NHWC
    int M, H, W, C1, C2, Ho, Wo, Cout;
    int KH, KW, stride, pad, act;
    int rows;                                // M*Ho*Wo
    int dst_sh;
    int res_f32;                             // residual is plain fp32 NHWC instead of SH
    int dbg;                                 // debug build only (OMNI_CONV_DBG): 4 = skip the epilogue
    int noxcd;                               // 1: identity block order (tuning, OMNI_CONV_NOXCD)
    int wt_major;                            // 1: an XCD's contiguous block range walks tile_m fastest — it owns a range of OUTPUT-CHANNEL tiles and touches only their weights (conv_sh_kernel)
    int epi_lds;                             // 1: SH epilogues through an LDS transposition (16-byte pieces), OMNI_CONV_EPI_LDS
    int splitk; float* ws;                   // >1: blockIdx.y owns a K range, raw fp32 partial sums to ws[y][rows][Cout]
    int wino_th, wino_tw, wino_pix;          // WINO kernels: tiles per image (H/2, W/2) and output pixels M*H*W (rows = tiles, Ho / Wo = the image)
    const float* post; unsigned post_rows;   // fp32 [post_rows][Cout] added AFTER the activation, row index modulo post_rows (layer1 + point_feat), or null
};

// Fused epilogue of NT accumulator tiles of ONE pixel row r (D = W x pixels: a lane holds, per register quad q, the four
// consecutive channels c0[j] + 8q + 4(lane>>5) .. +3 of its pixel).  Two phases: every bias / residual load is issued
// before the first store, so the loads overlap instead of serialising load -> wait -> store once per quad.
// QC: register quads of a tile whose loads are in flight together (4 = all; 2 where the register budget is tight)
template <int NT, int QC = 4>
__device__ __forceinline__ void epilogue_row(const f16v (&acc)[NT], const f16v (&acc1)[NT], const ShConvArgs& a, size_t r,
                                             const int (&c0)[NT], int lane, bool dst_sh)
{
    const float* post = a.post ? a.post + (size_t)((unsigned)r % a.post_rows) * a.Cout : nullptr;
#pragma unroll
    for (int q0 = 0; q0 < 4; q0 += QC) {
        f4v bq[NT * QC], rf[NT * QC]; h4v rh[NT * QC], rl[NT * QC];
#pragma unroll
        for (int j = 0; j < NT; ++j)
#pragma unroll
            for (int qq = 0; qq < QC; ++qq) {
                const int q = q0 + qq, c = c0[j] + 8 * q + 4 * (lane >> 5);
                bq[j * QC + qq] = a.bias ? *reinterpret_cast<const f4v*>(a.bias + c) : (f4v)(0.0f);
                if (a.res && a.res_f32) rf[j * QC + qq] = *reinterpret_cast<const f4v*>((const float*)a.res + r * a.Cout + c);
                else if (a.res) {
                    const unsigned char* rp = (const unsigned char*)a.res + sh_off(r * a.Cout + c);
                    rh[j * QC + qq] = *reinterpret_cast<const h4v*>(rp); rl[j * QC + qq] = *reinterpret_cast<const h4v*>(rp + 64);
                }
            }
#pragma unroll
        for (int j = 0; j < NT; ++j)
#pragma unroll
            for (int qq = 0; qq < QC; ++qq) {
                const int q = q0 + qq, c = c0[j] + 8 * q + 4 * (lane >> 5);
                f4v v;
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] = fmaf(acc1[j][4 * q + e], 4.8828125e-4f, acc[j][4 * q + e]);
                v += bq[j * QC + qq];
                if (a.res && a.res_f32) v += rf[j * QC + qq];
                else if (a.res) v += sh_join4(rh[j * QC + qq], rl[j * QC + qq]);
                if (a.act == OMNI_ACT_RELU) { v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f); }
                else if (a.act == OMNI_ACT_GELU) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[e] = 0.5f * v[e] * (1.0f + erff(v[e] * 0.70710678118654752440f));
                }
                if (post) v += *reinterpret_cast<const f4v*>(post + c);
                const size_t o = r * a.Cout + c;
                if (dst_sh) act_store4<true>(a.dst, o, v);
                else        act_store4<false>(a.dst, o, v);
            }
    }
}

// The same epilogue through LDS, for SH outputs (and SH or no residual): a wave's NT accumulator tiles of 32 CONSECUTIVE pixel rows
// r0 .. r0+31 go to a wave-private [32][32 NT + 4] float tile and come back as (pixel, 32-channel group, 8-channel piece) tasks, four
// consecutive lanes per pixel group: the residual arrives and the result leaves as 16-byte pieces, 64 contiguous bytes per pixel and
// half (hi | lo) per instruction.  epilogue_row moves 8 bytes per lane, 16 per pixel and instruction — 4.7 M sixteen-byte requests for
// layer1's 75 MB, which is what its 23-us skeleton is made of.  Same operations on every element in the same order: same bits.
// `tile` = 32 * (32 NT + 4) floats of LDS owned by this wave (the K loop's buffers, after a block barrier).
// (r1: the pixel row of accumulator column 16 when the 32 columns are two runs of 16 consecutive rows — the stem's 2 x 16 tiles; default r0 + 16)
// POST: the caller may carry a post-activation addend (a.post) — only the halo kernel does; the tile kernel compiles the addend's registers
// away.  The tasks are processed HALF at a time (loads of a half issued together, then its arithmetic and stores): the live set is what lets
// conv_sh_kernel<128,128,4,2,3,4> — twelve waves per block, a 168-register budget — run its epilogue without scratch (it carried 236 B).
template <int NT, bool POST = true>
__device__ __forceinline__ void epilogue_tile_lds(const f16v (&acc)[NT], const f16v (&acc1)[NT], const ShConvArgs& a, size_t r0, int nrows,
                                                  const int (&c0)[NT], int lane, float* tile, size_t r1 = ~(size_t)0)
{
    if (r1 == ~(size_t)0) r1 = r0 + 16;
    constexpr int PITCH = 32 * NT + 4;
    {
        const int px = lane & 31;
#pragma unroll
        for (int j = 0; j < NT; ++j)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                f4v v;
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] = fmaf(acc1[j][4 * q + e], 4.8828125e-4f, acc[j][4 * q + e]);
                *reinterpret_cast<f4v*>(tile + px * PITCH + 32 * j + 8 * q + 4 * (lane >> 5)) = v;
            }
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");            // (wave-private tile: the wave's own writes have landed)
    constexpr int TASKS = 32 * NT * 4 / 64;                       // (pixel, group, piece) tasks per lane
    constexpr int HALF = TASKS >= 4 ? TASKS / 2 : TASKS;          // tasks whose loads are in flight together
    const bool post = POST && a.post != nullptr;
#pragma unroll
    for (int k0 = 0; k0 < TASKS; k0 += HALF) {
        f4v va[HALF], vb[HALF], pa[HALF], pb[HALF]; h8v rh[HALF], rl[HALF];
        size_t off[HALF]; bool ok[HALF];
#pragma unroll
        for (int kk = 0; kk < HALF; ++kk) {
            const int task = (k0 + kk) * 64 + lane, px = task / (4 * NT), rem = task - px * (4 * NT), j = rem >> 2, pc = rem & 3;
            ok[kk] = px < nrows;
            va[kk] = *reinterpret_cast<const f4v*>(tile + px * PITCH + 32 * j + 8 * pc);
            vb[kk] = *reinterpret_cast<const f4v*>(tile + px * PITCH + 32 * j + 8 * pc + 4);
            off[kk] = ((px < 16 ? r0 + px : r1 + (px - 16)) * a.Cout + c0[j]) * 4 + 16 * pc;     // byte offset of the hi piece (the lo piece: + 64)
            if (a.bias) { va[kk] += *reinterpret_cast<const f4v*>(a.bias + c0[j] + 8 * pc); vb[kk] += *reinterpret_cast<const f4v*>(a.bias + c0[j] + 8 * pc + 4); }
            if (a.res && ok[kk]) {
                rh[kk] = *reinterpret_cast<const h8v*>((const unsigned char*)a.res + off[kk]);
                rl[kk] = *reinterpret_cast<const h8v*>((const unsigned char*)a.res + off[kk] + 64);
            }
            if (post && ok[kk]) {                                 // added AFTER the activation, as in epilogue_row
                const float* pp = a.post + (size_t)((unsigned)(px < 16 ? r0 + px : r1 + (px - 16)) % a.post_rows) * a.Cout + c0[j] + 8 * pc;
                pa[kk] = *reinterpret_cast<const f4v*>(pp); pb[kk] = *reinterpret_cast<const f4v*>(pp + 4);
            }
        }
#pragma unroll
        for (int kk = 0; kk < HALF; ++kk) {
            if (!ok[kk]) continue;
            f4v v0 = va[kk], v1 = vb[kk];
            if (a.res) {
#pragma unroll
                for (int e = 0; e < 4; ++e) { v0[e] += fmaf((float)rl[kk][e], 4.8828125e-4f, (float)rh[kk][e]); v1[e] += fmaf((float)rl[kk][4 + e], 4.8828125e-4f, (float)rh[kk][4 + e]); }
            }
            if (a.act == OMNI_ACT_RELU) {
#pragma unroll
                for (int e = 0; e < 4; ++e) { v0[e] = fmaxf(v0[e], 0.f); v1[e] = fmaxf(v1[e], 0.f); }
            } else if (a.act == OMNI_ACT_GELU) {
#pragma unroll
                for (int e = 0; e < 4; ++e) { v0[e] = 0.5f * v0[e] * (1.0f + erff(v0[e] * 0.70710678118654752440f)); v1[e] = 0.5f * v1[e] * (1.0f + erff(v1[e] * 0.70710678118654752440f)); }
            }
            if (post) { v0 += pa[kk]; v1 += pb[kk]; }
            h4v h0, l0, h1, l1;
            sh_split4(v0, h0, l0); sh_split4(v1, h1, l1);
            h8v oh, ol;
#pragma unroll
            for (int e = 0; e < 4; ++e) { oh[e] = h0[e]; oh[4 + e] = h1[e]; ol[e] = l0[e]; ol[4 + e] = l1[e]; }
            *reinterpret_cast<h8v*>((unsigned char*)a.dst + off[kk]) = oh;
            *reinterpret_cast<h8v*>((unsigned char*)a.dst + off[kk] + 64) = ol;
        }
    }
}

// dst[o .. o+3] = act(v + bias + res): the tail of a split-K sum (v = the partial sums added in slab order), 4 channels at flat index o
__device__ __forceinline__ void splitk_finish(f4v v, size_t o, const float* __restrict__ bias, const void* __restrict__ res, void* __restrict__ dst,
                                              int Cout, int act, int dst_sh, int res_f32)
{
    if (bias) v += *reinterpret_cast<const f4v*>(bias + (o % Cout));
    if (res && res_f32) v += *reinterpret_cast<const f4v*>((const float*)res + o);
    else if (res) {
        const unsigned char* rp = (const unsigned char*)res + sh_off(o);
        v += sh_join4(*reinterpret_cast<const h4v*>(rp), *reinterpret_cast<const h4v*>(rp + 64));
    }
    if (act == OMNI_ACT_RELU) { v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f); }
    else if (act == OMNI_ACT_GELU) {
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = 0.5f * v[e] * (1.0f + erff(v[e] * 0.70710678118654752440f));
    }
    if (dst_sh) {
        h4v hi, lo; sh_split4(v, hi, lo);
        unsigned char* dp = (unsigned char*)dst + sh_off(o);
        *reinterpret_cast<h4v*>(dp) = hi; *reinterpret_cast<h4v*>(dp + 64) = lo;
    } else {
        *reinterpret_cast<f4v*>((float*)dst + o) = v;
    }
}

// NL > 0: NL extra LOADER waves issue every LDS-DMA piece of the block and wait for them; the WM x WN matrix waves never touch vector memory
// inside the K loop (a piece costs the issuing wave 100-185 cycles between matrix instructions: four pieces per K-step against twelve
// matrix instructions of 32).  Same pieces, same LDS image, same K order: same bits.
//
// PP ("ping-pong", round 6; needs loader waves and 8 matrix waves): the two matrix waves of a SIMD (waves w and w + 4: a block's waves go to the four SIMDs
// in turn) run in ANTI-PHASE.  With one barrier per K-step all eight matrix waves start their fragment reads together — nobody has operands, the matrix
// pipes idle ~350 cycles — and then the two waves of a SIMD run their 12 matrix instructions one after the other: 1 150 cycles per step for 768 of
// matrix work (tools/tile_stamps.py, profiles/r05j_halo_stamps.txt).  Here a K-step is TWO phases behind two barriers: in phase a_k group A (waves 0-3)
// issues its 12 matrix instructions of step k while group B (waves 4-7) reads its fragments of step k; in phase b_k B computes step k and A reads step
// k+1.  Every SIMD's matrix pipe has work in every phase, the LDS port serves four waves' reads (48 KiB) under 384 cycles of matrix work.
//      group A:  b_{k-1} | reads(k)  wait | a_k | mfma(k)          | b_k | reads(k+1) ...
//      group B:  b_{k-1} | mfma(k-1)      | a_k | reads(k)   wait  | b_k | mfma(k)    ...
//      loaders:  wait(stage k landed) | b_{k-1} | issue stage k+NST-1 -> the slot of stage k-1 (B's reads of it ended in front of b_{k-1}) | a_k | ...
// Same pieces, same LDS image, same fragments, same order of the matrix instructions on every accumulator: same bits.
//
// WINO (round 6, experimental): the multiply stage AND the output transform of Winograd F(2x2, 3x3).  The "image" is the transformed input V [16 positions][tiles][C]
// (omni_wino_input_sh), a row of the GEMM is a 2 x 2 output TILE, the K order is (position p, 32-channel group) — sixteen taps whose pixel offset is p * tiles —,
// the weights are U_p = (G g G^T)[p] in the ordinary f16x3 split.  A wave (32 tiles x 32 channels: TM = TN = 1) keeps the four outputs of its tiles in registers:
// when a position's last K-step has been issued, M_p = acc + 2^-11 acc1 is folded into Y[i][j] += A^T[i][xi] A^T[j][nu] M_p (coefficients 0 / +1 / -1) and the
// accumulators start the next position from zero — 16 matrix products per four output pixels instead of 36.  splitk divides the POSITIONS.
__constant__ float wino_coef[16][4] = {                        // [p = 4 xi + nu][o = 2 i + j] = A^T[i][xi] * A^T[j][nu],  A^T = [[1, 1, 1, 0], [0, 1, -1, -1]]
    {1, 0, 0, 0}, {1, 1, 0, 0}, {1, -1, 0, 0}, {0, -1, 0, 0},
    {1, 0, 1, 0}, {1, 1, 1, 1}, {1, -1, 1, -1}, {0, -1, 0, -1},
    {1, 0, -1, 0}, {1, 1, -1, -1}, {1, -1, -1, 1}, {0, -1, 0, 1},
    {0, 0, -1, 0}, {0, 0, -1, -1}, {0, 0, -1, 1}, {0, 0, 0, 1}};
template <int BM, int BN, int WM, int WN, int NST = 3, int NL = 0, bool PP = false, bool WINO = false>           // NST stages in flight (the step loop is unrolled by it)
__global__ __launch_bounds__(64 * (WM * WN + NL)) void conv_sh_kernel(ShConvArgs a)
{
    static_assert(!PP || (NL > 0 && WM * WN == 8 && NST >= 3), "ping-pong: eight matrix waves (two per SIMD) + loader waves, three stages");
    static_assert(!WINO || (PP && BM / WM == 32 && BN / WN == 32), "Winograd: the ping-pong kernel with 32 x 32 wave tiles");
    constexpr int NW = WM * WN, LW = NL > 0 ? NL : NW, RPP = 8 * LW;   // matrix waves; waves that issue DMA; tile rows covered by one DMA pass of the block
    constexpr int TM = BM / WM / 32, TN = BN / WN / 32;         // 32x32 tiles per wave (waves WM x WN)
    constexpr int APASS = BM / RPP, BPASS = BN / RPP, LPS = APASS + BPASS;
    static_assert(APASS >= 1 && BPASS >= 1, "a tile side must cover at least one DMA pass");
    constexpr int A_BYTES = BM * 128, STAGE = (BM + BN) * 128;
    __shared__ __attribute__((aligned(1024))) unsigned char lds[NST * STAGE];
    // (ablation 32768, tools/tile_stamps.py: s_memtime of matrix wave 0 and of the first loader wave of block 0 around the parts of its first 28 K steps, dumped to a.ws)
    __shared__ long long cst[OMNI_ABL(32768) ? 256 : 1];
    const bool stamped = OMNI_ABL(32768) && blockIdx.x == 0 && blockIdx.y == 0 && a.ws != nullptr && a.splitk <= 1;
    auto cstamp = [&](int k) { if (OMNI_ABL(32768) && stamped && (threadIdx.x & 63) == 0 && k < 256) cst[k] = clock64(); };
    auto cdump = [&]() { if (OMNI_ABL(32768) && stamped && threadIdx.x == 0) for (int i = 0; i < 256; ++i) reinterpret_cast<long long*>(a.ws)[i] = cst[i]; };
    if (threadIdx.x == 0) cstamp(0);

    const int t = threadIdx.x, lane = t & 63;
    const int wave = __builtin_amdgcn_readfirstlane(t >> 6);     // wave-uniform: LDS-DMA bases stay in scalar registers
    const int wm = wave / WN, wn = wave % WN;
    const bool loader = NL > 0 && wave >= NW;                   // (wave-uniform)
    const int iw = NL > 0 ? wave - NW : wave;                   // index among the issuing waves (meaningless in a matrix wave when NL > 0)
    const int ntn = a.Cout / BN;
    // XCD-aware order: hardware block b runs on XCD b % 8; give each XCD a contiguous range of (tile_m, tile_n) so that the
    // blocks sharing an A row tile (and neighbouring pixels) share one L2
    // (wt_major — the WEIGHTS are the larger operand: layer4's 9.4 MB against 4.7 MB of pixels, the transformer's matrices against 144 token
    //  rows — an XCD gets a range of output-channel tiles instead and fetches 1/8 of the weights rather than all of them; same tiles, same bits)
    const int ntm = (a.rows + BM - 1) / BM;
    const unsigned lb = a.noxcd ? blockIdx.x : omni_xcd_remap(blockIdx.x, gridDim.x);
    const int tile_m = a.wt_major ? (int)(lb % (unsigned)ntm) : (int)(lb / (unsigned)ntn), tile_n = a.wt_major ? (int)(lb / (unsigned)ntm) : (int)(lb % (unsigned)ntn);
    const int row0 = tile_m * BM, col0 = tile_n * BN;
    const int G1 = a.C1 >> 5, G2 = a.C2 >> 5, G = G1 + G2;
    const int ksteps = a.KH * a.KW * G;

    // ---- DMA geometry: instruction j of a tile covers LDS row pairs 4j .. 4j+3; wave w issues j = w, w+NW, ...  Lane i
    // owns slot g' = i & 15 of pair d = 4j + (i >> 4), i.e. fetches piece g = g' ^ (d & 15) -> row 2d + (g >> 3), 16-byte
    // piece g & 7.  (d & 15 does not depend on the pass, so the lane's piece is fixed and its row advances by 8 NW per pass.)
    const int gs = (lane & 15) ^ ((4 * iw + (lane >> 4)) & 15);
    const int rl = 8 * iw + 2 * (lane >> 4) + (gs >> 3), pc16 = (gs & 7) * 16;
    int pix[APASS];                                              // pixel index of the (possibly padded) window origin
    unsigned vmask[APASS];                                       // bit (ky*KW+kx): tap inside the image
#pragma unroll
    for (int i = 0; i < APASS; ++i) {
        const int r = row0 + rl + RPP * i;
        vmask[i] = 0; pix[i] = 0;
        if constexpr (WINO) {
            if ((NL == 0 || loader) && r < a.rows) { pix[i] = r; vmask[i] = 0xffffu; }      // row = tile; tap p = position: offset p * tiles (a.W), always "inside"
        } else
        if ((NL == 0 || loader) && r < a.rows) {
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
    const rsrc_t rsw = make_rsrc(a.wt, (size_t)a.Cout * ksteps * 128);
    int wbase[BPASS];
#pragma unroll
    for (int i = 0; i < BPASS; ++i) wbase[i] = (col0 + rl + RPP * i) * ksteps * 128 + pc16;

    // ---- issue side.  The K order is (tap, source, 32-channel group); everything that depends on the lane is recomputed
    // only when (tap, source) changes — voff[i] = byte offset of this lane's piece for group 0, or out of range for a tap
    // outside the image — so that a K-step costs one scalar offset and LPS DMA instructions, nothing per lane.  (The matrix
    // pipe retires one MFMA per 32 cycles per SIMD, in which a SIMD has 8 issue slots: a loop with ~20 scalar/vector
    // instructions per MFMA, as the first version of this kernel had, is issue-bound at ~30 % of the MFMA rate.)
    int f_tap = 0, f_src = 0, f_gl = 0, f_ky = 0, f_kx = 0, f_gn = G1;
    int voff[APASS];
    auto refresh = [&]() {
        const int cs4 = (f_src ? a.C2 : a.C1) * 4;
        const int toff = (f_ky * a.W + f_kx) * cs4 + pc16;
#pragma unroll
        for (int i = 0; i < APASS; ++i)
            // (the range check of a raw buffer access sees the VGPR offset only and the origin of a padded window may lie
            //  before the tensor: the tap term is folded into the VGPR offset)
            voff[i] = ((vmask[i] >> f_tap) & 1u) ? pix[i] * cs4 + toff : (int)0x80000000;
    };
    auto seek = [&](int ks) {
        f_tap = ks / G; const int g = ks - f_tap * G;
        f_src = g >= G1; f_gl = f_src ? g - G1 : g; f_gn = f_src ? G2 : G1;
        f_ky = f_tap / a.KW; f_kx = f_tap - f_ky * a.KW;
        refresh();
    };
    // (a stage's pieces in two parts — the pixel operand, then the weights and the step to the next (tap, source, group) — so that the ping-pong schedule
    //  can spread them over its two phases: 32 KiB per K-step through the CU's address path take ~500 cycles, more than one phase's 384 of matrix work)
    auto issue_a = [&](auto slot_c) {
        constexpr int SLOT = decltype(slot_c)::value;
        unsigned char* sb = lds + SLOT * STAGE + iw * 1024;
        const int so = f_gl * 128;
        if (OMNI_ABL(128)) {}                                  // (ablation: no operand traffic)
        else if (f_src) {
#pragma unroll
            for (int i = 0; i < APASS; ++i) dma16(rs2, sb + i * (1024 * LW), voff[i], so);
        } else {
#pragma unroll
            for (int i = 0; i < APASS; ++i) dma16(rs1, sb + i * (1024 * LW), voff[i], so);
        }
    };
    auto issue_b = [&](int ks, auto slot_c) {
        constexpr int SLOT = decltype(slot_c)::value;
        unsigned char* sb = lds + SLOT * STAGE + iw * 1024;
        if (!OMNI_ABL(128)) {
#pragma unroll
            for (int i = 0; i < BPASS; ++i) dma16(rsw, sb + A_BYTES + i * (1024 * LW), wbase[i], ks * 128);
        }
        if (++f_gl == f_gn) {
            f_gl = 0;
            if (f_src == 0 && G2 > 0) { f_src = 1; f_gn = G2; }
            else { f_src = 0; f_gn = G1; ++f_tap; if (++f_kx == a.KW) { f_kx = 0; ++f_ky; } }
            refresh();
        }
    };
    auto issue = [&](int ks, auto slot_c) { issue_a(slot_c); issue_b(ks, slot_c); };

    f16v acc[TM][TN], acc1[TM][TN];                              // acc = hi.hi, acc1 = hi.lo + lo.hi (scaled by 2^11)
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) { acc[i][j] = (f16v)(0.0f); acc1[i][j] = (f16v)(0.0f); }

    // fragment offsets: row lane&31 of a 32-row tile, piece p = 2k + (lane>>5): k = 0,1 hi of the two 16-wide k chunks, 2,3 lo;
    // the wave's tile origin is folded in, the stage offset is an immediate of the unrolled step
    int foa[4], fob[4];
    {
        const int r = lane & 31, v = r >> 1, h = lane >> 5;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int f = v * 256 + ((((r & 1) * 8 + 2 * k + h) ^ v) * 16);
            foa[k] = f + wm * (BM / WM) * 128;
            fob[k] = f + A_BYTES + wn * (BN / WN) * 128;
        }
    }

    int ks_begin = 0, ks_end = ksteps;
    if (a.splitk > 1) {
        const int per = (ksteps + a.splitk - 1) / a.splitk;
        ks_begin = (int)blockIdx.y * per; ks_end = min(ksteps, ks_begin + per);
    }
    if (NL == 0 || loader) {
        seek(ks_begin);
        [&]<int... S>(std::integer_sequence<int, S...>) {        // prologue: stages 0 .. NST-2 in flight
            ((ks_begin + S < ks_end ? issue(ks_begin + S, std::integral_constant<int, S>()) : (void)0), ...);
        }(std::make_integer_sequence<int, NST - 1>());
    }
    auto pbarrier = [&]() {                                      // a phase boundary of the ping-pong schedule: nothing moves across it
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
    };
    if (loader) {
        if constexpr (PP) {
            auto lstep = [&](int ks, auto slot_c) {
                constexpr int SLOT = decltype(slot_c)::value;
                const int sk_ = 128 + 4 * (ks - ks_begin);
                if (wave == NW && ks - ks_begin < 28) cstamp(sk_);
                if (ks + NST - 2 < ks_end) wait_vm<(NST - 2) * LPS>();
                else                       wait_vm<0>();
                if (wave == NW && ks - ks_begin < 28) cstamp(sk_ + 1);
                pbarrier();                                       // b_{k-1}: stage ks has landed for everybody; group B is done with stage ks-1
                if (wave == NW && ks - ks_begin < 28) cstamp(sk_ + 2);
                if (ks + NST - 1 < ks_end) issue_a(std::integral_constant<int, (SLOT + NST - 1) % NST>());             // half of the pieces in each phase
                if (wave == NW && ks - ks_begin < 28) cstamp(sk_ + 3);
                pbarrier();                                       // a_k
                if (ks + NST - 1 < ks_end) issue_b(ks + NST - 1, std::integral_constant<int, (SLOT + NST - 1) % NST>());
            };
            int ks = ks_begin;
            for (; ks + NST - 1 < ks_end; ks += NST) {
                [&]<int... S>(std::integer_sequence<int, S...>) { (lstep(ks + S, std::integral_constant<int, S>()), ...); }
                (std::make_integer_sequence<int, NST>());
            }
            [&]<int... S>(std::integer_sequence<int, S...>) {
                ((ks + S < ks_end ? lstep(ks + S, std::integral_constant<int, S>()) : (void)0), ...);
            }(std::make_integer_sequence<int, NST - 1>());
            pbarrier();                                           // b_{n-1}
            if (!WINO && a.splitk <= 1 && a.dst_sh && !a.res_f32 && !a.post && a.epi_lds) __syncthreads();    // (the barrier in front of the LDS epilogue)
            return;
        }
        // ---- a loader wave's K loop: my pieces of stage ks have landed -> barrier (everybody's have; the matrix waves are done with stage
        // ks-1) -> the pieces of stage ks+NST-1 into the slot stage ks-1 occupied
        auto lstep = [&](int ks, auto slot_c) {
            constexpr int SLOT = decltype(slot_c)::value;
            const int sk_ = 128 + 4 * (ks - ks_begin);
            if (wave == NW && ks - ks_begin < 28) cstamp(sk_);
            if (ks + NST - 2 < ks_end) wait_vm<(NST - 2) * LPS>();
            else                       wait_vm<0>();
            if (wave == NW && ks - ks_begin < 28) cstamp(sk_ + 1);
            __builtin_amdgcn_s_barrier();
            asm volatile("" ::: "memory");
            if (wave == NW && ks - ks_begin < 28) cstamp(sk_ + 2);
            if (ks + NST - 1 < ks_end) issue(ks + NST - 1, std::integral_constant<int, (SLOT + NST - 1) % NST>());
            if (wave == NW && ks - ks_begin < 28) cstamp(sk_ + 3);
        };
        int ks = ks_begin;
        for (; ks + NST - 1 < ks_end; ks += NST) {
            [&]<int... S>(std::integer_sequence<int, S...>) { (lstep(ks + S, std::integral_constant<int, S>()), ...); }
            (std::make_integer_sequence<int, NST>());
        }
        [&]<int... S>(std::integer_sequence<int, S...>) {
            ((ks + S < ks_end ? lstep(ks + S, std::integral_constant<int, S>()) : (void)0), ...);
        }(std::make_integer_sequence<int, NST - 1>());
        if (a.splitk <= 1 && a.dst_sh && !a.res_f32 && !a.post && a.epi_lds) __syncthreads();    // (the barrier in front of the LDS epilogue)
        return;
    }

    if constexpr (PP) {
        const bool grp_b = wave >= NW / 2;                           // (wave-uniform) waves w and w + NW/2 share a SIMD
        h8v ah[2][TM], al[2][TM], bh[2][TN], bl[2][TN];
        auto reads = [&](auto slot_c) {
            constexpr int SLOT = decltype(slot_c)::value;
            const unsigned char* sl = lds + SLOT * STAGE;
#pragma unroll
            for (int kc = 0; kc < 2; ++kc) {
#pragma unroll
                for (int i = 0; i < TM; ++i) {
                    ah[kc][i] = *reinterpret_cast<const h8v*>(sl + i * 4096 + foa[kc]);
                    al[kc][i] = *reinterpret_cast<const h8v*>(sl + i * 4096 + foa[2 + kc]);
                }
#pragma unroll
                for (int j = 0; j < TN; ++j) {
                    bh[kc][j] = *reinterpret_cast<const h8v*>(sl + j * 4096 + fob[kc]);
                    bl[kc][j] = *reinterpret_cast<const h8v*>(sl + j * 4096 + fob[2 + kc]);
                }
            }
            wait_lds_reads();                                        // the fragments are in registers before the phase ends (the buffer may be refilled two phases later)
        };
        auto mfmas = [&]() {
            if constexpr (OMNI_PP_PRIO) __builtin_amdgcn_s_setprio(1);
#pragma unroll
            for (int kc = 0; kc < 2; ++kc)
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int j = 0; j < TN; ++j) {
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(bh[kc][j], ah[kc][i], acc[i][j], 0, 0, 0);
                        acc1[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(bl[kc][j], ah[kc][i], acc1[i][j], 0, 0, 0);
                        acc1[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(bh[kc][j], al[kc][i], acc1[i][j], 0, 0, 0);
                    }
            if constexpr (OMNI_PP_PRIO) __builtin_amdgcn_s_setprio(0);
        };
        f16v Y[WINO ? 4 : 1];
        int wino_g = 0, wino_p = ks_begin / (G > 0 ? G : 1);        // K-steps issued of the current position; the position
        if constexpr (WINO) {
#pragma unroll
            for (int o = 0; o < 4; ++o) Y[o] = (f16v)(0.0f);
        }
        auto wino_fold = [&]() {                                     // behind a K-step's matrix instructions: the position's last?  fold it into the four outputs
            if constexpr (WINO) {
                if (++wino_g < G) return;
                wino_g = 0;
                f16v m;
#pragma unroll
                for (int e = 0; e < 16; ++e) m[e] = fmaf(acc1[0][0][e], 4.8828125e-4f, acc[0][0][e]);
#pragma unroll
                for (int o = 0; o < 4; ++o) {
                    const float c = wino_coef[wino_p][o];            // (wave-uniform: a scalar load)
                    if (c != 0.0f) {
#pragma unroll
                        for (int e = 0; e < 16; ++e) Y[o][e] = fmaf(c, m[e], Y[o][e]);
                    }
                }
                acc[0][0] = (f16v)(0.0f); acc1[0][0] = (f16v)(0.0f);
                ++wino_p;
            }
        };
        pbarrier();                                                  // b_{-1}: stage ks_begin has landed
        if (!grp_b) {
            auto stepa = [&](int ks, auto slot_c) {
                if (wave == 0 && ks - ks_begin < 28) cstamp(8 + 4 * (ks - ks_begin));
                reads(slot_c);
                if (wave == 0 && ks - ks_begin < 28) cstamp(11 + 4 * (ks - ks_begin));
                pbarrier();                                          // a_k
                if (wave == 0 && ks - ks_begin < 28) cstamp(9 + 4 * (ks - ks_begin));
                mfmas();
                wino_fold();
                if (wave == 0 && ks - ks_begin < 28) cstamp(10 + 4 * (ks - ks_begin));
                pbarrier();                                          // b_k
            };
            int ks = ks_begin;
            for (; ks + NST - 1 < ks_end; ks += NST) {
                [&]<int... S>(std::integer_sequence<int, S...>) { (stepa(ks + S, std::integral_constant<int, S>()), ...); }
                (std::make_integer_sequence<int, NST>());
            }
            [&]<int... S>(std::integer_sequence<int, S...>) {
                ((ks + S < ks_end ? stepa(ks + S, std::integral_constant<int, S>()) : (void)0), ...);
            }(std::make_integer_sequence<int, NST - 1>());
        } else {
            auto stepb = [&](int ks, auto slot_c) {
                pbarrier();                                          // a_k
                reads(slot_c);
                pbarrier();                                          // b_k
                mfmas();
                wino_fold();
            };
            int ks = ks_begin;
            for (; ks + NST - 1 < ks_end; ks += NST) {
                [&]<int... S>(std::integer_sequence<int, S...>) { (stepb(ks + S, std::integral_constant<int, S>()), ...); }
                (std::make_integer_sequence<int, NST>());
            }
            [&]<int... S>(std::integer_sequence<int, S...>) {
                ((ks + S < ks_end ? stepb(ks + S, std::integral_constant<int, S>()) : (void)0), ...);
            }(std::make_integer_sequence<int, NST - 1>());
        }
        if constexpr (WINO) {
            // ---- the four output pixels of my tile (column lane & 31): tile -> (image, tile row, tile column) -> pixel (2 ty + i, 2 tx + j); per register quad the
            // lane holds four consecutive channels, as in every other epilogue.  splitk > 1 (the positions were divided): raw partial outputs to the workspace,
            // pixel-major like every split-K launch, for sh_splitk_reduce_kernel.
            const int tl = row0 + wm * (BM / WM) + (lane & 31);
            if (tl >= a.rows) return;
            const int per_img = a.wino_th * a.wino_tw, m = tl / per_img, rem = tl - m * per_img, ty = rem / a.wino_tw, tx = rem - ty * a.wino_tw;
            const int cj[1] = {col0 + wn * (BN / WN)};
#pragma unroll
            for (int o = 0; o < 4; ++o) {
                const size_t r = ((size_t)m * a.Ho + 2 * ty + (o >> 1)) * a.Wo + 2 * tx + (o & 1);
                if (a.splitk > 1) {
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        f4v v; v.x = Y[o][4 * q]; v.y = Y[o][4 * q + 1]; v.z = Y[o][4 * q + 2]; v.w = Y[o][4 * q + 3];
                        *reinterpret_cast<f4v*>(a.ws + ((size_t)blockIdx.y * a.wino_pix + r) * a.Cout + cj[0] + 8 * q + 4 * (lane >> 5)) = v;
                    }
                } else {
                    const f16v ea[1] = {Y[o]}, eb[1] = {(f16v)(0.0f)};
                    epilogue_row<1, 2>(ea, eb, a, r, cj, lane, a.dst_sh != 0);
                }
            }
            return;
        }
    }
    auto step = [&](int ks, auto slot_c) {
        constexpr int SLOT = decltype(slot_c)::value;
        // my pieces of stage ks have landed when at most the NST-2 younger stages are in flight (near the end fewer were issued:
        // a smaller count only waits longer, 0 is always safe)
        if (NL == 0) {
            if (ks + NST - 2 < ks_end) wait_vm<(NST - 2) * LPS>();
            else                       wait_vm<0>();
        }
        if (wave == 0 && ks - ks_begin < 28) cstamp(8 + 4 * (ks - ks_begin));
        wait_lds_reads();                                        // my fragment reads of stage ks-1 have returned ...
        if (!OMNI_ABL(256)) __builtin_amdgcn_s_barrier();     // ... everybody's pieces have landed; everybody is done reading stage ks-1
        asm volatile("" ::: "memory");
        if (wave == 0 && ks - ks_begin < 28) cstamp(9 + 4 * (ks - ks_begin));
        const unsigned char* sl = lds + SLOT * STAGE;
        h8v ah[2][TM], al[2][TM], bh[2][TN], bl[2][TN];
        if (OMNI_ABL(512)) {                                  // (ablation: no fragment reads)
#pragma unroll
            for (int kc = 0; kc < 2; ++kc) {
#pragma unroll
                for (int i = 0; i < TM; ++i) { ah[kc][i] = (h8v)((_Float16)1.0f); al[kc][i] = ah[kc][i]; }
#pragma unroll
                for (int j = 0; j < TN; ++j) { bh[kc][j] = (h8v)((_Float16)1.0f); bl[kc][j] = bh[kc][j]; }
            }
        } else
#pragma unroll
        for (int kc = 0; kc < 2; ++kc) {
#pragma unroll
            for (int i = 0; i < TM; ++i) {
                ah[kc][i] = *reinterpret_cast<const h8v*>(sl + i * 4096 + foa[kc]);
                al[kc][i] = *reinterpret_cast<const h8v*>(sl + i * 4096 + foa[2 + kc]);
            }
#pragma unroll
            for (int j = 0; j < TN; ++j) {
                bh[kc][j] = *reinterpret_cast<const h8v*>(sl + j * 4096 + fob[kc]);
                bl[kc][j] = *reinterpret_cast<const h8v*>(sl + j * 4096 + fob[2 + kc]);
            }
        }
#pragma unroll
        for (int kc = 0; kc < 2; ++kc) {
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j) {
                    if (!OMNI_ABL(64) && !OMNI_DBG(a, 64)) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(bh[kc][j], ah[kc][i], acc[i][j], 0, 0, 0);
                    else { acc[i][j][0] += (float)bh[kc][j][0] * (float)ah[kc][i][0]; }
                    if (!OMNI_ABL(16) && !OMNI_DBG(a, 16)) acc1[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(bl[kc][j], ah[kc][i], acc1[i][j], 0, 0, 0);   // precision map: weight-lo term
                    if (!OMNI_ABL(32) && !OMNI_DBG(a, 32)) acc1[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(bh[kc][j], al[kc][i], acc1[i][j], 0, 0, 0);   // ... activation-lo term
                }
            if (kc == 0) {                                       // stage ks+NST-1, issued under the first half's matrix work
                __builtin_amdgcn_sched_barrier(0);
                if (NL == 0 && ks + NST - 1 < ks_end) issue(ks + NST - 1, std::integral_constant<int, (SLOT + NST - 1) % NST>());
            }
        }
    };
    if constexpr (!PP) {
        int ks = ks_begin;
        for (; ks + NST - 1 < ks_end; ks += NST) {               // (no early exits inside: they would park the accumulators in VGPRs)
            [&]<int... S>(std::integer_sequence<int, S...>) { (step(ks + S, std::integral_constant<int, S>()), ...); }
            (std::make_integer_sequence<int, NST>());
        }
        [&]<int... S>(std::integer_sequence<int, S...>) {        // remainder: up to NST-1 steps
            ((ks + S < ks_end ? step(ks + S, std::integral_constant<int, S>()) : (void)0), ...);
        }(std::make_integer_sequence<int, NST - 1>());
    }

    if (wave == 0) cstamp(1);                                     // K loop issued
    if (OMNI_ABL(4) || OMNI_DBG(a, 4)) return;
    // ---- epilogue.  D = W x pixels: column (lane & 31) = pixel, row (reg & 3) + 8 (reg >> 2) + 4 (lane >> 5) = channel
    if (a.splitk <= 1 && a.dst_sh && !a.res_f32 && !a.post && a.epi_lds) {       // (`post`: only the halo kernel takes it through LDS — the registers it costs would spill here) split-half output: through LDS, 16-byte pieces (epilogue_tile_lds)
        static_assert(NW * 32 * (32 * TN + 4) * 4 <= NST * STAGE, "the transposition tiles must fit the K loop's buffers");
        wait_lds_reads();
        __syncthreads();                                          // every wave is done with the last stage
        float* tile = reinterpret_cast<float*>(lds) + wave * (32 * (32 * TN + 4));
        int c0[TN];
#pragma unroll
        for (int j = 0; j < TN; ++j) c0[j] = col0 + wn * (BN / WN) + j * 32;
#pragma unroll
        for (int i = 0; i < TM; ++i) {
            const int r0 = row0 + wm * (BM / WM) + i * 32;
            if (r0 < a.rows) epilogue_tile_lds<TN, false>(acc[i], acc1[i], a, (size_t)r0, min(32, a.rows - r0), c0, lane, tile);
            if (i + 1 < TM) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");    // (the tile is read back before it is written again)
        }
        if (OMNI_ABL(32768)) { if (wave == 0) cstamp(2); __syncthreads(); cdump(); }
        return;
    }
#pragma unroll
    for (int i = 0; i < TM; ++i) {
        const int r = row0 + wm * (BM / WM) + i * 32 + (lane & 31);
        if (r >= a.rows) continue;
        if (a.splitk > 1) {
#pragma unroll
            for (int j = 0; j < TN; ++j)
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const int c = col0 + wn * (BN / WN) + j * 32 + 8 * q + 4 * (lane >> 5);
                    f4v v;
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[e] = fmaf(acc1[i][j][4 * q + e], 4.8828125e-4f, acc[i][j][4 * q + e]);
                    *reinterpret_cast<f4v*>(a.ws + ((size_t)blockIdx.y * a.rows + r) * a.Cout + c) = v;
                }
        } else {
            if constexpr (NL > 0 && TN > 1) {
                // twelve waves per block = a 168-register budget: one channel tile at a time (4 instead of 4 TN bias / residual quads in flight)
#pragma unroll
                for (int j = 0; j < TN; ++j) {
                    const f16v ea[1] = {acc[i][j]}, eb[1] = {acc1[i][j]};
                    const int cj[1] = {col0 + wn * (BN / WN) + j * 32};
                    epilogue_row<1, 2>(ea, eb, a, (size_t)r, cj, lane, a.dst_sh != 0);
                }
            } else {
                int c0[TN];
#pragma unroll
                for (int j = 0; j < TN; ++j) c0[j] = col0 + wn * (BN / WN) + j * 32;
                epilogue_row<TN>(acc[i], acc1[i], a, (size_t)r, c0, lane, a.dst_sh != 0);
            }
        }
    }
}

// ------------------------------------------------------------------ 3x3 stride-1 convolution with halo-tile reuse (SH)
// The tile kernel above fetches every input pixel group once per tap (9x the input through L2 -> LDS), which is what
// bounds the wide, shallow decoder layers (de_conv3_x, de_conv4_0: 64^2 / 128^2 images, 32-128 channels).  Here a block
// owns a 4x32 pixel tile of ONE image and BN output channels; per 32-channel group it DMAs the 6x34 halo patch into LDS
// ONCE (pixel-major 128-B rows, same pair swizzle, out-of-image pixels arrive as zeros) and serves the nine taps from it:
// wave w owns image row y0+w (32 pixels = one MFMA column tile), its pixel fragment for tap (ky,kx) is the same LDS image
// shifted by ky rows and kx pixels.  The weights arrive one kernel row (3 taps) at a time through a double buffer: the
// next row's DMA is in flight under the 18*BN/32 MFMAs of the current one.  Requires W % 32 == 0, H % 4 == 0.
constexpr int HT_W = 32, HPW = HT_W + 2;

// TH = image rows per block = waves per block (4: 6x34 halo, 26 KiB; 8: 10x34 halo, 43 KiB, half the weight traffic per pixel)
//
// UP2: the convolution of the 2x bilinear up-sampling of src1 ([M, H/2, W/2, C1]; F.interpolate(align_corners=False) followed by
// ConvBnReLU, model/spherical_model.py:279-301) without the up-sampled tensor ever existing: the halo patch is COMPUTED into LDS
// instead of copied.  The 6 x 34 halo pixels are 3 x 17 cells of 2 x 2 pixels that share their four source pixels; thread
// (cell, 8 channels) loads those once (8 x 16 B), joins hi/lo, evaluates up-sample_sh8_kernel's expression for its 4 pixels and
// writes the 8 split pieces where the DMA would have put them (out-of-image halo pixels: zeros, the convolution's padding).
// Same arithmetic, same bits as the two kernels it replaces; one pass over HBM less in each direction for the widest tensors.
//
// IW > 0: images narrower than a 32-pixel tile row (layer2-4 and the first decoder stages: 16 x 16, 8 x 8, 4 x 4).  The tile is TH*32
// CONSECUTIVE pixels of the flattened [M, H, W] index — NSUB bands of SUBROWS whole image rows (8 rows of a 16 x 16 image; two 8 x 8 or
// eight 4 x 4 images) — each band with its own (SUBROWS+2) x (IW+2) halo in LDS; wave w owns pixels 32w .. 32w+31 of the tile.  Against
// conv_sh_kernel's im2col tiles (every pixel group fetched once per tap) a K-step brings the weights only: 0.6x the LDS-DMA pieces per
// matrix instruction at 128 x 128, which is what bounds those layers (tools/convabl.sh: the operand traffic of a layer3 convolution costs
// as much time as its matrix instructions and overlaps them for a third).  Needs H == W == IW and rows % (TH*32) == 0.
template <int BN, int TH, bool UP2 = false, int IW = 0>
__global__ __launch_bounds__(64 * TH, (TH == 8 && BN == 32) ? 4 : 1) void conv3x3_halo_sh_kernel(ShConvArgs a)      // (8 rows x 32 channels: 128 registers, two 8-wave blocks per CU)
{
    static_assert(!UP2 || (TH == 4 && IW == 0), "the cell decomposition of the up-sampling halo is written for 4-row tiles of wide images");
    constexpr int TN = BN / 32, NW = TH, RPP = 8 * NW;
    constexpr int IWD = IW > 0 ? IW : 1;
    constexpr int SUBROWS = (TH * 32 / IWD) < IWD ? (TH * 32 / IWD) : IWD, SUBPX = SUBROWS * IWD, NSUB = TH * 32 / SUBPX;
    constexpr int HPS = (SUBROWS + 2) * (IWD + 2);               // halo pixels of one band
    static_assert(IW == 0 || (NSUB * SUBPX == TH * 32 && IW * IW % SUBPX == 0), "bands must tile the images");
    constexpr int HPX = IW > 0 ? NSUB * HPS : (TH + 2) * HPW, HA_INSTR = (HPX * 8 + 63) / 64, HA_BYTES = HA_INSTR * 1024;
    constexpr int APASS = (HA_INSTR + NW - 1) / NW, BROWS = 3 * BN, BPASS = (BROWS + RPP - 1) / RPP, B_BYTES = BROWS * 128;
    __shared__ __attribute__((aligned(1024))) unsigned char lds[HA_BYTES + 2 * B_BYTES];
    // (ablation 32768, tools/halo_stamps.py: s_memtime of wave 0 of blocks 0 and 600 — start, prologue done, per K stage: before its waits / behind the barrier / weights
    //  issued / matrix instructions issued, epilogue done — dumped to a.ws)
    __shared__ long long hst[OMNI_ABL(32768) ? 64 : 1];
    const bool stamped = OMNI_ABL(32768) && (blockIdx.x == 0 || blockIdx.x == 600) && a.ws != nullptr;
    auto hstamp = [&](int k) { if (OMNI_ABL(32768) && stamped && threadIdx.x == 0 && k < 64) hst[k] = clock64(); };
    auto hdump = [&]() { if (OMNI_ABL(32768) && stamped && threadIdx.x == 0) for (int i = 0; i < 64; ++i) reinterpret_cast<long long*>(a.ws)[(blockIdx.x ? 64 : 0) + i] = hst[i]; };
    hstamp(0);

    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const int ntn = a.Cout / BN, tw = IW > 0 ? 1 : a.W / HT_W, th = IW > 0 ? 1 : a.H / TH;
    int bid = a.noxcd ? blockIdx.x : omni_xcd_remap(blockIdx.x, gridDim.x);   // neighbouring tiles (shared halos, same A for all tile_n) on one XCD
    const int tile_n = bid % ntn; bid /= ntn;
    const int tx = bid % tw; bid /= tw;
    const int ty = bid % th; const int m = bid / th;              // (IW > 0: m = tile index, first pixel m * TH * 32)
    const int y0 = ty * TH, x0 = tx * HT_W, col0 = tile_n * BN;
    const int G1 = a.C1 >> 5, G = (a.C1 + a.C2) >> 5, ksteps = 9 * G;
    const int pix0 = m * (TH * 32);                               // IW > 0: flattened index of the tile's first pixel

    // DMA geometry (as in conv_sh_kernel): lane -> row rl + 32*pass of the region, 16-byte piece pc16/16
    const int gs = (lane & 15) ^ ((4 * wave + (lane >> 4)) & 15);
    const int rl = 8 * wave + 2 * (lane >> 4) + (gs >> 3), pc16 = (gs & 7) * 16;
    int apix[APASS];                                              // image pixel index of halo pixel rl + 32*i, or -1
#pragma unroll
    for (int i = 0; i < APASS; ++i) {
        const int p = rl + RPP * i;
        if constexpr (IW > 0) {
            const int sb = p / HPS, q = p - sb * HPS, hy = q / (IW + 2), hx = q - hy * (IW + 2);
            const int first = pix0 + sb * SUBPX;                  // first pixel of the band: image first / IW^2, image row (first % IW^2) / IW
            const int img = first / (IW * IW), gy0 = (first - img * (IW * IW)) / IW;
            const int iy = gy0 - 1 + hy, ix = hx - 1;
            apix[i] = (p < HPX && (unsigned)iy < (unsigned)IW && (unsigned)ix < (unsigned)IW) ? (img * IW + iy) * IW + ix : -1;
        } else {
            const int hy = p / HPW, hx = p - hy * HPW;
            const int iy = y0 - 1 + hy, ix = x0 - 1 + hx;
            apix[i] = (p < HPX && (unsigned)iy < (unsigned)a.H && (unsigned)ix < (unsigned)a.W) ? (m * a.H + iy) * a.W + ix : -1;
        }
    }
    int wbase[BPASS];                                             // weight row (kx, co) = row rl + 32*i of a kernel-row stage
#pragma unroll
    for (int i = 0; i < BPASS; ++i) {
        const int r = rl + RPP * i, kx = r / BN, co = r - kx * BN;
        wbase[i] = r < BROWS ? ((col0 + co) * ksteps + kx * G) * 128 + pc16 : (int)0x80000000;
    }
    const rsrc_t rs1 = make_rsrc(a.src1, (size_t)a.M * a.H * a.W * a.C1 * 4);
    const rsrc_t rs2 = make_rsrc(a.src2 ? a.src2 : a.src1, a.src2 ? (size_t)a.M * a.H * a.W * a.C2 * 4 : 0);
    const rsrc_t rsw = make_rsrc(a.wt, (size_t)a.Cout * ksteps * 128);

    auto issue_a = [&](int g) {
        const bool first = g < G1;
        const int cs4 = (first ? a.C1 : a.C2) * 4;
        const int soff = (first ? g : g - G1) * 128 + pc16;
        unsigned char* sb = lds + wave * 1024;
#pragma unroll
        for (int i = 0; i < APASS; ++i) {
            if (wave + NW * i < HA_INSTR) {
                const int off = apix[i] >= 0 ? apix[i] * cs4 + soff : (int)0x80000000;
                if (OMNI_ABL(128)) {}
                else if (first) dma16(rs1, sb + i * (1024 * NW), off, 0);
                else       dma16(rs2, sb + i * (1024 * NW), off, 0);
            }
        }
    };
    // ---- UP2: this thread's cell of the halo and the byte offsets of its four source pixels
    const int Hl = a.H >> 1, Wl = a.W >> 1;
    const int u_c8 = t & 3, u_cell = t >> 2, u_ci = u_cell / 17, u_cj = u_cell - u_ci * 17;
    const int u_k = (y0 >> 1) - 1 + u_ci, u_j = (x0 >> 1) - 1 + u_cj;
    size_t u_src[4];
    float u_ly[2], u_lx[2];
    bool u_in[2][2];
    if constexpr (UP2) {
        const int ra = min(max(u_k, 0), Hl - 1), rb = min(max(u_k + 1, 0), Hl - 1), ca = min(max(u_j, 0), Wl - 1), cb = min(max(u_j + 1, 0), Wl - 1);
        const size_t pp = (size_t)a.C1 * 4, img = (size_t)m * Hl * Wl;
        u_src[0] = (img + (size_t)ra * Wl + ca) * pp + u_c8 * 16; u_src[1] = (img + (size_t)ra * Wl + cb) * pp + u_c8 * 16;
        u_src[2] = (img + (size_t)rb * Wl + ca) * pp + u_c8 * 16; u_src[3] = (img + (size_t)rb * Wl + cb) * pp + u_c8 * 16;
#pragma unroll
        for (int d = 0; d < 2; ++d) {                             // the weights of up-sample_sh8_kernel for rows / columns 2k+1+d
            const int oy = 2 * u_k + 1 + d, ox = 2 * u_j + 1 + d;
            const float fy = fmaxf(0.5f * ((float)oy + 0.5f) - 0.5f, 0.0f), fx = fmaxf(0.5f * ((float)ox + 0.5f) - 0.5f, 0.0f);
            u_ly[d] = fy - (float)(int)fy; u_lx[d] = fx - (float)(int)fx;
        }
#pragma unroll
        for (int dy = 0; dy < 2; ++dy)
#pragma unroll
            for (int dx = 0; dx < 2; ++dx)
                u_in[dy][dx] = (unsigned)(2 * u_k + 1 + dy) < (unsigned)a.H && (unsigned)(2 * u_j + 1 + dx) < (unsigned)a.W;
    }
    auto fill_a = [&](int g) {
        if (t >= 51 * 4) return;
        const unsigned char* sp = (const unsigned char*)a.src1 + (size_t)g * 128;
        h8v ch[4], cl[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) { ch[q] = *reinterpret_cast<const h8v*>(sp + u_src[q]); cl[q] = *reinterpret_cast<const h8v*>(sp + u_src[q] + 64); }
        float v[4][8];
#pragma unroll
        for (int q = 0; q < 4; ++q)
#pragma unroll
            for (int e = 0; e < 8; ++e) v[q][e] = fmaf((float)cl[q][e], 4.8828125e-4f, (float)ch[q][e]);
#pragma unroll
        for (int dy = 0; dy < 2; ++dy)
#pragma unroll
            for (int dx = 0; dx < 2; ++dx) {
                const float ly = u_ly[dy], lx = u_lx[dx], hy = 1.0f - ly, hx = 1.0f - lx;
                h8v oh, ol;
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    const float o = hy * (hx * v[0][e] + lx * v[1][e]) + ly * (hx * v[2][e] + lx * v[3][e]);
                    const _Float16 hh = (fabsf(o) < 6.103515625e-05f) ? (_Float16)0.0f : (_Float16)o;
                    oh[e] = u_in[dy][dx] ? hh : (_Float16)0.0f;
                    ol[e] = u_in[dy][dx] ? (_Float16)((o - (float)hh) * 2048.0f) : (_Float16)0.0f;
                }
                const int p = (2 * u_ci + dy) * HPW + 2 * u_cj + dx, d = p >> 1, pc = (p & 1) * 8 + u_c8;
                *reinterpret_cast<h8v*>(lds + d * 256 + ((pc ^ (d & 15)) * 16)) = oh;
                *reinterpret_cast<h8v*>(lds + d * 256 + (((pc + 4) ^ (d & 15)) * 16)) = ol;
            }
    };
    auto issue_b = [&](int g, int ky, int buf) {
        unsigned char* sb = lds + HA_BYTES + buf * B_BYTES + wave * 1024;
        const int soff = (ky * 3 * G + g) * 128;
#pragma unroll
        for (int i = 0; i < BPASS; ++i)
            if (wave + NW * i < BROWS / 8 && !OMNI_ABL(128)) dma16(rsw, sb + i * (1024 * NW), wbase[i], soff);
    };

    // pixel fragment offsets of the nine taps: halo pixel p = (wave+ky)*34 + (lane&31) + kx, row pair d = p >> 1,
    // first piece (hi, k chunk 0) at d*256 + 16*((8*(p&1) + (lane>>5)) ^ (d&15)); the other three pieces are that offset
    // XOR 32 / 64 / 96 (k chunk 1, lo chunk 0, lo chunk 1)
    int ao[9];
#pragma unroll
    for (int ky = 0; ky < 3; ++ky)
#pragma unroll
        for (int kx = 0; kx < 3; ++kx) {
            int p = (wave + ky) * HPW + (lane & 31) + kx;
            if constexpr (IW > 0) {
                const int tp = 32 * wave + (lane & 31), sb = tp / SUBPX, w_ = tp - sb * SUBPX, y = w_ / IW, x = w_ - y * IW;
                p = sb * HPS + (y + ky) * (IW + 2) + x + kx;
            }
            const int d = p >> 1;
            ao[ky * 3 + kx] = d * 256 + ((((p & 1) * 8 + (lane >> 5)) ^ (d & 15)) * 16);
        }
    int fo[4];                                                    // weight fragment offsets (32 consecutive rows)
    {
        const int r = lane & 31, v = r >> 1, h = lane >> 5;
#pragma unroll
        for (int k = 0; k < 4; ++k) fo[k] = v * 256 + ((((r & 1) * 8 + 2 * k + h) ^ v) * 16);
    }

    f16v acc[TN], acc1[TN];
#pragma unroll
    for (int j = 0; j < TN; ++j) { acc[j] = (f16v)(0.0f); acc1[j] = (f16v)(0.0f); }

    hstamp(63);
    if constexpr (UP2) { issue_b(0, 0, 0); fill_a(0); }            // (the weights travel while the halo is computed)
    else               { issue_a(0); issue_b(0, 0, 0); }
    int buf = 0;
    hstamp(1);
    for (int g = 0; g < G; ++g) {
#pragma unroll
        for (int ky = 0; ky < 3; ++ky) {
            hstamp(4 + 4 * (3 * g + ky));
            wait_vm<0>();                                         // halo (ky == 0) and this kernel row's weights have landed
            wait_lds_reads();                                     // ... and my reads of the other weight buffer have returned
            if (!OMNI_ABL(256)) __builtin_amdgcn_s_barrier();
            asm volatile("" ::: "memory");
            hstamp(5 + 4 * (3 * g + ky));
            if (ky < 2) issue_b(g, ky + 1, buf ^ 1);              // next weights under this row's matrix work
            else if (g + 1 < G) issue_b(g + 1, 0, buf ^ 1);
            hstamp(6 + 4 * (3 * g + ky));
            const unsigned char* sB = lds + HA_BYTES + buf * B_BYTES;
#pragma unroll
            for (int kx = 0; kx < 3; ++kx) {
                const int a0 = ao[ky * 3 + kx];
#pragma unroll
                for (int kc = 0; kc < 2; ++kc) {
                    const h8v ah = OMNI_ABL(512) ? (h8v)((_Float16)1.0f) : *reinterpret_cast<const h8v*>(lds + (a0 ^ (kc * 32)));
                    const h8v al = OMNI_ABL(512) ? (h8v)((_Float16)1.0f) : *reinterpret_cast<const h8v*>(lds + (a0 ^ (64 + kc * 32)));
#pragma unroll
                    for (int j = 0; j < TN; ++j) {
                        const unsigned char* bp = sB + (kx * BN + j * 32) * 128;
                        const h8v bh = OMNI_ABL(512) ? (h8v)((_Float16)1.0f) : *reinterpret_cast<const h8v*>(bp + fo[kc]);
                        const h8v bl = OMNI_ABL(512) ? (h8v)((_Float16)1.0f) : *reinterpret_cast<const h8v*>(bp + fo[2 + kc]);
                        if (!OMNI_ABL(64)) acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(bh, ah, acc[j], 0, 0, 0);
                        if (!OMNI_ABL(16) && !OMNI_DBG(a, 16)) acc1[j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(bl, ah, acc1[j], 0, 0, 0);
                        if (!OMNI_ABL(32) && !OMNI_DBG(a, 32)) acc1[j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(bh, al, acc1[j], 0, 0, 0);
                    }
                }
            }
            buf ^= 1;
            hstamp(7 + 4 * (3 * g + ky));
        }
        if (g + 1 < G) {                                          // everybody is done with this group's halo: fetch the next
            wait_lds_reads();
            __builtin_amdgcn_s_barrier();
            asm volatile("" ::: "memory");
            if constexpr (UP2) fill_a(g + 1); else issue_a(g + 1);
        }
    }

    // ---- epilogue (as conv_sh_kernel): column lane & 31 = pixel x0 + (lane & 31) of image row y0 + wave
    const int r = IW > 0 ? pix0 + 32 * wave + (lane & 31) : (m * a.H + y0 + wave) * a.W + x0 + (lane & 31);
    int c0[TN];
#pragma unroll
    for (int j = 0; j < TN; ++j) c0[j] = col0 + j * 32;
    if (a.dst_sh && !a.res_f32 && a.epi_lds) {
        static_assert(TH * 32 * (32 * TN + 4) * 4 <= (int)sizeof(lds), "the transposition tiles must fit the K loop's buffers");
        wait_lds_reads();
        __syncthreads();                                          // every wave is done with the halo and the weights
        hstamp(2);
        epilogue_tile_lds<TN>(acc, acc1, a, (size_t)(r - (lane & 31)), 32, c0, lane, reinterpret_cast<float*>(lds) + wave * (32 * (32 * TN + 4)));
        hstamp(62);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        hstamp(3); hdump();
        return;
    }
    hstamp(2);
    epilogue_row<TN>(acc, acc1, a, (size_t)r, c0, lane, a.dst_sh != 0);
    hstamp(3); hdump();
}

// ------------------------------------------------------------------ de_conv4_0: conv3x3(up2(x)), 32 -> 32 channels, PERSISTENT
// conv3x3_halo_sh_kernel<32, 4, UP2> spends 12.8 us per block at this shape (144 patches, 128 x 128 outputs: 18 432 blocks) around 0.72 us
// of matrix work: a chain of dependent round trips — the source pixels of the halo, three kernel-row weight stages each issued one 0.24-us
// matrix phase ahead of its use, the bias, the stores — that twelve waves per CU do not hide (19 % MFMA-busy).  With ONE input group all
// nine taps of the weights are 36 KiB: here a block keeps them in LDS for its whole life and walks over tiles, and its eight waves split
// the work by what they WAIT for (loads and stores retire through one in-order counter: a wave that does both waits for its previous
// tile's store acknowledges whenever it waits for pixels — measured: 277 us, as slow as the kernel this replaces):
//   waves 4-7, producers: source pixels global -> registers -> up-sampling arithmetic -> the halo of tile k+1 in LDS (two halo buffers);
//   waves 0-3, consumers: 54 matrix instructions per wave on the halo of tile k, bias from registers, stores — never a wait on memory.
// One block barrier per tile hands the buffers over.  Same cells, same K order (ky, kx, k chunk): same bits as the kernel it replaces.
//
// HEADS (round 5): the `pred` / `weight_pred` heads (3x3, 32 -> 1 each, model/spherical_model.py:223-224,304-306) start HERE instead of in a kernel
// that re-reads this one's output: de_conv4_0's result is the widest tensor of the network (302 MB at 8 panoramas, written once and read once by
// heads_kernel: 290 us for the pair) and never exists in this form.  out[q] = sum_{dy,dx} w[dy][dx] . x[q + (dy,dx)] is turned around: pixel p, where
// x[p] lives in registers, contributes w[dy][dx] . x[p] to q = p - (dy,dx) — eighteen 32-channel dot products per pixel (9 taps x 2 heads), which are
// ONE more matrix product: rows = (dy, head, dx), k = the 32 channels in the order the accumulator quads already hold them (a lane's 16 channels
// are its two k chunks: no data movement), three f16x3 terms like every other product = 6 matrix instructions per wave and tile beside the 54 of
// the convolution.  The three dx terms of a row are summed across neighbouring lanes (fixed order dx = -1, 0, +1), which leaves per tile row and
// (dy, head) 34 partial sums — pixels -1 .. 32: the two outer ones belong to the neighbouring tiles' pixels — written to `hr`
// [tile][row 4][(dy, head) 6][32 sums | pixel -1 | pixel 32 | 2 pad]: 3.4 KB per tile instead of 16 KB.  heads_finish_kernel adds the three rows (dy) and the neighbour tiles'
// outer sums in a fixed order, then bias, ReLU / sigmoid and the product.  Deterministic; equal to heads_kernel up to fp32 summation order.
constexpr int HR_PITCH = 36;                                 // floats per (tile row, (dy, head)) record of `hr`: 32 sums, pixel -1, pixel 32, 2 of padding (16-byte rows)
struct HeadsArgs { const void* w16; float* hr; };           // w16: the heads' weights in fragment order (Engine: heads.w16f), [hi kc0, hi kc1, lo kc0, lo kc1][64 lanes] x 16 B

template <bool HEADS>
__global__ __launch_bounds__(HEADS ? 64 * (4 + OMNI_G1_PW) : 512) void conv3x3_up2_g1_kernel(ShConvArgs a, int ntiles, HeadsArgs hd)
{
    constexpr int BN = 32, TH = 4, NW = 4, RPP = 8 * NW;
    constexpr int PW = HEADS ? OMNI_G1_PW : 4, CPT = 32 / PW;    // producer waves, channels per producer thread
    constexpr int HPX = (TH + 2) * HPW, HA_INSTR = (HPX * 8 + 63) / 64, HA_BYTES = HA_INSTR * 1024;
    constexpr int BROWS = 3 * BN, BPASS = (BROWS + RPP - 1) / RPP, B_BYTES = BROWS * 128, W_OFF = 2 * HA_BYTES;
    __shared__ __attribute__((aligned(1024))) unsigned char lds[2 * HA_BYTES + 3 * B_BYTES];
    // (ablation 32768, tools/g1_stamps.py: s_memtime of consumer wave 0 / producer wave 4 of block 0 at three points of each of its first 40 tiles, written over hd.hr at the end)
    constexpr int ST_N = 40, ST_K = 5;
    __shared__ long long stamps[OMNI_ABL(32768) ? 2 * ST_N * ST_K : 1];
    auto stamp = [&](int who, int it, int k) { if (OMNI_ABL(32768) && blockIdx.x == 0 && it < ST_N && (threadIdx.x & 63) == 0) stamps[(who * ST_N + it) * ST_K + k] = clock64(); };

    const int t = threadIdx.x, lane = t & 63;
    const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
    const bool consumer = wave < NW;
    const int tw = a.W / HT_W, th = a.H / TH, per_img = tw * th;
    // tiles of this block: XCD x (blocks x mod 8) owns one contiguous range of tiles, its blocks walk it with stride gridDim.x / 8
    const int xcd = blockIdx.x & 7, lb = blockIdx.x >> 3, nlb = gridDim.x >> 3;
    const int per = (ntiles + 7) >> 3, t_end = min(ntiles, (xcd + 1) * per);
    int tile = xcd * per + lb;
    if (tile >= t_end) return;
    // tile -> (patch, tile row, tile column): divided ONCE per wave, then stepped — a producer spent 2400 of its 6400 cycles per tile in front of its loads, most of them
    // in the three integer divisions per tile index (two indices per tile: profiles/r05h_up2_producer.txt, 8.)
    struct TileXY { int m, ty, tx; };
    auto coords = [&](int tl) { TileXY c; c.m = tl / per_img; const int r = tl - c.m * per_img; c.ty = r / tw; c.tx = r - c.ty * tw; return c; };
    const TileXY tstep = coords(nlb);
    auto advance = [&](TileXY& c) { c.tx += tstep.tx; c.ty += tstep.ty; c.m += tstep.m; if (c.tx >= tw) { c.tx -= tw; ++c.ty; } if (c.ty >= th) { c.ty -= th; ++c.m; } };

    if (consumer) {
        // ---- the nine taps' weights, once (three kernel-row stages of the halo kernel's layout, side by side)
        const int gs = (lane & 15) ^ ((4 * wave + (lane >> 4)) & 15);
        const int rl = 8 * wave + 2 * (lane >> 4) + (gs >> 3), pc16 = (gs & 7) * 16;
        const rsrc_t rsw = make_rsrc(a.wt, (size_t)a.Cout * 9 * 128);
#pragma unroll
        for (int ky = 0; ky < 3; ++ky)
#pragma unroll
            for (int i = 0; i < BPASS; ++i) {
                const int r = rl + RPP * i, kx = r / BN, co = r - kx * BN;
                if (wave + NW * i < BROWS / 8) dma16(rsw, lds + W_OFF + ky * B_BYTES + wave * 1024 + i * (1024 * NW), (co * 9 + kx) * 128 + pc16, ky * 3 * 128);
            }
        f4v bq[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) bq[q] = a.bias ? *reinterpret_cast<const f4v*>(a.bias + 8 * q + 4 * (lane >> 5)) : (f4v)(0.0f);
        h8v hwf[4];                                              // HEADS: this lane's weight fragments (row lane & 31, k chunk lane >> 5)
        if constexpr (HEADS) {
#pragma unroll
            for (int k = 0; k < 4; ++k) hwf[k] = *reinterpret_cast<const h8v*>((const unsigned char*)hd.w16 + k * 1024 + lane * 16);
        }
        int ao[9], fo[4];                                        // fragment offsets, as in conv3x3_halo_sh_kernel
#pragma unroll
        for (int ky = 0; ky < 3; ++ky)
#pragma unroll
            for (int kx = 0; kx < 3; ++kx) {
                const int p = (wave + ky) * HPW + (lane & 31) + kx, d = p >> 1;
                ao[ky * 3 + kx] = d * 256 + ((((p & 1) * 8 + (lane >> 5)) ^ (d & 15)) * 16);
            }
        {
            const int r = lane & 31, v = r >> 1, h = lane >> 5;
#pragma unroll
            for (int k = 0; k < 4; ++k) fo[k] = v * 256 + ((((r & 1) * 8 + 2 * k + h) ^ v) * 16);
        }
        wait_vm<0>();                                             // weights (and bias) have landed
        __syncthreads();                                          // ... everybody's; the first halo is there
        TileXY ct = coords(tile);                                  // (the epilogue's tile)
        for (int it = 0;; ++it, advance(ct)) {
            const unsigned char* ha = lds + (it & 1) * HA_BYTES;
            if (wave == 0) stamp(0, it, 0);
            // the eight fragments of tap k+1 are read while the six matrix instructions of tap k run
            f16v acc = (f16v)(0.0f), acc1 = (f16v)(0.0f), accx = (f16v)(0.0f), accy = (f16v)(0.0f);
            h8v fa[2][4], fb[2][4];                               // [buffer][hi k0, hi k1, lo k0, lo k1] of the pixels / of the weights
            auto read_tap = [&](int tap, int bf) {
                const int a0 = ao[tap];
                const unsigned char* bp = lds + W_OFF + (tap / 3) * B_BYTES + ((tap % 3) * BN) * 128;
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    if (OMNI_ABL(512)) { fa[bf][k] = (h8v)((_Float16)(float)(a0 & 3)); fb[bf][k] = (h8v)((_Float16)(float)(fo[k] & 3)); continue; }   // (ablation: no fragment reads)
                    fa[bf][k] = *reinterpret_cast<const h8v*>(ha + (a0 ^ (k * 32)));
                    fb[bf][k] = *reinterpret_cast<const h8v*>(bp + fo[k]);
                }
            };
            read_tap(0, 0);
#pragma unroll
            for (int tap = 0; tap < 9; ++tap) {
                const int bf = tap & 1;
                __builtin_amdgcn_sched_barrier(0);
                if (tap + 1 < 9) read_tap(tap + 1, bf ^ 1);
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int kc = 0; kc < 2; ++kc) {
                    if (OMNI_ABL(8192)) {                         // (ablation: four accumulators instead of two — another summation order)
                        if (kc == 0) { acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(fb[bf][kc], fa[bf][kc], acc, 0, 0, 0);
                                       acc1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(fb[bf][2 + kc], fa[bf][kc], acc1, 0, 0, 0);
                                       accx = __builtin_amdgcn_mfma_f32_32x32x16_f16(fb[bf][kc], fa[bf][2 + kc], accx, 0, 0, 0); }
                        else         { accy = __builtin_amdgcn_mfma_f32_32x32x16_f16(fb[bf][kc], fa[bf][kc], accy, 0, 0, 0);
                                       accx = __builtin_amdgcn_mfma_f32_32x32x16_f16(fb[bf][2 + kc], fa[bf][kc], accx, 0, 0, 0);
                                       acc1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(fb[bf][kc], fa[bf][2 + kc], acc1, 0, 0, 0); }
                        continue;
                    }
                    if (!OMNI_ABL(64)) acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(fb[bf][kc], fa[bf][kc], acc, 0, 0, 0);
                    else acc[0] += (float)fb[bf][kc][0] * (float)fa[bf][kc][0] + (float)fb[bf][2 + kc][0] * (float)fa[bf][2 + kc][0];
                    if (!OMNI_ABL(16)) acc1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(fb[bf][2 + kc], fa[bf][kc], acc1, 0, 0, 0);
                    if (!OMNI_ABL(32)) acc1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(fb[bf][kc], fa[bf][2 + kc], acc1, 0, 0, 0);
                }
            }
            __builtin_amdgcn_sched_barrier(0);
            if (OMNI_ABL(8192)) { acc += accy; acc1 += accx; }
            if (wave == 0) stamp(0, it, 1);
            if constexpr (HEADS) if (OMNI_ABL(16384)) { if (acc[0] == 12345.678f && acc1[3] == 3.0f) hd.hr[lane] = acc[1]; } else {
                // the tile's result stays in registers: v[q] = channels 8q + 4h .. + 3 of pixel lane & 31 — the lane's k chunk kc is its quads 2kc, 2kc + 1
                h8v ph[2], pl[2];
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    f4v v;
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[e] = fmaf(acc1[4 * q + e], 4.8828125e-4f, acc[4 * q + e]);
                    v += bq[q];
                    if (a.act == OMNI_ACT_RELU) { v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f); }
                    h4v hi, lo; sh_split4(v, hi, lo);            // (the split every SH epilogue does: range guard included)
#pragma unroll
                    for (int e = 0; e < 4; ++e) { ph[q >> 1][4 * (q & 1) + e] = hi[e]; pl[q >> 1][4 * (q & 1) + e] = lo[e]; }
                }
                if (OMNI_ABL(32768)) { if (ph[0][0] == (_Float16)77.0f && pl[1][3] == (_Float16)3.0f) hd.hr[1] = 1.0f; if (wave == 0) stamp(0, it, 2); }
                f16v d0 = (f16v)(0.0f), d1 = (f16v)(0.0f);
#pragma unroll
                for (int kc = 0; kc < 2; ++kc) {
                    d0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(hwf[kc], ph[kc], d0, 0, 0, 0);
                    d1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(hwf[2 + kc], ph[kc], d1, 0, 0, 0);
                    d1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(hwf[kc], pl[kc], d1, 0, 0, 0);
                }
                // rows (reg & 3) + 8 (reg >> 2) + 4 h: register group g = 0..2 is (dy, head) pair g + 3h, its registers 0..2 are dx = -1, 0, +1
                if (OMNI_ABL(32768)) { if (d0[0] == 12345.678f && d1[5] == 3.0f) hd.hr[2] = 1.0f; if (wave == 0) stamp(0, it, 3); }
                const int px = lane & 31, h = lane >> 5;
                float* hp = hd.hr + ((size_t)tile * TH + wave) * (6 * HR_PITCH);
#pragma unroll
                for (int g = 0; g < 3; ++g) {
                    const float tl = fmaf(d1[4 * g], 4.8828125e-4f, d0[4 * g]), tc = fmaf(d1[4 * g + 1], 4.8828125e-4f, d0[4 * g + 1]),
                                tr = fmaf(d1[4 * g + 2], 4.8828125e-4f, d0[4 * g + 2]);
                    // out[q] takes w[dx] . x[q + dx]: its dx = -1 term comes from pixel q - 1, its dx = +1 term from pixel q + 1
                    const float fl = __shfl_up(tl, 1, 32), fr = __shfl_down(tr, 1, 32);
                    const float sum = ((px > 0 ? fl : 0.0f) + tc) + (px < 31 ? fr : 0.0f);
                    float* row = hp + (g + 3 * h) * HR_PITCH;
                    row[px] = sum;
                    if (px == 0) row[32] = tr;                     // pixel -1 of this row (the left neighbour tile's column 31) takes my dx = +1 term
                    if (px == 31) row[33] = tl;                    // pixel 32 takes my dx = -1 term
                }
            } else
            {   // epilogue of this tile: column lane & 31 = pixel x0 + (lane & 31) of image row y0 + wave (through an LDS transposition, split-half or
                // fp32: 247 | 248 us — the stores are not this kernel's limit, and 18 KB of LDS more per block are felt beside other kernels)
                const int m = ct.m, y0 = ct.ty * TH, x0 = ct.tx * HT_W;
                const size_t r = (size_t)(m * a.H + y0 + wave) * a.W + x0 + (lane & 31);
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    f4v v;
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[e] = fmaf(acc1[4 * q + e], 4.8828125e-4f, acc[4 * q + e]);
                    v += bq[q];
                    if (a.act == OMNI_ACT_RELU) { v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f); }
                    else if (a.act == OMNI_ACT_GELU) {
#pragma unroll
                        for (int e = 0; e < 4; ++e) v[e] = 0.5f * v[e] * (1.0f + erff(v[e] * 0.70710678118654752440f));
                    }
                    const size_t o = r * BN + 8 * q + 4 * (lane >> 5);
                    if (OMNI_ABL(2048)) { if (v.x == 12345.678f) act_store4<false>(a.dst, o, v); }
                    else if (a.dst_sh) act_store4<true>(a.dst, o, v);
                    else               act_store4<false>(a.dst, o, v);
                }
            }
            if (wave == 0) stamp(0, it, 4);
            tile += nlb;
            if (tile >= t_end) break;
            wait_lds_reads();
            __syncthreads();                                      // this halo buffer is free, the other one is complete
        }
        if (OMNI_ABL(32768) && blockIdx.x == 0 && wave == 0 && lane == 0) for (int i = 0; i < ST_N * ST_K; ++i) reinterpret_cast<long long*>(hd.hr)[i] = stamps[i];
        return;
    }

    // ---- producers: thread (cell, CPT channels) of the 3 x 17 cells of 2 x 2 pixels of a tile's halo (see conv3x3_halo_sh_kernel, UP2); PW = 4 producer waves: 8 channels
    // per thread (the product); PW = 8: 4 channels per thread, two producer waves per SIMD (-DOMNI_G1_PW=8: measured equal, profiles/r05h_up2_producer.txt)
    using hcv = std::conditional_t<CPT == 8, h8v, h4v>;
    const int ft = t - 64 * NW;
    const int Hl = a.H >> 1, Wl = a.W >> 1;
    constexpr int TPC = 32 / CPT;                                 // threads per cell
    const int u_cg = ft % TPC, u_cell = ft / TPC, u_ci = u_cell / 17, u_cj = u_cell - u_ci * 17;
    const int u_c8 = u_cg * CPT / 8, u_sub = (u_cg * CPT % 8) * 2; // 16-byte piece (8 channels) and byte offset inside it
    const bool filler = ft < 51 * TPC;
    auto load_src = [&](const TileXY& c, hcv (&ch)[4], hcv (&cl)[4]) {
        if (!filler) return;
        const int m = c.m, y0 = c.ty * TH, x0 = c.tx * HT_W;
        const int u_k = (y0 >> 1) - 1 + u_ci, u_j = (x0 >> 1) - 1 + u_cj;
        const int ra = min(max(u_k, 0), Hl - 1), rb = min(max(u_k + 1, 0), Hl - 1), ca = min(max(u_j, 0), Wl - 1), cb = min(max(u_j + 1, 0), Wl - 1);
        const size_t img = (size_t)m * Hl * Wl;
        const unsigned char* sp = (const unsigned char*)a.src1 + u_cg * (CPT * 2);
        const size_t so[4] = {(img + (size_t)ra * Wl + ca) * 128, (img + (size_t)ra * Wl + cb) * 128, (img + (size_t)rb * Wl + ca) * 128, (img + (size_t)rb * Wl + cb) * 128};
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            if (OMNI_ABL(4096)) { ch[q] = (hcv)((_Float16)1.0f); cl[q] = ch[q]; }
            else { ch[q] = *reinterpret_cast<const hcv*>(sp + so[q]); cl[q] = *reinterpret_cast<const hcv*>(sp + so[q] + 64); }
        }
    };
    auto write_halo = [&](const TileXY& c, unsigned char* hb, const hcv (&ch)[4], const hcv (&cl)[4], int it) {
        if (!filler || OMNI_ABL(1024)) return;
        const int y0 = c.ty * TH, x0 = c.tx * HT_W;
        const int u_k = (y0 >> 1) - 1 + u_ci, u_j = (x0 >> 1) - 1 + u_cj;
        float v[4][CPT];
#pragma unroll
        for (int q = 0; q < 4; ++q)
#pragma unroll
            for (int e = 0; e < CPT; ++e) v[q][e] = fmaf((float)cl[q][e], 4.8828125e-4f, (float)ch[q][e]);
        if (OMNI_ABL(32768)) { float z = 0.0f; for (int q = 0; q < 4; ++q) for (int e = 0; e < CPT; ++e) z += v[q][e]; if (z == 12345.678f) hd.hr[0] = z; if (wave == NW) stamp(1, it, 1); }
#pragma unroll
        for (int dy = 0; dy < 2; ++dy) {
#pragma unroll
            for (int dx = 0; dx < 2; ++dx) {
                const int oy = 2 * u_k + 1 + dy, ox = 2 * u_j + 1 + dx;
                const float fy = fmaxf(0.5f * ((float)oy + 0.5f) - 0.5f, 0.0f), fx = fmaxf(0.5f * ((float)ox + 0.5f) - 0.5f, 0.0f);
                const float ly = fy - (float)(int)fy, lx = fx - (float)(int)fx, hy = 1.0f - ly, hx = 1.0f - lx;
                const bool in = (unsigned)oy < (unsigned)a.H && (unsigned)ox < (unsigned)a.W;
                hcv oh, ol;
#pragma unroll
                for (int e = 0; e < CPT; ++e) {
                    const float o = hy * (hx * v[0][e] + lx * v[1][e]) + ly * (hx * v[2][e] + lx * v[3][e]);
                    const _Float16 hh = (fabsf(o) < 6.103515625e-05f) ? (_Float16)0.0f : (_Float16)o;
                    oh[e] = in ? hh : (_Float16)0.0f;
                    ol[e] = in ? (_Float16)((o - (float)hh) * 2048.0f) : (_Float16)0.0f;
                }
                const int p = (2 * u_ci + dy) * HPW + 2 * u_cj + dx, d = p >> 1, pc = (p & 1) * 8 + u_c8;
                *reinterpret_cast<hcv*>(hb + d * 256 + ((pc ^ (d & 15)) * 16) + u_sub) = oh;
                *reinterpret_cast<hcv*>(hb + d * 256 + (((pc + 4) ^ (d & 15)) * 16) + u_sub) = ol;
            }
            if (dy == 0 && wave == NW) stamp(1, it, 2);
        }
    };
    // the pixels of tile k+2 are on their way while the halo of tile k+1 is computed (a producer issues no stores: its waits are for loads only)
    // Two register sets in turn, no copies: a set is re-loaded (tile k+2) as soon as its halo (tile k) is written — the loads are issued at the END of a tile's work,
    // the arithmetic starts right behind the barrier.
    hcv rh[2][4], rl_[2][4];
    TileXY cw = coords(tile), cn = cw;                            // the tile whose halo is written next / the tile loaded last
    load_src(cw, rh[0], rl_[0]);
    advance(cn);
    if (tile + nlb < t_end) load_src(cn, rh[1], rl_[1]);
    write_halo(cw, lds, rh[0], rl_[0], ST_N);
    cw = cn; advance(cn);
    if (tile + 2 * nlb < t_end) load_src(cn, rh[0], rl_[0]);
    __syncthreads();                                              // (the consumers' first barrier)
    int it = 0;
#define OMNI_G1_STEP(SET) { \
        const int next = tile + nlb; \
        if (wave == NW) stamp(1, it, 0); \
        if (next >= t_end) break;                                 /* (the consumers leave at the same point: no barrier after the last tile) */ \
        write_halo(cw, lds + ((it + 1) & 1) * HA_BYTES, rh[SET], rl_[SET], it); \
        if (wave == NW) stamp(1, it, 3); \
        cw = cn; advance(cn); \
        if (next + 2 * nlb < t_end) load_src(cn, rh[SET], rl_[SET]); \
        if (wave == NW) stamp(1, it, 4); \
        __syncthreads(); \
        tile = next; ++it; }
    for (;;) {
        OMNI_G1_STEP(1)
        OMNI_G1_STEP(0)
    }
#undef OMNI_G1_STEP
    if (OMNI_ABL(32768) && blockIdx.x == 0 && wave == NW && lane == 0) for (int i = 0; i < ST_N * ST_K; ++i) reinterpret_cast<long long*>(hd.hr)[ST_N * ST_K + i] = stamps[ST_N * ST_K + i];
}

// ------------------------------------------------------------------ stem: conv 7x7 s2 p3, 3 -> 64, + folded BN + ReLU (f16x3)
// model/spherical_model.py:254 (conv1, bn1, relu) as an implicit GEMM on the fp16 matrix cores.  K is laid out as
// (c, ky, kx padded 7 -> 8): one 8-wide MFMA fragment is then 8 CONSECUTIVE input pixels of one (channel, kernel row) — four
// 4-byte reads from the input patch parked in LDS as a hi and a lo half image (split once per pixel at load time); K = 3*7*8 = 168, padded to 192 = 6
// groups of 32 with zero weights.  A block owns an 8-row strip of one patch's output (Po columns in tiles of 16): the 64 x 192
// pre-split filter bank (48 KiB) is DMA'd into LDS once per block, wave w owns output rows 2w, 2w+1 of the strip.
// Output: SH [M, Po, Po, 64].
constexpr int SM_TH = 8, SM_TW = 16, SM_IH = 2 * SM_TH + 5, SM_IW = 2 * SM_TW + 5, SM_IP = 40, SM_G = 6;

__global__ __launch_bounds__(256) void stem_f16x3_kernel(const float* __restrict__ src, const void* __restrict__ wt16,
                                                         const float* __restrict__ bias, void* __restrict__ dst, int M, int P, int Po, int epi_lds)
{
    __shared__ __attribute__((aligned(1024))) unsigned char wl[64 * SM_G * 128];
    __shared__ __attribute__((aligned(16))) _Float16 imh[3 * SM_IH * SM_IP], iml[3 * SM_IH * SM_IP];   // the input patch, split ONCE per pixel
    __shared__ __attribute__((aligned(16))) float etile[4][32 * 36];                                     // a transposition tile per wave (epilogue_tile_lds, one 32-channel group at a time: two blocks per CU stay)
    const int t = threadIdx.x, lane = t & 63;
    const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
    const int strips = Po / SM_TH;
    const int m = blockIdx.x / strips, oy0 = (blockIdx.x % strips) * SM_TH;

    // filter bank -> LDS: SM_G regions of 64 rows x 128 B, same pair swizzle as the convolution tiles
    {
        const int gs = (lane & 15) ^ ((4 * wave + (lane >> 4)) & 15);
        const int rl = 8 * wave + 2 * (lane >> 4) + (gs >> 3), pc16 = (gs & 7) * 16;
        const rsrc_t rsw = make_rsrc(wt16, (size_t)64 * SM_G * 128);
#pragma unroll
        for (int g = 0; g < SM_G; ++g)
#pragma unroll
            for (int i = 0; i < 2; ++i)
                dma16(rsw, wl + g * 8192 + wave * 1024 + i * 4096, ((rl + 32 * i) * SM_G + g) * 128 + pc16, 0);
    }
    int fo[4];
    {
        const int r = lane & 31, v = r >> 1, h = lane >> 5;
#pragma unroll
        for (int k = 0; k < 4; ++k) fo[k] = v * 256 + ((((r & 1) * 8 + 2 * k + h) ^ v) * 16);
    }
    // this lane's output pixel inside a tile and its 12 fragment rows: fragment (g, kc) is input row (c, ky) = divmod(4g+2kc+h, 7)
    const int py = 2 * wave + ((lane & 31) >> 4), px = lane & 15;
    int rowoff[2 * SM_G];
#pragma unroll
    for (int f = 0; f < 2 * SM_G; ++f) {
        int rr = 2 * f + (lane >> 5);
        rr = rr < 21 ? rr : 20;                                   // rows 21..23 carry zero weights: any finite data will do
        rowoff[f] = ((rr / 7) * SM_IH + rr % 7 + 2 * py) * SM_IP + 2 * px;
    }
    ShConvArgs e;
    e.bias = bias; e.res = nullptr; e.res_f32 = 0; e.act = OMNI_ACT_RELU; e.Cout = 64; e.dst = dst; e.post = nullptr; e.post_rows = 1; e.epi_lds = epi_lds;

    // gridDim.y column ranges per strip (a lone panorama's 18 patches are 144 strips: a quarter strip per block fills the chip)
    const int ox_first = blockIdx.y * (Po / gridDim.y), ox_last = ox_first + Po / gridDim.y;
    // the next tile's input pixels travel (global -> registers) under the current tile's matrix work
    constexpr int IMG = 3 * SM_IH * SM_IP, IPT = (IMG + 255) / 256;
    float pre[IPT];
    auto prefetch = [&](int ox0) {
        const int iy0 = oy0 * 2 - 3, ix0 = ox0 * 2 - 3;
#pragma unroll
        for (int k = 0; k < IPT; ++k) {                           // (the pad columns 37..39 are read by the zero-weight kx = 7 lane slots)
            const int i = t + 256 * k;
            const int c = i / (SM_IH * SM_IP), r = (i % (SM_IH * SM_IP)) / SM_IP, q = i % SM_IP;
            const int iy = iy0 + r, ix = ix0 + q;
            pre[k] = (i < IMG && q < SM_IW && (unsigned)iy < (unsigned)P && (unsigned)ix < (unsigned)P) ? src[((size_t)m * 3 + c) * P * P + (size_t)iy * P + ix] : 0.0f;
        }
    };
    prefetch(ox_first);
    for (int ox0 = ox_first; ox0 < ox_last; ox0 += SM_TW) {
        __syncthreads();                                          // the previous tile's fragment reads are done
#pragma unroll
        for (int k = 0; k < IPT; ++k) {
            const int i = t + 256 * k;
            const float x = pre[k];
            const _Float16 hh = (fabsf(x) < 6.103515625e-05f) ? (_Float16)0.0f : (_Float16)x;
            if (i < IMG) { imh[i] = hh; iml[i] = (_Float16)((x - (float)hh) * 2048.0f); }
        }
        if (ox0 == ox_first) wait_vm<0>();                        // the filter bank has landed
        __syncthreads();
        if (ox0 + SM_TW < ox_last) prefetch(ox0 + SM_TW);
        f16v acc[2], acc1[2];
#pragma unroll
        for (int j = 0; j < 2; ++j) { acc[j] = (f16v)(0.0f); acc1[j] = (f16v)(0.0f); }
#pragma unroll
        for (int g = 0; g < SM_G; ++g)
#pragma unroll
            for (int kc = 0; kc < 2; ++kc) {
                // 8 consecutive pixels from an even column: four 4-byte reads per half image (every input pixel serves ~28 fragments and
                // is split once, at load time)
                const int ro = rowoff[2 * g + kc];
                h8v ah, al;
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const h2v xh = *reinterpret_cast<const h2v*>(imh + ro + 2 * u), xl = *reinterpret_cast<const h2v*>(iml + ro + 2 * u);
                    ah[2 * u] = xh[0]; ah[2 * u + 1] = xh[1]; al[2 * u] = xl[0]; al[2 * u + 1] = xl[1];
                }
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    const unsigned char* bp = wl + g * 8192 + j * 4096;
                    const h8v bh = *reinterpret_cast<const h8v*>(bp + fo[kc]);
                    const h8v bl = *reinterpret_cast<const h8v*>(bp + fo[2 + kc]);
                    acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(bh, ah, acc[j], 0, 0, 0);
                    acc1[j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(bl, ah, acc1[j], 0, 0, 0);
                    acc1[j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(bh, al, acc1[j], 0, 0, 0);
                }
            }
        const size_t r = ((size_t)m * Po + oy0 + py) * Po + ox0 + px;
        const int c0[2] = {0, 32};
        if (e.epi_lds) {                                          // 151 MB of output at 8 panoramas: as 16-byte pieces (the wave's two rows of 16 pixels)
            const size_t ra = ((size_t)m * Po + oy0 + 2 * wave) * Po + ox0;
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const f16v ea[1] = {acc[j]}, eb[1] = {acc1[j]};
                const int cj[1] = {32 * j};
                epilogue_tile_lds<1>(ea, eb, e, ra, 32, cj, lane, etile[wave], ra + Po);
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            }
        } else epilogue_row<2>(acc, acc1, e, r, c0, lane, true);
    }
}

// ---- the stem with producer / consumer waves (option conv_stem_pc)
// Twelve waves: 0-7 consume (fragment reads, 36 matrix instructions per tile, stores), 8-11 produce (input pixels global -> registers -> hi / lo
// split -> the NEXT tile's image in LDS, two image buffers) — loads and stores retire through one in-order counter, so a wave that does both
// waits for its previous stores' acknowledges whenever it waits for pixels (as conv3x3_up2_g1_kernel found); one block barrier per tile.
// Consumer wave w owns the two tile rows 2 (w & 3) and the 32 output channels of half w >> 2 (round 5; rounds 3-4: four consumers with both halves).
// With ONE consumer per SIMD a tile cost its K loop (2.8 us: 144 LDS reads whose latency nothing hid) PLUS its epilogue (2.7 us) — 95 us for 144
// patches with the matrix instructions themselves worth 12 (profiles/r05g_stem_ablations.txt); two consumers per SIMD run one's epilogue under the
// other's K loop.  The A fragments are read twice (LDS traffic per tile 2.3 -> 3.1 k cycles); every output element is the same sum as before.
__global__ __launch_bounds__(768) void stem_f16x3_pc_kernel(const float* __restrict__ src, const void* __restrict__ wt16,
                                                         const float* __restrict__ bias, void* __restrict__ dst, int M, int P, int Po, int epi_lds, int tpb)
{
    constexpr int IMG = 3 * SM_IH * SM_IP, IPT = (IMG + 255) / 256;
    // rows are STORED 48 halfs apart (SM_IP = 40 are written and read): a wave's fragment read takes pixel rows y and y + 1 of two image rows each —
    // 2 x 40 halfs = 40 dwords apart they share 8 of 32 banks (every read two passes), 48 dwords apart none
    constexpr int SM_IS = 48, IMGS = 3 * SM_IH * SM_IS;
    __shared__ __attribute__((aligned(1024))) unsigned char wl[64 * SM_G * 128];
    __shared__ __attribute__((aligned(16))) _Float16 imh[2][IMGS], iml[2][IMGS];                          // the input patch of a tile, split ONCE per pixel; two tiles
    __shared__ __attribute__((aligned(16))) float etile[8][32 * 36];                                     // a transposition tile per consumer wave (epilogue_tile_lds: its 32-channel group)
    const int t = threadIdx.x, lane = t & 63;
    const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
    // A block walks tpb consecutive tiles of the flat sequence (patch, strip of SM_TH rows, SM_TW columns): a quarter strip where the launch is small
    // (a lone panorama's 18 patches are 576 tiles), 18 tiles = 4.5 strips at 8 panoramas — ONE block per CU for the whole launch: the filter bank is
    // loaded once per CU instead of 4.5 times and the producers' pipeline is filled once (round 5; before: a strip per block, 4 tiles, 19 us per block
    // of which 4 x ~2.5 were its tiles).
    const int sps = Po / SM_TH, tps = Po / SM_TW, ntiles = M * sps * tps;
    const int T0 = blockIdx.x * tpb, T1 = min(T0 + tpb, ntiles);
    if (T0 >= T1) return;
    auto where = [&](int T, int& m, int& oy0, int& ox0) { const int strip = T / tps; ox0 = (T - strip * tps) * SM_TW; m = strip / sps; oy0 = (strip - m * sps) * SM_TH; };

    if (wave >= 8) {
        // ---- producers
        const int ft = t - 512;
        float pre[IPT], nxt[IPT];
        // what does not depend on the tile, once per thread: the pixel's offset inside the patch's 3 x P x P image, its (row, column) inside the tile's
        // input window, its LDS slot (the index arithmetic — divisions by 840 and 40, 64-bit address products — was ~30 quarter-rate integer
        // multiplies per tile and wave: profiles/r05g_stem_ablations.txt)
        int po[IPT], rq[IPT], ls[IPT];
#pragma unroll
        for (int k = 0; k < IPT; ++k) {                           // (the pad columns 37..39 are read by the zero-weight kx = 7 lane slots)
            const int i = ft + 256 * k;
            const int c = i / (SM_IH * SM_IP), r = (i % (SM_IH * SM_IP)) / SM_IP, q = i % SM_IP;
            po[k] = (c * P + r) * P + q;
            rq[k] = (i < IMG && q < SM_IW) ? (r | (q << 8)) : -1;
            ls[k] = i < IMG ? (i / SM_IP) * SM_IS + q : -1;
        }
        // the tile whose pixels are fetched next: (patch, strip, column tile), divided once and stepped
        int fm, fs, fc;
        { const int strip = T0 / tps; fc = T0 - strip * tps; fm = strip / sps; fs = strip - fm * sps; }
        auto fetch = [&](float (&v)[IPT]) {                       // ... and steps to the following tile
            const int m = fm, oy0 = fs * SM_TH, ox0 = fc * SM_TW;
            if (++fc == tps) { fc = 0; if (++fs == sps) { fs = 0; ++fm; } }
            const int iy0 = oy0 * 2 - 3, ix0 = ox0 * 2 - 3;
            const float* base = src + (size_t)m * 3 * P * P + ((long long)iy0 * P + ix0);       // (wave-uniform; dereferenced only where the pixel exists)
#pragma unroll
            for (int k = 0; k < IPT; ++k) {
                const int iy = iy0 + (rq[k] & 0xff), ix = ix0 + (rq[k] >> 8);
                v[k] = (rq[k] >= 0 && (unsigned)iy < (unsigned)P && (unsigned)ix < (unsigned)P) ? base[po[k]] : 0.0f;
            }
        };
        auto park = [&](int b, const float (&v)[IPT]) {
#pragma unroll
            for (int k = 0; k < IPT; ++k) {
                const float x = v[k];
                const _Float16 hh = (fabsf(x) < 6.103515625e-05f) ? (_Float16)0.0f : (_Float16)x;
                if (ls[k] >= 0) { imh[b][ls[k]] = hh; iml[b][ls[k]] = (_Float16)((x - (float)hh) * 2048.0f); }
            }
        };
        // Two register sets in turn: a set is re-fetched (tile T + 2) as soon as it is parked — the loads are issued at the END of a tile's work, when the consumers are
        // in their epilogues, not behind the barrier where their fragment reads start (conv3x3_up2_g1_kernel's producers stood a whole K loop in front of their loads there:
        // profiles/r05h_up2_producer.txt, 8.-9.)
        fetch(pre);
        if (T0 + 1 < T1) fetch(nxt);
        park(0, pre);
        if (T0 + 2 < T1) fetch(pre);
        __syncthreads();                                          // (the consumers' first barrier)
        int b = 0, T = T0 + 1;
        for (;;) {
            if (T >= T1) break;
            b ^= 1; park(b, nxt);
            if (T + 2 < T1) fetch(nxt);
            __syncthreads();
            if (++T >= T1) break;
            b ^= 1; park(b, pre);
            if (T + 2 < T1) fetch(pre);
            __syncthreads();
            ++T;
        }
        return;
    }

    // ---- consumers.  filter bank -> LDS: SM_G regions of 64 rows x 128 B, same pair swizzle as the convolution tiles
    const int pr = wave & 3, cj = wave >> 2;                      // tile rows 2 pr, 2 pr + 1; output channels 32 cj .. 32 cj + 31
    {
        const int gs = (lane & 15) ^ ((4 * pr + (lane >> 4)) & 15);
        const int rl = 8 * pr + 2 * (lane >> 4) + (gs >> 3), pc16 = (gs & 7) * 16;
        const rsrc_t rsw = make_rsrc(wt16, (size_t)64 * SM_G * 128);
#pragma unroll
        for (int g = 0; g < SM_G; ++g)
            dma16(rsw, wl + g * 8192 + pr * 1024 + cj * 4096, ((rl + 32 * cj) * SM_G + g) * 128 + pc16, 0);
    }
    int fo[4];
    {
        const int r = lane & 31, v = r >> 1, h = lane >> 5;
#pragma unroll
        for (int k = 0; k < 4; ++k) fo[k] = v * 256 + ((((r & 1) * 8 + 2 * k + h) ^ v) * 16);
    }
    // this lane's output pixel inside a tile and its 12 fragment rows: fragment (g, kc) is input row (c, ky) = divmod(4g+2kc+h, 7)
    const int py = 2 * pr + ((lane & 31) >> 4), px = lane & 15;
    int rowoff[2 * SM_G];
#pragma unroll
    for (int f = 0; f < 2 * SM_G; ++f) {
        int rr = 2 * f + (lane >> 5);
        rr = rr < 21 ? rr : 20;                                   // rows 21..23 carry zero weights: any finite data will do
        rowoff[f] = ((rr / 7) * SM_IH + rr % 7 + 2 * py) * SM_IS + 2 * px;
    }
    ShConvArgs e;
    e.bias = bias; e.res = nullptr; e.res_f32 = 0; e.act = OMNI_ACT_RELU; e.Cout = 64; e.dst = dst; e.post = nullptr; e.post_rows = 1; e.epi_lds = epi_lds;
    wait_vm<0>();                                                 // the filter bank has landed
    __syncthreads();                                              // ... everybody's; the first image is there
    // the K loop of one tile (image buffer b) and the epilogue of one tile, as two steps: the channel halves run them in OPPOSITE order between two
    // barriers — half 0: K loop(T), epilogue(T); half 1: epilogue(T - 1), K loop(T) — so that of the two consumers of a SIMD one is in its
    // fragment reads / matrix instructions while the other is in its conversions / stores (in the same order both sat in the same phase: the
    // tile cost the SUM of the two chains whatever the number of waves)
    auto kloop = [&](int b, f16v (&acc)[1], f16v (&acc1)[1]) {
        const _Float16* ih = imh[b];
        const _Float16* il = iml[b];
        acc[0] = (f16v)(0.0f); acc1[0] = (f16v)(0.0f);
#pragma unroll
        for (int g = 0; g < SM_G; ++g)
#pragma unroll
            for (int kc = 0; kc < 2; ++kc) {
                // 8 consecutive pixels from an even column: four 4-byte reads per half image (every input pixel serves ~28 fragments and
                // is split once, at load time)
                const int ro = rowoff[2 * g + kc];
                h8v ah, al;
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const h2v xh = *reinterpret_cast<const h2v*>(ih + ro + 2 * u), xl = *reinterpret_cast<const h2v*>(il + ro + 2 * u);
                    ah[2 * u] = xh[0]; ah[2 * u + 1] = xh[1]; al[2 * u] = xl[0]; al[2 * u + 1] = xl[1];
                }
                const unsigned char* bp = wl + g * 8192 + cj * 4096;
                const h8v bh = *reinterpret_cast<const h8v*>(bp + fo[kc]);
                const h8v bl = *reinterpret_cast<const h8v*>(bp + fo[2 + kc]);
                acc[0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(bh, ah, acc[0], 0, 0, 0);
                acc1[0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(bl, ah, acc1[0], 0, 0, 0);
                acc1[0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(bh, al, acc1[0], 0, 0, 0);
            }
    };
    auto epi = [&](int T, const f16v (&acc)[1], const f16v (&acc1)[1]) {
        int m, oy0, ox0;
        where(T, m, oy0, ox0);
        const int c0[1] = {32 * cj};
        if (e.epi_lds) {                                          // 151 MB of output at 8 panoramas: as 16-byte pieces (the wave's two rows of 16 pixels)
            const size_t ra = ((size_t)m * Po + oy0 + 2 * pr) * Po + ox0;
            epilogue_tile_lds<1>(acc, acc1, e, ra, 32, c0, lane, etile[wave], ra + Po);
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        } else epilogue_row<1>(acc, acc1, e, ((size_t)m * Po + oy0 + py) * Po + ox0 + px, c0, lane, true);
    };
    f16v acc[1], acc1[1];
    int b = 0;
    if (cj == 0) {
        for (int T = T0; T < T1; ++T) {
            kloop(b, acc, acc1);
            epi(T, acc, acc1);
            if (T + 1 >= T1) break;                               // (the producers leave at the same point: no barrier after the last tile)
            wait_lds_reads();
            __syncthreads();                                      // this image buffer is free, the other one is complete
            b ^= 1;
        }
    } else {
        for (int T = T0; T < T1; ++T) {
            if (T > T0) epi(T - 1, acc, acc1);
            kloop(b, acc, acc1);
            if (T + 1 >= T1) break;
            wait_lds_reads();
            __syncthreads();
            b ^= 1;
        }
        epi(T1 - 1, acc, acc1);
    }
}

// Second half of the fused heads (see conv3x3_up2_g1_kernel<HEADS>): one thread per output pixel adds, for each head, the partial sums of the three
// source rows (dy = -1, 0, +1: row y + dy of its tile, (dy, head) plane, position 1 + x % 32) and, at a tile's first / last column, the outer sums
// of the horizontally neighbouring tile (its pixel 32 / pixel -1 slots) — in that fixed order — then heads_kernel's own tail (bias, ReLU, sigmoid, product).
__global__ __launch_bounds__(256) void heads_finish_kernel(const float* __restrict__ hr, float bp, float bw, float* __restrict__ outa, float* __restrict__ outc,
                                                           int M, int P, int conf)
{
    const size_t i4 = (size_t)blockIdx.x * 256 + threadIdx.x;  // four consecutive pixels of a row per thread (16-byte loads and stores)
    if (i4 >= (size_t)M * P * P / 4) return;
    const size_t i = i4 * 4;
    const int x = (int)(i % P), y = (int)((i / P) % P), m = (int)(i / ((size_t)P * P));
    const int tw = P / HT_W, th = P / 4, c = x & 31;
    f4v s[2] = {(f4v)(0.0f), (f4v)(0.0f)};
#pragma unroll
    for (int dy = -1; dy <= 1; ++dy) {
        const int py = y + dy;
        if ((unsigned)py >= (unsigned)P) continue;
        const size_t tile = ((size_t)m * th + (py >> 2)) * tw + (x >> 5);
        const float* rowp = hr + (tile * 4 + (py & 3)) * (6 * HR_PITCH) + (dy + 1) * 2 * HR_PITCH;
#pragma unroll
        for (int hd = 0; hd < 2; ++hd) {
            s[hd] += *reinterpret_cast<const f4v*>(rowp + hd * HR_PITCH + c);
            if (c == 0 && x > 0) s[hd].x += rowp[hd * HR_PITCH + 33 - 4 * 6 * HR_PITCH];        // the left neighbour tile's pixel 32
            if (c == 28 && x + 4 < P) s[hd].w += rowp[hd * HR_PITCH + 32 + 4 * 6 * HR_PITCH];   // the right neighbour tile's pixel -1
        }
    }
    f4v oa, oc;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        const float ap = s[0][e] + bp, aw = s[1][e] + bw;
        const float pr = fmaxf(ap, 0.0f), cf = 1.0f / (1.0f + expf(-aw));
        oa[e] = conf ? pr * cf : pr; oc[e] = cf;
    }
    *reinterpret_cast<f4v*>(outa + i) = oa;
    if (outc) *reinterpret_cast<f4v*>(outc + i) = oc;
}

// dst = act(sum_s ws[s] + bias + res): the deterministic second pass of a split-K launch (4 channels per thread)
__global__ __launch_bounds__(256) void sh_splitk_reduce_kernel(const float* __restrict__ ws, const float* __restrict__ bias,
                                                               const void* __restrict__ res, void* __restrict__ dst,
                                                               size_t n4, int Cout, int splitk, size_t slab, int act, int dst_sh, int res_f32)
{
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n4) return;
    const size_t o = i * 4;
    f4v v = *reinterpret_cast<const f4v*>(ws + o);
    for (int s = 1; s < splitk; ++s) v += *reinterpret_cast<const f4v*>(ws + (size_t)s * slab + o);
    splitk_finish(v, o, bias, res, dst, Cout, act, dst_sh, res_f32);
}

// The same second pass for the transformer's fc2 (model/blocks.py:83-88: x = x + mlp(norm2(x)), then the next block's norm1 / encoder_norm) with the
// LayerNorm that follows it anyway in the same kernel: tok = sum_s ws[s] + bias + res (fp32, written: it is the next residual) and y = LayerNorm(tok)
// (SH: the next GEMM's operand, or fp32: encoder_norm).  One wave per row of 512; element for element the operations of sh_splitk_reduce_kernel
// followed by layernorm512_kernel (omni_net.hip) in their order: the bits of the two launches.
template <bool SH>
__global__ __launch_bounds__(256) void sh_splitk_reduce_ln512_kernel(const float* __restrict__ ws, const float* __restrict__ bias, const float* __restrict__ res,
                                                                     float* __restrict__ tok, const float* __restrict__ g, const float* __restrict__ b,
                                                                     void* __restrict__ y, int rows, int splitk, size_t slab, float eps)
{
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (row >= rows) return;
    const size_t o = (size_t)row * 512;
    f4v v0 = *reinterpret_cast<const f4v*>(ws + o + lane * 4), v1 = *reinterpret_cast<const f4v*>(ws + o + 256 + lane * 4);
    for (int s = 1; s < splitk; ++s) {
        v0 += *reinterpret_cast<const f4v*>(ws + (size_t)s * slab + o + lane * 4);
        v1 += *reinterpret_cast<const f4v*>(ws + (size_t)s * slab + o + 256 + lane * 4);
    }
    if (bias) { v0 += *reinterpret_cast<const f4v*>(bias + lane * 4); v1 += *reinterpret_cast<const f4v*>(bias + 256 + lane * 4); }
    if (res) { v0 += *reinterpret_cast<const f4v*>(res + o + lane * 4); v1 += *reinterpret_cast<const f4v*>(res + o + 256 + lane * 4); }
    *reinterpret_cast<f4v*>(tok + o + lane * 4) = v0; *reinterpret_cast<f4v*>(tok + o + 256 + lane * 4) = v1;
    float sum = (v0.x + v0.y) + (v0.z + v0.w) + (v1.x + v1.y) + (v1.z + v1.w);
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) sum += __shfl_xor(sum, d);
    const float mean = sum * (1.0f / 512.0f);
    v0 -= mean; v1 -= mean;
    float q = (v0.x * v0.x + v0.y * v0.y) + (v0.z * v0.z + v0.w * v0.w) + (v1.x * v1.x + v1.y * v1.y) + (v1.z * v1.z + v1.w * v1.w);
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) q += __shfl_xor(q, d);
    const float rstd = 1.0f / sqrtf(q * (1.0f / 512.0f) + eps);
    const f4v g0 = *reinterpret_cast<const f4v*>(g + lane * 4), g1 = *reinterpret_cast<const f4v*>(g + 256 + lane * 4);
    const f4v b0 = *reinterpret_cast<const f4v*>(b + lane * 4), b1 = *reinterpret_cast<const f4v*>(b + 256 + lane * 4);
    act_store4<SH>(y, o + lane * 4, v0 * rstd * g0 + b0);
    act_store4<SH>(y, o + 256 + lane * 4, v1 * rstd * g1 + b1);
}

// fp32 NHWC <-> SH (4 channels per thread)
__global__ __launch_bounds__(256) void sh_from_f32_kernel(const float* __restrict__ src, void* __restrict__ dst, size_t n4)
{
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n4) return;
    h4v hi, lo; sh_split4(*reinterpret_cast<const f4v*>(src + i * 4), hi, lo);
    unsigned char* dp = (unsigned char*)dst + sh_off(i * 4);
    *reinterpret_cast<h4v*>(dp) = hi; *reinterpret_cast<h4v*>(dp + 64) = lo;
}
__global__ __launch_bounds__(256) void sh_to_f32_kernel(const void* __restrict__ src, float* __restrict__ dst, size_t n4)
{
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n4) return;
    const unsigned char* sp = (const unsigned char*)src + sh_off(i * 4);
    *reinterpret_cast<f4v*>(dst + i * 4) = sh_join4(*reinterpret_cast<const h4v*>(sp), *reinterpret_cast<const h4v*>(sp + 64));
}

// ------------------------------------------------------------------ GEMM over a handful of rows (a lone panorama's tokens)
// out[rows <= 32, N] = act(x[rows, K] . W[N, K]^T + bias + res): the 24 transformer GEMMs of ONE panorama have 18 rows — one column
// tile of the matrix instruction — and are nothing but a stream of weights (1-4 MB each) behind a launch.  Through the tile kernel
// above they cost 8-19 us each (16-64 barrier-synchronised K-steps through LDS, fc2 a split-K launch plus its reduction); here a block
// owns 32 output channels, its 8 waves split K between them and fetch both operands STRAIGHT INTO REGISTERS in fragment order (no LDS,
// no barrier in the K loop, up to four K-steps = 32 sixteen-byte loads per lane in flight), and the 8 partial tiles meet once in LDS
// in a fixed order.  Within a 32-channel group lane half h takes halfs 16h .. 16h+15 (32 contiguous bytes) for BOTH operands: which k
// meets which inside one matrix instruction is free as long as the two sides agree.
struct RowsGemmArgs {
    const void* x; const void* wt; const float* bias; const float* res; void* dst;
    int rows, K, N, act, dst_sh;
    // K slices (round 6): gemm_rows_sh_kernel with blockIdx.y = slice s writes its RAW partial sums to parts[s][rows][N] (no bias / residual / activation);
    // gemm_rows_ln_sh_kernel with nparts > 0 takes its input as x = sum_s parts[s] + pbias + pres (and block 0 writes it to xout: the next residual)
    float* parts; int nparts; const float* pbias; const float* pres; float* xout;
};

// dst (fragment order, see gemm_rows_sh_kernel) <- src [N][K/32][hi32|lo32]; one 16-byte piece per thread
__global__ __launch_bounds__(256) void gemm_rows_pack_kernel(const unsigned char* __restrict__ src, unsigned char* __restrict__ dst, int ksteps, size_t pieces)
{
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= pieces) return;
    const int lane = i & 63, f = (i >> 6) & 3;
    const size_t bk = i >> 8;
    const int ks = bk % ksteps; const size_t b = bk / ksteps;
    const int r = lane & 31, h = lane >> 5, part = f >> 1, kc = f & 1;
    *reinterpret_cast<f4v*>(dst + i * 16) = *reinterpret_cast<const f4v*>(src + ((b * 32 + r) * ksteps + ks) * 128 + part * 64 + (h * 16 + kc * 8) * 2);
}

template <int KPW>                                               // K-steps per wave (K = 256 * KPW)
__global__ __launch_bounds__(512) void gemm_rows_sh_kernel(RowsGemmArgs a)
{
    constexpr int NWV = 8, DEPTH = KPW < 4 ? KPW : 4, PITCH = 36;
    __shared__ float red[NWV][32][PITCH];
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6, r = lane & 31, h = lane >> 5;
    const int col0 = blockIdx.x * 32, ksteps = a.K >> 5, ks0 = (blockIdx.y * NWV + wave) * KPW;       // (blockIdx.y: the K slice of a sliced launch)
    // weights in FRAGMENT ORDER (omni_gemm_rows_pack): [column tile][K-step][hi kc0, hi kc1, lo kc0, lo kc1][lane] x 16 B — a wave's load is
    // one contiguous KiB (8 cache lines) instead of 32 B out of each of 32 lines 8 KiB apart, which made the address unit the bound
    const unsigned char* wp = (const unsigned char*)a.wt + ((size_t)blockIdx.x * ksteps + ks0) * 4096 + lane * 16;
    const unsigned char* xp = (const unsigned char*)a.x + ((size_t)r * ksteps + ks0) * 128 + h * 32;
    const bool live = r < a.rows;                                // token columns past the end stay zero and are never stored
    h8v wh[DEPTH][2], wl[DEPTH][2], xh[DEPTH][2], xl[DEPTH][2];
    auto fetch = [&](int slot, int i) {
#pragma unroll
        for (int kc = 0; kc < 2; ++kc) {
            wh[slot][kc] = *reinterpret_cast<const h8v*>(wp + i * 4096 + kc * 1024);
            wl[slot][kc] = *reinterpret_cast<const h8v*>(wp + i * 4096 + 2048 + kc * 1024);
            xh[slot][kc] = live ? *reinterpret_cast<const h8v*>(xp + i * 128 + kc * 16) : (h8v)(_Float16)0.0f;
            xl[slot][kc] = live ? *reinterpret_cast<const h8v*>(xp + i * 128 + 64 + kc * 16) : (h8v)(_Float16)0.0f;
        }
    };
#pragma unroll
    for (int i = 0; i < DEPTH; ++i) fetch(i, i);
    f16v acc = (f16v)(0.0f), acc1 = (f16v)(0.0f);
#pragma unroll
    for (int i = 0; i < KPW; ++i) {
#pragma unroll
        for (int kc = 0; kc < 2; ++kc) {
            acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(wh[i % DEPTH][kc], xh[i % DEPTH][kc], acc, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(wl[i % DEPTH][kc], xh[i % DEPTH][kc], acc1, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(wh[i % DEPTH][kc], xl[i % DEPTH][kc], acc1, 0, 0, 0);
        }
        if (i + DEPTH < KPW) fetch(i % DEPTH, i + DEPTH);
    }
    // D = W x tokens: column (lane & 31) = token, row (reg & 3) + 8 (reg >> 2) + 4 (lane >> 5) = channel
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        f4v v;
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = fmaf(acc1[4 * q + e], 4.8828125e-4f, acc[4 * q + e]);
        *reinterpret_cast<f4v*>(&red[wave][r][8 * q + 4 * h]) = v;
    }
    __syncthreads();
    const int tok = t >> 3, c4 = (t & 7) * 4;
    if (t >= 256 || tok >= a.rows) return;
    f4v v = *reinterpret_cast<const f4v*>(&red[0][tok][c4]);
#pragma unroll
    for (int w = 1; w < NWV; ++w) v += *reinterpret_cast<const f4v*>(&red[w][tok][c4]);
    const size_t o = (size_t)tok * a.N + col0 + c4;
    if (a.parts) { *reinterpret_cast<f4v*>(a.parts + (size_t)blockIdx.y * a.rows * a.N + o) = v; return; }     // a K slice: raw partial sums
    if (a.bias) v += *reinterpret_cast<const f4v*>(a.bias + col0 + c4);
    if (a.res) v += *reinterpret_cast<const f4v*>(a.res + o);
    if (a.act == OMNI_ACT_RELU) { v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f); }
    else if (a.act == OMNI_ACT_GELU) {
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = 0.5f * v[e] * (1.0f + erff(v[e] * 0.70710678118654752440f));
    }
    if (a.dst_sh) act_store4<true>(a.dst, o, v);
    else          act_store4<false>(a.dst, o, v);
}

// LayerNorm(512) + the rows GEMM in one launch (K = 512): every block normalises all <= 32 rows itself — one wave per row, layernorm512_kernel's
// own loads, butterflies and expression, so the split-half values are the ones that kernel would have written — into LDS, where the
// fragment loads then find them; the block's weights are already travelling (they do not depend on x).  Saves a 3.4-us launch per
// LayerNorm of a lone panorama's transformer (12 of its 42); same bits as omni_layernorm512_sh + omni_gemm_rows_sh_f16x3.
__global__ __launch_bounds__(512) void gemm_rows_ln_sh_kernel(RowsGemmArgs a, const float* __restrict__ lg, const float* __restrict__ lb, float eps)
{
    constexpr int NWV = 8, KPW = 2, DEPTH = 2, PITCH = 36;
    __shared__ float red[NWV][32][PITCH];
    __shared__ __attribute__((aligned(16))) unsigned char xs[32 * 2048];           // LayerNorm(x) as split-half rows [32][16 groups][hi32|lo32]
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6, r = lane & 31, h = lane >> 5;
    const int col0 = blockIdx.x * 32, ksteps = 16, ks0 = wave * KPW;
    const unsigned char* wp = (const unsigned char*)a.wt + ((size_t)blockIdx.x * ksteps + ks0) * 4096 + lane * 16;
    h8v wh[DEPTH][2], wl[DEPTH][2];
#pragma unroll
    for (int i = 0; i < DEPTH; ++i)
#pragma unroll
        for (int kc = 0; kc < 2; ++kc) {
            wh[i][kc] = *reinterpret_cast<const h8v*>(wp + i * 4096 + kc * 1024);
            wl[i][kc] = *reinterpret_cast<const h8v*>(wp + i * 4096 + 2048 + kc * 1024);
        }
    // ---- LayerNorm, one wave per row (rows wave, wave + 8, ...): layernorm512_kernel<true>, writing to LDS
    for (int row = wave; row < a.rows; row += NWV) {
        f4v v0, v1;
        if (a.nparts > 0) {
            // the input is the previous GEMM's K slices: x = (slice 0 + slice 1 + ...) + bias + residual, in that order (sh_splitk_reduce_ln512_kernel's);
            // every block forms it for itself, block 0 also stores it (the next residual)
            const size_t o = (size_t)row * 512, slab = (size_t)a.rows * 512;
            v0 = *reinterpret_cast<const f4v*>(a.parts + o + lane * 4); v1 = *reinterpret_cast<const f4v*>(a.parts + o + 256 + lane * 4);
            for (int sl = 1; sl < a.nparts; ++sl) {
                v0 += *reinterpret_cast<const f4v*>(a.parts + sl * slab + o + lane * 4);
                v1 += *reinterpret_cast<const f4v*>(a.parts + sl * slab + o + 256 + lane * 4);
            }
            if (a.pbias) { v0 += *reinterpret_cast<const f4v*>(a.pbias + lane * 4); v1 += *reinterpret_cast<const f4v*>(a.pbias + 256 + lane * 4); }
            if (a.pres) { v0 += *reinterpret_cast<const f4v*>(a.pres + o + lane * 4); v1 += *reinterpret_cast<const f4v*>(a.pres + o + 256 + lane * 4); }
            if (blockIdx.x == 0 && a.xout) { *reinterpret_cast<f4v*>(a.xout + o + lane * 4) = v0; *reinterpret_cast<f4v*>(a.xout + o + 256 + lane * 4) = v1; }
        } else {
            const float* p = (const float*)a.x + (size_t)row * 512;
            v0 = *reinterpret_cast<const f4v*>(p + lane * 4); v1 = *reinterpret_cast<const f4v*>(p + 256 + lane * 4);
        }
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
    __syncthreads();
    const unsigned char* xp = xs + ((size_t)r * ksteps + ks0) * 128 + h * 32;
    const bool live = r < a.rows;
    f16v acc = (f16v)(0.0f), acc1 = (f16v)(0.0f);
#pragma unroll
    for (int i = 0; i < KPW; ++i)
#pragma unroll
        for (int kc = 0; kc < 2; ++kc) {
            const h8v xh = live ? *reinterpret_cast<const h8v*>(xp + i * 128 + kc * 16) : (h8v)(_Float16)0.0f;
            const h8v xl = live ? *reinterpret_cast<const h8v*>(xp + i * 128 + 64 + kc * 16) : (h8v)(_Float16)0.0f;
            acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(wh[i][kc], xh, acc, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(wl[i][kc], xh, acc1, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(wh[i][kc], xl, acc1, 0, 0, 0);
        }
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        f4v v;
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = fmaf(acc1[4 * q + e], 4.8828125e-4f, acc[4 * q + e]);
        *reinterpret_cast<f4v*>(&red[wave][r][8 * q + 4 * h]) = v;
    }
    __syncthreads();
    const int tok = t >> 3, c4 = (t & 7) * 4;
    if (t >= 256 || tok >= a.rows) return;
    f4v v = *reinterpret_cast<const f4v*>(&red[0][tok][c4]);
#pragma unroll
    for (int w = 1; w < NWV; ++w) v += *reinterpret_cast<const f4v*>(&red[w][tok][c4]);
    const size_t o = (size_t)tok * a.N + col0 + c4;
    if (a.bias) v += *reinterpret_cast<const f4v*>(a.bias + col0 + c4);
    if (a.res) v += *reinterpret_cast<const f4v*>(a.res + o);
    if (a.act == OMNI_ACT_RELU) { v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f); }
    else if (a.act == OMNI_ACT_GELU) {
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = 0.5f * v[e] * (1.0f + erff(v[e] * 0.70710678118654752440f));
    }
    if (a.dst_sh) act_store4<true>(a.dst, o, v);
    else          act_store4<false>(a.dst, o, v);
}

// Winograd F(2x2, 3x3) input transform: V[p = 4 xi + nu][tile][c] = (B^T d B)[xi][nu], d = the 4 x 4 window (rows 2 ty - 1 .. 2 ty + 2, columns 2 tx - 1 .. 2 tx + 2,
// zeros outside the image) of tile (m, ty, tx); B^T = [[1,0,-1,0],[0,1,1,0],[0,-1,1,0],[0,1,0,-1]].  SH in, SH out; one thread per (tile, 4 channels).
__global__ __launch_bounds__(256) void wino_input_sh_kernel(const void* __restrict__ src, void* __restrict__ V, int M, int H, int W, int C, size_t nt)
{
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    const int cq = C >> 2;
    if (i >= nt * cq) return;
    const size_t tl = i / cq;
    const int c = (int)(i - tl * cq) * 4;
    const int tw = W >> 1, th = H >> 1;
    const int m = (int)(tl / (size_t)(th * tw)), rem = (int)(tl - (size_t)m * th * tw), ty = rem / tw, tx = rem - ty * tw;
    f4v d[4][4];
#pragma unroll
    for (int a_ = 0; a_ < 4; ++a_)
#pragma unroll
        for (int b_ = 0; b_ < 4; ++b_) {
            const int y = 2 * ty - 1 + a_, x = 2 * tx - 1 + b_;
            d[a_][b_] = ((unsigned)y < (unsigned)H && (unsigned)x < (unsigned)W) ? act_load4<true>(src, (((size_t)m * H + y) * W + x) * C + c) : (f4v)(0.0f);
        }
    f4v u[4][4];                                                  // B^T d
#pragma unroll
    for (int b_ = 0; b_ < 4; ++b_) { u[0][b_] = d[0][b_] - d[2][b_]; u[1][b_] = d[1][b_] + d[2][b_]; u[2][b_] = d[2][b_] - d[1][b_]; u[3][b_] = d[1][b_] - d[3][b_]; }
#pragma unroll
    for (int xi = 0; xi < 4; ++xi) {                              // (B^T d) B
        const f4v v0 = u[xi][0] - u[xi][2], v1 = u[xi][1] + u[xi][2], v2 = u[xi][2] - u[xi][1], v3 = u[xi][1] - u[xi][3];
        act_store4<true>(V, ((size_t)(4 * xi + 0) * nt + tl) * C + c, v0);
        act_store4<true>(V, ((size_t)(4 * xi + 1) * nt + tl) * C + c, v1);
        act_store4<true>(V, ((size_t)(4 * xi + 2) * nt + tl) * C + c, v2);
        act_store4<true>(V, ((size_t)(4 * xi + 3) * nt + tl) * C + c, v3);
    }
}

template <int BM, int BN, int WM, int WN, int NST = 3, int NL = 0>
void launch_sh(ShConvArgs a, hipStream_t s)
{
    const int tiles = ((a.rows + BM - 1) / BM) * (a.Cout / BN);
    if constexpr (NL > 0 && WM * WN == 8) {
        if (omni_options().conv_pingpong) {                       // the SIMD's two matrix waves in anti-phase (same bits)
            hipLaunchKernelGGL((conv_sh_kernel<BM, BN, WM, WN, NST, NL, true>), dim3(tiles, a.splitk > 1 ? a.splitk : 1), dim3(64 * (WM * WN + NL)), 0, s, a);
            return;
        }
    }
    hipLaunchKernelGGL((conv_sh_kernel<BM, BN, WM, WN, NST, NL>), dim3(tiles, a.splitk > 1 ? a.splitk : 1), dim3(64 * (WM * WN + NL)), 0, s, a);
}

}  // namespace

// out[M,Ho,Wo,Cout] = act(conv(src1 ++ src2, wt16) + bias + res) with SH activations (see the file header).
// src1/src2: SH tensors; fmt bit 0: dst is SH (else fp32 NHWC); fmt bit 1: res is fp32 NHWC (else SH); fmt bit 2: latency form (few images: the im2col
// tile kernel also where the halo kernel for small images would be taken; equal to it up to the K summation order); wt16 as for
// omni_conv2d_nhwc_f16x3_ws.  A plain GEMM is the case H = W = KH = KW = 1 (rows = M).
// Requirements: C1, C2, Cout multiples of 32, kernels up to 3x3.  split-K as in omni_conv2d_nhwc_f32_ws.
// `post` (or null): fp32 [post_elems / Cout][Cout] added after the activation, output row index modulo its row count — layer1 + point_feat
// (model/spherical_model.py:258) inside layer1's last convolution instead of a pass of its own.  Not with split-K.
static int conv2d_sh_impl(const void* src1, const void* src2, const void* wt16, const float* bias,
                          const void* res, void* dst, int fmt, int M, int H, int W, int C1, int C2, int Cout,
                          int KH, int KW, int stride, int pad, int act, int splitk, float* ws, size_t ws_bytes,
                          const float* post, size_t post_elems, omni_stream_t stream, bool reduce = true);
extern "C" int omni_conv2d_sh_f16x3_ws(const void* src1, const void* src2, const void* wt16, const float* bias,
                                       const void* res, void* dst, int fmt, int M, int H, int W, int C1, int C2, int Cout,
                                       int KH, int KW, int stride, int pad, int act, int splitk, float* ws, size_t ws_bytes,
                                       omni_stream_t stream)
{
    return conv2d_sh_impl(src1, src2, wt16, bias, res, dst, fmt, M, H, W, C1, C2, Cout, KH, KW, stride, pad, act, splitk, ws, ws_bytes,
                          nullptr, 0, stream);
}
extern "C" int omni_conv2d_sh_f16x3_post_ws(const void* src1, const void* src2, const void* wt16, const float* bias,
                                            const void* res, void* dst, int fmt, int M, int H, int W, int C1, int C2, int Cout,
                                            int KH, int KW, int stride, int pad, int act, int splitk, float* ws, size_t ws_bytes,
                                            const float* post, size_t post_elems, omni_stream_t stream)
{
    return conv2d_sh_impl(src1, src2, wt16, bias, res, dst, fmt, M, H, W, C1, C2, Cout, KH, KW, stride, pad, act, splitk, ws, ws_bytes,
                          post, post_elems, stream);
}
// A split-K GEMM with 512 output columns (the transformer's fc2) whose second pass also applies the LayerNorm that follows: tok [rows,512] fp32 =
// x . wt16^T + bias + res (res fp32 [rows,512], may alias nothing), y = LayerNorm(tok; ln_g, ln_b, eps) as SH (fmt bit 0) or fp32.  splitk >= 2 (the
// caller's plan); one launch fewer per transformer layer than omni_conv2d_sh_f16x3_ws + omni_layernorm512_*, the same bits.
extern "C" int omni_gemm_sh_f16x3_ln512_ws(const void* x, const void* wt16, const float* bias, const float* res, float* tok, const float* ln_g, const float* ln_b,
                                           float eps, void* y, int fmt, int rows, int K, int splitk, float* ws, size_t ws_bytes, omni_stream_t stream)
{
    if (!x || !wt16 || !tok || !ln_g || !ln_b || !y) OMNI_FAIL(OMNI_ERR_INVALID, "omni_gemm_sh_f16x3_ln512: null pointer");
    if (rows <= 0 || K <= 0 || K % 32) OMNI_FAIL(OMNI_ERR_INVALID, "omni_gemm_sh_f16x3_ln512: bad shape");
    const int S = std::min(splitk, K / 32);
    if (S < 2) OMNI_FAIL(OMNI_ERR_INVALID, "omni_gemm_sh_f16x3_ln512: needs a split-K plan (splitk >= 2); an unsplit GEMM writes its result in its own epilogue");
    const int rc = conv2d_sh_impl(x, nullptr, wt16, bias, res, tok, 2, rows, 1, 1, K, 0, 512, 1, 1, 1, 0, OMNI_ACT_NONE, S, ws, ws_bytes, nullptr, 0, stream, false);
    if (rc != OMNI_OK) return rc;
    hipStream_t s = (hipStream_t)stream;
    if (fmt & 1) hipLaunchKernelGGL(sh_splitk_reduce_ln512_kernel<true>, dim3((rows + 3) / 4), dim3(256), 0, s, (const float*)ws, bias, res, tok, ln_g, ln_b, y, rows, S, (size_t)rows * 512, eps);
    else         hipLaunchKernelGGL(sh_splitk_reduce_ln512_kernel<false>, dim3((rows + 3) / 4), dim3(256), 0, s, (const float*)ws, bias, res, tok, ln_g, ln_b, y, rows, S, (size_t)rows * 512, eps);
    OMNI_HIP(hipGetLastError());
    return OMNI_OK;
}
extern "C" int omni_conv2d_splitk_plan(long long rows, int Cout, int ksteps);            // omni_conv.hip: the two-launch plan

static int conv2d_sh_impl(const void* src1, const void* src2, const void* wt16, const float* bias,
                          const void* res, void* dst, int fmt, int M, int H, int W, int C1, int C2, int Cout,
                          int KH, int KW, int stride, int pad, int act, int splitk, float* ws, size_t ws_bytes,
                          const float* post, size_t post_elems, omni_stream_t stream, bool reduce)
{
    const int dst_sh = fmt & 1;
    if (!src1 || !wt16 || !dst) OMNI_FAIL(OMNI_ERR_INVALID, "omni_conv2d_sh: null pointer");
    if (C1 <= 0 || C1 % 32 || C2 < 0 || C2 % 32 || Cout <= 0 || Cout % 32 || (C2 > 0 && !src2))
        OMNI_FAIL(OMNI_ERR_INVALID, "omni_conv2d_sh: channels must be multiples of 32");
    if (M <= 0 || H <= 0 || W <= 0 || KH <= 0 || KW <= 0 || KH > 3 || KW > 3 || stride <= 0 || pad < 0)
        OMNI_FAIL(OMNI_ERR_INVALID, "omni_conv2d_sh: bad shape (kernels up to 3x3)");
    ShConvArgs a;
    a.src1 = src1; a.src2 = src2; a.wt = wt16; a.bias = bias; a.res = res; a.dst = dst; a.dst_sh = dst_sh; a.res_f32 = (fmt >> 1) & 1;
    a.dbg = 0; a.noxcd = omni_options().conv_noxcd; a.wt_major = 0; a.epi_lds = omni_options().conv_epi_lds && !(fmt & 4);   // (fmt bit 2, one panorama: the extra barrier and LDS round trip cost more than the wider stores save)
#ifdef OMNI_DEBUG_BUILD
    a.dbg = omni_debug_bits("OMNI_CONV_DBG");
#endif
    a.M = M; a.H = H; a.W = W; a.C1 = C1; a.C2 = C2; a.Cout = Cout;
    a.KH = KH; a.KW = KW; a.stride = stride; a.pad = pad; a.act = act;
    a.Ho = (H + 2 * pad - KH) / stride + 1; a.Wo = (W + 2 * pad - KW) / stride + 1;
    const long long rows = (long long)M * a.Ho * a.Wo;
    if (rows <= 0 || rows >= (1ll << 31) || (long long)M * H * W >= (1ll << 31))
        OMNI_FAIL(OMNI_ERR_UNSUPPORTED, "omni_conv2d_sh: too many pixels for 32-bit row indices");
    a.rows = (int)rows;
    const int ksteps = KH * KW * ((C1 + C2) / 32);
    if ((long long)M * H * W * (C1 > C2 ? C1 : C2) * 4 >= (1ll << 31) || (long long)Cout * ksteps * 128 >= (1ll << 31))
        OMNI_FAIL(OMNI_ERR_UNSUPPORTED, "omni_conv2d_sh: an operand of 2 GiB or more (32-bit buffer offsets)");
    int S = splitk;
    if (S > ksteps) S = ksteps;
    if (S > 1 && (!ws || ws_bytes < (size_t)S * rows * Cout * sizeof(float)))
        OMNI_FAIL(OMNI_ERR_INVALID, "omni_conv2d_sh: split-K workspace too small");
    a.splitk = S > 1 ? S : 1; a.ws = ws;
    // block order of conv_sh_kernel: weight-stationary per XCD where the weight matrix is larger than the activation tensor(s) (option conv_wt_major:
    // 1 auto | 0 never | 2 always)
    a.wt_major = omni_options().conv_wt_major == 2 || (omni_options().conv_wt_major == 1 &&
                 (long long)Cout * ksteps * 128 > (long long)M * H * W * (C1 + C2) * 4) ? 1 : 0;
    a.post = post; a.post_rows = 1; a.wino_th = a.wino_tw = a.wino_pix = 0;
    if (post) {
        if (Cout <= 0 || post_elems == 0 || post_elems % (size_t)Cout || post_elems / (size_t)Cout > 0x7fffffffull)
            OMNI_FAIL(OMNI_ERR_INVALID, "omni_conv2d_sh: the post-activation addend must hold whole rows of Cout channels");
        if (a.splitk > 1) OMNI_FAIL(OMNI_ERR_UNSUPPORTED, "omni_conv2d_sh: no post-activation addend with split-K");
        a.post_rows = (unsigned)(post_elems / (size_t)Cout);
    }
    hipStream_t s = (hipStream_t)stream;
    // small square images: the halo kernel over bands of whole image rows (IW > 0) — 16 x 16 (layer2, de_conv1_x: 61.6 -> 49.3 us per layer2
    // convolution at 8 panoramas) by default, 8 x 8 as well with conv_img = 2 (layer3: 52.7 -> 50.9)
    // (the choice must not depend on the number of images: a panorama's bits are the same in every batch size; a caller that runs ONE panorama,
    //  where the launch would be a quarter block per CU, asks for the tile kernel with fmt bit 2)
    if (a.splitk <= 1 && KH == 3 && KW == 3 && stride == 1 && pad == 1 && H == W && (W == 16 || (W == 8 && omni_options().conv_img >= 2)) && rows % 128 == 0 &&
        Cout % 64 == 0 && !(fmt & 4) && omni_options().conv_img > 0 && !omni_options().conv_nohalo) {
        const int grid = (int)(rows / 128) * (Cout / 64);
        if (omni_options().conv_halo_bn == 32) {
            if (W == 16) hipLaunchKernelGGL((conv3x3_halo_sh_kernel<32, 4, false, 16>), dim3(2 * grid), dim3(256), 0, s, a);
            else         hipLaunchKernelGGL((conv3x3_halo_sh_kernel<32, 4, false, 8>), dim3(2 * grid), dim3(256), 0, s, a);
        }
        else if (W == 16) hipLaunchKernelGGL((conv3x3_halo_sh_kernel<64, 4, false, 16>), dim3(grid), dim3(256), 0, s, a);
        else         hipLaunchKernelGGL((conv3x3_halo_sh_kernel<64, 4, false, 8>), dim3(grid), dim3(256), 0, s, a);
        OMNI_HIP(hipGetLastError());
        return OMNI_OK;
    }
    if (a.splitk <= 1 && KH == 3 && KW == 3 && stride == 1 && pad == 1 && W % HT_W == 0 && H % 4 == 0 && !omni_options().conv_nohalo) {
        const int th = (H % 8 == 0 && omni_options().conv_halo_th == 8) ? 8 : 4;
        const int grid = M * (H / th) * (W / HT_W);
        if (th == 8) {
            if (Cout % 64 == 0 && omni_options().conv_halo_bn != 32) hipLaunchKernelGGL((conv3x3_halo_sh_kernel<64, 8>), dim3(grid * (Cout / 64)), dim3(512), 0, s, a);
            else                hipLaunchKernelGGL((conv3x3_halo_sh_kernel<32, 8>), dim3(grid * (Cout / 32)), dim3(512), 0, s, a);
        } else {
            // (fmt bit 2, ONE panorama: the launch is a fraction of a block per CU and costs the length of a block's life — 32-channel blocks are twice as many and
            //  half as long; option conv_halo_bn_lat)
            const bool bn32 = omni_options().conv_halo_bn == 32 || ((fmt & 4) && omni_options().conv_halo_bn_lat == 32);
            if (Cout % 64 == 0 && !bn32) hipLaunchKernelGGL((conv3x3_halo_sh_kernel<64, 4>), dim3(grid * (Cout / 64)), dim3(256), 0, s, a);
            else                hipLaunchKernelGGL((conv3x3_halo_sh_kernel<32, 4>), dim3(grid * (Cout / 32)), dim3(256), 0, s, a);
        }
        OMNI_HIP(hipGetLastError());
        return OMNI_OK;
    }
    // tile: results do not depend on it (every output element is the same k-ordered chain), so it is a pure tuning choice
    // -1 auto (= 8): the largest 8-wave tile the layer allows (256x128, 128x128, else 128x64: 3/8, 1/2, 3/4 of the L2 -> LDS bytes per MFMA
    // of a 64x64 tile) wherever the launch still has >= 128 blocks, 64x64 below that.  Measured interleaved in one process
    // (tools/pipe_ab.py, tools/plain_ab.py): +5.1 % panoramas/s with three forwards in flight (+1.2 % of it from 256x128), +1.7 % for plain
    // calls at 8 panoramas, -1 % at 4, 0 at 1.
    // 0: 64x64 everywhere; 1: 128x64 (4 waves); 2: 128x128 (4 waves); 3 / 4: the 8-wave forms everywhere; 5..7: auto without 256x128, with 64 / 128 / 256 blocks
    int tile = omni_options().conv_sh_tile;
    if (tile < 0) tile = 8;
    if (Cout % 64 != 0) launch_sh<128, 32, 4, 1>(a, s);
    else if (tile == 2 && Cout % 128 == 0) launch_sh<128, 128, 2, 2>(a, s);
    else if (tile == 1) launch_sh<128, 64, 2, 2>(a, s);
    else if (tile == 3) launch_sh<128, 64, 4, 2>(a, s);            // 8 waves
    else if (tile == 4 && Cout % 128 == 0) launch_sh<128, 128, 4, 2>(a, s);
    else if (tile == 8 && Cout % 128 == 0 && ((rows + 255) / 256) * (long long)(Cout / 128) * a.splitk >= std::max(1, omni_options().conv_big_blocks)) launch_sh<256, 128, 4, 2>(a, s);   // each wave a 64x64 tile: 0.67 KB of LDS reads per MFMA instead of 1
    // (128x128 and 128x64 with four LOADER waves beside the eight matrix waves: layer3 51.3 -> 45.8 us, de_conv0_0 90 -> 79, layer4 43.3 -> 41.3, same bits;
    //  tile = 9: without them.  256x128 has no registers to spare for a third wave per SIMD.)
    else if (tile == 8 && Cout % 128 == 0 && ((rows + 127) / 128) * (long long)(Cout / 128) * a.splitk >= 128) launch_sh<128, 128, 4, 2, 3, 4>(a, s);
    else if (tile == 8 && ((rows + 127) / 128) * (long long)(Cout / 64) * a.splitk >= 128) launch_sh<128, 64, 4, 2, 3, 4>(a, s);
    else if (tile == 9 && Cout % 128 == 0 && ((rows + 255) / 256) * (long long)(Cout / 128) * a.splitk >= 128) launch_sh<256, 128, 4, 2>(a, s);
    else if (tile == 9 && Cout % 128 == 0 && ((rows + 127) / 128) * (long long)(Cout / 128) * a.splitk >= 128) launch_sh<128, 128, 4, 2>(a, s);
    else if (tile == 9 && ((rows + 127) / 128) * (long long)(Cout / 64) * a.splitk >= 128) launch_sh<128, 64, 4, 2>(a, s);
    else if (tile >= 5 && tile <= 7 && Cout % 128 == 0 && ((rows + 127) / 128) * (long long)(Cout / 128) * a.splitk >= (32ll << (tile - 4))) launch_sh<128, 128, 4, 2>(a, s);
    else if (tile >= 5 && tile <= 7 && ((rows + 127) / 128) * (long long)(Cout / 64) * a.splitk >= (32ll << (tile - 4))) launch_sh<128, 64, 4, 2>(a, s);
    // one round of at most one block per CU (the transformer GEMMs; every deep layer at batch 1): the K loop is pure latency,
    // keep 5 stages in flight instead of 2 (96 KiB of LDS, which a single resident block can afford)
    else if (((rows + 63) / 64) * (long long)(Cout / 64) * a.splitk <= 256 && ksteps >= 8 && !omni_options().conv_nodeep) {
        // (conv_deep_loaders = 1: four loader waves beside the four matrix waves — at one block per CU a K-step is the four DMA pieces a matrix wave issues,
        //  ~400 cycles for its 192 of matrix work)
        if (omni_options().conv_deep_loaders) launch_sh<64, 64, 2, 2, 6, 4>(a, s);
        else launch_sh<64, 64, 2, 2, 6>(a, s);
    }
    else launch_sh<64, 64, 2, 2>(a, s);
    OMNI_HIP(hipGetLastError());
    if (a.splitk > 1 && reduce) {
        const size_t n4 = (size_t)rows * Cout / 4;
        hipLaunchKernelGGL(sh_splitk_reduce_kernel, dim3((unsigned)((n4 + 255) / 256)), dim3(256), 0, s, (const float*)ws, bias, res, dst,
                           n4, Cout, a.splitk, (size_t)rows * Cout, act, a.dst_sh, a.res_f32);
        OMNI_HIP(hipGetLastError());
    }
    return OMNI_OK;
}

// ---- Winograd F(2x2, 3x3) for 3x3 stride-1 pad-1 convolutions of small images (EXPERIMENTAL, round 6; conv_sh_kernel<.., WINO>)
// omni_wino_input_sh: src SH [M,H,W,C] (H, W even) -> V SH [16][M * H/2 * W/2][C].
extern "C" int omni_wino_input_sh(const void* src, void* V, int M, int H, int W, int C, omni_stream_t stream)
{
    if (!src || !V) OMNI_FAIL(OMNI_ERR_INVALID, "omni_wino_input_sh: null pointer");
    if (M <= 0 || H <= 0 || W <= 0 || (H & 1) || (W & 1) || C <= 0 || C % 32) OMNI_FAIL(OMNI_ERR_INVALID, "omni_wino_input_sh: even image sides, channels a multiple of 32");
    const size_t nt = (size_t)M * (H / 2) * (W / 2), n = nt * (C / 4);
    if (16 * nt * C * 4 >= (1ull << 31)) OMNI_FAIL(OMNI_ERR_UNSUPPORTED, "omni_wino_input_sh: the transformed tensor must stay under 2 GiB (32-bit buffer offsets)");
    hipLaunchKernelGGL(wino_input_sh_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, src, V, M, H, W, C, nt);
    OMNI_HIP(hipGetLastError());
    return OMNI_OK;
}

// dst [M,H,W,Cout] = act(conv3x3(x) + bias + res) from V = omni_wino_input_sh(x) and wt16 = the f16x3 split (omni_conv2d's weight format) of the matrix
// [Cout][16 * C], k = p * C + c, holding U_p = (G g G^T)[p] (G = [[1,0,0],[.5,.5,.5],[.5,-.5,.5],[0,0,1]]).  fmt as omni_conv2d_sh_f16x3_ws (bit 0: dst SH; bit 1: res fp32).
// splitk in {1, 2, 4}: the sixteen positions divided over that many blocks per tile, partial outputs to ws (splitk * M*H*W * Cout floats) and
// sh_splitk_reduce_kernel.  Cout % 64 == 0, C % 32 == 0.  Equal to the direct convolution up to rounding (measured 1e-6 against float64, tools/winograd_proto.py).
extern "C" int omni_conv3x3_wino_sh_f16x3(const void* V, const void* wt16, const float* bias, const void* res, void* dst, int fmt,
                                          int M, int H, int W, int C, int Cout, int act, int splitk, float* ws, size_t ws_bytes, omni_stream_t stream)
{
    if (!V || !wt16 || !dst) OMNI_FAIL(OMNI_ERR_INVALID, "omni_conv3x3_wino_sh: null pointer");
    if (M <= 0 || H <= 0 || W <= 0 || (H & 1) || (W & 1) || C <= 0 || C % 32 || Cout <= 0 || Cout % 64) OMNI_FAIL(OMNI_ERR_INVALID, "omni_conv3x3_wino_sh: even image sides, C % 32 == 0, Cout % 64 == 0");
    if (splitk != 1 && splitk != 2 && splitk != 4) OMNI_FAIL(OMNI_ERR_INVALID, "omni_conv3x3_wino_sh: splitk must be 1, 2 or 4 (it divides the sixteen positions)");
    const long long nt = (long long)M * (H / 2) * (W / 2), pix = (long long)M * H * W;
    if (16 * nt * C * 4 >= (1ll << 31) || (long long)Cout * 16 * C * 4 >= (1ll << 31) || pix >= (1ll << 31)) OMNI_FAIL(OMNI_ERR_UNSUPPORTED, "omni_conv3x3_wino_sh: an operand of 2 GiB or more");
    if (splitk > 1 && (!ws || ws_bytes < (size_t)splitk * pix * Cout * sizeof(float))) OMNI_FAIL(OMNI_ERR_INVALID, "omni_conv3x3_wino_sh: split workspace too small");
    ShConvArgs a;
    a.src1 = V; a.src2 = nullptr; a.wt = wt16; a.bias = bias; a.res = res; a.dst = dst; a.dst_sh = fmt & 1; a.res_f32 = (fmt >> 1) & 1;
    a.dbg = 0; a.noxcd = omni_options().conv_noxcd; a.wt_major = 0; a.epi_lds = 0;
    a.M = 1; a.H = 16; a.W = (int)nt; a.C1 = C; a.C2 = 0; a.Cout = Cout; a.KH = 16; a.KW = 1; a.stride = 1; a.pad = 0; a.act = act;
    a.Ho = H; a.Wo = W; a.rows = (int)nt; a.splitk = splitk; a.ws = ws; a.post = nullptr; a.post_rows = 1;
    a.wino_th = H / 2; a.wino_tw = W / 2; a.wino_pix = (int)pix;
    hipStream_t s = (hipStream_t)stream;
    const int tiles = (int)((nt + 127) / 128) * (Cout / 64);
    hipLaunchKernelGGL((conv_sh_kernel<128, 64, 4, 2, 3, 4, true, true>), dim3(tiles, splitk), dim3(64 * 12), 0, s, a);
    OMNI_HIP(hipGetLastError());
    if (splitk > 1) {
        const size_t n4 = (size_t)pix * Cout / 4;
        hipLaunchKernelGGL(sh_splitk_reduce_kernel, dim3((unsigned)((n4 + 255) / 256)), dim3(256), 0, s, (const float*)ws, bias, res, dst,
                           n4, Cout, splitk, (size_t)pix * Cout, act, a.dst_sh, a.res_f32);
        OMNI_HIP(hipGetLastError());
    }
    return OMNI_OK;
}

// dst = act(conv3x3(pad 1)(bilinear 2x up-sampling of src) + bias): F.interpolate(scale 2, align_corners=False) + ConvBnReLU of the
// decoder (model/spherical_model.py:279-301) in one kernel (conv3x3_halo_sh_kernel<.., UP2>).  src SH [M, Hl, Wl, C], dst [M, 2Hl, 2Wl, Cout]
// SH (fmt bit 0) or fp32; needs 2Wl % 32 == 0, 2Hl % 4 == 0 (OMNI_ERR_UNSUPPORTED otherwise: run omni_upsample_bilinear_sh +
// omni_conv2d_sh_f16x3_ws, which give the same bits).
extern "C" int omni_conv3x3_up2_sh_f16x3(const void* src, const void* wt16, const float* bias, void* dst, int fmt,
                                         int M, int Hl, int Wl, int C, int Cout, int act, omni_stream_t stream)
{
    if (!src || !wt16 || !dst) OMNI_FAIL(OMNI_ERR_INVALID, "omni_conv3x3_up2_sh: null pointer");
    if (M <= 0 || Hl <= 0 || Wl <= 0 || C <= 0 || C % 32 || Cout <= 0 || Cout % 32) OMNI_FAIL(OMNI_ERR_INVALID, "omni_conv3x3_up2_sh: bad shape (channels must be multiples of 32)");
    const int H = 2 * Hl, W = 2 * Wl;
    if (W % HT_W || H % 4) OMNI_FAIL(OMNI_ERR_UNSUPPORTED, "omni_conv3x3_up2_sh: the output must be a multiple of 4 rows x 32 columns");
    if ((long long)M * H * W >= (1ll << 31) || (long long)Cout * 9 * C * 4 >= (1ll << 31))
        OMNI_FAIL(OMNI_ERR_UNSUPPORTED, "omni_conv3x3_up2_sh: tensor too large for 32-bit indices");
    ShConvArgs a;
    a.src1 = src; a.src2 = nullptr; a.wt = wt16; a.bias = bias; a.res = nullptr; a.dst = dst; a.dst_sh = fmt & 1; a.res_f32 = 0;
    a.dbg = 0; a.noxcd = omni_options().conv_noxcd; a.wt_major = 0; a.epi_lds = omni_options().conv_epi_lds && !(fmt & 4);   // (fmt bit 2, one panorama: the extra barrier and LDS round trip cost more than the wider stores save)
#ifdef OMNI_DEBUG_BUILD
    a.dbg = omni_debug_bits("OMNI_CONV_DBG");
#endif
    a.M = M; a.H = H; a.W = W; a.C1 = C; a.C2 = 0; a.Cout = Cout; a.KH = 3; a.KW = 3; a.stride = 1; a.pad = 1; a.act = act;
    a.Ho = H; a.Wo = W; a.rows = M * H * W; a.splitk = 1; a.ws = nullptr; a.post = nullptr; a.post_rows = 1; a.wino_th = a.wino_tw = a.wino_pix = 0;
    const int grid = M * (H / 4) * (W / HT_W);
    if (C == 32 && Cout == 32 && omni_options().conv_up2_persist) {    // de_conv4_0: resident weights, one persistent block of 8 waves per CU
        hipLaunchKernelGGL(conv3x3_up2_g1_kernel<false>, dim3(grid < 256 ? (grid + 7) / 8 * 8 : 256), dim3(512), 0, (hipStream_t)stream, a, grid, HeadsArgs{nullptr, nullptr});
        OMNI_HIP(hipGetLastError());
        return OMNI_OK;
    }
    // (the up-sampling halo is COMPUTED per block — ~700 vector instructions per 2 x 2 cell: blocks of 32 output channels would do it twice — 64 per block here whatever conv_halo_bn says:
    //  de_conv2_0 51 -> 64 us, de_conv3_0 180 -> 240 us with 32, profiles/r06e_halo_bn.txt)
    if (Cout % 64 == 0 && !((fmt & 4) && omni_options().conv_halo_up2_bn_lat == 32)) hipLaunchKernelGGL((conv3x3_halo_sh_kernel<64, 4, true>), dim3(grid * (Cout / 64)), dim3(256), 0, (hipStream_t)stream, a);
    else                hipLaunchKernelGGL((conv3x3_halo_sh_kernel<32, 4, true>), dim3(grid * (Cout / 32)), dim3(256), 0, (hipStream_t)stream, a);
    OMNI_HIP(hipGetLastError());
    return OMNI_OK;
}

// de_conv4_0 + the two heads (model/spherical_model.py:300-307): a = relu(pred(y)) (* sigmoid(weight_pred(y)) when confidence), c = sigmoid(weight_pred(y)),
// y = relu(conv3x3(up2(src)) + bias) with 32 -> 32 channels — y never exists (conv3x3_up2_g1_kernel<HEADS> + heads_finish_kernel, see there).
// src SH [M, P/2, P/2, 32]; wt16 / bias: de_conv4_0's; heads_w16f: 4 KB, the heads' [2][9][32] weights in the fragment order of omni_heads_pack_f16x3;
// scratch: omni_up2_heads_scratch_bytes(M, P) bytes; out_a / out_c planar [M, P, P] (out_c may be NULL).  P % 32 == 0.
// Equal to omni_conv3x3_up2_sh_f16x3 (fp32 output) + omni_heads_f32 up to fp32 summation order (the heads' products run f16x3: ~1e-6 relative).
extern "C" size_t omni_up2_heads_scratch_bytes(int M, int P)
{
    if (M <= 0 || P <= 0 || P % 32) return 0;
    return (size_t)M * (P / 4) * (P / 32) * 4 * 6 * HR_PITCH * sizeof(float);
}
extern "C" int omni_conv3x3_up2_heads_sh_f16x3(const void* src, const void* wt16, const float* bias, const void* heads_w16f, float bias_pred, float bias_weight,
                                               float* scratch, size_t scratch_bytes, float* out_a, float* out_c, int M, int P, int confidence, omni_stream_t stream)
{
    if (!src || !wt16 || !heads_w16f || !scratch || !out_a) OMNI_FAIL(OMNI_ERR_INVALID, "omni_conv3x3_up2_heads_sh: null pointer");
    if (M <= 0 || P <= 0 || P % 32) OMNI_FAIL(OMNI_ERR_INVALID, "omni_conv3x3_up2_heads_sh: the patch size must be a multiple of 32");
    if (scratch_bytes < omni_up2_heads_scratch_bytes(M, P)) OMNI_FAIL(OMNI_ERR_INVALID, "omni_conv3x3_up2_heads_sh: scratch too small (omni_up2_heads_scratch_bytes)");
    if ((long long)M * P * P >= (1ll << 31)) OMNI_FAIL(OMNI_ERR_UNSUPPORTED, "omni_conv3x3_up2_heads_sh: tensor too large for 32-bit indices");
    ShConvArgs a;
    a.src1 = src; a.src2 = nullptr; a.wt = wt16; a.bias = bias; a.res = nullptr; a.dst = nullptr; a.dst_sh = 0; a.res_f32 = 0;
    a.dbg = 0; a.noxcd = omni_options().conv_noxcd; a.wt_major = 0; a.epi_lds = 0;
    a.M = M; a.H = P; a.W = P; a.C1 = 32; a.C2 = 0; a.Cout = 32; a.KH = 3; a.KW = 3; a.stride = 1; a.pad = 1; a.act = OMNI_ACT_RELU;
    a.Ho = P; a.Wo = P; a.rows = M * P * P; a.splitk = 1; a.ws = nullptr; a.post = nullptr; a.post_rows = 1; a.wino_th = a.wino_tw = a.wino_pix = 0;
    const int grid = M * (P / 4) * (P / HT_W);
    hipLaunchKernelGGL(conv3x3_up2_g1_kernel<true>, dim3(grid < 256 ? (grid + 7) / 8 * 8 : 256), dim3(64 * (4 + OMNI_G1_PW)), 0, (hipStream_t)stream, a, grid, HeadsArgs{heads_w16f, scratch});
    OMNI_HIP(hipGetLastError());
    const size_t n = (size_t)M * P * P / 4;
    hipLaunchKernelGGL(heads_finish_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, (const float*)scratch, bias_pred, bias_weight,
                       out_a, out_c, M, P, confidence);
    OMNI_HIP(hipGetLastError());
    return OMNI_OK;
}

// The heads' weights w [2 heads][9 taps][32 channels] (fp32, host or device memory readable by the host — 2.3 KB, packed once per checkpoint) in the
// fragment order of conv3x3_up2_g1_kernel<HEADS>: dst 4 x 64 x 8 halfs = [hi kc0 | hi kc1 | lo kc0 | lo kc1][lane = row + 32 kgroup][8], row r of the
// matrix product = (register group g = r >> 3, lane half hh = (r >> 2) & 1, dx = (r & 3) - 1): (dy, head) pair g + 3 hh; element e of k chunk kc, k group h
// = channel 16 kc + 8 (e >> 2) + 4 h + (e & 3) (the order in which a lane's accumulator quads hold the convolution's output channels).
extern "C" int omni_heads_pack_f16x3(const float* w_host, void* dst_host)
{
    if (!w_host || !dst_host) OMNI_FAIL(OMNI_ERR_INVALID, "omni_heads_pack: null pointer");
    _Float16* o = (_Float16*)dst_host;
    for (int kc = 0; kc < 2; ++kc)
        for (int lane = 0; lane < 64; ++lane)
            for (int e = 0; e < 8; ++e) {
                const int r = lane & 31, h = lane >> 5, g = r >> 3, hh = (r >> 2) & 1, dxi = r & 3;
                float w = 0.0f;
                if (g < 3 && dxi < 3) {
                    const int pair = g + 3 * hh, dy = pair / 2, head = pair % 2, ch = 16 * kc + 8 * (e >> 2) + 4 * h + (e & 3);
                    w = w_host[(head * 9 + dy * 3 + dxi) * 32 + ch];
                }
                const _Float16 hi = (w < 6.103515625e-05f && w > -6.103515625e-05f) ? (_Float16)0.0f : (_Float16)w;
                o[(kc * 64 + lane) * 8 + e] = hi;
                o[((2 + kc) * 64 + lane) * 8 + e] = (_Float16)((w - (float)hi) * 2048.0f);
            }
    return OMNI_OK;
}

// Weights of omni_gemm_rows_sh_f16x3: wt16 [N][K/32][hi32|lo32] (as for omni_conv2d_sh_f16x3_ws) -> fragment order, same size.
extern "C" int omni_gemm_rows_pack(const void* wt16, void* wt16r, int N, int K, omni_stream_t stream)
{
    if (!wt16 || !wt16r || N <= 0 || N % 32 || K <= 0 || K % 32) OMNI_FAIL(OMNI_ERR_INVALID, "omni_gemm_rows_pack: null pointer or N, K not multiples of 32");
    const size_t pieces = (size_t)N * (K / 32) * 8;
    hipLaunchKernelGGL(gemm_rows_pack_kernel, dim3((unsigned)((pieces + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                       (const unsigned char*)wt16, (unsigned char*)wt16r, K / 32, pieces);
    OMNI_HIP(hipGetLastError());
    return OMNI_OK;
}

// out[rows, N] = act(x . W^T + bias + res) for rows <= 32 (see gemm_rows_sh_kernel): x SH [rows, K], wt16r from omni_gemm_rows_pack,
// res fp32 [rows, N] or null, fmt bit 0: dst is SH (else fp32).  K in {512, 2048}, N % 32 == 0.  The K summation order differs from the
// tile kernel's: equal to it up to fp32 rounding, not bit for bit.
extern "C" int omni_gemm_rows_sh_f16x3(const void* x, const void* wt16, const float* bias, const float* res, void* dst, int fmt,
                                       int rows, int K, int N, int act, omni_stream_t stream)
{
    if (!x || !wt16 || !dst) OMNI_FAIL(OMNI_ERR_INVALID, "omni_gemm_rows_sh: null pointer");
    if (rows <= 0 || rows > 32 || N <= 0 || N % 32) OMNI_FAIL(OMNI_ERR_INVALID, "omni_gemm_rows_sh: 1..32 rows, N a multiple of 32");
    if (K != 512 && K != 2048) OMNI_FAIL(OMNI_ERR_UNSUPPORTED, "omni_gemm_rows_sh: K must be 512 or 2048");
    RowsGemmArgs a;
    a.x = x; a.wt = wt16; a.bias = bias; a.res = res; a.dst = dst; a.rows = rows; a.K = K; a.N = N; a.act = act; a.dst_sh = fmt & 1;
    a.parts = nullptr; a.nparts = 0; a.pbias = nullptr; a.pres = nullptr; a.xout = nullptr;
    if (K == 512) hipLaunchKernelGGL(gemm_rows_sh_kernel<2>, dim3(N / 32), dim3(512), 0, (hipStream_t)stream, a);
    else          hipLaunchKernelGGL(gemm_rows_sh_kernel<8>, dim3(N / 32), dim3(512), 0, (hipStream_t)stream, a);
    OMNI_HIP(hipGetLastError());
    return OMNI_OK;
}

// LayerNorm over 512 channels (weight lg, bias lb, eps) of x fp32 [rows, 512], then omni_gemm_rows_sh_f16x3 on the result, in one launch
// (gemm_rows_ln_sh_kernel): the same bits as omni_layernorm512_sh followed by omni_gemm_rows_sh_f16x3.  rows <= 32, K = 512.
extern "C" int omni_gemm_rows_ln_sh_f16x3(const float* x, const float* lg, const float* lb, float eps, const void* wt16, const float* bias,
                                          const float* res, void* dst, int fmt, int rows, int N, int act, omni_stream_t stream)
{
    if (!x || !lg || !lb || !wt16 || !dst) OMNI_FAIL(OMNI_ERR_INVALID, "omni_gemm_rows_ln_sh: null pointer");
    if (rows <= 0 || rows > 32 || N <= 0 || N % 32) OMNI_FAIL(OMNI_ERR_INVALID, "omni_gemm_rows_ln_sh: 1..32 rows, N a multiple of 32");
    RowsGemmArgs a;
    a.x = x; a.wt = wt16; a.bias = bias; a.res = res; a.dst = dst; a.rows = rows; a.K = 512; a.N = N; a.act = act; a.dst_sh = fmt & 1;
    a.parts = nullptr; a.nparts = 0; a.pbias = nullptr; a.pres = nullptr; a.xout = nullptr;
    hipLaunchKernelGGL(gemm_rows_ln_sh_kernel, dim3(N / 32), dim3(512), 0, (hipStream_t)stream, a, lg, lb, eps);
    OMNI_HIP(hipGetLastError());
    return OMNI_OK;
}

// The K = 2048 rows GEMM (a lone panorama's fc2) as `slices` K slices over blockIdx.y: N / 32 x slices blocks instead of N / 32 — 16 blocks stream
// the 4 MB of fc2's weights in ~12 us, 64 in a third of that — each writing its RAW partial sums to parts[slice][rows][N] (fp32, no bias / residual).
// The consumer sums them: omni_gemm_rows_ln_parts_sh_f16x3 (the next block's norm1 + qkv) or omni_splitk_reduce_ln512 (encoder_norm).
// slices in {1, 2, 4} (K / 32 / slices / 8 K-steps per wave).
extern "C" int omni_gemm_rows_slices_sh_f16x3(const void* x, const void* wt16, float* parts, int rows, int K, int N, int slices, omni_stream_t stream)
{
    if (!x || !wt16 || !parts) OMNI_FAIL(OMNI_ERR_INVALID, "omni_gemm_rows_slices_sh: null pointer");
    if (rows <= 0 || rows > 32 || N <= 0 || N % 32) OMNI_FAIL(OMNI_ERR_INVALID, "omni_gemm_rows_slices_sh: 1..32 rows, N a multiple of 32");
    if (K != 2048 || (slices != 1 && slices != 2 && slices != 4)) OMNI_FAIL(OMNI_ERR_UNSUPPORTED, "omni_gemm_rows_slices_sh: K = 2048 in 1, 2 or 4 slices");
    RowsGemmArgs a;
    a.x = x; a.wt = wt16; a.bias = nullptr; a.res = nullptr; a.dst = nullptr; a.rows = rows; a.K = K; a.N = N; a.act = OMNI_ACT_NONE; a.dst_sh = 0;
    a.parts = parts; a.nparts = slices; a.pbias = nullptr; a.pres = nullptr; a.xout = nullptr;
    const dim3 grid(N / 32, slices);
    if (slices == 4)      hipLaunchKernelGGL(gemm_rows_sh_kernel<2>, grid, dim3(512), 0, (hipStream_t)stream, a);
    else if (slices == 2) hipLaunchKernelGGL(gemm_rows_sh_kernel<4>, grid, dim3(512), 0, (hipStream_t)stream, a);
    else                  hipLaunchKernelGGL(gemm_rows_sh_kernel<8>, grid, dim3(512), 0, (hipStream_t)stream, a);
    OMNI_HIP(hipGetLastError());
    return OMNI_OK;
}

// omni_gemm_rows_ln_sh_f16x3 whose input is the previous GEMM's K slices: x = sum_s parts[s] + pbias + pres (fp32 [rows,512]; pbias / pres may be null),
// written to xout (the next residual) by one block; then LayerNorm + the rows GEMM as before.
extern "C" int omni_gemm_rows_ln_parts_sh_f16x3(const float* parts, int nparts, const float* pbias, const float* pres, float* xout,
                                                const float* lg, const float* lb, float eps, const void* wt16, const float* bias,
                                                void* dst, int fmt, int rows, int N, int act, omni_stream_t stream)
{
    if (!parts || nparts < 1 || nparts > 8 || !xout || !lg || !lb || !wt16 || !dst) OMNI_FAIL(OMNI_ERR_INVALID, "omni_gemm_rows_ln_parts_sh: null pointer or 1..8 slices");
    if (rows <= 0 || rows > 32 || N <= 0 || N % 32) OMNI_FAIL(OMNI_ERR_INVALID, "omni_gemm_rows_ln_parts_sh: 1..32 rows, N a multiple of 32");
    RowsGemmArgs a;
    a.x = nullptr; a.wt = wt16; a.bias = bias; a.res = nullptr; a.dst = dst; a.rows = rows; a.K = 512; a.N = N; a.act = act; a.dst_sh = fmt & 1;
    a.parts = const_cast<float*>(parts); a.nparts = nparts; a.pbias = pbias; a.pres = pres; a.xout = xout;
    hipLaunchKernelGGL(gemm_rows_ln_sh_kernel, dim3(N / 32), dim3(512), 0, (hipStream_t)stream, a, lg, lb, eps);
    OMNI_HIP(hipGetLastError());
    return OMNI_OK;
}

// tok [rows,512] = sum_s parts[s] + bias + res, y = LayerNorm(tok) as SH (fmt bit 0) or fp32: the second pass of a split-K / K-sliced GEMM with 512
// columns on its own (sh_splitk_reduce_ln512_kernel; omni_gemm_sh_f16x3_ln512_ws runs it behind its GEMM).
extern "C" int omni_splitk_reduce_ln512(const float* parts, int nparts, const float* bias, const float* res, float* tok, const float* lg, const float* lb,
                                        float eps, void* y, int fmt, int rows, omni_stream_t stream)
{
    if (!parts || nparts < 1 || !tok || !lg || !lb || !y || rows <= 0) OMNI_FAIL(OMNI_ERR_INVALID, "omni_splitk_reduce_ln512: null pointer or empty input");
    hipStream_t s = (hipStream_t)stream;
    if (fmt & 1) hipLaunchKernelGGL(sh_splitk_reduce_ln512_kernel<true>, dim3((rows + 3) / 4), dim3(256), 0, s, parts, bias, res, tok, lg, lb, y, rows, nparts, (size_t)rows * 512, eps);
    else         hipLaunchKernelGGL(sh_splitk_reduce_ln512_kernel<false>, dim3((rows + 3) / 4), dim3(256), 0, s, parts, bias, res, tok, lg, lb, y, rows, nparts, (size_t)rows * 512, eps);
    OMNI_HIP(hipGetLastError());
    return OMNI_OK;
}

// conv1 7x7 s2 p3 (3 -> 64) + bn1 + ReLU on the fp16 matrix cores.  src planar [M,3,P,P]; wt16: the folded filter bank as
// [64][192] with k = (c*7 + ky)*8 + kx (kx = 7 and k >= 168: zeros), split like every other f16x3 weight matrix
// ([64][6][hi32|lo32]); dst SH [M,P/2,P/2,64].
extern "C" int omni_stem_sh_f16x3(const float* src, const void* wt16, const float* bias, void* dst, int M, int P, omni_stream_t stream)
{
    if (!src || !wt16 || !dst) OMNI_FAIL(OMNI_ERR_INVALID, "omni_stem: null pointer");
    if (P % 32 || M <= 0) OMNI_FAIL(OMNI_ERR_INVALID, "omni_stem_sh_f16x3: patch size must be a multiple of 32");
    const int Po = P / 2;
    const int strips = M * (Po / SM_TH);
    const int split = (strips < 256 && Po % (4 * SM_TW) == 0) ? 4 : (strips < 512 && Po % (2 * SM_TW) == 0) ? 2 : 1;    // same bits either way
    if (omni_options().conv_stem_pc) {
        // tiles per block: the column range of the split above where the launch is small; one block per CU walking ntiles / CUs tiles where it is not
        const int tps = Po / SM_TW, ntiles = strips * tps;
        int tpb = tps / split;
        const int ncu = omni_num_cus();
        if (split == 1 && ntiles > ncu * tps) tpb = (ntiles + ncu - 1) / ncu;
        hipLaunchKernelGGL(stem_f16x3_pc_kernel, dim3((unsigned)((ntiles + tpb - 1) / tpb)), dim3(768), 0, (hipStream_t)stream, src, wt16, bias, dst, M, P, Po, omni_options().conv_epi_lds, tpb);
    }
    else hipLaunchKernelGGL(stem_f16x3_kernel, dim3(strips, split), dim3(256), 0, (hipStream_t)stream, src, wt16, bias, dst, M, P, Po, omni_options().conv_epi_lds);
    OMNI_HIP(hipGetLastError());
    return OMNI_OK;
}

// layout conversions (n = number of elements, a multiple of 32 channels per pixel)
OMNI_SH_OVERFLOW_ACCESSOR(omni_sh_overflow_conv)          // this translation unit's copy of the sticky range flag
int omni_sh_overflow_net(unsigned* out, int reset);      // omni_net.hip's

extern "C" int omni_sh_overflow(int* flag, int reset)
{
    if (!flag) OMNI_FAIL(OMNI_ERR_INVALID, "omni_sh_overflow: null output");
    OMNI_HIP(hipDeviceSynchronize());                    // diagnostic entry point, never on the hot path
    unsigned v = 0;
    if (omni_sh_overflow_conv(&v, reset) != 0 || omni_sh_overflow_net(&v, reset) != 0)
        OMNI_FAIL(OMNI_ERR_HIP, "omni_sh_overflow: could not read the device flag");
    *flag = v ? 1 : 0;
    return OMNI_OK;
}

extern "C" int omni_sh_from_f32(const float* src, void* dst, size_t n, omni_stream_t stream)
{
    if (!src || !dst || n % 32) OMNI_FAIL(OMNI_ERR_INVALID, "omni_sh_from_f32: null pointer or n % 32 != 0");
    if (n == 0) return OMNI_OK;
    hipLaunchKernelGGL(sh_from_f32_kernel, dim3((unsigned)((n / 4 + 255) / 256)), dim3(256), 0, (hipStream_t)stream, src, dst, n / 4);
    OMNI_HIP(hipGetLastError());
    return OMNI_OK;
}
extern "C" int omni_sh_to_f32(const void* src, float* dst, size_t n, omni_stream_t stream)
{
    if (!src || !dst || n % 32) OMNI_FAIL(OMNI_ERR_INVALID, "omni_sh_to_f32: null pointer or n % 32 != 0");
    if (n == 0) return OMNI_OK;
    hipLaunchKernelGGL(sh_to_f32_kernel, dim3((unsigned)((n / 4 + 255) / 256)), dim3(256), 0, (hipStream_t)stream, src, dst, n / 4);
    OMNI_HIP(hipGetLastError());
    return OMNI_OK;
}
