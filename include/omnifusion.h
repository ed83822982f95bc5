/*
 * omnifusion.h — C ABI of libomnifusion_hip.so (MI355X / gfx950).
 *
 * Drop-in boundary for the reference's `equi_pers` hot path.  Every entry point is
 * `extern "C"`, takes plain device pointers + sizes + a hipStream_t passed as void*,
 * allocates nothing the caller can see, never synchronises the host and returns an
 * int status (OMNI_OK == 0); omni_last_error() returns the thread-local message.
 * The reference has no FFI of its own (it is pure Python on stock PyTorch ops), so
 * each function cites the reference *Python* interface it replaces.
 *
 * Tensor layouts (all dense, row-major, last index fastest):
 *   ERP image          [B, C, H, W]
 *   patches, reference [B, C, ph, pw, N]   OMNI_LAYOUT_BCHWN  (N innermost,
 *                      equi2pers_v3.py:112-113; what the public Python API returns)
 *   patches, planar    [B, N, C, ph, pw]   OMNI_LAYOUT_BNCHW  (patch-major; what the
 *                      model uses internally so that no N-innermost tensor is ever
 *                      materialised between the sampler, the network and the blender)
 *   patches, NHWC      [B, N, ph, pw, C]   OMNI_LAYOUT_BNHWC  (network activations)
 */
#ifndef OMNIFUSION_H_
#define OMNIFUSION_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define OMNI_VERSION 200

enum omni_status {
    OMNI_OK = 0,
    OMNI_ERR_INVALID = 1,   /* bad nrows / shape / layout / null pointer  -> Python ValueError  */
    OMNI_ERR_HIP = 2,       /* a HIP runtime call failed                  -> Python RuntimeError */
    OMNI_ERR_UNSUPPORTED = 3
};

enum omni_layout { OMNI_LAYOUT_BCHWN = 0, OMNI_LAYOUT_BNCHW = 1, OMNI_LAYOUT_BNHWC = 2 };
enum omni_dtype { OMNI_F32 = 0, OMNI_F16 = 1 };

typedef void* omni_stream_t;            /* hipStream_t */
typedef struct omni_geometry omni_geometry_t;

int omni_version(void);
const char* omni_last_error(void);

/* Tuning options (tile shapes, kernel selection, cache size — csrc/omni_internal.h `OmniOptions`; each also has an
 * OMNI_<NAME> environment variable that is read ONCE, when the library first needs it).  No option changes a result
 * bit except "splitk_max" (K summation order).  Unknown name -> OMNI_ERR_INVALID.  No reference counterpart (the
 * reference has no tunables below PyTorch); used by tests/ and tools/ to sweep kernel variants. */
int omni_set_option(const char* name, int value);
int omni_get_option(const char* name, int* value);

/* Number of patches for an nrows preset (3,4,5,6 -> 10,18,26,46), -1 otherwise.
 * equi2pers_v3.py:32-47 / pers2equi_v3.py:36-51 (the reference raises
 * UnboundLocalError for any other value). */
int omni_num_patches(int nrows);

/* Patch centres in [-1,1] (`center_p`, equi2pers_v3.py:81-82) written to HOST memory
 * center_p[N*2]; `which` = 0 equi2pers table, 1 pers2equi table (nrows=3 differs). */
int omni_patch_centers(int nrows, int which, float* center_p_host);

/*
 * Geometry handle: the constants of one (nrows, fov, patch size, ERP size) configuration
 * resident on the current device — patch-centre trig, per-row/column trig of the ERP
 * grid and the per-tile candidate-patch masks of pers2equi.  A few KB; replaces the
 * reference's per-call CPU grid build + H2D (equi2pers_v3.py:24-109) and its 0.5 GB
 * ./grid/<layer_name>.pth tables (pers2equi_v3.py:24-29,155-167).  The convenience
 * entry points below look handles up in an internal per-device, mutex-protected cache,
 * so callers that mirror the reference's free functions never see this type.
 */
int omni_geometry_create(omni_geometry_t** out, int nrows, float fov_h, float fov_w,
                         int ph, int pw, int H, int W, omni_stream_t stream);
void omni_geometry_destroy(omni_geometry_t* g);
/* The internal cache is an LRU capped at option "geom_cache_max" (default 16) handles; evicted / cleared handles are
 * destroyed only after their device has drained.  omni_geometry_cache_size: handles currently cached (all devices). */
void omni_geometry_cache_clear(void);
int omni_geometry_cache_size(void);

/*
 * equi2pers — replaces equi_pers/equi2pers_v3.py:20 `equi2pers(erp_img, fov, nrows, patch_size)`
 * (grid build :24-109, F.grid_sample bilinear/border/align_corners=True :111, unfold+reshape
 * :112-113).  erp [B,C,H,W] -> pers in `layout`.  dtype: OMNI_F32 or OMNI_F16 storage
 * (arithmetic is fp32 either way).
 */
int omni_equi2pers(const void* erp, void* pers, int dtype, int B, int C, int H, int W,
                   int ph, int pw, int nrows, float fov_h, float fov_w, int layout,
                   omni_stream_t stream);

/* The other three return values of equi2pers (equi2pers_v3.py:115-122): xyz [N,3,ph,pw]
 * unit rays from the UNWRAPPED lon/lat, uv [N,2,ph,pw] (the reference's scrambled strip
 * reshape), both fp32 device buffers; either may be NULL. */
int omni_equi2pers_aux(float* xyz, float* uv, int ph, int pw, int nrows, float fov_h, float fov_w,
                       omni_stream_t stream);

/*
 * pers2equi — replaces equi_pers/pers2equi_v3.py:16
 * `pers2equi(pers_img, fov, nrows, patch_size, erp_size, layer_name)` (tables :109-152,
 * gathers :174-177, mask/threshold/L1-normalise/blend :179-196).  pers in `layout` -> erp
 * [B,C,H,W].  `layer_name` was only the reference's cache-file key and has no counterpart.
 */
int omni_pers2equi(const void* pers, void* erp, int dtype, int B, int C, int ph, int pw,
                   int H, int W, int nrows, float fov_h, float fov_w, int layout,
                   omni_stream_t stream);

/* [B,C,ph,pw,N] (OMNI_LAYOUT_BCHWN, the reference's N-innermost patch tensor, equi2pers_v3.py:112-113) -> [B,N,C,ph,pw]
 * (OMNI_LAYOUT_BNCHW), one coalesced pass.  The blend on planar patches is 5x faster than on the N-innermost tensor (every cache
 * line of which is shared by all N patches): conversion + omni_pers2equi(..., OMNI_LAYOUT_BNCHW) is what the drop-in pers2equi()
 * runs for reference-layout inputs (same result bits as omni_pers2equi(..., OMNI_LAYOUT_BCHWN)). */
int omni_patches_to_planar(const void* src, void* dst, int dtype, int B, int C, int ph, int pw, int N, omni_stream_t stream);

/*
 * Fused confidence blend — replaces model/spherical_model.py:307-311 (two pers2equi calls +
 * zero-safe division):  out = P(pred_w) / (P(conf) + 1e-8*[P(conf) <= 1e-8]),
 * pred_w = relu(pred)*sigmoid(weight_pred), conf = sigmoid(weight_pred), both C=1 patch
 * tensors in `layout`; out [B,1,H,W] fp32.
 */
int omni_pers2equi_conf(const void* pred_w, const void* conf, float* out, int dtype, int B,
                        int ph, int pw, int H, int W, int nrows, float fov_h, float fov_w,
                        int layout, omni_stream_t stream);

/* Explicit-handle forms of the three operators (same semantics, no cache lookup). */
int omni_equi2pers_g(const omni_geometry_t* g, const void* erp, void* pers, int dtype, int B, int C,
                     int layout, omni_stream_t stream);
int omni_pers2equi_g(const omni_geometry_t* g, const void* pers, void* erp, int dtype, int B, int C,
                     int layout, omni_stream_t stream);

/* ------------------------------------------------------------------------------------------------
 * Network operators (rows a3-a8 of SURVEY.md 8a).  Activations are NHWC fp32 over M = B*N patches
 * ([M,H,W,C]); eval-mode BatchNorm is folded into the weights by the caller at load time.  The
 * reference runs these as Conv3d(k,k,1)/BatchNorm3d/Linear modules over [B,C,P,P,N] tensors.
 * ---------------------------------------------------------------------------------------------- */
enum omni_act { OMNI_ACT_NONE = 0, OMNI_ACT_RELU = 1, OMNI_ACT_GELU = 2 };

/* Backward of the two operators: the vector-Jacobian products that autograd derives from
 * F.grid_sample (equi_pers/equi2pers_v3.py:111) and from the indexing gathers of equi_pers/pers2equi_v3.py:174-196 in the
 * reference's training scripts (train_erp_depth.py:255-300).  Both operators are linear in the image.  fp32 only; the output
 * gradient is overwritten; layouts as in the forward (grad_pers: OMNI_LAYOUT_BCHWN or _BNCHW).  No atomics: the result of a call is
 * a function of its inputs and the geometry handle only (fixed summation order).  The first backward of a geometry builds its tables
 * (synchronises the stream once); the first call of a (geometry, stream, B * C) allocates scratch the size of the incoming gradient
 * — OMNI_ERR_UNSUPPORTED if that first call happens while the stream is being captured (run the shape once before capturing). */
int omni_equi2pers_bwd(const void* grad_pers, void* grad_erp, int dtype, int B, int C, int H, int W,
                       int ph, int pw, int nrows, float fov_h, float fov_w, int layout, omni_stream_t stream);
int omni_pers2equi_bwd(const void* grad_erp, void* grad_pers, int dtype, int B, int C, int ph, int pw,
                       int H, int W, int nrows, float fov_h, float fov_w, int layout, omni_stream_t stream);
/* dst[M,Ho,Wo,Cout] = act(conv(src1 ++ src2 (channel concat), wt) + bias + res) on the fp32 matrix cores
 * (v_mfma_f32_32x32x2_f32).  wt: [Cout][KH*KW*(C1+C2)], k ordered (ky,kx,c).  C1,C2,Cout multiples of 32.
 * Replaces: encoder BasicBlock convs model/spherical_model.py:122-167,257-261 (residual+ReLU fused), decoder
 * ConvBnReLU_v2 :29-37,274-302 incl. torch.cat :275,282,289,296, `down` :211,263, and every nn.Linear of
 * model/blocks.py:19-21,43-46 (H=W=KH=KW=1; GELU of :20 and the residual adds of :86-87 fused). */
int omni_conv2d_nhwc_f32(const float* src1, const float* src2, const float* wt, const float* bias,
                         const float* res, float* dst, int M, int H, int W, int C1, int C2, int Cout,
                         int KH, int KW, int stride, int pad, int act, omni_stream_t stream);
/* Same, split along K into `splitk` ranges (workspace ws: splitk*rows*Cout floats) and reduced by a deterministic
 * second pass: for problems too small to fill the 256 CUs (layer4, de_conv0_x, the transformer's nn.Linear layers at
 * B*N rows).  omni_conv2d_splitk_plan(rows, Cout, ksteps) returns the factor the library recommends for `rows` output
 * pixels (ksteps = KH*KW*(C1+C2)/32); plan with a nominal batch to keep results bit-identical across batch sizes. */
int omni_conv2d_splitk_plan(long long rows, int Cout, int ksteps);
int omni_conv2d_nhwc_f32_ws(const float* src1, const float* src2, const float* wt, const float* bias,
                            const float* res, float* dst, int M, int H, int W, int C1, int C2, int Cout,
                            int KH, int KW, int stride, int pad, int act, int splitk, float* ws, size_t ws_bytes,
                            omni_stream_t stream);
/* The same operator with the products on the fp16 matrix cores at fp32-class accuracy ("f16x3"): every operand is split
 * into a pair of halfs x = hi + lo*2^-11 and a product block is three v_mfma_f32_32x32x16_f16 (3/16 of the fp32-MFMA
 * time, max error ~3e-6).  Activations, bias, residual and output stay fp32 NHWC (the A tile is split on its way into
 * LDS); wt16 is the weight matrix split once at load time: halfs [Cout][KH*KW*(C1+C2)/32][hi32|lo32],
 * hi = fp16(w) (0 below 2^-14), lo = fp16((w - hi) * 2048). */
int omni_conv2d_nhwc_f16x3_ws(const float* src1, const float* src2, const void* wt16, const float* bias,
                              const float* res, float* dst, int M, int H, int W, int C1, int C2, int Cout,
                              int KH, int KW, int stride, int pad, int act, int splitk, float* ws, size_t ws_bytes,
                              omni_stream_t stream);
/* The same operator with SPLIT-HALF ("SH") activations: per pixel and per group of 32 channels a tensor stores 32 hi
 * halfs then 32 lo halfs (x = hi + lo*2^-11; 128 bytes per group = the footprint of 32 floats).  The producer splits
 * once, consumers stream tiles straight into LDS by LDS-DMA.  src1 and src2 are SH tensors; fmt bit 0: dst is SH
 * (else fp32 NHWC), fmt bit 1: res is fp32 NHWC (else SH), fmt bit 2: latency form — the caller runs so few images (one panorama) that the
 * kernel choice for 16-pixel-wide images should stay with the im2col tiles (results equal up to the K summation order) and the epilogue with
 * the direct 8-byte stores (same bits); wt16 as above.
 * omni_sh_from_f32 / omni_sh_to_f32 convert n elements (n % 32 == 0). */
int omni_conv2d_sh_f16x3_ws(const void* src1, const void* src2, const void* wt16, const float* bias,
                            const void* res, void* dst, int fmt, int M, int H, int W, int C1, int C2, int Cout,
                            int KH, int KW, int stride, int pad, int act, int splitk, float* ws, size_t ws_bytes,
                            omni_stream_t stream);
/* The nn.Linear layers of ONE panorama's transformer (model/blocks.py:19-21,43-46 at 18 tokens): out[rows, N] =
 * act(x . W^T + bias + res) for rows <= 32.  x SH [rows, K], res fp32 [rows, N] or null, fmt bit 0: dst is SH
 * (else fp32).  K in {512, 2048}, N % 32 == 0.  Operands go straight into registers, K is split over the waves of a block and
 * summed in a fixed order: equal to omni_conv2d_sh_f16x3_ws (H = W = KH = KW = 1) up to fp32 rounding, not bit for bit.
 * wt16r is the weight matrix in FRAGMENT ORDER (a wave's load = one contiguous KiB): omni_gemm_rows_pack makes it, once at load
 * time, from wt16 [N][K/32][hi32|lo32] (same size). */
int omni_gemm_rows_sh_f16x3(const void* x, const void* wt16r, const float* bias, const float* res, void* dst, int fmt,
                            int rows, int K, int N, int act, omni_stream_t stream);
int omni_gemm_rows_pack(const void* wt16, void* wt16r, int N, int K, omni_stream_t stream);
/* LayerNorm (model/blocks.py:69,75: nn.LayerNorm(512), eps as given) + the following nn.Linear of one panorama's transformer in ONE launch:
 * x fp32 [rows <= 32, 512] is normalised by every block itself (the arithmetic of omni_layernorm512_sh), then multiplied as in
 * omni_gemm_rows_sh_f16x3 (K = 512).  Same bits as the two calls. */
int omni_gemm_rows_ln_sh_f16x3(const float* x, const float* ln_weight, const float* ln_bias, float eps, const void* wt16r, const float* bias,
                               const float* res, void* dst, int fmt, int rows, int N, int act, omni_stream_t stream);
/* F.interpolate(scale_factor 2, bilinear, align_corners=False) + ConvBnReLU (3x3, pad 1) of the decoder, model/spherical_model.py:
 * 279-301, in ONE kernel: the up-sampled tensor never exists.  src SH [M,Hl,Wl,C], dst [M,2Hl,2Wl,Cout] SH (fmt bit 0) or fp32;
 * 2Wl % 32 == 0 and 2Hl % 4 == 0 (else OMNI_ERR_UNSUPPORTED: omni_upsample_bilinear_sh + omni_conv2d_sh_f16x3_ws give the same bits). */
int omni_conv3x3_up2_sh_f16x3(const void* src, const void* wt16, const float* bias, void* dst, int fmt,
                              int M, int Hl, int Wl, int C, int Cout, int act, omni_stream_t stream);
/* de_conv4_0 AND the two heads in one pass over the widest tensor of the network (model/spherical_model.py:300-307: F.interpolate + ConvBnReLU 32 -> 32,
 * then pred (ReLU) / weight_pred (sigmoid), 3x3, 32 -> 1, and their product): a = relu(pred(y)) * (confidence ? sigmoid(weight_pred(y)) : 1),
 * c = sigmoid(weight_pred(y)); y = de_conv4_0's output never exists.  src SH [M,P/2,P/2,32]; heads_w16f from omni_heads_pack_f16x3; scratch of
 * omni_up2_heads_scratch_bytes(M, P) bytes; out_a / out_c planar [M,P,P] (out_c may be NULL).  Equal to omni_conv3x3_up2_sh_f16x3 + omni_heads_f32 up to
 * the fp32 summation order of the heads (whose products run f16x3 here).  Deterministic. */
size_t omni_up2_heads_scratch_bytes(int M, int P);
int omni_conv3x3_up2_heads_sh_f16x3(const void* src, const void* wt16, const float* bias, const void* heads_w16f, float bias_pred, float bias_weight,
                                    float* scratch, size_t scratch_bytes, float* out_a, float* out_c, int M, int P, int confidence, omni_stream_t stream);
/* w [2][9][32] fp32 (HOST memory) -> the 4 KB fragment-ordered f16x3 operand of omni_conv3x3_up2_heads_sh_f16x3 (HOST memory; copy it to the device). */
int omni_heads_pack_f16x3(const float* w_host, void* dst_host);
/* omni_conv2d_sh_f16x3_ws with `post` (fp32, post_elems = whole rows of Cout channels) added AFTER the activation, output row index
 * modulo its row count: `layer1 + point_feat` (model/spherical_model.py:258; point_feat [N,h,w,64] broadcast over the batch, or
 * full size) inside layer1's last convolution.  post = NULL: the plain operator.  Not combinable with split-K. */
int omni_conv2d_sh_f16x3_post_ws(const void* src1, const void* src2, const void* wt16, const float* bias,
                                 const void* res, void* dst, int fmt, int M, int H, int W, int C1, int C2, int Cout,
                                 int KH, int KW, int stride, int pad, int act, int splitk, float* ws, size_t ws_bytes,
                                 const float* post, size_t post_elems, omni_stream_t stream);
/* EXPERIMENTAL (round 6): Winograd F(2x2, 3x3) for the 3x3 stride-1 pad-1 convolutions of small images (model/spherical_model.py:122-143, layer3 / layer4 /
 * de_conv0: Conv3d (3,3,1) + BatchNorm3d + ReLU) — 16 matrix products per 2 x 2 output tile instead of 36.  omni_wino_input_sh: x SH [M,H,W,C] (H, W even) ->
 * V SH [16][M*H/2*W/2][C] = B^T d B per tile.  omni_conv3x3_wino_sh_f16x3: dst = act(conv3x3(x) + bias + res) from V and wt16 = the f16x3 split of
 * [Cout][16*C] (k = p*C + c) holding (G g G^T)[p]; multiply stage and output transform in one kernel (conv_sh_kernel<.., WINO>); splitk in {1,2,4} divides the
 * sixteen positions (ws: splitk * M*H*W * Cout floats).  Equal to omni_conv2d_sh_f16x3_ws up to rounding, not bit for bit. */
int omni_wino_input_sh(const void* src, void* V, int M, int H, int W, int C, omni_stream_t stream);
int omni_conv3x3_wino_sh_f16x3(const void* V, const void* wt16, const float* bias, const void* res, void* dst, int fmt,
                               int M, int H, int W, int C, int Cout, int act, int splitk, float* ws, size_t ws_bytes, omni_stream_t stream);
int omni_sh_from_f32(const float* src, void* dst, size_t n, omni_stream_t stream);
int omni_sh_to_f32(const void* src, float* dst, size_t n, omni_stream_t stream);
/* Range guard of the SH format: values with |x| > 65504 (the fp16 range) are SATURATED when an activation is split, and a
 * sticky per-device flag records it.  *flag = 1 if that happened on the current device since the flag was last cleared
 * (reset != 0 clears it).  Synchronises the device: diagnostic only.  The fp32 reference has no such limit
 * (model/spherical_model.py runs fp32 end to end): a checkpoint that trips the flag needs OMNI_NET_PRECISION=fp32. */
int omni_sh_overflow(int* flag, int reset);
/* conv1 7x7 s2 p3 (3->64) + bn1 + ReLU, model/spherical_model.py:254.  src planar [M,3,P,P]
 * (OMNI_LAYOUT_BNCHW patches), wt [147][64] (k = (ky*7+kx)*3+c), dst NHWC [M,P/2,P/2,64]. */
int omni_stem_f32(const float* src, const float* wt, const float* bias, float* dst, int M, int P, omni_stream_t stream);
/* F.max_pool3d((3,3,1),(2,2,1),(1,1,0)), model/spherical_model.py:255 */
int omni_maxpool3x3s2_f32(const float* src, float* dst, int M, int H, int W, int C, omni_stream_t stream);
/* F.interpolate(bilinear, align_corners=False), model/spherical_model.py:271,279,286,293,300 */
int omni_upsample_bilinear_f32(const float* src, float* dst, int M, int H, int W, int C, int Ho, int Wo, omni_stream_t stream);
/* x[m,hw,c] += y[m,c]: the transformer token added as a channel bias, model/spherical_model.py:267-268 */
int omni_add_hw_f32(float* x, const float* y, int M, int HW, int C, omni_stream_t stream);
/* x[i] += y[i % period]: layer1 + point_feat, model/spherical_model.py:258 */
int omni_add_period_f32(float* x, const float* y, size_t total, size_t period, omni_stream_t stream);
/* The element-wise operators above on split-half (SH) activations (see omni_conv2d_sh_f16x3_ws): same arguments,
 * activation tensors in the SH layout (C % 32 == 0); the broadcast operand y and the stem's planar input stay fp32. */
int omni_stem_sh(const float* src, const float* wt, const float* bias, void* dst, int M, int P, omni_stream_t stream);
/* the stem as an implicit GEMM on the fp16 matrix cores (f16x3): wt16 = the folded filter bank [64][192], k = (c*7+ky)*8+kx
 * (kx = 7 and k >= 168 zero), split into [64][6][hi32|lo32]; P % 32 == 0 */
int omni_stem_sh_f16x3(const float* src, const void* wt16, const float* bias, void* dst, int M, int P, omni_stream_t stream);
int omni_maxpool3x3s2_sh(const void* src, void* dst, int M, int H, int W, int C, omni_stream_t stream);
int omni_upsample_bilinear_sh(const void* src, void* dst, int M, int H, int W, int C, int Ho, int Wo, omni_stream_t stream);
int omni_add_hw_sh(void* x, const float* y, int M, int HW, int C, omni_stream_t stream);
int omni_add_period_sh(void* x, const float* y, size_t total, size_t period, omni_stream_t stream);
/* nn.LayerNorm(512) with an SH result, and the attention core on a fused q|k|v projection [B*N, 1536] (q at column 0,
 * k at 512, v at 1024; heads of 128) with an SH result: the f16x3 GEMMs of the transformer consume them directly. */
int omni_layernorm512_sh(const float* x, const float* g, const float* b, void* y, int rows, float eps, omni_stream_t stream);
/* A lone panorama's fc2 (K = 2048, <= 32 rows) in K slices: N / 32 x slices blocks stream the weights instead of N / 32; each slice's RAW partial sums go to
 * parts[slice][rows][N] (fp32).  The consumer adds them up: omni_gemm_rows_ln_parts_sh_f16x3 (x = sum parts + pbias + pres -> xout, then LayerNorm + the rows
 * GEMM of omni_gemm_rows_ln_sh_f16x3: the next Transformer_Block's norm1 + qkv, model/blocks.py:83-88) or omni_splitk_reduce_ln512 (encoder_norm,
 * model/spherical_model.py:186).  slices in {1, 2, 4}. */
int omni_gemm_rows_slices_sh_f16x3(const void* x, const void* wt16r, float* parts, int rows, int K, int N, int slices, omni_stream_t stream);
int omni_gemm_rows_ln_parts_sh_f16x3(const float* parts, int nparts, const float* pbias, const float* pres, float* xout,
                                     const float* ln_weight, const float* ln_bias, float eps, const void* wt16r, const float* bias,
                                     void* dst, int fmt, int rows, int N, int act, omni_stream_t stream);
int omni_splitk_reduce_ln512(const float* parts, int nparts, const float* bias, const float* res, float* tok, const float* ln_weight, const float* ln_bias,
                             float eps, void* y, int fmt, int rows, omni_stream_t stream);
/* x = x + mlp.fc2(h) FOLLOWED BY the next LayerNorm (model/blocks.py:83-88 into the next block's norm1, or into encoder_norm,
 * model/spherical_model.py:180-187) as a split-K GEMM whose second pass normalises: tok [rows,512] fp32 = x . wt16^T + bias + res (res fp32),
 * y = LayerNorm(tok) as SH (fmt bit 0) or fp32.  splitk >= 2; ws as for omni_conv2d_sh_f16x3_ws.  The bits of the separate calls, one launch fewer. */
int omni_gemm_sh_f16x3_ln512_ws(const void* x, const void* wt16, const float* bias, const float* res, float* tok, const float* ln_g, const float* ln_b,
                                float eps, void* y, int fmt, int rows, int K, int splitk, float* ws, size_t ws_bytes, omni_stream_t stream);
int omni_attention_qkv_sh(const float* qkv, void* out, int B, int N, omni_stream_t stream);
/* tokens: reshape(bs,-1,N).transpose(1,2) of the `down` output + pos_emb, model/spherical_model.py:264,181 */
int omni_token_pack_f32(const float* d, const float* pos, float* tok, int M, int N, int HW, int C, omni_stream_t stream);
/* nn.LayerNorm(512), model/blocks.py:74,81 (eps 1e-5) and model/spherical_model.py:173 (eps 1e-6) */
int omni_layernorm512_f32(const float* x, const float* g, const float* b, float* y, int rows, float eps, omni_stream_t stream);
/* softmax(q k^T * 128^-1/2) v, 4 heads x 128 over N <= 64 tokens, model/blocks.py:52-62.  q [B*N,512], kv [B*N,1024] */
int omni_attention_f32(const float* q, const float* kv, float* out, int B, int N, omni_stream_t stream);
/* pred (ReLU) and weight_pred (sigmoid) 3x3 heads, model/spherical_model.py:304-307.  x NHWC [M,P,P,32], w [2][9][32];
 * out_a = relu(pred) * (confidence ? sigmoid(weight) : 1), out_c = sigmoid(weight) (may be NULL); planar [M,P,P] */
int omni_heads_f32(const float* x, const float* w, float bias_pred, float bias_weight, float* out_a, float* out_c,
                   int M, int P, int confidence, omni_stream_t stream);
/* mlp_points1 / mlp_points2 (1x1 conv 3->16->64, BN folded, ReLU) of xyz[N,3,HW] (* depth[Mo,HW] if non-NULL),
 * model/spherical_model_iterative.py:290-305,319,387-393.  out NHWC [Mo,HW,64]. */
int omni_mlp_points_f32(const float* xyz, const float* depth, const float* w1, const float* b1, const float* w2,
                        const float* b2, float* out, int Mo, int N, int HW, omni_stream_t stream);

/* ------------------------------------------------------------------------------------------------
 * Evaluation metrics on the device (first "next" row): test.py:151-176 compute_eval_metrics and
 * metrics.py:7-26, without a device->host copy per batch.
 * ---------------------------------------------------------------------------------------------- */
/* x[mask > 0].median() (torch semantics: the lower middle element).  ws: >= 260 unsigned; out: one device float. */
int omni_masked_median_f32(const float* x, const float* mask, size_t n, unsigned* ws, float* out, omni_stream_t stream);
/* pred *= *scale_num / *scale_den in place (test.py:161-162; device scalars, both NULL = no scaling), then
 * out[9] = abs_rel, sq_rel, rms_sq_lin, rms_sq_log, d1, d2, d3 (masked means, metrics.py:7-26), N = mask.sum(), N_log.
 * ws: >= 2048*9 doubles. */
int omni_depth_metrics_f32(float* pred, const float* gt, const float* mask, const float* scale_num, const float* scale_den,
                           size_t n, double* ws, float* out, omni_stream_t stream);

/* ---------------------------------------------------------------------------------------------------------------
 * Host-facing ends of the hot path, csrc/omni_io.hip.
 *
 * omni_preprocess_rgb_u8: decoded BGR uint8 frames [B,Hs,Ws,3] (device memory; the Python side stages them through pinned
 * buffers on a copy stream) -> cv2.INTER_AREA down-scale to HxW (identity when equal) -> /255 -> float32 [B,3,H,W].
 * Replaces dataset_loader_stanford.py:54,85,92-97 (`readRGBPano`, `rgb.astype(np.float32)/255`, transpose(2,0,1)).
 * omni_preprocess_depth_u16: 16-bit depth frames [B,Hs,Ws] -> float32 -> INTER_AREA -> /65535*128 -> mask (min, max] ->
 * depth [B,1,H,W] (* mask) and mask uint8.  Replaces dataset_loader_stanford.py:76-80,99-109. */
int omni_preprocess_rgb_u8(const unsigned char* src_hwc, float* dst_chw, int B, int Hs, int Ws, int H, int W, omni_stream_t stream);
int omni_preprocess_depth_u16(const unsigned short* src, float* depth, unsigned char* mask, int B, int Hs, int Ws, int H, int W,
                              float min_depth, float max_depth, omni_stream_t stream);
/* PNG decoding on the host (csrc/omni_png.hip; no GPU involved) — the two decode calls of dataset_loader_stanford.py:
 *   :85 cv2.imread(path)      -> kind 0: uint8 [H,W,3] in B,G,R order (alpha dropped, gray replicated, 16-bit samples >> 8)
 *   :96 cv2.imread(path, -1)  -> kind 1: the file's own samples, single-channel gray only: uint8 or uint16 (host byte order) [H,W]
 * `data` is the file's bytes; omni_png_info reads the header only.  Colour types 0/2/3/4/6 at 8 bits, 0/2/4/6 at 16 bits, not interlaced;
 * anything else returns OMNI_ERR_UNSUPPORTED, a damaged stream (signature, chunk CRC, zlib, size) OMNI_ERR_INVALID.
 * omni_png_decode_batch decodes n files of one size on the library's persistent decoder pool (one thread per core, at most 64, started on first use;
 * `threads`: at most that many of THIS call's images at a time, 0: no limit, 1: in the calling thread) — the loader's worker pool (test.py:90-97 uses 8
 * DataLoader workers); several calls may run side by side; dsts[i] may point into pinned memory so that the H2D copy needs no staging. */
int omni_png_info(const void* data, size_t nbytes, int* width, int* height, int* bit_depth, int* color_type);
int omni_png_decode(const void* data, size_t nbytes, void* dst, int H, int W, int kind);
int omni_png_decode_batch(const void* const* datas, const size_t* nbytes, void* const* dsts, int n, int H, int W, int kind, int threads);
/* The decoder's own inflate and checksums (csrc/omni_inflate.h), exposed so that they can be checked against an independent zlib (tests/test_png.py
 * decodes the same valid, truncated and damaged streams with Python's zlib module): what cv2.imread gets from libpng -> libz.
 * omni_zlib_inflate: one zlib stream (RFC 1950) -> at most cap bytes at dst; *produced = bytes written (also on failure: how far it got), *consumed =
 * bytes of the stream including its Adler-32.  OMNI_OK | OMNI_ERR_INVALID (damaged stream, or more than cap bytes) | OMNI_ERR_UNSUPPORTED (the input ends
 * inside the stream).  omni_png_checksums: CRC-32 (chunk check) and Adler-32 (zlib trailer) of a buffer, continuing from *crc32 / *adler32 (start: 0 / 1). */
int omni_zlib_inflate(const void* src, size_t nbytes, void* dst, size_t cap, size_t* produced, size_t* consumed);
int omni_png_checksums(const void* data, size_t nbytes, unsigned* crc32, unsigned* adler32);
/* Reverse-Huber loss of supervision/direct.py:3-18 (train_erp_depth.py:267): *loss = mean_b(sum(loss * mask * weights)_b / sum(mask)_b)
 * with c = max|gt - pred| / 5 evaluated on the device.  `workspace` (omni_berhu_workspace_bytes(B) bytes) carries c and the
 * per-item counts to omni_berhu_grad_f32, which writes dloss/dpred * (*grad_out). */
size_t omni_berhu_workspace_bytes(int B);
int omni_berhu_loss_f32(const float* pred, const float* gt, const float* mask, const float* weights, int B, size_t per_item,
                        void* workspace, float* loss, omni_stream_t stream);
int omni_berhu_grad_f32(const float* pred, const float* gt, const float* mask, const float* weights, int B, size_t per_item,
                        const void* workspace, const float* grad_out, float* grad_pred, omni_stream_t stream);
/* Point cloud of test.py:210-240: depth [B,1,H,W], rgb [B,3,H,W] -> B*H*W packed 15-byte PLY vertex records
 * (x, y, z float32 = uv2xyz(coords2uv(pixel)) * depth, util.py:159-174; three uint8 colours = uint8(rgb * 255)). */
int omni_pointcloud_ply_f32(const float* depth, const float* rgb, unsigned char* records, int B, int H, int W, omni_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* OMNIFUSION_H_ */
