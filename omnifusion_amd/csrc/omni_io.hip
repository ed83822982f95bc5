// omni_io.hip — the host-facing ends of the hot path: what sits between the loader / the training and
// export code of the reference and the equi_pers operators, moved onto the device so that the GPU is never fed or drained by
// per-pixel host work.  gfx950 only.  All kernels are HBM-bound streaming passes (coalesced 16-byte accesses where the layout allows).
//
//   omni_preprocess_rgb_u8     decoded BGR uint8 HWC frame(s) -> cv2.INTER_AREA resize -> /255 -> float32 CHW
//                              (dataset_loader_stanford.py:54,92-97 `readRGBPano` + `rgb.astype(np.float32)/255` + transpose(2,0,1) :85)
//   omni_preprocess_depth_u16  16-bit depth frame -> float32 -> INTER_AREA resize -> /65535*128 -> mask (0.1, 8] -> depth *= mask
//                              (dataset_loader_stanford.py:99-109 `readDepthPano`, :76-80)
//   omni_berhu_loss_f32        reverse-Huber loss + its gradient w.r.t. the prediction (supervision/direct.py:3-18; train_erp_depth.py:267)
//   omni_pointcloud_ply_f32    depth map + panorama -> binary PLY vertex records x,y,z,blue,green,red (test.py:210-240, util.py:159-174)
#include "omni_internal.h"

namespace {

// ------------------------------------------------------------------ cv2.INTER_AREA
// OpenCV's area resampling of a down-scale by (sx, sy) = (src / dst): output pixel d covers the source interval
// [d*s, (d+1)*s); every source pixel contributes with the length of its overlap, normalised by min(s, src - d*s)
// (resize.cpp `computeResizeAreaTab`).  For an integer scale this is the plain box average.  uint8 images are averaged in float and
// rounded to nearest-even back to uint8 (saturate_cast<uchar>(float) = cvRound), BEFORE the loader's /255.
// The oracle (oracle/io_ref.py) restates the same published algorithm; cv2 itself is not installed here: parity unpinned.
struct AreaSpan { int i0, i1; float w0, w1, inv; };   // full cells [i0, i1) weight 1, partial cells i0-1 (w0) and i1 (w1), 1/cell width

__device__ __forceinline__ AreaSpan area_span(int d, float scale, int ssize)
{
    const float fs1 = (float)d * scale, fs2 = fminf(fs1 + scale, (float)ssize);
    const float cell = fminf(scale, (float)ssize - fs1);
    AreaSpan s;
    s.i0 = (int)ceilf(fs1); s.i1 = (int)floorf(fs2);
    if (s.i1 > ssize) s.i1 = ssize;
    if (s.i0 > s.i1) s.i0 = s.i1;
    s.w0 = (float)s.i0 - fs1;                          // overlap with cell i0-1 (0 when aligned)
    s.w1 = fs2 - (float)s.i1;                          // overlap with cell i1
    if (s.w0 < 1e-3f) s.w0 = 0.0f;                     // OpenCV drops overlaps <= 1e-3
    if (s.w1 < 1e-3f) s.w1 = 0.0f;
    s.inv = 1.0f / cell;
    return s;
}

// area average of channel c at destination pixel (y, x); src interleaved [Hs][Ws][C] of SrcT
template <typename SrcT>
__device__ __forceinline__ float area_sample(const SrcT* __restrict__ src, int Hs, int Ws, int C, int c, const AreaSpan& sy, const AreaSpan& sx)
{
    float acc = 0.0f;
    auto row = [&](int yy, float wy) {
        const SrcT* r = src + (size_t)yy * Ws * C + c;
        float a = 0.0f;
        if (sx.w0 > 0.0f) a += (float)r[(size_t)(sx.i0 - 1) * C] * sx.w0;
        for (int xx = sx.i0; xx < sx.i1; ++xx) a += (float)r[(size_t)xx * C];
        if (sx.w1 > 0.0f) a += (float)r[(size_t)sx.i1 * C] * sx.w1;
        acc += a * wy;
    };
    if (sy.w0 > 0.0f) row(sy.i0 - 1, sy.w0);
    for (int yy = sy.i0; yy < sy.i1; ++yy) row(yy, 1.0f);
    if (sy.w1 > 0.0f) row(sy.i1, sy.w1);
    return acc * (sx.inv * sy.inv);
}

__global__ __launch_bounds__(256) void prep_rgb_kernel(const unsigned char* __restrict__ src, float* __restrict__ dst, int B, int Hs, int Ws,
                                                       int H, int W, float sy_, float sx_)
{
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;          // one thread per destination pixel (all 3 channels)
    if (i >= (size_t)B * H * W) return;
    const int x = (int)(i % W), y = (int)((i / W) % H), b = (int)(i / ((size_t)H * W));
    const unsigned char* s = src + (size_t)b * Hs * Ws * 3;
    float v[3];
    if (Hs == H && Ws == W) {
#pragma unroll
        for (int c = 0; c < 3; ++c) v[c] = (float)s[((size_t)y * Ws + x) * 3 + c];
    } else {
        const AreaSpan ay = area_span(y, sy_, Hs), ax = area_span(x, sx_, Ws);
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            const float avg = area_sample<unsigned char>(s, Hs, Ws, 3, c, ay, ax);
            // back to uint8: saturate_cast (nearest-even); OpenCV's 2x2 fast path computes (a + b + c + d + 2) >> 2 (half up)
            v[c] = fminf(255.0f, fmaxf(0.0f, (Hs == 2 * H && Ws == 2 * W) ? floorf(avg + 0.5f) : rintf(avg)));
        }
    }
#pragma unroll
    for (int c = 0; c < 3; ++c) dst[(((size_t)b * 3 + c) * H + y) * W + x] = v[c] / 255.0f;     // :54  (BGR order kept: quirk q10)
}

// frames that already have the network's size (no INTER_AREA pass): four pixels per thread — three aligned 4-byte loads, one 16-byte store
// per channel (the byte-per-lane form above: 44 us for 8 x 512 x 1024, 1.4 TB/s; same arithmetic, same bits)
__global__ __launch_bounds__(256) void prep_rgb_same_kernel(const unsigned* __restrict__ src, float* __restrict__ dst, size_t quads, size_t plane /* H*W */)
{
    const size_t q = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (q >= quads) return;
    const unsigned w0 = src[3 * q], w1 = src[3 * q + 1], w2 = src[3 * q + 2];      // 12 bytes: b0 g0 r0 b1 | g1 r1 b2 g2 | r2 b3 g3 r3
    const size_t px = 4 * q, b = px / plane, o = px - b * plane;
    float* d = dst + b * 3 * plane + o;
    const float c0[4] = {(float)(w0 & 255u), (float)(w0 >> 24), (float)((w1 >> 16) & 255u), (float)((w2 >> 8) & 255u)};
    const float c1[4] = {(float)((w0 >> 8) & 255u), (float)(w1 & 255u), (float)(w1 >> 24), (float)((w2 >> 16) & 255u)};
    const float c2[4] = {(float)((w0 >> 16) & 255u), (float)((w1 >> 8) & 255u), (float)(w2 & 255u), (float)(w2 >> 24)};
    *reinterpret_cast<float4*>(d) = make_float4(c0[0] / 255.0f, c0[1] / 255.0f, c0[2] / 255.0f, c0[3] / 255.0f);
    *reinterpret_cast<float4*>(d + plane) = make_float4(c1[0] / 255.0f, c1[1] / 255.0f, c1[2] / 255.0f, c1[3] / 255.0f);
    *reinterpret_cast<float4*>(d + 2 * plane) = make_float4(c2[0] / 255.0f, c2[1] / 255.0f, c2[2] / 255.0f, c2[3] / 255.0f);
}

__global__ __launch_bounds__(256) void prep_depth_kernel(const unsigned short* __restrict__ src, float* __restrict__ depth, unsigned char* __restrict__ mask,
                                                         int B, int Hs, int Ws, int H, int W, float sy_, float sx_, float min_d, float max_d)
{
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= (size_t)B * H * W) return;
    const int x = (int)(i % W), y = (int)((i / W) % H), b = (int)(i / ((size_t)H * W));
    const unsigned short* s = src + (size_t)b * Hs * Ws;
    float v;
    if (Hs == H && Ws == W) v = (float)s[(size_t)y * Ws + x];
    else { const AreaSpan ay = area_span(y, sy_, Hs), ax = area_span(x, sx_, Ws); v = area_sample<unsigned short>(s, Hs, Ws, 1, 0, ay, ax); }
    v = v / 65535.0f * 128.0f;                                         // :108
    const bool m = (v <= max_d) && (v > min_d);                        // :76
    depth[i] = m ? v : 0.0f;                                           // :79  depth *= mask
    mask[i] = m ? 1 : 0;
}

// ------------------------------------------------------------------ BerHu (reverse Huber), supervision/direct.py:3-18
__device__ __forceinline__ float wave_max_f(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o));
    return v;
}
__device__ __forceinline__ double wave_sum_d(double v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    return v;
}

// pass 1: max |gt - pred| over EVERYTHING (:7 — not only the masked elements), non-negative floats order like their bits
__global__ __launch_bounds__(256) void berhu_max_kernel(const float* __restrict__ pred, const float* __restrict__ gt, size_t n, unsigned* __restrict__ maxbits)
{
    float m = 0.0f;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) m = fmaxf(m, fabsf(gt[i] - pred[i]));
    m = wave_max_f(m);
    if ((threadIdx.x & 63) == 0) atomicMax(maxbits, __float_as_uint(m));
}

// pass 2: per batch item, partial sums of loss*mask*weight and of mask (fixed block -> slot mapping: deterministic)
__global__ __launch_bounds__(256) void berhu_sum_kernel(const float* __restrict__ pred, const float* __restrict__ gt, const float* __restrict__ mask,
                                                        const float* __restrict__ wt, size_t per, const unsigned* __restrict__ maxbits,
                                                        double* __restrict__ part /* [B][gridDim.x][2] */)
{
    __shared__ double red[2][4];
    const int b = blockIdx.y;
    const float c = __uint_as_float(*maxbits) / 5.0f;                  // :7
    double s = 0.0, cnt = 0.0;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < per; i += (size_t)gridDim.x * 256) {
        const size_t j = (size_t)b * per + i;
        const float d = gt[j] - pred[j], ad = fabsf(d);
        const float l = (ad <= c) ? ad : (d * d + c * c) / (2.0f * c);             // :8-10
        s += (double)(l * mask[j] * wt[j]);                                        // :16-17
        cnt += (double)mask[j];                                                    // :15
    }
    s = wave_sum_d(s); cnt = wave_sum_d(cnt);
    if ((threadIdx.x & 63) == 0) { red[0][threadIdx.x >> 6] = s; red[1][threadIdx.x >> 6] = cnt; }
    __syncthreads();
    if (threadIdx.x == 0) {
        double* p = part + ((size_t)b * gridDim.x + blockIdx.x) * 2;
        p[0] = (red[0][0] + red[0][1]) + (red[0][2] + red[0][3]);
        p[1] = (red[1][0] + red[1][1]) + (red[1][2] + red[1][3]);
    }
}

// pass 3: loss = mean_b(sum_b / count_b) (:18); counts[b] kept for the gradient
__global__ void berhu_final_kernel(const double* __restrict__ part, int B, int nblk, float* __restrict__ loss, float* __restrict__ counts)
{
    if (threadIdx.x != 0) return;
    double tot = 0.0;
    for (int b = 0; b < B; ++b) {
        double s = 0.0, c = 0.0;
        for (int k = 0; k < nblk; ++k) { s += part[((size_t)b * nblk + k) * 2]; c += part[((size_t)b * nblk + k) * 2 + 1]; }
        counts[b] = (float)c;
        tot += (double)((float)s / (float)c);          // fp32 division like torch (0/0 -> NaN for an empty mask, like the reference)
    }
    *loss = (float)(tot / B);
}

// gradient w.r.t. pred (c is a Python float in the reference — `.item()` — hence a constant):
// dL/dpred = -(g / B) * mask * weight / count_b * (|d| <= c ? sign(d) : d / c)
__global__ __launch_bounds__(256) void berhu_grad_kernel(const float* __restrict__ pred, const float* __restrict__ gt, const float* __restrict__ mask,
                                                         const float* __restrict__ wt, size_t per, size_t n, const unsigned* __restrict__ maxbits,
                                                         const float* __restrict__ counts, const float* __restrict__ gout, int B, float* __restrict__ grad)
{
    const size_t j = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (j >= n) return;
    const float c = __uint_as_float(*maxbits) / 5.0f;
    const int b = (int)(j / per);
    const float d = gt[j] - pred[j], ad = fabsf(d);
    const float dl = (ad <= c) ? (d > 0.0f ? 1.0f : (d < 0.0f ? -1.0f : 0.0f)) : d / c;      // d loss / d diff
    grad[j] = -(*gout / (float)B) * (mask[j] * wt[j] / counts[b]) * dl;
}

// ------------------------------------------------------------------ point cloud, test.py:210-240
// vertex i of image b (row-major, x fastest — np.meshgrid(range(w), range(h)) reshaped, :211-213):
//   coords = (x+1, y+1);  u = (cx - (w/2 + 0.5)) / w * 2 pi,  v = -(cy - (h/2 + 0.5)) / h * pi          util.py:159-165
//   xyz = (cos v sin u, cos v cos u, sin v) * depth                                                      util.py:168-173, test.py:218-219
//   colour = uint8(rgb * 255) of the three image channels in loader order, written as blue, green, red   test.py:229,236
// record: 3 x float32 + 3 x uint8 = 15 bytes, packed (the numpy structured array ply.py:303-314 writes)
__global__ __launch_bounds__(256) void pointcloud_kernel(const float* __restrict__ depth, const float* __restrict__ rgb, unsigned char* __restrict__ out,
                                                         int B, int H, int W)
{
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= (size_t)B * H * W) return;
    const int x = (int)(i % W), y = (int)((i / W) % H), b = (int)(i / ((size_t)H * W));
    const float PI_F = 3.14159265358979323846f;
    // numpy float32 arithmetic of coords2uv: (int - python float) -> float64, / w * 2 * pi in float64, stored to float32
    const double u64 = ((double)(x + 1) - ((double)W / 2.0 + 0.5)) / (double)W * 2.0 * 3.141592653589793;
    const double v64 = -((double)(y + 1) - ((double)H / 2.0 + 0.5)) / (double)H * 3.141592653589793;
    const float u = (float)u64, v = (float)v64;
    (void)PI_F;
    const float cv = cosf(v), sv = sinf(v), su = sinf(u), cu = cosf(u);
    const float d = depth[i];
    float p[3] = {cv * su * d, cv * cu * d, sv * d};
    unsigned char* o = out + i * 15;
    __builtin_memcpy(o, p, 12);
    const size_t plane = (size_t)H * W, pix = (size_t)y * W + x;
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        const float col = rgb[((size_t)b * 3 + c) * plane + pix] * 255.0f;
        o[12 + c] = (unsigned char)(int)fminf(255.0f, fmaxf(0.0f, col));            // astype(np.uint8): truncation
    }
}

}  // namespace

extern "C" int omni_preprocess_rgb_u8(const unsigned char* src_hwc, float* dst_chw, int B, int Hs, int Ws, int H, int W, omni_stream_t stream)
{
    if (!src_hwc || !dst_chw) OMNI_FAIL(OMNI_ERR_INVALID, "omni_preprocess_rgb_u8: null device pointer");
    if (B < 0 || Hs < 1 || Ws < 1 || H < 1 || W < 1) OMNI_FAIL(OMNI_ERR_INVALID, "omni_preprocess_rgb_u8: bad shape");
    if (H > Hs || W > Ws) OMNI_FAIL(OMNI_ERR_UNSUPPORTED, "omni_preprocess_rgb_u8: INTER_AREA is a down-scale here (the loader shrinks 4096x2048 scans)");
    if (B == 0) return OMNI_OK;
    const size_t n = (size_t)B * H * W;
    if (Hs == H && Ws == W && ((size_t)H * W) % 4 == 0 && (uintptr_t)src_hwc % 4 == 0 && (uintptr_t)dst_chw % 16 == 0) {
        hipLaunchKernelGGL(prep_rgb_same_kernel, dim3((unsigned)((n / 4 + 255) / 256)), dim3(256), 0, (hipStream_t)stream, (const unsigned*)src_hwc, dst_chw, n / 4, (size_t)H * W);
        OMNI_HIP(hipGetLastError());
        return OMNI_OK;
    }
    hipLaunchKernelGGL(prep_rgb_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, src_hwc, dst_chw, B, Hs, Ws, H, W,
                       (float)((double)Hs / H), (float)((double)Ws / W));
    OMNI_HIP(hipGetLastError());
    return OMNI_OK;
}

extern "C" int omni_preprocess_depth_u16(const unsigned short* src, float* depth, unsigned char* mask, int B, int Hs, int Ws, int H, int W,
                                         float min_depth, float max_depth, omni_stream_t stream)
{
    if (!src || !depth || !mask) OMNI_FAIL(OMNI_ERR_INVALID, "omni_preprocess_depth_u16: null device pointer");
    if (B < 0 || Hs < 1 || Ws < 1 || H < 1 || W < 1) OMNI_FAIL(OMNI_ERR_INVALID, "omni_preprocess_depth_u16: bad shape");
    if (H > Hs || W > Ws) OMNI_FAIL(OMNI_ERR_UNSUPPORTED, "omni_preprocess_depth_u16: INTER_AREA is a down-scale here");
    if (B == 0) return OMNI_OK;
    const size_t n = (size_t)B * H * W;
    hipLaunchKernelGGL(prep_depth_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, src, depth, mask, B, Hs, Ws, H, W,
                       (float)((double)Hs / H), (float)((double)Ws / W), min_depth, max_depth);
    OMNI_HIP(hipGetLastError());
    return OMNI_OK;
}

extern "C" size_t omni_berhu_workspace_bytes(int B) { return 64 + sizeof(double) * 2 * 256 * (size_t)(B > 0 ? B : 1) + sizeof(float) * (size_t)(B > 0 ? B : 1); }

extern "C" int omni_berhu_loss_f32(const float* pred, const float* gt, const float* mask, const float* weights, int B, size_t per_item,
                                   void* workspace, float* loss, omni_stream_t stream)
{
    if (!pred || !gt || !mask || !weights || !workspace || !loss) OMNI_FAIL(OMNI_ERR_INVALID, "omni_berhu_loss_f32: null device pointer");
    if (B < 1 || per_item < 1) OMNI_FAIL(OMNI_ERR_INVALID, "omni_berhu_loss_f32: empty batch");
    hipStream_t s = (hipStream_t)stream;
    unsigned* maxbits = (unsigned*)workspace;
    double* part = (double*)((char*)workspace + 64);
    float* counts = (float*)(part + 2 * 256 * (size_t)B);
    const size_t n = (size_t)B * per_item;
    OMNI_HIP(hipMemsetAsync(maxbits, 0, sizeof(unsigned), s));
    const unsigned g1 = (unsigned)((n + 255) / 256 < 2048 ? (n + 255) / 256 : 2048);
    hipLaunchKernelGGL(berhu_max_kernel, dim3(g1), dim3(256), 0, s, pred, gt, n, maxbits);
    const unsigned nblk = (unsigned)((per_item + 255) / 256 < 256 ? (per_item + 255) / 256 : 256);
    hipLaunchKernelGGL(berhu_sum_kernel, dim3(nblk, B), dim3(256), 0, s, pred, gt, mask, weights, per_item, (const unsigned*)maxbits, part);
    hipLaunchKernelGGL(berhu_final_kernel, dim3(1), dim3(64), 0, s, (const double*)part, B, (int)nblk, loss, counts);
    OMNI_HIP(hipGetLastError());
    return OMNI_OK;
}

extern "C" int omni_berhu_grad_f32(const float* pred, const float* gt, const float* mask, const float* weights, int B, size_t per_item,
                                   const void* workspace, const float* grad_out, float* grad_pred, omni_stream_t stream)
{
    if (!pred || !gt || !mask || !weights || !workspace || !grad_out || !grad_pred) OMNI_FAIL(OMNI_ERR_INVALID, "omni_berhu_grad_f32: null device pointer");
    if (B < 1 || per_item < 1) OMNI_FAIL(OMNI_ERR_INVALID, "omni_berhu_grad_f32: empty batch");
    const unsigned* maxbits = (const unsigned*)workspace;
    const float* counts = (const float*)((const char*)workspace + 64 + sizeof(double) * 2 * 256 * (size_t)B);
    const size_t n = (size_t)B * per_item;
    hipLaunchKernelGGL(berhu_grad_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, pred, gt, mask, weights, per_item, n,
                       maxbits, counts, grad_out, B, grad_pred);
    OMNI_HIP(hipGetLastError());
    return OMNI_OK;
}

extern "C" int omni_pointcloud_ply_f32(const float* depth, const float* rgb, unsigned char* records, int B, int H, int W, omni_stream_t stream)
{
    if (!depth || !rgb || !records) OMNI_FAIL(OMNI_ERR_INVALID, "omni_pointcloud_ply_f32: null device pointer");
    if (B < 0 || H < 1 || W < 1) OMNI_FAIL(OMNI_ERR_INVALID, "omni_pointcloud_ply_f32: bad shape");
    if (B == 0) return OMNI_OK;
    const size_t n = (size_t)B * H * W;
    hipLaunchKernelGGL(pointcloud_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, depth, rgb, records, B, H, W);
    OMNI_HIP(hipGetLastError());
    return OMNI_OK;
}
