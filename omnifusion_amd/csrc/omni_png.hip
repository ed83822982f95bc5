// omni_png.hip — PNG decoding for the host input pipeline (host code only; no kernel in this file).
//
// Replaces the two decode calls of /root/reference/dataset_loader_stanford.py:
//   :85  rgb   = cv2.imread(path)        -> uint8  [H,W,3], B G R order (IMREAD_COLOR: alpha dropped, gray replicated, 16-bit >> 8)
//   :96  depth = cv2.imread(path, -1)    -> the file's own sample type (IMREAD_UNCHANGED): Stanford2D3D depth is 16-bit gray -> uint16 [H,W]
// ("PNG/EXR decode -> INTER_AREA resize -> BGR/255 -> pinned-memory async H2D"; the loader's EXR reader `read_exr` is
// dead code in the reference — :93 is commented out — so PNG is the only codec on the path.)
//
// PNG is lossless: every conforming decoder returns the same samples, so parity here is conformance to the PNG specification
// (ISO/IEC 15948: chunk layout, zlib stream over the concatenated IDAT chunks, the five scan-line filters) plus OpenCV's documented
// conversions listed above.  Everything is here: chunk walk, CRC check, inflate (omni_inflate.h: rounds 4-5 called libz, which was 70 % of a panorama's decode time), un-filtering,
// sample conversion, the worker pool.
// Supported: colour types 0 (gray), 2 (RGB), 3 (palette), 4 (gray + alpha), 6 (RGB + alpha) at 8 bits, types 0 / 2 / 4 / 6 at 16 bits; no
// interlacing (Adam7) and no 1/2/4-bit samples (OMNI_ERR_UNSUPPORTED; neither occurs in the dataset).
#include <string.h>
#include <pthread.h>
#include <stdio.h>
#include <stdlib.h>
#include <atomic>
#include <condition_variable>
#include <deque>
#include <functional>
#include <memory>
#include <new>
#include <stdexcept>
#include <thread>
#include <vector>
#include "omni_internal.h"
#include "omni_inflate.h"

namespace {

struct PngInfo { int w, h, depth, ctype, interlace; };

inline unsigned be32(const unsigned char* p) { return ((unsigned)p[0] << 24) | ((unsigned)p[1] << 16) | ((unsigned)p[2] << 8) | (unsigned)p[3]; }

int channels_of(int ctype) { return ctype == 0 ? 1 : ctype == 2 ? 3 : ctype == 3 ? 1 : ctype == 4 ? 2 : ctype == 6 ? 4 : 0; }

constexpr size_t RAW_SLACK = 32;        // bytes behind the last scan line that the 4- / 16-byte accesses of the row kernels may read

// walks the chunks: fills `info`, the palette, and inflates the zlib stream of the IDAT chunks into `raw` (filter byte + samples per scan line).
// One IDAT: inflated where it lies; several (libpng writes 8-KiB chunks): their payloads are laid end to end in `zbuf` first.
// (expect_w / expect_h > 0: the destination's size — a header that announces anything else is refused BEFORE a byte is allocated for it)
int png_unpack(const unsigned char* d, size_t n, PngInfo& info, std::vector<unsigned char>& raw, std::vector<unsigned char>& zbuf, unsigned char (*pal)[3], int* npal,
               bool info_only, int expect_w = 0, int expect_h = 0)
{
    static const unsigned char sig[8] = {0x89, 'P', 'N', 'G', 0x0d, 0x0a, 0x1a, 0x0a};
    if (n < 8 + 25 || memcmp(d, sig, 8) != 0) OMNI_FAIL(OMNI_ERR_INVALID, "png: not a PNG stream");
    size_t pos = 8;
    bool have_ihdr = false, done = false;
    size_t rawsize = 0;
    const unsigned char* z0 = nullptr;                             // the first non-empty IDAT payload ...
    size_t zn = 0, zparts = 0, ztotal = 0;                         // ... its length, the number of payloads, their total length
    while (pos + 12 <= n && !done) {
        const unsigned len = be32(d + pos);
        const unsigned char* type = d + pos + 4;
        const unsigned char* body = d + pos + 8;
        if ((size_t)len > n - pos - 12) OMNI_FAIL(OMNI_ERR_INVALID, "png: truncated chunk");
        if (be32(body + len) != omni_inflate::crc32_fast(0u, type, (size_t)len + 4)) OMNI_FAIL(OMNI_ERR_INVALID, "png: chunk CRC mismatch");
        if (!memcmp(type, "IHDR", 4)) {
            if (have_ihdr) OMNI_FAIL(OMNI_ERR_INVALID, "png: a second IHDR chunk");
            if (len != 13) OMNI_FAIL(OMNI_ERR_INVALID, "png: bad IHDR");
            info.w = (int)be32(body); info.h = (int)be32(body + 4); info.depth = body[8]; info.ctype = body[9]; info.interlace = body[12];
            have_ihdr = true;
            if (info.w <= 0 || info.h <= 0 || channels_of(info.ctype) == 0 || body[10] != 0 || body[11] != 0) OMNI_FAIL(OMNI_ERR_INVALID, "png: bad IHDR fields");
            if (info_only) return OMNI_OK;
            if (info.interlace != 0) OMNI_FAIL(OMNI_ERR_UNSUPPORTED, "png: interlaced (Adam7) files are not supported");
            if (!(info.depth == 8 || (info.depth == 16 && info.ctype != 3))) OMNI_FAIL(OMNI_ERR_UNSUPPORTED, "png: only 8- and 16-bit samples are supported");
            // the header is untrusted input: the sizes are checked (against the destination, and against 2 GiB in 64-bit arithmetic —
            // w, h < 2^31 and <= 8 bytes per pixel cannot wrap it) before anything is allocated for them
            if ((expect_w > 0 && info.w != expect_w) || (expect_h > 0 && info.h != expect_h))
                OMNI_FAIL(OMNI_ERR_INVALID, "png: the image is " + std::to_string(info.h) + " x " + std::to_string(info.w) + ", the destination " + std::to_string(expect_h) + " x " + std::to_string(expect_w));
            const unsigned long long stride = (unsigned long long)info.w * channels_of(info.ctype) * (info.depth / 8);
            if (stride + 1 > 0x7fffffffull || (stride + 1) * (unsigned long long)info.h > 0x7fffffffull) OMNI_FAIL(OMNI_ERR_UNSUPPORTED, "png: image too large (2 GiB of samples or more)");
            rawsize = (size_t)((stride + 1) * (unsigned long long)info.h);
            if (raw.size() < rawsize + RAW_SLACK) raw.resize(rawsize + RAW_SLACK);   // (grows only: the caller keeps it across images; inflate writes every byte it reports)
        } else if (!have_ihdr) {
            OMNI_FAIL(OMNI_ERR_INVALID, "png: IHDR is not the first chunk");
        } else if (!memcmp(type, "PLTE", 4)) {
            if (len % 3 != 0 || len > 768) OMNI_FAIL(OMNI_ERR_INVALID, "png: bad PLTE");
            *npal = (int)(len / 3);
            memcpy(pal, body, len);
        } else if (!memcmp(type, "IDAT", 4)) {
            if (len != 0) {                                        // an empty IDAT is legal (cv2.imread reads such files): nothing to inflate
                if (zparts == 0) { z0 = body; zn = len; }
                else {
                    if (zparts == 1) { zbuf.clear(); zbuf.insert(zbuf.end(), z0, z0 + zn); }
                    zbuf.insert(zbuf.end(), body, body + len);
                }
                ++zparts; ztotal += len;
            }
        } else if (!memcmp(type, "IEND", 4)) {
            done = true;
        } else if (!(type[0] & 0x20)) {                            // an unknown CRITICAL chunk
            OMNI_FAIL(OMNI_ERR_UNSUPPORTED, "png: unknown critical chunk");
        }
        pos += 12 + (size_t)len;
    }
    if (!have_ihdr) OMNI_FAIL(OMNI_ERR_INVALID, "png: no IHDR");
    if (!done) OMNI_FAIL(OMNI_ERR_INVALID, "png: no IEND");
    const unsigned char* zp = zparts > 1 ? zbuf.data() : z0;
    size_t produced = 0;
    const int zr = zparts == 0 ? omni_inflate::INF_TRUNCATED : omni_inflate::zlib_decompress(zp, ztotal, raw.data(), rawsize, nullptr, &produced);
    if (zr == omni_inflate::INF_OUTPUT_FULL) OMNI_FAIL(OMNI_ERR_INVALID, "png: more image data than the header announces");
    if (zr == omni_inflate::INF_CORRUPT) OMNI_FAIL(OMNI_ERR_INVALID, "png: corrupt zlib stream");
    if (produced != rawsize) OMNI_FAIL(OMNI_ERR_INVALID, "png: image data ends early");
    // (INF_TRUNCATED with every sample present — a stream cut inside its Adler-32 trailer — is accepted, as libz's streaming inflate accepted it)
    return OMNI_OK;
}

// ------------------------------------------------------------------ scan-line reconstruction (PNG spec 9.2), one row at a time
inline int paeth(int a, int b, int c)
{
    const int p = a + b - c, pa = abs(p - a), pb = abs(p - b), pc = abs(p - c);
    return (pa <= pb && pa <= pc) ? a : (pb <= pc ? b : c);
}

// x: the row's n bytes (in place), prev: the reconstructed row above (nullptr for the first row = a row of zeros), BPP bytes per complete pixel
template <int BPP>
bool unfilter_row(int ft, unsigned char* x, const unsigned char* prev, size_t n)
{
    switch (ft) {
    case 0: return true;
    case 1: for (size_t i = BPP; i < n; ++i) x[i] = (unsigned char)(x[i] + x[i - BPP]); return true;
    case 2: if (prev) for (size_t i = 0; i < n; ++i) x[i] = (unsigned char)(x[i] + prev[i]); return true;
    case 3:
        if (prev) {
            for (size_t i = 0; i < (size_t)BPP && i < n; ++i) x[i] = (unsigned char)(x[i] + (prev[i] >> 1));
            for (size_t i = BPP; i < n; ++i) x[i] = (unsigned char)(x[i] + ((x[i - BPP] + prev[i]) >> 1));
        } else
            for (size_t i = BPP; i < n; ++i) x[i] = (unsigned char)(x[i] + (x[i - BPP] >> 1));
        return true;
    case 4:
        if (prev) {
            for (size_t i = 0; i < (size_t)BPP && i < n; ++i) x[i] = (unsigned char)(x[i] + prev[i]);                      // paeth(0, b, 0) = b
            for (size_t i = BPP; i < n; ++i) x[i] = (unsigned char)(x[i] + paeth(x[i - BPP], prev[i], prev[i - BPP]));
        } else
            for (size_t i = BPP; i < n; ++i) x[i] = (unsigned char)(x[i] + x[i - BPP]);                                       // paeth(a, 0, 0) = a
        return true;
    default: return false;
    }
}

// The three filters with a serial dependence on the pixel to the left (Sub, Average, Paeth), one PIXEL per step in a vector register instead of one
// byte: 8-byte loads (pixels have 1-8 bytes; the bytes behind the pixel belong to its right neighbours and are ignored), BPP-byte stores (the row is
// reconstructed in place and the neighbours' raw bytes must survive).  prev != nullptr for Average / Paeth; the first pixel (no left neighbour) and the
// last ones (their 8-byte load would leave the row) go through the byte loops.  8-bit RGB panoramas: 3 bytes per pixel; 16-bit depth maps: 2.
inline __m128i load8(const unsigned char* p) { long long v; memcpy(&v, p, 8); return _mm_cvtsi64_si128(v); }
template <int BPP> inline void store_px(unsigned char* p, __m128i v) { const long long r = _mm_cvtsi128_si64(v); memcpy(p, &r, BPP); }

template <int BPP>
inline void sub_row_v(unsigned char* x, size_t n)
{
    if (n < 16) { unfilter_row<BPP>(1, x, nullptr, n); return; }
    __m128i a = load8(x);
    size_t i = BPP;
    for (; i + 8 <= n; i += BPP) { a = _mm_add_epi8(a, load8(x + i)); store_px<BPP>(x + i, a); }
    for (; i < n; ++i) x[i] = (unsigned char)(x[i] + x[i - BPP]);
}

template <int BPP>
inline void avg_row_v(unsigned char* x, const unsigned char* prev, size_t n)
{
    if (n < 16) { unfilter_row<BPP>(3, x, prev, n); return; }
    for (size_t i = 0; i < (size_t)BPP; ++i) x[i] = (unsigned char)(x[i] + (prev[i] >> 1));
    __m128i a = load8(x);
    const __m128i one = _mm_set1_epi8(1);
    size_t i = BPP;
    for (; i + 8 <= n; i += BPP) {
        const __m128i b = load8(prev + i);
        const __m128i avg = _mm_sub_epi8(_mm_avg_epu8(a, b), _mm_and_si128(_mm_xor_si128(a, b), one));      // pavgb rounds up: floor((a + b) / 2) = it - ((a ^ b) & 1)
        a = _mm_add_epi8(load8(x + i), avg);
        store_px<BPP>(x + i, a);
    }
    for (; i < n; ++i) x[i] = (unsigned char)(x[i] + ((x[i - BPP] + prev[i]) >> 1));
}

template <int BPP>
__attribute__((target("ssse3"))) inline void paeth_row_v(unsigned char* x, const unsigned char* prev, size_t n)
{
    for (size_t i = 0; i < (size_t)BPP; ++i) x[i] = (unsigned char)(x[i] + prev[i]);
    const __m128i zero = _mm_setzero_si128(), low = _mm_set1_epi16(0xff);
    __m128i a = _mm_unpacklo_epi8(load8(x), zero), c = _mm_unpacklo_epi8(load8(prev), zero);           // 16-bit lanes
    size_t i = BPP;
    for (; i + 8 <= n; i += BPP) {
        const __m128i b = _mm_unpacklo_epi8(load8(prev + i), zero);
        const __m128i dbc = _mm_sub_epi16(b, c), dac = _mm_sub_epi16(a, c);                                // p - a = b - c, p - b = a - c, p - c = their sum
        const __m128i pa = _mm_abs_epi16(dbc), pb = _mm_abs_epi16(dac), pc = _mm_abs_epi16(_mm_add_epi16(dbc, dac));
        const __m128i smallest = _mm_min_epi16(pc, _mm_min_epi16(pa, pb));
        const __m128i ma = _mm_cmpeq_epi16(smallest, pa), mb = _mm_cmpeq_epi16(smallest, pb);             // ties: a, then b, then c (the spec's order)
        const __m128i bc = _mm_or_si128(_mm_and_si128(mb, b), _mm_andnot_si128(mb, c));
        const __m128i pred = _mm_or_si128(_mm_and_si128(ma, a), _mm_andnot_si128(ma, bc));
        a = _mm_and_si128(_mm_add_epi16(pred, _mm_unpacklo_epi8(load8(x + i), zero)), low);
        c = b;
        store_px<BPP>(x + i, _mm_packus_epi16(a, a));
    }
    for (; i < n; ++i) x[i] = (unsigned char)(x[i] + paeth(x[i - BPP], prev[i], prev[i - BPP]));
}

bool cpu_has_ssse3() { static const bool v = __builtin_cpu_supports("ssse3"); return v; }

template <int BPP>
bool unfilter_px(int ft, unsigned char* x, const unsigned char* prev, size_t n)
{
    if (ft == 1) { sub_row_v<BPP>(x, n); return true; }
    if (prev && ft == 3) { avg_row_v<BPP>(x, prev, n); return true; }
    if (prev && ft == 4 && n >= 16 && cpu_has_ssse3()) { paeth_row_v<BPP>(x, prev, n); return true; }
    return unfilter_row<BPP>(ft, x, prev, n);
}

// RGB -> BGR, 8 bits: five pixels per 16-byte load / store (the 16th byte is rewritten by the next store; the loop ends while 6 pixels remain)
__attribute__((target("ssse3"))) inline void rgb_to_bgr_row_ssse3(const unsigned char* s, unsigned char* q, int W)
{
    const __m128i sh = _mm_setr_epi8(2, 1, 0, 5, 4, 3, 8, 7, 6, 11, 10, 9, 14, 13, 12, 15);
    int x = 0;
    for (; x + 6 <= W; x += 5, s += 15, q += 15) _mm_storeu_si128((__m128i*)q, _mm_shuffle_epi8(_mm_loadu_si128((const __m128i*)s), sh));
    for (; x < W; ++x, s += 3, q += 3) { const unsigned char r = s[0], g = s[1], b = s[2]; q[0] = b; q[1] = g; q[2] = r; }
}

bool unfilter_any(int bpp, int ft, unsigned char* x, const unsigned char* prev, size_t n)
{
    switch (bpp) {
    case 1: return unfilter_px<1>(ft, x, prev, n);
    case 2: return unfilter_px<2>(ft, x, prev, n);
    case 3: return unfilter_px<3>(ft, x, prev, n);
    case 4: return unfilter_px<4>(ft, x, prev, n);
    case 6: return unfilter_px<6>(ft, x, prev, n);
    case 8: return unfilter_px<8>(ft, x, prev, n);
    default: return false;
    }
}

// the free list of scratch buffers (png_decode_body's Lease) and its lock: at namespace scope so that the fork handlers below can hold the lock across fork()
std::mutex& lease_mu() { static std::mutex m; return m; }
std::vector<std::pair<std::vector<unsigned char>, std::vector<unsigned char>>>& lease_pool() { static std::vector<std::pair<std::vector<unsigned char>, std::vector<unsigned char>>> p; return p; }

// kind 0: cv2.imread(path) -> uint8 BGR [H,W,3];  kind 1: cv2.imread(path, -1) of a single-channel file -> uint8 / uint16 [H,W] in host byte order
int png_decode_body(const unsigned char* d, size_t n, void* dst, int H, int W, int kind, int* elem_bytes)
{
    if (H <= 0 || W <= 0) OMNI_FAIL(OMNI_ERR_INVALID, "png: the destination must have a positive size");
    PngInfo info{};
    // scratch buffers are kept between images (a free list: the decoder threads of a batch call are short-lived): a fresh 1.5-25 MB vector per image
    // is an mmap + page faults + a zero fill each time, and with 64 decoder threads the kernel's address-space lock serialises them (round 5: 216 MB/s
    // per thread on 8 threads, 80 MB/s on 64)
    struct Lease {
        std::vector<unsigned char> buf, zbuf;
        Lease() { std::lock_guard<std::mutex> lk(lease_mu()); if (!lease_pool().empty()) { buf.swap(lease_pool().back().first); zbuf.swap(lease_pool().back().second); lease_pool().pop_back(); } }
        ~Lease() { std::lock_guard<std::mutex> lk(lease_mu()); if (lease_pool().size() < 256 && buf.capacity() <= (64u << 20) && zbuf.capacity() <= (64u << 20)) lease_pool().emplace_back(std::move(buf), std::move(zbuf)); }
    } lease;
    std::vector<unsigned char>& raw = lease.buf;
    unsigned char pal[256][3];
    int npal = 0;
    int rc = png_unpack(d, n, info, raw, lease.zbuf, pal, &npal, false, W, H);
    if (rc != OMNI_OK) return rc;
    const int ch = channels_of(info.ctype), bs = info.depth / 8, bpp = ch * bs;
    const size_t stride = (size_t)W * bpp;
    if (kind == 1 && (ch != 1 || info.ctype == 3)) OMNI_FAIL(OMNI_ERR_UNSUPPORTED, "png: the unchanged (-1) read is implemented for single-channel gray files (depth maps)");
    if (elem_bytes) *elem_bytes = kind == 0 ? 1 : bs;
    const bool rgb8 = info.ctype == 2 && bs == 1 && cpu_has_ssse3();
    const unsigned char* prev = nullptr;
    // a row is reconstructed and converted while it sits in the L1 cache (one pass over the image instead of two)
    for (int y = 0; y < H; ++y) {
        unsigned char* row = raw.data() + (size_t)y * (stride + 1);
        unsigned char* s = row + 1;
        if (!unfilter_any(bpp, row[0], s, prev, stride)) OMNI_FAIL(OMNI_ERR_INVALID, "png: unknown filter type");
        prev = s;
        if (kind == 0) {
            unsigned char* q = (unsigned char*)dst + (size_t)y * W * 3;
            if (rgb8) { rgb_to_bgr_row_ssse3(s, q, W); continue; }
            for (int x = 0; x < W; ++x, s += bpp, q += 3) {
                unsigned char r, g, b;                              // 16-bit samples: the high byte (OpenCV's IMREAD_COLOR scales 16 -> 8 bits by >> 8)
                if (info.ctype == 3) { const int i = s[0]; if (i >= npal) OMNI_FAIL(OMNI_ERR_INVALID, "png: palette index out of range"); r = pal[i][0]; g = pal[i][1]; b = pal[i][2]; }
                else if (ch <= 2) { r = g = b = s[0]; }
                else { r = s[0]; g = s[bs]; b = s[2 * bs]; }
                q[0] = b; q[1] = g; q[2] = r;
            }
        } else if (bs == 1) {
            memcpy((unsigned char*)dst + (size_t)y * W, s, W);
        } else {
            unsigned short* q = (unsigned short*)dst + (size_t)y * W;
            for (int x = 0; x < W; ++x) q[x] = (unsigned short)((s[2 * x] << 8) | s[2 * x + 1]);      // network (big-endian) -> host order
        }
    }
    return OMNI_OK;
}

// no exception leaves the decoder: it runs behind extern "C" entry points and on std::thread workers, where one would end the process
int png_decode(const unsigned char* d, size_t n, void* dst, int H, int W, int kind, int* elem_bytes)
{
    try {
        return png_decode_body(d, n, dst, H, W, kind, elem_bytes);
    } catch (const std::bad_alloc&) {
        OMNI_FAIL(OMNI_ERR_UNSUPPORTED, "png: out of host memory while decoding");
    } catch (const std::exception& e) {
        OMNI_FAIL(OMNI_ERR_INVALID, std::string("png: ") + e.what());
    }
}
}  // namespace

extern "C" int omni_png_info(const void* data, size_t nbytes, int* width, int* height, int* bit_depth, int* color_type)
{
    if (!data) OMNI_FAIL(OMNI_ERR_INVALID, "omni_png_info: null buffer");
    PngInfo info{};
    std::vector<unsigned char> raw;
    int npal = 0;
    int rc;
    std::vector<unsigned char> zbuf;
    try { rc = png_unpack((const unsigned char*)data, nbytes, info, raw, zbuf, nullptr, &npal, true); }
    catch (const std::exception& e) { OMNI_FAIL(OMNI_ERR_INVALID, std::string("png: ") + e.what()); }
    if (rc != OMNI_OK) return rc;
    if (width) *width = info.w;
    if (height) *height = info.h;
    if (bit_depth) *bit_depth = info.depth;
    if (color_type) *color_type = info.ctype;
    return OMNI_OK;
}

extern "C" int omni_png_decode(const void* data, size_t nbytes, void* dst, int H, int W, int kind)
{
    if (!data || !dst) OMNI_FAIL(OMNI_ERR_INVALID, "omni_png_decode: null buffer");
    if (kind != 0 && kind != 1) OMNI_FAIL(OMNI_ERR_INVALID, "omni_png_decode: kind must be 0 (BGR uint8, cv2.imread) or 1 (unchanged gray, cv2.imread(path, -1))");
    return png_decode((const unsigned char*)data, nbytes, dst, H, W, kind, nullptr);
}

extern "C" int omni_zlib_inflate(const void* src, size_t nbytes, void* dst, size_t cap, size_t* produced, size_t* consumed)
{
    if (produced) *produced = 0;
    if (consumed) *consumed = 0;
    if (!src || (!dst && cap)) OMNI_FAIL(OMNI_ERR_INVALID, "omni_zlib_inflate: null buffer");
    unsigned char none = 0;
    const int r = omni_inflate::zlib_decompress((const unsigned char*)src, nbytes, dst ? (unsigned char*)dst : &none, cap, consumed, produced);
    if (r == omni_inflate::INF_OK) return OMNI_OK;
    if (r == omni_inflate::INF_TRUNCATED) OMNI_FAIL(OMNI_ERR_UNSUPPORTED, "omni_zlib_inflate: the input ends inside the stream");
    if (r == omni_inflate::INF_OUTPUT_FULL) OMNI_FAIL(OMNI_ERR_INVALID, "omni_zlib_inflate: the stream holds more than the destination");
    OMNI_FAIL(OMNI_ERR_INVALID, "omni_zlib_inflate: corrupt stream");
}

extern "C" int omni_png_checksums(const void* data, size_t nbytes, unsigned* crc32, unsigned* adler32)
{
    if (!data && nbytes) OMNI_FAIL(OMNI_ERR_INVALID, "omni_png_checksums: null buffer");
    if (crc32) *crc32 = omni_inflate::crc32_fast(*crc32, (const unsigned char*)data, nbytes);
    if (adler32) *adler32 = omni_inflate::adler32(*adler32, (const unsigned char*)data, nbytes);
    return OMNI_OK;
}

// A process-wide pool of decoder threads, started on first use: a batch call used to create (and join) its own std::threads — with several batch
// decoders side by side that was ~18 000 thread creations per second, each an 8-MB stack mmap / munmap under the process's address-space lock, and the
// whole loader levelled off at 2300 panoramas/s whatever the number of decoders (round 5, tools/png_fed_ab.py).
//
// fork(): torch DataLoader workers (the reference's test.py:90-97) are forked children; a child inherits the pool OBJECT but none of its threads, and
// any mutex another thread held at the moment of the fork stays locked for ever.  pthread_atfork handlers (registered with the pool's first use) hold
// the three locks of this file across the fork — so the child finds them unlocked and the structures consistent — and the child ABANDONS the inherited
// pool (never destroyed: its std::thread members are joinable and their threads do not exist); the child's first batch call starts a pool of its own.
namespace {
class PngPool {
public:
    static PngPool& get()
    {
        std::lock_guard<std::mutex> lk(create_mu());
        static const int registered = pthread_atfork(&PngPool::fork_prepare, &PngPool::fork_parent, &PngPool::fork_child);
        (void)registered;
        if (!instance()) instance() = new PngPool();               // (never deleted: the threads sleep on the queue until the process ends)
        return *instance();
    }
    void submit(std::function<void()> f) { { std::lock_guard<std::mutex> lk(mu_); q_.push_back(std::move(f)); } cv_.notify_one(); }
    int size() const { return (int)th_.size(); }
private:
    static std::mutex& create_mu() { static std::mutex m; return m; }
    static PngPool*& instance() { static PngPool* p = nullptr; return p; }
    static void fork_prepare() { create_mu().lock(); lease_mu().lock(); if (instance()) instance()->mu_.lock(); }
    static void fork_parent() { if (instance()) instance()->mu_.unlock(); lease_mu().unlock(); create_mu().unlock(); }
    static void fork_child() { if (instance()) instance()->mu_.unlock(); instance() = nullptr; lease_mu().unlock(); create_mu().unlock(); }
    PngPool()
    {
        int n = (int)std::thread::hardware_concurrency();
        n = std::max(2, std::min(n > 0 ? n / 2 : 8, 64));          // one decoder per core (two hardware threads each), at most 64 ...
        // ... and no more than the CPU time the container may use (cgroup v2 cpu.max "quota period" / v1 cfs quota): the GPU boxes of this pool show 256
        // hardware threads and grant 16 CPUs — 64 decoder threads there only take turns
        long long quota = -1, period = 100000;
        if (FILE* f = fopen("/sys/fs/cgroup/cpu.max", "r")) { char q[32] = {0}; if (fscanf(f, "%31s %lld", q, &period) >= 1 && strcmp(q, "max") != 0) quota = atoll(q); fclose(f); }
        else if (FILE* f1 = fopen("/sys/fs/cgroup/cpu/cpu.cfs_quota_us", "r")) {
            if (fscanf(f1, "%lld", &quota) != 1) quota = -1;
            fclose(f1);
            if (FILE* f2 = fopen("/sys/fs/cgroup/cpu/cpu.cfs_period_us", "r")) { if (fscanf(f2, "%lld", &period) != 1) period = 100000; fclose(f2); }
        }
        if (quota > 0 && period > 0) n = std::max(2, std::min(n, (int)((quota + period - 1) / period)));
        for (int i = 0; i < n; ++i) th_.emplace_back([this] { run(); });
    }
    void run()
    {
        for (;;) {
            std::function<void()> f;
            {
                std::unique_lock<std::mutex> lk(mu_);
                cv_.wait(lk, [this] { return !q_.empty(); });
                f = std::move(q_.front()); q_.pop_front();
            }
            f();
        }
    }
    std::mutex mu_; std::condition_variable cv_; std::deque<std::function<void()>> q_; std::vector<std::thread> th_;
};
}  // namespace

// n files decoded on the library's decoder pool (threads: at most that many of this call's images at a time; 0: no limit; 1: in the calling thread).
// dsts[i] receives image i; the first failure's status is returned and its message kept (the other images are still decoded).
extern "C" int omni_png_decode_batch(const void* const* datas, const size_t* nbytes, void* const* dsts, int n, int H, int W, int kind, int threads)
{
    if (n < 0 || (n > 0 && (!datas || !nbytes || !dsts))) OMNI_FAIL(OMNI_ERR_INVALID, "omni_png_decode_batch: null argument");
    if (kind != 0 && kind != 1) OMNI_FAIL(OMNI_ERR_INVALID, "omni_png_decode_batch: kind must be 0 or 1");
    if (n == 0) return OMNI_OK;
    struct Call {
        std::atomic<int> next{0}, status{OMNI_OK};
        int running = 0; bool closed = false;                       // (under mu) helpers inside work(); the caller has run out of images: later helpers return at once
        std::string first_error; std::mutex mu; std::condition_variable done;
    };
    // (shared with the helpers: a helper that the pool schedules after the caller has returned finds `closed` and touches nothing else)
    const std::shared_ptr<Call> callp = std::make_shared<Call>();
    Call& call = *callp;
    auto work = [&, callp]() {                                      // takes images until none is left (a pool thread, or the caller)
        for (;;) {
            const int i = callp->next.fetch_add(1);
            if (i >= n) return;
            int rc = (datas[i] && dsts[i]) ? png_decode((const unsigned char*)datas[i], nbytes[i], dsts[i], H, W, kind, nullptr) : OMNI_ERR_INVALID;
            if (rc != OMNI_OK) {
                std::lock_guard<std::mutex> lk(callp->mu);
                if (callp->status.load() == OMNI_OK) { callp->status.store(rc); callp->first_error = "image " + std::to_string(i) + ": " + omni_last_error(); }
            }
        }
    };
    int helpers = 0;
    if (threads != 1 && n > 1) {
        helpers = std::min(n, PngPool::get().size());
        if (threads > 1) helpers = std::min(helpers, threads);
        helpers -= 1;                                               // the calling thread decodes too
    }
    for (int t = 0; t < helpers; ++t)
        PngPool::get().submit([callp, &work] {
            {
                std::lock_guard<std::mutex> lk(callp->mu);
                if (callp->closed) return;                          // every image is taken (the caller may have returned: `work` is gone) — a busy pool costs the caller no wait
                ++callp->running;
            }
            work();                                                 // (`work` and what it references live on the caller's stack: the caller does not return while running > 0)
            std::lock_guard<std::mutex> lk(callp->mu);
            if (--callp->running == 0) callp->done.notify_all();
        });
    work();
    if (helpers > 0) {
        std::unique_lock<std::mutex> lk(call.mu);
        call.closed = true;                                         // next >= n here: a helper that starts now has nothing to take
        call.done.wait(lk, [&] { return call.running == 0; });
    }
    if (call.status.load() != OMNI_OK) omni_set_error("omni_png_decode_batch: " + call.first_error);
    return call.status.load();
}
