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
// conversions listed above.  Inflate is zlib's (the image ships libz; the reference's OpenCV links the same library through libpng);
// everything else — chunk walk, CRC check, un-filtering, sample conversion, the worker pool — is here.
// Supported: colour types 0 (gray), 2 (RGB), 3 (palette), 4 (gray + alpha), 6 (RGB + alpha) at 8 bits, types 0 / 2 / 4 / 6 at 16 bits; no
// interlacing (Adam7) and no 1/2/4-bit samples (OMNI_ERR_UNSUPPORTED; neither occurs in the dataset).
#include <zlib.h>
#include <string.h>
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

namespace {

struct PngInfo { int w, h, depth, ctype, interlace; };

inline unsigned be32(const unsigned char* p) { return ((unsigned)p[0] << 24) | ((unsigned)p[1] << 16) | ((unsigned)p[2] << 8) | (unsigned)p[3]; }

int channels_of(int ctype) { return ctype == 0 ? 1 : ctype == 2 ? 3 : ctype == 3 ? 1 : ctype == 4 ? 2 : ctype == 6 ? 4 : 0; }

// walks the chunks: fills `info`, the palette, and inflates the concatenated IDAT payloads into `raw` (filter byte + samples per scan line)
// (expect_w / expect_h > 0: the destination's size — a header that announces anything else is refused BEFORE a byte is allocated for it)
int png_unpack(const unsigned char* d, size_t n, PngInfo& info, std::vector<unsigned char>& raw, unsigned char (*pal)[3], int* npal, bool info_only,
               int expect_w = 0, int expect_h = 0)
{
    static const unsigned char sig[8] = {0x89, 'P', 'N', 'G', 0x0d, 0x0a, 0x1a, 0x0a};
    if (n < 8 + 25 || memcmp(d, sig, 8) != 0) OMNI_FAIL(OMNI_ERR_INVALID, "png: not a PNG stream");
    size_t pos = 8;
    bool have_ihdr = false, done = false, inflating = false;
    z_stream zs;
    memset(&zs, 0, sizeof(zs));
    size_t rawsize = 0;
    int rc = OMNI_OK;
    while (pos + 12 <= n && !done) {
        const unsigned len = be32(d + pos);
        const unsigned char* type = d + pos + 4;
        const unsigned char* body = d + pos + 8;
        if ((size_t)len > n - pos - 12) { rc = OMNI_ERR_INVALID; omni_set_error("png: truncated chunk"); break; }
        if (be32(body + len) != (unsigned)crc32(crc32(0L, Z_NULL, 0), type, len + 4)) { rc = OMNI_ERR_INVALID; omni_set_error("png: chunk CRC mismatch"); break; }
        if (!memcmp(type, "IHDR", 4)) {
            if (have_ihdr) { rc = OMNI_ERR_INVALID; omni_set_error("png: a second IHDR chunk"); break; }
            if (len != 13) { rc = OMNI_ERR_INVALID; omni_set_error("png: bad IHDR"); break; }
            info.w = (int)be32(body); info.h = (int)be32(body + 4); info.depth = body[8]; info.ctype = body[9]; info.interlace = body[12];
            have_ihdr = true;
            if (info.w <= 0 || info.h <= 0 || channels_of(info.ctype) == 0 || body[10] != 0 || body[11] != 0) { rc = OMNI_ERR_INVALID; omni_set_error("png: bad IHDR fields"); break; }
            if (info_only) { done = true; break; }
            if (info.interlace != 0) { rc = OMNI_ERR_UNSUPPORTED; omni_set_error("png: interlaced (Adam7) files are not supported"); break; }
            if (!(info.depth == 8 || (info.depth == 16 && info.ctype != 3))) { rc = OMNI_ERR_UNSUPPORTED; omni_set_error("png: only 8- and 16-bit samples are supported"); break; }
            // the header is untrusted input: the sizes are checked (against the destination, and against 2 GiB in 64-bit arithmetic —
            // w, h < 2^31 and <= 8 bytes per pixel cannot wrap it) before anything is allocated for them
            if ((expect_w > 0 && info.w != expect_w) || (expect_h > 0 && info.h != expect_h)) {
                rc = OMNI_ERR_INVALID;
                omni_set_error("png: the image is " + std::to_string(info.h) + " x " + std::to_string(info.w) + ", the destination " + std::to_string(expect_h) + " x " + std::to_string(expect_w));
                break;
            }
            const unsigned long long stride = (unsigned long long)info.w * channels_of(info.ctype) * (info.depth / 8);
            if (stride + 1 > 0x7fffffffull || (stride + 1) * (unsigned long long)info.h > 0x7fffffffull) { rc = OMNI_ERR_UNSUPPORTED; omni_set_error("png: image too large (2 GiB of samples or more)"); break; }
            rawsize = (size_t)((stride + 1) * (unsigned long long)info.h);
            if (raw.size() < rawsize) raw.resize(rawsize);          // (grows only: the caller may keep it across images; inflate overwrites every byte it reports)
            if (inflateInit(&zs) != Z_OK) { rc = OMNI_ERR_HIP; omni_set_error("png: inflateInit failed"); break; }
            inflating = true;
            zs.next_out = raw.data(); zs.avail_out = (uInt)rawsize;
        } else if (!have_ihdr) {
            rc = OMNI_ERR_INVALID; omni_set_error("png: IHDR is not the first chunk"); break;
        } else if (!memcmp(type, "PLTE", 4)) {
            if (len % 3 != 0 || len > 768) { rc = OMNI_ERR_INVALID; omni_set_error("png: bad PLTE"); break; }
            *npal = (int)(len / 3);
            memcpy(pal, body, len);
        } else if (!memcmp(type, "IDAT", 4)) {
            if (len == 0) { pos += 12; continue; }                 // an empty IDAT is legal (cv2.imread reads such files): nothing to inflate
            zs.next_in = const_cast<unsigned char*>(body); zs.avail_in = len;
            const int zr = inflate(&zs, Z_NO_FLUSH);
            if (zr != Z_OK && zr != Z_STREAM_END) { rc = OMNI_ERR_INVALID; omni_set_error("png: corrupt zlib stream"); break; }
            if (zs.avail_in != 0 && zr != Z_STREAM_END) { rc = OMNI_ERR_INVALID; omni_set_error("png: more image data than the header announces"); break; }
        } else if (!memcmp(type, "IEND", 4)) {
            done = true;
        } else if (!(type[0] & 0x20)) {                            // an unknown CRITICAL chunk
            rc = OMNI_ERR_UNSUPPORTED; omni_set_error("png: unknown critical chunk"); break;
        }
        pos += 12 + (size_t)len;
    }
    if (inflating) {
        if (rc == OMNI_OK && zs.total_out != rawsize) { rc = OMNI_ERR_INVALID; omni_set_error("png: image data ends early"); }
        inflateEnd(&zs);
    }
    if (rc == OMNI_OK && !have_ihdr) { rc = OMNI_ERR_INVALID; omni_set_error("png: no IHDR"); }
    if (rc == OMNI_OK && !info_only && !done) { rc = OMNI_ERR_INVALID; omni_set_error("png: no IEND"); }
    return rc;
}

inline int paeth(int a, int b, int c)
{
    const int p = a + b - c, pa = abs(p - a), pb = abs(p - b), pc = abs(p - c);
    return (pa <= pb && pa <= pc) ? a : (pb <= pc ? b : c);
}

// in-place reconstruction of the scan lines (PNG spec 9.2): raw = h x (1 + stride); bpp = bytes per complete pixel
int png_unfilter(std::vector<unsigned char>& raw, int h, size_t stride, int bpp)
{
    const unsigned char* prev = nullptr;
    for (int y = 0; y < h; ++y) {
        unsigned char* row = raw.data() + (size_t)y * (stride + 1);
        const int ft = row[0];
        unsigned char* x = row + 1;
        switch (ft) {
        case 0: break;
        case 1: for (size_t i = bpp; i < stride; ++i) x[i] = (unsigned char)(x[i] + x[i - bpp]); break;
        case 2: if (prev) for (size_t i = 0; i < stride; ++i) x[i] = (unsigned char)(x[i] + prev[i]); break;
        case 3:
            for (size_t i = 0; i < stride; ++i) {
                const int a = i >= (size_t)bpp ? x[i - bpp] : 0, b = prev ? prev[i] : 0;
                x[i] = (unsigned char)(x[i] + ((a + b) >> 1));
            }
            break;
        case 4:
            for (size_t i = 0; i < stride; ++i) {
                const int a = i >= (size_t)bpp ? x[i - bpp] : 0, b = prev ? prev[i] : 0, c = (prev && i >= (size_t)bpp) ? prev[i - bpp] : 0;
                x[i] = (unsigned char)(x[i] + paeth(a, b, c));
            }
            break;
        default: OMNI_FAIL(OMNI_ERR_INVALID, "png: unknown filter type");
        }
        prev = x;
    }
    return OMNI_OK;
}

// kind 0: cv2.imread(path) -> uint8 BGR [H,W,3];  kind 1: cv2.imread(path, -1) of a single-channel file -> uint8 / uint16 [H,W] in host byte order
int png_decode_body(const unsigned char* d, size_t n, void* dst, int H, int W, int kind, int* elem_bytes)
{
    if (H <= 0 || W <= 0) OMNI_FAIL(OMNI_ERR_INVALID, "png: the destination must have a positive size");
    PngInfo info{};
    // scratch buffers are kept between images (a free list: the decoder threads of a batch call are short-lived): a fresh 1.5-25 MB vector per image
    // is an mmap + page faults + a zero fill each time, and with 64 decoder threads the kernel's address-space lock serialises them (round 5: 216 MB/s
    // per thread on 8 threads, 80 MB/s on 64)
    struct Lease {
        std::vector<unsigned char> buf;
        Lease() { std::lock_guard<std::mutex> lk(mu()); if (!pool().empty()) { buf.swap(pool().back()); pool().pop_back(); } }
        ~Lease() { std::lock_guard<std::mutex> lk(mu()); if (pool().size() < 256 && buf.capacity() <= (64u << 20)) pool().emplace_back(std::move(buf)); }
        static std::mutex& mu() { static std::mutex m; return m; }
        static std::vector<std::vector<unsigned char>>& pool() { static std::vector<std::vector<unsigned char>> p; return p; }
    } lease;
    std::vector<unsigned char>& raw = lease.buf;
    unsigned char pal[256][3];
    int npal = 0;
    int rc = png_unpack(d, n, info, raw, pal, &npal, false, W, H);
    if (rc != OMNI_OK) return rc;
    const int ch = channels_of(info.ctype), bs = info.depth / 8, bpp = ch * bs;
    const size_t stride = (size_t)W * bpp;
    rc = png_unfilter(raw, H, stride, bpp);
    if (rc != OMNI_OK) return rc;
    if (kind == 0) {
        unsigned char* o = (unsigned char*)dst;
        for (int y = 0; y < H; ++y) {
            const unsigned char* s = raw.data() + (size_t)y * (stride + 1) + 1;
            unsigned char* q = o + (size_t)y * W * 3;
            for (int x = 0; x < W; ++x, s += bpp, q += 3) {
                unsigned char r, g, b;                              // 16-bit samples: the high byte (OpenCV's IMREAD_COLOR scales 16 -> 8 bits by >> 8)
                if (info.ctype == 3) { const int i = s[0]; if (i >= npal) OMNI_FAIL(OMNI_ERR_INVALID, "png: palette index out of range"); r = pal[i][0]; g = pal[i][1]; b = pal[i][2]; }
                else if (ch <= 2) { r = g = b = s[0]; }
                else { r = s[0]; g = s[bs]; b = s[2 * bs]; }
                q[0] = b; q[1] = g; q[2] = r;
            }
        }
        if (elem_bytes) *elem_bytes = 1;
        return OMNI_OK;
    }
    if (ch != 1 || info.ctype == 3) OMNI_FAIL(OMNI_ERR_UNSUPPORTED, "png: the unchanged (-1) read is implemented for single-channel gray files (depth maps)");
    if (elem_bytes) *elem_bytes = bs;
    for (int y = 0; y < H; ++y) {
        const unsigned char* s = raw.data() + (size_t)y * (stride + 1) + 1;
        if (bs == 1) memcpy((unsigned char*)dst + (size_t)y * W, s, W);
        else {
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
    try { rc = png_unpack((const unsigned char*)data, nbytes, info, raw, nullptr, &npal, true); }
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

// A process-wide pool of decoder threads, started on first use: a batch call used to create (and join) its own std::threads — with several batch
// decoders side by side that was ~18 000 thread creations per second, each an 8-MB stack mmap / munmap under the process's address-space lock, and the
// whole loader levelled off at 2300 panoramas/s whatever the number of decoders (round 5, tools/png_fed_ab.py).
namespace {
class PngPool {
public:
    static PngPool& get() { static PngPool p; return p; }
    void submit(std::function<void()> f) { { std::lock_guard<std::mutex> lk(mu_); q_.push_back(std::move(f)); } cv_.notify_one(); }
    int size() const { return (int)th_.size(); }
private:
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
    ~PngPool()
    {
        { std::lock_guard<std::mutex> lk(mu_); stop_ = true; }
        cv_.notify_all();
        for (auto& t : th_) t.join();
    }
    void run()
    {
        for (;;) {
            std::function<void()> f;
            {
                std::unique_lock<std::mutex> lk(mu_);
                cv_.wait(lk, [this] { return stop_ || !q_.empty(); });
                if (stop_ && q_.empty()) return;
                f = std::move(q_.front()); q_.pop_front();
            }
            f();
        }
    }
    std::mutex mu_; std::condition_variable cv_; std::deque<std::function<void()>> q_; std::vector<std::thread> th_; bool stop_ = false;
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
        std::atomic<int> next{0}, left{0}, status{OMNI_OK};
        std::string first_error; std::mutex mu; std::condition_variable done;
    };
    // (shared with the helpers: the last helper may still be inside its unlock / notify when the caller wakes up and returns)
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
    call.left.store(helpers);
    for (int t = 0; t < helpers; ++t)
        PngPool::get().submit([callp, &work] {
            work();                                                 // (`work` and what it references live on the caller's stack: the caller does not return before left == 0)
            std::lock_guard<std::mutex> lk(callp->mu);
            if (callp->left.fetch_sub(1) == 1) callp->done.notify_all();
        });
    work();
    if (helpers > 0) {
        std::unique_lock<std::mutex> lk(call.mu);
        call.done.wait(lk, [&] { return call.left.load() == 0; });
    }
    if (call.status.load() != OMNI_OK) omni_set_error("omni_png_decode_batch: " + call.first_error);
    return call.status.load();
}
