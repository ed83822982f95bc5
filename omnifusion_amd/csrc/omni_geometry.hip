// omni_geometry.hip — patch presets, geometry handles and the per-device handle cache.
//
// Replaces the per-call CPU work of the reference: the patch-centre tables
// (equi2pers_v3.py:32-47,52-84; pers2equi_v3.py:36-74), the ERP lat/lon grid
// (pers2equi_v3.py:109-111) and the ./grid/<layer_name>.pth cache (:24-29,155-167).
// Everything here is a constant of (nrows, fov, patch size, ERP size): a few KB that
// stay resident on the device.
#include <math.h>
#include <stdlib.h>
#include <string.h>
#include <mutex>
#include <vector>
#include <memory>

#include "omni_internal.h"

static thread_local std::string g_err;
void omni_set_error(const std::string& msg) { g_err = msg; }
extern "C" const char* omni_last_error(void) { return g_err.c_str(); }
extern "C" int omni_version(void) { return OMNI_VERSION; }

// ---------------------------------------------------------------- options
namespace {
struct OptName { const char* name; const char* env; int OmniOptions::* field; int dflt; };
const OptName kOpts[] = {
    {"conv_sh_tile", "OMNI_CONV_SH_TILE", &OmniOptions::conv_sh_tile, -1},
    {"conv_nohalo", "OMNI_CONV_NOHALO", &OmniOptions::conv_nohalo, 0},
    {"conv_halo_th", "OMNI_CONV_HALO_TH", &OmniOptions::conv_halo_th, 4},
    {"conv_big_blocks", "OMNI_CONV_BIG_BLOCKS", &OmniOptions::conv_big_blocks, 128},
    {"conv_halo_bn_lat", "OMNI_CONV_HALO_BN_LAT", &OmniOptions::conv_halo_bn_lat, 32},
    {"conv_deep_loaders", "OMNI_CONV_DEEP_LOADERS", &OmniOptions::conv_deep_loaders, 1},
    {"conv_halo_up2_bn_lat", "OMNI_CONV_HALO_UP2_BN_LAT", &OmniOptions::conv_halo_up2_bn_lat, 64},
    {"conv_halo_bn", "OMNI_CONV_HALO_BN", &OmniOptions::conv_halo_bn, 64},
    {"conv_img", "OMNI_CONV_IMG", &OmniOptions::conv_img, 1},
    {"conv_nodeep", "OMNI_CONV_NODEEP", &OmniOptions::conv_nodeep, 0},
    {"conv_pingpong", "OMNI_CONV_PINGPONG", &OmniOptions::conv_pingpong, 1},
    {"conv_noxcd", "OMNI_CONV_NOXCD", &OmniOptions::conv_noxcd, 0},
    {"conv_wt_major", "OMNI_CONV_WT_MAJOR", &OmniOptions::conv_wt_major, 1},
    {"conv_stem_pc", "OMNI_CONV_STEM_PC", &OmniOptions::conv_stem_pc, 1},
    {"conv_epi_lds", "OMNI_CONV_EPI_LDS", &OmniOptions::conv_epi_lds, 1},
    {"conv_up2_persist", "OMNI_CONV_UP2_PERSIST", &OmniOptions::conv_up2_persist, 1},
    {"splitk_max", "OMNI_SPLITK_MAX", &OmniOptions::splitk_max, 0},
    {"e2p_gather", "OMNI_E2P_GATHER", &OmniOptions::e2p_gather, 0},
    {"e2p_notab", "OMNI_E2P_NOTAB", &OmniOptions::e2p_notab, 0},
    {"e2p_verbose", "OMNI_E2P_VERBOSE", &OmniOptions::e2p_verbose, 0},
    {"e2p_bwd_simple", "OMNI_E2P_BWD_SIMPLE", &OmniOptions::e2p_bwd_simple, 0},
    {"p2e_bwd_simple", "OMNI_P2E_BWD_SIMPLE", &OmniOptions::p2e_bwd_simple, 0},
    {"bwd_table_mb", "OMNI_BWD_TABLE_MB", &OmniOptions::bwd_table_mb, 1024},
    {"bwd_wide", "OMNI_BWD_WIDE", &OmniOptions::bwd_wide, 1},
    {"bwd_lmax", "OMNI_BWD_LMAX", &OmniOptions::bwd_lmax, 0},
    {"bwd_chunk", "OMNI_BWD_CHUNK", &OmniOptions::bwd_chunk, 0},
    {"p2e_gather", "OMNI_P2E_GATHER", &OmniOptions::p2e_gather, 0},
    {"e2p_nbuf", "OMNI_E2P_NBUF", &OmniOptions::e2p_nbuf, 0},
    {"e2p_slot_kb", "OMNI_E2P_SLOT_KB", &OmniOptions::e2p_slot_kb, 6},
    {"e2p_tile_h", "OMNI_E2P_TILE_H", &OmniOptions::e2p_tile_h, 0},
    {"e2p_ref_lds", "OMNI_E2P_REF_LDS", &OmniOptions::e2p_ref_lds, 1},
    {"e2p_store", "OMNI_E2P_STORE", &OmniOptions::e2p_store, 1},
    {"e2p_slots", "OMNI_E2P_SLOTS", &OmniOptions::e2p_slots, 0},
    {"e2p_split", "OMNI_E2P_SPLIT", &OmniOptions::e2p_split, 0},
    {"e2p_full", "OMNI_E2P_FULL", &OmniOptions::e2p_full, -1},
    {"e2p_fb_planes", "OMNI_E2P_FB_PLANES", &OmniOptions::e2p_fb_planes, 0},
    {"e2p_fb_pos", "OMNI_E2P_FB_POS", &OmniOptions::e2p_fb_pos, 0},
    {"e2p_region", "OMNI_E2P_REGION", &OmniOptions::e2p_region, 0},
    {"p2e_band", "OMNI_P2E_BAND", &OmniOptions::p2e_band, 0},
    {"p2e_nbuf", "OMNI_P2E_NBUF", &OmniOptions::p2e_nbuf, 0},
    {"p2e_lds_kb", "OMNI_P2E_LDS_KB", &OmniOptions::p2e_lds_kb, 0},
    {"p2e_planes", "OMNI_P2E_PLANES", &OmniOptions::p2e_planes, 0},
    {"p2e_walk", "OMNI_P2E_WALK", &OmniOptions::p2e_walk, 1},
    {"p2e_tile8", "OMNI_P2E_TILE8", &OmniOptions::p2e_tile8, 1},
    {"p2e_store", "OMNI_P2E_STORE", &OmniOptions::p2e_store, 1},
    {"geom_cache_max", "OMNI_GEOM_CACHE_MAX", &OmniOptions::geom_cache_max, 16},
};
}  // namespace

OmniOptions& omni_options()
{
    static OmniOptions o = [] {                       // thread-safe one-time initialisation (C++11 magic static)
        OmniOptions v;
        for (const OptName& k : kOpts) {
            const char* e = getenv(k.env);
            v.*(k.field) = e ? (*e ? atoi(e) : 1) : k.dflt;
        }
        return v;
    }();
    return o;
}

int omni_num_cus()
{
    static int cus[64] = {0};                         // per device; a repeated query writes the same value
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return 256;
    if (cus[dev] == 0) {
        int n = 0;
        cus[dev] = (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && n > 0) ? n : 256;
    }
    return cus[dev];
}

extern "C" int omni_set_option(const char* name, int value)
{
    if (!name) OMNI_FAIL(OMNI_ERR_INVALID, "omni_set_option: null name");
    for (const OptName& k : kOpts)
        if (strcmp(k.name, name) == 0) { omni_options().*(k.field) = value; return OMNI_OK; }
    OMNI_FAIL(OMNI_ERR_INVALID, std::string("omni_set_option: unknown option '") + name + "'");
}

extern "C" int omni_get_option(const char* name, int* value)
{
    if (!name || !value) OMNI_FAIL(OMNI_ERR_INVALID, "omni_get_option: null argument");
    for (const OptName& k : kOpts)
        if (strcmp(k.name, name) == 0) { *value = omni_options().*(k.field); return OMNI_OK; }
    OMNI_FAIL(OMNI_ERR_INVALID, std::string("omni_get_option: unknown option '") + name + "'");
}

#ifdef OMNI_DEBUG_BUILD
int omni_debug_bits(const char* env_name) { const char* d = getenv(env_name); return d ? atoi(d) : 0; }
static long long* g_omni_trace = nullptr;
long long* omni_debug_trace_buf() { return g_omni_trace; }
// tools/: per-block time stamps of the resample kernels (bit 16 of OMNI_E2P_DBG / OMNI_P2E_DBG), 4 x int64 per block
extern "C" int omni_debug_set_trace(long long* buf) { g_omni_trace = buf; return OMNI_OK; }
#endif

// ---------------------------------------------------------------- presets
namespace {
struct Preset { int nr; const int* cols; const double* phis; };

bool preset(int nrows, int which, Preset& p)
{
    static const int c4[] = {3, 6, 6, 3};            static const double p4[] = {-67.5, -22.5, 22.5, 67.5};
    static const int c6[] = {3, 8, 12, 12, 8, 3};    static const double p6[] = {-75.2, -45.93, -15.72, 15.72, 45.93, 75.2};
    static const int c3[] = {3, 4, 3};               static const double p3e[] = {-60, 0, 60};
    static const double p3p[] = {-59.6, 0, 59.6};    // pers2equi_v3.py:47 (differs from equi2pers)
    static const int c5[] = {3, 6, 8, 6, 3};         static const double p5[] = {-72.2, -36.1, 0, 36.1, 72.2};
    switch (nrows) {
    case 4: p = {4, c4, p4}; return true;
    case 6: p = {6, c6, p6}; return true;
    case 3: p = {3, c3, which ? p3p : p3e}; return true;
    case 5: p = {5, c5, p5}; return true;
    default: return false;
    }
}

// fp32 centre angles exactly as the reference rounds them (equi2pers_v3.py:77-84).
int centers(int nrows, int which, float* lam0, float* phi1, float* center_p)
{
    Preset p;
    if (!preset(nrows, which, p)) return -1;
    const float PI_F = (float)M_PI, PI_2_F = (float)(M_PI * 0.5);
    int n = 0;
    for (int i = 0; i < p.nr; ++i)
        for (int j = 0; j < p.cols[i]; ++j) {
            double ti = 360.0 / p.cols[i];
            double tc = j * ti + ti / 2;
            float cx = (float)tc / 360.0f;
            float cy = ((float)p.phis[i] + 90.0f) / 180.0f;
            float px = cx * 2.0f - 1.0f, py = cy * 2.0f - 1.0f;
            if (center_p) { center_p[2 * n] = px; center_p[2 * n + 1] = py; }
            if (lam0) lam0[n] = px * PI_F;
            if (phi1) phi1[n] = py * PI_2_F;
            ++n;
        }
    return n;
}

void fill_tab(PatchTab& t, int nrows, int which)
{
    float lam[OMNI_MAX_PATCH], phi[OMNI_MAX_PATCH];
    t.N = centers(nrows, which, lam, phi, nullptr);
    for (int n = 0; n < t.N; ++n) {
        t.lam0[n] = lam[n];
        t.slam[n] = (float)sin((double)lam[n]); t.clam[n] = (float)cos((double)lam[n]);
        t.sphi[n] = (float)sin((double)phi[n]); t.cphi[n] = (float)cos((double)phi[n]);
    }
}

// torch.linspace fp32 semantics (two-sided evaluation), pers2equi_v3.py:109.
float linspace_f(float start, float end, int steps, int idx)
{
    if (steps == 1) return start;
    float step = (end - start) / (float)(steps - 1);
    int half = steps / 2;
    // ATen contracts both branches to a single FMA (checked bit-exact against torch.linspace)
    return idx < half ? fmaf(step, (float)idx, start) : fmaf(-step, (float)(steps - idx - 1), end);
}
}  // namespace

extern "C" int omni_num_patches(int nrows) { return centers(nrows, 0, nullptr, nullptr, nullptr); }

extern "C" int omni_patch_centers(int nrows, int which, float* center_p_host)
{
    if (!center_p_host) OMNI_FAIL(OMNI_ERR_INVALID, "omni_patch_centers: null output");
    int n = centers(nrows, which, nullptr, nullptr, center_p_host);
    if (n < 0) OMNI_FAIL(OMNI_ERR_INVALID, "unsupported nrows " + std::to_string(nrows) + " (presets: 3,4,5,6)");
    return OMNI_OK;
}

// ---------------------------------------------------------------- handles
extern "C" int omni_geometry_create(omni_geometry_t** out, int nrows, float fov_h, float fov_w,
                                    int ph, int pw, int H, int W, omni_stream_t stream_)
{
    hipStream_t stream = (hipStream_t)stream_;
    if (!out) OMNI_FAIL(OMNI_ERR_INVALID, "omni_geometry_create: null out");
    if (omni_num_patches(nrows) < 0)
        OMNI_FAIL(OMNI_ERR_INVALID, "unsupported nrows " + std::to_string(nrows) + " (presets: 3,4,5,6)");
    if (ph < 1 || pw < 1 || H < 0 || W < 0 || ph > 32768 || pw > 32768 || H > 32768 || W > 32768)
        OMNI_FAIL(OMNI_ERR_INVALID, "omni_geometry_create: bad patch/ERP size");
    if (!(fov_h > 0.f) || !(fov_w > 0.f)) OMNI_FAIL(OMNI_ERR_INVALID, "omni_geometry_create: fov must be > 0");

    std::unique_ptr<omni_geometry> g(new omni_geometry());
    OMNI_HIP(hipGetDevice(&g->device));
    g->nrows = nrows; g->fov_h = fov_h; g->fov_w = fov_w; g->ph = ph; g->pw = pw; g->H = H; g->W = W;
    fill_tab(g->e2p, nrows, 0);
    fill_tab(g->p2e, nrows, 1);
    g->N = g->e2p.N;
    centers(nrows, 0, nullptr, nullptr, g->center_p);
    g->row_trig = nullptr; g->col_trig = nullptr; g->cand = nullptr; g->ntx = (W + 63) / 64;
    g->e2p_fb_tiles = nullptr; g->e2p_nfb = 0; g->e2p_ixy = nullptr; g->e2p_ts = 32;
    for (auto& t : g->p2e_tiles) { t.ent = nullptr; t.ord = nullptr; t.walk = nullptr; t.nslots = 0; t.max_chunks = 0; t.max_cand = 0; t.ok = 0; t.sum_chunks = 0; }
    for (auto& t : g->e2p_boxes) { t.ent = nullptr; t.fb = nullptr; t.order = nullptr; t.norder = 0; t.nfb = 0; t.max_chunks = 0; t.ok = 0; t.tw = t.th = t.tx = t.ty = 0; }
    g->p2e_tx = g->p2e_ty = 0; g->pinned = 0;
    g->p2e_bwd_box = nullptr; g->p2e_rden = nullptr; g->p2e_btx = g->p2e_bty = g->p2e_bwd_ok = 0;
    g->p2e_bwd_ids = nullptr; g->p2e_bwd_nsmall = g->p2e_bwd_nbig = 0;
    g->e2p_bwd_box = nullptr; g->e2p_bwd_ids = nullptr; g->e2p_bwd_nsmall = g->e2p_bwd_nbig = g->e2p_gtx = g->e2p_gty = g->e2p_bwd_ok = 0;

    if (H > 0 && W > 0) {
        const float PI_F = (float)M_PI, PI_2_F = (float)(M_PI * 0.5);
        std::vector<float2> rt(H), ct(W);
        for (int i = 0; i < H; ++i) { double a = (double)linspace_f(-PI_2_F, PI_2_F, H, i); rt[i] = make_float2((float)sin(a), (float)cos(a)); }
        for (int j = 0; j < W; ++j) { double a = (double)linspace_f(-PI_F, PI_F, W, j);     ct[j] = make_float2((float)sin(a), (float)cos(a)); }
        OMNI_HIP(hipMalloc((void**)&g->row_trig, sizeof(float2) * H));
        OMNI_HIP(hipMalloc((void**)&g->col_trig, sizeof(float2) * W));
        OMNI_HIP(hipMalloc((void**)&g->cand, sizeof(unsigned long long) * (size_t)H * g->ntx));
        // synchronous copies: the staging vectors die at scope exit (one-time setup, never on the hot path)
        OMNI_HIP(hipMemcpy(g->row_trig, rt.data(), sizeof(float2) * H, hipMemcpyHostToDevice));
        OMNI_HIP(hipMemcpy(g->col_trig, ct.data(), sizeof(float2) * W, hipMemcpyHostToDevice));
        int rc = omni_p2e_build_candidates(g.get(), stream);
        if (rc == OMNI_OK) rc = omni_p2e_build_tiles(g.get(), stream);
        if (rc == OMNI_OK) rc = omni_e2p_build_tileflags(g.get(), stream);
        if (rc == OMNI_OK) rc = omni_e2p_build_boxes(g.get(), stream);
        if (rc != OMNI_OK) { omni_geometry_destroy(g.release()); return rc; }
    }
    *out = g.release();
    return OMNI_OK;
}

// The calling stream's scratch buffer of at least `bytes` (stream-ordered use: a buffer is only ever touched by launches on its own stream).
// A handle that has been used under stream capture keeps every buffer it ever handed out (a graph may hold the pointer).
int omni_bwd_workspace(omni_geometry* g, hipStream_t stream, size_t bytes, float** out)
{
    std::lock_guard<std::mutex> lk(g->ws_mu);
    omni_geometry::Ws* w = nullptr;
    for (auto& x : g->bwd_ws)
        if (x.stream == stream) {
            if (x.bytes >= bytes) { *out = x.ptr; return OMNI_OK; }
            w = &x;
        }
    hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
    if (stream && hipStreamIsCapturing(stream, &cs) == hipSuccess && cs == hipStreamCaptureStatusActive)
        OMNI_FAIL(OMNI_ERR_UNSUPPORTED, "backward: first use of a batch size on this geometry and stream while the stream is being captured (run the shape once before capturing)");
    if (!w || g->pinned) { g->bwd_ws.push_back({stream, nullptr, 0}); w = &g->bwd_ws.back(); }
    if (w->ptr) { OMNI_HIP(hipStreamSynchronize(stream)); OMNI_HIP(hipFree(w->ptr)); w->ptr = nullptr; w->bytes = 0; }
    OMNI_HIP(hipMalloc((void**)&w->ptr, bytes));
    w->bytes = bytes; *out = w->ptr;
    return OMNI_OK;
}

void omni_sp_free(OmniSpTable& t)
{
    if (t.ent) (void)hipFree(t.ent);
    if (t.slice_off) (void)hipFree(t.slice_off);
    if (t.cnt) (void)hipFree(t.cnt);
    if (t.long_ent) (void)hipFree(t.long_ent);
    if (t.long_off) (void)hipFree(t.long_off);
    if (t.long_row) (void)hipFree(t.long_row);
    t = OmniSpTable();
}

extern "C" void omni_geometry_destroy(omni_geometry_t* g)
{
    if (!g) return;
    if (g->row_trig) (void)hipFree(g->row_trig);
    if (g->col_trig) (void)hipFree(g->col_trig);
    if (g->cand) (void)hipFree(g->cand);
    for (auto& t : g->p2e_tiles) { if (t.ent) (void)hipFree(t.ent); if (t.ord) (void)hipFree(t.ord); if (t.walk) (void)hipFree(t.walk); }
    if (g->p2e_bwd_box) (void)hipFree(g->p2e_bwd_box);
    if (g->p2e_rden) (void)hipFree(g->p2e_rden);
    if (g->p2e_bwd_ids) (void)hipFree(g->p2e_bwd_ids);
    for (auto& t : g->e2p_boxes) { if (t.ent) (void)hipFree(t.ent); if (t.fb) (void)hipFree(t.fb); if (t.order) (void)hipFree(t.order); for (auto& w : t.work) (void)hipFree(w.dev); }
    if (g->e2p_fb_tiles) (void)hipFree(g->e2p_fb_tiles);
    if (g->e2p_ixy) (void)hipFree(g->e2p_ixy);
    if (g->e2p_bwd_box) (void)hipFree(g->e2p_bwd_box);
    if (g->e2p_bwd_ids) (void)hipFree(g->e2p_bwd_ids);
    omni_sp_free(g->p2e_sp); omni_sp_free(g->e2p_sp);
    for (auto& w : g->bwd_ws) if (w.ptr) (void)hipFree(w.ptr);
    delete g;
}

// ---------------------------------------------------------------- cache
// LRU over (device, nrows, fov, patch size, ERP size), capped at OmniOptions::geom_cache_max handles (an entry holds the
// device tables of its configuration, e.g. the 8 B/sample equi2pers coordinate table: inputs of ever-changing size must not
// leak HBM).  A handle is destroyed only after its device has drained (in-flight launches may still read its tables); a
// hit moves the entry to the front, so a handle in use is never the eviction candidate unless >= cap OTHER configurations
// were created in between.
namespace {
std::mutex g_mu;
std::vector<omni_geometry*> g_cache;           // front = most recently used

void destroy_drained(omni_geometry* g)
{
    int cur = 0;
    if (hipGetDevice(&cur) == hipSuccess) {
        if (cur != g->device) (void)hipSetDevice(g->device);
        (void)hipDeviceSynchronize();
        if (cur != g->device) (void)hipSetDevice(cur);
    }
    omni_geometry_destroy(g);
}
}

int omni_geometry_lookup(const omni_geometry** out, int nrows, float fov_h, float fov_w,
                         int ph, int pw, int H, int W, hipStream_t stream)
{
    int dev = 0;
    OMNI_HIP(hipGetDevice(&dev));
    std::lock_guard<std::mutex> lk(g_mu);      // replica threads (nn.DataParallel) may race here
    for (size_t i = 0; i < g_cache.size(); ++i) {
        omni_geometry* g = g_cache[i];
        if (g->device == dev && g->nrows == nrows && g->fov_h == fov_h && g->fov_w == fov_w &&
            g->ph == ph && g->pw == pw && g->H == H && g->W == W) {
            if (i) { g_cache.erase(g_cache.begin() + i); g_cache.insert(g_cache.begin(), g); }
            // a launch that is being CAPTURED bakes this handle's table pointers into a hipGraph that may be replayed at any later time:
            // such a handle is never evicted (a replay after >= geom_cache_max other shapes read freed memory)
            hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
            if (stream && hipStreamIsCapturing(stream, &cs) == hipSuccess && cs == hipStreamCaptureStatusActive) g->pinned = 1;
            *out = g; return OMNI_OK;
        }
    }
    {   // a new shape under capture cannot be built (allocations, synchronous copies): the caller warms every shape up before capturing
        hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
        if (stream && hipStreamIsCapturing(stream, &cs) == hipSuccess && cs == hipStreamCaptureStatusActive)
            OMNI_FAIL(OMNI_ERR_UNSUPPORTED, "omni_geometry_lookup: first use of a geometry while its stream is being captured (run the shape once before capturing)");
    }
    omni_geometry* g = nullptr;
    int rc = omni_geometry_create(&g, nrows, fov_h, fov_w, ph, pw, H, W, stream);
    if (rc != OMNI_OK) return rc;
    g_cache.insert(g_cache.begin(), g);
    const size_t cap = (size_t)(omni_options().geom_cache_max > 0 ? omni_options().geom_cache_max : 1);
    // evict the least recently used handle that no graph holds (pinned handles stay: the cache may then exceed its cap by their number)
    for (size_t i = g_cache.size(); i-- > 1 && g_cache.size() > cap;)
        if (!g_cache[i]->pinned) { destroy_drained(g_cache[i]); g_cache.erase(g_cache.begin() + i); }
    *out = g;
    return OMNI_OK;
}

extern "C" int omni_geometry_cache_size(void)
{
    std::lock_guard<std::mutex> lk(g_mu);
    return (int)g_cache.size();
}

extern "C" void omni_geometry_cache_clear(void)
{
    std::lock_guard<std::mutex> lk(g_mu);
    for (omni_geometry* g : g_cache) destroy_drained(g);
    g_cache.clear();
}
