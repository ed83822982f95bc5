// omni_internal.h — shared host/device declarations of libomnifusion_hip.so (gfx950 only).
#pragma once
#include <mutex>
#include <hip/hip_runtime.h>
#include <hip/hip_fp16.h>
#include <stdint.h>
#include <string>
#include <vector>

#include "../../include/omnifusion.h"

#define OMNI_MAX_PATCH 46          // nrows=6 preset
#define OMNI_WAVE 64

// ---------------------------------------------------------------- error plumbing
void omni_set_error(const std::string& msg);
#define OMNI_FAIL(code, msg) do { omni_set_error(msg); return (code); } while (0)
#define OMNI_HIP(expr) do { hipError_t e_ = (expr); if (e_ != hipSuccess) {                   \
        omni_set_error(std::string(#expr) + ": " + hipGetErrorString(e_)); return OMNI_ERR_HIP; } } while (0)

// ---------------------------------------------------------------- tuning options
// Read ONCE from the environment (OMNI_* variables, DESIGN.md "Environment switches") the first time the library needs them;
// `omni_set_option()` (include/omnifusion.h) changes one at run time (tests and tools sweep tile shapes with it).  No option
// changes a result bit except the three that change the K summation ORDER of a convolution: `splitk_max`, `conv_img` (16-wide images on
// the halo kernel: K runs (tap, group) per channel group instead of im2col order) and the latency form a lone panorama selects (fmt bit 2
// of omni_conv2d_sh_f16x3_ws) — results then agree to ~2e-5, not bit for bit (tests assert exactly that).  Launch paths read the struct, never getenv().
struct OmniOptions {
    int conv_sh_tile;     // OMNI_CONV_SH_TILE   -1 auto (8-wave 256x128 / 128x128 / 128x64 where >= 128 blocks remain) | 0 64x64 | 1 128x64 | 2 128x128 | 3, 4 the 8-wave forms | 5..7 auto without 256x128 (64 / 128 / 256 blocks) | 8 = auto | 9 = auto without the loader waves
    int conv_nohalo;      // OMNI_CONV_NOHALO    1: never take the halo-reuse 3x3 kernel
    int conv_halo_th;     // OMNI_CONV_HALO_TH   rows per halo block: 4 (default) | 8
    int conv_halo_bn_lat; // OMNI_CONV_HALO_BN_LAT  the same choice for the LATENCY form (fmt bit 2: one panorama per forward, a fraction of a block per CU): 32 (default: 0.956 -> 0.942 ms per forward) | 64
    int conv_halo_up2_bn_lat; // OMNI_CONV_HALO_UP2_BN_LAT  ... and for the up-sampling halo kernels in the latency form: 64 | 32 (twice the halo arithmetic, half the block life)
    int conv_deep_loaders; // OMNI_CONV_DEEP_LOADERS  1 (default, round 6: one panorama 0.966 -> 0.930 ms per forward, batches unchanged): the 64 x 64 six-stage tile kernel of single-round launches (one panorama; the transformer GEMMs) with four loader waves | 0
    int conv_big_blocks;  // OMNI_CONV_BIG_BLOCKS  least number of 256 x 128 blocks a launch must have to take that tile (default 128; the chip is power-bound in the timed region — profiles/r06h_clocks_under_load.txt — so fewer LDS / L2 bytes per matrix instruction may beat filling more CUs)
    int conv_halo_bn;     // OMNI_CONV_HALO_BN   output channels per block of the 3x3 halo kernels that COPY their halo (not the up-sampling ones): 64 (default: 74 KiB of LDS, 244 registers -> two blocks per CU) | 32 (50 KiB, 164 registers -> THREE blocks per CU; round 6: each kernel ALONE 11-15 % faster — layer1 46.9 -> 41.8 us, layer2 47.4 -> 40.1, de_conv1_0 44.5 -> 39.3 — but whole forwards equal (plain) or 1 % slower (three in flight: 3992 -> 3955 panoramas/s): twice the halo fetches; profiles/r06e_halo_bn.txt); same bits
    int conv_img;         // OMNI_CONV_IMG       1 (default): 3x3 stride-1 convolutions of 16-pixel-wide images on the halo kernel (bands of whole rows) when the launch has >= 256 blocks and no split-K | 2: 8-wide too | 0: im2col tiles
    int conv_nodeep;      // OMNI_CONV_NODEEP    1: 3 pipeline stages even for single-round launches
    int conv_pingpong;    // OMNI_CONV_PINGPONG  1 (default): the 8 + 4-wave tile kernels run the two matrix waves of a SIMD in anti-phase (conv_sh_kernel<.., PP>; same bits) | 0: one barrier per K-step
    int conv_noxcd;       // OMNI_CONV_NOXCD     1: identity block order (no XCD-aware remap)
    int conv_wt_major;    // OMNI_CONV_WT_MAJOR  block order of the tile kernel: 1 (default) weight-stationary per XCD where the weights are the larger operand | 0 never | 2 always
    int conv_stem_pc;     // OMNI_CONV_STEM_PC   1 (default): the stem with producer / consumer waves (8 waves, two image buffers: +2 % one forward at a time, nothing pipelined) | 0: 4 waves
    int conv_epi_lds;     // OMNI_CONV_EPI_LDS   1 (default): split-half outputs of the convolution kernels leave through an LDS transposition, 16 bytes per lane | 0: 8 bytes per lane straight from the accumulators
    int conv_up2_persist; // OMNI_CONV_UP2_PERSIST 1 (default): conv3x3(up2(x)) with 32 -> 32 channels (de_conv4_0) on the persistent kernel with resident weights | 0: the halo kernel
    int splitk_max;       // OMNI_SPLITK_MAX     cap of the split-K plan (0 = none)
    int e2p_gather;       // OMNI_E2P_GATHER     1: equi2pers always takes the direct-gather kernel (no LDS staging)
    int e2p_notab;        // OMNI_E2P_NOTAB      1: no per-geometry sampling-coordinate table
    int e2p_verbose;      // OMNI_E2P_VERBOSE    1: print tile statistics when a geometry handle is built
    int e2p_bwd_simple;   // OMNI_E2P_BWD_SIMPLE 0 (default) | 4: the sparse-matrix gather (omni_spgather.h; no atomics, fixed summation order) | 1: plain scatter | 2: LDS boxes + global atomics | 3: ERP-tile gathers with LDS atomics
    int p2e_bwd_simple;   // OMNI_P2E_BWD_SIMPLE 0 (default): the sparse-matrix gather | 1: global atomics (the round-1 kernel) | 2: patch-tile gathers with LDS atomics
    int p2e_gather;       // OMNI_P2E_GATHER     1: pers2equi always takes the direct-gather kernel (no LDS staging) | 2: never (not even for ONE plane of a large ERP)
    int e2p_nbuf;         // OMNI_E2P_NBUF       LDS ring slots (boxes in flight) per wave of the equi2pers LDS kernel: 0 auto | 1 | 2 | 4
    int e2p_store;        // OMNI_E2P_STORE      patch stores of the equi2pers box kernel: 0 plain | 1 non-temporal (default: 106 -> 64-74 us at 16 panoramas, whose 327 MB per launch exceed the 256-MB memory-side cache)
    int e2p_slots;        // OMNI_E2P_SLOTS      wave slots per CU the equi2pers work table plans for (0: 12, what the 12-KiB ring admits)
    int e2p_split;        // OMNI_E2P_SPLIT      plane ranges a tile beyond the first round of slots is cut into (0: 3)
    int e2p_full;         // OMNI_E2P_FULL       -1: one whole tile per slot first; else that percentage of the slots
    int e2p_fb_planes;    // OMNI_E2P_FB_PLANES  planes per gather block of a pole tile (0: 12)
    int e2p_region;       // OMNI_E2P_REGION     which tiles share an XCD: 0 ERP sectors (4 longitudes x 2 hemispheres) | 1 eight latitude bands of equal cost
    int e2p_fb_pos;       // OMNI_E2P_FB_POS     where the gather blocks go: 0 / 3 first | 1 behind the whole tiles | 2 behind the first plane range of the cut tiles | 4 last
    int e2p_ref_lds;      // OMNI_E2P_REF_LDS    1 (default): the reference layout [B,C,ph,pw,N] by the LDS-staged kernel (e2p_ref_kernel) where a block fits the CU | 0: gathers (e2p_reflayout_kernel)
    int e2p_tile_h;       // OMNI_E2P_TILE_H     sample rows per tile of the equi2pers box kernel: 0 / 8 (default) | 4 | 2: 4 where > 1/8 of the 8-row tiles would gather (measured: wins only at 8 panoramas of 128^2 patches)
    int e2p_slot_kb;      // OMNI_E2P_SLOT_KB    largest tap box staged in LDS (KiB, 1..8; default 6); tiles with a larger box take the gather path
    int p2e_band;         // OMNI_P2E_BAND       tile rows per XCD band of the pers2equi block order (0: max(1, tile rows / 8) — ONE contiguous range of tile rows per XCD)
    int p2e_lds_kb;       // OMNI_P2E_LDS_KB     0 (default): the ring | n: at least n KiB of LDS per wave of the pers2equi LDS / walk kernels — caps the waves resident per CU at 160 / n (tuning: a launch whose waves are all resident at once marches in step; fewer resident waves are back-filled out of phase)
    int p2e_nbuf;         // OMNI_P2E_NBUF       LDS ring slots (boxes in flight) per wave of the pers2equi LDS kernel: 0 auto | 1 | 2 | 4
    int p2e_planes;       // OMNI_P2E_PLANES     cap of the image planes per wave of the pers2equi LDS kernel: 0 auto (8) | 1 | 2 | 4 | 8
    int p2e_store;        // OMNI_P2E_STORE      ERP stores of the pers2equi LDS kernels: 1 (default) non-temporal | 0 plain
    int p2e_tile8;        // OMNI_P2E_TILE8      1 (default): ONE plane per wave (a lone depth map) on 8 x 32 ERP tiles, four pixels per lane | 0: 4 x 32 like every other form (same bits)
    int p2e_walk;         // OMNI_P2E_WALK       1 (default): the flat-pipeline kernel p2e_walk_kernel (one stage stream per tile across its patches, stages consumed in pairs) | 0: p2e_lds_kernel (patch by patch)
    int bwd_wide;         // OMNI_BWD_WIDE       1 (default): the backward gathers read a plane-interleaved copy of the gradient (16-byte gathers) | 0: the gradient itself (4-byte gathers, no scratch)
    int bwd_chunk;        // OMNI_BWD_CHUNK      blocks per XCD chunk of the backward gathers (0: 16; 4 .. 64 measured within 3 %)
    int bwd_lmax;         // OMNI_BWD_LMAX       rows with more entries than this go to the long-row list (0: OMNI_SP_LMAX; read when a table is built)
    int bwd_table_mb;     // OMNI_BWD_TABLE_MB   largest sparse-matrix table of a backward operator kept per geometry, MiB (default 1024; a geometry past it keeps the tile kernels)
    int geom_cache_max;   // OMNI_GEOM_CACHE_MAX geometry handles kept per process (LRU), default 16
};
OmniOptions& omni_options();
int omni_num_cus();                // compute units of the current device (cached per device)

// Ablation bits that change RESULTS (skip the epilogue, suppress loads ...) exist only in the debug build of the library
// (python -m omnifusion_amd.build --debug -> libomnifusion_hip_dbg.so, -DOMNI_DEBUG_BUILD); the product never carries them.
#ifdef OMNI_DEBUG_BUILD
#define OMNI_DBG(a, bit) (((a).dbg & (bit)) != 0)
int omni_debug_bits(const char* env_name);
long long* omni_debug_trace_buf();   // device buffer set by omni_debug_set_trace (4 x int64 per block: start, set-up done, end, HW_ID | XCC_ID << 32), or null
#else
#define OMNI_DBG(a, bit) false
#endif

// ---------------------------------------------------------------- geometry constants
// Per-patch constants, passed to kernels BY VALUE (kernarg segment -> scalar loads,
// the patch index is wave-uniform everywhere).  Trig of the fp32 centre angles is
// evaluated on the host in double and rounded once.
struct PatchTab {
    int   N;
    float lam0[OMNI_MAX_PATCH];    // fp32 centre longitude  (equi2pers_v3.py:83)
    float slam[OMNI_MAX_PATCH];
    float clam[OMNI_MAX_PATCH];
    float sphi[OMNI_MAX_PATCH];    // sin / cos of the fp32 centre latitude (:84)
    float cphi[OMNI_MAX_PATCH];
};

// A linear operator of the geometry as a sparse matrix in sliced-ELL form (omni_spgather.h): rows in slices of 64 (one per lane of a wave), slice s
// holds K_s = slice_off[s+1] - slice_off[s] entries per row, entry k of row r at ent[(slice_off[s] * 64) + k * 64 + (r & 63)]; cnt[r] of them are
// real.  Rows with more than OMNI_SP_LMAX entries (patch pixels at a pole: a whole ERP row maps onto them) live in a CSR side list instead
// (cnt[r] = -1) and get a wave each.  Entries of a row are sorted by source index: the summation order is a constant of the geometry.
constexpr int OMNI_SP_LMAX = 48;   // (24 .. 64 measured: equi2pers^T has no row past 48 and gains 12 % from keeping them all in the slices; pers2equi^T is flat)
struct OmniSpTable {
    uint2* ent = nullptr; int* slice_off = nullptr; int* cnt = nullptr;
    uint2* long_ent = nullptr; int* long_off = nullptr; int* long_row = nullptr;
    int nrows = 0, nslices = 0, nlong = 0, ok = 0;
    long long nent = 0, npadded = 0, nlong_ent = 0;
};
void omni_sp_free(OmniSpTable& t);
struct omni_geometry;
int omni_bwd_workspace(omni_geometry* g, hipStream_t stream, size_t bytes, float** out);    // omni_geometry.hip

struct omni_geometry {
    int device;
    int pinned;                    // used by a launch under stream capture: a hipGraph holds its table pointers, the LRU never evicts it
    int nrows, N;
    float fov_h, fov_w;
    int ph, pw, H, W;
    PatchTab e2p;                  // equi2pers centre table
    PatchTab p2e;                  // pers2equi centre table (differs for nrows=3)
    float center_p[2 * OMNI_MAX_PATCH];
    // device-resident tables (a few KB)
    float2* row_trig;              // [H]  (sin lat_i, cos lat_i)   pers2equi_v3.py:109
    float2* col_trig;              // [W]  (sin lon_j, cos lon_j)
    unsigned long long* cand;      // [H][ntx] bit n set <=> patch n covers >=1 pixel of the 64-px tile
    int ntx;
    // pers2equi LDS path: per ERP tile (P2E_TH x P2E_TW pixels) the list of covering patches with the bounding box of their
    // bilinear taps inside the patch (omni_pers2equi.hip); index 0: 4-byte elements, 1: 2-byte elements (16-byte chunk alignment)
    struct P2ETiles { uint2* ent; int max_chunks; int max_cand; int ok; int sum_chunks; uint2* ord; int nslots; unsigned char* walk; int nslots_walk; } p2e_tiles[4];   // ord: the table in block order (+ tile id), what the kernels read; [2], [3]: the 8-row tile sets of the one-plane walk kernel (walk table only)
    int p2e_tx, p2e_ty;            // tiles per ERP row / column
    // pers2equi backward by gathers (omni_pers2equi.hip): per (patch, 4 x 32 patch tile) the ERP box of the pixels whose taps touch it
    // (columns relative to the patch's centre column: the box may cross the +-pi seam), and 1 / (L1 norm of the tap weights) per ERP pixel
    int4* p2e_bwd_box; float* p2e_rden; int p2e_btx, p2e_bty, p2e_bwd_ok;
    int* p2e_bwd_ids; int p2e_bwd_nsmall, p2e_bwd_nbig;   // tile ids: [0, nsmall) boxes of <= 2048 pixels (one wave each), then the big ones (1024 threads each)
    // equi2pers LDS path (omni_equi2pers.hip, e2p_box_kernel): per (patch, sample tile) the bounding box of the bilinear taps on
    // the ERP; index 0: 4-byte elements (8 x 32 sample tiles), 1: 2-byte elements (4 x 64); fb = tiles whose box exceeds the slot
    struct E2PWorkTab { long long key; uint4* dev; int nblocks; };
    struct E2PTiles { uint2* ent; int* fb; int nfb; int max_chunks; int tw, th, tx, ty; int ok;
                      int* order; int norder;                 // order: LDS-path tiles grouped by ERP longitude sector, one sector per XCD (-1 = padding)
                      std::vector<uint2> h_ent; std::vector<int> h_order, h_fb, h_region;   // host copies: the work tables are built from them
                      std::vector<E2PWorkTab> work; } e2p_boxes[2];               // work tables of e2p_box_kernel, one per plane count (omni_equi2pers.hip)
    std::mutex work_mu;
    int* e2p_fb_tiles;             // equi2pers backward: (patch, 32x32 tile) ids whose ERP footprint does not fit the LDS box
    int e2p_nfb;
    int e2p_ts;                    // equi2pers: tile side (32 or 16 samples) chosen so that the footprints fit the LDS box
    float2* e2p_ixy;               // equi2pers: clamped sampling coordinates (ix, iy) of every patch sample [N][ph][pw]
    // equi2pers backward by gathers (omni_equi2pers.hip): per (4 x 32 ERP tile, patch) the box of the patch samples whose taps touch the tile
    int4* e2p_bwd_box; int* e2p_bwd_ids; int e2p_bwd_nsmall, e2p_bwd_nbig, e2p_gtx, e2p_gty, e2p_bwd_ok;
    // the backward operators as constant sparse matrices (omni_spgather.h): one row per OUTPUT element, entries (source index, weight)
    OmniSpTable p2e_sp, e2p_sp;
    // scratch of the backward calls (the plane-interleaved copy of the gradient they gather from): one buffer per stream that has called,
    // grown on demand (never under capture), freed with the handle
    struct Ws { hipStream_t stream; float* ptr; size_t bytes; };
    std::vector<Ws> bwd_ws; std::mutex ws_mu;
    // the two backward tables are built by the FIRST backward call of the geometry (3.7 ms of one-time kernels a forward-only user never pays)
    std::mutex bwd_mu; int p2e_bwd_tried = 0, e2p_bwd_tried = 0;
};

// implemented in omni_geometry.hip
int omni_geometry_lookup(const omni_geometry** out, int nrows, float fov_h, float fov_w,
                         int ph, int pw, int H, int W, hipStream_t stream);
// implemented in omni_pers2equi.hip: fills g->cand on `stream`
int omni_p2e_build_candidates(omni_geometry* g, hipStream_t stream);
// implemented in omni_pers2equi.hip: fills g->p2e_tiles (needs g->cand)
int omni_p2e_build_tiles(omni_geometry* g, hipStream_t stream);
// implemented in omni_pers2equi.hip: fills g->p2e_bwd_box / p2e_rden (needs g->cand)
int omni_p2e_build_bwd(omni_geometry* g, hipStream_t stream);
// implemented in omni_equi2pers.hip: fills g->e2p_ixy, g->e2p_fb_tiles / e2p_nfb, then g->e2p_boxes
int omni_e2p_build_tileflags(omni_geometry* g, hipStream_t stream);
int omni_e2p_build_boxes(omni_geometry* g, hipStream_t stream);
int omni_e2p_build_bwd(omni_geometry* g, hipStream_t stream);        // fills g->e2p_bwd_* (needs g->e2p_ixy)

// ---------------------------------------------------------------- storage types
template <typename T> struct Store;
template <> struct Store<float> {
    static __device__ __forceinline__ float ld(const float* p) { return *p; }
    static __device__ __forceinline__ void st(float* p, float v) { *p = v; }
    // non-temporal form for outputs the kernel never re-reads: the lines stream to memory while the kernel runs instead of sitting dirty in
    // the XCD's L2 until the end-of-kernel write-back
    static __device__ __forceinline__ void st_nt(float* p, float v) { __builtin_nontemporal_store(v, p); }
};
template <> struct Store<__half> {
    static __device__ __forceinline__ float ld(const __half* p) { return __half2float(*p); }
    static __device__ __forceinline__ void st(__half* p, float v) { *p = __float2half_rn(v); }
    static __device__ __forceinline__ void st_nt(__half* p, float v)
    { const __half h = __float2half_rn(v); __builtin_nontemporal_store(*reinterpret_cast<const unsigned short*>(&h), reinterpret_cast<unsigned short*>(p)); }
};

// XCD-aware block remap: the dispatcher places block b on XCD b % 8 (MI355X_MICROARCH.md,
// "Workgroup dispatch"); give every XCD one CONTIGUOUS range of logical tiles so that
// neighbouring tiles (which share gather footprints) share one L2.  Pure speed: results
// never depend on placement.
__device__ __forceinline__ unsigned omni_xcd_remap(unsigned bid, unsigned nblocks)
{
    const unsigned per = nblocks >> 3;                 // tiles per XCD (full part)
    const unsigned full = per << 3;
    if (bid >= full) return bid;                       // ragged tail keeps identity
    return (bid & 7u) * per + (bid >> 3);
}

// Band-interleaved XCD map for kernels tiled over the ERP image.  ERP tile rows differ in cost (a polar row is covered by 3-4
// patches, an equatorial row by 2): one contiguous latitude range per XCD leaves the polar XCDs with ~1.6x the work of the
// equatorial ones.  But vertically adjacent tiles share most of their gather footprint, and only tiles on ONE XCD share an L2
// (interleaving single tile rows measured 3.6x the compulsory HBM traffic).  So the image is cut into bands of `band` tile rows,
// XCD x (blocks = x mod 8) owns bands x, x+8, x+16, ... and walks each band row by row.
// Launch omni_xcd_bands_grid() blocks; false = padding block (row past the end).
__host__ __device__ inline int omni_xcd_band_rows(int rows) { const int b = rows / 16; return b < 1 ? 1 : (b > 8 ? 8 : b); }
__host__ __device__ inline int omni_xcd_rows_grid(int rows, int cols)
{
    const int band = omni_xcd_band_rows(rows);
    return 8 * ((rows + 8 * band - 1) / (8 * band)) * band * cols;
}
__device__ __forceinline__ bool omni_xcd_rows(unsigned bid, int rows, int cols, int& row, int& col)
{
    const int band = omni_xcd_band_rows(rows);
    const unsigned x = bid & 7u, k = bid >> 3, per = (unsigned)(band * cols);
    const unsigned bl = k / per, wi = k - bl * per;                // band index within this XCD's list, tile within the band
    row = (int)((bl * 8u + x) * (unsigned)band + wi / (unsigned)cols);
    col = (int)(wi % (unsigned)cols);
    return row < rows;
}
