// gfo_internal.h -- context, level geometry and kernel launch prototypes of libgfo.so.
// gfx950 (MI355X) only.  Nothing here is part of the ABI (include/gfo.h is).
// (a classic include guard, not #pragma once: tests/host/combine_tsan.cc compiles gfo_combine.hip for the CPU under
//  ThreadSanitizer against a fake of this header and defines the guard first)
#ifndef GFO_INTERNAL_H
#define GFO_INTERNAL_H

#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include <memory>
#include <string>
#include <vector>

#include "../../include/gfo.h"

// [OCV] build variants (include/gfo.h gfo_build_variant; same values as the checker's ocv_variants.json).  `make EXTRA="-DGFO_OCV_RESIZE=1"`.
#ifndef GFO_OCV_RESIZE
#define GFO_OCV_RESIZE 0        // 0: 11-bit fixed-point bilinear (resizeGeneric_), 1: float bilinear, rounded half to even once
#endif
#ifndef GFO_OCV_ATAN_FMA
#define GFO_OCV_ATAN_FMA 0      // 0: fastAtan2's polynomial in separate multiplies and adds, 1: Horner steps fused
#endif
#ifndef GFO_OCV_BLUR_ROUND
#define GFO_OCV_BLUR_ROUND 0    // 0: exact accumulation, one (v + 2^15) >> 16; 1: each pass rounded to 8 bits, (v + 128) >> 8
#endif
#ifndef GFO_GAUSS_TAPS
#define GFO_GAUSS_TAPS 18, 34, 49, 55   // the outer three taps and the centre one, scale 256 (symmetric kernel)
#endif

#define GFO_EDGE 19          // EDGE_THRESHOLD, ORBextractor.cc:74
#define GFO_HALF_PATCH 15    // HALF_PATCH_SIZE, ORBextractor.cc:73
#define GFO_PATCH 31         // PATCH_SIZE, ORBextractor.cc:72
#define GFO_MIN_BORDER 16    // EDGE_THRESHOLD-3, ORBextractor.cc:775
#ifndef GFO_FAST_QCAP
#define GFO_FAST_QCAP 768       // k_fast: queue entries per wave (a stage-A pass adds up to 256)
#endif
#define GFO_FAST_XOFF 1       // k_fast: the LDS tile starts this many bytes left of the cell (scan column 0 at tile column 4)
#define GFO_CELL_W 30        // W, ORBextractor.cc:771
#define GFO_CNT_STRIDE 32     // ints between per-(image,level) candidate counters: one 128-B line each,
                             // so the per-cell atomicAdds of different levels/images never share a line

// Geometry of one pyramid level, shared by host planning and every kernel (passed by value).
#ifndef GFO_BLUR_STRIP
#define GFO_BLUR_STRIP 24   // rows a blur thread walks (6 halo rows each): shared by plan() and k_blur
#endif

struct GfoLevel {
    int w, h, pitch;          // plane size; pitch multiple of 64 B
    long long plane_off;      // byte offset of the plane inside one image's pyramid block (level>=1)
    long long blur_off;       // byte offset inside one image's blurred-pyramid block
    // FAST cell grid (ORBextractor.cc:775-789); ncols == 0 when the level is too small
    int ncols, nrows, wcell, hcell;
    int max_bx, max_by;       // maxBorderX/Y = w-16, h-16
    int cell_base;            // prefix of ncols*nrows over levels
    // blur tiling
    int tiles_x, tiles_y, tile_base;   // blur: quads per row, 32-row strips, first block of the interior launch
    int blur_base_b;                   // blur: first block of the border launch
    // quadtree (ORBextractor.cc:539-563)
    int quota;                // mnFeaturesPerLevel[level]
    int n_ini;                // root nodes
    float hx;                 // root width
    int node_cap;             // max nodes the selection can hold
    // buffers (element offsets inside one image's block)
    long long cand_off;       // u32 candidates
    int cand_cap;
    int sel_off, sel_cap;     // u32 selected keypoints
    // scaling (ORBextractor.cc:839-849, 1164-1170)
    float scale;
    int patch_size;           // int(31*scale)
    // resize tables: offsets (in elements) into the table buffers
    int xtab_off, ytab_off;
};

// One launch of the banded pyramid: levels [lb, le) in nb bands per image
struct GfoBandGroup {
    int lb, le, nb, lds_bytes, tab_off;   // tab_off in int4 elements into d_band
};

struct GfoGeom {
    int nlevels;
    int w0, h0;
    int total_cells;
    int total_tiles;          // blocks of the blur interior launch
    int blur_total_b;         // blocks of the blur border launch
    int total_sel_cap;        // per image: sum of sel_cap
    int kp_stride;            // per image output capacity (>= total_sel_cap)
    int ini_th, min_th;
    int fast_tile_pitch, fast_tile_rows, fast_smap_pitch, fast_smap_rows; // LDS plan per wave
    int fast_npx_max;         // largest scan area of a cell, multiple of 8
    int fast_q_cap;           // entries of a wave's survivor / corner queue: min(fast_npx_max, GFO_FAST_QCAP)
    long long pyr_img_stride;   // bytes per image of levels 1..L-1
    long long blur_img_stride;  // bytes per image of blurred levels 0..L-1
    long long cand_img_stride;  // u32 elements per image
    // banded pyramid (k_pyramid_bands): smallest band count of the groups (0 = not planned), LDS offset and
    // pitch of each level's band
    int pyr_nb;
    int band_lds_off[GFO_MAX_LEVELS], band_lp[GFO_MAX_LEVELS];
    GfoLevel lv[GFO_MAX_LEVELS];
};

// Input description of level 0 (either the arena staging buffer or a caller's device buffer)
struct GfoInput {
    const uint8_t* base;
    long long pitch;
    long long img_stride;
};

struct GfoStereoDev {
    float* u_right;     // [pairs][kp_stride]
    float* depth;
    int* best_dist;
    int* best_idx;
    int* nmatched;      // [pairs]
    unsigned char* counted;  // [pairs][kp_stride] 1 where the reference's loop reaches nmatched++ (Frame.cc:1286)
};

// Results of a small batch, gathered by ONE kernel straight into the context's pinned host buffer (mapped into the
// device's address space): nine separate D2H copies cost ~75 us of copy-engine latency on the per-frame path
// (profiles/latency_timeline_r02.txt), this costs ~10 us of PCIe writes at the end of the launch sequence.
// cut_pairs > 0 (round 5): the first `cut_pairs` workgroups of the kernel make the outlier cut of one stereo pair each
// (Frame.cc:1290-1313, what k_stereo_cut does as a launch of its own) and write that pair's final u_right / depth / nmatched
// to the device arrays AND to the host block; the other workgroups copy the segments (k_pack_results_cut, k_stereo.hip).
#define GFO_PACK_MAX 10
struct GfoPack {
    const uint4* src[GFO_PACK_MAX];
    uint4* dst[GFO_PACK_MAX];
    int n16[GFO_PACK_MAX];     // 16-byte units
    int nseg;
    int cut_pairs;
    const int* cut_cnt_dev;
    int cut_nl_host;
    GfoStereoDev cut_out;
    int cut_out_stride;
    float* h_u_right;          // host (mapped) destinations of the cut's outputs, [pairs][cut_out_stride] / [pairs]
    float* h_depth;
    int* h_nmatched;
};

// right keypoints of a pair counting-sorted by floor(y), compact SoA (k_stereo_bucket)
struct GfoStereoSort {
    float* sx;
    float* sy;
    unsigned* soi;      // octave << 16 | original index
    uint8_t* sdesc;     // 32 B per keypoint
    int* row_start;     // [pairs][n_rows + 1]
    int* lorder;        // LEFT keypoints whose row is inside the image, counting-sorted by (int)y: original indices [pairs][stride]
    int* lrow_start;    // [pairs][n_rows + 1]
};

struct GfoStereoLaunch {
    const gfo_keypoint* kl; const uint8_t* dl;
    const gfo_keypoint* kr; const uint8_t* dr;
    const int* cnt_dev; int nl_host, nr_host;
    long long pair_stride_kp;
    int npairs;
    const float* d_scale;
    gfo_stereo_params p;
    const float* min_d; const float* max_d; long long win_stride;   // per-keypoint disparity windows, floats between pairs
    GfoStereoDev out; int out_stride;
    GfoStereoSort sort; int sort_stride;
    int window;
    int nlevels;          // entries of d_scale
    bool cut_in_pack;     // the caller's pack kernel makes the outlier cut (GfoPack::cut_pairs): no k_stereo_cut launch
};

// work buffers of the batched projection search (k_project.hip), one block per frame
struct GfoProjBuf {
    void* base = nullptr;
    int frames_cap = 0, m_cap = 0, n_cap = 0;
    int* cell_start = nullptr;
    void* cell_xy = nullptr;
    unsigned* cell_meta = nullptr;
    void* cand = nullptr;
    int* pick = nullptr;
    int* pick_dist = nullptr;
    unsigned* live = nullptr;
    int* rot_bin = nullptr;
    int* spill_off = nullptr;               // [frames][m] by live slot: start of the point's full candidate list in `spill`, -1 none
    unsigned long long* spill = nullptr;    // [spill_cap] candidate keys of the points whose list outgrew the cache (k_proj_round0_wave)
    int spill_cap = 0;
    int* tab_g = nullptr;
    int* counters = nullptr;
    int* out_mp = nullptr;
    int* out_score = nullptr;
    int* out_q = nullptr;                   // [frames][m] per-point outcomes (gfo_search_by_projection_points)
};

enum GfoStage {
    ST_RESIZE = 0, ST_BLUR, ST_FAST, ST_QUADTREE, ST_ORIENT_DESC, ST_STEREO_BUCKET, ST_STEREO, ST_STEREO_CUT,
    ST_PROJECT, ST_BOW, ST_COUNT
};

struct GfoEngine;   // gfo_combine.hip
struct GfoPair;     // gfo_combine.hip: two contexts declared the left / right extractor of one stereo rig (gfo_ctx_pair)

struct gfo_ctx {
    gfo_params prm{};
    uint64_t id = 0;                 // process-wide serial number (gfo_ctx_id), never reused
    int device = 0;
    hipStream_t own_stream = nullptr;
    hipStream_t stream = nullptr;
    // the blur depends only on the pyramid and is needed only by orient_desc: it runs on this side stream next to
    // FAST and the quadtree (fork / join by events); all work is still ordered on `stream` for the caller
    hipStream_t side_stream = nullptr;
    hipEvent_t ev_fork = nullptr, ev_join = nullptr;
    bool zero_cnt_pending = false;   // the next pyramid launch clears the candidate counters (k_pyramid.hip)
    // gfo_ctx_chain: this context's extractions wait for `chain_after`'s pace event; `pace_stage` != 0 means some
    // context waits on THIS one, which then records ev_pace after that stage of every extraction
    gfo_ctx* chain_after = nullptr;
    hipEvent_t ev_pace = nullptr;
    int pace_stage = 0;
    bool pace_recorded = false;
    bool fork_blur = true;
    std::string err;
    std::string launch_err;
    bool debug_sync = false;
    int cur_stage = 0;
    // host tables (ORBextractor.cc:409-469)
    std::vector<float> scale, inv_scale, sigma2, inv_sigma2;
    std::vector<int> quota;
    int umax[16]{};
    // geometry + arena
    bool planned = false;
    GfoGeom g{};
    GfoGeom* d_geom = nullptr;       // device copy of g (kernels read it through scalar loads)
    int cap_batch = 0;
    uint8_t* d_input = nullptr;      // staging for host images (pitch = lv[0].pitch)
    uint8_t* d_pyr = nullptr;
    uint8_t* d_blur = nullptr;
    uint32_t* d_cand = nullptr;
    int* d_cand_cnt = nullptr;       // [batch][nlevels]
    uint16_t* d_node_of = nullptr;   // [batch][cand_img_stride] quadtree scratch
    uint32_t* d_sel = nullptr;       // [batch][total_sel_cap]
    int* d_sel_cnt = nullptr;        // [batch][nlevels]
    gfo_keypoint* d_kp = nullptr;    // [batch][kp_stride]
    uint8_t* d_desc = nullptr;       // [batch][kp_stride][32]
    int* d_kp_cnt = nullptr;         // [batch]
    int* d_flags = nullptr;          // [4] overflow flags
    int* d_xofs = nullptr;           // resize tables, all levels
    short* d_xcoef = nullptr;
    int* d_yofs = nullptr;
    uint8_t* d_qt_scratch = nullptr;   // quadtree state in HBM, only when a level's state exceeds LDS
    size_t qt_scratch_stride = 0;      // bytes per (image, level) workgroup
    size_t qt_lds_granted = 0;         // dynamic LDS limit already raised for k_quadtree on this context's device
    int* d_cell_tab = nullptr;    // FAST: four ints per cell: level | row << 4 | column << 16, the two lane maps (plan())
    int* d_od_tab = nullptr;      // k_orient_desc: level | pair << 4 per wavefront of an image (inside d_cell_tab's allocation)
    int od_pairs = 0;             // its length
    int* d_band = nullptr;        // per group: int4 [nb][nlevels] = {c0, c1, o0, o1}
    GfoBandGroup band_groups[GFO_MAX_LEVELS];   // nb == 0: a single level launched as k_resize
    int n_band_groups = 0, band_threads = 0;
    short* d_ycoef = nullptr;
    float* d_scale = nullptr;        // mvScaleFactor on the device
    float* d_inv_scale = nullptr;    // mvInvScaleFactor on the device
    GfoStereoDev st{};
    GfoStereoSort st_sort{};
    int st_rows_cap = 0;             // rows the row_start table of st_sort is sized for
    // state of the last batch
    int last_nimg = 0;
    GfoInput last_in{};
    bool have_batch = false;
    bool have_pyramid = false;
    bool have_stereo = false;
    // profiling
    bool profiling = false;
    double stage_ms[ST_COUNT]{};
    int stage_launches[ST_COUNT]{};
    struct PendingEv { int stage; hipEvent_t a, b; };
    std::vector<PendingEv> pending;
    std::vector<hipEvent_t> ev_pool;
    // scratch for host-array matcher entry points
    void* d_scratch = nullptr;
    size_t scratch_bytes = 0;
    void* d_pj_cand = nullptr;        // gfo_projection_candidates: offsets [m + 1], entries [cap] and the unsorted keys [cap] of the last call
    size_t pj_cand_bytes = 0;
    int last_project_rounds = 0;
    // batched projection search: resident local-map descriptors + per-frame work buffers
    GfoProjBuf pj{};
    uint8_t* d_map_desc = nullptr;
    int map_cap = 0, map_m = 0;
    bool have_projection = false;
    int proj_frames = 0, proj_step = 1;
    // latency path (host images, small batches): pinned staging for one-copy / one-sync transfers and a captured
    // hipGraph of the fixed launch sequence
    uint8_t* h_in = nullptr;  size_t h_in_bytes = 0;     // hipHostMalloc
    uint8_t* h_out = nullptr; size_t h_out_bytes = 0;
    // host-array matcher calls (SearchByProjection / SearchByBoW / ComputeBoW on caller arrays): every input of a call is packed
    // here and crosses in ONE H2D copy, every output comes back in ONE D2H (a pageable hipMemcpyAsync costs ~16-22 us each)
    uint8_t* h_min = nullptr;  size_t h_min_bytes = 0;
    uint8_t* h_mout = nullptr; size_t h_mout_bytes = 0;
    bool graph_ok = false;   // GFO_GRAPH=1 opts in (see run_extract)
    hipGraphExec_t graph_exec = nullptr;
    struct GraphKey { const void* base; const void* pack_dst; long long pitch, img_stride; int nimg, stereo; gfo_stereo_params sp; int plan_gen; } graph_key{};
    int plan_gen = 0;
    // gfo_batch_deliver: D2H of a batch's results on a stream of its own; the next extraction waits for it before
    // k_orient_desc overwrites the outputs
    hipStream_t copy_stream = nullptr;
    hipEvent_t ev_results = nullptr, ev_delivered = nullptr;
    bool deliver_pending = false;
    // frame combiner (gfo_ctx_set_combining): per-frame host calls of this context may run inside a shared device batch
    bool flags_snapshot = false;        // the last batch's overflow flags were moved to d_flags[4..7] by gfo_batch_deliver
    bool combining = false;
    std::shared_ptr<GfoEngine> engine;
    std::shared_ptr<GfoPair> pair;      // gfo_ctx_pair
    // resident vocabulary tree (gfo_vocabulary_upload)
    void* d_voc = nullptr;
    size_t voc_desc_off = 0, voc_fc_off = 0, voc_nc_off = 0, voc_wid_off = 0, voc_w_off = 0, voc_w64_off = 0;
    int voc_nodes = 0, voc_depth = 0;     // Jacobi rounds of the last gfo_search_by_projection
};

// ---- kernel launchers (each in its own .hip file) ------------------------------------------
// the candidate-counter pointer for the first pyramid launch of an extraction (nullptr afterwards)
inline int* gfo_take_zero_cnt(gfo_ctx* c)
{
    if (!c->zero_cnt_pending) return nullptr;
    c->zero_cnt_pending = false;
    return c->d_cand_cnt;
}
// every kernel of a translation unit (for gfo_preload_kernels, gfo_api.hip)
void gfo_kernels_fast(std::vector<const void*>& v);
void gfo_kernels_pyramid(std::vector<const void*>& v);
void gfo_kernels_blur(std::vector<const void*>& v);
void gfo_kernels_quadtree(std::vector<const void*>& v);
void gfo_kernels_orient_desc(std::vector<const void*>& v);
void gfo_kernels_stereo(std::vector<const void*>& v);
void gfo_kernels_project(std::vector<const void*>& v);
void gfo_kernels_bow(std::vector<const void*>& v);
void gfo_launch_resize(gfo_ctx* c, const GfoInput& in, int level, int nimg);
void gfo_launch_resize_tail(gfo_ctx* c, const GfoInput& in, int level_begin, int nimg);
size_t gfo_quadtree_lds_bytes(int ncap, int klds);
int gfo_take_launch_err(gfo_ctx* c);
void gfo_launch_pyramid_bands(gfo_ctx* c, const GfoInput& in, int nimg);
int gfo_pyramid_bands_prepare(int lds_bytes);
void gfo_launch_blur(gfo_ctx* c, const GfoInput& in, int nimg);
void gfo_launch_fast(gfo_ctx* c, const GfoInput& in, int nimg);
// images up to which a batch counts as a per-frame batch: 1024-thread one-thread-per-node quadtree workgroups, quadtree + blur as one launch
// (GFO_FEW_MAX; the frame combiner's batches at K = 8 .. 16 camera threads are 16 .. 32 images)
int gfo_few_max();
void gfo_launch_quadtree(gfo_ctx* c, int nimg);
bool gfo_launch_quadtree_blur(gfo_ctx* c, const GfoInput& in, int nimg);   // per-frame batches: quadtree + blur as one launch (false: not applicable, nothing launched)
void gfo_launch_orient_desc(gfo_ctx* c, const GfoInput& in, int nimg);
void gfo_launch_stereo(gfo_ctx* c, const GfoStereoLaunch& s);
int gfo_pinned(gfo_ctx* c, uint8_t** buf, size_t* cap, size_t bytes);   // grow-only hipHostMalloc buffer (synchronises the stream when it grows)
void gfo_launch_copy16(gfo_ctx* c, void* dst, const void* src, size_t bytes, hipStream_t st);   // gfo_api.hip
// host-array matcher calls: the last kernel of a call writes the answer into the pinned block itself (1, default) or the block is filled
// by a copy behind it (GFO_MATCHER_HOST_WRITES=0; same-box A/B in profiles/matcher_call_latency_r05.txt)
inline bool gfo_matcher_host_writes()
{
    static const bool on = !(getenv("GFO_MATCHER_HOST_WRITES") && atoi(getenv("GFO_MATCHER_HOST_WRITES")) == 0);
    return on;
}
// One host-array call's transfers.  in(): reserve the pinned mirror of the device scratch [0, bytes); put(): memcpy one input to its
// offset; up(): the ONE H2D.  out()/down(): ONE D2H of a contiguous device range into pinned memory, read after the stream sync.
struct GfoXfer {
    gfo_ctx* c;
    uint8_t* H = nullptr;    // pinned mirror of the inputs
    uint8_t* HO = nullptr;   // pinned copy of the outputs
    explicit GfoXfer(gfo_ctx* ctx) : c(ctx) {}
    int in(size_t bytes)
    {
        int rc = bytes <= c->h_min_bytes ? 0 : gfo_pinned(c, &c->h_min, &c->h_min_bytes, bytes + bytes / 2);   // headroom: frames vary
        H = c->h_min;
        return rc;
    }
    void put(size_t off, const void* src, size_t bytes) const { if (bytes) memcpy(H + off, src, bytes); }
    hipError_t up(void* d_dst, size_t bytes, hipStream_t st) const
    {
        // up to 1 MB: a copy KERNEL in front of the call's kernels (the pinned block is device-visible); larger blocks take the copy engine
        static const long kernel_max = getenv("GFO_MATCHER_UPLOAD_KERNEL_MAX") ? atol(getenv("GFO_MATCHER_UPLOAD_KERNEL_MAX")) : (1L << 20);
        if ((long)bytes <= kernel_max && (bytes & 15) == 0 && ((uintptr_t)d_dst & 15) == 0) {
            gfo_launch_copy16(c, d_dst, H, bytes, st);
            return hipSuccess;   // (a launch error is collected by gfo_take_launch_err with the call's other launches)
        }
        return hipMemcpyAsync(d_dst, H, bytes, hipMemcpyHostToDevice, st);
    }
    int out(size_t bytes)
    {
        int rc = bytes <= c->h_mout_bytes ? 0 : gfo_pinned(c, &c->h_mout, &c->h_mout_bytes, bytes + bytes / 2);
        HO = c->h_mout;
        return rc;
    }
    hipError_t down(const void* d_src, size_t bytes, hipStream_t st) const
    {
        // (the calls whose last kernel cannot write the answer itself) up to 1 MB: a copy kernel into the pinned block, as in up()
        static const long kernel_max = getenv("GFO_MATCHER_UPLOAD_KERNEL_MAX") ? atol(getenv("GFO_MATCHER_UPLOAD_KERNEL_MAX")) : (1L << 20);
        if (gfo_matcher_host_writes() && (long)bytes <= kernel_max && (bytes & 15) == 0 && ((uintptr_t)d_src & 15) == 0) {
            gfo_launch_copy16(c, HO, d_src, bytes, st);
            return hipSuccess;
        }
        return hipMemcpyAsync(HO, d_src, bytes, hipMemcpyDeviceToHost, st);
    }
};
void gfo_launch_pack_cut(gfo_ctx* c, const GfoPack& p, hipStream_t st);   // k_pack_results with the stereo cut in it (GfoPack::cut_pairs)
int gfo_stereo_window(const float* scale, int nlevels);
void gfo_launch_stereo_sad(gfo_ctx* c, const GfoStereoLaunch& s, const GfoInput& in, const float* d_inv_scale);

// host-side pieces of gfo_api.hip the frame combiner (gfo_combine.hip) builds on
int gfo_fail(gfo_ctx* c, int code, const char* fmt, ...);
int gfo_plan(gfo_ctx* c, int w, int h, int batch);
struct GfoSmallLayout {
    int nimg_cap;
    int pitch;            // of the staged images: the width when that keeps rows 16-byte aligned (one memcpy / one DMA per image), else the arena's
    size_t img_bytes;
    size_t o_fl, o_cnt, o_kp, o_ds, o_ur, o_dp, o_bd, o_bi, o_nm;   // offsets into the pinned result buffer
};
int gfo_small_prepare(gfo_ctx* c, int nimg_cap, GfoSmallLayout* L);
int gfo_small_upload(gfo_ctx* c, gfo_ctx* ec, const GfoSmallLayout& L, int first, int count, const uint8_t* const* imgs, int w, int h, int stride,
                     hipStream_t st, bool lone_caller = false);
int gfo_small_submit(gfo_ctx* c, const GfoSmallLayout& L, int nimg, const gfo_stereo_params* sp, bool copy_in);
int gfo_small_collect(gfo_ctx* c, const GfoSmallLayout& L, int i, gfo_keypoint* kp, uint8_t* desc, int cap, int* n);
void gfo_small_collect_stereo(gfo_ctx* c, const GfoSmallLayout& L, int pair, int n_left, int cap, float* u_right, float* depth,
                              int32_t* best_dist, int32_t* best_idx_r, int* nmatched);
// host-array stereo association of several pairs as one device batch (the frame combiner's kind 3)
struct GfoPairBlock {      // one pair's inputs, staged contiguously: {nl, nr, -, -} | kl | dl | kr | dr | min_d | max_d, every part 16-byte aligned
    size_t bytes, o_kl, o_dl, o_kr, o_dr, o_min, o_max;
};
GfoPairBlock gfo_pair_block(int kp_stride);
int gfo_small_submit_pairs(gfo_ctx* c, const GfoSmallLayout& L, int npairs, const gfo_stereo_params* sp, const uint8_t* d_stage);

// gfo_combine.hip
#define GFO_COMBINE_DIRECT 1   // (positive: never an ABI status) the engine cannot serve this request -- no slot could be prepared, or the
                               // batch it joined failed as a whole: the caller runs it alone on the direct path
int gfo_combined_extract(gfo_ctx* c, int kind, const uint8_t* const* imgs, int w, int h, int stride, const gfo_stereo_params* sp,
                         gfo_keypoint* const* kp, uint8_t* const* desc, int cap, int* n, float* u_right, float* depth,
                         int32_t* best_dist, int32_t* best_idx_r, int* nmatched);
int gfo_combined_stereo_match(gfo_ctx* c, const gfo_keypoint* kl, const uint8_t* dl, int nl, const gfo_keypoint* kr, const uint8_t* dr, int nr,
                              const float* sf, int nlevels, const gfo_stereo_params* p, const float* min_d, const float* max_d,
                              float* u_right, float* depth, int32_t* best_dist, int32_t* best_idx_r, int* nmatched, int* status);
void gfo_engine_release(gfo_ctx* c);
#define GFO_COMBINER_COUNTERS 8
void gfo_pair_release(gfo_ctx* c);
bool gfo_has_pair(const gfo_ctx* c);   // (gfo_ctx::pair is only ever touched through the atomic shared_ptr functions, gfo_combine.hip)
int gfo_pair_extract(gfo_ctx* c, const uint8_t* img, int w, int h, int stride, gfo_keypoint* kp, uint8_t* desc, int cap, int* n);
int gfo_pair_lookup(gfo_ctx* c, const gfo_keypoint* kl, const uint8_t* dl, int nl, const gfo_keypoint* kr, const uint8_t* dr, int nr,
                    const float* sf, int nlevels, const gfo_stereo_params* p, const float* min_d, const float* max_d,
                    float* u_right, float* depth, int32_t* best_dist, int32_t* best_idx_r, int* nmatched);
void gfo_note_pinned(const uint8_t* p, size_t bytes, bool add);   // gfo_api.hip: the registry gfo_small_upload consults

// profiling helpers (gfo_api.hip)
void gfo_prof_begin(gfo_ctx* c, int stage);
void gfo_prof_end(gfo_ctx* c);
void gfo_prof_kernel_events(gfo_ctx* c, hipEvent_t* a, hipEvent_t* b);

// Every kernel launch of the library.  With profiling on (gfo_profile_enable) the launch carries a start / stop event
// pair (hipExtLaunchKernelGGL): they take the kernel's own begin / end timestamps, i.e. the duration rocprofv3's kernel
// trace reports -- events recorded around a plain launch also count the dispatch latency behind the previous
// kernel (measured +20 us on a 166 us kernel).
#define GFO_LAUNCH(c, kern, grid, block, lds, stream, ...)                                                               \
    do {                                                                                                                 \
        hipEvent_t ea_ = nullptr, eb_ = nullptr;                                                                         \
        if ((c)->profiling) gfo_prof_kernel_events((c), &ea_, &eb_);                                                     \
        if (ea_) hipExtLaunchKernelGGL(kern, grid, block, (std::uint32_t)(lds), stream, ea_, eb_, 0, __VA_ARGS__);       \
        else hipLaunchKernelGGL(kern, grid, block, lds, stream, __VA_ARGS__);                                            \
    } while (0)

// device-side address helpers shared by kernels
__device__ __forceinline__ const uint8_t* gfo_level_ptr(const GfoGeom& g, const GfoInput& in, const uint8_t* pyr,
                                                         int level, int img, int* pitch)
{
    if (level == 0) {
        *pitch = (int)in.pitch;
        return in.base + (long long)img * in.img_stride;
    }
    *pitch = g.lv[level].pitch;
    return pyr + (long long)img * g.pyr_img_stride + g.lv[level].plane_off;
}

#endif  // GFO_INTERNAL_H
