// k_project.hip -- Frame::AssignFeaturesToGrid / GetFeaturesInArea (Frame.cc:461-476, 593-658) and
// ORBmatcher::SearchByProjection, both overloads (ORBmatcher.cc:155-249 map points, :1440-1593 frame to frame),
// for a BATCH of frames in three launches, nothing returning to the host in between:
//
//   k_proj_grid     one workgroup per frame: a 128x96 scan grid over the keypoints the frame assigns to its 64x48 grid,
//                   as a CSR of positions and meta words (index | octave | reference cell) in cell order
//   k_proj_round0   one THREAD per projected point (a 50 000-point local map against 4000 keypoints leaves
//                   0-3 candidates per point: a wavefront per point would idle 60 lanes): every point is
//                   evaluated against the entry state; points that can ever match go to the frame's live list
//                   TOGETHER WITH their sorted candidate list (what blocks a candidate later is the only thing
//                   that changes between rounds: window, level, mvuRight gate and distance are static)
//   k_proj_resolve  one workgroup per frame: the order dependence, resolved in LDS over the cached candidate
//                   lists (below), then the owner / score / rotation-histogram epilogue.  Measured on MI355X:
//                   re-evaluating the live points from the grid each round cost 885 us per batch (a workgroup's
//                   fully divergent gathers go through ONE CU's address unit, ~2 clk per distinct line); the
//                   cached lists turn a round into a coalesced 32-byte read and a few LDS lookups per point
//
// The reference is ORDER-DEPENDENT: points are visited in vector order, every accepted match writes
// F.mvpMapPoints[bestIdx], and later points skip keypoints already taken by a point with observations
// (ORBmatcher.cc:197-199, 233).  The device version reproduces the serial result exactly with a fixed-point
// iteration:
//   state  = each live point's tentative pick;
//   update = every point i re-evaluates its candidates treating keypoint k as taken iff it was taken on entry
//            or some point j < i (with observations) currently picks k (claim table: atomicMin per keypoint);
//   the serial answer is the unique fixed point, point i is final after at most (its rank among the matching
//   points) rounds, and "nothing changed in a round" proves convergence.
// Points whose best distance exceeds the threshold under the entry state can never match (more blocking only
// removes candidates), so only the few thousand live points are re-evaluated after round 0.
//
// Candidate order matters only for ties (strict '<' keeps the first candidate, :212-225).  The reference
// iterates cells (ix, iy) then the cell's list in keypoint order, so "first" is the minimum of
// (dist, ix, iy, idx): candidates can be scanned in any order with that 64-bit key, and the best /
// second-best pair is the two smallest keys (a stable sort by distance).
#include "gfo_internal.h"
#include "k_wave.inc"
#include <limits.h>
#include <stdlib.h>

#define GRID_COLS 64   // FRAME_GRID_COLS, Frame.h:92
#define GRID_ROWS 48   // FRAME_GRID_ROWS, Frame.h:93
#define NCELL (GRID_COLS * GRID_ROWS)
// The SCAN grid.  Which keypoints GetFeaturesInArea returns does not depend on the grid: every keypoint the frame has
// assigned to a cell (PosInGrid, Frame.cc:648-658) that passes |dx| < r, |dy| < r lies in a cell of the query's range
// (round() of a value inside [floor(lo), ceil(hi)] stays inside), so any spatial index over the assigned keypoints
// finds the same set; the reference's 64x48 cell only enters the ORDER of the candidates (the key below) and is
// carried in each item.  The scan therefore uses cells of half the size: a level-0 window (15 px) in 30 x 22.5-px cells
// visits 12x its own area, in 15 x 11-px cells 3.5x -- and a wave's scan loop runs as long as its unluckiest lane.
#define SG_COLS 128
#define SG_ROWS 96
#define NSG (SG_COLS * SG_ROWS)
#define TH_HIGH 100    // ORBmatcher.cc:57
#define HISTO_LENGTH 30  // ORBmatcher.cc:59

// what a point that matched nothing did (the reference's three ways out of the loop body, src/ORBmatcher.cc:60-92, 137-138):
//   PJ_PT_NONE      not in view / bad / GetFeaturesInArea returned nothing -- `continue` before any distance is computed
//   PJ_PT_RATIO     best candidate within TH_HIGH, rejected by the ratio test -- `continue`
//   PJ_PT_FAR       candidates in the window, none usable within TH_HIGH -- falls through to the END of the loop body
//                   (where SearchByProjection_Budget reads its clock, :96-102)
enum { PJ_PT_NONE = -1, PJ_PT_RATIO = -2, PJ_PT_FAR = -3 };
#define PJ_DIST_NONE 256    // accept_rule: no candidate, or the best one beyond the threshold
#define PJ_DIST_RATIO 257   // accept_rule: rejected by the ratio test
enum { PJ_NLIVE = 0, PJ_ROUNDS = 1, PJ_NMATCH = 2, PJ_ERR = 3, PJ_FALLBACK = 4, PJ_SPILL = 5, PJ_CNT = 8 };
#define PJ_SPILL_PER_POINT 64   // the spill pool holds this many candidate keys per point of a wavefront-per-point call, on average
#define PJ_K 7   // cached candidates per live point: 7 entries + 1 header word = 32 bytes
#define PJ_RR 4  // k_proj_resolve keeps up to 1024 * PJ_RR live points in registers across the rounds

// Everything is [frame][...]: frame f of a per-frame array starts at base + f * stride (stride 0 = shared).
struct ProjB {
    const gfo_keypoint* kp; long long kp_stride;     // elements
    const uint8_t* desc;                              // [frame][kp_stride][32]
    const float* u_right; long long ur_stride;        // may be null
    const uint8_t* taken0; long long tk_stride;       // may be null
    const float* kp_angle; long long ang_stride;      // rotation check; null = kp[i].angle
    const int* n_dev; int n_dev_stride; int n_host;   // keypoints of frame f = n_dev ? n_dev[f * n_dev_stride] : n_host
    gfo_frame_bounds fb;
    float inv_w, inv_h;
    // projected points: form 0 = gfo_proj_query, form 1 = gfo_map_point (turned into a query on the fly)
    int form;
    const void* q; long long q_stride;                // elements per frame
    const uint8_t* q_desc; long long qd_stride;       // bytes per frame (0: the resident map, shared)
    int m;
    float th; int bfactor; float scale[GFO_MAX_LEVELS]; int nlevels;   // form 1: r * mvScaleFactors[level]
    int use_ratio; float nn_ratio; int th_dist; int check_ori;
    // per-frame scratch
    int n_cap;                 // stride of the per-keypoint arrays
    float sinv_w, sinv_h;      // scan-grid cells per pixel
    int* cell_start;           // [NSG + 1] CSR over the scan grid, column-major
    float2* cell_xy;           // [n_cap] keypoint positions in scan-cell order
    unsigned* cell_meta;       // [n_cap] index | octave << 16 | reference cell column << 20 | reference cell row << 26
    int* pick;                 // [m] by live slot: keypoint picked, -1 none
    int* pick_dist;            // [m] by live slot
    unsigned* live;            // [m] live point | has-observations << 31
    uint4* cand;               // [m][2] by live slot: header (count | truncated << 8) + PJ_K entries
                               //        dist << 23 | octave << 16 | index, in the reference's candidate order
    int* rot_bin;              // [m] by live slot
    int* spill_off;            // [m] by live slot: where the point's FULL candidate list starts in `spill` (header bits 9.. = its length), -1 none
    unsigned long long* spill; // [spill_cap], one pool for the frames of a call (cursor: counters[PJ_SPILL] of frame 0): candidate keys, unsorted,
                               // of the points with more than PJ_K candidates
    int spill_cap;
    int* tab_g;                // [2 * n_cap] claim / owner and score tables when they do not fit LDS
    int* counters;             // [PJ_CNT]
    int* out_mp; int* out_score;   // [n_cap]
    // host-array calls (one frame): the results ALSO go straight to the caller-side pinned block (device-visible host memory) -- a D2H
    // copy behind the last kernel is ~5 us of copy and ~10 us of hand-over between the compute queue and the copy engine
    int* h_counters; int* h_out_mp; int* h_out_score;   // null: the batch forms
    // host-array calls: k_proj_grid ALSO brings the call's inputs over -- workgroups behind the frames' copy cp_n16 16-byte units from
    // the pinned block to the scratch, while the grid workgroup reads the keypoints from the pinned block itself (kp_grid): one launch
    // instead of a copy kernel and a grid kernel
    const uint4* cp_src; uint4* cp_dst; int cp_n16;
    const gfo_keypoint* kp_grid;   // null: a.kp
    int grid_frames;               // frames of the launch (the copy workgroups come behind them)
    int max_matches;               // > 0: BUDGETING_FEATURE_MATCHING (gfo_proj_mode::max_matches), host-array calls only
    // per-POINT outcomes (gfo_search_by_projection_points: SearchByProjection_OnePoint's return value for every point taken in vector
    // order, include/ORBmatcher.h:71-150): [m] per frame, null = not wanted.  keypoint | distance << 16, or PJ_PT_* below
    int* out_q; int* h_out_q;
    // ORBmatcher::Fuse(KeyFrame*, MapPoints, th) (ORBmatcher.cc:1019-1051): a candidate is dropped when its reprojection error, weighted with
    // its own level's inverse sigma^2, exceeds the chi-square bound (5.99; 7.8 with the right-image coordinate) -- INSTEAD of the mvuRight
    // window gate of the tracking overloads.  0 = off
    int fuse_gate; const float* inv_sigma2;   // [GFO_MAX_LEVELS] in device memory (a table in the kernel arguments would be indexed through scratch)
#ifdef GFO_PROJ_DEBUG
    int dbg_stop;
#endif
};

struct ProjQ {
    float u, v, ur, radius, angle;
    int min_level, max_level;
    bool active, obs;
};

// float -> int as the reference's x86 build converts (cvttss2si): NaN, the infinities and anything beyond +-2^31 give INT_MIN.
// (C++ leaves those conversions undefined; gfx950's v_cvt_i32_f32 saturates instead.)
__device__ __forceinline__ int pj_cvt_x86(float f) { return (f >= 2147483648.f || f < -2147483648.f || f != f) ? (int)0x80000000 : (int)f; }

// The early returns of Frame::GetFeaturesInArea (Frame.cc:602-617) with that conversion: a query whose window arithmetic leaves the
// int range -- a NaN or infinite projection, a radius of 1e30 -- selects NOTHING in the reference (nMaxCell = min(63, INT_MIN) < 0),
// where saturating conversions would scan the whole frame.  For every finite window inside the int range this says "empty" only
// where the window lies wholly beside the grid, which the scan below finds empty as well.
__device__ __forceinline__ bool pj_ref_window_empty(const ProjB& a, float x, float y, float r)
{
    if (max(0, pj_cvt_x86(floorf((x - a.fb.min_x - r) * a.inv_w))) >= GRID_COLS) return true;
    if (min(GRID_COLS - 1, pj_cvt_x86(ceilf((x - a.fb.min_x + r) * a.inv_w))) < 0) return true;
    if (max(0, pj_cvt_x86(floorf((y - a.fb.min_y - r) * a.inv_h))) >= GRID_ROWS) return true;
    if (min(GRID_ROWS - 1, pj_cvt_x86(ceilf((y - a.fb.min_y + r) * a.inv_h))) < 0) return true;
    return false;
}

__device__ __forceinline__ ProjQ load_query(const ProjB& a, int f, int iq)
{
    ProjQ r;
    if (a.form == 0) {
        const gfo_proj_query p = reinterpret_cast<const gfo_proj_query*>(a.q)[(long long)f * a.q_stride + iq];
        r.u = p.u; r.v = p.v; r.ur = p.ur; r.radius = p.radius; r.angle = p.angle;
        r.min_level = p.min_level; r.max_level = p.max_level;
        r.active = (p.flags & 1) != 0;
        r.obs = (p.flags & 4) != 0;
    } else {
        // ORBmatcher.cc:163-180: mbTrackInView && !isBad(); r = RadiusByViewingCos(viewCos) [* th];
        // window r * mvScaleFactors[level], levels [level - 1, level]
        const gfo_map_point p = reinterpret_cast<const gfo_map_point*>(a.q)[(long long)f * a.q_stride + iq];
        const int lvl = p.level;
        const bool lvl_ok = lvl >= 0 && lvl < a.nlevels;
        float rr = (double)p.view_cos > 0.998 ? 2.5f : 4.0f;   // RadiusByViewingCos, :243-249
        if (a.bfactor) rr *= a.th;
        r.u = p.proj_x; r.v = p.proj_y; r.ur = p.proj_xr; r.angle = 0.f;
        r.radius = lvl_ok ? rr * a.scale[lvl] : 0.f;
        r.min_level = lvl - 1; r.max_level = lvl;
        r.active = (p.flags & 1) && !(p.flags & 2) && lvl_ok;
        r.obs = (p.flags & 4) != 0;
    }
    r.active = r.active && !pj_ref_window_empty(a, r.u, r.v, r.radius);
    return r;
}

__device__ __forceinline__ int frame_n(const ProjB& a, int f)
{
    int n = a.n_dev ? a.n_dev[(long long)f * a.n_dev_stride] : a.n_host;
    return n < 0 ? 0 : (n > a.n_cap ? a.n_cap : n);
}

// one workgroup builds one frame's scan grid (N <= 65535); it also clears the frame's counters
__global__ __launch_bounds__(1024) void k_proj_grid(ProjB a)
{
    __shared__ int s_cnt[NSG];
    __shared__ int s_part[16];
    const int tid = threadIdx.x, f = blockIdx.x;
    if (a.cp_n16 > 0 && (int)blockIdx.x >= a.grid_frames) {   // a copy workgroup (host-array calls)
        const int t = ((int)blockIdx.x - a.grid_frames) * 1024 + tid, stride = ((int)gridDim.x - a.grid_frames) * 1024;
        for (int i = t; i < a.cp_n16; i += stride) a.cp_dst[i] = a.cp_src[i];
        return;
    }
    const int n = frame_n(a, f);
    const gfo_keypoint* kp = (a.kp_grid ? a.kp_grid : a.kp) + (long long)f * a.kp_stride;
    int* cell_start = a.cell_start + (long long)f * (NSG + 1);
    float2* cell_xy = a.cell_xy + (long long)f * a.n_cap;
    unsigned* cell_meta = a.cell_meta + (long long)f * a.n_cap;
    if (tid < PJ_CNT) a.counters[f * PJ_CNT + tid] = 0;
    for (int c = tid; c < NSG; c += 1024) s_cnt[c] = 0;
    __syncthreads();
    auto scan_cell = [&](float x, float y) {
        const int fx = min(max((int)floorf((x - a.fb.min_x) * a.sinv_w), 0), SG_COLS - 1);
        const int fy = min(max((int)floorf((y - a.fb.min_y) * a.sinv_h), 0), SG_ROWS - 1);
        return fx * SG_ROWS + fy;
    };
    // the first 4096 keypoints stay in registers between the counting and the filling pass (four a thread): on a host-array call they
    // are read from the pinned block over PCIe (kp_grid), and once is enough
    float kx[4], ky[4];
    int ko[4];
#pragma unroll
    for (int r = 0; r < 4; r++) {
        const int i = tid + r * 1024;
        kx[r] = ky[r] = 0.f; ko[r] = 0;
        if (i < n) { kx[r] = kp[i].x; ky[r] = kp[i].y; ko[r] = kp[i].octave; }
    }
    // Frame::PosInGrid, Frame.cc:648-658 (round half away from zero): only assigned keypoints can be candidates
    auto count_one = [&](float x, float y) {
        const int px = (int)roundf((x - a.fb.min_x) * a.inv_w);
        const int py = (int)roundf((y - a.fb.min_y) * a.inv_h);
        if (!(px < 0 || px >= GRID_COLS || py < 0 || py >= GRID_ROWS)) atomicAdd(&s_cnt[scan_cell(x, y)], 1);
    };
#pragma unroll
    for (int r = 0; r < 4; r++)
        if (tid + r * 1024 < n) count_one(kx[r], ky[r]);
    for (int i = tid + 4096; i < n; i += 1024) count_one(kp[i].x, kp[i].y);
    __syncthreads();
    // exclusive scan of 12288 counters: 12 per thread
    int loc[NSG / 1024], s = 0;
#pragma unroll
    for (int k = 0; k < NSG / 1024; k++) { loc[k] = s_cnt[tid * (NSG / 1024) + k]; s += loc[k]; }
    int total_in;
    int run = st_block_incl_scan(s, s_part, &total_in) - s;   // DPP scan inside the waves, 16 partial sums (k_wave.inc): two barriers
#pragma unroll
    for (int k = 0; k < NSG / 1024; k++) {
        cell_start[tid * (NSG / 1024) + k] = run;
        s_cnt[tid * (NSG / 1024) + k] = run;   // becomes the fill cursor
        run += loc[k];
    }
    if (tid == 1023) cell_start[NSG] = run;
    __syncthreads();
    auto fill_one = [&](int i, float x, float y, int octave) {
        const int px = (int)roundf((x - a.fb.min_x) * a.inv_w);
        const int py = (int)roundf((y - a.fb.min_y) * a.inv_h);
        if (px < 0 || px >= GRID_COLS || py < 0 || py >= GRID_ROWS) return;
        const int slot = atomicAdd(&s_cnt[scan_cell(x, y)], 1);
        cell_xy[slot] = make_float2(x, y);
        // octaves outside 0..15 cannot come out of the extractor (GFO_MAX_LEVELS); host arrays are checked at the ABI
        cell_meta[slot] = (unsigned)i | ((unsigned)(octave & 0xF) << 16) | ((unsigned)px << 20) | ((unsigned)py << 26);
    };
#pragma unroll
    for (int r = 0; r < 4; r++)
        if (tid + r * 1024 < n) fill_one(tid + r * 1024, kx[r], ky[r], ko[r]);
    for (int i = tid + 4096; i < n; i += 1024) fill_one(i, kp[i].x, kp[i].y, kp[i].octave);
}

__device__ __forceinline__ int hamming_u4(const uint4 a0, const uint4 a1, const uint4* __restrict__ b)
{
    const uint4 b0 = b[0], b1 = b[1];
    return __popc(a0.x ^ b0.x) + __popc(a0.y ^ b0.y) + __popc(a0.z ^ b0.z) + __popc(a0.w ^ b0.w) +
           __popc(a1.x ^ b1.x) + __popc(a1.y ^ b1.y) + __popc(a1.z ^ b1.z) + __popc(a1.w ^ b1.w);
}

// key of a candidate = (dist, cell column, cell row, index): the reference's iteration order breaks distance ties.
// The octave rides in the lowest 7 bits (it cannot disturb the order: two candidates never share an index).
//   [43:35] dist  [34:29] cell column  [28:23] cell row  [22:7] index  [6:0] octave
__device__ __forceinline__ unsigned long long cand_key(unsigned dist, unsigned meta)
{
    return ((unsigned long long)dist << 35) | ((unsigned long long)((meta >> 20) & 63u) << 29) | ((unsigned long long)(meta >> 26) << 23) |
           ((unsigned long long)(meta & 0xFFFFu) << 7) | (unsigned long long)((meta >> 16) & 0xFu);
}

// The acceptance rule on the best and second-best UNBLOCKED candidates (entries dist << 23 | octave << 16 | index,
// -1 = none): ORBmatcher.cc:228-233 / :1536.
__device__ __forceinline__ void accept_rule(const ProjB& a, int e1, int e2, int* pick, int* dist)
{
    *pick = -1;
    *dist = PJ_DIST_NONE;
    if (e1 < 0) return;
    const int bestDist = e1 >> 23;
    if (bestDist > a.th_dist) return;
    if (a.use_ratio && e2 >= 0) {
        const int bestDist2 = e2 >> 23;
        // same level and not distinctive enough (:230); no second candidate: bestLevel2 = -1, always accepted
        if (((e1 >> 16) & 0x7F) == ((e2 >> 16) & 0x7F) && (float)bestDist > a.nn_ratio * (float)bestDist2) { *dist = PJ_DIST_RATIO; return; }
    }
    *pick = e1 & 0xFFFF;
    *dist = bestDist;
}

// the gate between a candidate and the descriptor distance: true = skipped.  Tracking overloads: |ur - uR| > r where the keypoint has a
// right-image coordinate (ORBmatcher.cc:201-206).  Fuse: the chi-square test on the reprojection error (:1026-1050), float arithmetic as
// the reference writes it (sum of squares left to right, un-fused; the product with the level's inverse sigma^2; the comparison in double)
__device__ __forceinline__ bool pj_gated(const ProjB& a, const ProjQ& q, float rs, float kx, float ky, float ur, int octave)
{
    if (!a.fuse_gate) return ur > 0 && fabsf(q.ur - ur) > rs;
    const float ex = q.u - kx, ey = q.v - ky;
    if (ur >= 0) {
        const float er = q.ur - ur;
        const float e2 = ex * ex + ey * ey + er * er;
        return (double)(e2 * a.inv_sigma2[octave]) > 7.8;
    }
    const float e2 = ex * ex + ey * ey;
    return (double)(e2 * a.inv_sigma2[octave]) > 5.99;
}

// Scans the grid window of one projected point in two phases, so that no lane waits for HBM inside the divergent
// scan loop: phase 1 walks the cells and keeps the (at most PJ_HOLD) items that pass the window and level tests --
// for a 50 000-point map most lanes keep none; phase 2 fetches, for every kept item at once, what the remaining
// filters need (taken on entry, mvuRight gate, `blocked(i)`, the keypoint's descriptor) and offers the survivors to
// `sink(key)`.  Items beyond PJ_HOLD are finished on the spot (contended maps only).
#ifndef PJ_HOLD
#define PJ_HOLD 4
#endif
template <class StartT, class Blocked, class Sink>
__device__ __forceinline__ bool scan_candidates(const ProjB& a, int f, int n, int iq, const ProjQ& q, const StartT* cell_start,
                                                const float2* cell_xy, const unsigned* cell_meta, Blocked blocked, Sink sink)
{   // returns: the window holds a keypoint of the level range (the reference's vIndices is not empty)
    const float rs = q.radius, x = q.u, y = q.v;
    // the scan cells the window touches, with a hundredth of a cell of slack on either side: the window test below is
    // the reference's own (GetFeaturesInArea, Frame.cc:627-640), the cell range only has to be a superset
    // (both ends are clamped INTO the grid, as the keypoints' cells are: a keypoint slightly outside the frame bounds is
    //  still assigned -- PosInGrid rounds -- and sits in a border cell, so a window that lies entirely outside the bounds
    //  must still visit that border cell)
    const float sgx = (float)(SG_COLS - 1), sgy = (float)(SG_ROWS - 1);
    int cx0 = (int)fminf(fmaxf(floorf((x - a.fb.min_x - rs) * a.sinv_w - 0.01f), 0.f), sgx);
    int cx1 = (int)fminf(fmaxf(floorf((x - a.fb.min_x + rs) * a.sinv_w + 0.01f), 0.f), sgx);
    const int cy0 = (int)fminf(fmaxf(floorf((y - a.fb.min_y - rs) * a.sinv_h - 0.01f), 0.f), sgy);
    const int cy1 = (int)fminf(fmaxf(floorf((y - a.fb.min_y + rs) * a.sinv_h + 0.01f), 0.f), sgy);
    if (!q.active || !(rs > 0.f)) { cx0 = 0; cx1 = -1; }   // no cells (a zero or NaN radius admits nothing: the window test is strict)
    const bool check_levels = (q.min_level > 0) || (q.max_level >= 0);
    const uint8_t* desc = a.desc + (long long)f * a.kp_stride * 32;
    const float* u_right = a.u_right ? a.u_right + (long long)f * a.ur_stride : nullptr;
    const uint8_t* taken0 = a.taken0 ? a.taken0 + (long long)f * a.tk_stride : nullptr;
    uint4 a0 = make_uint4(0, 0, 0, 0), a1 = a0;
    bool have_desc = false;
    // finishes one kept item (j = position in the cell-ordered item arrays)
    auto finish = [&](bool valid, int j) {
        const unsigned meta = cell_meta[valid ? j : 0];
        const int i = min((int)(meta & 0xFFFF), n - 1);
        const uint4* dk = reinterpret_cast<const uint4*>(desc + (long long)i * 32);
        const uint4 b0 = dk[0], b1 = dk[1];
        const int tk = taken0 ? (int)taken0[i] : 0;            // F.mvpMapPoints[idx] with Observations() > 0, :197-199
        const float ur = u_right ? u_right[i] : -1.0f;         // :201-206
        if (!valid || tk || blocked(i) || (ur > 0 && fabsf(q.ur - ur) > rs)) return;   // (the fusion gate lives in the wavefront form only: pj_launch)
        const unsigned dist = (unsigned)(__popc(a0.x ^ b0.x) + __popc(a0.y ^ b0.y) + __popc(a0.z ^ b0.z) + __popc(a0.w ^ b0.w) +
                                         __popc(a1.x ^ b1.x) + __popc(a1.y ^ b1.y) + __popc(a1.z ^ b1.z) + __popc(a1.w ^ b1.w));
        sink(cand_key(dist, meta));
    };
    auto need_desc = [&]() {
        if (!have_desc) {
            const uint4* dq = reinterpret_cast<const uint4*>(a.q_desc + (long long)f * a.qd_stride + (long long)iq * 32);
            a0 = dq[0]; a1 = dq[1];
            have_desc = true;
        }
    };
    int hold[PJ_HOLD], nh = 0;
#pragma unroll
    for (int s = 0; s < PJ_HOLD; s++) hold[s] = 0;
    for (int ix = cx0; ix <= cx1; ix++) {
        // cells (ix, cy0..cy1) are contiguous in the CSR
        const int beg = (int)cell_start[ix * SG_ROWS + cy0];
        const int end = (int)cell_start[ix * SG_ROWS + cy1 + 1];
        for (int j = beg; j < end; j++) {
            const float2 it = cell_xy[j];
            if (!(fabsf(it.x - x) < rs && fabsf(it.y - y) < rs)) continue;
            if (check_levels) {
                const int oct = (int)((cell_meta[j] >> 16) & 0xF);
                if (oct < q.min_level) continue;
                if (q.max_level >= 0 && oct > q.max_level) continue;
            }
            if (nh < PJ_HOLD) {
#pragma unroll
                for (int s = 0; s < PJ_HOLD; s++) hold[s] = nh == s ? j : hold[s];
                nh++;
            } else {
                need_desc();
                finish(true, j);
            }
        }
    }
#ifdef GFO_PROJ_DEBUG
    if (a.dbg_stop == 2) { if (nh == 12345) sink(0ull); return nh > 0; }   // tools/pmc_proj_phases.sh: stop after the grid scan
#endif
    if (__builtin_amdgcn_ballot_w64(nh > 0) == 0) return false;
    if (nh > 0) need_desc();
#pragma unroll
    for (int s0 = 0; s0 < PJ_HOLD; s0 += 2) {
        if (s0 > 0 && __builtin_amdgcn_ballot_w64(nh > s0) == 0) break;
        // two items per step: their loads are independent and go out together
        finish(nh > s0, hold[s0]);
        finish(nh > s0 + 1, hold[s0 + 1]);
    }
    return nh > 0;
}

// cached form of a candidate: dist << 23 | octave << 16 | index (-1 = none).  A distance of 256 -- every bit differs --
// sets bit 31 and so reads as "none" too (every consumer tests `< 0`).  That is the reference's behaviour, not an
// accident of the packing: bestDist and bestDist2 START at 256 and are replaced on strict `<` only (ORBmatcher.cc:186-224),
// so a candidate at distance 256 becomes neither best nor second-best and the ratio test never sees its level
// (tests/test_gpu_projection.py::test_candidate_at_distance_256_is_no_candidate).
__device__ __forceinline__ int key_entry(unsigned long long key)
{
    if (key == ~0ull) return -1;
    return (int)(((unsigned)(key >> 35) << 23) | ((unsigned)(key & 0x7Fu) << 16) | (unsigned)((key >> 7) & 0xFFFFu));   // octave < 16
}

// round 0: every projected point against the entry state; the points that can ever match (best distance within
// the threshold before anyone blocks anything) get a live slot holding their PJ_K first candidates in order.
// LDSGRID: the workgroup first copies the frame's scan grid (cell table + items, 74 KB for 4128 keypoints) into LDS and
// walks its share of the points against that copy: the scan is all gathers, and a CU's address unit retires a fully
// divergent global gather at ~2 clk per distinct line where LDS serves 128 B/clk.  Grids too large for LDS (more
// than ~4000 keypoints) are read from HBM in place.
template <bool LDSGRID>
__global__ __launch_bounds__(LDSGRID ? 1024 : 256, 8) void k_proj_round0(ProjB a, int chunk)
{
    extern __shared__ __attribute__((aligned(16))) uint8_t lds_grid[];
    const int f = blockIdx.y;
    const int n = frame_n(a, f);
    const int* g_start = a.cell_start + (long long)f * (NSG + 1);
    const float2* cell_xy = a.cell_xy + (long long)f * a.n_cap;
    const unsigned* cell_meta = a.cell_meta + (long long)f * a.n_cap;
    const unsigned short* l_start = nullptr;
    const int lane = threadIdx.x & 63;
    if (LDSGRID) {
        // LDS copy of the frame's scan grid: positions, meta words, cell table as u16 (N <= 65535)
        float2* w_xy = reinterpret_cast<float2*>(lds_grid);
        unsigned* w_meta = reinterpret_cast<unsigned*>(lds_grid + (size_t)a.n_cap * 8);
        unsigned short* w_start = reinterpret_cast<unsigned short*>(lds_grid + (size_t)a.n_cap * 12);
        const int n_in = min(g_start[NSG], a.n_cap);   // keypoints inside the grid
        for (int i = threadIdx.x; i < n_in; i += 1024) { w_xy[i] = cell_xy[i]; w_meta[i] = cell_meta[i]; }
        for (int i = threadIdx.x; i <= NSG; i += 1024) w_start[i] = (unsigned short)g_start[i];
        __syncthreads();
        cell_xy = w_xy;
        cell_meta = w_meta;
        l_start = w_start;
    }
    const int q_begin = blockIdx.x * chunk;
    const int q_end = min(a.m, q_begin + chunk);
    for (int iq0 = q_begin; iq0 < q_end; iq0 += (int)blockDim.x) {
        const int iq = iq0 + (int)threadIdx.x;
        int pick = -1, dist = 256;
        bool live = false, obs = false, trunc = false, seen = false;
        unsigned long long k[PJ_K];   // the PJ_K smallest keys, ascending
#pragma unroll
        for (int s = 0; s < PJ_K; s++) k[s] = ~0ull;
        if (iq < q_end && n > 0) {
            const ProjQ q = load_query(a, f, iq);
            obs = q.obs;
#ifdef GFO_PROJ_DEBUG
            if (a.dbg_stop == 1) { if (q.radius == 12345.f) live = true; } else   // stop after the query load
#endif
            {
                auto sink0 = [&](unsigned long long key) {
                    unsigned long long x = key;
#pragma unroll
                    for (int s = 0; s < PJ_K; s++) {
                        const unsigned long long lo = k[s] < x ? k[s] : x;
                        x = k[s] < x ? x : k[s];
                        k[s] = lo;
                    }
                    if (x != ~0ull) trunc = true;
                };
                if (LDSGRID) seen = scan_candidates(a, f, n, iq, q, l_start, cell_xy, cell_meta, [](int) { return false; }, sink0);
                else seen = scan_candidates(a, f, n, iq, q, g_start, cell_xy, cell_meta, [](int) { return false; }, sink0);
                if (k[0] != ~0ull) {
                    const int e1 = key_entry(k[0]), e2 = key_entry(k[1]);
                    live = (e1 >> 23) <= a.th_dist;   // :228 / :1536
                    accept_rule(a, e1, e2, &pick, &dist);
                }
            }
        }
        if (a.out_q && iq < q_end) a.out_q[(long long)f * a.m + iq] = seen ? PJ_PT_FAR : PJ_PT_NONE;   // the live points' entries: k_proj_resolve
        const unsigned long long mask = __builtin_amdgcn_ballot_w64(live);
        if (mask == 0) continue;
        const int leader = __ffsll((long long)mask) - 1;
        int base = 0;
        if (lane == leader) base = atomicAdd(&a.counters[f * PJ_CNT + PJ_NLIVE], __popcll(mask));
        base = __shfl(base, leader);
        if (live) {
            const long long slot = (long long)f * a.m + base + __popcll(mask & ((1ull << lane) - 1));
            a.live[slot] = (unsigned)iq | (obs ? 0x80000000u : 0u);
            a.pick[slot] = pick;
            a.pick_dist[slot] = dist;
            a.spill_off[slot] = -1;
            int e[PJ_K], cnt = 0;
#pragma unroll
            for (int s = 0; s < PJ_K; s++) {
                e[s] = key_entry(k[s]);
                cnt += k[s] != ~0ull;
            }
            a.cand[2 * slot] = make_uint4((unsigned)cnt | (trunc ? 0x100u : 0u), (unsigned)e[0], (unsigned)e[1], (unsigned)e[2]);
            a.cand[2 * slot + 1] = make_uint4((unsigned)e[3], (unsigned)e[4], (unsigned)e[5], (unsigned)e[6]);
        }
    }
}

// Minimum over the wave of a candidate key (44 bits, ~0 = none) in two 32-bit DPP reductions: the upper 21 bits (distance, reference
// cell), then the lower 23 (index, octave) among the lanes that hold the winning upper part.
__device__ __forceinline__ unsigned long long pj_wave_min_key(unsigned long long x)
{
    const unsigned hi = (unsigned)(x >> 23), lo = (unsigned)x & 0x7FFFFFu;
    const unsigned mh = st_wave_min(hi);
    if (mh == 0xFFFFFFFFu) return ~0ull;
    const unsigned ml = st_wave_min(hi == mh ? lo : 0xFFFFFFFFu);
    return ((unsigned long long)mh << 23) | ml;
}

// round 0 for FEW points (frames * m <= 16 384: the call Tracking makes per frame -- a thousand or two map points, or the last
// frame's tracked points, against ONE frame): a wavefront per projected point.  One thread a point makes the call a chain of
// dependent ~1 us loads as long as the point's window has items (1500 points at th 7: 105 us on six workgroups); here the lanes
// take the window's items side by side -- column ranges of the scan grid, flattened over the lanes -- so a point costs four
// round trips to memory whatever its window holds, and the points spread over every CU.  Same outputs as k_proj_round0: the
// PJ_K smallest keys in order, the count, the truncation mark, the pick under the entry state.
// (Sixteen points a workgroup, and ONE atomic on the frame's live counter per workgroup: the L2 retires atomics on one address at
//  ~20 ns each -- a returning atomic per point made the 16 000-point call 0.42 ms instead of 0.31.)
#define PJ_WAVES 16
__global__ __launch_bounds__(64 * PJ_WAVES) void k_proj_round0_wave(ProjB a)
{
    __shared__ int s_n, s_base, s_sp_base;
    __shared__ int s_want[PJ_WAVES];
    const int f = blockIdx.y;
    const int n = frame_n(a, f);
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
    const int iq = blockIdx.x * PJ_WAVES + wave;
    if (threadIdx.x == 0) s_n = 0;
    ProjQ q{};
    if (iq < a.m && n > 0) q = load_query(a, f, iq);
    const float rs = q.radius, x = q.u, y = q.v;
    const bool scan = iq < a.m && n > 0 && q.active && rs > 0.f;   // a zero or NaN radius admits nothing: the window test is strict
    const int* g_start = a.cell_start + (long long)f * (NSG + 1);
    const float2* cell_xy = a.cell_xy + (long long)f * a.n_cap;
    const unsigned* cell_meta = a.cell_meta + (long long)f * a.n_cap;
    const uint8_t* desc = a.desc + (long long)f * a.kp_stride * 32;
    const float* u_right = a.u_right ? a.u_right + (long long)f * a.ur_stride : nullptr;
    const uint8_t* taken0 = a.taken0 ? a.taken0 + (long long)f * a.tk_stride : nullptr;
    uint4 a0 = make_uint4(0, 0, 0, 0), a1 = a0;
    if (scan) {
        const uint4* dq = reinterpret_cast<const uint4*>(a.q_desc + (long long)f * a.qd_stride + (long long)iq * 32);
        a0 = dq[0]; a1 = dq[1];
    }
    // the scan cells the window touches: see scan_candidates
    const float sgx = (float)(SG_COLS - 1), sgy = (float)(SG_ROWS - 1);
    const int cx0 = (int)fminf(fmaxf(floorf((x - a.fb.min_x - rs) * a.sinv_w - 0.01f), 0.f), sgx);
    const int cx1 = scan ? (int)fminf(fmaxf(floorf((x - a.fb.min_x + rs) * a.sinv_w + 0.01f), 0.f), sgx) : cx0 - 1;   // not scanning: no columns
    const int cy0 = (int)fminf(fmaxf(floorf((y - a.fb.min_y - rs) * a.sinv_h - 0.01f), 0.f), sgy);
    const int cy1 = (int)fminf(fmaxf(floorf((y - a.fb.min_y + rs) * a.sinv_h + 0.01f), 0.f), sgy);
    const bool check_levels = (q.min_level > 0) || (q.max_level >= 0);
    // lane c holds column cg + c of a group of up to 64: cells (column, cy0..cy1) are contiguous in the CSR
    int beg = 0, cnt = 0;
    auto load_columns = [&](int cg, int ncol) {
        beg = 0; cnt = 0;
        if (lane < ncol) {
            beg = g_start[(cg + lane) * SG_ROWS + cy0];
            cnt = g_start[(cg + lane) * SG_ROWS + cy1 + 1] - beg;
        }
    };
    load_columns(cx0, min(64, cx1 - cx0 + 1));
    // Room for the point's FULL candidate list, for k_proj_resolve to fall back on when the point's PJ_K cached candidates are all
    // taken (contended maps: several points per keypoint, wide windows): the items of the window bound it.  One cursor for the call
    // (frame 0's counter), advanced once per workgroup.
    const int items0 = __builtin_amdgcn_readlane(st_wave_incl_scan(cnt), 63);
    const int want = cx1 - cx0 < 64 && items0 > PJ_K ? items0 : 0;
    if (lane == 0) s_want[wave] = want;
    __syncthreads();
    if (threadIdx.x == 0) {
        int sum = 0;
        for (int w = 0; w < PJ_WAVES; w++) sum += s_want[w];
        s_sp_base = sum ? atomicAdd(&a.counters[PJ_SPILL], sum) : 0;
    }
    __syncthreads();
    int sp_off = -1;
    if (want) {
        int off = s_sp_base;
        for (int w = 0; w < wave; w++) off += s_want[w];
        if (off >= 0 && off + want <= a.spill_cap) sp_off = off;   // pool exhausted: the point keeps the grid rescan
    }
    unsigned long long k[PJ_K];   // the PJ_K smallest keys of the point, ascending; the same in every lane
#pragma unroll
    for (int s = 0; s < PJ_K; s++) k[s] = ~0ull;
    int total = 0;
    bool seen = false;   // vIndices not empty
    for (int cg = cx0; cg <= cx1; cg += 64) {
        const int ncol = min(64, cx1 - cg + 1);
        if (cg != cx0) load_columns(cg, ncol);
        const int incl = st_wave_incl_scan(cnt);
        const int excl = incl - cnt;
        const int T = __builtin_amdgcn_readlane(incl, 63);
        for (int t0 = 0; t0 < T; t0 += 64) {
            const int t = t0 + lane;
            int j = -1;
            for (int cc = 0; cc < ncol; cc++) {
                const int p = __builtin_amdgcn_readlane(excl, cc), cn = __builtin_amdgcn_readlane(cnt, cc), b = __builtin_amdgcn_readlane(beg, cc);
                if (t >= p && t < p + cn) j = b + (t - p);
            }
            unsigned long long key = ~0ull;
            if (j >= 0) {
                const float2 it = cell_xy[j];
                const unsigned meta = cell_meta[j];
                bool ok = fabsf(it.x - x) < rs && fabsf(it.y - y) < rs;   // GetFeaturesInArea, Frame.cc:627-640
                if (ok && check_levels) {
                    const int oct = (int)((meta >> 16) & 0xF);
                    ok = !(oct < q.min_level) && !(q.max_level >= 0 && oct > q.max_level);
                }
                if (ok) {
                    seen = true;
                    const int i = min((int)(meta & 0xFFFF), n - 1);
                    const uint4* dk = reinterpret_cast<const uint4*>(desc + (long long)i * 32);
                    const uint4 b0 = dk[0], b1 = dk[1];
                    const int tk = taken0 ? (int)taken0[i] : 0;            // F.mvpMapPoints[idx] with Observations() > 0, :197-199
                    const float ur = u_right ? u_right[i] : -1.0f;         // :201-206
                    if (!(tk || pj_gated(a, q, rs, it.x, it.y, ur, (int)((meta >> 16) & 0xF)))) {
                        const unsigned dist = (unsigned)(__popc(a0.x ^ b0.x) + __popc(a0.y ^ b0.y) + __popc(a0.z ^ b0.z) + __popc(a0.w ^ b0.w) +
                                                         __popc(a1.x ^ b1.x) + __popc(a1.y ^ b1.y) + __popc(a1.z ^ b1.z) + __popc(a1.w ^ b1.w));
                        key = cand_key(dist, meta);
                    }
                }
            }
            const unsigned long long have = __builtin_amdgcn_ballot_w64(key != ~0ull);
            if (sp_off >= 0 && key != ~0ull) a.spill[sp_off + total + __popcll(have & ((1ull << lane) - 1))] = key;
            total += __popcll(have);
            // at most PJ_K keys of this batch can enter the list: take the batch's minimum while it beats the list's last
            for (int s = 0; s < PJ_K; s++) {
                const unsigned long long mk = pj_wave_min_key(key);
                if (mk >= k[PJ_K - 1]) break;
                if (key == mk) key = ~0ull;
                unsigned long long xk = mk;
#pragma unroll
                for (int s2 = 0; s2 < PJ_K; s2++) {
                    const unsigned long long lo = k[s2] < xk ? k[s2] : xk;
                    xk = k[s2] < xk ? xk : k[s2];
                    k[s2] = lo;
                }
            }
        }
    }
    int pick = -1, dist = 256, rank = 0;
    bool live = false;
    if (k[0] != ~0ull) {
        const int e1 = key_entry(k[0]), e2 = key_entry(k[1]);
        live = (e1 >> 23) <= a.th_dist;   // :228 / :1536: otherwise it can never match
        accept_rule(a, e1, e2, &pick, &dist);
    }
    if (a.out_q && iq < a.m) {   // the live points' entries: k_proj_resolve
        const bool any = __builtin_amdgcn_ballot_w64(seen) != 0;
        if (lane == 0) a.out_q[(long long)f * a.m + iq] = any ? PJ_PT_FAR : PJ_PT_NONE;
    }
    if (live && lane == 0) rank = atomicAdd(&s_n, 1);
    __syncthreads();
    if (threadIdx.x == 0 && s_n > 0) s_base = atomicAdd(&a.counters[f * PJ_CNT + PJ_NLIVE], s_n);
    __syncthreads();
    if (!live || lane != 0) return;
    const long long slot = (long long)f * a.m + s_base + rank;
    a.live[slot] = (unsigned)iq | (q.obs ? 0x80000000u : 0u);
    a.pick[slot] = pick;
    a.pick_dist[slot] = dist;
    a.spill_off[slot] = total > PJ_K ? sp_off : -1;
    int e[PJ_K];
#pragma unroll
    for (int s = 0; s < PJ_K; s++) e[s] = key_entry(k[s]);
    const int ncached = total < PJ_K ? total : PJ_K;
    // header: cached entries | truncated << 8 | length of the full list << 9 (with spill_off)
    a.cand[2 * slot] = make_uint4((unsigned)ncached | (total > PJ_K ? 0x100u : 0u) | ((unsigned)total << 9), (unsigned)e[0], (unsigned)e[1], (unsigned)e[2]);
    a.cand[2 * slot + 1] = make_uint4((unsigned)e[3], (unsigned)e[4], (unsigned)e[5], (unsigned)e[6]);
}

// One workgroup per frame: claim / re-evaluate rounds until nothing changes, then the epilogue:
//   owner of a keypoint = the LAST accepted point that picked it (:233 overwrites), its distance the score;
//   rotation consistency (ORBmatcher.cc:1548-1591): bin of every accepted point, histogram, the reference's
//   three-maxima scan, then every point in a discarded bin clears the keypoint it took and costs one match.
// TABG: the per-keypoint tables live in HBM (more than 12 288 keypoints) and are read past the L1, because the
// other waves of the workgroup update them with L2 atomics.
template <bool TABG>
__global__ __launch_bounds__(1024) void k_proj_resolve(ProjB a)
{
    extern __shared__ int lds_tab[];
    __shared__ int histo[HISTO_LENGTH];
    __shared__ int keep[3];
    __shared__ int s_acc[3];
    const int tid = threadIdx.x, f = blockIdx.x;
    const int n = frame_n(a, f);
    int* tab = TABG ? a.tab_g + 2LL * f * a.n_cap : lds_tab;   // claim table, then owner table
    int* sc = tab + (TABG ? a.n_cap : n);                       // score of the owner
    int* pick = a.pick + (long long)f * a.m;
    int* pick_dist = a.pick_dist + (long long)f * a.m;
    const unsigned* live = a.live + (long long)f * a.m;
    const uint4* cand = a.cand + 2LL * f * a.m;
    int* rot_bin = a.rot_bin + (long long)f * a.m;
    int* out_mp = a.out_mp + (long long)f * a.n_cap;
    int* out_score = a.out_score + (long long)f * a.n_cap;
    auto tab_load = [&](const int* p) { return TABG ? __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : *p; };
    int nlive = a.counters[f * PJ_CNT + PJ_NLIVE];
    if (nlive > a.m) nlive = a.m;
    if (tid < 3) s_acc[tid] = 0;
    int rounds = 0, fallbacks = 0;
    // one live point under the current claims: the first two candidates of its cached list that no lower point holds.
    // false: the cached prefix ran out before a best and a second were found and the point has more candidates than the cache
    // holds -- it needs `rescan` (rare)
    auto from_cache = [&](int iq, const uint4& c0, const uint4* c1p, int* e1, int* e2) {
        const int cnt = (int)(c0.x & 0xFF);
        *e1 = -1; *e2 = -1;
        auto offer = [&](int s, int ent) {
            if (s < cnt && *e2 < 0) {
                const bool free_ = tab_load(&tab[ent & 0xFFFF]) >= iq;   // not claimed by a lower point
                if (free_) { if (*e1 < 0) *e1 = ent; else *e2 = ent; }
            }
        };
        offer(0, (int)c0.y); offer(1, (int)c0.z); offer(2, (int)c0.w);
        if (cnt > 3 && *e2 < 0) {   // the second half of the cached list: fetched only by the points that get this far
            const uint4 c1 = *c1p;
            offer(3, (int)c1.x); offer(4, (int)c1.y); offer(5, (int)c1.z); offer(6, (int)c1.w);
        }
        return !((c0.x & 0x100u) && *e2 < 0);
    };
    // full re-evaluation of live point t (projected point iq) under the current claims: from its full candidate list when round 0
    // left one (k_proj_round0_wave: the static filters are already applied, what remains is "not claimed by a lower point"),
    // otherwise from the grid
    const int* spill_off = a.spill_off + (long long)f * a.m;
    auto rescan = [&](int t, int iq, unsigned header, int* e1, int* e2) {
        unsigned long long k1 = ~0ull, k2 = ~0ull;
        auto two_smallest = [&](unsigned long long x) {
            const unsigned long long hi = x < k1 ? k1 : x;
            k1 = x < k1 ? x : k1;
            k2 = hi < k2 ? hi : k2;
        };
        const int so = spill_off[t];
        if (so >= 0) {
            const unsigned long long* sp = a.spill + so;
            const int len = (int)(header >> 9);
            int j = 0;
            for (; j + 4 <= len; j += 4) {   // four keys in flight
                const unsigned long long x0 = sp[j], x1 = sp[j + 1], x2 = sp[j + 2], x3 = sp[j + 3];
                const int t0 = tab_load(&tab[(x0 >> 7) & 0xFFFFu]), t1 = tab_load(&tab[(x1 >> 7) & 0xFFFFu]),
                          t2 = tab_load(&tab[(x2 >> 7) & 0xFFFFu]), t3 = tab_load(&tab[(x3 >> 7) & 0xFFFFu]);
                if (t0 >= iq) two_smallest(x0);
                if (t1 >= iq) two_smallest(x1);
                if (t2 >= iq) two_smallest(x2);
                if (t3 >= iq) two_smallest(x3);
            }
            for (; j < len; j++) {
                const unsigned long long x0 = sp[j];
                if (tab_load(&tab[(x0 >> 7) & 0xFFFFu]) >= iq) two_smallest(x0);
            }
        } else {
            const ProjQ q = load_query(a, f, iq);
            scan_candidates(a, f, n, iq, q, a.cell_start + (long long)f * (NSG + 1), a.cell_xy + (long long)f * a.n_cap,
                            a.cell_meta + (long long)f * a.n_cap, [&](int i) { return tab_load(&tab[i]) < iq; }, two_smallest);
        }
        *e1 = key_entry(k1);
        *e2 = key_entry(k2);
        fallbacks++;
    };
    if (!TABG && nlive > 0 && nlive <= 1024 * PJ_RR) {   // (TABG: more than 12 288 keypoints, the tables are in device memory anyway)
        // The usual case -- a frame's few thousand live points: every thread keeps its points (index, cached candidates, pick) in
        // registers for all the rounds, so a round is LDS traffic and three barriers; walking the live arrays in device memory
        // every round made a lone call's resolve a chain of ~1 us loads (6 rounds over 1721 points: 44 us).
        unsigned lv[PJ_RR];
        int pk[PJ_RR];     // (the pick's distance is a function of the pick: it goes to pick_dist[] when the pick changes, not in a register)
        uint4 c0[PJ_RR];   // header + the first three candidates; the other four stay in memory (from_cache)
#pragma unroll
        for (int r = 0; r < PJ_RR; r++) {
            const int t = tid + r * 1024;
            lv[r] = 0; pk[r] = -1;
            c0[r] = make_uint4(0, 0, 0, 0);
            if (t < nlive) { lv[r] = live[t]; pk[r] = pick[t]; c0[r] = cand[2 * t]; }
        }
        // two claim tables, used in turn (the second one is the score table's place, idle until the epilogue): the table of the next
        // round is cleared while this round's is being read -- two barriers a round instead of three
        int* const tab_a = tab;
        int* const tab_b = sc;
        for (int k = tid; k < n; k += 1024) tab[k] = 0x7FFFFFFF;
        __syncthreads();
        for (;;) {
            rounds++;
#pragma unroll
            for (int r = 0; r < PJ_RR; r++)
                if ((lv[r] & 0x80000000u) && pk[r] >= 0) atomicMin(&tab[pk[r]], (int)(lv[r] & 0x7FFFFFFFu));
            __syncthreads();
            int* const tab_next = tab == tab_a ? tab_b : tab_a;
            for (int k = tid; k < n; k += 1024) tab_next[k] = 0x7FFFFFFF;
            int changed = 0;
            unsigned redo = 0;   // slots whose cache ran out: rescanned below, ONE inlined copy of the grid scan
#pragma unroll
            for (int r = 0; r < PJ_RR; r++) {
                if (tid + r * 1024 < nlive) {
                    int e1, e2, np, nd;
                    if (from_cache((int)(lv[r] & 0x7FFFFFFFu), c0[r], &cand[2 * (tid + r * 1024) + 1], &e1, &e2)) {
                        accept_rule(a, e1, e2, &np, &nd);
                        if (np != pk[r]) { changed = 1; pk[r] = np; pick_dist[tid + r * 1024] = nd; }
                        else if (a.out_q && np < 0) pick_dist[tid + r * 1024] = nd;   // WHY it matches nothing may change while the pick does not
                    } else redo |= 1u << r;
                }
            }
            while (redo) {
                const int r = __builtin_ctz(redo);
                redo &= redo - 1;
                unsigned e = lv[0], hdr = c0[0].x;
#pragma unroll
                for (int r2 = 1; r2 < PJ_RR; r2++) { e = r == r2 ? lv[r2] : e; hdr = r == r2 ? c0[r2].x : hdr; }
                int e1, e2, np, nd;
                rescan(tid + r * 1024, (int)(e & 0x7FFFFFFFu), hdr, &e1, &e2);
                accept_rule(a, e1, e2, &np, &nd);
#pragma unroll
                for (int r2 = 0; r2 < PJ_RR; r2++)
                    if (r == r2 && np != pk[r2]) { changed = 1; pk[r2] = np; pick_dist[tid + r2 * 1024] = nd; }
                if (a.out_q && np < 0) pick_dist[tid + r * 1024] = nd;
            }
            if (!__syncthreads_or(changed)) break;
            if (rounds > nlive + 1) {   // cannot happen (point i is final after rank(i) rounds); never spin
                if (tid == 0) a.counters[f * PJ_CNT + PJ_ERR] = 1;
                break;
            }
            tab = tab_next;
        }
        tab = tab_a;
#pragma unroll
        for (int r = 0; r < PJ_RR; r++) {   // the epilogue below reads them back with the same thread
            const int t = tid + r * 1024;
            if (t < nlive) pick[t] = pk[r];
        }
    } else if (nlive > 0) {
        for (;;) {
            rounds++;
            for (int k = tid; k < n; k += 1024) tab[k] = 0x7FFFFFFF;
            __syncthreads();
            for (int t = tid; t < nlive; t += 1024) {
                const unsigned e = live[t];
                const int k = pick[t];
                if ((e & 0x80000000u) && k >= 0) atomicMin(&tab[k], (int)(e & 0x7FFFFFFFu));
            }
            __syncthreads();
            int changed = 0;
            for (int t = tid; t < nlive; t += 1024) {
                const int iq = (int)(live[t] & 0x7FFFFFFFu);
                int e1, e2, np, nd;
                const uint4 c0 = cand[2 * t];
                if (!from_cache(iq, c0, &cand[2 * t + 1], &e1, &e2)) rescan(t, iq, c0.x, &e1, &e2);
                accept_rule(a, e1, e2, &np, &nd);
                if (np != pick[t] || nd != pick_dist[t]) {
                    changed = 1;
                    pick[t] = np;
                    pick_dist[t] = nd;
                }
            }
            if (!__syncthreads_or(changed)) break;
            if (rounds > nlive + 1) {   // cannot happen (point i is final after rank(i) rounds); never spin
                if (tid == 0) a.counters[f * PJ_CNT + PJ_ERR] = 1;
                break;
            }
        }
    }
    // ---- BUDGETING_FEATURE_MATCHING (ORBmatcher.h:36-37, ORBmatcher.cc:1547-1552): the reference's loop over the queries ends with
    //      the query whose match makes nmatches reach the budget.  A query's outcome depends on earlier queries only, so the budgeted
    //      answer is the fixed point above with every query behind that one taken out: the rank of an accepted query among the accepted
    //      ones, in query order, by counting (a variant that is off by default: the live list is in no particular order and is not
    //      sorted for it), the K-th one's index is the cut.  That K-th match is kept but never enters the rotation histogram (the
    //      reference breaks in front of it).
    __shared__ int s_cut;
    if (a.max_matches > 0) {
        if (tid == 0) s_cut = 0x7FFFFFFF;
        __threadfence_block();
        __syncthreads();
        for (int t = tid; t < nlive; t += 1024) {
            if (pick[t] < 0) continue;
            const int iq = (int)(live[t] & 0x7FFFFFFFu);
            int rank = 0;
            for (int u = 0; u < nlive; u++) rank += (pick[u] >= 0 && (int)(live[u] & 0x7FFFFFFFu) < iq) ? 1 : 0;
            if (rank == a.max_matches - 1) s_cut = iq;     // (one thread at most: ranks are distinct)
        }
        __syncthreads();
        const int cut = s_cut;
        __syncthreads();
        for (int t = tid; t < nlive; t += 1024)
            if (pick[t] >= 0 && (int)(live[t] & 0x7FFFFFFFu) > cut) pick[t] = -1;
        __threadfence_block();
        __syncthreads();
    }
    // ---- per-point outcomes (round 0 wrote PJ_PT_NONE / PJ_PT_FAR for every point; the live ones are settled here) ----
    if (a.out_q) {
        int* oq = a.out_q + (long long)f * a.m;
        for (int t = tid; t < nlive; t += 1024) {
            const int k = pick[t], d = pick_dist[t];
            oq[live[t] & 0x7FFFFFFFu] = k >= 0 ? (k | (d << 16)) : (d == PJ_DIST_RATIO ? PJ_PT_RATIO : PJ_PT_FAR);
        }
        if (a.h_out_q) {
            __threadfence_block();
            __syncthreads();
            for (int i = tid; i < a.m; i += 1024) a.h_out_q[i] = oq[i];
        }
    }
    // ---- epilogue ----
    if (tid < HISTO_LENGTH) histo[tid] = 0;
    for (int k = tid; k < n; k += 1024) { tab[k] = -1; sc[k] = 0; }
    __syncthreads();
    int cnt = 0;
    for (int t = tid; t < nlive; t += 1024) {
        const int k = pick[t];
        if (k < 0) continue;
        const int iq = (int)(live[t] & 0x7FFFFFFFu);
        atomicMax(&tab[k], iq);
        cnt++;
        if (a.check_ori && a.max_matches > 0 && iq == s_cut) rot_bin[t] = HISTO_LENGTH;   // the match that reached the budget: in no bin, never cleared
        else if (a.check_ori) {
            const ProjQ q = load_query(a, f, iq);
            const float ka = a.kp_angle ? a.kp_angle[(long long)f * a.ang_stride + k] : a.kp[(long long)f * a.kp_stride + k].angle;
            float rot = q.angle - ka;
            if (rot < 0.0f) rot += 360.0f;
            int bin = (int)roundf(rot * (1.0f / HISTO_LENGTH));
            if (bin == HISTO_LENGTH) bin = 0;
            atomicAdd(&histo[bin], 1);
            rot_bin[t] = bin;
        }
    }
    if (cnt) atomicAdd(&s_acc[0], cnt);
    if (fallbacks) atomicAdd(&s_acc[2], fallbacks);
    __syncthreads();
    for (int t = tid; t < nlive; t += 1024) {   // the owner's distance is the keypoint's score
        const int k = pick[t];
        if (k >= 0 && tab_load(&tab[k]) == (int)(live[t] & 0x7FFFFFFFu)) {
            if (TABG) __hip_atomic_store(&sc[k], pick_dist[t], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            else sc[k] = pick_dist[t];
        }
    }
    if (a.check_ori) {
        if (tid == 0) {  // ComputeThreeMaxima, ORBmatcher.cc:1723-1764
            int max1 = 0, max2 = 0, max3 = 0, ind1 = -1, ind2 = -1, ind3 = -1;
            for (int i = 0; i < HISTO_LENGTH; i++) {
                const int s = histo[i];
                if (s > max1) { max3 = max2; max2 = max1; max1 = s; ind3 = ind2; ind2 = ind1; ind1 = i; }
                else if (s > max2) { max3 = max2; max2 = s; ind3 = ind2; ind2 = i; }
                else if (s > max3) { max3 = s; ind3 = i; }
            }
            if ((float)max2 < 0.1f * (float)max1) { ind2 = -1; ind3 = -1; }
            else if ((float)max3 < 0.1f * (float)max1) ind3 = -1;
            keep[0] = ind1; keep[1] = ind2; keep[2] = ind3;
        }
        __syncthreads();
        int drop = 0;
        for (int t = tid; t < nlive; t += 1024) {
            const int k = pick[t];
            if (k < 0) continue;
            const int b = rot_bin[t];
            if (b < HISTO_LENGTH && b != keep[0] && b != keep[1] && b != keep[2]) {
                // -2: matched by this call, then cleared by its rotation check (the reference stores NULL there, :1586)
                if (TABG) __hip_atomic_store(&tab[k], -2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                else tab[k] = -2;   // benign race: every writer stores the same value
                drop++;
            }
        }
        if (drop) atomicAdd(&s_acc[1], drop);
    }
    __syncthreads();
    for (int k = tid; k < n; k += 1024) {
        const int mp = tab_load(&tab[k]), score = tab_load(&sc[k]);
        out_mp[k] = mp;
        out_score[k] = score;
        if (a.h_out_mp) { a.h_out_mp[k] = mp; a.h_out_score[k] = score; }
    }
    if (tid == 0) {
        a.counters[f * PJ_CNT + PJ_NMATCH] = s_acc[0] - s_acc[1];
        a.counters[f * PJ_CNT + PJ_ROUNDS] = rounds + 1;
        a.counters[f * PJ_CNT + PJ_FALLBACK] = s_acc[2];
        if (a.h_counters) {
            a.h_counters[PJ_NLIVE] = nlive;
            a.h_counters[PJ_ROUNDS] = rounds + 1;
            a.h_counters[PJ_NMATCH] = s_acc[0] - s_acc[1];
            a.h_counters[PJ_ERR] = a.counters[f * PJ_CNT + PJ_ERR];   // (written by this thread, if at all)
            a.h_counters[PJ_FALLBACK] = s_acc[2];
        }
    }
}

// ------------------------------------------------------------------------------------------------------------
// The candidate table (gfo_projection_candidates): ORBmatcher::GetCandidates (include/ORBmatcher.h:152-172) for every map point at
// once -- pMP->mvMatchCandidates = F.GetFeaturesInArea(...) -- and, per candidate, the two things MatchCandidates /
// SearchByProjection_OnePoint compute that do not depend on what the frame's slots hold: the mvuRight gate (:118-123) and the
// descriptor distance (:127).  A wavefront per point, two passes over the same scan: PASS 0 counts, k_proj_cand_scan turns the counts
// into offsets, PASS 1 writes.  GetFeaturesInArea returns a window's keypoints in (grid column, grid row, index) order; the scan grid
// has its own cells, so a point's entries are ranked by that key before they are written (in LDS up to PJ_CAND_LDS candidates -- every
// tracking window and the 100-px windows of SearchForInitialization -- through device memory beyond).
//   entry: keypoint index | octave << 16 | distance << 20 | (mvuRight gate closed) << 31
#define PJ_CAND_LDS 256   // candidates of one point ranked in LDS (32 KB per workgroup of 16 points); a longer list goes through device memory
template <int PASS>
__global__ __launch_bounds__(64 * PJ_WAVES) void k_proj_candidates(ProjB a, int* start, unsigned* cand, unsigned* h_cand, unsigned long long* tmp, int cap)
{
    __shared__ unsigned long long s_keys[PASS ? PJ_WAVES : 1][PASS ? PJ_CAND_LDS : 1];
    __shared__ int2 s_col[PJ_WAVES][64];
    const int n = frame_n(a, 0);
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
    const int iq = blockIdx.x * PJ_WAVES + wave;
    ProjQ q{};
    if (iq < a.m && n > 0) q = load_query(a, 0, iq);
    const float rs = q.radius, x = q.u, y = q.v;
    bool scan = iq < a.m && n > 0 && q.active && rs > 0.f;
    int base = 0, T = 0;
    if (PASS == 1) {
        if (iq < a.m) { base = start[iq]; T = start[iq + 1] - base; }
        scan = scan && T > 0 && base >= 0 && (long long)base + T <= (long long)cap;   // a table that does not fit the caller's array is not written (GFO_ERR_CAPACITY)
    }
    const int* g_start = a.cell_start;
    const float2* cell_xy = a.cell_xy;
    const unsigned* cell_meta = a.cell_meta;
    uint4 a0 = make_uint4(0, 0, 0, 0), a1 = a0;
    if (PASS == 1 && scan) {
        const uint4* dq = reinterpret_cast<const uint4*>(a.q_desc + (long long)iq * 32);
        a0 = dq[0]; a1 = dq[1];
    }
    // the scan cells the window touches: see scan_candidates
    const float sgx = (float)(SG_COLS - 1), sgy = (float)(SG_ROWS - 1);
    const int cx0 = (int)fminf(fmaxf(floorf((x - a.fb.min_x - rs) * a.sinv_w - 0.01f), 0.f), sgx);
    const int cx1 = scan ? (int)fminf(fmaxf(floorf((x - a.fb.min_x + rs) * a.sinv_w + 0.01f), 0.f), sgx) : cx0 - 1;
    const int cy0 = (int)fminf(fmaxf(floorf((y - a.fb.min_y - rs) * a.sinv_h - 0.01f), 0.f), sgy);
    const int cy1 = (int)fminf(fmaxf(floorf((y - a.fb.min_y + rs) * a.sinv_h + 0.01f), 0.f), sgy);
    const bool check_levels = (q.min_level > 0) || (q.max_level >= 0);
    int total = 0;
    for (int cg = cx0; cg <= cx1; cg += 64) {
        const int ncol = min(64, cx1 - cg + 1);
        int beg = 0, cnt = 0;
        if (lane < ncol) {   // lane c holds column cg + c: cells (column, cy0..cy1) are contiguous in the CSR
            beg = g_start[(cg + lane) * SG_ROWS + cy0];
            cnt = g_start[(cg + lane) * SG_ROWS + cy1 + 1] - beg;
        }
        const int incl = st_wave_incl_scan(cnt);
        const int excl = incl - cnt;
        const int items = __builtin_amdgcn_readlane(incl, 63);
        // item t of the window belongs to the LAST column whose first item is <= t (empty columns share their successor's start): a
        // six-step search of the wave's own 64 starts in LDS.  (A loop over the columns with three readlanes each cost 37 + 44 us per call
        // on the 35-column windows of SearchForInitialization; LDS operations of one wave execute in order, a compiler fence suffices.)
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        s_col[wave][lane] = make_int2(excl, beg);             // lanes beyond ncol: excl = items, never chosen
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        for (int t0 = 0; t0 < items; t0 += 64) {
            const int t = t0 + lane;
            int j = -1;
            if (t < items) {
                int lo = 0;
#pragma unroll
                for (int step = 32; step > 0; step >>= 1) {
                    const int c = lo + step;
                    if (c < 64 && s_col[wave][c].x <= t) lo = c;
                }
                const int2 e = s_col[wave][lo];
                j = e.y + (t - e.x);
            }
            bool ok = false;
            unsigned meta = 0;
            if (j >= 0) {
                const float2 it = cell_xy[j];
                meta = cell_meta[j];
                ok = fabsf(it.x - x) < rs && fabsf(it.y - y) < rs;   // GetFeaturesInArea, Frame.cc:627-640
                if (ok && check_levels) {
                    const int oct = (int)((meta >> 16) & 0xF);
                    ok = !(oct < q.min_level) && !(q.max_level >= 0 && oct > q.max_level);
                }
            }
            const unsigned long long have = __builtin_amdgcn_ballot_w64(ok);
            if (PASS == 1 && ok) {
                const int i = min((int)(meta & 0xFFFF), n - 1);
                const uint4* dk = reinterpret_cast<const uint4*>(a.desc + (long long)i * 32);
                const uint4 b0 = dk[0], b1 = dk[1];
                const float ur = a.u_right ? a.u_right[i] : -1.0f;
                const bool gated = ur > 0 && fabsf(q.ur - ur) > rs;   // ORBmatcher.h:118-123
                const unsigned dist = (unsigned)(__popc(a0.x ^ b0.x) + __popc(a0.y ^ b0.y) + __popc(a0.z ^ b0.z) + __popc(a0.w ^ b0.w) +
                                                 __popc(a1.x ^ b1.x) + __popc(a1.y ^ b1.y) + __popc(a1.z ^ b1.z) + __popc(a1.w ^ b1.w));
                const unsigned entry = (meta & 0xFFFFFu) | (dist << 20) | (gated ? 0x80000000u : 0u);
                const unsigned order = (((meta >> 20) & 63u) << 22) | ((meta >> 26) << 16) | (meta & 0xFFFFu);   // grid column, grid row, index
                const unsigned long long key = ((unsigned long long)order << 32) | entry;
                const int pos = total + __popcll(have & ((1ull << lane) - 1));
                if (pos < T) {
                    if (T <= PJ_CAND_LDS) s_keys[wave][pos] = key;
                    else tmp[base + pos] = key;
                }
            }
            total += __popcll(have);
        }
    }
    if (PASS == 0) {
        if (lane == 0 && iq < a.m) start[iq] = total;
        return;
    }
    __syncthreads();
    if (!scan) return;
    if (T <= PJ_CAND_LDS) {
        for (int t0 = 0; t0 < T; t0 += 64) {                   // (wave-uniform trip count; keys are distinct: the keypoint index is in them)
            const int t = t0 + lane;
            const unsigned long long key = t < T ? s_keys[wave][t] : ~0ull;
            int rank = 0;
            for (int u = 0; u < T; u++) rank += s_keys[wave][u] < key ? 1 : 0;
            if (t < T) {
                cand[base + rank] = (unsigned)key;
                if (h_cand) h_cand[base + rank] = (unsigned)key;
            }
        }
    } else {
        __threadfence();   // the wave's own keys, written above through the vector memory path, read back by other lanes
        for (int t = lane; t < T; t += 64) {
            const unsigned long long key = __hip_atomic_load(&tmp[base + t], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            int rank = 0;
            for (int u = 0; u < T; u++) rank += __hip_atomic_load(&tmp[base + u], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < key ? 1 : 0;
            cand[base + rank] = (unsigned)key;
            if (h_cand) h_cand[base + rank] = (unsigned)key;
        }
    }
}

// counts [m] -> offsets [m + 1] in place (one workgroup; m <= a few ten thousand), mirrored into the caller-side pinned block
__global__ __launch_bounds__(1024) void k_proj_cand_scan(int* start, int m, int* h_start)
{
    __shared__ int s_part[16];
    __shared__ unsigned long long s_tot64;   // the table's size without wrap-around (a point has at most 65 535 candidates, a thread's chunk fits an int)
    const int tid = threadIdx.x;
    if (tid == 0) s_tot64 = 0;
    __syncthreads();
    const int chunk = (m + 1023) / 1024;
    const int b = min(m, tid * chunk), e = min(m, b + chunk);
    int sum = 0;
    for (int i = b; i < e; i++) sum += start[i];
    if (sum) atomicAdd(&s_tot64, (unsigned long long)sum);
    int tot = 0;
    const int incl = st_block_incl_scan(sum, s_part, &tot);
    int run = incl - sum;
    for (int i = b; i < e; i++) {
        const int cnt = start[i];
        start[i] = run;
        if (h_start) h_start[i] = run;
        run += cnt;
    }
    if (tid == 0) {
        if (s_tot64 > 0x7FFFFFFFull) tot = -1;   // more than 2^31 entries: refused by the host (no caller array holds them)
        start[m] = tot;
        if (h_start) h_start[m] = tot;
    }
}

// ------------------------------------------------------------------------------------------------------------
// host side
// ------------------------------------------------------------------------------------------------------------
#define PTRY(c, expr)                                                                             \
    do {                                                                                          \
        hipError_t e_ = (expr);                                                                   \
        if (e_ != hipSuccess) {                                                                   \
            (c)->err = std::string(#expr) + ": " + hipGetErrorString(e_);                         \
            return GFO_ERR_DEVICE;                                                                \
        }                                                                                         \
    } while (0)

static inline size_t al256(size_t v) { return (v + 255) / 256 * 256; }

static int pj_fail(gfo_ctx* c, int code, const char* msg)
{
    c->err = msg;
    return code;
}

// work buffers of the projection search, grown on demand and kept (never on a steady-state call)
static int pj_reserve(gfo_ctx* c, int frames, int m, int n_cap)
{
    GfoProjBuf& b = c->pj;
    if (frames <= b.frames_cap && m <= b.m_cap && n_cap <= b.n_cap) return GFO_OK;
    PTRY(c, hipStreamSynchronize(c->stream));
    if (b.base) (void)hipFree(b.base);
    b = GfoProjBuf{};
    c->have_projection = false;
    const size_t F = (size_t)(frames > 1 ? frames : 1), M = (size_t)(m > 1 ? m : 1), N = (size_t)(n_cap > 1 ? n_cap : 1);
    size_t off = 0;
    auto take = [&](size_t bytes) { size_t o = off; off = al256(off + bytes); return o; };
    const size_t o_cs = take(F * (NSG + 1) * 4), o_it = take(F * N * 8), o_me = take(F * N * 4), o_pk = take(F * M * 4),
                 o_pd = take(F * M * 4), o_lv = take(F * M * 4), o_cd = take(F * M * 32), o_rb = take(F * M * 4),
                 o_so = take(F * M * 4), o_sp = take(8 * (size_t)PJ_SPILL_PER_POINT * (F * M < 16384 ? F * M : 16384)),
                 o_tg = take(F * N * 8), o_ct = take(F * PJ_CNT * 4), o_om = take(F * N * 4), o_os = take(F * N * 4), o_oq = take(F * M * 4);
    PTRY(c, hipMalloc(&b.base, off));
    uint8_t* S = (uint8_t*)b.base;
    b.cell_start = (int*)(S + o_cs); b.cell_xy = S + o_it; b.cell_meta = (unsigned*)(S + o_me);
    b.pick = (int*)(S + o_pk); b.pick_dist = (int*)(S + o_pd); b.live = (unsigned*)(S + o_lv); b.cand = S + o_cd; b.rot_bin = (int*)(S + o_rb);
    b.spill_off = (int*)(S + o_so); b.spill = (unsigned long long*)(S + o_sp); b.spill_cap = PJ_SPILL_PER_POINT * (int)(F * M < 16384 ? F * M : 16384);
    b.tab_g = (int*)(S + o_tg); b.counters = (int*)(S + o_ct); b.out_mp = (int*)(S + o_om); b.out_score = (int*)(S + o_os);
    b.out_q = (int*)(S + o_oq);
    b.frames_cap = (int)F; b.m_cap = (int)M; b.n_cap = (int)N;
    return GFO_OK;
}

static void pj_bind(const gfo_ctx* c, ProjB* a)
{
#ifdef GFO_PROJ_DEBUG
    a->dbg_stop = getenv("GFO_PROJ_STOP") ? atoi(getenv("GFO_PROJ_STOP")) : 0;
#endif
    const GfoProjBuf& b = c->pj;
    a->n_cap = b.n_cap;
    a->cell_start = b.cell_start; a->cell_xy = (float2*)b.cell_xy; a->cell_meta = b.cell_meta;
    a->pick = b.pick; a->pick_dist = b.pick_dist; a->live = b.live; a->cand = (uint4*)b.cand; a->rot_bin = b.rot_bin;
    a->spill_off = b.spill_off; a->spill = b.spill; a->spill_cap = b.spill_cap;
    if (const char* e = getenv("GFO_PROJ_SPILL_CAP")) {   // tests: a pool that runs out (read per call)
        const int cap = atoi(e);
        if (cap >= 0 && cap < a->spill_cap) a->spill_cap = cap;
    }
    a->tab_g = b.tab_g; a->counters = b.counters; a->out_mp = b.out_mp; a->out_score = b.out_score;
}

// the three launches; n_max bounds the keypoints of any frame (sizes the LDS table)
static int pj_launch(gfo_ctx* c, const ProjB& a, int frames, int n_max)
{
    hipStream_t st = c->stream;
    gfo_prof_begin(c, ST_PROJECT);
    {
        ProjB g = a;
        g.grid_frames = frames;
        const int copy_blocks = a.cp_n16 > 0 ? (a.cp_n16 + 1023) / 1024 : 0;
        GFO_LAUNCH(c, k_proj_grid, dim3(frames + copy_blocks), dim3(1024), 0, st, g);
    }
    // round 0 against an LDS copy of the grid when it fits twice per CU: workgroups of 1024 threads, as many per
    // frame as it takes to put ~2 on every CU (each pays the 78 KB copy once, then walks its chunk of the points)
    const size_t grid_bytes = (size_t)a.n_cap * 12 + (NSG + 1) * 2 + 16;
    static const int lds_grid_on = getenv("GFO_PROJ_LDSGRID") ? atoi(getenv("GFO_PROJ_LDSGRID")) : 1;
    // few points (the host-array call Tracking makes per frame: a thousand or two against one frame): a WAVEFRONT per point
    const int wave_on = getenv("GFO_PROJ_WAVE") ? atoi(getenv("GFO_PROJ_WAVE")) : 1;   // read per call: the tests run both forms
    if ((wave_on || a.fuse_gate) && (long long)frames * a.m <= 16384) {   // (a fusion search always takes this form: gfo_search_for_fusion feeds it at most 16 384 points a call)
        GFO_LAUNCH(c, k_proj_round0_wave, dim3((a.m + PJ_WAVES - 1) / PJ_WAVES, frames), dim3(64 * PJ_WAVES), 0, st, a);
    } else if (lds_grid_on && grid_bytes <= 78 * 1024 && a.m >= 4096) {
        int per_frame = (512 + frames - 1) / frames;
        const int max_pf = (a.m + 2047) / 2048;     // at least two passes of 1024 points per workgroup
        if (per_frame > max_pf) per_frame = max_pf;
        if (per_frame < 1) per_frame = 1;
        const int chunk = ((a.m + per_frame - 1) / per_frame + 63) / 64 * 64;
        if (grid_bytes > 48 * 1024)
            (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&k_proj_round0<true>), hipFuncAttributeMaxDynamicSharedMemorySize, 80 * 1024);
        GFO_LAUNCH(c, k_proj_round0<true>, dim3((a.m + chunk - 1) / chunk, frames), dim3(1024), grid_bytes, st, a, chunk);
    } else {
        GFO_LAUNCH(c, k_proj_round0<false>, dim3((a.m + 255) / 256, frames), dim3(256), 0, st, a, 256);
    }
    const size_t tab_bytes = (size_t)n_max * 8;   // claim / owner table + score table
    if (tab_bytes <= 96 * 1024) {
        if (tab_bytes > 48 * 1024)   // beyond the default dynamic-LDS grant: raised per call (per device), rare
            (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&k_proj_resolve<false>), hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024);
        GFO_LAUNCH(c, k_proj_resolve<false>, dim3(frames), dim3(1024), tab_bytes, st, a);
    } else {
        GFO_LAUNCH(c, k_proj_resolve<true>, dim3(frames), dim3(1024), 0, st, a);
    }
    gfo_prof_end(c);
    if (int lrc = gfo_take_launch_err(c)) return lrc;
    PTRY(c, hipGetLastError());
    return GFO_OK;
}

// Stages the inputs of a host-array call: the arrays are packed into the pinned mirror of the scratch layout and cross in one copy
// (or inside the grid launch, ProjB::cp_*), the work buffers are reserved, and `a` describes frame, queries and scratch -- everything
// but the mode of the search.
static int pj_stage(gfo_ctx* c, GfoXfer& x, const gfo_keypoint* kp_un, const uint8_t* desc, const float* u_right, const float* kp_angle, int n,
                    const gfo_frame_bounds* fb, const gfo_proj_query* queries, const uint8_t* q_desc, int m, const uint8_t* kp_taken, ProjB& a,
                    const float* fuse_inv_sigma2 = nullptr, int fuse_nlevels = 0)
{
    PTRY(c, hipSetDevice(c->device));
    size_t off = 0;
    auto take = [&](size_t bytes) { size_t o = off; off = al256(off + bytes); return o; };
    const size_t o_kp = take(sizeof(gfo_keypoint) * n), o_desc = take(32 * (size_t)n), o_ur = take(4 * (size_t)n),
                 o_tk = take(n), o_ang = take(4 * (size_t)n), o_q = take(sizeof(gfo_proj_query) * m),
                 o_mpd = take(32 * (size_t)m), o_sig = take(4 * GFO_MAX_LEVELS);
    if (off > c->scratch_bytes) {
        (void)hipStreamSynchronize(c->stream);
        if (c->d_scratch) (void)hipFree(c->d_scratch);
        c->d_scratch = nullptr;
        c->scratch_bytes = 0;
        PTRY(c, hipMalloc(&c->d_scratch, off + off / 2));   // headroom: the next, slightly larger frame does not reallocate
        c->scratch_bytes = off + off / 2;
    }
    if (int rc = pj_reserve(c, 1, m + m / 2, n + n / 2)) return rc;
    uint8_t* S = (uint8_t*)c->d_scratch;
    hipStream_t st = c->stream;
    // the seven input arrays are packed into the pinned mirror of the scratch layout and cross in one copy: seven pageable
    // hipMemcpyAsync calls cost ~95 us of the call's 190 us (profiles/proj_call_latency_r05.txt)
    if (int rc = x.in(off)) return rc;
    x.put(o_kp, kp_un, sizeof(gfo_keypoint) * n);
    x.put(o_desc, desc, 32 * (size_t)n);
    if (u_right) x.put(o_ur, u_right, 4 * (size_t)n);
    if (kp_taken) x.put(o_tk, kp_taken, n);
    if (kp_angle) x.put(o_ang, kp_angle, 4 * (size_t)n);
    x.put(o_q, queries, sizeof(gfo_proj_query) * m);
    x.put(o_mpd, q_desc, 32 * (size_t)m);
    {
        float sig[GFO_MAX_LEVELS] = {0};
        for (int l = 0; l < GFO_MAX_LEVELS && l < fuse_nlevels && fuse_inv_sigma2; l++) sig[l] = fuse_inv_sigma2[l];
        x.put(o_sig, sig, sizeof sig);
    }
    // up to 1 MB the inputs travel inside the grid launch (ProjB::cp_*); larger calls take the copy engine first
    static const long fused_max = getenv("GFO_MATCHER_UPLOAD_KERNEL_MAX") ? atol(getenv("GFO_MATCHER_UPLOAD_KERNEL_MAX")) : (1L << 20);
    static const bool fuse_ok = !(getenv("GFO_PROJ_FUSED_UPLOAD") && atoi(getenv("GFO_PROJ_FUSED_UPLOAD")) == 0);
    const bool fused = fuse_ok && (long)off <= fused_max;
    if (!fused) PTRY(c, x.up(S, off, st));
    a = ProjB{};
    if (fused) {
        a.cp_src = (const uint4*)x.H; a.cp_dst = (uint4*)S; a.cp_n16 = (int)(off / 16);
        a.kp_grid = (const gfo_keypoint*)(x.H + o_kp);
    }
    a.kp = (const gfo_keypoint*)(S + o_kp);
    a.desc = S + o_desc;
    a.u_right = u_right ? (const float*)(S + o_ur) : nullptr;
    a.taken0 = kp_taken ? S + o_tk : nullptr;
    a.kp_angle = kp_angle ? (const float*)(S + o_ang) : nullptr;
    a.n_dev = nullptr; a.n_host = n;
    a.fb = *fb;
    a.inv_w = (float)GRID_COLS / (fb->max_x - fb->min_x);  // Frame.cc:129-130
    a.inv_h = (float)GRID_ROWS / (fb->max_y - fb->min_y);
    a.sinv_w = (float)SG_COLS / (fb->max_x - fb->min_x);
    a.sinv_h = (float)SG_ROWS / (fb->max_y - fb->min_y);
    a.form = 0;
    a.q = S + o_q;
    a.q_desc = S + o_mpd;
    a.m = m;
    a.fuse_gate = fuse_inv_sigma2 ? 1 : 0;
    a.inv_sigma2 = (const float*)(S + o_sig);
    pj_bind(c, &a);
    return GFO_OK;
}

// out_point (optional, [m]): the per-point outcomes of gfo_search_by_projection_points
static int pj_queries(gfo_ctx* c, const gfo_keypoint* kp_un, const uint8_t* desc, const float* u_right, const float* kp_angle, int n,
                      const gfo_frame_bounds* fb, const gfo_proj_query* queries, const uint8_t* q_desc, int m, const gfo_proj_mode* mode,
                      const uint8_t* kp_taken, int32_t* out_mp, int32_t* out_score, int* nmatches, int32_t* out_point,
                      const float* fuse_inv_sigma2 = nullptr, int fuse_nlevels = 0)
{
    if (!c) return GFO_ERR_INVALID;
    if (!fb || !mode || !out_mp || !out_score || !nmatches || n < 0 || m < 0 || (n > 0 && (!kp_un || !desc)) ||
        (m > 0 && (!queries || !q_desc)) || (mode->check_orientation && n > 0 && !kp_angle))
        return pj_fail(c, GFO_ERR_INVALID, "gfo_search_by_projection: bad argument");
    if (n > 65535) return pj_fail(c, GFO_ERR_INVALID, "gfo_search_by_projection: more than 65535 keypoints");
    // "no candidate" is bestDist = 256 in the reference (:187, :1517): a threshold of 256 or more would accept it
    if (mode->th_dist < 0 || mode->th_dist > 255) return pj_fail(c, GFO_ERR_INVALID, "gfo_search_by_projection: th_dist must be 0..255");
    if (!(fb->max_x > fb->min_x) || !(fb->max_y > fb->min_y)) return pj_fail(c, GFO_ERR_INVALID, "gfo_search_by_projection: empty frame bounds");
    for (int i = 0; i < n; i++)
        if (kp_un[i].octave < 0 || kp_un[i].octave >= GFO_MAX_LEVELS) return pj_fail(c, GFO_ERR_INVALID, "gfo_search_by_projection: keypoint octave outside 0..15");
    if (mode->check_orientation) {
        // the rotation histogram is indexed with round((angle_q - angle_kp [+ 360]) / 30) (ORBmatcher.cc:1557-1565, which asserts the
        // bin): angles are cv::KeyPoint::angle of oriented keypoints, 0..360 -- anything else would index beside the 30 bins
        for (int i = 0; i < n; i++)
            if (!(kp_angle[i] >= 0.f && kp_angle[i] <= 360.f)) return pj_fail(c, GFO_ERR_INVALID, "gfo_search_by_projection: keypoint angle outside 0..360");
        for (int i = 0; i < m; i++)
            if (!(queries[i].angle >= 0.f && queries[i].angle <= 360.f)) return pj_fail(c, GFO_ERR_INVALID, "gfo_search_by_projection: query angle outside 0..360");
    }
    *nmatches = 0;
    for (int i = 0; i < n; i++) { out_mp[i] = -1; out_score[i] = 0; }
    if (out_point) for (int i = 0; i < m; i++) out_point[i] = PJ_PT_NONE;
    if (n == 0 || m == 0) return GFO_OK;
    GfoXfer x(c);
    ProjB a{};
    if (int rc = pj_stage(c, x, kp_un, desc, u_right, kp_angle, n, fb, queries, q_desc, m, kp_taken, a, fuse_inv_sigma2, fuse_nlevels)) return rc;
    hipStream_t st = c->stream;
    a.use_ratio = mode->use_ratio;
    a.nn_ratio = mode->nn_ratio;
    a.th_dist = mode->th_dist;
    a.check_ori = mode->check_orientation;
    a.max_matches = mode->max_matches > 0 ? mode->max_matches : 0;
    // the resolve kernel writes counters, out_mp and out_score into the pinned block itself (ProjB::h_*): no copy back
    const size_t o_hm = 256, o_hs = o_hm + al256(4 * (size_t)n), o_hq = o_hs + al256(4 * (size_t)n);
    if (int rc = x.out(o_hq + (out_point ? 4 * (size_t)m : 0))) return rc;
    const bool direct = gfo_matcher_host_writes();
    if (out_point) a.out_q = c->pj.out_q;
    if (direct) {
        a.h_counters = (int*)x.HO;
        a.h_out_mp = (int*)(x.HO + o_hm);
        a.h_out_score = (int*)(x.HO + o_hs);
        if (out_point) a.h_out_q = (int*)(x.HO + o_hq);
    }
    if (int rc = pj_launch(c, a, 1, n)) return rc;
    if (!direct) {   // three (four) copies into the same places
        PTRY(c, hipMemcpyAsync(x.HO, a.counters, PJ_CNT * 4, hipMemcpyDeviceToHost, st));
        PTRY(c, hipMemcpyAsync(x.HO + o_hm, a.out_mp, 4 * (size_t)n, hipMemcpyDeviceToHost, st));
        PTRY(c, hipMemcpyAsync(x.HO + o_hs, a.out_score, 4 * (size_t)n, hipMemcpyDeviceToHost, st));
        if (out_point) PTRY(c, hipMemcpyAsync(x.HO + o_hq, a.out_q, 4 * (size_t)m, hipMemcpyDeviceToHost, st));
    }
    PTRY(c, hipStreamSynchronize(st));
    int cnt[PJ_CNT];
    memcpy(cnt, x.HO, sizeof cnt);
    memcpy(out_mp, x.HO + o_hm, 4 * (size_t)n);
    memcpy(out_score, x.HO + o_hs, 4 * (size_t)n);
    if (out_point) memcpy(out_point, x.HO + o_hq, 4 * (size_t)m);
    if (cnt[PJ_ERR]) return pj_fail(c, GFO_ERR_STATE, "gfo_search_by_projection: fixed point not reached");
    *nmatches = cnt[PJ_NMATCH];
    c->last_project_rounds = cnt[PJ_ROUNDS];
    static const bool stats = getenv("GFO_PROJ_STATS") != nullptr;   // diagnosis only: one line per call on stderr
    if (stats) fprintf(stderr, "[gfo] projection: m %d n %d live %d rounds %d fallbacks %d matches %d\n", m, n, cnt[PJ_NLIVE], cnt[PJ_ROUNDS], cnt[PJ_FALLBACK], cnt[PJ_NMATCH]);
    c->have_projection = false;
    return GFO_OK;
}

extern "C" int gfo_search_by_projection_queries(gfo_ctx* c, const gfo_keypoint* kp_un, const uint8_t* desc,
                                                const float* u_right, const float* kp_angle, int n,
                                                const gfo_frame_bounds* fb, const gfo_proj_query* queries,
                                                const uint8_t* q_desc, int m, const gfo_proj_mode* mode,
                                                const uint8_t* kp_taken, int32_t* out_mp, int32_t* out_score, int* nmatches)
{
    return pj_queries(c, kp_un, desc, u_right, kp_angle, n, fb, queries, q_desc, m, mode, kp_taken, out_mp, out_score, nmatches, nullptr);
}

// The query form with what every query did at its turn (before any rotation check).  Queries that block nothing (flags bit 2 clear)
// against a frame whose slots only count as they were on entry are INDEPENDENT best-match searches: the form of ORBmatcher::Fuse(KF, Scw, ..)
// (ORBmatcher.cc:1089-1212: best keypoint of the predicted levels in a window, TH_LOW, no ratio) and of either direction of SearchBySim3
// (:1214-1438: TH_HIGH), whose side effects stay with the caller.
extern "C" int gfo_search_by_projection_queries_points(gfo_ctx* c, const gfo_keypoint* kp_un, const uint8_t* desc,
                                                       const float* u_right, const float* kp_angle, int n,
                                                       const gfo_frame_bounds* fb, const gfo_proj_query* queries,
                                                       const uint8_t* q_desc, int m, const gfo_proj_mode* mode,
                                                       const uint8_t* kp_taken, int32_t* out_q, int32_t* out_score, int32_t* out_point,
                                                       int* nmatches)
{
    if (!c) return GFO_ERR_INVALID;
    if (!out_point && m > 0) return pj_fail(c, GFO_ERR_INVALID, "gfo_search_by_projection_queries_points: null out_point");
    if (mode && mode->max_matches > 0) return pj_fail(c, GFO_ERR_INVALID, "gfo_search_by_projection_queries_points: no feature budget in this form");
    return pj_queries(c, kp_un, desc, u_right, kp_angle, n, fb, queries, q_desc, m, mode, kp_taken, out_q, out_score, nmatches, out_point);
}

// ORBmatcher::Fuse(KeyFrame*, const vector<MapPoint*>&, th) (ORBmatcher.cc:937-1087), its search: per point the best keypoint of the two
// predicted levels in the window among those whose reprojection error passes the chi-square test with the keypoint's own level sigma
// (:1019-1051), TH_LOW, nothing blocks.  What is done with a find (Replace / AddObservation, :1067-1083) stays with the caller.
extern "C" int gfo_search_for_fusion(gfo_ctx* c, const gfo_keypoint* kp_un, const uint8_t* desc, const float* u_right, int n,
                                     const gfo_frame_bounds* fb, const float* inv_level_sigma2, int nlevels, const gfo_proj_query* queries,
                                     const uint8_t* q_desc, int m, int th_dist, int32_t* out_point)
{
    if (!c) return GFO_ERR_INVALID;
    if (!inv_level_sigma2 || nlevels < 1 || nlevels > GFO_MAX_LEVELS || (m > 0 && (!out_point || !queries || !q_desc)) || n < 0 || m < 0)
        return pj_fail(c, GFO_ERR_INVALID, "gfo_search_for_fusion: bad argument");
    for (int i = 0; i < n && kp_un; i++)
        if (kp_un[i].octave < 0 || kp_un[i].octave >= nlevels) return pj_fail(c, GFO_ERR_INVALID, "gfo_search_for_fusion: keypoint octave outside the sigma table");
    std::vector<gfo_proj_query> q(queries && m > 0 ? queries : nullptr, queries && m > 0 ? queries + m : nullptr);
    for (auto& e : q) e.flags &= ~4;                      // nothing a point finds hides a keypoint from the next one
    std::vector<int32_t> out_q((size_t)(n > 0 ? n : 1)), out_score((size_t)(n > 0 ? n : 1));
    gfo_proj_mode mode = {0, 0.f, th_dist, 0, 0};
    int nm = 0;
    if (m == 0) return pj_queries(c, kp_un, desc, u_right, nullptr, n, fb, nullptr, q_desc, 0, &mode, nullptr, out_q.data(), out_score.data(), &nm, out_point,
                                  inv_level_sigma2, nlevels);
    // the points are independent of each other, so a long list goes in pieces the wavefront-per-point form takes (the one that carries the gate)
    for (int m0 = 0; m0 < m; m0 += 16384) {
        const int mm = m - m0 < 16384 ? m - m0 : 16384;
        const int rc = pj_queries(c, kp_un, desc, u_right, nullptr, n, fb, q.data() + m0, q_desc + (size_t)m0 * 32, mm, &mode, nullptr, out_q.data(),
                                  out_score.data(), &nm, out_point + m0, inv_level_sigma2, nlevels);
        if (rc != GFO_OK) return rc;
    }
    return GFO_OK;
}

// ORBmatcher::SearchByProjection(Frame&, const vector<MapPoint*>&, th): every map point becomes a query with
// r = RadiusByViewingCos(viewCos) (* th), window r * scale[level], levels [level-1, level] (ORBmatcher.cc:171-180).
static void pj_queries_of_map_points(const gfo_map_point* mps, int m, const float* sf, int nlevels, float th, gfo_proj_query* q)
{
    const bool bFactor = th != 1.0f;
    for (int i = 0; i < m; i++) {
        const gfo_map_point& p = mps[i];
        gfo_proj_query& d = q[i];
        const int lvl = p.level;
        float r = (double)p.view_cos > 0.998 ? 2.5f : 4.0f;  // RadiusByViewingCos, :243-249
        if (bFactor) r *= th;
        d.u = p.proj_x;
        d.v = p.proj_y;
        d.ur = p.proj_xr;
        d.radius = (lvl >= 0 && lvl < nlevels) ? r * sf[lvl] : 0.f;
        d.min_level = lvl - 1;
        d.max_level = lvl;
        d.angle = 0.f;
        d.flags = (((p.flags & 1) && !(p.flags & 2) && lvl >= 0 && lvl < nlevels) ? 1 : 0) | (p.flags & 4);
    }
}

static int pj_map_points(gfo_ctx* c, const gfo_keypoint* kp_un, const uint8_t* desc, const float* u_right,
                         int n, const float* sf, int nlevels, const gfo_frame_bounds* fb,
                         const gfo_map_point* mps, const uint8_t* mp_desc, int m, float th, float nn_ratio,
                         const uint8_t* kp_taken, int32_t* out_mp, int32_t* out_score, int* nmatches, int32_t* out_point)
{
    if (!c) return GFO_ERR_INVALID;
    if (!sf || nlevels < 1 || nlevels > GFO_MAX_LEVELS || m < 0 || (m > 0 && !mps))
        return pj_fail(c, GFO_ERR_INVALID, "gfo_search_by_projection: bad argument");
    std::vector<gfo_proj_query> q((size_t)(m > 0 ? m : 1));
    pj_queries_of_map_points(mps, m, sf, nlevels, th, q.data());
    gfo_proj_mode mode = {1, nn_ratio, TH_HIGH, 0, 0};
    return pj_queries(c, kp_un, desc, u_right, nullptr, n, fb, q.data(), mp_desc, m, &mode, kp_taken, out_mp, out_score, nmatches, out_point);
}

extern "C" int gfo_search_by_projection(gfo_ctx* c, const gfo_keypoint* kp_un, const uint8_t* desc, const float* u_right,
                                        int n, const float* sf, int nlevels, const gfo_frame_bounds* fb,
                                        const gfo_map_point* mps, const uint8_t* mp_desc, int m, float th, float nn_ratio,
                                        const uint8_t* kp_taken, int32_t* out_mp, int32_t* out_score, int* nmatches)
{
    return pj_map_points(c, kp_un, desc, u_right, n, sf, nlevels, fb, mps, mp_desc, m, th, nn_ratio, kp_taken, out_mp, out_score, nmatches, nullptr);
}

// ORBmatcher::SearchByProjection_Budget (src/ORBmatcher.cc:45-153) and SearchByProjection_OnePoint taken in vector order
// (include/ORBmatcher.h:71-150): the same loop body as the overload above, plus what every point did at its turn.
extern "C" int gfo_search_by_projection_points(gfo_ctx* c, const gfo_keypoint* kp_un, const uint8_t* desc, const float* u_right,
                                               int n, const float* sf, int nlevels, const gfo_frame_bounds* fb,
                                               const gfo_map_point* mps, const uint8_t* mp_desc, int m, float th, float nn_ratio,
                                               const uint8_t* kp_taken, int32_t* out_mp, int32_t* out_score, int32_t* out_point,
                                               int* nmatches)
{
    if (!c) return GFO_ERR_INVALID;
    if (!out_point && m > 0) return pj_fail(c, GFO_ERR_INVALID, "gfo_search_by_projection_points: null out_point");
    return pj_map_points(c, kp_un, desc, u_right, n, sf, nlevels, fb, mps, mp_desc, m, th, nn_ratio, kp_taken, out_mp, out_score, nmatches, out_point);
}

// The state of a frame after the FIRST `prefix` points of a gfo_search_by_projection_points call: a point's outcome depends on the
// points in front of it only, so every way the reference cuts its loop short -- SearchByProjection_Budget's clock (:96-102), the
// weighted match budget of Observability::runBaselineMapMatching (src/Observability.cc:1233-1262) -- is a prefix of the full answer.
// Pure bookkeeping on the host (no distance is computed here): slot i holds the LAST point of the prefix that matched it (:89 overwrites).
extern "C" int gfo_projection_points_prefix(const int32_t* out_point, int m, int prefix, int n, int32_t* out_mp, int32_t* out_score,
                                            int* nmatches)
{
    if (m < 0 || n < 0 || prefix < 0 || (m > 0 && !out_point) || (n > 0 && (!out_mp || !out_score))) return GFO_ERR_INVALID;
    if (prefix > m) prefix = m;
    for (int i = 0; i < n; i++) { out_mp[i] = -1; out_score[i] = 0; }
    int cnt = 0;
    for (int p = 0; p < prefix; p++) {
        const int v = out_point[p];
        if (v < 0) continue;
        const int k = v & 0xFFFF;
        if (k >= n) return GFO_ERR_INVALID;
        out_mp[k] = p;
        out_score[k] = v >> 16;
        cnt++;
    }
    if (nmatches) *nmatches = cnt;
    return GFO_OK;
}

// ORBmatcher::GetCandidates for all of vpMapPoints (include/ORBmatcher.h:152-172) + the static half of MatchCandidates (:176-250)
// the candidate table of m window queries (GetFeaturesInArea's order within a query, each entry with its descriptor distance): the body of
// gfo_projection_candidates, on queries
static int pj_candidates(gfo_ctx* c, const gfo_keypoint* kp_un, const uint8_t* desc, const float* u_right, int n, const gfo_frame_bounds* fb,
                         const gfo_proj_query* q, const uint8_t* q_desc, int m, int32_t* cand_start, uint32_t* cand, int cap, int* total)
{
    if (n > 65535) return pj_fail(c, GFO_ERR_INVALID, "gfo_projection_candidates: more than 65535 keypoints");
    if (!(fb->max_x > fb->min_x) || !(fb->max_y > fb->min_y)) return pj_fail(c, GFO_ERR_INVALID, "gfo_projection_candidates: empty frame bounds");
    for (int i = 0; i < n; i++)
        if (kp_un[i].octave < 0 || kp_un[i].octave >= GFO_MAX_LEVELS) return pj_fail(c, GFO_ERR_INVALID, "gfo_projection_candidates: keypoint octave outside 0..15");
    *total = 0;
    for (int i = 0; i <= m; i++) cand_start[i] = 0;
    if (n == 0 || m == 0) return GFO_OK;
    GfoXfer x(c);
    ProjB a{};
    if (int rc = pj_stage(c, x, kp_un, desc, u_right, nullptr, n, fb, q, q_desc, m, nullptr, a)) return rc;
    hipStream_t st = c->stream;
    // offsets [m + 1] | entries [cap] | unsorted keys [cap] (only the points with more than 64 candidates use them)
    const size_t o_st = 0, o_cd = al256(4 * ((size_t)m + 1)), o_tmp = o_cd + al256(4 * (size_t)cap), need = o_tmp + 8 * (size_t)cap + 256;
    if (need > c->pj_cand_bytes) {
        PTRY(c, hipStreamSynchronize(st));
        if (c->d_pj_cand) (void)hipFree(c->d_pj_cand);
        c->d_pj_cand = nullptr;
        c->pj_cand_bytes = 0;
        PTRY(c, hipMalloc(&c->d_pj_cand, need + need / 2));
        c->pj_cand_bytes = need + need / 2;
    }
    uint8_t* D = (uint8_t*)c->d_pj_cand;
    int* d_start = (int*)(D + o_st);
    unsigned* d_cand = (unsigned*)(D + o_cd);
    unsigned long long* d_tmp = (unsigned long long*)(D + o_tmp);
    const size_t o_hc = al256(4 * ((size_t)m + 1));
    if (int rc = x.out(o_hc + 4 * (size_t)cap)) return rc;
    const bool direct = gfo_matcher_host_writes();
    int* h_start = direct ? (int*)x.HO : nullptr;
    unsigned* h_cand = direct ? (unsigned*)(x.HO + o_hc) : nullptr;
    gfo_prof_begin(c, ST_PROJECT);
    {
        ProjB g = a;
        g.grid_frames = 1;
        const int copy_blocks = a.cp_n16 > 0 ? (a.cp_n16 + 1023) / 1024 : 0;
        GFO_LAUNCH(c, k_proj_grid, dim3(1 + copy_blocks), dim3(1024), 0, st, g);
    }
    const int blocks = (m + PJ_WAVES - 1) / PJ_WAVES;
    GFO_LAUNCH(c, k_proj_candidates<0>, dim3(blocks), dim3(64 * PJ_WAVES), 0, st, a, d_start, d_cand, h_cand, d_tmp, cap);
    GFO_LAUNCH(c, k_proj_cand_scan, dim3(1), dim3(1024), 0, st, d_start, m, h_start);
    if (cap > 0) GFO_LAUNCH(c, k_proj_candidates<1>, dim3(blocks), dim3(64 * PJ_WAVES), 0, st, a, d_start, d_cand, h_cand, d_tmp, cap);
    gfo_prof_end(c);
    if (int lrc = gfo_take_launch_err(c)) return lrc;
    PTRY(c, hipGetLastError());
    if (!direct) PTRY(c, hipMemcpyAsync(x.HO, d_start, 4 * ((size_t)m + 1), hipMemcpyDeviceToHost, st));
    PTRY(c, hipStreamSynchronize(st));
    memcpy(cand_start, x.HO, 4 * ((size_t)m + 1));
    const int tot = cand_start[m];
    *total = tot;
    c->have_projection = false;
    if (tot < 0) return pj_fail(c, GFO_ERR_CAPACITY, "gfo_projection_candidates: the table has more than 2^31 entries");
    if (tot > cap) {
        c->err = "gfo_projection_candidates: the table has " + std::to_string(tot) + " entries, the caller's array " + std::to_string(cap);
        return GFO_ERR_CAPACITY;
    }
    if (tot > 0) {
        if (!direct) {
            PTRY(c, hipMemcpyAsync(x.HO + o_hc, d_cand, 4 * (size_t)tot, hipMemcpyDeviceToHost, st));
            PTRY(c, hipStreamSynchronize(st));
        }
        memcpy(cand, x.HO + o_hc, 4 * (size_t)tot);
    }
    return GFO_OK;
}

extern "C" int gfo_projection_candidates(gfo_ctx* c, const gfo_keypoint* kp_un, const uint8_t* desc, const float* u_right, int n,
                                         const float* sf, int nlevels, const gfo_frame_bounds* fb, const gfo_map_point* mps,
                                         const uint8_t* mp_desc, int m, float th, int32_t* cand_start, uint32_t* cand, int cap, int* total)
{
    if (!c) return GFO_ERR_INVALID;
    if (!sf || nlevels < 1 || nlevels > GFO_MAX_LEVELS || !fb || !cand_start || !total || n < 0 || m < 0 || cap < 0 || (cap > 0 && !cand) ||
        (n > 0 && (!kp_un || !desc)) || (m > 0 && (!mps || !mp_desc)))
        return pj_fail(c, GFO_ERR_INVALID, "gfo_projection_candidates: bad argument");
    std::vector<gfo_proj_query> q((size_t)(m > 0 ? m : 1));
    pj_queries_of_map_points(mps, m, sf, nlevels, th, q.data());
    return pj_candidates(c, kp_un, desc, u_right, n, fb, q.data(), mp_desc, m, cand_start, cand, cap, total);
}

// ORBmatcher::SearchForInitialization (ORBmatcher.cc:520-633), the monocular bootstrap's matcher.  Two halves: every level-0 keypoint of
// F1 looks at F2's level-0 keypoints within windowSize of where it was matched last (GetFeaturesInArea, :538) and takes their descriptor
// distances -- the device's candidate table, one call --; then the reference's loop over that table on the host, in keypoint order, because
// a later keypoint may TAKE a keypoint of F2 from an earlier one when it is strictly closer (vMatchedDistance / vnMatches21, :557-581): a
// chain of thefts has no bound, and the whole pass is a few ten thousand table entries.
extern "C" int gfo_search_for_initialization(gfo_ctx* c, const gfo_keypoint* kp1, const uint8_t* desc1, int n1, float* prev_matched,
                                             const gfo_keypoint* kp2, const uint8_t* desc2, int n2, const gfo_frame_bounds* fb, int window_size,
                                             float nn_ratio, int check_orientation, int32_t* matches12, int* nmatches)
{
    if (!c) return GFO_ERR_INVALID;
    if (!fb || !nmatches || n1 < 0 || n2 < 0 || window_size < 0 || (n1 > 0 && (!kp1 || !desc1 || !prev_matched || !matches12)) ||
        (n2 > 0 && (!kp2 || !desc2)))
        return pj_fail(c, GFO_ERR_INVALID, "gfo_search_for_initialization: bad argument");
    // everything the table pass would refuse, before the caller's arrays are touched
    if (n2 > 65535) return pj_fail(c, GFO_ERR_INVALID, "gfo_search_for_initialization: more than 65535 keypoints in F2");
    if (!(fb->max_x > fb->min_x) || !(fb->max_y > fb->min_y)) return pj_fail(c, GFO_ERR_INVALID, "gfo_search_for_initialization: empty frame bounds");
    for (int i = 0; i < n2; i++)
        if (kp2[i].octave < 0 || kp2[i].octave >= GFO_MAX_LEVELS) return pj_fail(c, GFO_ERR_INVALID, "gfo_search_for_initialization: keypoint octave outside 0..15");
    *nmatches = 0;
    for (int i = 0; i < n1; i++) matches12[i] = -1;                                  // :523
    if (n1 == 0 || n2 == 0) return GFO_OK;
    std::vector<gfo_proj_query> q((size_t)n1);
    for (int i = 0; i < n1; i++) {
        gfo_proj_query& d = q[i];
        d.u = prev_matched[2 * i];
        d.v = prev_matched[2 * i + 1];
        d.ur = -1.f;
        d.radius = (float)window_size;                                              // `const float& r` of GetFeaturesInArea
        d.min_level = 0;                                                             // level1, level1 (:538) with level1 == 0 (:534-536)
        d.max_level = 0;
        d.angle = 0.f;
        d.flags = kp1[i].octave > 0 ? 0 : 1;
    }
    std::vector<int32_t> start((size_t)n1 + 1);
    std::vector<uint32_t> cand;
    int cap = 64 * n1, total = 0, rc = GFO_OK;
    for (int attempt = 0; attempt < 2; attempt++) {                                  // a table larger than the guess: the first pass says how large
        cand.resize((size_t)(cap > 0 ? cap : 1));
        rc = pj_candidates(c, kp2, desc2, nullptr, n2, fb, q.data(), desc1, n1, start.data(), cand.data(), cap, &total);
        if (rc != GFO_ERR_CAPACITY || total < 0) break;
        cap = total;
    }
    if (rc != GFO_OK) return rc;
    const int HISTO = 30, TH_LOW_ = 50;
    const float factor = 1.0f / HISTO;
    std::vector<int> matched_dist((size_t)n2, INT_MAX), matches21((size_t)n2, -1), rot_bin((size_t)n1, -1);
    int histo[30] = {0};
    int nm = 0;
    for (int i1 = 0; i1 < n1; i1++) {
        if (kp1[i1].octave > 0) continue;
        int bestDist = INT_MAX, bestDist2 = INT_MAX, bestIdx2 = -1;
        for (int k = start[i1]; k < start[i1 + 1]; k++) {
            const int i2 = (int)(cand[k] & 0xFFFFu), dist = (int)((cand[k] >> 20) & 0x7FFu);
            if (matched_dist[i2] <= dist) continue;                                  // :557
            if (dist < bestDist) { bestDist2 = bestDist; bestDist = dist; bestIdx2 = i2; }
            else if (dist < bestDist2) bestDist2 = dist;
        }
        if (bestDist > TH_LOW_ || !((float)bestDist < (float)bestDist2 * nn_ratio)) continue;
        if (matches21[bestIdx2] >= 0) {                                              // :575-579: taken from the keypoint that had it
            matches12[matches21[bestIdx2]] = -1;
            nm--;
        }
        matches12[i1] = bestIdx2;
        matches21[bestIdx2] = i1;
        matched_dist[bestIdx2] = bestDist;
        nm++;
        if (check_orientation) {
            float rot = kp1[i1].angle - kp2[bestIdx2].angle;
            if (rot < 0.0f) rot += 360.0f;
            int bin = (int)roundf(rot * factor);
            if (bin == HISTO) bin = 0;
            if (bin >= 0 && bin < HISTO) { rot_bin[i1] = bin; histo[bin]++; }        // (the reference asserts the range)
        }
    }
    if (check_orientation) {                                                         // ComputeThreeMaxima, :1723-1764
        int max1 = 0, max2 = 0, max3 = 0, ind1 = -1, ind2 = -1, ind3 = -1;
        for (int i = 0; i < HISTO; i++) {
            const int sz = histo[i];
            if (sz > max1) { max3 = max2; max2 = max1; max1 = sz; ind3 = ind2; ind2 = ind1; ind1 = i; }
            else if (sz > max2) { max3 = max2; max2 = sz; ind3 = ind2; ind2 = i; }
            else if (sz > max3) { max3 = sz; ind3 = i; }
        }
        if ((float)max2 < 0.1f * (float)max1) { ind2 = -1; ind3 = -1; }
        else if ((float)max3 < 0.1f * (float)max1) ind3 = -1;
        for (int i1 = 0; i1 < n1; i1++) {
            const int b = rot_bin[i1];
            if (b < 0 || b == ind1 || b == ind2 || b == ind3) continue;
            if (matches12[i1] >= 0) { matches12[i1] = -1; nm--; }                    // (a keypoint robbed since is in the histogram still, :612)
        }
    }
    for (int i1 = 0; i1 < n1; i1++)                                                  // :626-629
        if (matches12[i1] >= 0) {
            prev_matched[2 * i1] = kp2[matches12[i1]].x;
            prev_matched[2 * i1 + 1] = kp2[matches12[i1]].y;
        }
    *nmatches = nm;
    return GFO_OK;
}

// ORBmatcher::MatchCandidates (include/ORBmatcher.h:176-250) = the candidate loop of SearchByProjection_OnePoint (:101-149) on one
// point's entries of the table: what is left once window, level, gate and distance are known -- skip the slots a point with
// observations holds NOW (:113-115), best and second best on strict `<` in the list's order, TH_HIGH, the ratio test between
// candidates of one level.  Host bookkeeping (a few compares per candidate; no descriptor is read): the caller interleaves it
// with its own selection loop (Observability::runActiveMapMatching) and updates slot_taken after each match.
extern "C" int gfo_match_candidates(const uint32_t* cand, int ncand, const uint8_t* slot_taken, int n, float nn_ratio, int* best_dist)
{
    int bestDist = 256, bestLevel = -1, bestDist2 = 256, bestLevel2 = -1, bestIdx = -1;
    if (best_dist) *best_dist = 256;
    if (ncand < 0 || (ncand > 0 && !cand) || n < 0) return GFO_ERR_INVALID - 16;   // (below every GFO_POINT_* code: not an answer)
    for (int k = 0; k < ncand; k++) {
        const uint32_t e = cand[k];
        const int idx = (int)(e & 0xFFFFu);
        if (idx >= n) continue;                       // an entry of another frame's table: never a keypoint of this one
        if (slot_taken && slot_taken[idx]) continue;
        if (e & 0x80000000u) continue;
        const int dist = (int)((e >> 20) & 0x1FFu), level = (int)((e >> 16) & 0xFu);
        if (dist < bestDist) {
            bestDist2 = bestDist; bestDist = dist;
            bestLevel2 = bestLevel; bestLevel = level;
            bestIdx = idx;
        } else if (dist < bestDist2) {
            bestLevel2 = level;
            bestDist2 = dist;
        }
    }
    if (best_dist) *best_dist = bestDist;
    if (bestDist <= TH_HIGH) {
        if (bestLevel == bestLevel2 && (float)bestDist > nn_ratio * (float)bestDist2) return GFO_POINT_RATIO;
        return bestIdx;
    }
    return ncand > 0 ? GFO_POINT_FAR : GFO_POINT_NONE;
}

// ---- the resident local map and the batched, device-chained search --------------------------------------------
extern "C" int gfo_map_upload(gfo_ctx* c, const uint8_t* mp_desc, int m)
{
    if (!c) return GFO_ERR_INVALID;
    if (m < 0 || (m > 0 && !mp_desc)) return pj_fail(c, GFO_ERR_INVALID, "gfo_map_upload: bad argument");
    PTRY(c, hipSetDevice(c->device));
    if (m > c->map_cap) {
        PTRY(c, hipStreamSynchronize(c->stream));
        if (c->d_map_desc) (void)hipFree(c->d_map_desc);
        c->d_map_desc = nullptr;
        c->map_cap = 0;
        PTRY(c, hipMalloc(&c->d_map_desc, 32 * (size_t)(m + m / 4 + 64)));
        c->map_cap = m + m / 4 + 64;
    }
    if (m > 0) PTRY(c, hipMemcpyAsync(c->d_map_desc, mp_desc, 32 * (size_t)m, hipMemcpyHostToDevice, c->stream));
    PTRY(c, hipStreamSynchronize(c->stream));   // the caller's buffer is free on return
    c->map_m = m;
    return GFO_OK;
}

extern "C" int gfo_search_by_projection_batch(gfo_ctx* c, const gfo_projection_batch* p)
{
    if (!c || !p) return GFO_ERR_INVALID;
    if (!c->have_batch) return pj_fail(c, GFO_ERR_STATE, "gfo_search_by_projection_batch: no batch has been extracted");
    if (c->map_m <= 0 || !c->d_map_desc) return pj_fail(c, GFO_ERR_STATE, "gfo_search_by_projection_batch: no map uploaded (gfo_map_upload)");
    if (!p->mps) return pj_fail(c, GFO_ERR_INVALID, "gfo_search_by_projection_batch: null projections");
    if (p->stereo && !c->have_stereo) return pj_fail(c, GFO_ERR_STATE, "gfo_search_by_projection_batch: stereo chain requested but no stereo batch has been matched");
    if (!(p->bounds.max_x > p->bounds.min_x) || !(p->bounds.max_y > p->bounds.min_y))
        return pj_fail(c, GFO_ERR_INVALID, "gfo_search_by_projection_batch: empty frame bounds");
    PTRY(c, hipSetDevice(c->device));
    const int frames = p->stereo ? c->last_nimg / 2 : c->last_nimg;
    const int m = c->map_m, ks = c->g.kp_stride;
    if (int rc = pj_reserve(c, c->cap_batch > frames ? c->cap_batch : frames, m, ks)) return rc;
    hipStream_t st = c->stream;
    const gfo_map_point* d_mps = p->mps;
    const uint8_t* d_taken = p->kp_taken;
    if (!p->on_device) {   // host projections: staged through the scratch buffer
        const size_t b_mps = al256(sizeof(gfo_map_point) * (size_t)m * frames), b_tk = p->kp_taken ? al256((size_t)ks * frames) : 0;
        if (b_mps + b_tk > c->scratch_bytes) {
            PTRY(c, hipStreamSynchronize(st));
            if (c->d_scratch) (void)hipFree(c->d_scratch);
            c->d_scratch = nullptr;
            c->scratch_bytes = 0;
            PTRY(c, hipMalloc(&c->d_scratch, b_mps + b_tk));
            c->scratch_bytes = b_mps + b_tk;
        }
        uint8_t* S = (uint8_t*)c->d_scratch;
        PTRY(c, hipMemcpyAsync(S, p->mps, sizeof(gfo_map_point) * (size_t)m * frames, hipMemcpyHostToDevice, st));
        d_mps = (const gfo_map_point*)S;
        if (p->kp_taken) {
            PTRY(c, hipMemcpyAsync(S + b_mps, p->kp_taken, (size_t)ks * frames, hipMemcpyHostToDevice, st));
            d_taken = S + b_mps;
        }
    }
    ProjB a{};
    const int step = p->stereo ? 2 : 1;   // stereo: frame k = left image 2k
    a.kp = c->d_kp; a.kp_stride = (long long)step * ks;
    a.desc = c->d_desc;
    a.u_right = p->stereo ? c->st.u_right : nullptr; a.ur_stride = ks;
    a.taken0 = d_taken; a.tk_stride = ks;
    a.kp_angle = nullptr;
    a.n_dev = c->d_kp_cnt; a.n_dev_stride = step; a.n_host = 0;
    a.fb = p->bounds;
    a.inv_w = (float)GRID_COLS / (p->bounds.max_x - p->bounds.min_x);  // Frame.cc:129-130
    a.inv_h = (float)GRID_ROWS / (p->bounds.max_y - p->bounds.min_y);
    a.sinv_w = (float)SG_COLS / (p->bounds.max_x - p->bounds.min_x);
    a.sinv_h = (float)SG_ROWS / (p->bounds.max_y - p->bounds.min_y);
    a.form = 1;
    a.q = d_mps; a.q_stride = m;
    a.q_desc = c->d_map_desc; a.qd_stride = 0;
    a.m = m;
    a.th = p->th; a.bfactor = p->th != 1.0f;
    a.nlevels = c->g.nlevels;
    for (int l = 0; l < c->g.nlevels; l++) a.scale[l] = c->scale[l];
    a.use_ratio = 1; a.nn_ratio = p->nn_ratio; a.th_dist = TH_HIGH; a.check_ori = 0;
    pj_bind(c, &a);
    if (int rc = pj_launch(c, a, frames, ks)) return rc;
    c->have_projection = true;
    c->proj_frames = frames;
    c->proj_step = step;
    return GFO_OK;
}

extern "C" int gfo_projection_fetch(gfo_ctx* c, int frame, int32_t* out_mp, int32_t* out_score, int cap, int* nmatches)
{
    if (!c) return GFO_ERR_INVALID;
    if (!c->have_projection) return pj_fail(c, GFO_ERR_STATE, "gfo_projection_fetch: no batched projection search has run");
    PTRY(c, hipSetDevice(c->device));
    if (frame < 0 || frame >= c->proj_frames) return pj_fail(c, GFO_ERR_INVALID, "gfo_projection_fetch: frame out of range");
    hipStream_t st = c->stream;
    const int step = c->proj_step;
    int cnt[PJ_CNT], n = 0;
    PTRY(c, hipMemcpyAsync(cnt, c->pj.counters + frame * PJ_CNT, sizeof cnt, hipMemcpyDeviceToHost, st));
    PTRY(c, hipMemcpyAsync(&n, c->d_kp_cnt + frame * step, sizeof n, hipMemcpyDeviceToHost, st));
    PTRY(c, hipStreamSynchronize(st));
    if (cnt[PJ_ERR]) return pj_fail(c, GFO_ERR_STATE, "gfo_search_by_projection_batch: fixed point not reached");
    const int k = n < cap ? n : cap;
    if (k > 0 && out_mp) PTRY(c, hipMemcpyAsync(out_mp, c->pj.out_mp + (size_t)frame * c->pj.n_cap, 4 * (size_t)k, hipMemcpyDeviceToHost, st));
    if (k > 0 && out_score) PTRY(c, hipMemcpyAsync(out_score, c->pj.out_score + (size_t)frame * c->pj.n_cap, 4 * (size_t)k, hipMemcpyDeviceToHost, st));
    PTRY(c, hipStreamSynchronize(st));
    if (nmatches) *nmatches = cnt[PJ_NMATCH];
    c->last_project_rounds = cnt[PJ_ROUNDS];
    if (n > cap) {
        c->err = "gfo_projection_fetch: more keypoints than the caller capacity";
        return GFO_ERR_CAPACITY;
    }
    return GFO_OK;
}

extern "C" int gfo_projection_device_views(gfo_ctx* c, const int32_t** d_out_mp, const int32_t** d_out_score,
                                           const int32_t** d_counters, int* stride, int* counters_stride)
{
    if (!c) return GFO_ERR_INVALID;
    if (!c->have_projection) return pj_fail(c, GFO_ERR_STATE, "gfo_projection_device_views: no batched projection search has run");
    if (d_out_mp) *d_out_mp = c->pj.out_mp;
    if (d_out_score) *d_out_score = c->pj.out_score;
    if (d_counters) *d_counters = c->pj.counters;
    if (stride) *stride = c->pj.n_cap;
    if (counters_stride) *counters_stride = PJ_CNT;
    return GFO_OK;
}

// Every __global__ of this translation unit, for gfo_preload_kernels (gfo_api.hip): the runtime loads a code object and
// registers a kernel lazily, on the first launch that needs it; gfo_ctx_create resolves them all once per device under a
// mutex so that no two host threads ever race through that first-launch path (round 3: eight threads, first k_pack_results).
void gfo_kernels_project(std::vector<const void*>& v)
{
    v.push_back((const void*)k_proj_grid); v.push_back((const void*)k_proj_round0<true>); v.push_back((const void*)k_proj_round0<false>);
    v.push_back((const void*)k_proj_round0_wave);
    v.push_back((const void*)k_proj_resolve<true>); v.push_back((const void*)k_proj_resolve<false>);
    v.push_back((const void*)k_proj_candidates<0>); v.push_back((const void*)k_proj_candidates<1>); v.push_back((const void*)k_proj_cand_scan);
}
