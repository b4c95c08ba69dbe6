// k_quadtree.hip -- ORBextractor::DistributeOctTree (ORBextractor.cc:539-763) on the device.
//
// The reference walks a std::list of nodes, pushes children to the FRONT of the list and
// erases parents.  Two facts make an exact data-parallel restatement possible:
//   (1) every new node goes to the front and nodes never move, so the list is always sorted
//       by creation time, newest first.  "Position in the list" and "creation order" are the
//       same thing, and the tie-break of the final phase (equal-sized nodes: newest first --
//       the deterministic rule this build documents) is "smaller position";
//   (2) a child's box depends only on its parent's box, so every key can route itself.
// One breadth pass of the reference (:594-666) therefore becomes: histogram keys into the four
// children of their node, prefix-sum the non-empty child counts over the nodes in list order,
// and write the new list as [children of the last parent ... children of the first parent]
// followed by the untouched single-key nodes.  The final phase (:671-737) sorts the expandable
// nodes by (size desc, position asc), prefix-sums how many nodes each split adds, cuts where
// the total reaches N, and rebuilds the list the same way.
//
// One workgroup per (image, level); node tables and (up to 7 x quota of) the keys live in LDS, a level with more candidates
// keeps its keys in L2, a level whose tables exceed LDS runs from an HBM scratch block (k_quadtree_gmem).  Two statements of
// the algorithm: k_quadtree_body.inc (any workgroup size: batches) and k_quadtree_wide.inc (per-frame batches: 1024 threads,
// one thread per node, a third of the barriers), launched there together with the blur as k_quadtree_blur.
#include "gfo_internal.h"
#include <stdlib.h>

#define QT_MAX_THREADS 1024
#define QT_PART 32                     // ints of scan scratch: one per wave of the workgroup (<= 16), rounded up
#define QT_THREADS ((int)blockDim.x)   // 256 (small quotas) or 1024 (large ones): see gfo_launch_quadtree

struct QtBox {
    short ulx, uly, urx, bry;
};

// exclusive scan of vals[0..n) in place (LDS), returns the total; all threads call it.
// Per-thread chunk sums are scanned inside each wave with lane shuffles; only the four wave totals
// cross a workgroup barrier (3 barriers per scan).
__device__ __forceinline__ int qt_wave_incl_scan(int v)
{
    // inclusive scan over the 64 lanes in the DPP network (row shifts, then the row_bcast:15 / row_bcast:31
    // steps): six dependent VALU operations instead of six LDS round trips of a shuffle-based scan
    v += __builtin_amdgcn_update_dpp(0, v, 0x111, 0xF, 0xF, false);  // row_shr:1
    v += __builtin_amdgcn_update_dpp(0, v, 0x112, 0xF, 0xF, false);  // row_shr:2
    v += __builtin_amdgcn_update_dpp(0, v, 0x114, 0xF, 0xF, false);  // row_shr:4
    v += __builtin_amdgcn_update_dpp(0, v, 0x118, 0xF, 0xF, false);  // row_shr:8
    v += __builtin_amdgcn_update_dpp(0, v, 0x142, 0xA, 0xF, false);  // row_bcast:15 -> rows 1 and 3
    v += __builtin_amdgcn_update_dpp(0, v, 0x143, 0xC, 0xF, false);  // row_bcast:31 -> rows 2 and 3
    return v;
}

__device__ int qt_scan(int* vals, int n, int* part)
{
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int chunk = (n + QT_THREADS - 1) / QT_THREADS;
    const int b = tid * chunk, e = min(b + chunk, n);
    int s = 0;
    for (int i = b; i < e; i++) s += vals[i];
    const int incl = qt_wave_incl_scan(s);
    if (lane == 63) part[wave] = incl;
    __syncthreads();
    int wbase = 0, total = 0;
    for (int w = 0; w < QT_THREADS / 64; w++) {
        const int t = part[w];
        if (w < wave) wbase += t;
        total += t;
    }
    int run = wbase + incl - s;
    for (int i = b; i < e; i++) {
        const int v = vals[i];
        vals[i] = run;
        run += v;
    }
    __syncthreads();
    return total;
}

// Two exclusive scans over the same index range in one pass: the values ride in the two halves of a word
// (totals stay below 2^16: the caller checks 4 * ncap < 65536, otherwise it scans twice).
__device__ int2 qt_scan2(int* a, int* b2, int n, int* part)
{
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int chunk = (n + QT_THREADS - 1) / QT_THREADS;
    const int b = tid * chunk, e = min(b + chunk, n);
    unsigned s = 0;
    for (int i = b; i < e; i++) s += (unsigned)a[i] | ((unsigned)b2[i] << 16);
    const unsigned incl = (unsigned)qt_wave_incl_scan((int)s);
    if (lane == 63) part[wave] = (int)incl;
    __syncthreads();
    unsigned wbase = 0, total = 0;
    for (int w = 0; w < QT_THREADS / 64; w++) {
        const unsigned t = (unsigned)part[w];
        if (w < wave) wbase += t;
        total += t;
    }
    unsigned run = wbase + incl - s;
    for (int i = b; i < e; i++) {
        const unsigned v = (unsigned)a[i] | ((unsigned)b2[i] << 16);
        a[i] = (int)(run & 0xFFFFu);
        b2[i] = (int)(run >> 16);
        run += v;
    }
    __syncthreads();
    return make_int2((int)(total & 0xFFFFu), (int)(total >> 16));
}

__device__ __forceinline__ int qt_quadrant(uint32_t key, QtBox b)
{
    // ExtractorNode::DivideNode, ORBextractor.cc:483-484,515-525
    const int x = key & 0xFFF, y = (key >> 12) & 0xFFF;
    const int midx = b.ulx + (int)ceilf((float)(b.urx - b.ulx) / 2);
    const int midy = b.uly + (int)ceilf((float)(b.bry - b.uly) / 2);
    return (x < midx ? 0 : 1) + (y < midy ? 0 : 2);  // 0=n1 1=n2 2=n3 3=n4
}

__device__ __forceinline__ QtBox qt_child_box(QtBox b, int q)
{
    const short midx = (short)(b.ulx + (int)ceilf((float)(b.urx - b.ulx) / 2));
    const short midy = (short)(b.uly + (int)ceilf((float)(b.bry - b.uly) / 2));
    QtBox c;
    c.ulx = (q & 1) ? midx : b.ulx;
    c.urx = (q & 1) ? b.urx : midx;
    c.uly = (q & 2) ? midy : b.uly;
    c.bry = (q & 2) ? b.bry : midy;
    return c;
}

// register budget: see DESIGN.md "footprint" (GFO_QT_WAVES = waves per SIMD the allocation aims at; 0 = compiler default)
#ifndef GFO_QT_WAVES
#define GFO_QT_WAVES 6   // 80 registers instead of 102: +0.9 % pipeline throughput; 8 (64, spilling) costs 12 us of its own
#endif
#if GFO_QT_WAVES > 0
#define QT_OCC_ATTR __attribute__((amdgpu_waves_per_eu(GFO_QT_WAVES, GFO_QT_WAVES)))
#else
#define QT_OCC_ATTR
#endif
#include "k_quadtree_body.inc"
#include "k_quadtree_wide.inc"
#include "k_blur_dev.inc"

__global__ __launch_bounds__(QT_MAX_THREADS) QT_OCC_ATTR void k_quadtree(const GfoGeom* __restrict__ gp, const uint32_t* __restrict__ cand,
                                                                         const int* __restrict__ cand_cnt, uint16_t* __restrict__ node_of_all,
                                                                         uint32_t* __restrict__ sel, int* __restrict__ sel_cnt, int* __restrict__ flags,
                                                                         int ncap, int klds, int level0, unsigned long long* __restrict__ dbg_ts)
{
    extern __shared__ __attribute__((aligned(16))) uint8_t qt_lds[];
    qt_body(gp, cand, cand_cnt, node_of_all, sel, sel_cnt, flags, ncap, klds, dbg_ts, blockIdx.x, level0 + blockIdx.y, qt_lds);
}

__global__ __launch_bounds__(QT_MAX_THREADS) QT_OCC_ATTR void k_quadtree_gmem(const GfoGeom* __restrict__ gp, const uint32_t* __restrict__ cand,
                                                                              const int* __restrict__ cand_cnt, uint16_t* __restrict__ node_of_all,
                                                                              uint32_t* __restrict__ sel, int* __restrict__ sel_cnt, int* __restrict__ flags,
                                                                              int ncap, unsigned long long* __restrict__ dbg_ts,
                                                                              uint8_t* __restrict__ scratch, unsigned long long scratch_stride)
{
    uint8_t* state = scratch + ((unsigned long long)blockIdx.y * gridDim.x + blockIdx.x) * scratch_stride;
    qt_body(gp, cand, cand_cnt, node_of_all, sel, sel_cnt, flags, ncap, 0, dbg_ts, blockIdx.x, blockIdx.y, state);
}

// Quadtree AND blur of a per-frame batch (<= 8 images) as one launch (round 5).  The two are independent (both read the pyramid;
// the quadtree the FAST candidates) and used to run side by side on two streams: an event fork behind k_fast and a join in front
// of k_orient_desc, which on the device are ~7 + ~6 us of nothing on the critical path of a 0.19-ms frame
// (profiles/stereo_direct_r05.txt, the timeline).  Here workgroups [0, nqt) are k_quadtree's, level-major, and the rest are the
// blur's 256-thread blocks, four to a workgroup: no second stream, no events, and the blur's blocks fill the CUs the sixteen
// quadtree workgroups leave idle.  (Batches keep the two launches: there the blur wants its own launch shape and XCD placement.)
// (1024-thread workgroups: four waves per SIMD is all a CU can hold of them -- the whole register file of 128 is theirs)
__global__ __launch_bounds__(QT_MAX_THREADS) __attribute__((amdgpu_waves_per_eu(4, 4))) void k_quadtree_blur(const GfoGeom* __restrict__ gp, const uint32_t* __restrict__ cand,
                                                                              const int* __restrict__ cand_cnt, uint16_t* __restrict__ node_of_all,
                                                                              uint32_t* __restrict__ sel, int* __restrict__ sel_cnt, int* __restrict__ flags,
                                                                              int ncap, int klds, unsigned long long* __restrict__ dbg_ts, GfoInput in,
                                                                              const uint8_t* __restrict__ pyr, uint8_t* __restrict__ blur, int nimg, int nqt,
                                                                              int blur_blocks, int wide)
{
    extern __shared__ __attribute__((aligned(16))) uint8_t qt_lds[];
    const int b = blockIdx.x;
    if (b < nqt) {
        const int img = b % nimg, level = b / nimg;
        // the one-thread-per-node form (k_quadtree_wide.inc) when the node list fits the workgroup and the level's keys fit LDS
        const GfoGeom& g0 = *gp;
        const int kraw = min(cand_cnt[(img * g0.nlevels + level) * GFO_CNT_STRIDE], g0.lv[level].cand_cap);
        if (wide && kraw <= klds)
            qt_body_wide(gp, cand, cand_cnt, sel, sel_cnt, flags, ncap, klds, dbg_ts, img, level, qt_lds);
        else
            qt_body(gp, cand, cand_cnt, node_of_all, sel, sel_cnt, flags, ncap, klds, dbg_ts, img, level, qt_lds);
        return;
    }
    const GfoGeom& g = *gp;
    const int v = (b - nqt) * (QT_MAX_THREADS / 256) + (int)(threadIdx.x >> 8);      // the blur's block index over all images
    const int img = v / blur_blocks, bx = v - img * blur_blocks;
    if (img >= nimg) return;
    const int nb = g.blur_total_b, tid = threadIdx.x & 255;
    if (bx < nb)
        blur_body<true>(g, in, pyr, blur, bx, img, tid);
    else
        blur_body<false>(g, in, pyr, blur, bx - nb, img, tid);
}

size_t gfo_quadtree_lds_bytes(int ncap, int klds)
{
    int p2 = 1;
    while (p2 < ncap) p2 <<= 1;
    return (size_t)p2 * 8 + (size_t)ncap * (2 * sizeof(QtBox) + 2 * 4 + 4 * 4 + 4 * 4 + 3 * 4) + QT_PART * 4 + (size_t)klds * 6 + 64;
}

// Keys a level keeps in LDS (6 B each): seven times its quota, in steps of 512, at most `cap`.  Measured candidate counts per
// level are 2-6 x the quota (752x480 @2000: level 0 up to 2700 for a quota of 434, level 7 ~550 for 122); a level with more
// candidates than that runs the same code on its keys in L2.
static int qt_level_klds(const GfoLevel& L, int cap)
{
    int k = ((7 * L.quota > 1024 ? 7 * L.quota : 1024) + 511) & ~511;
    if (k > L.cand_cap) k = (L.cand_cap + 63) & ~63;
    return k < cap ? k : cap;
}

int gfo_few_max()
{
    // 16: the frame combiner's batches at K = 8 .. 16 camera threads are 6 .. 16 images (8: 59.1 / 57.2 k images/s at K = 16,
    // 16: 62.3 / 63.3 k, 32: 59.4 / 68.6 k; K = 4, 8 within the noise -- profiles/stereo_direct_r05.txt)
    static const int v = getenv("GFO_FEW_MAX") ? atoi(getenv("GFO_FEW_MAX")) : 16;
    return v < 1 ? 1 : v;
}

static bool launch_quadtree(gfo_ctx* c, int nimg, const GfoInput* blur_in)
{
    const int nl = c->g.nlevels;
    int ncap = 0;
    for (int l = 0; l < nl; l++) ncap = c->g.lv[l].node_cap > ncap ? c->g.lv[l].node_cap : ncap;
    // Keys kept in LDS (6 B each).  Alone, a large batch ran 4 % faster with the keys in L2 (four workgroups per CU instead
    // of two); in the running pipeline LDS-resident keys win (214.4k -> 219.1k frames/s, same-box A/B: less L2 traffic next to
    // the other contexts' kernels).  For a handful of images (the per-frame latency path) the level-0 workgroup IS the critical
    // path: 6144 keys of LDS and 1024 threads (0.237 -> 0.214 ms per stereo frame).
    // Round 5: batches keep 7 x quota keys of the LARGEST level instead of 6144 (752x480 @2000: 3072; measured candidate counts
    // are <= 2700 at level 0, ~2000 below) and the scan scratch is one int per wave instead of per thread: 75 -> 52.7 KB per
    // workgroup = THREE workgroups per CU instead of two for all eight levels.  k_quadtree alone 163 -> 116 us per 256 images
    // (profiles/quadtree_occupancy_r05.txt); a level with more candidates than that runs on its keys in L2, same code.
    static const int klds_env = getenv("GFO_QT_KLDS") ? atoi(getenv("GFO_QT_KLDS")) : -1;
    const bool few = nimg <= gfo_few_max();
    int klds = klds_env >= 0 ? klds_env : (few ? 6144 : qt_level_klds(c->g.lv[0], 6144));
    while (klds > 0 && gfo_quadtree_lds_bytes(ncap, klds) > 150 * 1024) klds -= 1024;
    const size_t lds = gfo_quadtree_lds_bytes(ncap, klds);
    const bool gmem = lds > 160 * 1024;   // state of the largest level does not fit LDS: scratch in HBM (plan() sized it)
    // the per-pass key loops are latency-bound inside a workgroup: large quotas (1080p @4000 features) get
    // 1024 threads per (image, level), the 752x480 @2000 case runs best with 256
    static const int nt_env = getenv("GFO_QT_THREADS") ? atoi(getenv("GFO_QT_THREADS")) : 0;
    const bool nt_ok = nt_env >= 64 && nt_env <= QT_MAX_THREADS && (nt_env & 63) == 0;   // anything else: the default
    // (with LDS-resident keys 128 / 192 / 256 threads give the same pipeline rate, 219k; 256 is the fastest alone: 96 us)
    const int nthreads = nt_ok ? nt_env : (c->g.lv[0].quota >= 600 && nimg < 48 ? 1024 : (few ? 1024 : 256));

    // Level groups (round 5, opt-in: GFO_QT_GROUPS=auto | a,b).  The dynamic LDS of a launch is one figure for all its
    // workgroups, sized by level 0, although level 7 (122 nodes, ~550 keys) needs 16 KB.  With GFO_QT_GROUPS a batch is launched
    // as up to three consecutive level ranges, each with the node and key capacity of ITS first (largest) level (`auto`: ranges
    // cut where the workgroups a CU can hold -- LDS, and 24 waves -- step from <= 3 to 4-5 to >= 6; `a,b`: cut in front of
    // levels a and b).  MEASURED AND NOT ADOPTED: 136-148 us alone against 116 us for the single launch at three workgroups per
    // CU -- a launch boundary makes the small levels wait for the slowest level-0 workgroup instead of back-filling behind it --
    // and the pipeline rate is the same within noise for every form (277.6-281.5k frames/s).  Kept as a knob, parity-tested.
    struct Grp { int l0, n, ncap, klds; size_t lds; };
    Grp grp[GFO_MAX_LEVELS];
    int ngrp = 0;
    const char* groups_env = getenv("GFO_QT_GROUPS");   // (read per launch: the tests switch it inside one process)
    const bool auto_groups = groups_env && groups_env[0] == 'a';
    const bool grouped = !gmem && !few && groups_env && !(groups_env[0] == '0' && groups_env[1] == 0);
    if (grouped) {
        int cut_a = -1, cut_b = -1;
        if (!auto_groups) sscanf(groups_env, "%d,%d", &cut_a, &cut_b);
        const int wave_wgs = 24 / (nthreads / 64) > 0 ? 24 / (nthreads / 64) : 1;
        int prev_class = -1;
        for (int l = 0; l < nl; l++) {
            const GfoLevel& L = c->g.lv[l];
            const int kl = qt_level_klds(L, klds);
            const size_t need = gfo_quadtree_lds_bytes(L.node_cap, kl);
            int wgs = (int)((size_t)160 * 1024 / ((need + 1279) / 1280 * 1280));
            if (wgs > wave_wgs) wgs = wave_wgs;
            const int cls = wgs <= 3 ? 0 : (wgs <= 5 ? 1 : 2);
            const bool cut = auto_groups ? (cls != prev_class) : (l == cut_a || l == cut_b);
            if (l == 0 || cut) {
                grp[ngrp].l0 = l; grp[ngrp].n = 0; grp[ngrp].ncap = L.node_cap; grp[ngrp].klds = kl; grp[ngrp].lds = need;
                ngrp++;
            }
            Grp& G = grp[ngrp - 1];
            G.n++;
            if (L.node_cap > G.ncap) G.ncap = L.node_cap;      // (quotas shrink with the level; kept general)
            if (kl > G.klds) G.klds = kl;
            G.lds = gfo_quadtree_lds_bytes(G.ncap, G.klds);
            prev_class = cls;
        }
    } else {
        grp[0].l0 = 0; grp[0].n = nl; grp[0].ncap = ncap; grp[0].klds = klds; grp[0].lds = lds;
        ngrp = 1;
    }
    size_t lds_max = 0;
    for (int i = 0; i < ngrp; i++) lds_max = grp[i].lds > lds_max ? grp[i].lds : lds_max;
    if (!gmem && lds_max > 64 * 1024 && lds_max > c->qt_lds_granted) {
        // raised per context (= per device; contexts may be driven from different threads): the grant only grows
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(k_quadtree), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_max);
        if (e == hipSuccess) e = hipFuncSetAttribute(reinterpret_cast<const void*>(k_quadtree_blur), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_max);
        if (e != hipSuccess) {
            (void)hipGetLastError();
            if (c->launch_err.empty()) c->launch_err = std::string("stage quadtree: cannot raise the dynamic LDS limit: ") + hipGetErrorString(e);
            return true;      // (the error is recorded: the pipeline driver stops before anything consumes the selection)
        }
        c->qt_lds_granted = lds_max;
    }
    static const bool timing = getenv("GFO_QT_TIMING") != nullptr;
    unsigned long long* d_ts = nullptr;
    if (timing && hipMalloc(&d_ts, 128 * sizeof(unsigned long long)) == hipSuccess) (void)hipMemsetAsync(d_ts, 0, 128 * sizeof(unsigned long long), c->stream);
    // per-frame batches: quadtree and blur as ONE launch (k_quadtree_blur) when the caller asks for it and the shapes allow
    const int blur_blocks = c->g.total_tiles + c->g.blur_total_b;
    const bool fused = blur_in && !gmem && ngrp == 1 && nthreads == QT_MAX_THREADS && blur_blocks > 0;
    if (blur_in && !fused) return false;      // the caller launches the two side by side as before
    gfo_prof_begin(c, ST_QUADTREE);
    if (fused) {
        const int nqt = nimg * nl;
        const int blur_wgs = (nimg * blur_blocks + QT_MAX_THREADS / 256 - 1) / (QT_MAX_THREADS / 256);
        // (GFO_QT_WIDE=0: the general body for every workgroup)
        const char* wide_env = getenv("GFO_QT_WIDE");
        const int wide = grp[0].ncap <= QT_MAX_THREADS && grp[0].klds > 0 && !(wide_env && wide_env[0] == '0') ? 1 : 0;
        GFO_LAUNCH(c, k_quadtree_blur, dim3(nqt + blur_wgs), dim3(QT_MAX_THREADS), grp[0].lds, c->stream, c->d_geom, c->d_cand, c->d_cand_cnt, c->d_node_of,
                           c->d_sel, c->d_sel_cnt, c->d_flags, grp[0].ncap, grp[0].klds, d_ts, *blur_in, c->d_pyr, c->d_blur, nimg, nqt, blur_blocks, wide);
    } else if (gmem)
        GFO_LAUNCH(c, k_quadtree_gmem, dim3(nimg, nl), dim3(1024), 0, c->stream, c->d_geom, c->d_cand, c->d_cand_cnt, c->d_node_of,
                           c->d_sel, c->d_sel_cnt, c->d_flags, ncap, d_ts, c->d_qt_scratch, (unsigned long long)c->qt_scratch_stride);
    else
        for (int i = 0; i < ngrp; i++)
            GFO_LAUNCH(c, k_quadtree, dim3(nimg, grp[i].n), dim3(nthreads), grp[i].lds, c->stream, c->d_geom, c->d_cand, c->d_cand_cnt, c->d_node_of,
                               c->d_sel, c->d_sel_cnt, c->d_flags, grp[i].ncap, grp[i].klds, grp[i].l0, d_ts);
    if (d_ts) {   // debugging aid: blocks until the kernel is done and prints the phase times of block (0, 0)
        unsigned long long ts[128];
        (void)hipStreamSynchronize(c->stream);
        (void)hipMemcpy(ts, d_ts, sizeof ts, hipMemcpyDeviceToHost);
        (void)hipFree(d_ts);
        fprintf(stderr, "[gfo] quadtree block (0,0), 10 ns ticks since start:");
        for (int i = 0; i < 128 && ts[i]; i++) fprintf(stderr, " %llu:%llu", ts[i] & 255, (ts[i] >> 8) - (ts[0] >> 8));
        fprintf(stderr, "\n");
    }
    gfo_prof_end(c);
    return true;
}

void gfo_launch_quadtree(gfo_ctx* c, int nimg) { (void)launch_quadtree(c, nimg, nullptr); }

// quadtree + blur of a per-frame batch in one launch; false (nothing launched) when the configuration does not allow it
bool gfo_launch_quadtree_blur(gfo_ctx* c, const GfoInput& in, int nimg) { return launch_quadtree(c, nimg, &in); }

// Every __global__ of this translation unit, for gfo_preload_kernels (gfo_api.hip): the runtime loads a code object and
// registers a kernel lazily, on the first launch that needs it; gfo_ctx_create resolves them all once per device under a
// mutex so that no two host threads ever race through that first-launch path (round 3: eight threads, first k_pack_results).
void gfo_kernels_quadtree(std::vector<const void*>& v)
{
    v.push_back((const void*)k_quadtree); v.push_back((const void*)k_quadtree_gmem); v.push_back((const void*)k_quadtree_blur);
}
