// k_orient_desc.hip -- IC_Angle (ORBextractor.cc:76-103), computeOrbDescriptor (:107-146) and
// the keypoint bookkeeping of ComputeKeyPointsOctTree / operator() (:839-849, 1144-1172).
//
// TWO keypoints per wavefront, one per 32-lane half.  The per-keypoint work is a chain of dependent
// memory accesses (selection word -> window addresses -> window bytes), so the kernel is latency-bound;
// two independent chains per wave halve the number of waves the same occupancy has to retire.
//   staging    : the 31x31 patch of the level and the 37x37 window of the BLURRED level (the rotated
//                pattern reaches 18 px) go to LDS as aligned dwords, all loads of a
//                lane in flight before the first LDS store;
//   angle      : one disc row per step, 31 lanes of the half; the int32 moments are reduced inside
//                the half with lane shuffles; fastAtan2 is the shared plain-fp32 polynomial of
//                include/gfo_sincos.h;
//   descriptor : each lane evaluates 8 of the 256 pair tests; the half of the wave ballot that belongs
//                to a keypoint in round r IS its descriptor bytes 4r..4r+3, so the 32 bytes leave as
//                eight 32-bit words without any bit shuffling.
// Output rows are laid out level by level, inside a level in list order (:1144-1161).
// No workgroup barrier is used: each wave owns its LDS window (LDS operations of one wave
// execute in issue order), so trailing waves may exit early.
#include "gfo_internal.h"
#include "../../include/gfo_sincos.h"

struct PatQuad { signed char x0, y0, x1, y1; };
__device__ const PatQuad k_pattern[256] = {
#include "../../include/gfo_pattern.inc"
};
__device__ const int k_umax[16] = {15, 15, 15, 15, 14, 14, 14, 13, 13, 12, 11, 10, 9, 8, 6, 3};  // ORBextractor.cc:451-468

#define DW 37          // descriptor window (blurred level), rows
#define DWP 44         // its LDS pitch: 11 dwords cover 37 px + up to 3 px of alignment slack
#define OW 31          // orientation patch (unblurred level), rows
#define OWP 40         // its LDS pitch: 9 dwords (+1 spare)

#define OD_PAT_K 9      // dword loads per lane for the 31-row patch  (<= 9 dwords x 31 rows = 279, 32 lanes)
#define OD_WIN_K 13     // dword loads per lane for the 37-row window (<= 11 dwords x 37 rows = 407, 32 lanes)

__global__ __launch_bounds__(256) void k_orient_desc(const GfoGeom* __restrict__ gp, GfoInput in,
                                                     const uint8_t* __restrict__ pyr, const uint8_t* __restrict__ blur,
                                                     const uint32_t* __restrict__ sel, const int* __restrict__ sel_cnt,
                                                     gfo_keypoint* __restrict__ kp_out, uint8_t* __restrict__ desc_out,
                                                     int* __restrict__ kp_cnt, int* __restrict__ flags, int nimg,
                                                     int blocks_per_img)
{
    __shared__ __attribute__((aligned(16))) uint8_t s_win[8][DW * DWP];   // [wave*2 + half]
    __shared__ __attribute__((aligned(16))) uint8_t s_pat[8][OW * OWP];
    const GfoGeom& g = *gp;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;  // uniform: keep it scalar
    const int half = lane >> 5, hl = lane & 31;
    // XCD-aware placement (speed only): workgroups are dealt round-robin over the 8 XCDs, so workgroup b runs
    // on XCD b % 8.  All workgroups of one image are given the same b % 8, so an image's windows (2.2 MB of
    // pyramid + blurred pyramid, overlapping heavily between keypoints) are fetched into ONE 4-MB L2 instead
    // of all eight.  Any placement gives the same results.
    int img, blk;
    {
        const int b = blockIdx.x;
        const int groups = nimg >> 3;  // full groups of 8 images
        const int swz = groups * 8 * blocks_per_img;
        if (b < swz) {
            const int xcd = b & 7, s_ = b >> 3;
            img = (s_ / blocks_per_img) * 8 + xcd;
            blk = s_ - (s_ / blocks_per_img) * blocks_per_img;
        } else {  // the < 8 images left over: plain order
            const int r = b - swz;
            img = groups * 8 + r / blocks_per_img;
            blk = r - (r / blocks_per_img) * blocks_per_img;
        }
    }
    // per-level counts (wave-uniform): a slot's level follows from their prefix; output rows are level by
    // level, list order inside a level (:1144-1161)
    int total = 0;
    for (int l = 0; l < g.nlevels; l++) total += sel_cnt[img * g.nlevels + l];
    if (blk == 0 && wave == 0 && lane == 0) {
        kp_cnt[img] = min(total, g.kp_stride);
        if (total > g.kp_stride) atomicOr(&flags[0], 8);
    }
    const int nkp = min(total, g.kp_stride);
    const int slot0 = (blk * 4 + wave) * 2;
    if (slot0 >= nkp) return;                      // wave-uniform
    const bool act = slot0 + half < nkp;           // the second half may run past the end: it redoes the last keypoint
    const int slot = min(slot0 + half, nkp - 1);
    int level = 0, idx = slot, acc = 0;
    for (int l = 0; l < g.nlevels; l++) {
        const int c = sel_cnt[img * g.nlevels + l];
        if (slot >= acc && slot < acc + c) {
            level = l;
            idx = slot - acc;
        }
        acc += c;
    }
    const GfoLevel& L = g.lv[level];
    const uint32_t key = sel[(long long)img * g.total_sel_cap + L.sel_off + idx];
    const int x = (int)(key & 0xFFF) + GFO_MIN_BORDER, y = (int)((key >> 12) & 0xFFF) + GFO_MIN_BORDER;  // :845-846
    const int score = (int)(key >> 24);

    // ---- stage both windows of this half's keypoint (dword loads, all in flight together) ----
    int pitch;
    const uint8_t* lv = gfo_level_ptr(g, in, pyr, level, img, &pitch);
    const int ox_al = (x - GFO_HALF_PATCH) & ~3, ooff = (x - GFO_HALF_PATCH) - ox_al;
    const int odpr = (ooff + OW + 3) >> 2;  // <= 9
    const int wx_al = (x - 18) & ~3, woff = (x - 18) - wx_al;
    const int wdpr = (woff + DW + 3) >> 2;  // <= 11
    const int lpitch = L.pitch;
    const uint8_t* psrc = lv + (long long)(y - GFO_HALF_PATCH) * pitch + ox_al;
    const uint8_t* wsrc = blur + (long long)img * g.blur_img_stride + L.blur_off + (long long)(y - 18) * lpitch + wx_al;
    uint8_t* pat = s_pat[wave * 2 + half];
    uint8_t* win = s_win[wave * 2 + half];
    {
        // the LDS offsets are recomputed at store time rather than kept: 22 fewer live registers while the
        // loads are in flight (occupancy matters more than the extra integer operations here)
        unsigned vp[OD_PAT_K], vw[OD_WIN_K];
        const int pn = odpr * OW, wn = wdpr * DW;
        const float pinv = 1.0f / (float)odpr, winv = 1.0f / (float)wdpr;
#pragma unroll
        for (int k = 0; k < OD_PAT_K; k++) {
            const int tt = min(k * 32 + hl, pn - 1);
            const int r = (int)(((float)tt + 0.5f) * pinv);
            vp[k] = *reinterpret_cast<const unsigned*>(psrc + (long long)r * pitch + 4 * (tt - r * odpr));
        }
#pragma unroll
        for (int k = 0; k < OD_WIN_K; k++) {
            const int tt = min(k * 32 + hl, wn - 1);
            const int r = (int)(((float)tt + 0.5f) * winv);
            vw[k] = *reinterpret_cast<const unsigned*>(wsrc + (long long)r * lpitch + 4 * (tt - r * wdpr));
        }
#pragma unroll
        for (int k = 0; k < OD_PAT_K; k++) {
            const int t = k * 32 + hl;
            if (t < pn) {
                const int r = (int)(((float)t + 0.5f) * pinv);
                *reinterpret_cast<unsigned*>(pat + r * OWP + 4 * (t - r * odpr)) = vp[k];
            }
        }
#pragma unroll
        for (int k = 0; k < OD_WIN_K; k++) {
            const int t = k * 32 + hl;
            if (t < wn) {
                const int r = (int)(((float)t + 0.5f) * winv);
                *reinterpret_cast<unsigned*>(win + r * DWP + 4 * (t - r * wdpr)) = vw[k];
            }
        }
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");

    // ---- IC_Angle on the unblurred patch: lane = column u, one disc row per step ----
    const uint8_t* center = pat + GFO_HALF_PATCH * OWP + ooff + GFO_HALF_PATCH;
    const int u = hl - GFO_HALF_PATCH;
    int m10 = 0, m01 = 0;
    if (hl < 31) {
#pragma unroll
        for (int v = -GFO_HALF_PATCH; v <= GFO_HALF_PATCH; v++) {
            const int d = k_umax[v < 0 ? -v : v];
            if (u >= -d && u <= d) {
                const int val = center[v * OWP + u];
                m10 += u * val;
                m01 += v * val;
            }
        }
    }
#pragma unroll
    for (int o = 16; o > 0; o >>= 1) {  // stays inside the 32-lane half
        m10 += __shfl_xor(m10, o);
        m01 += __shfl_xor(m01, o);
    }
    const float angle = gfo_fast_atan2f((float)m01, (float)m10);

    // ---- descriptor on the blurred window ----
    const float factorPI = (float)(3.1415926535897932384626433832795 / 180.f);
    float a, b;
    gfo_sincosf(angle * factorPI, &b, &a);
    const uint8_t* wc = win + 18 * DWP + woff + 18;
    unsigned word = 0;
#pragma unroll
    for (int r = 0; r < 8; r++) {
        const PatQuad p = k_pattern[r * 32 + hl];
        const float x0 = (float)p.x0, y0 = (float)p.y0, x1 = (float)p.x1, y1 = (float)p.y1;
        const int iy0 = (int)rintf(x0 * b + y0 * a), ix0 = (int)rintf(x0 * a - y0 * b);
        const int iy1 = (int)rintf(x1 * b + y1 * a), ix1 = (int)rintf(x1 * a - y1 * b);
        const int t0 = wc[iy0 * DWP + ix0], t1 = wc[iy1 * DWP + ix1];
        const unsigned long long m = __ballot(t0 < t1);
        if (hl == r) word = (unsigned)(m >> (32 * half));  // tests 32r..32r+31 of THIS half's keypoint
    }
    if (!act) return;
    const long long o = (long long)img * g.kp_stride + slot;
    if (hl < 8) reinterpret_cast<unsigned*>(desc_out + o * 32)[hl] = word;
    if (hl == 0) {
        gfo_keypoint q;
        q.x = (float)x;
        q.y = (float)y;
        if (level != 0) {  // :1164-1170
            q.x = q.x * L.scale;
            q.y = q.y * L.scale;
        }
        q.size = (float)L.patch_size;
        q.angle = angle;
        q.response = (float)score;
        q.octave = level;
        q.class_id = -1;
        kp_out[o] = q;
    }
}

void gfo_launch_orient_desc(gfo_ctx* c, const GfoInput& in, int nimg)
{
    const int bpi = (c->g.kp_stride + 7) / 8;  // 4 waves x 2 keypoints per workgroup
    dim3 grid((unsigned)bpi * (unsigned)nimg);
    gfo_prof_begin(c, ST_ORIENT_DESC);
    hipLaunchKernelGGL(k_orient_desc, grid, dim3(256), 0, c->stream, c->d_geom, in, c->d_pyr, c->d_blur, c->d_sel,
                       c->d_sel_cnt, c->d_kp, c->d_desc, c->d_kp_cnt, c->d_flags, nimg, bpi);
    gfo_prof_end(c);
}
