// k_orient_desc.hip -- IC_Angle (ORBextractor.cc:76-103), computeOrbDescriptor (:107-146) and
// the keypoint bookkeeping of ComputeKeyPointsOctTree / operator() (:839-849, 1144-1172).
//
// One wavefront per selected keypoint.
//   angle      : the 749-px disc is swept two rows (+v / -v) per step, 31 lanes each; the
//                int32 moments are wave-reduced with lane shuffles; fastAtan2 is the shared
//                plain-fp32 polynomial of include/gfo_sincos.h.
//   descriptor : 37x37 window of the BLURRED level staged in LDS (the rotated pattern reaches
//                18 px); each lane evaluates 4 of the 256 pair tests; the wave
//                ballot of test r*64+lane IS descriptor bytes 8r..8r+7, so the 32 bytes leave
//                as four 64-bit words without any bit shuffling.
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

// Copies `rows` rows of `dpr` dwords from global (row pitch `pitch`) to LDS (row pitch `lpitch` bytes);
// every lane issues all of its global loads before the first LDS store.
template <int MAXK>
__device__ __forceinline__ void stage_rows(const uint8_t* __restrict__ src, long long pitch, uint8_t* lds, int lpitch,
                                           int dpr, int rows, int lane)
{
    const int ndw = dpr * rows;
    const float inv = 1.0f / (float)dpr;
    unsigned v[MAXK];
    int dst[MAXK];
#pragma unroll
    for (int k = 0; k < MAXK; k++) {
        const int t = k * 64 + lane;
        const int tt = min(t, ndw - 1);
        const int r = (int)(((float)tt + 0.5f) * inv);
        const int c = tt - r * dpr;
        dst[k] = t < ndw ? r * lpitch + 4 * c : -1;
        v[k] = *reinterpret_cast<const unsigned*>(src + (long long)r * pitch + 4 * c);
    }
#pragma unroll
    for (int k = 0; k < MAXK; k++)
        if (dst[k] >= 0) *reinterpret_cast<unsigned*>(lds + dst[k]) = v[k];
}

__global__ __launch_bounds__(256) void k_orient_desc(const GfoGeom* __restrict__ gp, GfoInput in,
                                                     const uint8_t* __restrict__ pyr, const uint8_t* __restrict__ blur,
                                                     const uint32_t* __restrict__ sel, const int* __restrict__ sel_cnt,
                                                     gfo_keypoint* __restrict__ kp_out, uint8_t* __restrict__ desc_out,
                                                     int* __restrict__ kp_cnt, int* __restrict__ flags, int nimg,
                                                     int blocks_per_img)
{
    __shared__ __attribute__((aligned(16))) uint8_t s_win[4][DW * DWP];
    __shared__ __attribute__((aligned(16))) uint8_t s_pat[4][OW * OWP];
    const GfoGeom& g = *gp;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
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
            const int xcd = b & 7, s = b >> 3;
            img = (s / blocks_per_img) * 8 + xcd;
            blk = s - (s / blocks_per_img) * blocks_per_img;
        } else {  // the < 8 images left over: plain order
            const int r = b - swz;
            img = groups * 8 + r / blocks_per_img;
            blk = r - (r / blocks_per_img) * blocks_per_img;
        }
    }
    const int slot = blk * 4 + wave;
    // level of this slot: prefix over the per-level counts (wave-uniform)
    int level = -1, idx = 0, acc = 0;
    for (int l = 0; l < g.nlevels; l++) {
        const int c = sel_cnt[img * g.nlevels + l];
        if (level < 0 && slot < acc + c) {
            level = l;
            idx = slot - acc;
        }
        acc += c;
    }
    const int total = acc;
    if (slot == 0 && lane == 0) {
        kp_cnt[img] = min(total, g.kp_stride);
        if (total > g.kp_stride) atomicOr(&flags[0], 8);
    }
    if (level < 0 || slot >= g.kp_stride) return;
    const GfoLevel& L = g.lv[level];
    const uint32_t key = sel[(long long)img * g.total_sel_cap + L.sel_off + idx];
    const int x = (int)(key & 0xFFF) + GFO_MIN_BORDER, y = (int)((key >> 12) & 0xFFF) + GFO_MIN_BORDER;  // :845-846
    const int score = (int)(key >> 24);

    // ---- stage both windows (dword loads, all in flight together) ----
    int pitch;
    const uint8_t* lv = gfo_level_ptr(g, in, pyr, level, img, &pitch);
    const int ox_al = (x - GFO_HALF_PATCH) & ~3, ooff = (x - GFO_HALF_PATCH) - ox_al;
    const int odpr = (ooff + OW + 3) >> 2;  // <= 9
    uint8_t* pat = s_pat[wave];
    stage_rows<5>(lv + (long long)(y - GFO_HALF_PATCH) * pitch + ox_al, pitch, pat, OWP, odpr, OW, lane);
    const int wx_al = (x - 18) & ~3, woff = (x - 18) - wx_al;
    const int wdpr = (woff + DW + 3) >> 2;  // <= 11
    uint8_t* win = s_win[wave];
    stage_rows<7>(blur + (long long)img * g.blur_img_stride + L.blur_off + (long long)(y - 18) * L.pitch + wx_al, L.pitch, win,
                  DWP, wdpr, DW, lane);
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");

    // ---- IC_Angle on the unblurred patch ----
    const uint8_t* center = pat + GFO_HALF_PATCH * OWP + ooff + GFO_HALF_PATCH;
    const int half = lane >> 5;           // 0: row +v, 1: row -v
    const int u = (lane & 31) - GFO_HALF_PATCH;
    const bool col_ok = (lane & 31) < 31;
    int m10 = 0, m01 = 0;
    if (lane < 31) m10 = u * (int)center[u];
#pragma unroll
    for (int v = 1; v <= GFO_HALF_PATCH; v++) {
        const int d = k_umax[v];
        if (col_ok && u >= -d && u <= d) {
            const int val = center[(half ? -v : v) * OWP + u];
            m10 += u * val;
            m01 += half ? -v * val : v * val;
        }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        m10 += __shfl_xor(m10, o);
        m01 += __shfl_xor(m01, o);
    }
    const float angle = gfo_fast_atan2f((float)m01, (float)m10);

    // ---- descriptor on the blurred window ----
    const float factorPI = (float)(3.1415926535897932384626433832795 / 180.f);
    float a, b;
    gfo_sincosf(angle * factorPI, &b, &a);
    const uint8_t* wc = win + 18 * DWP + woff + 18;
    unsigned long long word = 0;
#pragma unroll
    for (int r = 0; r < 4; r++) {
        const PatQuad p = k_pattern[r * 64 + lane];
        const float x0 = (float)p.x0, y0 = (float)p.y0, x1 = (float)p.x1, y1 = (float)p.y1;
        const int iy0 = (int)rintf(x0 * b + y0 * a), ix0 = (int)rintf(x0 * a - y0 * b);
        const int iy1 = (int)rintf(x1 * b + y1 * a), ix1 = (int)rintf(x1 * a - y1 * b);
        const int t0 = wc[iy0 * DWP + ix0], t1 = wc[iy1 * DWP + ix1];
        const unsigned long long m = __ballot(t0 < t1);
        if (lane == r) word = m;
    }
    const long long o = (long long)img * g.kp_stride + slot;
    if (lane < 4) reinterpret_cast<unsigned long long*>(desc_out + o * 32)[lane] = word;
    if (lane == 0) {
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
    const int bpi = (c->g.kp_stride + 3) / 4;
    dim3 grid((unsigned)bpi * (unsigned)nimg);
    gfo_prof_begin(c, ST_ORIENT_DESC);
    hipLaunchKernelGGL(k_orient_desc, grid, dim3(256), 0, c->stream, c->d_geom, in, c->d_pyr, c->d_blur, c->d_sel,
                       c->d_sel_cnt, c->d_kp, c->d_desc, c->d_kp_cnt, c->d_flags, nimg, bpi);
    gfo_prof_end(c);
}
