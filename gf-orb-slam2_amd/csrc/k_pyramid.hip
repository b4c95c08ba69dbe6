// k_pyramid.hip -- ComputePyramid (ORBextractor.cc:1176-1201): level l = cv::resize(level l-1,
// INTER_LINEAR) in OpenCV's 11-bit fixed point.  The coefficient tables (xofs/alpha, yofs/beta)
// are built on the host in double exactly as cv::resize builds them (gfo_api.hip, plan()).
//
// HBM-bound byte work.  A thread owns a 4x4 block of output pixels: the column tables are read once,
// the <= 6 source rows the block touches are fetched up front as three coalesced dwords each (12 bytes
// cover the <= 11-byte footprint of 4 outputs for scale factors up to 2) so every load of the thread is
// in flight at once, each source row is filtered horizontally once, and the four output rows are
// blended from those sums.  (Blocks that touch more than 6 source rows -- scale factors above 1.5 --
// take the row-by-row path.)
// One launch per level (level l needs all of level l-1); all images of the batch per launch.
// The 19-px reflect frame of the reference is never materialised: nothing on the extraction
// path reads it (gfo_pyramid_level rebuilds it on request).
#include "gfo_internal.h"
#include <stdlib.h>

#define RS_STRIP 4
#define RS_MAXR 6   // source rows a 4-row strip touches at scale factors up to 1.5 (4*1.5 rows)

struct HQuad {
    int h[4];
};

typedef unsigned short gfo_u16x2 __attribute__((ext_vector_type(2)));

#if GFO_OCV_RESIZE == 1
// the float variant's horizontal step: S[x0] * (1 - fx) + S[x1] * fx in float (un-fused: -ffp-contract=off), kept as float bits
__device__ __forceinline__ int resize_hlerp(unsigned p0, unsigned p1, int fx_bits)
{
    const float fx = __int_as_float(fx_bits);
    return __float_as_int((float)p0 * (1.f - fx) + (float)p1 * fx);
}
#endif

// One source row segment -> the four horizontal sums.
// `fast`: the 12 bytes at base_x cover every tap.  The window is first shifted so that it starts at sx[0]
// (two v_alignbyte), after which the byte pair of output k sits at the fixed offsets (rel[k], rel[k]+1) < 8:
// one v_perm_b32 spreads the pair into two u16 and one v_dot2_u32_u16 multiplies it by the packed
// coefficients {alpha0, alpha1} of the table -- 2.5 VALU operations per output and row.
__device__ __forceinline__ HQuad resize_hrow(const uint8_t* __restrict__ row, int base_x, bool fast, const int* sx,
                                             const unsigned* sel, const int* cw, int sw)
{
    HQuad q;
    if (fast) {
        const uint32_t* p = reinterpret_cast<const uint32_t*>(row + base_x);
        const unsigned w0 = p[0], w1 = p[1], w2 = p[2];
        const unsigned o0 = (unsigned)(sx[0] - base_x);  // 0..3
        const unsigned A = __builtin_amdgcn_alignbyte(w1, w0, o0), B = __builtin_amdgcn_alignbyte(w2, w1, o0);
#pragma unroll
        for (int k = 0; k < 4; k++) {
            const unsigned two = __builtin_amdgcn_perm(B, A, sel[k]);  // {byte rel, 0, byte rel+1, 0}
#if GFO_OCV_RESIZE == 1
            q.h[k] = resize_hlerp(two & 0xFFFFu, two >> 16, cw[k]);
#else
            q.h[k] = (int)__builtin_amdgcn_udot2(__builtin_bit_cast(gfo_u16x2, two), __builtin_bit_cast(gfo_u16x2, (unsigned)cw[k]), 0u, false);
#endif
        }
    } else {
#pragma unroll
        for (int k = 0; k < 4; k++) {
            const int s1 = min(sx[k] + 1, sw - 1);
#if GFO_OCV_RESIZE == 1
            q.h[k] = resize_hlerp(row[sx[k]], row[s1], cw[k]);
#else
            q.h[k] = (int)row[sx[k]] * (cw[k] & 0xFFFF) + (int)row[s1] * (int)((unsigned)cw[k] >> 16);
#endif
        }
    }
    return q;
}

// One output byte from the horizontal results of its two source rows and the row-table entry.
//   GFO_OCV_RESIZE 0: VResizeLinear<uchar,int,short>'s (((b0 * (r0 >> 4)) >> 16) + ((b1 * (r1 >> 4)) >> 16) + 2) >> 2 [OCV]
//   GFO_OCV_RESIZE 1: cvRound(top * (1 - fy) + bot * fy) in float, saturated
__device__ __forceinline__ unsigned resize_vblend(int bwj, int ha, int hb)
{
#if GFO_OCV_RESIZE == 1
    const float fy = __int_as_float(bwj);
    const float v = __int_as_float(ha) * (1.f - fy) + __int_as_float(hb) * fy;
    return (unsigned)min(max(__float2int_rn(v), 0), 255);
#else
    const unsigned b0 = bwj & 0xFFFF, b1 = (unsigned)bwj >> 16;
    return (((__umul24(b0, ha >> 4) >> 16) + (__umul24(b1, hb >> 4) >> 16) + 2u) >> 2) & 255u;   // operands < 2^16
#endif
}

// Tables: one int2 per output column / row = {source offset, coef0 | coef1 << 16}; every level's table is
// padded with 3 copies of its last entry so a thread may read its 4 entries as two 16-byte loads.
//
// One 4x4 output block (column quad `quad`, rows dy0 .. dy0+3) of a level.
//   source : rows [srow0, srow_hi] of the level below live at src + (r - srow0) * spitch (a whole plane in HBM,
//            or the band of it a workgroup keeps in LDS); sw x sh is the full size of that level
//   output : rows < row_end; a row in [own0, own1) goes to the plane in HBM (dstg), every row to the LDS band
//            (dstl, row lrow0 first) when BAND
// UNI: the strip (hence every row index) is the same for all lanes of the wave, so the source-row pair of an
// output row is picked by a scalar branch instead of per-lane select chains.
template <bool UNI, bool BAND>
__device__ __forceinline__ void resize_block(const uint8_t* __restrict__ src, int spitch, int srow0, int srow_hi, int sh, int sw,
                                             uint8_t* __restrict__ dstg, int gpitch, int own0, int own1,
                                             uint8_t* __restrict__ dstl, int lpitch, int lrow0, int row_end, int quad,
                                             int dy0, const int2* __restrict__ xtab, const int2* __restrict__ ytab, int over_mode)
{
    const int dx0 = quad * 4;
    const int4* xt = reinterpret_cast<const int4*>(xtab + dx0);
    const int4 xa = xt[0], xb = xt[1];
    const int sx[4] = {xa.x, xa.z, xb.x, xb.z};
    const int cw[4] = {xa.y, xa.w, xb.y, xb.w};
    unsigned sel[4];
#pragma unroll
    for (int k = 0; k < 4; k++) {
        const unsigned rel = (unsigned)(sx[k] - sx[0]) & 7u;
        sel[k] = rel | (0x0Cu << 8) | ((rel + 1u) << 16) | (0x0Cu << 24);  // selector 0x0C = constant zero byte
    }
    const int base_x = sx[0] & ~3;
    const int dy1 = min(dy0 + RS_STRIP, row_end);   // rows [dy0, dy1) of the level
    const int4* yt = reinterpret_cast<const int4*>(ytab + dy0);
    const int4 ya = yt[0], yb = yt[1];
    const int syv[4] = {ya.x, ya.z, yb.x, yb.z};
    const int bw[4] = {ya.y, ya.w, yb.y, yb.w};
    const int r_first = min(max(syv[0], 0), sh - 1);
    // The 12-byte row window: eight bytes from sx[0] must cover sx[3]+1 (scale factors up to 2), and the twelve from
    // base_x must be readable.  Inside the row they are; in the last quad(s) of a row they run up to 11 bytes past it,
    // into the next row -- harmless (the selectors never pick those bytes) wherever a next row or arena slack exists:
    // over_mode 1 = the source is a pyramid plane of the arena (256 bytes of slack behind the last one), 2 = the source
    // is the caller's image: allowed unless one of the rows this block loads is the image's last, 0 = never (LDS bands).
    // Without this the last quads take the byte path, and a wave holding one of them (one in two or three) runs BOTH.
    // (the last row this block loads is r_first + RS_MAXR - 1 in the six-row path and r_last in the row-by-row path of
    //  steep scale factors: both must stay clear of the image's last row)
    const int r_last = min(max(syv[dy1 - dy0 - 1] + 1, 0), sh - 1);
    const bool over = over_mode == 1 || (over_mode == 2 && max(r_first + RS_MAXR - 1, r_last) < sh - 1);
    const bool fast = (sx[3] + 1 - sx[0]) < 8 && (base_x + 12 <= sw || over);
    src -= (long long)srow0 * spitch;
    if (r_last - r_first < RS_MAXR) {
        // all source rows of the block in flight together
        HQuad rows[RS_MAXR];
        // one branch for the six rows (only the last one or two quads of a row take the byte path)
        if (fast) {
#pragma unroll
            for (int k = 0; k < RS_MAXR; k++) {
                const int r = min(r_first + k, srow_hi);
                rows[k] = resize_hrow(src + (long long)r * spitch, base_x, true, sx, sel, cw, sw);
            }
        } else {
#pragma unroll
            for (int k = 0; k < RS_MAXR; k++) {
                const int r = min(r_first + k, srow_hi);
                rows[k] = resize_hrow(src + (long long)r * spitch, base_x, false, sx, sel, cw, sw);
            }
        }
#pragma unroll
        for (int j = 0; j < RS_STRIP; j++) {
            const int dy = dy0 + j;
            if (dy >= dy1) break;
            const int i0 = min(max(syv[j], 0), sh - 1) - r_first, i1 = min(max(syv[j] + 1, 0), sh - 1) - r_first;
            HQuad ra = rows[0], rb = rows[0];
            // Output row j of the strip blends source rows (i0, i0 + 1) with i0 = j or j + 1 at scale factors up
            // to 1.33 (sy advances by 1 or 2 per row): with wave-uniform rows those two cases are two scalar
            // compares; anything else (steeper factors, the clamped last row) takes the per-lane select chains.
            bool generic = true;
            if (UNI) {
                const int i0u = __builtin_amdgcn_readfirstlane(i0), i1u = __builtin_amdgcn_readfirstlane(i1);
                if (i1u == i0u + 1 && i0u == j) {
                    ra = rows[j];
                    rb = rows[j + 1];
                    generic = false;
                } else if (i1u == i0u + 1 && i0u == j + 1) {
                    ra = rows[j + 1];
                    rb = rows[j + 2];
                    generic = false;
                }
            }
            if (!UNI) {
                // rows differ between the lanes of a wave (k_resize_tail: a wave spans several strips of a small level),
                // but at scale factors up to 1.33 every lane still has i0 in {j, j+1} and i1 = i0 + 1: ONE select per
                // value between the two cases instead of the five-deep chains (the clamped last row of a level and
                // steeper factors take the chains -- a wave-uniform branch)
                const bool two = i1 == i0 + 1 && (i0 == j || i0 == j + 1);
                if (__builtin_amdgcn_ballot_w64(!two) == 0) {
                    const bool first = i0 == j;
#pragma unroll
                    for (int k = 0; k < 4; k++) {
                        ra.h[k] = first ? rows[j].h[k] : rows[j + 1].h[k];
                        rb.h[k] = first ? rows[j + 1].h[k] : rows[j + 2].h[k];
                    }
                    generic = false;
                }
            }
            if (generic) {
#pragma unroll
                for (int k = 0; k < 4; k++) {
#pragma unroll
                    for (int m = 1; m < RS_MAXR; m++) {  // register select (no dynamic indexing of the row array)
                        ra.h[k] = i0 == m ? rows[m].h[k] : ra.h[k];
                        rb.h[k] = i1 == m ? rows[m].h[k] : rb.h[k];
                    }
                }
            }
            unsigned packed = 0;
#pragma unroll
            for (int k = 0; k < 4; k++) packed |= resize_vblend(bw[j], ra.h[k], rb.h[k]) << (8 * k);
            // pitches are multiples of 16: the tail dword stays in-row
            if (!BAND || (dy >= own0 && dy < own1)) *reinterpret_cast<unsigned*>(dstg + (long long)dy * gpitch + dx0) = packed;
            if (BAND && dstl) *reinterpret_cast<unsigned*>(dstl + (dy - lrow0) * lpitch + dx0) = packed;
        }
        return;
    }
#pragma unroll
    for (int j = 0; j < RS_STRIP; j++) {  // steep scale factors: row by row
        const int dy = dy0 + j;
        if (dy >= dy1) break;
        const int sy0 = min(max(syv[j], 0), sh - 1), sy1 = min(max(syv[j] + 1, 0), sh - 1);
        const HQuad r0 = resize_hrow(src + (long long)sy0 * spitch, base_x, fast, sx, sel, cw, sw);
        const HQuad r1 = resize_hrow(src + (long long)sy1 * spitch, base_x, fast, sx, sel, cw, sw);
        unsigned packed = 0;
#pragma unroll
        for (int k = 0; k < 4; k++) packed |= resize_vblend(bw[j], r0.h[k], r1.h[k]) << (8 * k);
        if (!BAND || (dy >= own0 && dy < own1)) *reinterpret_cast<unsigned*>(dstg + (long long)dy * gpitch + dx0) = packed;
        if (BAND && dstl) *reinterpret_cast<unsigned*>(dstl + (dy - lrow0) * lpitch + dx0) = packed;
    }
}

// The first pyramid launch of an extraction also clears the per-(image, level) candidate counters FAST appends to
// (one word each, a 128-byte line apart): a separate 5-us fill kernel per batch otherwise.
__device__ __forceinline__ void zero_candidate_counters(int* __restrict__ zero_cnt, int img, int nlevels, int tid)
{
    if (zero_cnt && tid < nlevels) zero_cnt[(img * nlevels + tid) * GFO_CNT_STRIDE] = 0;
}

// Whole-plane form: level `level` of image `img` from the plane below it in HBM.
template <bool UNI>
__device__ __forceinline__ void resize_plane_block(const GfoGeom& g, const GfoInput& in, uint8_t* __restrict__ pyr, int level,
                                                   int img, int quad, int strip, const int2* __restrict__ xtab_all,
                                                   const int2* __restrict__ ytab_all)
{
    const GfoLevel& L = g.lv[level];
    int spitch;
    const uint8_t* src = gfo_level_ptr(g, in, pyr, level - 1, img, &spitch);
    const int sh = g.lv[level - 1].h, sw = g.lv[level - 1].w;
    uint8_t* dst = pyr + (long long)img * g.pyr_img_stride + L.plane_off;
    resize_block<UNI, false>(src, spitch, 0, sh - 1, sh, sw, dst, L.pitch, 0, L.h, nullptr, 0, 0, L.h, quad, strip * RS_STRIP,
                             xtab_all + L.xtab_off, ytab_all + L.ytab_off, level - 1 == 0 ? 2 : 1);
}

// register budget of the per-level kernel (the 1080p path: its levels do not fit the banded form): 8 waves per SIMD =
// 64 registers instead of 73; 55.9k -> 58.8k frames/s for extract-only 1080p (same-box A/B), 7 gives half of that
#ifndef GFO_RESIZE_WAVES
#define GFO_RESIZE_WAVES 8
#endif
#if GFO_RESIZE_WAVES > 0
#define RESIZE_OCC_ATTR __attribute__((amdgpu_waves_per_eu(GFO_RESIZE_WAVES, GFO_RESIZE_WAVES)))
#else
#define RESIZE_OCC_ATTR
#endif
__global__ __launch_bounds__(256) RESIZE_OCC_ATTR void k_resize(const GfoGeom* __restrict__ gp, GfoInput in, uint8_t* __restrict__ pyr,
                                                int level, const int2* __restrict__ xtab_all,
                                                const int2* __restrict__ ytab_all, int* __restrict__ zero_cnt)
{
    const GfoGeom& g = *gp;
    if (blockIdx.x == 0) zero_candidate_counters(zero_cnt, blockIdx.y, g.nlevels, threadIdx.x);
    const GfoLevel& L = g.lv[level];
    const int quads = (L.w + 3) >> 2;
    const int strips = (L.h + RS_STRIP - 1) / RS_STRIP;
    // one wave = up to 64 quads of ONE strip, so the row tables are wave-uniform
    const int wps = (quads + 63) >> 6;
    const int wv = __builtin_amdgcn_readfirstlane((blockIdx.x * 256 + threadIdx.x) >> 6);
    const int strip = wv / wps;
    const int quad = (wv - strip * wps) * 64 + (threadIdx.x & 63);
    if (strip >= strips || quad >= quads) return;
    resize_plane_block<true>(g, in, pyr, level, blockIdx.y, quad, strip, xtab_all, ytab_all);
}

// The small top levels of the pyramid are launch-latency bound as separate kernels (each is a dependent
// launch that cannot fill the chip).  One 1024-thread workgroup per image computes levels
// [level_begin, nlevels) back to back: a workgroup barrier orders level l's stores before level l+1's loads
// (same CU, same L1/L2 path).
#ifndef GFO_TAIL_WAVES
#define GFO_TAIL_WAVES 0
#endif
#if GFO_TAIL_WAVES > 0
#define TAIL_OCC_ATTR __attribute__((amdgpu_waves_per_eu(GFO_TAIL_WAVES, GFO_TAIL_WAVES)))
#else
#define TAIL_OCC_ATTR
#endif
__global__ __launch_bounds__(1024) TAIL_OCC_ATTR void k_resize_tail(const GfoGeom* __restrict__ gp, GfoInput in, uint8_t* __restrict__ pyr,
                                                      int level_begin, const int2* __restrict__ xtab_all,
                                                      const int2* __restrict__ ytab_all, int* __restrict__ zero_cnt)
{
    const GfoGeom& g = *gp;
    const int img = blockIdx.x;
    zero_candidate_counters(zero_cnt, img, g.nlevels, threadIdx.x);
    for (int level = level_begin; level < g.nlevels; level++) {
        const GfoLevel& L = g.lv[level];
        const int quads = (L.w + 3) >> 2;
        const int strips = (L.h + RS_STRIP - 1) / RS_STRIP;
        const int ntask = quads * strips;
        for (int t = threadIdx.x; t < ntask; t += (int)blockDim.x) {
            const int strip = t / quads;
            resize_plane_block<false>(g, in, pyr, level, img, t - strip * quads, strip, xtab_all, ytab_all);
        }
        __threadfence_block();
        __syncthreads();
    }
}

// Banded form (large batches): a workgroup owns a horizontal band of the image through a GROUP of consecutive
// levels [lb, le).  It computes the band of level lb from the plane below it in HBM, keeps it in LDS, computes
// the band of level lb+1 from that, and so on: inside a group a level is read from HBM zero times and written
// once, and the group is one launch.  (A band must also compute the rows the level above needs beyond its own
// share; that halo grows by 1 + 1.2x per level, which is why the pyramid is split into two groups rather than
// run as one chain of seven.)
// band_tab[band][level] = {c0, c1, o0, o1}: rows [c0,c1) are computed, rows [o0,o1) are written to HBM.
// register budget of the banded kernel (DESIGN.md "footprint"): 6 waves per SIMD = 80 registers is faster alone
// (124 -> 116 us for the two launches) and under overlap (+1.8 % pipeline throughput); 7 and 8 spill (140 / 172 us)
#ifndef GFO_BANDS_WAVES
#define GFO_BANDS_WAVES 6
#endif
#if GFO_BANDS_WAVES > 0
#define BANDS_OCC_ATTR __attribute__((amdgpu_waves_per_eu(GFO_BANDS_WAVES, GFO_BANDS_WAVES)))
#else
#define BANDS_OCC_ATTR
#endif
__global__ __launch_bounds__(1024) BANDS_OCC_ATTR void k_pyramid_bands(const GfoGeom* __restrict__ gp, GfoInput in, uint8_t* __restrict__ pyr,
                                                        const int2* __restrict__ xtab_all, const int2* __restrict__ ytab_all,
                                                        const int4* __restrict__ band_tab, int lb, int le, int* __restrict__ zero_cnt)
{
    extern __shared__ __align__(16) uint8_t band_lds[];
    const GfoGeom& g = *gp;
    const int band = blockIdx.x, img = blockIdx.y, nl = g.nlevels;
    if (band == 0) zero_candidate_counters(zero_cnt, img, nl, threadIdx.x);
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
    const int nwaves = blockDim.x >> 6;
    int4 below = make_int4(0, 0, 0, 0);
    for (int level = lb; level < le; level++) {
        const GfoLevel& L = g.lv[level];
        const int4 rg = band_tab[band * nl + level];
        const int quads = (L.w + 3) >> 2, wps = (quads + 63) >> 6;
        const int nstrips = (rg.y - rg.x + RS_STRIP - 1) / RS_STRIP;
        uint8_t* dstg = pyr + (long long)img * g.pyr_img_stride + L.plane_off;
        uint8_t* dstl = level < le - 1 ? band_lds + g.band_lds_off[level] : nullptr;
        const int lp = g.band_lp[level];
        const int2* xt = xtab_all + L.xtab_off;
        const int2* yt = ytab_all + L.ytab_off;
        const int sh = g.lv[level - 1].h, sw = g.lv[level - 1].w;
        const unsigned inv_wps = (65536u + wps - 1) / wps;   // wt / wps by multiplication: exact while wt * wps < 2^16
        for (int wt = wave; wt < wps * nstrips; wt += nwaves) {
            const int s = (int)(((unsigned)wt * inv_wps) >> 16);
            const int quad = (wt - s * wps) * 64 + lane;
            if (quad >= quads) continue;
            const int dy0 = rg.x + s * RS_STRIP;
            if (level == lb) {
                int spitch;
                const uint8_t* src = gfo_level_ptr(g, in, pyr, lb - 1, img, &spitch);
                resize_block<true, true>(src, spitch, 0, sh - 1, sh, sw, dstg, L.pitch, rg.z, rg.w, dstl, lp, rg.x, rg.y, quad,
                                         dy0, xt, yt, lb - 1 == 0 ? 2 : 1);
            } else {
                const uint8_t* src = band_lds + g.band_lds_off[level - 1];
                resize_block<true, true>(src, g.band_lp[level - 1], below.x, below.y - 1, sh, sw, dstg, L.pitch, rg.z, rg.w, dstl,
                                         lp, rg.x, rg.y, quad, dy0, xt, yt, 0);
            }
        }
        below = rg;
        __syncthreads();
    }
}

void gfo_launch_resize(gfo_ctx* c, const GfoInput& in, int level, int nimg)
{
    const GfoLevel& L = c->g.lv[level];
    const int quads = (L.w + 3) / 4, strips = (L.h + RS_STRIP - 1) / RS_STRIP;
    const int waves = ((quads + 63) / 64) * strips;
    dim3 grid((waves + 3) / 4, nimg);
    gfo_prof_begin(c, ST_RESIZE);
    GFO_LAUNCH(c, k_resize, grid, dim3(256), 0, c->stream, c->d_geom, in, c->d_pyr, level,
                       reinterpret_cast<const int2*>(c->d_xofs), reinterpret_cast<const int2*>(c->d_yofs), gfo_take_zero_cnt(c));
    gfo_prof_end(c);
}

void gfo_launch_resize_tail(gfo_ctx* c, const GfoInput& in, int level_begin, int nimg)
{
    gfo_prof_begin(c, ST_RESIZE);
    static const int tt_env = getenv("GFO_TAIL_THREADS") ? atoi(getenv("GFO_TAIL_THREADS")) : 0;
    const int tail_threads = tt_env >= 64 && tt_env <= 1024 && (tt_env & 63) == 0 ? tt_env : 1024;
    GFO_LAUNCH(c, k_resize_tail, dim3(nimg), dim3(tail_threads), 0, c->stream, c->d_geom, in, c->d_pyr, level_begin,
                       reinterpret_cast<const int2*>(c->d_xofs), reinterpret_cast<const int2*>(c->d_yofs), gfo_take_zero_cnt(c));
    gfo_prof_end(c);
}

void gfo_launch_pyramid_bands(gfo_ctx* c, const GfoInput& in, int nimg)
{
    for (int k = 0; k < c->n_band_groups; k++) {
        const GfoBandGroup& bg = c->band_groups[k];
        if (bg.nb == 0) {   // a level whose neighbours do not fit a band with it
            gfo_launch_resize(c, in, bg.lb, nimg);
            continue;
        }
        gfo_prof_begin(c, ST_RESIZE);
        // 256 threads when the launch has workgroups to spare (more of them resident per CU), 512 for a handful of
        // images, where the time of ONE workgroup is what counts
        const bool bt_ok = c->band_threads >= 64 && c->band_threads <= 1024 && (c->band_threads & 63) == 0;
        const int threads = bt_ok ? c->band_threads : (bg.nb * nimg >= 2048 ? 256 : 512);
        GFO_LAUNCH(c, k_pyramid_bands, dim3(bg.nb, nimg), dim3(threads), bg.lds_bytes, c->stream, c->d_geom, in,
                           c->d_pyr, reinterpret_cast<const int2*>(c->d_xofs), reinterpret_cast<const int2*>(c->d_yofs),
                           reinterpret_cast<const int4*>(c->d_band) + bg.tab_off, bg.lb, bg.le, gfo_take_zero_cnt(c));
        gfo_prof_end(c);
    }
}

int gfo_pyramid_bands_prepare(int lds_bytes)
{
    if (lds_bytes <= 64 * 1024) return 0;
    return (int)hipFuncSetAttribute(reinterpret_cast<const void*>(k_pyramid_bands), hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes);
}

// Every __global__ of this translation unit, for gfo_preload_kernels (gfo_api.hip): the runtime loads a code object and
// registers a kernel lazily, on the first launch that needs it; gfo_ctx_create resolves them all once per device under a
// mutex so that no two host threads ever race through that first-launch path (round 3: eight threads, first k_pack_results).
void gfo_kernels_pyramid(std::vector<const void*>& v)
{
    v.push_back((const void*)k_resize); v.push_back((const void*)k_resize_tail); v.push_back((const void*)k_pyramid_bands);
}
