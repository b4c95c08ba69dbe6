// k_pyramid.hip -- ComputePyramid (ORBextractor.cc:1176-1201): level l = cv::resize(level l-1,
// INTER_LINEAR) in OpenCV's 11-bit fixed point.  The coefficient tables (xofs/alpha, yofs/beta)
// are built on the host in double exactly as cv::resize builds them (gfo_api.hip, plan()).
//
// HBM-bound byte work.  A thread owns 4 adjacent output columns and walks down a strip of 16 output
// rows: the column tables are read once per thread, each source row segment is fetched as three
// coalesced dwords (12 bytes cover the <= 11-byte footprint of 4 outputs for scale factors up to 2),
// and the horizontal pass of a source row is reused when the next output row needs the same row
// (sy advances by ~1.2 per output row, so ~1.2 source rows are filtered per output row, not 2).
// One launch per level (level l needs all of level l-1); all images of the batch per launch.
// The 19-px reflect frame of the reference is never materialised: nothing on the extraction
// path reads it (gfo_pyramid_level rebuilds it on request).
#include "gfo_internal.h"

#define RS_STRIP 16

struct HQuad {
    int h[4];
};

__device__ __forceinline__ HQuad resize_hrow(const uint8_t* __restrict__ row, int base_x, bool fast, const int* sx,
                                             const int* a0, const int* a1, int sw)
{
    HQuad q;
    if (fast) {
        const unsigned w0 = *reinterpret_cast<const unsigned*>(row + base_x);
        const unsigned w1 = *reinterpret_cast<const unsigned*>(row + base_x + 4);
        const unsigned w2 = *reinterpret_cast<const unsigned*>(row + base_x + 8);
#pragma unroll
        for (int k = 0; k < 4; k++) {
            const int off = sx[k] - base_x;  // 0..10
            const unsigned lo = off < 4 ? w0 : (off < 8 ? w1 : w2);
            const unsigned hi = off < 4 ? w1 : (off < 8 ? w2 : 0u);
            const unsigned two = __builtin_amdgcn_alignbyte(hi, lo, off & 3);
            q.h[k] = (int)(two & 255u) * a0[k] + (int)((two >> 8) & 255u) * a1[k];
        }
    } else {
#pragma unroll
        for (int k = 0; k < 4; k++) {
            const int s1 = min(sx[k] + 1, sw - 1);
            q.h[k] = (int)row[sx[k]] * a0[k] + (int)row[s1] * a1[k];
        }
    }
    return q;
}

__global__ __launch_bounds__(256) void k_resize(const GfoGeom* __restrict__ gp, GfoInput in, uint8_t* __restrict__ pyr,
                                                int level, const int* __restrict__ xofs_all,
                                                const short* __restrict__ xcoef_all, const int* __restrict__ yofs_all,
                                                const short* __restrict__ ycoef_all)
{
    const GfoGeom& g = *gp;
    const GfoLevel& L = g.lv[level];
    const int img = blockIdx.y;
    const int quads = (L.w + 3) >> 2;
    const int strips = (L.h + RS_STRIP - 1) / RS_STRIP;
    // one wave = up to 64 quads of ONE strip, so the row tables and the row-reuse branch are wave-uniform
    const int wps = (quads + 63) >> 6;
    const int wv = __builtin_amdgcn_readfirstlane((blockIdx.x * 256 + threadIdx.x) >> 6);
    const int strip = wv / wps;
    const int quad = (wv - strip * wps) * 64 + (threadIdx.x & 63);
    if (strip >= strips || quad >= quads) return;
    int spitch;
    const uint8_t* src = gfo_level_ptr(g, in, pyr, level - 1, img, &spitch);
    const int sh = g.lv[level - 1].h, sw = g.lv[level - 1].w;
    uint8_t* dst = pyr + (long long)img * g.pyr_img_stride + L.plane_off;

    const int* xofs = xofs_all + L.xtab_off;
    const short* xcoef = xcoef_all + 2 * L.xtab_off;
    const int dx0 = quad * 4;
    int sx[4], a0[4], a1[4];
#pragma unroll
    for (int k = 0; k < 4; k++) {
        const int dx = min(dx0 + k, L.w - 1);
        sx[k] = xofs[dx];
        a0[k] = xcoef[2 * dx];
        a1[k] = xcoef[2 * dx + 1];
    }
    const int base_x = sx[0] & ~3;
    // three dwords must cover sx[3]+1 and stay inside the source row
    const bool fast = (sx[3] + 1 - base_x) < 12 && base_x + 12 <= sw && sx[1] >= sx[0] && sx[2] >= sx[0] && sx[3] >= sx[0];
    const int dy0 = strip * RS_STRIP, dy1 = min(dy0 + RS_STRIP, L.h);
    int prev_row = -1;
    HQuad prev;
    prev.h[0] = prev.h[1] = prev.h[2] = prev.h[3] = 0;
    for (int dy = dy0; dy < dy1; dy++) {
        const int sy = yofs_all[L.ytab_off + dy];
        const int b0 = ycoef_all[2 * (L.ytab_off + dy)], b1 = ycoef_all[2 * (L.ytab_off + dy) + 1];
        const int sy0 = min(max(sy, 0), sh - 1), sy1 = min(max(sy + 1, 0), sh - 1);
        const HQuad r0 = sy0 == prev_row ? prev : resize_hrow(src + (long long)sy0 * spitch, base_x, fast, sx, a0, a1, sw);
        const HQuad r1 = sy1 == sy0 ? r0 : resize_hrow(src + (long long)sy1 * spitch, base_x, fast, sx, a0, a1, sw);
        unsigned packed = 0;
#pragma unroll
        for (int k = 0; k < 4; k++) {
            const int v = (((b0 * (r0.h[k] >> 4)) >> 16) + ((b1 * (r1.h[k] >> 4)) >> 16) + 2) >> 2;
            packed |= (unsigned)(v & 255) << (8 * k);
        }
        *reinterpret_cast<unsigned*>(dst + (long long)dy * L.pitch + dx0) = packed;  // pitch multiple of 64: tail dword stays in-row
        prev = r1;
        prev_row = sy1;
    }
}

void gfo_launch_resize(gfo_ctx* c, const GfoInput& in, int level, int nimg)
{
    const GfoLevel& L = c->g.lv[level];
    const int quads = (L.w + 3) / 4, strips = (L.h + RS_STRIP - 1) / RS_STRIP;
    const int waves = ((quads + 63) / 64) * strips;
    dim3 grid((waves + 3) / 4, nimg);
    gfo_prof_begin(c, ST_RESIZE);
    hipLaunchKernelGGL(k_resize, grid, dim3(256), 0, c->stream, c->d_geom, in, c->d_pyr, level, c->d_xofs, c->d_xcoef,
                       c->d_yofs, c->d_ycoef);
    gfo_prof_end(c);
}
