// k_pyramid.hip -- ComputePyramid (ORBextractor.cc:1176-1201): level l = cv::resize(level l-1,
// INTER_LINEAR) in OpenCV's 11-bit fixed point.  The coefficient tables (xofs/alpha, yofs/beta)
// are built on the host in double exactly as cv::resize builds them (gfo_api.hip, plan_geometry).
//
// HBM-bound byte work: one thread produces 4 horizontally adjacent output pixels (one dword
// store); the two source rows it reads are shared by neighbouring lanes through L1.
// The 19-px reflect frame of the reference is never materialised: nothing on the extraction
// path reads it (gfo_pyramid_level rebuilds it on request).
#include "gfo_internal.h"

__global__ __launch_bounds__(256) void k_resize(const GfoGeom* __restrict__ gp, GfoInput in, uint8_t* __restrict__ pyr,
                                                int level, const int* __restrict__ xofs_all,
                                                const short* __restrict__ xcoef_all, const int* __restrict__ yofs_all,
                                                const short* __restrict__ ycoef_all)
{
    const GfoGeom& g = *gp;
    const GfoLevel& L = g.lv[level];
    const int img = blockIdx.z;
    const int dy = blockIdx.y;
    const int dx0 = (blockIdx.x * blockDim.x + threadIdx.x) * 4;
    if (dx0 >= L.w) return;
    int spitch;
    const uint8_t* src = gfo_level_ptr(g, in, pyr, level - 1, img, &spitch);
    const int sh = g.lv[level - 1].h, sw = g.lv[level - 1].w;
    uint8_t* dst = pyr + (long long)img * g.pyr_img_stride + L.plane_off + (long long)dy * L.pitch;

    const int* xofs = xofs_all + L.xtab_off;
    const short* xcoef = xcoef_all + 2 * L.xtab_off;
    const int sy = yofs_all[L.ytab_off + dy];
    const int b0 = ycoef_all[2 * (L.ytab_off + dy)], b1 = ycoef_all[2 * (L.ytab_off + dy) + 1];
    const int sy0 = min(max(sy, 0), sh - 1), sy1 = min(max(sy + 1, 0), sh - 1);
    const uint8_t* S0 = src + (long long)sy0 * spitch;
    const uint8_t* S1 = src + (long long)sy1 * spitch;
    uint32_t packed = 0;
#pragma unroll
    for (int k = 0; k < 4; k++) {
        const int dx = min(dx0 + k, L.w - 1);
        const int sx = xofs[dx];
        const int sx1 = min(sx + 1, sw - 1);
        const int a0 = xcoef[2 * dx], a1 = xcoef[2 * dx + 1];
        const int r0 = S0[sx] * a0 + S0[sx1] * a1;
        const int r1 = S1[sx] * a0 + S1[sx1] * a1;
        const int v = (((b0 * (r0 >> 4)) >> 16) + ((b1 * (r1 >> 4)) >> 16) + 2) >> 2;
        packed |= (uint32_t)(v & 255) << (8 * k);
    }
    *reinterpret_cast<uint32_t*>(dst + dx0) = packed;  // pitch is a multiple of 64: the tail dword stays in-row
}

void gfo_launch_resize(gfo_ctx* c, const GfoInput& in, int level, int nimg)
{
    const GfoLevel& L = c->g.lv[level];
    dim3 block(256);
    dim3 grid((L.w + 4 * 256 - 1) / (4 * 256), L.h, nimg);
    gfo_prof_begin(c, ST_RESIZE);
    hipLaunchKernelGGL(k_resize, grid, block, 0, c->stream, c->d_geom, in, c->d_pyr, level, c->d_xofs, c->d_xcoef,
                       c->d_yofs, c->d_ycoef);
    gfo_prof_end(c);
}
