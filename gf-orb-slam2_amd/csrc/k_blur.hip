// k_blur.hip -- cv::GaussianBlur(level.clone(), 7x7, sigma 2, BORDER_REFLECT_101) for CV_8UC1
// (ORBextractor.cc:1154-1155).  Integer taps {18,34,49,55,49,34,18} (sum 257), exact
// accumulation, one rounding: min(255, (sum + 2^15) >> 16).  See DESIGN.md "blur".
//
// One 256-thread workgroup produces a 64x16 output tile of one level of one image: the
// 70x22 source window is staged in LDS (reflected at the level's own border), the horizontal
// pass lands in LDS as u16 (max 257*255 = 65535), the vertical pass writes the bytes.
// All levels of all images are one launch (tile index -> level through the tile prefix table).
#include "gfo_internal.h"

#define BT_W 64
#define BT_H 16
#define BT_SRC_PITCH 72

__device__ __forceinline__ int gfo_reflect101(int p, int n)
{
    if (p < 0) p = -p;
    if (p >= n) p = 2 * n - 2 - p;
    return min(max(p, 0), n - 1);
}

__global__ __launch_bounds__(256) void k_blur(const GfoGeom* __restrict__ gp, GfoInput in,
                                              const uint8_t* __restrict__ pyr, uint8_t* __restrict__ blur)
{
    __shared__ uint8_t s_src[(BT_H + 6) * BT_SRC_PITCH];
    __shared__ uint16_t s_h[(BT_H + 6) * BT_W];
    const GfoGeom& g = *gp;
    const int tile = blockIdx.x, img = blockIdx.y, tid = threadIdx.x;
    int level = 0;
    while (level + 1 < g.nlevels && tile >= g.lv[level + 1].tile_base) level++;
    const GfoLevel& L = g.lv[level];
    const int t = tile - L.tile_base;
    const int ty = t / L.tiles_x, tx = t - ty * L.tiles_x;
    const int x0 = tx * BT_W, y0 = ty * BT_H;
    int pitch;
    const uint8_t* src = gfo_level_ptr(g, in, pyr, level, img, &pitch);

    for (int i = tid; i < (BT_H + 6) * (BT_W + 6); i += 256) {
        const int r = i / (BT_W + 6), c = i - r * (BT_W + 6);
        const int y = gfo_reflect101(y0 - 3 + r, L.h), x = gfo_reflect101(x0 - 3 + c, L.w);
        s_src[r * BT_SRC_PITCH + c] = src[(long long)y * pitch + x];
    }
    __syncthreads();
    for (int i = tid; i < (BT_H + 6) * BT_W; i += 256) {
        const int r = i >> 6, c = i & 63;
        const uint8_t* p = &s_src[r * BT_SRC_PITCH + c];
        const int acc = 18 * (p[0] + p[6]) + 34 * (p[1] + p[5]) + 49 * (p[2] + p[4]) + 55 * p[3];
        s_h[i] = (uint16_t)acc;
    }
    __syncthreads();
    const int c = tid & 63, rb = (tid >> 6) * 4;
    uint8_t* dst = blur + (long long)img * g.blur_img_stride + L.blur_off;
    if (x0 + c < L.w) {
#pragma unroll
        for (int k = 0; k < 4; k++) {
            const int r = rb + k;
            if (y0 + r >= L.h) break;
            const uint16_t* q = &s_h[r * BT_W + c];
            const unsigned acc = 18u * (q[0] + q[6 * BT_W]) + 34u * (q[BT_W] + q[5 * BT_W]) +
                                 49u * (q[2 * BT_W] + q[4 * BT_W]) + 55u * q[3 * BT_W];
            const unsigned v = (acc + 32768u) >> 16;
            dst[(long long)(y0 + r) * L.pitch + x0 + c] = (uint8_t)min(v, 255u);
        }
    }
}

void gfo_launch_blur(gfo_ctx* c, const GfoInput& in, int nimg)
{
    dim3 grid(c->g.total_tiles, nimg);
    gfo_prof_begin(c, ST_BLUR);
    hipLaunchKernelGGL(k_blur, grid, dim3(256), 0, c->stream, c->d_geom, in, c->d_pyr, c->d_blur);
    gfo_prof_end(c);
}
