// k_blur.hip -- cv::GaussianBlur(level.clone(), 7x7, sigma 2, BORDER_REFLECT_101) for CV_8UC1
// (ORBextractor.cc:1154-1155).  Integer taps {18,34,49,55,49,34,18} (sum 257), exact
// accumulation, one rounding: min(255, (sum + 2^15) >> 16).  See DESIGN.md "blur".
//
// Streaming form, no LDS and no barriers: a thread owns 4 adjacent columns and walks down a
// 32-row strip.  Per row it loads the 12 bytes [x-4, x+8) as three coalesced dwords (the lanes of a
// wave read one contiguous 256-B run three times, shifted by 4 B: served by L1), forms the four
// horizontal sums in packed u16 (max 257*255 = 65535 fits exactly), keeps the last seven rows of
// sums in registers, and emits one dword of output per row.  Vertical halo: 6 extra rows per 32.
// All levels of all images are one launch (block index -> level through the prefix table).
#include "gfo_internal.h"

#define BLUR_STRIP 32

__device__ __forceinline__ int gfo_reflect101(int p, int n)
{
    if (p < 0) p = -p;
    if (p >= n) p = 2 * n - 2 - p;
    return min(max(p, 0), n - 1);
}

typedef unsigned short __attribute__((ext_vector_type(2))) us2;

struct HRow {
    unsigned lo, hi;  // four u16 horizontal sums: (x0, x0+1), (x0+2, x0+3)
};

__device__ __forceinline__ unsigned hsum2(unsigned a06, unsigned a15, unsigned a24, unsigned a3)
{
    // packed u16: 18*(p0+p6) + 34*(p1+p5) + 49*(p2+p4) + 55*p3, two pixels per register
    const unsigned k18 = 18u | (18u << 16), k34 = 34u | (34u << 16), k49 = 49u | (49u << 16), k55 = 55u | (55u << 16);
    us2 acc = __builtin_bit_cast(us2, a3) * __builtin_bit_cast(us2, k55);
    acc += __builtin_bit_cast(us2, a06) * __builtin_bit_cast(us2, k18);
    acc += __builtin_bit_cast(us2, a15) * __builtin_bit_cast(us2, k34);
    acc += __builtin_bit_cast(us2, a24) * __builtin_bit_cast(us2, k49);
    return __builtin_bit_cast(unsigned, acc);
}

__device__ __forceinline__ unsigned pk(unsigned a, unsigned b) { return a | (b << 16); }

// horizontal pass for 4 pixels from the 10 source pixels p[0..9] = columns x0-3 .. x0+6
__device__ __forceinline__ HRow hpass(const unsigned* p)
{
    HRow r;
    // outputs 0,1 use p[0..6], p[1..7]; outputs 2,3 use p[2..8], p[3..9]
    r.lo = hsum2(pk(p[0] + p[6], p[1] + p[7]), pk(p[1] + p[5], p[2] + p[6]), pk(p[2] + p[4], p[3] + p[5]), pk(p[3], p[4]));
    r.hi = hsum2(pk(p[2] + p[8], p[3] + p[9]), pk(p[3] + p[7], p[4] + p[8]), pk(p[4] + p[6], p[5] + p[7]), pk(p[5], p[6]));
    return r;
}

__device__ __forceinline__ HRow load_hrow(const uint8_t* __restrict__ row, int x0, int w, bool interior)
{
    unsigned p[10];
    if (interior) {
        const unsigned d0 = *reinterpret_cast<const unsigned*>(row + x0 - 4);
        const unsigned d1 = *reinterpret_cast<const unsigned*>(row + x0);
        const unsigned d2 = *reinterpret_cast<const unsigned*>(row + x0 + 4);
        p[0] = (d0 >> 8) & 255; p[1] = (d0 >> 16) & 255; p[2] = d0 >> 24;
        p[3] = d1 & 255; p[4] = (d1 >> 8) & 255; p[5] = (d1 >> 16) & 255; p[6] = d1 >> 24;
        p[7] = d2 & 255; p[8] = (d2 >> 8) & 255; p[9] = (d2 >> 16) & 255;
    } else {
#pragma unroll
        for (int k = 0; k < 10; k++) p[k] = row[gfo_reflect101(x0 - 3 + k, w)];
    }
    return hpass(p);
}

__device__ __forceinline__ unsigned vout(unsigned a06, unsigned a15, unsigned a24, unsigned a3)
{
    const unsigned acc = 18u * a06 + 34u * a15 + 49u * a24 + 55u * a3;
    return min((acc + 32768u) >> 16, 255u);
}

__global__ __launch_bounds__(256) void k_blur(const GfoGeom* __restrict__ gp, GfoInput in,
                                              const uint8_t* __restrict__ pyr, uint8_t* __restrict__ blur)
{
    const GfoGeom& g = *gp;
    const int blk = blockIdx.x, img = blockIdx.y;
    int level = 0;
    while (level + 1 < g.nlevels && blk >= g.lv[level + 1].tile_base) level++;
    const GfoLevel& L = g.lv[level];
    const int t = (blk - L.tile_base) * 256 + threadIdx.x;
    const int quads = L.tiles_x;  // ceil(w/4)
    const int strip = t / quads, quad = t - strip * quads;
    if (strip >= L.tiles_y) return;
    const int x0 = quad * 4, y0 = strip * BLUR_STRIP;
    const int w = L.w, h = L.h;
    int pitch;
    const uint8_t* src = gfo_level_ptr(g, in, pyr, level, img, &pitch);
    uint8_t* dst = blur + (long long)img * g.blur_img_stride + L.blur_off;
    const bool interior = x0 >= 4 && x0 + 8 <= w;  // the three dwords stay inside the row
    const int y1 = min(y0 + BLUR_STRIP, h);

    HRow r0, r1, r2, r3, r4, r5, r6;
    r0 = load_hrow(src + (long long)gfo_reflect101(y0 - 3, h) * pitch, x0, w, interior);
    r1 = load_hrow(src + (long long)gfo_reflect101(y0 - 2, h) * pitch, x0, w, interior);
    r2 = load_hrow(src + (long long)gfo_reflect101(y0 - 1, h) * pitch, x0, w, interior);
    r3 = load_hrow(src + (long long)y0 * pitch, x0, w, interior);
    r4 = load_hrow(src + (long long)gfo_reflect101(y0 + 1, h) * pitch, x0, w, interior);
    r5 = load_hrow(src + (long long)gfo_reflect101(y0 + 2, h) * pitch, x0, w, interior);
    for (int y = y0; y < y1; y++) {
        r6 = load_hrow(src + (long long)gfo_reflect101(y + 3, h) * pitch, x0, w, interior);
        // vertical pass on the four columns (u32 accumulation: 257 * 65535 < 2^32)
        const unsigned o0 = vout((r0.lo & 0xFFFF) + (r6.lo & 0xFFFF), (r1.lo & 0xFFFF) + (r5.lo & 0xFFFF), (r2.lo & 0xFFFF) + (r4.lo & 0xFFFF), r3.lo & 0xFFFF);
        const unsigned o1 = vout((r0.lo >> 16) + (r6.lo >> 16), (r1.lo >> 16) + (r5.lo >> 16), (r2.lo >> 16) + (r4.lo >> 16), r3.lo >> 16);
        const unsigned o2 = vout((r0.hi & 0xFFFF) + (r6.hi & 0xFFFF), (r1.hi & 0xFFFF) + (r5.hi & 0xFFFF), (r2.hi & 0xFFFF) + (r4.hi & 0xFFFF), r3.hi & 0xFFFF);
        const unsigned o3 = vout((r0.hi >> 16) + (r6.hi >> 16), (r1.hi >> 16) + (r5.hi >> 16), (r2.hi >> 16) + (r4.hi >> 16), r3.hi >> 16);
        *reinterpret_cast<unsigned*>(dst + (long long)y * L.pitch + x0) = o0 | (o1 << 8) | (o2 << 16) | (o3 << 24);
        r0 = r1; r1 = r2; r2 = r3; r3 = r4; r4 = r5; r5 = r6;
    }
}

void gfo_launch_blur(gfo_ctx* c, const GfoInput& in, int nimg)
{
    dim3 grid(c->g.total_tiles, nimg);
    gfo_prof_begin(c, ST_BLUR);
    hipLaunchKernelGGL(k_blur, grid, dim3(256), 0, c->stream, c->d_geom, in, c->d_pyr, c->d_blur);
    gfo_prof_end(c);
}
