// k_blur.hip -- cv::GaussianBlur(level.clone(), 7x7, sigma 2, BORDER_REFLECT_101) for CV_8UC1
// (ORBextractor.cc:1154-1155).  Integer taps {18,34,49,55,49,34,18} (sum 257), exact
// accumulation, one rounding: min(255, (sum + 2^15) >> 16).  See DESIGN.md "blur".
//
// Streaming form, no LDS and no barriers: a thread owns 4 adjacent columns and walks down a
// GFO_BLUR_STRIP-row (24) strip.  Per row it loads the 12 bytes [x-4, x+8) as three coalesced dwords (the lanes of a
// wave read one contiguous 256-B run three times, shifted by 4 B: served by L1), forms the four
// horizontal sums with two v_dot4_u32_u8 each (no byte unpacking; max 257*255 = 65535 fits u16 exactly),
// keeps the sums of the last four row PAIRS in registers (a column's even and odd row share a register: the vertical pass is
// then four v_dot2_u32_u16 per output instead of seven), and emits two rows of output per step.
// Vertical halo: 6 extra rows per strip (32-row strips measured 4 % slower alone, 16-row ones 2 % slower in the pipeline).
// All levels of all images are one launch (block index -> level through the prefix table).
#include "gfo_internal.h"
#include <stdlib.h>

#include "k_blur_dev.inc"

// One launch: the first blur_total_b blocks of every image are its border blocks (long, thin chains of
// scattered rows -- dispatched first so they run underneath the streaming bulk), the rest the interior.
// Register budget.  The kernel alone is indifferent (6 waves per SIMD at 77 registers, 8 at 64 with a 16-byte spill:
// 109 vs 105 us), but it shares the chip with the quadtree of its own batch and with the other contexts' kernels, and
// there every register it does not hold is a wave of somebody else: 194.0k -> 201.3k frames/s for the whole pipeline
// (same-box A/B, tools/ab_variant.sh; DESIGN.md "footprint").
#ifndef GFO_BLUR_WAVES
#define GFO_BLUR_WAVES 6
#endif
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(GFO_BLUR_WAVES, GFO_BLUR_WAVES))) void k_blur(const GfoGeom* __restrict__ gp, GfoInput in,
                                              const uint8_t* __restrict__ pyr, uint8_t* __restrict__ blur, int xcd8, int nimg)
{
    const GfoGeom& g = *gp;
    const int nb = g.blur_total_b;
    // xcd8: grid (8 * blocks, ceil(images / 8)), blockIdx.x & 7 picks the image inside a group of eight -- all blocks of an image
    // on ONE XCD (as in k_fast), so that the three halo rows a strip shares with the strips above and below it are fetched through
    // the fabric once instead of by two XCDs
    const int img = xcd8 ? (int)(blockIdx.y * 8 + (blockIdx.x & 7)) : (int)blockIdx.y;
    if (img >= nimg) return;
    const int bx = xcd8 ? (int)(blockIdx.x >> 3) : (int)blockIdx.x;
    if (bx < nb)
        blur_body<true>(g, in, pyr, blur, bx, img, threadIdx.x);
    else
        blur_body<false>(g, in, pyr, blur, bx - nb, img, threadIdx.x);
}

void gfo_launch_blur(gfo_ctx* c, const GfoInput& in, int nimg)
{
    gfo_prof_begin(c, ST_BLUR);
    if (c->g.total_tiles + c->g.blur_total_b > 0) {
        static const int xcd_env = getenv("GFO_BLUR_XCD") ? atoi(getenv("GFO_BLUR_XCD")) : 1;
        const int xcd8 = xcd_env && nimg >= 8 ? 1 : 0;
        const unsigned blocks = (unsigned)(c->g.total_tiles + c->g.blur_total_b);
        const dim3 grid = xcd8 ? dim3(blocks * 8u, (unsigned)(nimg + 7) / 8u) : dim3(blocks, (unsigned)nimg);
        GFO_LAUNCH(c, k_blur, grid, dim3(256), 0, c->stream, c->d_geom, in, c->d_pyr, c->d_blur, xcd8, nimg);
    }
    gfo_prof_end(c);
}

// Every __global__ of this translation unit, for gfo_preload_kernels (gfo_api.hip): the runtime loads a code object and
// registers a kernel lazily, on the first launch that needs it; gfo_ctx_create resolves them all once per device under a
// mutex so that no two host threads ever race through that first-launch path (round 3: eight threads, first k_pack_results).
void gfo_kernels_blur(std::vector<const void*>& v) { v.push_back((const void*)k_blur); }
