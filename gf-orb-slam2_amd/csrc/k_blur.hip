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

// ------------------------------------------------------------------------------------------------------------------------------
// The blur on the matrix cores (round 6).  The streaming form above is VALU-bound (0.70 of the vector pipe at 3.2 TB/s): 18 lane
// operations a pixel.  A 7-tap pass over a run of pixels is a product with a banded matrix, and the i8 matrix instructions sum signed
// bytes exactly, so a 16 x 64 block of the image costs twelve matrix instructions and ~130 vector instructions a wavefront:
//   pass 1 (horizontal)  H = (P - 128) x T1 + 128 S      A = 16 rows x 64 pixels, 16 bytes a lane straight from memory;  B = the band
//   pass 2 (vertical)    D2[col][row] = sum_k H[k][col] x T2[k][row]: pass 1's result as the A operand, the band as B
// What makes it free of data movement (tools/c/mfma_i8_layout.hip checks both facts on the device):
//   * the K index of both operands is (lane group g, byte j) alike, so WHICH row or column a k is never matters as long as A and B
//     are built by the same rule -- the band matrices are;
//   * an accumulator tile has its column on the lane (l & 15) and rows 4 (l >> 4) + i in its four registers: exactly an A operand
//     whose m is that column and whose k are those rows.  Pass 2 takes pass 1's result where it lies (two stacked tiles = the 8 bytes
//     of a lane), and ITS result has the output row on the lane and four adjacent output columns in the registers.
// Which output column an accumulator column stands for is the band matrix's choice: tile k = 0..3 of a block computes columns
// 16 (n >> 2) + 4 k + (n & 3), so that a lane group ends with SIXTEEN ADJACENT columns of its row -- one 16-byte store a lane, 64 aligned
// bytes a row.  BORDER_REFLECT_101 is linear as well: near the left and right edge the band's taps fold onto the pixels they reflect
// to (two taps on one pixel: at most 98 < 128), so there are no border blocks; rows reflect by index arithmetic (a lane loads one row).
// H is up to 16 bits (S * 255 = 65 535 with taps summing to S = 257): its low and high bytes go through pass 2 separately, both exact;
// (V_hi << 8) + V_lo + 2^15 >> 16, clamped, is the streaming form's single rounding.  Signed bytes: every operand is offset by 128 and
// the offset's share (128 S) sits in the accumulator's start value.
// A wavefront walks BLUR_MF_TILES tile rows down one 64-column strip (plus one pass-1 tile row of halo per run).
#ifndef BLUR_MF_TILES
#define BLUR_MF_TILES 15   // 240 rows a run (one tile row of halo each): 6 / 10 / 15 measured 288.4 / 287.6 / 290.5 k frames/s, 3: 282.3 k
#endif
typedef int gfo_v4i __attribute__((ext_vector_type(4)));

// wavefront tasks of a level: 64-column strips x runs of tile rows
__host__ __device__ __forceinline__ int blur_mf_cols(int w) { return (w + 63) / 64; }
__host__ __device__ __forceinline__ int blur_mf_runs(int h) { return (h + 16 * BLUR_MF_TILES - 1) / (16 * BLUR_MF_TILES); }
__host__ __device__ __forceinline__ int blur_mf_blocks(int w, int h) { return (blur_mf_cols(w) * blur_mf_runs(h) + 3) / 4; }

// eight consecutive entries of the tap sequence (... 0 0 a b c d c b a 0 0 ...) from position o (the first tap at 0) as packed bytes
__device__ __forceinline__ unsigned long long blur_tap_window(int o)
{
    constexpr GfoGaussTaps T = gfo_gauss_taps();
    constexpr unsigned long long TP = (unsigned long long)T.a | ((unsigned long long)T.b << 8) | ((unsigned long long)T.c << 16) | ((unsigned long long)T.d << 24) |
                                      ((unsigned long long)T.c << 32) | ((unsigned long long)T.b << 40) | ((unsigned long long)T.a << 48);
    if (o >= 8 || o <= -8) return 0ull;
    return o >= 0 ? TP >> (8 * o) : TP << (-8 * o);
}

__device__ __forceinline__ void blur_body_mfma(const GfoGeom& g, const GfoInput& in, const uint8_t* __restrict__ pyr, uint8_t* __restrict__ blur,
                                               int blk, int img, int tid)
{
    constexpr GfoGaussTaps T = gfo_gauss_taps();
    constexpr int S7 = (int)(2 * (T.a + T.b + T.c) + T.d);
    constexpr unsigned TAP7[7] = {T.a, T.b, T.c, T.d, T.c, T.b, T.a};
    // signed bytes: a tap, and two taps that BORDER_REFLECT_101 folds onto one pixel (taps d and d' with d + d' = 6, 4 or 2), stay below 128
    static_assert(T.d <= 127 && 2 * T.c <= 127 && T.b + T.d <= 127 && 2 * T.b <= 127 && T.a + T.c <= 127, "folded taps must fit a signed byte");
    int level = 0, base = 0;
    for (;;) {   // (scalar: a handful of levels)
        const int nb = blur_mf_blocks(g.lv[level].w, g.lv[level].h);
        if (blk < base + nb || level + 1 >= g.nlevels) break;
        base += nb;
        level++;
    }
    const GfoLevel& L = g.lv[level];
    const int w = L.w, h = L.h;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
    const int task = (blk - base) * 4 + wave;
    const int cols = blur_mf_cols(w);
    if (task >= cols * blur_mf_runs(h)) return;
    const int run = task / cols, tc = task - run * cols;
    const int x0 = 64 * tc;
    const int r0 = run * 16 * BLUR_MF_TILES;
    const int nt = min(BLUR_MF_TILES, (h - r0 + 15) / 16);
    int pitch;
    const uint8_t* src = gfo_level_ptr(g, in, pyr, level, img, &pitch);
    uint8_t* dst = blur + (long long)img * g.blur_img_stride + L.blur_off;
    const int g4 = lane >> 4, n16 = lane & 15;
    // Tile k = 0..3 computes columns x0 + 16 k .. + 15 (accumulator column n = column x0 + 16 k + n).  Tiles 0, 1 read one 64-pixel window,
    // tiles 2, 3 another: [wp, wp + 64) holds every pixel the pair's valid outputs read (reflected ones included), never leaves the row
    // (level 0 may be the caller's image: nothing beyond w - 1 is read), and starts 16-aligned in memory where that leaves a choice
    // (x0 - 16 and x0 + 16 in the interior of a plane whose rows are 16-aligned).
    int wp[2];
#pragma unroll
    for (int q = 0; q < 2; q++) {
        const int lo = max(0, x0 + 32 * q - 3), hi = min(w - 1, x0 + 32 * q + 34);
        const int w_min = max(0, hi - 63), w_max = max(w_min, min(lo, w - 64));
        const int a = (int)((unsigned long long)(src + w_max) & 15);          // w_max - a is 16-aligned
        int ws = w_max - a;
        if (ws < w_min) ws = w_max - (a & 7);                                  // 8-aligned
        if (ws < w_min) ws = w_max;
        wp[q] = ws;
    }
#ifdef BLUR_MF_SHARE
    const bool share = wp[1] == wp[0] + 32;   // wave-uniform
#endif
    gfo_v4i B1[4];
#pragma unroll
    for (int k = 0; k < 4; k++) {
        // band matrix B1[kappa = 16 g + j][n] = sum of the taps whose pixel reflect101(x + d) is column ws + kappa
        const int ws = wp[k >> 1];
        const int x = x0 + 16 * k + n16;
        unsigned long long blo, bhi;
        if (x >= 3 && x + 3 <= w - 1) {
            const int o = ws + 16 * g4 - (x - 3);
            blo = blur_tap_window(o); bhi = blur_tap_window(o + 8);
        } else {
            blo = 0; bhi = 0;
            if (x < w) {
#pragma unroll
                for (int d = 0; d < 7; d++) {
                    const int pos = gfo_reflect101(x + d - 3, w) - ws - 16 * g4;
                    if (pos >= 0 && pos < 8) blo += (unsigned long long)TAP7[d] << (8 * pos);
                    else if (pos >= 8 && pos < 16) bhi += (unsigned long long)TAP7[d] << (8 * (pos - 8));
                }
            }
        }
        B1[k] = gfo_v4i{(int)(unsigned)blo, (int)(unsigned)(blo >> 32), (int)(unsigned)bhi, (int)(unsigned)(bhi >> 32)};
    }
    // pass 2 (any tile): bytes 0-3 tap[4 g + j - n], bytes 4-7 tap[4 g + 16 + (j - 4) - n]
    const unsigned long long B2 = (blur_tap_window(4 * g4 - n16) & 0xFFFFFFFFull) | (blur_tap_window(4 * g4 + 16 - n16) << 32);
    const gfo_v4i C1 = {128 * S7, 128 * S7, 128 * S7, 128 * S7};
    const gfo_v4i C2lo = {128 * S7 + 32768, 128 * S7 + 32768, 128 * S7 + 32768, 128 * S7 + 32768};
    struct Raw { gfo_v4i v[2]; };
    auto load_tile = [&](int t) {   // the lane's row of pass-1 tile row t: image row r0 + 16 t - 3 + (l & 15), reflected; 16 of each window's 64 columns
        const int y = gfo_reflect101(r0 + 16 * t - 3 + n16, h);
        const uint8_t* row = src + (long long)y * pitch + 16 * g4;
        Raw r;
#ifdef BLUR_MF_SHARE
        // (experiment) interior strips: the second window starts 32 bytes behind the first, so its lower half IS the first one's upper
        // half -- only the lanes of its upper half load (96 bytes a row instead of 128), the rest comes over the lanes in pass1
        if (share) {
            r.v[0] = *reinterpret_cast<const gfo_v4i*>(row + wp[0]);
            r.v[1] = r.v[0];
            if (g4 >= 2) r.v[1] = *reinterpret_cast<const gfo_v4i*>(row + wp[1]);
            return r;
        }
#endif
#pragma unroll
        for (int q = 0; q < 2; q++) {
#ifdef BLUR_MF_NOLOAD
            r.v[q] = gfo_v4i{y, wp[q], y, y};
#else
            r.v[q] = *reinterpret_cast<const gfo_v4i*>(row + wp[q]);
#endif
        }
        return r;
    };
    // pass 1 of one tile row: four matrix instructions, the 16-bit sums split into signed low and high bytes
    auto pass1 = [&](const Raw& raw, unsigned* lo, unsigned* hi) {
        gfo_v4i a[2];
#pragma unroll
        for (int q = 0; q < 2; q++)
            a[q] = gfo_v4i{raw.v[q][0] ^ (int)0x80808080, raw.v[q][1] ^ (int)0x80808080, raw.v[q][2] ^ (int)0x80808080, raw.v[q][3] ^ (int)0x80808080};
#ifdef BLUR_MF_SHARE
        if (share) {   // lanes 0-31 of the second window: what lanes 32-63 hold of the first
#pragma unroll
            for (int i = 0; i < 4; i++) {
#ifdef BLUR_MF_SHARE_LDS
                const int up = __builtin_amdgcn_ds_bpermute(4 * ((lane + 32) & 63), a[0][i]);
                a[1][i] = g4 < 2 ? up : a[1][i];
#else
                // v_permlane32_swap: the upper half of its first operand changes places with the lower half of the second -- the second
                // result is (first window's upper lanes, second window's own upper lanes)
                const auto sw = __builtin_amdgcn_permlane32_swap((unsigned)a[0][i], (unsigned)a[1][i], false, false);
                a[1][i] = (int)sw[1];
#endif
            }
        }
#endif
#pragma unroll
        for (int k = 0; k < 4; k++) {
            const gfo_v4i d = __builtin_amdgcn_mfma_i32_16x16x64_i8(a[k >> 1], B1[k], C1, 0, 0, 0);
            const unsigned x01 = __builtin_amdgcn_perm((unsigned)d[1], (unsigned)d[0], 0x05010400u);   // lo0 lo1 hi0 hi1
            const unsigned x23 = __builtin_amdgcn_perm((unsigned)d[3], (unsigned)d[2], 0x05010400u);
            lo[k] = __builtin_amdgcn_perm(x23, x01, 0x05040100u) ^ 0x80808080u;
            hi[k] = __builtin_amdgcn_perm(x23, x01, 0x07060302u) ^ 0x80808080u;
        }
    };
    unsigned hlo[4], hhi[4];   // pass-1 tile row t: rows 4 g + i of the lane's column, per 16-column tile
    Raw raw1 = load_tile(1), raw2 = load_tile(2);
    pass1(load_tile(0), hlo, hhi);
    for (int t = 0; t < nt; t++) {
        const Raw raw = raw1;
        raw1 = raw2;
        raw2 = load_tile(t + 3);   // (a row index beyond the image reflects back inside: always a valid row)
        unsigned nlo[4], nhi[4];
        pass1(raw, nlo, nhi);
        unsigned out[4];
#pragma unroll
        for (int k = 0; k < 4; k++) {
            const gfo_v4i vlo = __builtin_amdgcn_mfma_i32_16x16x32_i8((long)((unsigned long long)hlo[k] | ((unsigned long long)nlo[k] << 32)), (long)B2, C2lo, 0, 0, 0);
            const gfo_v4i vhi = __builtin_amdgcn_mfma_i32_16x16x32_i8((long)((unsigned long long)hhi[k] | ((unsigned long long)nhi[k] << 32)), (long)B2, C1, 0, 0, 0);
            unsigned o[4];
#pragma unroll
            for (int i = 0; i < 4; i++) o[i] = min((unsigned)((vhi[i] << 8) + vlo[i]) >> 16, 255u);
            out[k] = o[0] | (o[1] << 8) | (o[2] << 16) | (o[3] << 24);      // row l & 15, columns x0 + 16 k + 4 g .. + 3
            hlo[k] = nlo[k]; hhi[k] = nhi[k];
        }
        // 4 x 4 transpose over (register k, lane group g): afterwards the lane holds columns x0 + 16 g + 4 k .. + 3 in register k --
        // sixteen adjacent bytes.  Two butterfly stages of row swaps (v_permlane32_swap: the upper half of one register with the lower
        // half of the other; v_permlane16_swap: the odd 16-lane rows of one with the even rows of the other).
        {
            auto s0 = __builtin_amdgcn_permlane32_swap(out[0], out[2], false, false); out[0] = s0[0]; out[2] = s0[1];
            auto s1 = __builtin_amdgcn_permlane32_swap(out[1], out[3], false, false); out[1] = s1[0]; out[3] = s1[1];
            auto s2 = __builtin_amdgcn_permlane16_swap(out[0], out[1], false, false); out[0] = s2[0]; out[1] = s2[1];
            auto s3 = __builtin_amdgcn_permlane16_swap(out[2], out[3], false, false); out[2] = s3[0]; out[3] = s3[1];
        }
        const int y = r0 + 16 * t + n16;
#ifdef BLUR_MF_NOSTORE   // (experiments only: results are wrong by construction)
        if (y == -12345)
#else
        if (y < h)
#endif
            *reinterpret_cast<gfo_v4i*>(dst + (long long)y * L.pitch + x0 + 16 * g4) = gfo_v4i{(int)out[0], (int)out[1], (int)out[2], (int)out[3]};   // (columns from w on: the plane's padding)
    }
}

#ifndef GFO_BLUR_MF_WAVES
#define GFO_BLUR_MF_WAVES 5
#endif
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(GFO_BLUR_MF_WAVES, 8))) void k_blur_mfma(const GfoGeom* __restrict__ gp, GfoInput in, const uint8_t* __restrict__ pyr,
                                                                                                   uint8_t* __restrict__ blur, int xcd8, int nimg)
{
    const GfoGeom& g = *gp;
    const int img = xcd8 ? (int)(blockIdx.y * 8 + (blockIdx.x & 7)) : (int)blockIdx.y;
    if (img >= nimg) return;
    const int bx = xcd8 ? (int)(blockIdx.x >> 3) : (int)blockIdx.x;
#ifndef BLUR_MF_NOBODY   // (experiments only)
    blur_body_mfma(g, in, pyr, blur, bx, img, threadIdx.x);
#endif
}

// the matrix-core form needs every level to be at least one 64-pixel window wide, a blurred plane whose pitch covers whole 64-column strips
// (the last strip's store runs into the padding), and the build's single rounding (GFO_OCV_BLUR_ROUND 0; any taps of gfo_internal.h fit
// the signed-byte arithmetic: a row sum <= 257, two folded taps <= 127): otherwise the streaming form for the whole launch
static int blur_mfma_blocks(const gfo_ctx* c)
{
#if GFO_OCV_BLUR_ROUND == 1
    return 0;
#else
    // GFO_BLUR_MFMA (read per call: tests/test_gpu_blur.py runs both forms in one process): 0 = the streaming form, 1 = the matrix cores
    // whatever the image, unset = by image width.  The matrix-core form wins where an image's levels stay in its XCD's L2 between the
    // overlapping window loads (752 x 480: +5.4 % extract-only, +2.5 % with the stereo association) and loses on larger images
    // (640 x 480: +5.5 %, 848 x 480: +0.6 %, 960 x 540: -2.3 %, 1024 x 768: -2.7 %, 1280 x 720: -4.0 %, 1920 x 1080: -6 %;
    // tools/ab_blur_width.py): up to 896 px wide.
    const char* e = getenv("GFO_BLUR_MFMA");
    if (e && atoi(e) == 0) return 0;
    if (!e && c->g.lv[0].w > 896) return 0;
    int blocks = 0;
    for (int l = 0; l < c->g.nlevels; l++) {
        const int w = c->g.lv[l].w, h = c->g.lv[l].h;
        if (w < 64 || h < 1 || c->g.lv[l].pitch < 64 * blur_mf_cols(w)) return 0;
        blocks += blur_mf_blocks(w, h);
    }
    return blocks;
#endif
}

void gfo_launch_blur(gfo_ctx* c, const GfoInput& in, int nimg)
{
    gfo_prof_begin(c, ST_BLUR);
    if (c->g.total_tiles + c->g.blur_total_b > 0) {
        static const int xcd_env = getenv("GFO_BLUR_XCD") ? atoi(getenv("GFO_BLUR_XCD")) : 1;
        const int xcd8 = xcd_env && nimg >= 8 ? 1 : 0;
        const int mf = blur_mfma_blocks(c);   // (the matrix-core form has no border blocks: reflection is folded into its band matrices)
        const unsigned blocks = mf > 0 ? (unsigned)mf : (unsigned)(c->g.total_tiles + c->g.blur_total_b);
        const dim3 grid = xcd8 ? dim3(blocks * 8u, (unsigned)(nimg + 7) / 8u) : dim3(blocks, (unsigned)nimg);
        if (mf > 0) GFO_LAUNCH(c, k_blur_mfma, grid, dim3(256), 0, c->stream, c->d_geom, in, c->d_pyr, c->d_blur, xcd8, nimg);
        else GFO_LAUNCH(c, k_blur, grid, dim3(256), 0, c->stream, c->d_geom, in, c->d_pyr, c->d_blur, xcd8, nimg);
    }
    gfo_prof_end(c);
}

// Every __global__ of this translation unit, for gfo_preload_kernels (gfo_api.hip): the runtime loads a code object and
// registers a kernel lazily, on the first launch that needs it; gfo_ctx_create resolves them all once per device under a
// mutex so that no two host threads ever race through that first-launch path (round 3: eight threads, first k_pack_results).
void gfo_kernels_blur(std::vector<const void*>& v) { v.push_back((const void*)k_blur); v.push_back((const void*)k_blur_mfma); }
