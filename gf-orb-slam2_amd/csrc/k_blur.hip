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

#define BLUR_STRIP GFO_BLUR_STRIP

__device__ __forceinline__ int gfo_reflect101(int p, int n)
{
    if (p < 0) p = -p;
    if (p >= n) p = 2 * n - 2 - p;
    return min(max(p, 0), n - 1);
}

struct HRow {
    unsigned lo, hi;  // four u16 horizontal sums: (x0, x0+1), (x0+2, x0+3)
};

// Horizontal pass of 4 pixels as v_dot4_u32_u8 over the 12 loaded bytes B = pixels x0-4 .. x0+7: output j (pixel x0+j) is
// B[j+1 .. j+7] . (18,34,49,55,49,34,18).  The dwords are used where they lie and the TAPS are shifted instead -- a tap
// vector per (output, dword) with zeros where the window does not reach: 2 + 3 + 3 + 2 dot products, all against scalar
// constants.  (Round 2 cut byte-shifted windows out of the dwords first: 6 v_alignbyte + 8 dot products.)
__device__ __forceinline__ HRow hpass_dot(unsigned d0, unsigned d1, unsigned d2)
{
#define GFO_T4(a, b, c, d) ((unsigned)(a) | ((unsigned)(b) << 8) | ((unsigned)(c) << 16) | ((unsigned)(d) << 24))
    const unsigned s0 = __builtin_amdgcn_udot4(d1, GFO_T4(55, 49, 34, 18), __builtin_amdgcn_udot4(d0, GFO_T4(0, 18, 34, 49), 0u, false), false);
    const unsigned s1 = __builtin_amdgcn_udot4(d2, GFO_T4(18, 0, 0, 0), __builtin_amdgcn_udot4(d1, GFO_T4(49, 55, 49, 34),
                                               __builtin_amdgcn_udot4(d0, GFO_T4(0, 0, 18, 34), 0u, false), false), false);
    const unsigned s2 = __builtin_amdgcn_udot4(d2, GFO_T4(34, 18, 0, 0), __builtin_amdgcn_udot4(d1, GFO_T4(34, 49, 55, 49),
                                               __builtin_amdgcn_udot4(d0, GFO_T4(0, 0, 0, 18), 0u, false), false), false);
    const unsigned s3 = __builtin_amdgcn_udot4(d2, GFO_T4(49, 34, 18, 0), __builtin_amdgcn_udot4(d1, GFO_T4(18, 34, 49, 55), 0u, false), false);
#undef GFO_T4
    HRow r;
    r.lo = s0 | (s1 << 16);  // each sum <= 257 * 255 = 65535
    r.hi = s2 | (s3 << 16);
    return r;
}

struct RawRow {
    unsigned d0, d1, d2;  // pixels x0-4 .. x0+7
};

__device__ __forceinline__ RawRow load_raw(const uint8_t* __restrict__ row, int x0, int w, bool interior)
{
    RawRow r;
    if (interior) {
        r.d0 = *reinterpret_cast<const unsigned*>(row + x0 - 4);
        r.d1 = *reinterpret_cast<const unsigned*>(row + x0);
        r.d2 = *reinterpret_cast<const unsigned*>(row + x0 + 4);
    } else {
        unsigned b[12];
#pragma unroll
        for (int k = 0; k < 12; k++) b[k] = row[gfo_reflect101(x0 - 4 + k, w)];
        r.d0 = b[0] | (b[1] << 8) | (b[2] << 16) | (b[3] << 24);
        r.d1 = b[4] | (b[5] << 8) | (b[6] << 16) | (b[7] << 24);
        r.d2 = b[8] | (b[9] << 8) | (b[10] << 16) | (b[11] << 24);
    }
    return r;
}

// Border quads (the first of a row and the last one or two): the twelve bytes are a fixed, per-thread
// rearrangement of one 16-byte window of the row (bytes [0,16) on the left, [w-16,w) on the right) --
// BORDER_REFLECT_101 folds every out-of-row position back inside it.  The rearrangement is three v_perm_b32
// selectors computed once per thread, so a row costs one wide load instead of twelve byte loads (the lanes of
// a border wave sit in different rows: every load instruction touches 64 cache lines, so their count is what
// the launch costs).
struct BorderSel {
    int ws;            // window start; < 0: row narrower than 16 px, byte path
    unsigned sel[3];   // v_perm selector of output dword d over the dword pair (lo[d], lo[d]+1)
    int lo[3];
};

__device__ __forceinline__ BorderSel border_sel(int x0, int w)
{
    BorderSel b;
    b.ws = w < 16 ? -1 : (x0 == 0 ? 0 : w - 16);
#pragma unroll
    for (int d = 0; d < 3; d++) {
        int j[4], m = 15;
#pragma unroll
        for (int k = 0; k < 4; k++) {
            j[k] = gfo_reflect101(x0 - 4 + 4 * d + k, w) - max(b.ws, 0);
            m = min(m, j[k]);
        }
        b.lo[d] = min(max(m >> 2, 0), 3);
        unsigned sel = 0;
#pragma unroll
        for (int k = 0; k < 4; k++) sel |= (unsigned)((j[k] - 4 * b.lo[d]) & 7) << (8 * k);
        b.sel[d] = sel;
    }
    return b;
}

__device__ __forceinline__ RawRow load_raw_border(const uint8_t* __restrict__ row, int x0, int w, const BorderSel& b)
{
    if (b.ws < 0) return load_raw(row, x0, w, false);
    const uint4 v = *reinterpret_cast<const uint4*>(row + b.ws);
    const unsigned dw[5] = {v.x, v.y, v.z, v.w, 0u};
    unsigned out[3];
#pragma unroll
    for (int d = 0; d < 3; d++) {
        const int l = b.lo[d];
        const unsigned lo = l == 0 ? dw[0] : (l == 1 ? dw[1] : (l == 2 ? dw[2] : dw[3]));
        const unsigned hi = l == 0 ? dw[1] : (l == 1 ? dw[2] : (l == 2 ? dw[3] : dw[4]));
        out[d] = __builtin_amdgcn_perm(hi, lo, b.sel[d]);
    }
    RawRow r;
    r.d0 = out[0]; r.d1 = out[1]; r.d2 = out[2];
    return r;
}

template <bool BORDER>
__device__ __forceinline__ RawRow load_row(const uint8_t* __restrict__ row, int x0, int w, const BorderSel& b)
{
    if (BORDER) return load_raw_border(row, x0, w, b);
    return load_raw(row, x0, w, true);
}

__device__ __forceinline__ HRow hrow(const RawRow& r) { return hpass_dot(r.d0, r.d1, r.d2); }

typedef unsigned short gfo_bu16x2 __attribute__((ext_vector_type(2)));

// Vertical pass.  The window holds ROW PAIRS: a register carries one column's horizontal sums of rows (2k, 2k+1) in its two
// halves, so v_dot2_u32_u16 against a pair of taps does two useful multiply-adds -- an output row is FOUR dot products over
// four pairs (the seven taps and a zero), where a window of single rows packed by column pair needed seven, each with one
// tap wasted on the neighbour column.  u32 accumulation (257 * 65535 + 2^15 < 2^32), one rounding.
struct HPair {
    unsigned c0, c1, c2, c3;   // column j: sums of the pair's even row | odd row << 16
};

__device__ __forceinline__ HPair hpair(const HRow& e, const HRow& o)
{
    HPair p;   // (the four v_perm / v_lshl_or that re-pack two rows of column pairs into four columns of row pairs)
    p.c0 = __builtin_amdgcn_perm(o.lo, e.lo, 0x05040100u);   // e.lo low half | o.lo low half << 16
    p.c1 = __builtin_amdgcn_perm(o.lo, e.lo, 0x07060302u);   // the high halves
    p.c2 = __builtin_amdgcn_perm(o.hi, e.hi, 0x05040100u);
    p.c3 = __builtin_amdgcn_perm(o.hi, e.hi, 0x07060302u);
    return p;
}

#define GFO_DOT2(v, klo, khi, acc) __builtin_amdgcn_udot2(__builtin_bit_cast(gfo_bu16x2, (unsigned)(v)), __builtin_bit_cast(gfo_bu16x2, (unsigned)((klo) | ((khi) << 16))), acc, false)
// output row y (even) of one column from the pairs (y-4, y-3), (y-2, y-1), (y, y+1), (y+2, y+3)
__device__ __forceinline__ unsigned vcol_even(unsigned m2, unsigned m1, unsigned p0, unsigned p1)
{
    unsigned acc = GFO_DOT2(m2, 0u, 18u, 32768u);
    acc = GFO_DOT2(m1, 34u, 49u, acc);
    acc = GFO_DOT2(p0, 55u, 49u, acc);
    acc = GFO_DOT2(p1, 34u, 18u, acc);
    return min(acc >> 16, 255u);
}
// output row y + 1 from the pairs (y-2, y-1), (y, y+1), (y+2, y+3), (y+4, y+5)
__device__ __forceinline__ unsigned vcol_odd(unsigned m1, unsigned p0, unsigned p1, unsigned p2)
{
    unsigned acc = GFO_DOT2(m1, 18u, 34u, 32768u);
    acc = GFO_DOT2(p0, 49u, 55u, acc);
    acc = GFO_DOT2(p1, 49u, 34u, acc);
    acc = GFO_DOT2(p2, 18u, 0u, acc);
    return min(acc >> 16, 255u);
}

// BORDER = false: the quads whose three dwords lie inside the row (no reflection, no byte loads) -- the bulk.
// BORDER = true : the first quad of every row and the last one or two (reflected bytes), in blocks of their
// own, so the streaming loop carries no slow path at all.
template <bool BORDER>
__device__ __forceinline__ void blur_body(const GfoGeom& g, const GfoInput& in, const uint8_t* __restrict__ pyr,
                                          uint8_t* __restrict__ blur, int blk, int img)
{
    int level = 0;
    while (level + 1 < g.nlevels && blk >= (BORDER ? g.lv[level + 1].blur_base_b : g.lv[level + 1].tile_base)) level++;
    const GfoLevel& L = g.lv[level];
    const int t = (blk - (BORDER ? L.blur_base_b : L.tile_base)) * 256 + threadIdx.x;
    const int w = L.w, h = L.h;
    const int quads = L.tiles_x;                       // ceil(w/4)
    const int nint = max(0, min((w - 8) / 4, quads - 1));   // interior quads are 1 .. nint
    const int per_row = BORDER ? quads - nint : nint;
    if (per_row <= 0) return;
    const int strip = t / per_row, qi = t - strip * per_row;
    if (strip >= L.tiles_y) return;
    const int quad = BORDER ? (qi == 0 ? 0 : nint + qi) : qi + 1;
    const int x0 = quad * 4, y0 = strip * BLUR_STRIP;
    int pitch;
#ifdef GFO_BLUR_DEBUG
    // tools/blur_read_bound.sh only: every image reads image (img & 1)'s levels, so the reads are served by the caches
    // and the kernel's time shows what removing its HBM read could gain at most (results are wrong by construction)
    const uint8_t* src = gfo_level_ptr(g, in, pyr, level, img & 1, &pitch);
#else
    const uint8_t* src = gfo_level_ptr(g, in, pyr, level, img, &pitch);
#endif
    uint8_t* dst = blur + (long long)img * g.blur_img_stride + L.blur_off;
    BorderSel bs;
    if (BORDER) bs = border_sel(x0, w);
    const int y1 = min(y0 + BLUR_STRIP, h);

    // strips start at even rows (GFO_BLUR_STRIP is even): the pairs are (y0 + 2k, y0 + 2k + 1)
    static_assert(BLUR_STRIP % 2 == 0, "row pairs need an even strip height");
#define GFO_BLUR_ROW(yy) hrow(load_row<BORDER>(src + (long long)gfo_reflect101((yy), h) * pitch, x0, w, bs))
    HPair m2 = hpair(GFO_BLUR_ROW(y0 - 4), GFO_BLUR_ROW(y0 - 3));
    HPair m1 = hpair(GFO_BLUR_ROW(y0 - 2), GFO_BLUR_ROW(y0 - 1));
    HPair p0 = hpair(GFO_BLUR_ROW(y0), GFO_BLUR_ROW(y0 + 1));
    HPair p1 = hpair(GFO_BLUR_ROW(y0 + 2), GFO_BLUR_ROW(y0 + 3));
#undef GFO_BLUR_ROW
    // the raw dwords of rows y+4, y+5 are requested one step before they are filtered (two rows of loads in flight
    // per thread).  Measured alternatives that were slower: a 7-row unrolled window (106 VGPRs, 4 waves/SIMD),
    // and one load per lane with the neighbour dwords fetched by lane shuffle (the ds_bpermute traffic costs
    // more than the L1-served overlapping loads).
    RawRow nxe = load_row<BORDER>(src + (long long)gfo_reflect101(y0 + 4, h) * pitch, x0, w, bs);
    RawRow nxo = load_row<BORDER>(src + (long long)gfo_reflect101(y0 + 5, h) * pitch, x0, w, bs);
    const int h2 = 2 * h - 2;
    for (int y = y0; y < y1; y += 2) {
        const RawRow ce = nxe, co = nxo;
        // rows y + 6, y + 7 reflected: never above the image, so BORDER_REFLECT_101 is min(a, 2h - 2 - a), and the clamp for
        // images of a few rows adds max(., 0): three operations instead of the general form's eight
        const int ya = y + 6, yb = y + 7;
        nxe = load_row<BORDER>(src + (long long)max(min(ya, h2 - ya), 0) * pitch, x0, w, bs);
        nxo = load_row<BORDER>(src + (long long)max(min(yb, h2 - yb), 0) * pitch, x0, w, bs);
        const HPair p2 = hpair(hpass_dot(ce.d0, ce.d1, ce.d2), hpass_dot(co.d0, co.d1, co.d2));
        {
            const unsigned o0 = vcol_even(m2.c0, m1.c0, p0.c0, p1.c0), o1 = vcol_even(m2.c1, m1.c1, p0.c1, p1.c1);
            const unsigned o2 = vcol_even(m2.c2, m1.c2, p0.c2, p1.c2), o3 = vcol_even(m2.c3, m1.c3, p0.c3, p1.c3);
            *reinterpret_cast<unsigned*>(dst + (long long)y * L.pitch + x0) = o0 | (o1 << 8) | (o2 << 16) | (o3 << 24);
        }
        if (y + 1 < y1) {
            const unsigned o0 = vcol_odd(m1.c0, p0.c0, p1.c0, p2.c0), o1 = vcol_odd(m1.c1, p0.c1, p1.c1, p2.c1);
            const unsigned o2 = vcol_odd(m1.c2, p0.c2, p1.c2, p2.c2), o3 = vcol_odd(m1.c3, p0.c3, p1.c3, p2.c3);
            *reinterpret_cast<unsigned*>(dst + (long long)(y + 1) * L.pitch + x0) = o0 | (o1 << 8) | (o2 << 16) | (o3 << 24);
        }
        m2 = m1; m1 = p0; p0 = p1; p1 = p2;
    }
}
#undef GFO_DOT2

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
        blur_body<true>(g, in, pyr, blur, bx, img);
    else
        blur_body<false>(g, in, pyr, blur, bx - nb, img);
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
