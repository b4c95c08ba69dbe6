// k_blur.hip -- cv::GaussianBlur(level.clone(), 7x7, sigma 2, BORDER_REFLECT_101) for CV_8UC1
// (ORBextractor.cc:1154-1155).  Integer taps {18,34,49,55,49,34,18} (sum 257), exact
// accumulation, one rounding: min(255, (sum + 2^15) >> 16).  See DESIGN.md "blur".
//
// Streaming form, no LDS and no barriers: a thread owns 4 adjacent columns and walks down a
// GFO_BLUR_STRIP-row (24) strip.  Per row it loads the 12 bytes [x-4, x+8) as three coalesced dwords (the lanes of a
// wave read one contiguous 256-B run three times, shifted by 4 B: served by L1), forms the four
// horizontal sums with two v_dot4_u32_u8 each (no byte unpacking; max 257*255 = 65535 fits u16 exactly),
// keeps the last seven rows of sums packed in registers, and emits one dword of output per row.
// Vertical halo: 6 extra rows per strip (32-row strips measured 4 % slower alone, 16-row ones 2 % slower in the pipeline).
// All levels of all images are one launch (block index -> level through the prefix table).
#include "gfo_internal.h"

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

// Vertical pass of one column: the seven rows' u16 sums sit in the low (HI = false) or high half of their
// registers; v_dot2_u32_u16 against {tap, 0} (or {0, tap}) multiplies the wanted half and adds it to the
// running sum without unpacking anything.  u32 accumulation (257 * 65535 + 2^15 < 2^32), one rounding.
template <bool HI>
__device__ __forceinline__ unsigned vcol(unsigned a0, unsigned a1, unsigned a2, unsigned a3, unsigned a4, unsigned a5, unsigned a6)
{
#define GFO_TAP(v, k, acc) __builtin_amdgcn_udot2(__builtin_bit_cast(gfo_bu16x2, v), __builtin_bit_cast(gfo_bu16x2, (unsigned)(HI ? ((k) << 16) : (k))), acc, false)
    unsigned acc = GFO_TAP(a0, 18u, 32768u);
    acc = GFO_TAP(a6, 18u, acc);
    acc = GFO_TAP(a1, 34u, acc);
    acc = GFO_TAP(a5, 34u, acc);
    acc = GFO_TAP(a2, 49u, acc);
    acc = GFO_TAP(a4, 49u, acc);
    acc = GFO_TAP(a3, 55u, acc);
#undef GFO_TAP
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

    HRow r0, r1, r2, r3, r4, r5, r6;
    r0 = hrow(load_row<BORDER>(src + (long long)gfo_reflect101(y0 - 3, h) * pitch, x0, w, bs));
    r1 = hrow(load_row<BORDER>(src + (long long)gfo_reflect101(y0 - 2, h) * pitch, x0, w, bs));
    r2 = hrow(load_row<BORDER>(src + (long long)gfo_reflect101(y0 - 1, h) * pitch, x0, w, bs));
    r3 = hrow(load_row<BORDER>(src + (long long)y0 * pitch, x0, w, bs));
    r4 = hrow(load_row<BORDER>(src + (long long)gfo_reflect101(y0 + 1, h) * pitch, x0, w, bs));
    r5 = hrow(load_row<BORDER>(src + (long long)gfo_reflect101(y0 + 2, h) * pitch, x0, w, bs));
    // the raw dwords of row y+4 are requested one step before they are filtered (two rows of loads in flight
    // per thread).  Measured alternatives that were slower: a 7-row unrolled window (106 VGPRs, 4 waves/SIMD),
    // and one load per lane with the neighbour dwords fetched by lane shuffle (the ds_bpermute traffic costs
    // more than the L1-served overlapping loads).
    RawRow nxt = load_row<BORDER>(src + (long long)gfo_reflect101(y0 + 3, h) * pitch, x0, w, bs);
    const int h2 = 2 * h - 2;
    for (int y = y0; y < y1; y++) {
        const RawRow cur = nxt;
        // row y + 4 reflected: it is never above the image, so BORDER_REFLECT_101 is min(a, 2h - 2 - a), and the clamp for
        // images of a few rows adds max(., 0): three operations instead of the general form's eight
        const int ya = y + 4;
        nxt = load_row<BORDER>(src + (long long)max(min(ya, h2 - ya), 0) * pitch, x0, w, bs);
        r6 = hpass_dot(cur.d0, cur.d1, cur.d2);
        const unsigned o0 = vcol<false>(r0.lo, r1.lo, r2.lo, r3.lo, r4.lo, r5.lo, r6.lo);
        const unsigned o1 = vcol<true>(r0.lo, r1.lo, r2.lo, r3.lo, r4.lo, r5.lo, r6.lo);
        const unsigned o2 = vcol<false>(r0.hi, r1.hi, r2.hi, r3.hi, r4.hi, r5.hi, r6.hi);
        const unsigned o3 = vcol<true>(r0.hi, r1.hi, r2.hi, r3.hi, r4.hi, r5.hi, r6.hi);
        *reinterpret_cast<unsigned*>(dst + (long long)y * L.pitch + x0) = o0 | (o1 << 8) | (o2 << 16) | (o3 << 24);
        r0 = r1; r1 = r2; r2 = r3; r3 = r4; r4 = r5; r5 = r6;
    }
}

// One launch: the first blur_total_b blocks of every image are its border blocks (long, thin chains of
// scattered rows -- dispatched first so they run underneath the streaming bulk), the rest the interior.
// Register budget.  The kernel alone is indifferent (6 waves per SIMD at 77 registers, 8 at 64 with a 16-byte spill:
// 109 vs 105 us), but it shares the chip with the quadtree of its own batch and with the other contexts' kernels, and
// there every register it does not hold is a wave of somebody else: 194.0k -> 201.3k frames/s for the whole pipeline
// (same-box A/B, tools/ab_variant.sh; DESIGN.md "footprint").
#ifndef GFO_BLUR_WAVES
#define GFO_BLUR_WAVES 8
#endif
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(GFO_BLUR_WAVES, GFO_BLUR_WAVES))) void k_blur(const GfoGeom* __restrict__ gp, GfoInput in,
                                              const uint8_t* __restrict__ pyr, uint8_t* __restrict__ blur)
{
    const GfoGeom& g = *gp;
    const int nb = g.blur_total_b;
    if ((int)blockIdx.x < nb)
        blur_body<true>(g, in, pyr, blur, blockIdx.x, blockIdx.y);
    else
        blur_body<false>(g, in, pyr, blur, blockIdx.x - nb, blockIdx.y);
}

void gfo_launch_blur(gfo_ctx* c, const GfoInput& in, int nimg)
{
    gfo_prof_begin(c, ST_BLUR);
    if (c->g.total_tiles + c->g.blur_total_b > 0)
        GFO_LAUNCH(c, k_blur, dim3(c->g.total_tiles + c->g.blur_total_b, nimg), dim3(256), 0, c->stream, c->d_geom, in,
                           c->d_pyr, c->d_blur);
    gfo_prof_end(c);
}
