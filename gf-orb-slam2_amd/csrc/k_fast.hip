// k_fast.hip -- per-cell FAST-9/16 with non-max suppression and the per-cell threshold
// fallback of ORBextractor::ComputeKeyPointsOctTree (ORBextractor.cc:786-831), restating
// cv::FAST / FAST_t<16> (in-tree mirror: FAST_NEON.cc:91-287).
//
// One wavefront owns one 30-px cell (+6 px overlap, +3 px ring halo): the cell's pixels are
// staged in LDS once, every lane scores its share of the scan area, and the cell-local
// 3x3 suppression and the "retry with minThFAST when the cell came up empty" decision are
// wave-uniform (ballot / popcount), so no workgroup-wide state is needed.  Four cells per
// 256-thread workgroup; all levels of all images are ONE launch.
//
// Score map S: S(p) = max over the 16 nine-pixel arcs of min |v - ring| with a common sign,
// i.e. the largest t for which p is still a FAST corner, plus one.  It does not depend on the
// threshold, so  corner(t) <=> S > t  and  cornerScore = S - 1  (FAST_NEON.cc:231,
// Fast_gpu.cu:196-219).  Because suppressed-by relations only involve pixels with S >= S(p),
// the reference's "compare against the score buffer of corners at threshold t" reduces to
// "strict local maximum of S inside the cell's scan area, and S > t" (FAST).
//
// Candidates are appended to the level's list with one atomicAdd per cell; their order in
// memory is unspecified.  The order the reference hands to DistributeOctTree (cell-major,
// row-major inside a cell) matters only as a tie-break on equal response, and it is a pure
// function of (x, y) that the quadtree kernel recomputes.
#include "gfo_internal.h"

__device__ __forceinline__ int fast_score16(const uint8_t* __restrict__ c, int tp, int tq)
{
    // ring in the order of FAST_NEON.cc:3-7
    const int v = c[0];
    int d[16];
    d[0] = v - c[3 * tp];
    d[1] = v - c[3 * tp + 1];
    d[2] = v - c[2 * tp + 2];
    d[3] = v - c[tp + 3];
    d[4] = v - c[3];
    d[5] = v - c[-tp + 3];
    d[6] = v - c[-2 * tp + 2];
    d[7] = v - c[-3 * tp + 1];
    d[8] = v - c[-3 * tp];
    d[9] = v - c[-3 * tp - 1];
    d[10] = v - c[-2 * tp - 2];
    d[11] = v - c[-tp - 3];
    d[12] = v - c[-3];
    d[13] = v - c[tp - 3];
    d[14] = v - c[2 * tp - 2];
    d[15] = v - c[3 * tp - 1];
    // necessary condition for S > tq: every opposite pair holds a pixel beyond the threshold
    bool dark = true, bright = true;
#pragma unroll
    for (int k = 0; k < 8; k++) {
        dark = dark && (d[k] > tq || d[k + 8] > tq);
        bright = bright && (d[k] < -tq || d[k + 8] < -tq);
    }
    if (!dark && !bright) return 0;
    int lo2[16], hi2[16], lo4[16], hi4[16];
#pragma unroll
    for (int k = 0; k < 16; k++) {
        lo2[k] = min(d[k], d[(k + 1) & 15]);
        hi2[k] = max(d[k], d[(k + 1) & 15]);
    }
#pragma unroll
    for (int k = 0; k < 16; k++) {
        lo4[k] = min(lo2[k], lo2[(k + 2) & 15]);
        hi4[k] = max(hi2[k], hi2[(k + 2) & 15]);
    }
    int a = -255, b = 255;
#pragma unroll
    for (int k = 0; k < 16; k++) {
        const int lo9 = min(min(lo4[k], lo4[(k + 4) & 15]), d[(k + 8) & 15]);
        const int hi9 = max(max(hi4[k], hi4[(k + 4) & 15]), d[(k + 8) & 15]);
        a = max(a, lo9);
        b = min(b, hi9);
    }
    const int s = max(a, -b);
    return s > tq ? s : 0;
}

__global__ __launch_bounds__(256) void k_fast(const GfoGeom* __restrict__ gp, GfoInput in,
                                              const uint8_t* __restrict__ pyr, uint32_t* __restrict__ cand,
                                              int* __restrict__ cand_cnt, int* __restrict__ flags)
{
    extern __shared__ __attribute__((aligned(16))) uint8_t lds[];
    const GfoGeom& g = *gp;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int img = blockIdx.y;
    const int cell = blockIdx.x * 4 + wave;
    const int TP = g.fast_tile_pitch, SP = g.fast_smap_pitch;
    const int tile_bytes = TP * g.fast_tile_rows, smap_bytes = SP * g.fast_smap_rows;
    uint8_t* tile = lds + wave * (tile_bytes + smap_bytes);
    uint8_t* smap = tile + tile_bytes;

    bool active = cell < g.total_cells;
    int level = 0, cw = 0, ch = 0, sw = 0, sh = 0, ci = 0, cj = 0;
    if (active) {
        while (level + 1 < g.nlevels && cell >= g.lv[level + 1].cell_base) level++;
    }
    const GfoLevel& L = g.lv[level];
    if (active) {
        const int cidx = cell - L.cell_base;
        ci = cidx / L.ncols;
        cj = cidx - ci * L.ncols;
        const int iniX = GFO_MIN_BORDER + cj * L.wcell, iniY = GFO_MIN_BORDER + ci * L.hcell;
        const int maxX = min(iniX + L.wcell + 6, L.max_bx), maxY = min(iniY + L.hcell + 6, L.max_by);
        if (iniY >= L.max_by - 3 || iniX >= L.max_bx - 6) active = false;  // ORBextractor.cc:796,805
        cw = maxX - iniX;
        ch = maxY - iniY;
        sw = cw - 6;
        sh = ch - 6;
        if (sw <= 0 || sh <= 0) active = false;
        if (active) {
            int pitch;
            const uint8_t* src = gfo_level_ptr(g, in, pyr, level, img, &pitch);
            src += (long long)iniY * pitch + iniX;
            const float inv_cw = 1.0f / (float)cw;
            for (int t = lane; t < cw * ch; t += 64) {
                const int r = (int)(((float)t + 0.5f) * inv_cw);
                const int c = t - r * cw;
                tile[r * TP + c] = src[(long long)r * pitch + c];
            }
            uint32_t* sm32 = reinterpret_cast<uint32_t*>(smap);
            for (int t = lane; t < ((sh + 2) * SP) / 4; t += 64) sm32[t] = 0;
        }
    }
    __syncthreads();
    const int tq = min(g.ini_th, g.min_th);
    const int npx = sw * sh;
    const float inv_sw = active ? 1.0f / (float)sw : 0.f;
    if (active) {
        for (int p = lane; p < npx; p += 64) {
            const int py = (int)(((float)p + 0.5f) * inv_sw);
            const int px = p - py * sw;
            const int s = fast_score16(tile + (py + 3) * TP + px + 3, TP, tq);
            smap[(py + 1) * SP + px + 1] = (uint8_t)s;
        }
    }
    __syncthreads();
    if (!active) return;
    // strict local maxima of S; per-lane bitmask over this lane's pixels (<= 64 iterations)
    unsigned long long ismax = 0;
    int n_ini = 0, n_min = 0;
    int it = 0;
    for (int p0 = 0; p0 < npx; p0 += 64, it++) {
        const int p = p0 + lane;
        bool mx = false;
        int s = 0;
        if (p < npx) {
            const int py = (int)(((float)p + 0.5f) * inv_sw);
            const int px = p - py * sw;
            const uint8_t* q = smap + (py + 1) * SP + px + 1;
            s = q[0];
            mx = s >= 2 && s > q[-1] && s > q[1] && s > q[-SP - 1] && s > q[-SP] && s > q[-SP + 1] &&
                 s > q[SP - 1] && s > q[SP] && s > q[SP + 1];
        }
        if (mx) ismax |= 1ull << it;
        n_ini += __popcll(__ballot(mx && s > g.ini_th));
        n_min += __popcll(__ballot(mx && s > g.min_th));
    }
    const int th = n_ini > 0 ? g.ini_th : g.min_th;  // ORBextractor.cc:811-818
    const int total = n_ini > 0 ? n_ini : n_min;
    if (total == 0) return;
    int base = 0;
    int* cnt = cand_cnt + img * g.nlevels + level;
    if (lane == 0) base = atomicAdd(cnt, total);
    base = __shfl(base, 0);
    if (base + total > L.cand_cap) {
        if (lane == 0) atomicOr(&flags[0], 1);
        return;
    }
    uint32_t* out = cand + (long long)img * g.cand_img_stride + L.cand_off + base;
    int run = 0;
    it = 0;
    for (int p0 = 0; p0 < npx; p0 += 64, it++) {
        const int p = p0 + lane;
        bool emit = false;
        uint32_t key = 0;
        if (p < npx && ((ismax >> it) & 1)) {
            const int py = (int)(((float)p + 0.5f) * inv_sw);
            const int px = p - py * sw;
            const int s = smap[(py + 1) * SP + px + 1];
            if (s > th) {
                emit = true;
                const int x = px + 3 + cj * L.wcell, y = py + 3 + ci * L.hcell;  // ORBextractor.cc:824-825
                key = (uint32_t)x | ((uint32_t)y << 12) | ((uint32_t)(s - 1) << 24);
            }
        }
        const unsigned long long m = __ballot(emit);
        if (emit) out[run + __popcll(m & ((1ull << lane) - 1))] = key;
        run += __popcll(m);
    }
}

void gfo_launch_fast(gfo_ctx* c, const GfoInput& in, int nimg)
{
    const GfoGeom& g = c->g;
    const size_t lds = 4 * (size_t)(g.fast_tile_pitch * g.fast_tile_rows + g.fast_smap_pitch * g.fast_smap_rows);
    dim3 grid((g.total_cells + 3) / 4, nimg);
    gfo_prof_begin(c, ST_FAST);
    hipLaunchKernelGGL(k_fast, grid, dim3(256), lds, c->stream, c->d_geom, in, c->d_pyr, c->d_cand, c->d_cand_cnt,
                       c->d_flags);
    gfo_prof_end(c);
}
