// gfo_api.hip -- the extern "C" boundary of libgfo.so (include/gfo.h): context, host tables,
// arena planning, pipeline orchestration.  Host-side C++; every device stage is a HIP kernel
// in its own k_*.hip file.  No CPU fallback exists anywhere in this library.
#include "gfo_internal.h"

#include <math.h>
#include <stdarg.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

static std::string g_create_err;
// live contexts, for gfo_ctx_chain's edges (a destroyed context must disappear from the others' `chain_after`)
#include <mutex>
#include <vector>
#include <atomic>
static std::mutex g_ctx_mu;
static std::vector<gfo_ctx*> g_ctx_live;
static std::atomic<int> g_contexts_created{0}, g_arenas_planned{0};

int gfo_fail(gfo_ctx* c, int code, const char* fmt, ...)
{
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    if (c) c->err = buf;
    else g_create_err = buf;
    return code;
}

#define fail gfo_fail
#define HIP_TRY(c, expr)                                                                        \
    do {                                                                                        \
        hipError_t e_ = (expr);                                                                 \
        if (e_ != hipSuccess) return fail((c), GFO_ERR_DEVICE, "%s: %s", #expr, hipGetErrorString(e_)); \
    } while (0)

static inline long long align_up(long long v, long long a) { return (v + a - 1) / a * a; }

// ---------------------------------------------------------------------------------------------
// profiling
// ---------------------------------------------------------------------------------------------
static const char* k_stage_names[ST_COUNT] = {"resize", "blur", "fast", "quadtree", "orient_desc",
                                              "stereo_bucket", "stereo_match", "stereo_cut", "project", "bow"};

static hipEvent_t ev_get(gfo_ctx* c)
{
    if (!c->ev_pool.empty()) {
        hipEvent_t e = c->ev_pool.back();
        c->ev_pool.pop_back();
        return e;
    }
    hipEvent_t e = nullptr;
    (void)hipEventCreateWithFlags(&e, hipEventReleaseToDevice);   // device-scope release: a system-scope one writes back and invalidates the L2 around the kernel being timed
    return e;
}

void gfo_prof_begin(gfo_ctx* c, int stage) { c->cur_stage = stage; }

// start / stop events of the next kernel launch of the current stage (GFO_LAUNCH)
void gfo_prof_kernel_events(gfo_ctx* c, hipEvent_t* a, hipEvent_t* b)
{
    gfo_ctx::PendingEv p{c->cur_stage, ev_get(c), ev_get(c)};
    if (!p.a || !p.b) return;
    c->pending.push_back(p);
    *a = p.a;
    *b = p.b;
}

void gfo_prof_end(gfo_ctx* c)
{
    // a refused launch (bad configuration) is recorded with its stage; the pipeline driver stops before it
    // launches anything that would consume the missing results.  GFO_DEBUG_SYNC=1 additionally waits for the
    // stage, so that runtime faults are attributed too.
    hipError_t e = hipGetLastError();
    if (e == hipSuccess && c->debug_sync) e = hipStreamSynchronize(c->stream);
    if (e != hipSuccess && c->launch_err.empty())
        c->launch_err = std::string("stage ") + k_stage_names[c->cur_stage] + ": " + hipGetErrorString(e);
}

// Launch refusals recorded by gfo_prof_end() become the call's error (GFO_ERR_DEVICE, stage named).
int gfo_take_launch_err(gfo_ctx* c)
{
    if (c->launch_err.empty()) return GFO_OK;
    const std::string m = c->launch_err;
    c->launch_err.clear();
    return fail(c, GFO_ERR_DEVICE, "%s", m.c_str());
}

static void prof_collect(gfo_ctx* c)
{
    if (c->pending.empty()) return;
    (void)hipStreamSynchronize(c->stream);
    for (auto& p : c->pending) {
        float ms = 0.f;
        if (hipEventElapsedTime(&ms, p.a, p.b) == hipSuccess) {
            c->stage_ms[p.stage] += ms;
            c->stage_launches[p.stage]++;
        }
        c->ev_pool.push_back(p.a);
        c->ev_pool.push_back(p.b);
    }
    c->pending.clear();
}

// ---------------------------------------------------------------------------------------------
// host tables -- ORBextractor::ORBextractor, ORBextractor.cc:409-469
// ---------------------------------------------------------------------------------------------
static void build_tables(gfo_ctx* c)
{
    const int n = c->prm.nlevels;
    c->scale.assign(n, 1.f);
    c->inv_scale.assign(n, 1.f);
    c->sigma2.assign(n, 1.f);
    c->inv_sigma2.assign(n, 1.f);
    c->quota.assign(n, 0);
    for (int i = 1; i < n; i++) {
        c->scale[i] = c->scale[i - 1] * c->prm.scale_factor;
        c->sigma2[i] = c->scale[i] * c->scale[i];
    }
    for (int i = 0; i < n; i++) {
        c->inv_scale[i] = 1.0f / c->scale[i];
        c->inv_sigma2[i] = 1.0f / c->sigma2[i];
    }
    const float factor = (float)(1.0f / (double)c->prm.scale_factor);  // scaleFactor is a double member (ORBextractor.h:155)
    float per = c->prm.nfeatures * (1 - factor) / (1 - (float)pow((double)factor, (double)n));
    int sum = 0;
    for (int l = 0; l < n - 1; l++) {
        c->quota[l] = (int)lrintf(per);  // cvRound: half to even
        sum += c->quota[l];
        per *= factor;
    }
    c->quota[n - 1] = c->prm.nfeatures - sum > 0 ? c->prm.nfeatures - sum : 0;
}

// ---------------------------------------------------------------------------------------------
// arena
// ---------------------------------------------------------------------------------------------
static void free_arena(gfo_ctx* c)
{
    void* ptrs[] = {c->d_geom, c->d_input, c->d_pyr, c->d_blur, c->d_cand, c->d_cand_cnt, c->d_node_of, c->d_sel,
                    c->d_sel_cnt, c->d_kp, c->d_desc, c->d_kp_cnt, c->d_flags, c->d_xofs, c->d_xcoef, c->d_yofs, c->d_band, c->d_cell_tab, c->d_qt_scratch,
                    c->d_ycoef, c->st.u_right, c->st.depth, c->st.best_dist, c->st.best_idx, c->st.nmatched, c->st.counted,
                    c->d_scale, c->d_inv_scale, c->st_sort.sx, c->st_sort.sy, c->st_sort.soi, c->st_sort.sdesc, c->st_sort.row_start, c->st_sort.lorder, c->st_sort.lrow_start};
    for (void* p : ptrs)
        if (p) (void)hipFree(p);
    c->d_geom = nullptr; c->d_input = c->d_pyr = c->d_blur = nullptr; c->d_cand = nullptr; c->d_cand_cnt = nullptr;
    c->d_node_of = nullptr; c->d_sel = nullptr; c->d_sel_cnt = nullptr; c->d_kp = nullptr; c->d_desc = nullptr;
    c->d_kp_cnt = nullptr; c->d_flags = nullptr; c->d_xofs = nullptr; c->d_xcoef = nullptr; c->d_yofs = nullptr; c->d_band = nullptr; c->d_cell_tab = nullptr; c->d_od_tab = nullptr; c->od_pairs = 0; c->d_qt_scratch = nullptr;
    c->d_ycoef = nullptr; c->st = GfoStereoDev{}; c->d_scale = nullptr; c->d_inv_scale = nullptr; c->st_sort = GfoStereoSort{}; c->st_rows_cap = 0;
    c->planned = false;
    c->have_batch = c->have_pyramid = c->have_stereo = false;
    c->plan_gen++;                         // every captured launch sequence points into the old arena
    if (c->graph_exec) {
        (void)hipGraphExecDestroy(c->graph_exec);
        c->graph_exec = nullptr;
    }
}

// cv::resize(INTER_LINEAR) coefficient tables, built exactly as OpenCV builds them (double
// scale, float fraction, 11-bit rounding; x clamps the fraction at the borders, y clips rows).
static void resize_tables(int ssize, int dsize, int* ofs, short* coef, bool clamp_x)
{
    const double inv_scale = (double)dsize / ssize;
    const double scale = 1. / inv_scale;
    for (int d = 0; d < dsize; d++) {
        float f = (float)((d + 0.5) * scale - 0.5);
        int s = (int)floorf(f);
        f -= s;
        if (clamp_x) {
            if (s < 0) { f = 0; s = 0; }
            if (s >= ssize - 1) { f = 0; s = ssize - 1; }
        }
        ofs[d] = s;
        coef[2 * d] = (short)lrintf((1.f - f) * 2048.f);
        coef[2 * d + 1] = (short)lrintf(f * 2048.f);
    }
}

#if GFO_OCV_RESIZE == 1
// the float variant: the same sample positions, the FRACTION itself (as float bits) instead of the 11-bit pair; both axes clamp
// (source index into [0, ssize - 1] with fraction 0 at either end), as a float bilinear does
static void resize_tables_float(int ssize, int dsize, int* ofs, int* frac_bits)
{
    const double scale = 1. / ((double)dsize / ssize);
    for (int d = 0; d < dsize; d++) {
        float f = (float)((d + 0.5) * scale - 0.5);
        int s = (int)floorf(f);
        f -= s;
        if (s < 0) { f = 0; s = 0; }
        if (s >= ssize - 1) { f = 0; s = ssize - 1; }
        ofs[d] = s;
        memcpy(&frac_bits[d], &f, 4);
    }
}
#endif

int gfo_plan(gfo_ctx* c, int w, int h, int batch)
{
    if (c->planned && c->g.w0 == w && c->g.h0 == h && batch <= c->cap_batch) return GFO_OK;
    if (w > 4000 || h > 4000) return fail(c, GFO_ERR_INVALID, "image %dx%d exceeds the 4000-px coordinate packing", w, h);
    const int keep_batch = c->planned && c->g.w0 == w && c->g.h0 == h ? c->cap_batch : 0;
    (void)hipStreamSynchronize(c->stream);
    free_arena(c);
    batch = batch > keep_batch ? batch : keep_batch;
    if (batch < c->prm.max_batch) batch = c->prm.max_batch;
    GfoGeom& g = c->g;
    memset(&g, 0, sizeof g);
    g.nlevels = c->prm.nlevels;
    g.w0 = w;
    g.h0 = h;
    g.ini_th = c->prm.ini_th_fast;
    g.min_th = c->prm.min_th_fast;
    long long pyr_off = 0, blur_off = 0, cand_off = 0;
    int cell_base = 0, tile_base = 0, blur_b = 0, sel_off = 0, xtab = 0, ytab = 0;
    int max_cw = 8, max_ch = 8;
    for (int l = 0; l < g.nlevels; l++) {
        GfoLevel& L = g.lv[l];
        const float s = c->inv_scale[l];
        L.w = (int)lrintf((float)w * s);  // ORBextractor.cc:1180-1181
        L.h = (int)lrintf((float)h * s);
        if (L.w < 1 || L.h < 1) return fail(c, GFO_ERR_INVALID, "level %d of a %dx%d image is empty", l, w, h);
        L.pitch = (int)align_up(L.w, 64);
        L.plane_off = pyr_off;
        if (l > 0) pyr_off += align_up((long long)L.pitch * L.h, 256);
        L.blur_off = blur_off;
        blur_off += align_up((long long)L.pitch * L.h, 256);
        L.max_bx = L.w - GFO_EDGE + 3;
        L.max_by = L.h - GFO_EDGE + 3;
        const float width = (float)(L.max_bx - GFO_MIN_BORDER), height = (float)(L.max_by - GFO_MIN_BORDER);
        L.cell_base = cell_base;
        if (width >= (float)GFO_CELL_W && height >= (float)GFO_CELL_W) {  // :783-789
            L.ncols = (int)(width / GFO_CELL_W);
            L.nrows = (int)(height / GFO_CELL_W);
            L.wcell = (int)ceilf(width / L.ncols);
            L.hcell = (int)ceilf(height / L.nrows);
            cell_base += L.ncols * L.nrows;
            max_cw = L.wcell + 6 > max_cw ? L.wcell + 6 : max_cw;
            max_ch = L.hcell + 6 > max_ch ? L.hcell + 6 : max_ch;
            L.cand_cap = L.ncols * L.nrows * ((L.wcell + 1) / 2) * ((L.hcell + 1) / 2);
        } else {
            L.ncols = L.nrows = 0;
            L.wcell = L.hcell = 1;
            L.cand_cap = 0;
        }
        L.cand_off = cand_off;
        cand_off += align_up(L.cand_cap, 64);
        L.tiles_x = (L.w + 3) / 4;      // blur: column quads per row
        L.tiles_y = (L.h + GFO_BLUR_STRIP - 1) / GFO_BLUR_STRIP;    // blur: strips of GFO_BLUR_STRIP rows
        {
            int nint = (L.w - 8) / 4;
            if (nint > L.tiles_x - 1) nint = L.tiles_x - 1;
            if (nint < 0) nint = 0;
            L.tile_base = tile_base;        // blur: first block of this level in the interior launch
            tile_base += (nint * L.tiles_y + 255) / 256;
            L.blur_base_b = blur_b;         // ... and in the border launch
            blur_b += ((L.tiles_x - nint) * L.tiles_y + 255) / 256;
        }
        L.quota = c->quota[l];
        int nini = 1;
        if (L.max_by - GFO_MIN_BORDER > 0) nini = (int)roundf(width / height);  // :543
        if (nini < 1) nini = 1;
        L.n_ini = nini;
        L.hx = width / nini;
        L.node_cap = (L.quota > 4 * nini ? L.quota : 4 * nini) + 8;
        L.sel_off = sel_off;
        L.sel_cap = L.node_cap;
        sel_off += L.sel_cap;
        L.scale = c->scale[l];
        L.patch_size = (int)(GFO_PATCH * c->scale[l]);  // :839
        L.xtab_off = xtab;
        L.ytab_off = ytab;
        if (l > 0) {
            xtab += (int)align_up(L.w + 3, 4) + 4;   // padded: see the table construction below
            ytab += (int)align_up(L.h + 3, 4) + 4;
        }
        if (L.node_cap > 60000) return fail(c, GFO_ERR_INVALID, "level %d takes %d of the %d features: more than 60000 nodes per level are not supported", l, L.quota, c->prm.nfeatures);
    }
    if (max_cw - 6 > 64 || max_ch - 6 > 64)
        return fail(c, GFO_ERR_INVALID, "FAST cell %dx%d exceeds the per-wave plan", max_cw - 6, max_ch - 6);
    const int xtab_n = xtab, ytab_n = ytab;
    g.total_cells = cell_base;
    g.total_tiles = tile_base;
    g.blur_total_b = blur_b;
    g.total_sel_cap = sel_off;
    g.kp_stride = (int)align_up(sel_off, 4);
    // +3: tile rows start at the aligned dword left of the cell; 16-B segments.  Bucketed to the three pitches
    // k_fast is instantiated for (48/44, 64/60, 80/76: tile / score-map pitch).
    g.fast_tile_pitch = max_cw <= 45 ? 48 : (max_cw <= 61 ? 64 : 80);
    g.fast_npx_max = (int)align_up((long long)(max_cw - 6) * (max_ch - 6), 8);
    {
        const int qcap_env = getenv("GFO_FAST_QCAP") ? atoi(getenv("GFO_FAST_QCAP")) : 0;   // read per plan: tests force a small queue (>= 264) to reach the overflow paths
        const int qc = qcap_env >= 264 ? (qcap_env & ~7) : GFO_FAST_QCAP;   // (264: a pass of 256 pixels must fit; at that size nearly every cell takes the dense pass)
        g.fast_q_cap = g.fast_npx_max < qc ? g.fast_npx_max : qc;
    }
    g.fast_tile_rows = max_ch;
    g.fast_smap_pitch = g.fast_tile_pitch - 4;
    g.fast_smap_rows = (int)align_up((long long)(max_ch - 6 + 2) * g.fast_smap_pitch, 16) / g.fast_smap_pitch + 1;  // zeroed in 16-B steps
    g.pyr_img_stride = align_up(pyr_off, 256);
    g.blur_img_stride = align_up(blur_off, 256);
    g.cand_img_stride = align_up(cand_off, 64);
    if (g.kp_stride > 65535) return fail(c, GFO_ERR_INVALID, "more than 65535 keypoints per image are not supported");

    // resize tables: one {offset, coef0 | coef1 << 16} pair per output column / row, each level padded with
    // 3 copies of its last entry and kept 16-byte aligned (k_resize reads a thread's 4 entries as two int4)
    std::vector<int> xtabv(2 * (size_t)(xtab_n > 0 ? xtab_n : 4)), ytabv(2 * (size_t)(ytab_n > 0 ? ytab_n : 4));
    for (int l = 1; l < g.nlevels; l++) {
        const int dw = g.lv[l].w, dh = g.lv[l].h;
        std::vector<int> ofs(dw > dh ? dw : dh);
        std::vector<short> coef(2 * ofs.size());
#if GFO_OCV_RESIZE == 1
        std::vector<int> fb(ofs.size());
        resize_tables_float(g.lv[l - 1].w, dw, ofs.data(), fb.data());
        for (int d = 0; d < dw + 3 + 4; d++) {
            const int s_ = d < dw ? d : dw - 1;
            if (g.lv[l].xtab_off + d >= xtab_n) break;
            xtabv[2 * (g.lv[l].xtab_off + d)] = ofs[s_];
            xtabv[2 * (g.lv[l].xtab_off + d) + 1] = fb[s_];
        }
        resize_tables_float(g.lv[l - 1].h, dh, ofs.data(), fb.data());
        for (int d = 0; d < dh + 3 + 4; d++) {
            const int s_ = d < dh ? d : dh - 1;
            if (g.lv[l].ytab_off + d >= ytab_n) break;
            ytabv[2 * (g.lv[l].ytab_off + d)] = ofs[s_];
            ytabv[2 * (g.lv[l].ytab_off + d) + 1] = fb[s_];
        }
        (void)coef;
#else
        resize_tables(g.lv[l - 1].w, dw, ofs.data(), coef.data(), true);
        for (int d = 0; d < dw + 3 + 4; d++) {
            const int s_ = d < dw ? d : dw - 1;
            if (g.lv[l].xtab_off + d >= xtab_n) break;
            xtabv[2 * (g.lv[l].xtab_off + d)] = ofs[s_];
            xtabv[2 * (g.lv[l].xtab_off + d) + 1] = (int)((unsigned short)coef[2 * s_] | ((unsigned)(unsigned short)coef[2 * s_ + 1] << 16));
        }
        resize_tables(g.lv[l - 1].h, dh, ofs.data(), coef.data(), false);
        for (int d = 0; d < dh + 3 + 4; d++) {
            const int s_ = d < dh ? d : dh - 1;
            if (g.lv[l].ytab_off + d >= ytab_n) break;
            ytabv[2 * (g.lv[l].ytab_off + d)] = ofs[s_];
            ytabv[2 * (g.lv[l].ytab_off + d) + 1] = (int)((unsigned short)coef[2 * s_] | ((unsigned)(unsigned short)coef[2 * s_ + 1] << 16));
        }
#endif
    }
    // banded pyramid: the levels are cut into groups of consecutive levels, each one launch of k_pyramid_bands
    // (a group of ONE level is an ordinary k_resize launch).  Greedy from level 1: the longest group (up to
    // GFO_PYR_GROUP levels) for which some band count <= 64 keeps the per-workgroup LDS footprint within the
    // budget; inside a group the fewest such bands.  Ranges are derived top-down inside a group: the rows a level
    // must hold are its own share plus the source rows the level above reads for ITS computed rows.
    std::vector<int> bandv;
    g.pyr_nb = 0;
    c->n_band_groups = 0;
    if (g.nlevels >= 2) {
        const int budget = (getenv("GFO_PYR_LDS_KB") ? atoi(getenv("GFO_PYR_LDS_KB")) : 24) * 1024;   // measured best with 256 threads (752x480)
        const int max_group = getenv("GFO_PYR_GROUP") ? atoi(getenv("GFO_PYR_GROUP")) : 4;
        const int nl = g.nlevels;
        auto sy_of = [&](int l, int dy) {
            const int v = ytabv[2 * (size_t)(g.lv[l].ytab_off + dy)], sh = g.lv[l - 1].h;
            return v < 0 ? 0 : (v > sh - 1 ? sh - 1 : v);
        };
        // plan of one group [lb, le): fills bg / tab / lds offsets on success
        double px_own = 0.0, px_done = 0.0;   // pixels the levels hold / pixels the bands compute (halo included)
        auto plan_group = [&](int lb, int le, GfoBandGroup* out_bg, std::vector<int>* out_tab, int* lp_out, int* off_out) {
            for (int nb = 1; nb <= 64; nb++) {
                if (g.lv[le - 1].h < 2 * nb && nb > 1) break;
                std::vector<int> tab(4 * (size_t)nb * nl, 0);
                int maxrows[GFO_MAX_LEVELS] = {0};
                for (int b = 0; b < nb; b++) {
                    int c0_up = 0, c1_up = 0;
                    for (int l = le - 1; l >= lb; l--) {
                        const int hl = g.lv[l].h;
                        const int o0 = (int)((long long)b * hl / nb), o1 = (int)((long long)(b + 1) * hl / nb);
                        int c0 = o0, c1 = o1;
                        if (l < le - 1 && c1_up > c0_up) {
                            const int lo = sy_of(l + 1, c0_up);
                            int hi = sy_of(l + 1, c1_up - 1) + 1;
                            if (hi > hl - 1) hi = hl - 1;
                            c0 = lo < c0 ? lo : c0;
                            c1 = hi + 1 > c1 ? hi + 1 : c1;
                        }
                        int* t = &tab[4 * ((size_t)b * nl + l)];
                        t[0] = c0; t[1] = c1; t[2] = o0; t[3] = o1;
                        if (c1 - c0 > maxrows[l]) maxrows[l] = c1 - c0;
                        c0_up = c0;
                        c1_up = c1;
                    }
                }
                int lds = 0;
                for (int l = lb; l < le - 1; l++) lds += (int)align_up((long long)align_up(g.lv[l].w, 16) * maxrows[l], 16);
                if (getenv("GFO_DEBUG_PLAN")) fprintf(stderr, "[gfo] pyramid bands, levels %d..%d: nb %d -> %d B of LDS (budget %d)\n", lb, le - 1, nb, lds, budget);
                if (lds <= budget) {
                    out_bg->lb = lb; out_bg->le = le; out_bg->nb = nb; out_bg->lds_bytes = lds;
                    int off = 0;
                    for (int l = lb; l < le - 1; l++) {
                        lp_out[l] = (int)align_up(g.lv[l].w, 16);
                        off_out[l] = off;
                        off += (int)align_up((long long)lp_out[l] * maxrows[l], 16);
                    }
                    for (int l = lb; l < le; l++) {
                        px_own += (double)g.lv[l].w * g.lv[l].h;
                        for (int b = 0; b < nb; b++) px_done += (double)g.lv[l].w * (tab[4 * ((size_t)b * nl + l) + 1] - tab[4 * ((size_t)b * nl + l)]);
                    }
                    out_tab->swap(tab);
                    return true;
                }
            }
            return false;
        };
        bool ok = true;
        int lb = 1;
        while (lb < nl && ok) {
            int take = 0;
            GfoBandGroup bg = {0, 0, 0, 0, 0};
            std::vector<int> tab;
            for (int size = (nl - lb < max_group ? nl - lb : max_group); size >= 2 && take == 0; size--)
                if (plan_group(lb, lb + size, &bg, &tab, g.band_lp, g.band_lds_off)) take = size;
            if (take == 0) {          // not even two levels fit: this level is an ordinary per-level launch
                bg.lb = lb; bg.le = lb + 1; bg.nb = 0; bg.lds_bytes = 0;
                take = 1;
            } else {
                bg.tab_off = (int)(bandv.size() / 4);
                bandv.insert(bandv.end(), tab.begin(), tab.end());
                if (gfo_pyramid_bands_prepare(bg.lds_bytes) != 0) {
                    (void)hipGetLastError();
                    ok = false;
                    break;
                }
            }
            if (c->n_band_groups >= GFO_MAX_LEVELS) { ok = false; break; }
            c->band_groups[c->n_band_groups++] = bg;
            if (bg.nb > 0) g.pyr_nb = g.pyr_nb == 0 || bg.nb < g.pyr_nb ? bg.nb : g.pyr_nb;
            lb += take;
        }
        if (getenv("GFO_DEBUG_PLAN") && px_own > 0) fprintf(stderr, "[gfo] pyramid bands: %d groups, computed / owned pixels = %.3f\n", c->n_band_groups, px_done / px_own);
        // wide images pay the halo rows at full width: past 1.3x recomputation the per-level launches win (1080p: 1.4)
        const double max_over = getenv("GFO_PYR_MAX_OVERHEAD") ? atof(getenv("GFO_PYR_MAX_OVERHEAD")) : 1.30;
        if (px_own > 0 && px_done / px_own > max_over) ok = false;
        // measured: for rows wider than ~1000 px (1080p: level 1 = 1600 px) the per-level launches are already efficient
        // and beat every band configuration (477 vs >= 499 us per 128 images); bands are for the VGA-class sizes
        const int max_band_w = getenv("GFO_PYR_MAX_W") ? atoi(getenv("GFO_PYR_MAX_W")) : 1024;
        if (g.lv[1].w > max_band_w) ok = false;
        if (!ok || g.pyr_nb == 0) {   // nothing banded at all: the plain per-level path
            c->n_band_groups = 0;
            g.pyr_nb = 0;
        }
        c->band_threads = getenv("GFO_PYR_THREADS") ? atoi(getenv("GFO_PYR_THREADS")) : 0;   // 0: by launch size (k_pyramid.hip)
    }
    const size_t B = (size_t)batch;
    HIP_TRY(c, hipMalloc(&c->d_geom, sizeof(GfoGeom)));
    HIP_TRY(c, hipMalloc(&c->d_input, B * g.lv[0].pitch * (size_t)h + 64));   // +64: the resize row window may look past a row (never past the last row of a caller's image)
    HIP_TRY(c, hipMalloc(&c->d_pyr, B * (size_t)(g.pyr_img_stride > 0 ? g.pyr_img_stride : 256) + 256));
    HIP_TRY(c, hipMalloc(&c->d_blur, B * (size_t)g.blur_img_stride + 256));  // +256: window staging may read 3 B past a row end
    HIP_TRY(c, hipMalloc(&c->d_cand, B * (size_t)(g.cand_img_stride + 64) * sizeof(uint32_t)));
    HIP_TRY(c, hipMalloc(&c->d_node_of, B * (size_t)(g.cand_img_stride + 64) * sizeof(uint16_t)));
    HIP_TRY(c, hipMalloc(&c->d_cand_cnt, B * g.nlevels * GFO_CNT_STRIDE * sizeof(int)));
    HIP_TRY(c, hipMalloc(&c->d_sel, B * (size_t)g.total_sel_cap * sizeof(uint32_t)));
    HIP_TRY(c, hipMalloc(&c->d_sel_cnt, B * g.nlevels * sizeof(int)));
    HIP_TRY(c, hipMalloc(&c->d_kp, B * (size_t)g.kp_stride * sizeof(gfo_keypoint)));
    HIP_TRY(c, hipMalloc(&c->d_desc, B * (size_t)g.kp_stride * 32));
    HIP_TRY(c, hipMalloc(&c->d_kp_cnt, B * sizeof(int) + 16));   // +16: k_pack_results copies in 16-byte units
    HIP_TRY(c, hipMalloc(&c->d_flags, 8 * sizeof(int)));   // [0..3] the live flags the kernels OR into, [4..7] the snapshot gfo_batch_deliver hands out
    HIP_TRY(c, hipMalloc(&c->d_xofs, xtabv.size() * sizeof(int) + 64));
    HIP_TRY(c, hipMalloc(&c->d_yofs, ytabv.size() * sizeof(int) + 64));
    {
        // FAST cell table, four ints per cell: {level | row << 4 | column << 16, tile-load map, stage-A map, 0}.  The two maps
        // are the lane -> (row, column) decompositions k_fast uses, which depend on the cell's clipped size only: dividing a
        // lane index by a wave-uniform count costs a wave ~25 vector instructions (no scalar division, no scalar float unit),
        // a multiplication by ceil(4096 / n) and a shift cost two (exact for lanes 0..63 and n <= 16, checked below).
        //   map = n | (64 / n) << 5 | ceil(4096 / n) << 12      n = 16-byte segments per tile row / dwords per scan row
        std::vector<int> cells(4 * ((size_t)g.total_cells + 4), 0);
        auto lane_map = [&](int n) -> int {
            if (n < 1 || n > 16) return 0;   // (cells beyond 64 px are refused above)
            const int m = (4096 + n - 1) / n;
            for (int lane = 0; lane < 64; lane++)
                if (((lane * m) >> 12) != lane / n) return 0;
            return n | ((64 / n) << 5) | (m << 12);
        };
        for (int l = 0; l < g.nlevels; l++)
            for (int i = 0; i < g.lv[l].nrows; i++)
                for (int j = 0; j < g.lv[l].ncols; j++) {
                    const GfoLevel& L = g.lv[l];
                    const int iniX = GFO_MIN_BORDER + j * L.wcell;
                    const int maxX = iniX + L.wcell + 6 < L.max_bx ? iniX + L.wcell + 6 : L.max_bx;
                    const int cw = maxX - iniX;          // as k_fast computes it (a cell with cw <= 6 is skipped there)
                    int* e = &cells[4 * ((size_t)L.cell_base + i * L.ncols + j)];
                    e[0] = l | (i << 4) | (j << 16);
                    if (cw > 6) {
                        e[1] = lane_map((GFO_FAST_XOFF + cw + 15) >> 4);
                        e[2] = lane_map((cw - 6 + 3) >> 2);
                        if (!e[1] || !e[2]) return fail(c, GFO_ERR_INVALID, "FAST cell of %d px exceeds the per-wave plan", cw - 6);
                    }
                }
        // behind the cells: the keypoint-pair table of k_orient_desc, two ints per wavefront of an image: level | pair << 4 and the level's sel_off --
        // a wave owns selection slots (2 * pair, 2 * pair + 1) of ONE level, so everything that depends on the level is
        // wave-uniform there; -1 pads the table to whole workgroups
        if (cells.size() & 1) cells.push_back(0);   // (the pair table is read as int2)
        const size_t od_base = cells.size();
        for (int l = 0; l < g.nlevels; l++)
            for (int pr = 0; pr < (g.lv[l].sel_cap + 1) / 2; pr++) {
                cells.push_back(l | (pr << 4));
                cells.push_back(g.lv[l].sel_off);    // the level's first selection slot: the kernel requests a pair's selection words without looking the level up
            }
        c->od_pairs = (int)((cells.size() - od_base) / 2);
        for (int k = 0; k < 16; k++) { cells.push_back(-1); cells.push_back(0); }
        HIP_TRY(c, hipMalloc(&c->d_cell_tab, cells.size() * sizeof(int)));
        HIP_TRY(c, hipMemcpy(c->d_cell_tab, cells.data(), cells.size() * sizeof(int), hipMemcpyHostToDevice));
        c->d_od_tab = c->d_cell_tab + od_base;
    }
    {
        // quadtree: a level whose node tables exceed the 160 KB of LDS runs the global-memory variant of the kernel
        // (the usual 8-level configurations never do: that takes more than 2040 features on one level)
        int ncap_max = 0;
        for (int l = 0; l < g.nlevels; l++) ncap_max = g.lv[l].node_cap > ncap_max ? g.lv[l].node_cap : ncap_max;
        c->qt_scratch_stride = 0;
        if (gfo_quadtree_lds_bytes(ncap_max, 0) > 160 * 1024) {
            c->qt_scratch_stride = align_up((long long)gfo_quadtree_lds_bytes(ncap_max, 0), 256);
            const size_t total = c->qt_scratch_stride * B * (size_t)g.nlevels;
            if (total > ((size_t)4 << 30))
                return fail(c, GFO_ERR_INVALID, "quadtree scratch for %d features on one level x %zu images (%zu MB) exceeds 4 GB: lower max_batch", c->prm.nfeatures, B, total >> 20);
            HIP_TRY(c, hipMalloc(&c->d_qt_scratch, total));
        }
    }
    HIP_TRY(c, hipMalloc(&c->d_band, (bandv.size() + 4) * sizeof(int)));
    if (!bandv.empty()) HIP_TRY(c, hipMemcpy(c->d_band, bandv.data(), bandv.size() * sizeof(int), hipMemcpyHostToDevice));
    HIP_TRY(c, hipMalloc(&c->d_scale, GFO_MAX_LEVELS * sizeof(float)));
    HIP_TRY(c, hipMalloc(&c->d_inv_scale, GFO_MAX_LEVELS * sizeof(float)));
    const size_t P = (B + 1) / 2;
    HIP_TRY(c, hipMalloc(&c->st.u_right, P * g.kp_stride * sizeof(float)));
    HIP_TRY(c, hipMalloc(&c->st.depth, P * g.kp_stride * sizeof(float)));
    HIP_TRY(c, hipMalloc(&c->st.best_dist, P * g.kp_stride * sizeof(int)));
    HIP_TRY(c, hipMalloc(&c->st.best_idx, P * g.kp_stride * sizeof(int)));
    HIP_TRY(c, hipMalloc(&c->st.nmatched, P * sizeof(int) + 16));
    HIP_TRY(c, hipMalloc(&c->st.counted, P * g.kp_stride));
    HIP_TRY(c, hipMalloc(&c->st_sort.sx, P * g.kp_stride * sizeof(float)));
    HIP_TRY(c, hipMalloc(&c->st_sort.sy, P * g.kp_stride * sizeof(float)));
    HIP_TRY(c, hipMalloc(&c->st_sort.soi, P * g.kp_stride * sizeof(unsigned)));
    HIP_TRY(c, hipMalloc(&c->st_sort.sdesc, P * g.kp_stride * 32));
    c->st_rows_cap = h + 64;
    HIP_TRY(c, hipMalloc(&c->st_sort.row_start, P * (size_t)(c->st_rows_cap + 1) * sizeof(int)));
    HIP_TRY(c, hipMalloc(&c->st_sort.lrow_start, P * (size_t)(c->st_rows_cap + 1) * sizeof(int)));
    HIP_TRY(c, hipMalloc(&c->st_sort.lorder, P * g.kp_stride * sizeof(int)));
    HIP_TRY(c, hipMemcpy(c->d_geom, &g, sizeof g, hipMemcpyHostToDevice));
    HIP_TRY(c, hipMemcpy(c->d_xofs, xtabv.data(), xtabv.size() * sizeof(int), hipMemcpyHostToDevice));
    HIP_TRY(c, hipMemcpy(c->d_yofs, ytabv.data(), ytabv.size() * sizeof(int), hipMemcpyHostToDevice));
    HIP_TRY(c, hipMemcpy(c->d_scale, c->scale.data(), g.nlevels * sizeof(float), hipMemcpyHostToDevice));
    HIP_TRY(c, hipMemcpy(c->d_inv_scale, c->inv_scale.data(), g.nlevels * sizeof(float), hipMemcpyHostToDevice));
    HIP_TRY(c, hipMemset(c->d_flags, 0, 8 * sizeof(int)));
    c->cap_batch = batch;
    c->planned = true;
    g_arenas_planned++;
    return GFO_OK;
}

// ---------------------------------------------------------------------------------------------
// context
// ---------------------------------------------------------------------------------------------
extern "C" int gfo_version(void) { return GFO_VERSION; }

extern "C" int gfo_build_variant(int key)
{
    static const int taps[4] = {GFO_GAUSS_TAPS};
    switch (key) {
    case 0: return GFO_OCV_RESIZE;
    case 1: return GFO_OCV_ATAN_FMA;
    case 2: return GFO_OCV_BLUR_ROUND;
    case 3: case 9: return taps[0];
    case 4: case 8: return taps[1];
    case 5: case 7: return taps[2];
    case 6: return taps[3];
    default: return -1;
    }
}

extern "C" const char* gfo_last_error(const gfo_ctx* ctx) { return ctx ? ctx->err.c_str() : g_create_err.c_str(); }

// The HIP runtime loads a translation unit's code object and registers each kernel on the FIRST launch that needs it.
// Round 3 saw eight host threads reach their first k_pack_results launch together and one of them fault inside the runtime
// (under rocprofv3, which hooks exactly that code-object load; profiles/boundary_trace_r04.txt).  Whatever the runtime's own
// locking is worth there, this library no longer depends on it: the first gfo_ctx_create on a device resolves every kernel
// of every translation unit (hipFuncGetAttributes = load + register, no launch) under one mutex, and later contexts find
// the device marked.  GFO_PRELOAD=0 restores the lazy behaviour (experiments only).
static void gfo_kernels_api(std::vector<const void*>& v);
static std::mutex g_preload_mu;
static uint64_t g_preloaded_devices = 0;
static std::atomic<int> g_kernels_preloaded{0};
static int gfo_preload_kernels(int device)
{
    std::lock_guard<std::mutex> lk(g_preload_mu);
    if (device < 64 && (g_preloaded_devices >> device & 1)) return GFO_OK;
    if (const char* e = getenv("GFO_PRELOAD")) if (atoi(e) == 0) return GFO_OK;
    std::vector<const void*> ks;
    gfo_kernels_pyramid(ks); gfo_kernels_blur(ks); gfo_kernels_fast(ks); gfo_kernels_quadtree(ks); gfo_kernels_orient_desc(ks);
    gfo_kernels_stereo(ks); gfo_kernels_project(ks); gfo_kernels_bow(ks); gfo_kernels_api(ks);
    for (const void* k : ks) {
        hipFuncAttributes fa;
        const hipError_t e = hipFuncGetAttributes(&fa, k);
        if (e != hipSuccess) {
            (void)hipGetLastError();
            return fail(nullptr, GFO_ERR_DEVICE, "kernel %d of %d did not load on device %d: %s", (int)(&k - ks.data()), (int)ks.size(), device, hipGetErrorString(e));
        }
    }
    g_kernels_preloaded += (int)ks.size();
    if (device < 64) g_preloaded_devices |= 1ull << device;
    return GFO_OK;
}
extern "C" int gfo_kernels_preloaded(void) { return g_kernels_preloaded.load(); }

extern "C" int gfo_ctx_create(const gfo_params* p, int device, gfo_ctx** out)
{
    if (!p || !out) return fail(nullptr, GFO_ERR_INVALID, "null argument");
    *out = nullptr;
    if (p->nlevels < 1 || p->nlevels > GFO_MAX_LEVELS) return fail(nullptr, GFO_ERR_INVALID, "nlevels must be 1..%d", GFO_MAX_LEVELS);
    if (p->nfeatures < 1 || !(p->scale_factor > 1.0f)) return fail(nullptr, GFO_ERR_INVALID, "nfeatures >= 1 and scale_factor > 1 required");
    if (p->ini_th_fast < 1 || p->min_th_fast < 1 || p->ini_th_fast > 254 || p->min_th_fast > 254)
        return fail(nullptr, GFO_ERR_INVALID, "FAST thresholds must be in 1..254");
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev < 1)
        return fail(nullptr, GFO_ERR_DEVICE, "no HIP device available (this library has no CPU fallback)");
    if (device < 0 || device >= ndev) return fail(nullptr, GFO_ERR_INVALID, "device %d out of range (0..%d)", device, ndev - 1);
    hipDeviceProp_t prop;
    HIP_TRY(nullptr, hipGetDeviceProperties(&prop, device));
    if (strncmp(prop.gcnArchName, "gfx950", 6) != 0)
        return fail(nullptr, GFO_ERR_DEVICE, "device %d is %s; libgfo is built for gfx950 only", device, prop.gcnArchName);
    HIP_TRY(nullptr, hipSetDevice(device));
    if (const int prc = gfo_preload_kernels(device)) return prc;
    gfo_ctx* c = new gfo_ctx();
    c->prm = *p;
    if (c->prm.max_batch < 1) c->prm.max_batch = 1;
    c->device = device;
    if (hipStreamCreateWithFlags(&c->own_stream, hipStreamNonBlocking) != hipSuccess) {
        delete c;
        return fail(nullptr, GFO_ERR_DEVICE, "hipStreamCreate failed");
    }
    c->stream = c->own_stream;
    if (hipStreamCreateWithFlags(&c->side_stream, hipStreamNonBlocking) != hipSuccess ||
        hipEventCreateWithFlags(&c->ev_fork, hipEventDisableTiming) != hipSuccess ||
        hipEventCreateWithFlags(&c->ev_join, hipEventDisableTiming) != hipSuccess) {
        (void)hipGetLastError();
        c->fork_blur = false;   // extraction then simply runs in one stream
    }
    if (getenv("GFO_FORK_BLUR")) c->fork_blur = c->fork_blur && atoi(getenv("GFO_FORK_BLUR")) != 0;
    const char* dbg = getenv("GFO_DEBUG_SYNC");
    c->debug_sync = dbg && dbg[0] == '1';
    if (getenv("GFO_GRAPH")) c->graph_ok = atoi(getenv("GFO_GRAPH")) != 0;
    build_tables(c);
    {   // registered only now that nothing above can fail: gfo_ctx_chain / gfo_ctx_destroy dereference every entry
        std::lock_guard<std::mutex> lk(g_ctx_mu);
        g_ctx_live.push_back(c);
    }
    c->id = (uint64_t)++g_contexts_created;
    *out = c;
    return GFO_OK;
}

extern "C" int gfo_contexts_created(void) { return g_contexts_created.load(); }
extern "C" int gfo_arenas_planned(void) { return g_arenas_planned.load(); }
extern "C" uint64_t gfo_ctx_id(const gfo_ctx* c) { return c ? c->id : 0; }
extern "C" int gfo_vocabulary_nodes(const gfo_ctx* c) { return c && c->d_voc ? c->voc_nodes : 0; }

extern "C" void gfo_ctx_destroy(gfo_ctx* c)
{
    if (!c) return;
    {   // first: no other context may start a new wait on this one's events (gfo_ctx_chain)
        std::lock_guard<std::mutex> lk(g_ctx_mu);
        for (size_t i = 0; i < g_ctx_live.size(); i++)
            if (g_ctx_live[i] == c) { g_ctx_live.erase(g_ctx_live.begin() + i); break; }
        for (gfo_ctx* o : g_ctx_live)
            if (o->chain_after == c) o->chain_after = nullptr;
    }
    gfo_pair_release(c);     // a stereo rig this context was declared part of (gfo_ctx_pair)
    gfo_engine_release(c);   // the combiner's engine (and its batch contexts) goes with its last member
    (void)hipSetDevice(c->device);
    (void)hipStreamSynchronize(c->stream);
    prof_collect(c);
    for (hipEvent_t e : c->ev_pool) (void)hipEventDestroy(e);
    free_arena(c);
    if (c->d_scratch) (void)hipFree(c->d_scratch);
    if (c->d_voc) (void)hipFree(c->d_voc);
    if (c->h_in) (void)hipHostFree(c->h_in);
    if (c->h_out) (void)hipHostFree(c->h_out);
    if (c->h_min) (void)hipHostFree(c->h_min);
    if (c->h_mout) (void)hipHostFree(c->h_mout);
    if (c->pj.base) (void)hipFree(c->pj.base);
    if (c->d_pj_cand) (void)hipFree(c->d_pj_cand);
    if (c->d_map_desc) (void)hipFree(c->d_map_desc);
    if (c->side_stream) {
        (void)hipStreamSynchronize(c->side_stream);
        (void)hipStreamDestroy(c->side_stream);
    }
    if (c->copy_stream) {
        (void)hipStreamSynchronize(c->copy_stream);
        (void)hipStreamDestroy(c->copy_stream);
    }
    if (c->ev_results) (void)hipEventDestroy(c->ev_results);
    if (c->ev_delivered) (void)hipEventDestroy(c->ev_delivered);
    if (c->ev_pace) (void)hipEventDestroy(c->ev_pace);
    if (c->ev_fork) (void)hipEventDestroy(c->ev_fork);
    if (c->ev_join) (void)hipEventDestroy(c->ev_join);
    if (c->own_stream) (void)hipStreamDestroy(c->own_stream);
    delete c;
}

extern "C" int gfo_ctx_set_stream(gfo_ctx* c, void* s)
{
    if (!c) return GFO_ERR_INVALID;
    (void)hipSetDevice(c->device);
    (void)hipStreamSynchronize(c->stream);
    c->stream = s ? (hipStream_t)s : c->own_stream;
    return GFO_OK;
}

extern "C" int gfo_ctx_synchronize(gfo_ctx* c)
{
    if (!c) return GFO_ERR_INVALID;
    HIP_TRY(c, hipSetDevice(c->device));   // the calling thread may be on another device
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    return GFO_OK;
}

extern "C" int gfo_ctx_chain(gfo_ctx* c, gfo_ctx* after, int stage)
{
    if (!c) return GFO_ERR_INVALID;
    std::lock_guard<std::mutex> lk(g_ctx_mu);
    if (!after) {
        c->chain_after = nullptr;
        return GFO_OK;
    }
    if (after == c || stage < GFO_STAGE_PYRAMID || stage > GFO_STAGE_DESCRIPTORS) return fail(c, GFO_ERR_INVALID, "gfo_ctx_chain: bad context or stage");
    bool live = false;
    for (gfo_ctx* o : g_ctx_live) live = live || o == after;
    if (!live) return fail(c, GFO_ERR_INVALID, "gfo_ctx_chain: `after` is not a live context");
    if (after->device != c->device) return fail(c, GFO_ERR_INVALID, "gfo_ctx_chain: the two contexts are on different devices");
    if (!after->ev_pace) {
        (void)hipSetDevice(after->device);
        if (hipEventCreateWithFlags(&after->ev_pace, hipEventDisableTiming) != hipSuccess) {
            (void)hipGetLastError();
            after->ev_pace = nullptr;
            return fail(c, GFO_ERR_DEVICE, "gfo_ctx_chain: hipEventCreate failed");
        }
    }
    after->pace_stage = stage;
    c->chain_after = after;
    return GFO_OK;
}

extern "C" int gfo_ctx_tables(const gfo_ctx* c, float* scale, float* inv_scale, float* sigma2, float* inv_sigma2,
                              int32_t* quota)
{
    if (!c) return GFO_ERR_INVALID;
    const int n = c->prm.nlevels;
    if (scale) memcpy(scale, c->scale.data(), n * sizeof(float));
    if (inv_scale) memcpy(inv_scale, c->inv_scale.data(), n * sizeof(float));
    if (sigma2) memcpy(sigma2, c->sigma2.data(), n * sizeof(float));
    if (inv_sigma2) memcpy(inv_sigma2, c->inv_sigma2.data(), n * sizeof(float));
    if (quota) memcpy(quota, c->quota.data(), n * sizeof(int));
    return GFO_OK;
}

extern "C" int gfo_ctx_max_keypoints(const gfo_ctx* c)
{
    if (!c) return GFO_ERR_INVALID;
    if (c->planned) return c->g.kp_stride;
    int s = 0;
    for (int l = 0; l < c->prm.nlevels; l++) s += (c->quota[l] > 32 ? c->quota[l] : 32) + 8;
    return (int)align_up(s, 4);
}

// ---------------------------------------------------------------------------------------------
// extraction pipeline
// ---------------------------------------------------------------------------------------------
static int run_pyramid(gfo_ctx* c, const GfoInput& in, int nimg)
{
    // big levels: one launch each (they fill the chip); the small top levels: one fused launch, one workgroup
    // per image (each of them alone is latency-bound)
    static const int tail_px = getenv("GFO_RESIZE_TAIL_PX") ? atoi(getenv("GFO_RESIZE_TAIL_PX")) : 60000;  // measured: fusing levels of <= 60k px wins, larger ones lose
    // The banded form (one launch per level group, levels chained through LDS) for batches below 32 images.
    // From 64 images on the per-level kernels win IN THE PIPELINE although they are slower alone (127 vs 116 us per 128
    // images) and move more bytes: they hold no LDS and 64 registers, so the other contexts' kernels run beside them
    // (same-box A/B, stereo752: 64 images 190.3k -> 201.3k, 128 images 205.2k -> 214.0k, 256 images 210.5k -> 211.7k).
    // GFO_PYR_BAND_MIN_WG (workgroups a launch must have) and GFO_PYR_BAND_MAX_IMG let the tests force either path.
    const char* bm = getenv("GFO_PYR_BAND_MIN_WG");
    const int band_min_wg = bm ? atoi(bm) : 0;
    // (after the row-window change in k_resize the per-level form also wins at 32 images -- 179.5k vs 176k; for one frame
    //  or one stereo pair the two forms are within the run-to-run spread of the latency harness, +-4 %)
    static const int band_max_img = getenv("GFO_PYR_BAND_MAX_IMG") ? atoi(getenv("GFO_PYR_BAND_MAX_IMG")) : 32;
    if (c->g.pyr_nb > 0 && nimg * c->g.pyr_nb >= band_min_wg && nimg < band_max_img) {
        gfo_launch_pyramid_bands(c, in, nimg);
        c->last_in = in;
        c->last_nimg = nimg;
        c->have_pyramid = true;
        return GFO_OK;
    }
    int l = 1;
    for (; l < c->g.nlevels; l++) {
        if (c->g.lv[l].w * c->g.lv[l].h <= tail_px && nimg >= 32) break;
        gfo_launch_resize(c, in, l, nimg);
    }
    if (l < c->g.nlevels) gfo_launch_resize_tail(c, in, l, nimg);
    c->last_in = in;
    c->last_nimg = nimg;
    c->have_pyramid = true;
    return GFO_OK;
}

__global__ __launch_bounds__(256) void k_pack_results(GfoPack p)
{
    const int t = blockIdx.x * 256 + threadIdx.x, stride = gridDim.x * 256;
    for (int s = 0; s < p.nseg; s++)
        for (int i = t; i < p.n16[s]; i += stride) p.dst[s][i] = p.src[s][i];
}

// `bytes` (a multiple of 16, both pointers 16-byte aligned) from device-visible pinned host memory to device memory by a kernel of the
// compute queue instead of the copy engine: for a few hundred KB in front of a kernel chain the engine costs ~8 us of hand-over
// between the two queues on top of the copy (GfoXfer::up).
void gfo_launch_copy16(gfo_ctx* c, void* dst, const void* src, size_t bytes, hipStream_t st)
{
    GfoPack pk{};
    pk.src[0] = (const uint4*)src; pk.dst[0] = (uint4*)dst; pk.n16[0] = (int)(bytes / 16); pk.nseg = 1;
    const int blocks = (pk.n16[0] + 255) / 256;
    (void)c;   // (no per-stage timing events: a transfer is not a stage)
    hipLaunchKernelGGL(k_pack_results, dim3(blocks > 0 ? (blocks < 4096 ? blocks : 4096) : 1), dim3(256), 0, st, pk);
}

static GfoStereoLaunch stereo_batch_launch(gfo_ctx* c, const gfo_stereo_params& p)
{
    GfoStereoLaunch sl{};
    sl.kl = c->d_kp; sl.dl = c->d_desc;
    sl.kr = c->d_kp + c->g.kp_stride; sl.dr = c->d_desc + (size_t)c->g.kp_stride * 32;
    sl.cnt_dev = c->d_kp_cnt; sl.nl_host = 0; sl.nr_host = 0;
    sl.pair_stride_kp = 2LL * c->g.kp_stride; sl.npairs = c->last_nimg / 2;
    sl.d_scale = c->d_scale;
    sl.p = p;
    sl.min_d = nullptr; sl.max_d = nullptr; sl.win_stride = 0;
    sl.out = c->st; sl.out_stride = c->g.kp_stride;
    sl.sort = c->st_sort; sl.sort_stride = c->g.kp_stride;
    sl.window = gfo_stereo_window(c->scale.data(), c->g.nlevels);
    sl.nlevels = c->g.nlevels;
    return sl;
}

// the launches of one extraction (+ the stereo association of its pairs when sp is given), in stream order
static int extract_launches(gfo_ctx* c, const GfoInput& in, int nimg, const gfo_stereo_params* sp, const GfoPack* pack)
{
    // gfo_ctx_chain: start behind the stage event of the context this one is chained after (work already submitted there)
    // (under g_ctx_mu: gfo_ctx_destroy(after) on another thread clears the edge and frees the event under the same lock)
    if (c->chain_after) {
        std::lock_guard<std::mutex> lk(g_ctx_mu);
        if (gfo_ctx* a = c->chain_after)
            if (a->ev_pace && a->pace_recorded) HIP_TRY(c, hipStreamWaitEvent(c->stream, a->ev_pace, 0));
    }
#define GFO_PACE_POINT(S)                                                    \
    do {                                                                     \
        if (c->pace_stage == (S) && c->ev_pace) {                            \
            HIP_TRY(c, hipEventRecord(c->ev_pace, c->stream));               \
            c->pace_recorded = true;                                         \
        }                                                                    \
    } while (0)
    c->zero_cnt_pending = true;     // cleared by the first pyramid kernel; by a fill only when there is none (one level)
    run_pyramid(c, in, nimg);
    if (c->zero_cnt_pending) {
        c->zero_cnt_pending = false;
        HIP_TRY(c, hipMemsetAsync(c->d_cand_cnt, 0, sizeof(int) * nimg * c->g.nlevels * GFO_CNT_STRIDE, c->stream));
    }
    GFO_PACE_POINT(GFO_STAGE_PYRAMID);
    // fork: the blur (vector-pipe bound) next to FAST and the quadtree (the latter mostly barrier waits); join
    // before the descriptors.  Per-kernel profiling and debug runs keep everything in one stream.
    const bool fork = c->fork_blur && !c->profiling && !c->debug_sync;
    // per-frame batches: FAST, then quadtree and blur as ONE launch (k_quadtree_blur) -- no second stream, no fork / join events
    // (GFO_QT_FUSE_BLUR=0: the forked form)
    static const bool fuse_env = !(getenv("GFO_QT_FUSE_BLUR") && getenv("GFO_QT_FUSE_BLUR")[0] == '0');
    bool fused_done = false;
    if (fork && fuse_env && nimg <= gfo_few_max()) {
        gfo_launch_fast(c, in, nimg);
        GFO_PACE_POINT(GFO_STAGE_FAST);
        fused_done = gfo_launch_quadtree_blur(c, in, nimg);
        if (!fused_done) {      // (shapes that need the HBM-scratch quadtree: the forked form, FAST already launched)
            hipStream_t main_stream = c->stream;
            HIP_TRY(c, hipEventRecord(c->ev_fork, main_stream));
            HIP_TRY(c, hipStreamWaitEvent(c->side_stream, c->ev_fork, 0));
            gfo_launch_quadtree(c, nimg);
            c->stream = c->side_stream;
            gfo_launch_blur(c, in, nimg);
            c->stream = main_stream;
            HIP_TRY(c, hipEventRecord(c->ev_join, c->side_stream));
            HIP_TRY(c, hipStreamWaitEvent(c->stream, c->ev_join, 0));
        }
    } else if (fork) {
        // FAST first (it fills the vector pipes by itself); the fork point is after it, so that the blur shares the
        // chip with the quadtree, whose workgroups mostly wait at barriers
        hipStream_t main_stream = c->stream;
        gfo_launch_fast(c, in, nimg);
        GFO_PACE_POINT(GFO_STAGE_FAST);
        HIP_TRY(c, hipEventRecord(c->ev_fork, main_stream));
        HIP_TRY(c, hipStreamWaitEvent(c->side_stream, c->ev_fork, 0));
        gfo_launch_quadtree(c, nimg);
        c->stream = c->side_stream;
        gfo_launch_blur(c, in, nimg);
        c->stream = main_stream;
        HIP_TRY(c, hipEventRecord(c->ev_join, c->side_stream));
        HIP_TRY(c, hipStreamWaitEvent(c->stream, c->ev_join, 0));
    } else {
        // same order as the forked form (FAST reads the levels while the pyramid kernel's output is still cache-resident;
        // with the blur in between it measured 190 instead of 165 us), so that per-kernel profiles describe the real pipeline
        gfo_launch_fast(c, in, nimg);
        GFO_PACE_POINT(GFO_STAGE_FAST);
        gfo_launch_quadtree(c, nimg);
        gfo_launch_blur(c, in, nimg);
    }
    GFO_PACE_POINT(GFO_STAGE_SELECT);
    // nothing that consumes the selection may run if a stage before it was refused
    if (c->deliver_pending) {   // gfo_batch_deliver is still reading the previous batch's outputs
        HIP_TRY(c, hipStreamWaitEvent(c->stream, c->ev_delivered, 0));
        c->deliver_pending = false;
    }
    if (c->launch_err.empty()) gfo_launch_orient_desc(c, in, nimg);
    GFO_PACE_POINT(GFO_STAGE_DESCRIPTORS);
#undef GFO_PACE_POINT
    if (c->launch_err.empty() && sp) {
        c->last_nimg = nimg;
        GfoStereoLaunch sl = stereo_batch_launch(c, *sp);
        sl.cut_in_pack = pack && pack->cut_pairs > 0;      // the per-frame path: the pack kernel's first workgroups make the cut
        gfo_launch_stereo(c, sl);
    }
    if (c->launch_err.empty() && pack) {
        if (pack->cut_pairs > 0) {
            gfo_launch_pack_cut(c, *pack, c->stream);
        } else {
            int total = 0;
            for (int s_ = 0; s_ < pack->nseg; s_++) total += pack->n16[s_];
            GFO_LAUNCH(c, k_pack_results, dim3((total + 1023) / 1024 > 0 ? (total + 1023) / 1024 : 1), dim3(256), 0, c->stream, *pack);
        }
    }
    return GFO_OK;
}

// Small batches of host images (the per-frame path of the drop-in adapter) CAN replay the fixed launch sequence as a
// captured hipGraph (GFO_GRAPH=1): one submission instead of 8-11.  Off by default, for two measured reasons
// (MI355X, ROCm 7.2): (1) it buys nothing -- 0.2714 vs 0.2736 ms per stereo frame (profiles/latency_pair_r02*.json):
// the launches are already asynchronous and the kernels are 10-50 us each, so the host stays ahead of the GPU either
// way; (2) while ANY stream of the process is capturing -- in every capture mode, relaxed included -- a synchronous
// HIP call of ANOTHER host thread (hipMemcpy / hipMalloc, e.g. the other extractor thread of Frame.cc:84-87 planning
// its arena) fails with "operation failed due to a previous error during capture"
// (tests/test_gpu_extract.py::test_capture_in_one_thread_planning_in_another reproduces it when the graph is on).
// A library that other host threads share a process with cannot impose that.
static bool graph_key_eq(const gfo_ctx::GraphKey& a, const gfo_ctx::GraphKey& b)
{
    return a.base == b.base && a.pack_dst == b.pack_dst && a.pitch == b.pitch && a.img_stride == b.img_stride && a.nimg == b.nimg && a.stereo == b.stereo &&
           memcmp(&a.sp, &b.sp, sizeof a.sp) == 0 && a.plan_gen == b.plan_gen;
}

static int run_extract(gfo_ctx* c, const GfoInput& in, int nimg, const gfo_stereo_params* sp = nullptr, const GfoPack* pack = nullptr)
{
    static const int graph_max_img = getenv("GFO_GRAPH_MAX_IMAGES") ? atoi(getenv("GFO_GRAPH_MAX_IMAGES")) : 8;
    bool done = false;
    if (c->graph_ok && !c->profiling && !c->debug_sync && nimg <= graph_max_img && in.base == c->d_input && c->stream == c->own_stream &&
        !c->chain_after && !c->pace_stage) {   // (a chained context orders itself against another context's events: not capturable)
        gfo_ctx::GraphKey key{};
        key.base = in.base; key.pitch = in.pitch; key.img_stride = in.img_stride; key.nimg = nimg; key.stereo = sp ? 1 : 0;
        if (sp) key.sp = *sp;
        key.plan_gen = c->plan_gen;
        key.pack_dst = pack ? (const void*)pack->dst[0] : nullptr;
        if (c->graph_exec && !graph_key_eq(key, c->graph_key)) {
            (void)hipGraphExecDestroy(c->graph_exec);
            c->graph_exec = nullptr;
        }
        if (!c->graph_exec) {
            hipGraph_t graph = nullptr;
            if (hipStreamBeginCapture(c->stream, hipStreamCaptureModeRelaxed) == hipSuccess) {
                const int lrc = extract_launches(c, in, nimg, sp, pack);
                const hipError_t e = hipStreamEndCapture(c->stream, &graph);
                if (lrc == GFO_OK && e == hipSuccess && graph && c->launch_err.empty() &&
                    hipGraphInstantiate(&c->graph_exec, graph, nullptr, nullptr, 0) == hipSuccess) {
                    c->graph_key = key;
                    if (getenv("GFO_DEBUG_PLAN")) fprintf(stderr, "[gfo] captured the launch sequence of %d image(s)%s as a hipGraph\n", nimg, sp ? " + stereo association" : "");
                } else {
                    if (getenv("GFO_DEBUG_PLAN")) fprintf(stderr, "[gfo] launch-sequence capture failed (launches %d, end capture %s, launch error '%s'): plain launches from now on\n", lrc, hipGetErrorString(e), c->launch_err.c_str());
                    c->graph_exec = nullptr;
                    c->graph_ok = false;          // this runtime / configuration does not capture: plain launches from now on
                    c->launch_err.clear();
                    (void)hipGetLastError();
                }
                if (graph) (void)hipGraphDestroy(graph);
            } else {
                (void)hipGetLastError();
                c->graph_ok = false;
            }
        }
        if (c->graph_exec) {
            if (hipGraphLaunch(c->graph_exec, c->stream) == hipSuccess) done = true;
            else {
                (void)hipGetLastError();
                c->graph_ok = false;
            }
        }
    }
    if (!done) {
        const int lrc = extract_launches(c, in, nimg, sp, pack);
        if (lrc) return lrc;
    }
    if (!c->launch_err.empty()) {
        c->have_batch = false;
        return gfo_take_launch_err(c);
    }
    HIP_TRY(c, hipGetLastError());
    c->flags_snapshot = false;      // a new batch: its flags are in the live word
    c->last_in = in;
    c->last_nimg = nimg;
    c->have_pyramid = true;
    c->have_batch = true;
    c->have_stereo = sp != nullptr;
    c->have_projection = false;
    return GFO_OK;
}

static int check_flags(gfo_ctx* c)
{
    int f[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    HIP_TRY(c, hipMemcpyAsync(f, c->d_flags, sizeof f, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    if (c->flags_snapshot) f[0] |= f[4];    // gfo_batch_deliver moved this batch's flags into the snapshot word
#ifdef GFO_FAST_DEBUG
    if (getenv("GFO_FAST_STOP") && atoi(getenv("GFO_FAST_STOP")) == 9) {
        fprintf(stderr, "[gfo] FAST selectivity: scan px %d, after compass %d, after pair test %d\n", f[1], f[2], f[3]);
        (void)hipMemsetAsync(c->d_flags, 0, sizeof f, c->stream);
    }
#endif
    if (f[0]) {
        (void)hipMemsetAsync(c->d_flags, 0, sizeof f, c->stream);
        c->flags_snapshot = false;
        return fail(c, GFO_ERR_OVERFLOW, "internal buffer overflow (flags 0x%x: 1 candidates, 2 quadtree nodes, 4 selection, 8 keypoints)", f[0]);
    }
    return GFO_OK;
}

static int upload_images(gfo_ctx* c, const uint8_t* const* imgs, int nimg, int w, int h, int stride, GfoInput* in)
{
    const int pitch = c->g.lv[0].pitch;
    for (int i = 0; i < nimg; i++)
        HIP_TRY(c, hipMemcpy2DAsync(c->d_input + (size_t)i * pitch * h, pitch, imgs[i], stride, w, h,
                                    hipMemcpyHostToDevice, c->stream));
    in->base = c->d_input;
    in->pitch = pitch;
    in->img_stride = (long long)pitch * h;
    return GFO_OK;
}

extern "C" int gfo_extract_batch_device(gfo_ctx* c, const uint8_t* d_imgs, int nimg, int w, int h, size_t pitch,
                                        size_t img_stride)
{
    if (!c || !d_imgs || nimg < 1 || w < 1 || h < 1 || pitch < (size_t)w) return fail(c, GFO_ERR_INVALID, "bad argument");
    if (nimg > 1 && img_stride < pitch * (size_t)h) return fail(c, GFO_ERR_INVALID, "img_stride %zu < pitch * h = %zu: images overlap", img_stride, pitch * (size_t)h);
    HIP_TRY(c, hipSetDevice(c->device));
    int rc = gfo_plan(c, w, h, nimg);
    if (rc) return rc;
    GfoInput in{d_imgs, (long long)pitch, (long long)img_stride};
    return run_extract(c, in, nimg);
}

extern "C" int gfo_batch_counts(gfo_ctx* c, int* n, int* per_level)
{
    if (!c || !n) return GFO_ERR_INVALID;
    if (!c->have_batch) return fail(c, GFO_ERR_STATE, "no batch has been extracted");
    HIP_TRY(c, hipSetDevice(c->device));   // the calling thread may be on another device
    int rc = check_flags(c);
    if (rc) return rc;
    HIP_TRY(c, hipMemcpyAsync(n, c->d_kp_cnt, sizeof(int) * c->last_nimg, hipMemcpyDeviceToHost, c->stream));
    if (per_level)
        HIP_TRY(c, hipMemcpyAsync(per_level, c->d_sel_cnt, sizeof(int) * c->last_nimg * c->g.nlevels, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    return GFO_OK;
}

extern "C" int gfo_batch_fetch(gfo_ctx* c, int image, gfo_keypoint* kp, uint8_t* desc, int cap, int* n)
{
    if (!c || !n) return GFO_ERR_INVALID;
    if (!c->have_batch) return fail(c, GFO_ERR_STATE, "no batch has been extracted");
    HIP_TRY(c, hipSetDevice(c->device));   // the calling thread may be on another device
    if (image < 0 || image >= c->last_nimg) return fail(c, GFO_ERR_INVALID, "image %d out of range", image);
    int rc = check_flags(c);
    if (rc) return rc;
    int cnt = 0;
    HIP_TRY(c, hipMemcpyAsync(&cnt, c->d_kp_cnt + image, sizeof(int), hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    *n = cnt;
    const int m = cnt < cap ? cnt : cap;
    if (m > 0 && kp)
        HIP_TRY(c, hipMemcpyAsync(kp, c->d_kp + (size_t)image * c->g.kp_stride, sizeof(gfo_keypoint) * m, hipMemcpyDeviceToHost, c->stream));
    if (m > 0 && desc)
        HIP_TRY(c, hipMemcpyAsync(desc, c->d_desc + (size_t)image * c->g.kp_stride * 32, 32 * (size_t)m, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    return cnt > cap ? fail(c, GFO_ERR_CAPACITY, "%d keypoints, caller capacity %d", cnt, cap) : GFO_OK;
}

extern "C" int gfo_batch_device_views(gfo_ctx* c, const gfo_keypoint** d_kp, const uint8_t** d_desc,
                                      const int32_t** d_counts, int* kp_stride)
{
    if (!c) return GFO_ERR_INVALID;
    if (!c->have_batch) return fail(c, GFO_ERR_STATE, "no batch has been extracted");
    if (d_kp) *d_kp = c->d_kp;
    if (d_desc) *d_desc = c->d_desc;
    if (d_counts) *d_counts = c->d_kp_cnt;
    if (kp_stride) *kp_stride = c->g.kp_stride;
    return GFO_OK;
}

extern "C" int gfo_batch_deliver(gfo_ctx* c, void* host_dst, size_t host_bytes, gfo_delivery* L)
{
    if (!c || !L) return GFO_ERR_INVALID;
    if (!c->have_batch) return fail(c, GFO_ERR_STATE, "no batch has been extracted");
    const int nimg = c->last_nimg, ks = c->g.kp_stride, np = nimg / 2;
    const bool stereo = c->have_stereo;
    size_t off = 0;
    auto take = [&](size_t bytes) { size_t o = off; off = (size_t)align_up((long long)(off + bytes), 256); return o; };
    L->nimg = nimg; L->kp_stride = ks; L->stereo = stereo ? 1 : 0;
    L->off_flags = take(16); L->off_counts = take(4 * (size_t)nimg); L->off_kp = take(sizeof(gfo_keypoint) * (size_t)ks * nimg);
    L->off_desc = take(32 * (size_t)ks * nimg);
    L->off_u_right = take(stereo ? 4 * (size_t)ks * np : 0); L->off_depth = take(stereo ? 4 * (size_t)ks * np : 0);
    L->off_best_dist = take(stereo ? 4 * (size_t)ks * np : 0); L->off_best_idx = take(stereo ? 4 * (size_t)ks * np : 0);
    L->off_nmatched = take(stereo ? 4 * (size_t)np : 0);
    L->bytes = off;
    if (!host_dst) return GFO_OK;
    if (host_bytes < off) return fail(c, GFO_ERR_CAPACITY, "delivery block of %zu bytes, %zu needed", host_bytes, off);
    HIP_TRY(c, hipSetDevice(c->device));
    if (!c->copy_stream) {
        HIP_TRY(c, hipStreamCreateWithFlags(&c->copy_stream, hipStreamNonBlocking));
        HIP_TRY(c, hipEventCreateWithFlags(&c->ev_results, hipEventDisableTiming));
        HIP_TRY(c, hipEventCreateWithFlags(&c->ev_delivered, hipEventDisableTiming));
    }
    uint8_t* H = (uint8_t*)host_dst;
    hipStream_t cs = c->copy_stream;
    // The overflow flags of THIS batch, frozen on the main stream before the copy stream may run beside the next extraction
    // (ADVICE r3): that extraction's k_fast / k_quadtree OR into the live word while this batch is still being delivered, and the
    // live word was never cleared on this path -- batch N could report batch N+1's overflow, and one overflow stuck to every later
    // delivery.  Snapshot, then clear; the snapshot is what is delivered.  (The next snapshot is written after the next
    // extraction's k_orient_desc, which waits for ev_delivered: the copy stream has read this one by then.)
    HIP_TRY(c, hipMemcpyAsync(c->d_flags + 4, c->d_flags, 16, hipMemcpyDeviceToDevice, c->stream));
    HIP_TRY(c, hipMemsetAsync(c->d_flags, 0, 16, c->stream));
    c->flags_snapshot = true;
    HIP_TRY(c, hipEventRecord(c->ev_results, c->stream));
    HIP_TRY(c, hipStreamWaitEvent(cs, c->ev_results, 0));
    // Pinned memory the device can address (hipHostMalloc, a pinned torch tensor, hipHostRegister'ed pages) is written by
    // ONE kernel through its device mapping: stores travel up the link while the DMA engine brings the next batch down --
    // D2H copies would queue on that same engine behind the 46-MB input copy (measured, 128 images per step: 93 k frames/s
    // delivered against 124 k with the input copy alone).  Anything else falls back to D2H copies.  GFO_DELIVER_DMA=1 forces them.
    static const bool force_dma = getenv("GFO_DELIVER_DMA") && atoi(getenv("GFO_DELIVER_DMA")) != 0;
    void* dmap = nullptr;
    if (!force_dma && (off & 15) == 0 && ((uintptr_t)host_dst & 15) == 0 && hipHostGetDevicePointer(&dmap, host_dst, 0) == hipSuccess && dmap) {
        uint8_t* D = (uint8_t*)dmap;
        GfoPack pk{};
        auto seg = [&](const void* src, size_t dst_off, size_t bytes) {
            pk.src[pk.nseg] = (const uint4*)src; pk.dst[pk.nseg] = (uint4*)(D + dst_off); pk.n16[pk.nseg] = (int)((bytes + 15) / 16); pk.nseg++;
        };
        seg(c->d_flags + 4, L->off_flags, 16);
        seg(c->d_kp_cnt, L->off_counts, 4 * (size_t)nimg);
        seg(c->d_kp, L->off_kp, sizeof(gfo_keypoint) * (size_t)ks * nimg);
        seg(c->d_desc, L->off_desc, 32 * (size_t)ks * nimg);
        if (stereo) {
            seg(c->st.u_right, L->off_u_right, 4 * (size_t)ks * np);
            seg(c->st.depth, L->off_depth, 4 * (size_t)ks * np);
            seg(c->st.best_dist, L->off_best_dist, 4 * (size_t)ks * np);
            seg(c->st.best_idx, L->off_best_idx, 4 * (size_t)ks * np);
            seg(c->st.nmatched, L->off_nmatched, 4 * (size_t)np);
        }
        int total = 0;
        for (int s_ = 0; s_ < pk.nseg; s_++) total += pk.n16[s_];
        const int blocks = (total + 1023) / 1024;   // (16 ... 2048 workgroups deliver at the same rate: 18 MB in 0.34 ms = 53 GB/s, the link's)
        hipLaunchKernelGGL(k_pack_results, dim3(blocks > 0 ? (blocks < 2048 ? blocks : 2048) : 1), dim3(256), 0, cs, pk);
        HIP_TRY(c, hipGetLastError());
    } else {
        (void)hipGetLastError();
        HIP_TRY(c, hipMemcpyAsync(H + L->off_flags, c->d_flags + 4, 16, hipMemcpyDeviceToHost, cs));
        HIP_TRY(c, hipMemcpyAsync(H + L->off_counts, c->d_kp_cnt, 4 * (size_t)nimg, hipMemcpyDeviceToHost, cs));
        HIP_TRY(c, hipMemcpyAsync(H + L->off_kp, c->d_kp, sizeof(gfo_keypoint) * (size_t)ks * nimg, hipMemcpyDeviceToHost, cs));
        HIP_TRY(c, hipMemcpyAsync(H + L->off_desc, c->d_desc, 32 * (size_t)ks * nimg, hipMemcpyDeviceToHost, cs));
        if (stereo) {
            HIP_TRY(c, hipMemcpyAsync(H + L->off_u_right, c->st.u_right, 4 * (size_t)ks * np, hipMemcpyDeviceToHost, cs));
            HIP_TRY(c, hipMemcpyAsync(H + L->off_depth, c->st.depth, 4 * (size_t)ks * np, hipMemcpyDeviceToHost, cs));
            HIP_TRY(c, hipMemcpyAsync(H + L->off_best_dist, c->st.best_dist, 4 * (size_t)ks * np, hipMemcpyDeviceToHost, cs));
            HIP_TRY(c, hipMemcpyAsync(H + L->off_best_idx, c->st.best_idx, 4 * (size_t)ks * np, hipMemcpyDeviceToHost, cs));
            HIP_TRY(c, hipMemcpyAsync(H + L->off_nmatched, c->st.nmatched, 4 * (size_t)np, hipMemcpyDeviceToHost, cs));
        }
    }
    HIP_TRY(c, hipEventRecord(c->ev_delivered, cs));
    c->deliver_pending = true;
    return GFO_OK;
}

extern "C" int gfo_deliver_wait(gfo_ctx* c)
{
    if (!c) return GFO_ERR_INVALID;
    if (!c->copy_stream) return GFO_OK;
    HIP_TRY(c, hipSetDevice(c->device));
    HIP_TRY(c, hipStreamSynchronize(c->copy_stream));
    return GFO_OK;
}

// grow-only pinned host buffers of the latency path
int gfo_pinned(gfo_ctx* c, uint8_t** buf, size_t* cap, size_t bytes)
{
    if (bytes <= *cap) return GFO_OK;
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    if (*buf) (void)hipHostFree(*buf);
    *buf = nullptr;
    *cap = 0;
    HIP_TRY(c, hipHostMalloc((void**)buf, bytes, hipHostMallocDefault));
    *cap = bytes;
    return GFO_OK;
}

#define GFO_SMALL_BATCH 8   // up to this many host images go through pinned staging: one H2D, one D2H, one sync

// Small host batch: images -> pinned -> one H2D; kernels; every result -> pinned by ONE kernel (k_pack_results);
// ONE stream synchronisation; then plain memcpy into the caller's arrays.  In four steps so that the frame combiner
// (gfo_combine.hip) can let every caller stage and collect its own frame while one of them submits the batch:
//   gfo_small_prepare   pinned buffers + the result layout for up to nimg_cap images
//   gfo_small_upload    images -> (pinned staging ->) the device input, in as few DMA copies as possible
//   gfo_small_submit    the launches, the pack kernel, the synchronisation
//   gfo_small_collect   image i's keypoints / descriptors (and pair i/2's stereo outputs) -> the caller's arrays
// ranges the caller has pinned (gfo_host_register): images inside one need no staging copy
static std::mutex g_pin_mu;
static std::vector<std::pair<const uint8_t*, const uint8_t*>> g_pinned;

extern "C" int gfo_host_register(void* p, size_t bytes)
{
    if (!p || bytes == 0) return GFO_ERR_INVALID;
    if (hipHostRegister(p, bytes, hipHostRegisterDefault) != hipSuccess) {
        (void)hipGetLastError();
        return fail(nullptr, GFO_ERR_DEVICE, "hipHostRegister of %zu bytes failed", bytes);
    }
    std::lock_guard<std::mutex> lk(g_pin_mu);
    g_pinned.push_back({(const uint8_t*)p, (const uint8_t*)p + bytes});
    return GFO_OK;
}

extern "C" int gfo_host_unregister(void* p)
{
    {
        std::lock_guard<std::mutex> lk(g_pin_mu);
        size_t i = 0;
        for (; i < g_pinned.size(); i++)
            if (g_pinned[i].first == (const uint8_t*)p) break;
        if (i == g_pinned.size()) return GFO_ERR_INVALID;
        g_pinned.erase(g_pinned.begin() + i);
    }
    if (hipHostUnregister(p) != hipSuccess) {
        (void)hipGetLastError();
        return GFO_ERR_DEVICE;
    }
    return GFO_OK;
}

// pinned memory of the library's own that images are staged in (the stereo rigs of gfo_combine.hip): same registry
void gfo_note_pinned(const uint8_t* p, size_t bytes, bool add)
{
    std::lock_guard<std::mutex> lk(g_pin_mu);
    if (add) {
        g_pinned.push_back({p, p + bytes});
        return;
    }
    for (size_t i = 0; i < g_pinned.size(); i++)
        if (g_pinned[i].first == p) { g_pinned.erase(g_pinned.begin() + i); break; }
}

static bool host_pinned(const uint8_t* p, size_t bytes)
{
    std::lock_guard<std::mutex> lk(g_pin_mu);
    for (const auto& r : g_pinned)
        if (p >= r.first && p + bytes <= r.second) return true;
    return false;
}

int gfo_small_prepare(gfo_ctx* c, int nimg_cap, GfoSmallLayout* L)
{
    const int ks = c->g.kp_stride, h = c->g.h0, w = c->g.w0;
    L->nimg_cap = nimg_cap;
    L->pitch = (w & 15) == 0 ? w : c->g.lv[0].pitch;    // 752, 640, 1920, ...: tight rows, the caller's own layout
    L->img_bytes = (size_t)L->pitch * h;
    int rc = gfo_pinned(c, &c->h_in, &c->h_in_bytes, L->img_bytes * nimg_cap);
    if (rc) return rc;
    // result layout in the pinned buffer (fixed for a planned geometry and capacity)
    size_t off = 0;
    auto take = [&](size_t bytes) { size_t o = off; off = (size_t)align_up((long long)(off + bytes), 64); return o; };
    const size_t npair = (size_t)(nimg_cap + 1) / 2;
    L->o_fl = take(16); L->o_cnt = take(16 * (size_t)((nimg_cap + 3) / 4)); L->o_kp = take(sizeof(gfo_keypoint) * (size_t)ks * nimg_cap);
    L->o_ds = take(32 * (size_t)ks * nimg_cap);
    L->o_ur = take(4 * (size_t)ks * npair); L->o_dp = take(4 * (size_t)ks * npair); L->o_bd = take(4 * (size_t)ks * npair);
    L->o_bi = take(4 * (size_t)ks * npair); L->o_nm = take(16 * ((npair + 3) / 4));
    return gfo_pinned(c, &c->h_out, &c->h_out_bytes, off);
}

// Images `first .. first + count` of the small batch on their way to the device, on stream `st`, in as few DMA copies as
// the caller's memory allows (a copy costs the engine ~10 us whatever its size):
//   every image pinned by the caller (gfo_host_register), tight rows, one behind the other in memory -> ONE copy, no staging;
//   pinned but apart -> one copy each, no staging;
//   otherwise -> memcpy into the context's pinned staging (one memcpy per image when the rows are tight), ONE copy.
// `ec` takes the error message: the context of the CALLING thread (in the combiner several callers upload into one batch
// context at once -- its error string is not theirs to write).
// lone_caller: nobody else is feeding this device through the library right now as far as the caller can tell (a context used on its
// own; in the combiner, an engine that serves ONE camera).  Then a staged upload of up to GFO_UPLOAD_KERNEL_MAX bytes (512 KB: one
// image of a mono camera) comes over by a copy kernel of the compute queue instead of the copy engine: -4 to -6 us per frame (same-box
// A/B, profiles/NOTEBOOK.md "Round 5"); a stereo frame's 722 KB gain nothing, and with several cameras the engine's copies run
// underneath the other batches' kernels where a copy kernel competes with them (-17 % at K = 8).
int gfo_small_upload(gfo_ctx* c, gfo_ctx* ec, const GfoSmallLayout& L, int first, int count, const uint8_t* const* imgs, int w, int h, int stride,
                     hipStream_t st, bool lone_caller)
{
    uint8_t* dst = c->d_input + (size_t)first * L.img_bytes;
    const bool tight = stride == L.pitch;
    bool all_pinned = tight;
    for (int i = 0; i < count && all_pinned; i++) all_pinned = host_pinned(imgs[i], L.img_bytes);
    if (all_pinned) {
        bool contiguous = true;
        for (int i = 1; i < count; i++) contiguous = contiguous && imgs[i] == imgs[i - 1] + L.img_bytes;
        if (contiguous) {
            HIP_TRY(ec, hipMemcpyAsync(dst, imgs[0], L.img_bytes * count, hipMemcpyHostToDevice, st));
        } else {
            for (int i = 0; i < count; i++) HIP_TRY(ec, hipMemcpyAsync(dst + (size_t)i * L.img_bytes, imgs[i], L.img_bytes, hipMemcpyHostToDevice, st));
        }
        return GFO_OK;
    }
    uint8_t* stage = c->h_in + (size_t)first * L.img_bytes;
    for (int i = 0; i < count; i++) {
        uint8_t* d = stage + (size_t)i * L.img_bytes;
        if (tight) memcpy(d, imgs[i], L.img_bytes);
        else
            for (int y = 0; y < h; y++) memcpy(d + (size_t)y * L.pitch, imgs[i] + (size_t)y * stride, w);
    }
    static const long kernel_max = getenv("GFO_UPLOAD_KERNEL_MAX") ? atol(getenv("GFO_UPLOAD_KERNEL_MAX")) : 512 * 1024;
    if (lone_caller && (long)(L.img_bytes * count) <= kernel_max && (L.img_bytes & 15) == 0) gfo_launch_copy16(c, dst, stage, L.img_bytes * count, st);
    else HIP_TRY(ec, hipMemcpyAsync(dst, stage, L.img_bytes * count, hipMemcpyHostToDevice, st));
    return GFO_OK;
}

int gfo_small_submit(gfo_ctx* c, const GfoSmallLayout& L, int nimg, const gfo_stereo_params* sp, bool copy_in)
{
    const int pitch = L.pitch, ks = c->g.kp_stride;
    uint8_t* H = c->h_out;
    hipStream_t st = c->stream;
    GfoPack pk{};
    auto seg = [&](const void* src, size_t dst_off, size_t bytes) {
        pk.src[pk.nseg] = (const uint4*)src; pk.dst[pk.nseg] = (uint4*)(H + dst_off); pk.n16[pk.nseg] = (int)((bytes + 15) / 16); pk.nseg++;
    };
    const size_t npair = (size_t)nimg / 2;
    seg(c->d_flags, L.o_fl, 16);
    seg(c->d_kp_cnt, L.o_cnt, 4 * (size_t)nimg);          // the count vector is allocated in 16-byte multiples (plan)
    seg(c->d_kp, L.o_kp, sizeof(gfo_keypoint) * (size_t)ks * nimg);
    seg(c->d_desc, L.o_ds, 32 * (size_t)ks * nimg);
    if (sp) {
        // u_right / depth / nmatched reach the host block from the cut itself, made by the pack kernel's first workgroups
        // (GFO_STEREO_CUT_IN_PACK=0: k_stereo_cut as a launch of its own, then plain copies)
        static const bool cut_in_pack = !(getenv("GFO_STEREO_CUT_IN_PACK") && getenv("GFO_STEREO_CUT_IN_PACK")[0] == '0');
        if (cut_in_pack && npair > 0) {
            pk.cut_pairs = (int)npair; pk.cut_cnt_dev = c->d_kp_cnt; pk.cut_nl_host = 0; pk.cut_out = c->st; pk.cut_out_stride = ks;
            pk.h_u_right = reinterpret_cast<float*>(H + L.o_ur); pk.h_depth = reinterpret_cast<float*>(H + L.o_dp);
            pk.h_nmatched = reinterpret_cast<int*>(H + L.o_nm);
        } else {
            seg(c->st.u_right, L.o_ur, 4 * (size_t)ks * npair);
            seg(c->st.depth, L.o_dp, 4 * (size_t)ks * npair);
            seg(c->st.nmatched, L.o_nm, 4 * npair);             // allocated with 16 bytes of slack (plan)
        }
        seg(c->st.best_dist, L.o_bd, 4 * (size_t)ks * npair);
        seg(c->st.best_idx, L.o_bi, 4 * (size_t)ks * npair);
    }
    (void)copy_in;   // the images are on their way already (gfo_small_upload)
    GfoInput in{c->d_input, pitch, (long long)L.img_bytes};
    int rc = run_extract(c, in, nimg, sp, &pk);
    if (rc) return rc;
    HIP_TRY(c, hipStreamSynchronize(st));
    const int* fl = reinterpret_cast<const int*>(H + L.o_fl);
    if (fl[0]) {
        (void)hipMemsetAsync(c->d_flags, 0, 16, st);
        return fail(c, GFO_ERR_OVERFLOW, "internal buffer overflow (flags 0x%x: 1 candidates, 2 quadtree nodes, 4 selection, 8 keypoints)", fl[0]);
    }
    return GFO_OK;
}

// returns 1 when the image produced more keypoints than `cap` (outputs truncated), else 0
int gfo_small_collect(gfo_ctx* c, const GfoSmallLayout& L, int i, gfo_keypoint* kp, uint8_t* desc, int cap, int* n)
{
    const int ks = c->g.kp_stride;
    const uint8_t* H = c->h_out;
    *n = reinterpret_cast<const int*>(H + L.o_cnt)[i];
    const int m = *n < cap ? *n : cap;
    if (m > 0 && kp) memcpy(kp, H + L.o_kp + sizeof(gfo_keypoint) * (size_t)ks * i, sizeof(gfo_keypoint) * (size_t)m);
    if (m > 0 && desc) memcpy(desc, H + L.o_ds + 32 * (size_t)ks * i, 32 * (size_t)m);
    return *n > cap;
}

void gfo_small_collect_stereo(gfo_ctx* c, const GfoSmallLayout& L, int pair, int n_left, int cap, float* u_right, float* depth,
                              int32_t* best_dist, int32_t* best_idx_r, int* nmatched)
{
    const int ks = c->g.kp_stride;
    const uint8_t* H = c->h_out;
    const int m = n_left < cap ? n_left : cap;
    const size_t o = 4 * (size_t)ks * pair;
    if (m > 0) {
        memcpy(u_right, H + L.o_ur + o, 4 * (size_t)m);
        memcpy(depth, H + L.o_dp + o, 4 * (size_t)m);
        if (best_dist) memcpy(best_dist, H + L.o_bd + o, 4 * (size_t)m);
        if (best_idx_r) memcpy(best_idx_r, H + L.o_bi + o, 4 * (size_t)m);
    }
    *nmatched = reinterpret_cast<const int*>(H + L.o_nm)[pair];
}

GfoPairBlock gfo_pair_block(int ks)
{
    GfoPairBlock b;
    size_t off = 16;
    auto take = [&](size_t bytes) { size_t o = off; off = (size_t)align_up((long long)(off + bytes), 16); return o; };
    b.o_kl = take(sizeof(gfo_keypoint) * (size_t)ks); b.o_dl = take(32 * (size_t)ks);
    b.o_kr = take(sizeof(gfo_keypoint) * (size_t)ks); b.o_dr = take(32 * (size_t)ks);
    b.o_min = take(4 * (size_t)ks); b.o_max = take(4 * (size_t)ks);
    b.bytes = (size_t)align_up((long long)off, 256);
    return b;
}

// pair k's staged block -> the arena's batch layout (images 2k, 2k + 1 = its left / right keypoints and descriptors, counts)
__global__ __launch_bounds__(256) void k_unpack_pairs(const uint8_t* __restrict__ stage, GfoPairBlock b, gfo_keypoint* __restrict__ d_kp,
                                                      uint8_t* __restrict__ d_desc, int* __restrict__ d_cnt, int ks)
{
    const int pair = blockIdx.y;
    const uint8_t* S = stage + (size_t)pair * b.bytes;
    const int nl = reinterpret_cast<const int*>(S)[0], nr = reinterpret_cast<const int*>(S)[1];
    if (blockIdx.x == 0 && threadIdx.x == 0) { d_cnt[2 * pair] = nl; d_cnt[2 * pair + 1] = nr; }
    const int t = blockIdx.x * 256 + threadIdx.x, stride = gridDim.x * 256;
    // 16-byte units; the keypoint arrays are 28-byte records: nl * 28 bytes rounded up stays inside the ks-sized part
    uint4* kpl = reinterpret_cast<uint4*>(d_kp + (size_t)(2 * pair) * ks);
    uint4* kpr = reinterpret_cast<uint4*>(d_kp + (size_t)(2 * pair + 1) * ks);
    uint4* dl = reinterpret_cast<uint4*>(d_desc + (size_t)(2 * pair) * ks * 32);
    uint4* dr = reinterpret_cast<uint4*>(d_desc + (size_t)(2 * pair + 1) * ks * 32);
    const uint4* skl = reinterpret_cast<const uint4*>(S + b.o_kl); const uint4* skr = reinterpret_cast<const uint4*>(S + b.o_kr);
    const uint4* sdl = reinterpret_cast<const uint4*>(S + b.o_dl); const uint4* sdr = reinterpret_cast<const uint4*>(S + b.o_dr);
    for (int i = t; i < (nl * 28 + 15) / 16; i += stride) kpl[i] = skl[i];
    for (int i = t; i < (nr * 28 + 15) / 16; i += stride) kpr[i] = skr[i];
    for (int i = t; i < 2 * nl; i += stride) dl[i] = sdl[i];
    for (int i = t; i < 2 * nr; i += stride) dr[i] = sdr[i];
}

// The frame combiner's host-array stereo batch: the pairs' blocks are in d_stage (each joiner copied its own); unpack, the
// three association kernels over all pairs, the results into the pinned result buffer, one synchronisation.
int gfo_small_submit_pairs(gfo_ctx* c, const GfoSmallLayout& L, int npairs, const gfo_stereo_params* sp, const uint8_t* d_stage)
{
    const int ks = c->g.kp_stride;
    const GfoPairBlock b = gfo_pair_block(ks);
    hipStream_t st = c->stream;
    hipLaunchKernelGGL(k_unpack_pairs, dim3(8, npairs), dim3(256), 0, st, d_stage, b, c->d_kp, c->d_desc, c->d_kp_cnt, ks);
    HIP_TRY(c, hipGetLastError());
    c->last_nimg = 2 * npairs;
    GfoStereoLaunch sl = stereo_batch_launch(c, *sp);
    sl.min_d = reinterpret_cast<const float*>(d_stage + b.o_min);
    sl.max_d = reinterpret_cast<const float*>(d_stage + b.o_max);
    sl.win_stride = (long long)(b.bytes / 4);
    sl.cut_in_pack = true;            // the pack kernel below makes the cut
    gfo_launch_stereo(c, sl);
    if (int lrc = gfo_take_launch_err(c)) return lrc;
    uint8_t* H = c->h_out;
    GfoPack pk{};
    auto seg = [&](const void* src, size_t dst_off, size_t bytes) {
        pk.src[pk.nseg] = (const uint4*)src; pk.dst[pk.nseg] = (uint4*)(H + dst_off); pk.n16[pk.nseg] = (int)((bytes + 15) / 16); pk.nseg++;
    };
    pk.cut_pairs = npairs; pk.cut_cnt_dev = c->d_kp_cnt; pk.cut_nl_host = 0; pk.cut_out = c->st; pk.cut_out_stride = ks;
    pk.h_u_right = reinterpret_cast<float*>(H + L.o_ur); pk.h_depth = reinterpret_cast<float*>(H + L.o_dp);
    pk.h_nmatched = reinterpret_cast<int*>(H + L.o_nm);
    seg(c->st.best_dist, L.o_bd, 4 * (size_t)ks * npairs);
    seg(c->st.best_idx, L.o_bi, 4 * (size_t)ks * npairs);
    gfo_launch_pack_cut(c, pk, st);
    if (int lrc = gfo_take_launch_err(c)) return lrc;
    HIP_TRY(c, hipGetLastError());
    HIP_TRY(c, hipStreamSynchronize(st));
    c->have_batch = c->have_stereo = false;     // the arena holds unpacked host arrays, not an extraction
    return GFO_OK;
}

static int extract_small(gfo_ctx* c, const uint8_t* const* imgs, int nimg, int w, int h, int stride, const gfo_stereo_params* sp,
                         gfo_keypoint* const* kp, uint8_t* const* desc, int cap, int* n, float* u_right, float* depth,
                         int32_t* best_dist, int32_t* best_idx_r, int* nmatched)
{
    GfoSmallLayout L;
    int rc = gfo_small_prepare(c, nimg, &L);
    if (rc) return rc;
    rc = gfo_small_upload(c, c, L, 0, nimg, imgs, w, h, stride, c->stream, true);   // a context used on its own (no combiner)
    if (rc) return rc;
    rc = gfo_small_submit(c, L, nimg, sp, false);
    if (rc) return rc;
    int over = 0;
    for (int i = 0; i < nimg; i++) over |= gfo_small_collect(c, L, i, kp[i], desc[i], cap, &n[i]);
    if (sp) gfo_small_collect_stereo(c, L, 0, n[0], cap, u_right, depth, best_dist, best_idx_r, nmatched);
    return over ? fail(c, GFO_ERR_CAPACITY, "an image produced more keypoints than the caller capacity %d", cap) : GFO_OK;
}

extern "C" int gfo_extract_batch(gfo_ctx* c, const uint8_t* const* imgs, int nimg, int w, int h, int stride,
                                 gfo_keypoint* kp, uint8_t* desc, int cap, int* n)
{
    if (!c || !n) return GFO_ERR_INVALID;
    if (!imgs || nimg < 1 || w <= 0 || h <= 0) {  // ORBextractor.cc:1115-1116: empty image, outputs untouched
        for (int i = 0; i < nimg && n; i++) n[i] = 0;
        return GFO_OK;
    }
    if (stride < w) return fail(c, GFO_ERR_INVALID, "stride < width");
    if (cap < 0) return fail(c, GFO_ERR_INVALID, "negative capacity");   // (found by the sanitizer harness of round 6: image i's arrays are kp + i * cap)
    if (c->combining && nimg == 1) {   // one frame of one caller: may share a device batch with other callers' frames
        gfo_keypoint* kps[1] = {kp};
        uint8_t* ds[1] = {desc};
        c->have_batch = c->have_pyramid = c->have_stereo = false;   // the device-side state lives in the combiner's arena
        if (gfo_has_pair(c)) {      // one camera of a declared stereo rig: this frame and its partner's go down as ONE stereo request
            const int prc = gfo_pair_extract(c, imgs[0], w, h, stride, kp, desc, cap, n);
            if (prc != GFO_COMBINE_DIRECT) return prc;
        }
        const int crc = gfo_combined_extract(c, 1, imgs, w, h, stride, nullptr, kps, ds, cap, n, nullptr, nullptr, nullptr, nullptr, nullptr);
        if (crc != GFO_COMBINE_DIRECT) return crc;
        // the engine cannot serve this frame (no slot could be prepared, or its batch failed as a whole): alone, below
    }
    HIP_TRY(c, hipSetDevice(c->device));
    int rc = gfo_plan(c, w, h, nimg);
    if (rc) return rc;
    if (nimg <= GFO_SMALL_BATCH) {
        gfo_keypoint* kps[GFO_SMALL_BATCH];
        uint8_t* ds[GFO_SMALL_BATCH];
        for (int i = 0; i < nimg; i++) {
            kps[i] = kp ? kp + (size_t)i * cap : nullptr;
            ds[i] = desc ? desc + (size_t)i * cap * 32 : nullptr;
        }
        return extract_small(c, imgs, nimg, w, h, stride, nullptr, kps, ds, cap, n, nullptr, nullptr, nullptr, nullptr, nullptr);
    }
    GfoInput in;
    rc = upload_images(c, imgs, nimg, w, h, stride, &in);
    if (rc) return rc;
    rc = run_extract(c, in, nimg);
    if (rc) return rc;
    rc = gfo_batch_counts(c, n, nullptr);
    if (rc) return rc;
    int over = 0;
    for (int i = 0; i < nimg; i++) {
        const int m = n[i] < cap ? n[i] : cap;
        if (n[i] > cap) over = 1;
        if (m > 0 && kp)
            HIP_TRY(c, hipMemcpyAsync(kp + (size_t)i * cap, c->d_kp + (size_t)i * c->g.kp_stride, sizeof(gfo_keypoint) * m, hipMemcpyDeviceToHost, c->stream));
        if (m > 0 && desc)
            HIP_TRY(c, hipMemcpyAsync(desc + (size_t)i * cap * 32, c->d_desc + (size_t)i * c->g.kp_stride * 32, 32 * (size_t)m, hipMemcpyDeviceToHost, c->stream));
    }
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    return over ? fail(c, GFO_ERR_CAPACITY, "an image produced more keypoints than the caller capacity %d", cap) : GFO_OK;
}

extern "C" int gfo_extract_stereo(gfo_ctx* c, const uint8_t* img_l, const uint8_t* img_r, int w, int h, int stride,
                                  const gfo_stereo_params* p, gfo_keypoint* kp_l, uint8_t* desc_l, gfo_keypoint* kp_r,
                                  uint8_t* desc_r, int cap, int* n_l, int* n_r, float* u_right, float* depth,
                                  int32_t* best_dist, int32_t* best_idx_r, int* nmatched)
{
    if (!c || !p || !n_l || !n_r || !u_right || !depth || !nmatched) return fail(c, GFO_ERR_INVALID, "bad argument");
    *n_l = *n_r = *nmatched = 0;
    if (!img_l || !img_r || w <= 0 || h <= 0) return GFO_OK;   // empty image: outputs untouched (:1115)
    if (stride < w) return fail(c, GFO_ERR_INVALID, "stride < width");
    if (cap < 0) return fail(c, GFO_ERR_INVALID, "negative capacity");
    const uint8_t* imgs[2] = {img_l, img_r};
    gfo_keypoint* kps[2] = {kp_l, kp_r};
    uint8_t* ds[2] = {desc_l, desc_r};
    int n[2] = {0, 0};
    if (c->combining) {
        c->have_batch = c->have_pyramid = c->have_stereo = false;
        const int crc = gfo_combined_extract(c, 2, imgs, w, h, stride, p, kps, ds, cap, n, u_right, depth, best_dist, best_idx_r, nmatched);
        if (crc != GFO_COMBINE_DIRECT) {
            *n_l = n[0];
            *n_r = n[1];
            return crc;
        }
        n[0] = n[1] = 0;      // the engine cannot serve this frame: alone, below
    }
    HIP_TRY(c, hipSetDevice(c->device));
    int rc = gfo_plan(c, w, h, 2);
    if (rc) return rc;
    if (p->n_rows < 1 || p->n_rows > c->st_rows_cap) return fail(c, GFO_ERR_INVALID, "n_rows %d exceeds the planned %d", p->n_rows, c->st_rows_cap);
    rc = extract_small(c, imgs, 2, w, h, stride, p, kps, ds, cap, n, u_right, depth, best_dist, best_idx_r, nmatched);
    *n_l = n[0];
    *n_r = n[1];
    return rc;
}

extern "C" int gfo_extract(gfo_ctx* c, const uint8_t* img, int w, int h, int stride, gfo_keypoint* kp, uint8_t* desc,
                           int cap, int* n)
{
    if (!c || !n) return GFO_ERR_INVALID;
    if (!img || w <= 0 || h <= 0) {
        *n = 0;
        return GFO_OK;
    }
    const uint8_t* one[1] = {img};
    return gfo_extract_batch(c, one, 1, w, h, stride, kp, desc, cap, n);
}

// ---------------------------------------------------------------------------------------------
// pyramid access -- ComputePyramid / mvImagePyramid
// ---------------------------------------------------------------------------------------------
extern "C" int gfo_compute_pyramid(gfo_ctx* c, const uint8_t* img, int w, int h, int stride)
{
    if (!c || !img || w <= 0 || h <= 0 || stride < w) return fail(c, GFO_ERR_INVALID, "bad argument");
    HIP_TRY(c, hipSetDevice(c->device));
    int rc = gfo_plan(c, w, h, 1);
    if (rc) return rc;
    GfoInput in;
    const uint8_t* one[1] = {img};
    rc = upload_images(c, one, 1, w, h, stride, &in);
    if (rc) return rc;
    c->have_batch = false;
    run_pyramid(c, in, 1);
    if (int lrc = gfo_take_launch_err(c)) return lrc;
    HIP_TRY(c, hipGetLastError());
    return GFO_OK;
}

static int reflect101_host(int p, int n)
{
    if (n == 1) return 0;
    while (p < 0 || p >= n) p = p < 0 ? -p : 2 * (n - 1) - p;
    return p;
}

extern "C" int gfo_pyramid_level(gfo_ctx* c, int image, int level, int border, uint8_t* out, int out_stride, int* w, int* h)
{
    if (!c || !out) return GFO_ERR_INVALID;
    if (!c->have_pyramid) return fail(c, GFO_ERR_STATE, "no pyramid has been computed");
    HIP_TRY(c, hipSetDevice(c->device));   // the calling thread may be on another device
    if (image < 0 || image >= c->last_nimg || level < 0 || level >= c->g.nlevels || border < 0)
        return fail(c, GFO_ERR_INVALID, "image/level out of range");
    const GfoLevel& L = c->g.lv[level];
    if (out_stride < L.w + 2 * border) return fail(c, GFO_ERR_CAPACITY, "out_stride too small");
    const uint8_t* src;
    size_t spitch;
    if (level == 0) {
        src = c->last_in.base + (size_t)image * c->last_in.img_stride;
        spitch = (size_t)c->last_in.pitch;
    } else {
        src = c->d_pyr + (size_t)image * c->g.pyr_img_stride + L.plane_off;
        spitch = L.pitch;
    }
    uint8_t* centre = out + (size_t)border * out_stride + border;
    HIP_TRY(c, hipMemcpy2DAsync(centre, out_stride, src, spitch, L.w, L.h, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    if (border > 0) {  // copyMakeBorder(BORDER_REFLECT_101), ORBextractor.cc:1191-1197
        for (int y = -border; y < L.h + border; y++) {
            uint8_t* D = centre + (ptrdiff_t)y * out_stride;
            const uint8_t* S = centre + (ptrdiff_t)reflect101_host(y, L.h) * out_stride;
            if (y < 0 || y >= L.h) memcpy(D, S, L.w);
            for (int x = 1; x <= border; x++) {
                D[-x] = D[reflect101_host(-x, L.w)];
                D[L.w - 1 + x] = D[reflect101_host(L.w - 1 + x, L.w)];
            }
        }
    }
    if (w) *w = L.w;
    if (h) *h = L.h;
    return GFO_OK;
}

// ---------------------------------------------------------------------------------------------
// matchers
// ---------------------------------------------------------------------------------------------
extern "C" int gfo_hamming256(const void* a, const void* b)
{
    // ORBmatcher::DescriptorDistance, ORBmatcher.cc:1768-1784 (popcount of the XOR)
    uint64_t x[4], y[4];
    memcpy(x, a, 32);
    memcpy(y, b, 32);
    return __builtin_popcountll(x[0] ^ y[0]) + __builtin_popcountll(x[1] ^ y[1]) +
           __builtin_popcountll(x[2] ^ y[2]) + __builtin_popcountll(x[3] ^ y[3]);
}

static int scratch(gfo_ctx* c, size_t bytes)
{
    if (bytes <= c->scratch_bytes) return GFO_OK;
    (void)hipStreamSynchronize(c->stream);
    if (c->d_scratch) (void)hipFree(c->d_scratch);
    c->d_scratch = nullptr;
    c->scratch_bytes = 0;
    HIP_TRY(c, hipMalloc(&c->d_scratch, bytes));
    c->scratch_bytes = bytes;
    return GFO_OK;
}

extern "C" int gfo_stereo_match(gfo_ctx* c, const gfo_keypoint* kl, const uint8_t* dl, int nl, const gfo_keypoint* kr,
                                const uint8_t* dr, int nr, const float* sf, int nlevels, const gfo_stereo_params* p,
                                const float* min_d, const float* max_d, float* u_right, float* depth,
                                int32_t* best_dist, int32_t* best_idx_r, int* nmatched)
{
    if (!c || !p || !sf || !u_right || !depth || !nmatched || nl < 0 || nr < 0 || nlevels < 1 || nlevels > GFO_MAX_LEVELS ||
        (nl > 0 && (!kl || !dl)) || (nr > 0 && (!kr || !dr)) || ((min_d == nullptr) != (max_d == nullptr)))
        return fail(c, GFO_ERR_INVALID, "bad argument");
    if (nr > 65535) return fail(c, GFO_ERR_INVALID, "more than 65535 right keypoints");
    if (nl == 0) { *nmatched = 0; return GFO_OK; }
    // a declared stereo rig (gfo_ctx_pair) whose last frame these arrays are, bit for bit: the association was computed with that
    // frame (and arrays this library delivered need no validation)
    if (gfo_has_pair(c) && gfo_pair_lookup(c, kl, dl, nl, kr, dr, nr, sf, nlevels, p, min_d, max_d, u_right, depth, best_dist, best_idx_r, nmatched) == 0)
        return GFO_OK;
    // the kernels index scale[octave] (Frame.h:244, Frame.cc:1204-1206 do the same, unchecked): refuse what would read past it
    for (int i = 0; i < nl; i++)
        if (kl[i].octave < 0 || kl[i].octave >= nlevels) return fail(c, GFO_ERR_INVALID, "left keypoint %d: octave %d outside 0..%d", i, kl[i].octave, nlevels - 1);
    for (int i = 0; i < nr; i++)
        if (kr[i].octave < 0 || kr[i].octave >= nlevels) return fail(c, GFO_ERR_INVALID, "right keypoint %d: octave %d outside 0..%d", i, kr[i].octave, nlevels - 1);
    if (p->n_rows < 1 || p->n_rows > 8192) return fail(c, GFO_ERR_INVALID, "n_rows must be 1..8192");
    if (c->combining) {    // the pairs several threads associate at once share one launch (gfo_combine.hip); 1 = not eligible
        int status = GFO_OK;
        if (gfo_combined_stereo_match(c, kl, dl, nl, kr, dr, nr, sf, nlevels, p, min_d, max_d, u_right, depth, best_dist, best_idx_r, nmatched, &status) == 0)
            return status;
    }
    HIP_TRY(c, hipSetDevice(c->device));
    // scratch layout: inputs first, outputs next, work buffers last -- the input and output regions are mirrored in
    // pinned host memory, so the call is one H2D copy, three kernels, one D2H copy and one synchronisation
    size_t off = 0;
    auto take = [&](size_t bytes) { size_t o = off; off = (size_t)align_up((long long)(off + bytes), 256); return o; };
    const int nr1 = nr > 0 ? nr : 1;
    const size_t o_kl = take(sizeof(gfo_keypoint) * nl), o_dl = take(32 * (size_t)nl);
    const size_t o_kr = take(sizeof(gfo_keypoint) * nr1), o_dr = take(32 * (size_t)nr1);
    const size_t o_sf = take(sizeof(float) * GFO_MAX_LEVELS);
    const size_t o_min = take(sizeof(float) * nl), o_max = take(sizeof(float) * nl);
    const size_t in_bytes = off;
    const size_t o_u = take(sizeof(float) * nl), o_dp = take(sizeof(float) * nl);
    const size_t o_bd = take(sizeof(int) * nl), o_bi = take(sizeof(int) * nl), o_nm = take(sizeof(int));
    const size_t out_bytes = off - in_bytes;
    const size_t o_ct = take(nl);
    const size_t o_sx = take(4 * (size_t)nr1), o_sy = take(4 * (size_t)nr1), o_soi = take(4 * (size_t)nr1),
                 o_sd = take(32 * (size_t)nr1), o_rs = take(4 * (size_t)(p->n_rows + 1)), o_lo = take(4 * (size_t)nl),
                 o_lrs = take(4 * (size_t)(p->n_rows + 1));
    int rc = scratch(c, off);
    if (rc) return rc;
    rc = gfo_pinned(c, &c->h_in, &c->h_in_bytes, in_bytes);
    if (rc) return rc;
    rc = gfo_pinned(c, &c->h_out, &c->h_out_bytes, out_bytes);
    if (rc) return rc;
    uint8_t* S = (uint8_t*)c->d_scratch;
    uint8_t* HI = c->h_in;
    const bool win = min_d && max_d;
    memcpy(HI + o_kl, kl, sizeof(gfo_keypoint) * nl);
    memcpy(HI + o_dl, dl, 32 * (size_t)nl);
    if (nr > 0) {
        memcpy(HI + o_kr, kr, sizeof(gfo_keypoint) * nr);
        memcpy(HI + o_dr, dr, 32 * (size_t)nr);
    }
    memcpy(HI + o_sf, sf, sizeof(float) * nlevels);
    if (win) {
        memcpy(HI + o_min, min_d, sizeof(float) * nl);
        memcpy(HI + o_max, max_d, sizeof(float) * nl);
    }
    // as the other host-array matcher calls (GfoXfer): a few hundred KB in front of and behind three short kernels go by copy KERNELS of
    // the compute queue -- the copy engine's hand-overs cost more than the copies
    static const long kernel_max = getenv("GFO_MATCHER_UPLOAD_KERNEL_MAX") ? atol(getenv("GFO_MATCHER_UPLOAD_KERNEL_MAX")) : (1L << 20);
    if ((long)in_bytes <= kernel_max) gfo_launch_copy16(c, S, HI, in_bytes, c->stream);
    else HIP_TRY(c, hipMemcpyAsync(S, HI, in_bytes, hipMemcpyHostToDevice, c->stream));
    GfoStereoDev out{(float*)(S + o_u), (float*)(S + o_dp), (int*)(S + o_bd), (int*)(S + o_bi), (int*)(S + o_nm), S + o_ct};
    GfoStereoLaunch sl{};
    sl.kl = (const gfo_keypoint*)(S + o_kl); sl.dl = S + o_dl;
    sl.kr = (const gfo_keypoint*)(S + o_kr); sl.dr = S + o_dr;
    sl.cnt_dev = nullptr; sl.nl_host = nl; sl.nr_host = nr;
    sl.pair_stride_kp = 0; sl.npairs = 1;
    sl.d_scale = (const float*)(S + o_sf);
    sl.p = *p;
    sl.min_d = win ? (const float*)(S + o_min) : nullptr;
    sl.max_d = win ? (const float*)(S + o_max) : nullptr;
    sl.out = out; sl.out_stride = nl;
    sl.sort = GfoStereoSort{(float*)(S + o_sx), (float*)(S + o_sy), (unsigned*)(S + o_soi), S + o_sd, (int*)(S + o_rs), (int*)(S + o_lo), (int*)(S + o_lrs)};
    sl.sort_stride = nr1;
    sl.window = gfo_stereo_window(sf, nlevels);
    sl.nlevels = nlevels;
    gfo_launch_stereo(c, sl);
    if (int lrc = gfo_take_launch_err(c)) return lrc;
    HIP_TRY(c, hipGetLastError());
    uint8_t* HO = c->h_out;
    if (gfo_matcher_host_writes() && (long)out_bytes <= kernel_max) gfo_launch_copy16(c, HO, S + in_bytes, out_bytes, c->stream);
    else HIP_TRY(c, hipMemcpyAsync(HO, S + in_bytes, out_bytes, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    memcpy(u_right, HO + (o_u - in_bytes), sizeof(float) * nl);
    memcpy(depth, HO + (o_dp - in_bytes), sizeof(float) * nl);
    if (best_dist) memcpy(best_dist, HO + (o_bd - in_bytes), sizeof(int) * nl);
    if (best_idx_r) memcpy(best_idx_r, HO + (o_bi - in_bytes), sizeof(int) * nl);
    memcpy(nmatched, HO + (o_nm - in_bytes), sizeof(int));
    return GFO_OK;
}

extern "C" int gfo_stereo_match_batch(gfo_ctx* c, const gfo_stereo_params* p)
{
    if (!c || !p) return GFO_ERR_INVALID;
    if (!c->have_batch) return fail(c, GFO_ERR_STATE, "no batch has been extracted");
    HIP_TRY(c, hipSetDevice(c->device));   // the calling thread may be on another device
    if (c->last_nimg < 2 || (c->last_nimg & 1)) return fail(c, GFO_ERR_STATE, "stereo needs an even number of images (L,R,L,R,...)");
    if (p->n_rows < 1 || p->n_rows > c->st_rows_cap) return fail(c, GFO_ERR_INVALID, "n_rows %d exceeds the planned %d", p->n_rows, c->st_rows_cap);
    gfo_launch_stereo(c, stereo_batch_launch(c, *p));
    if (int lrc = gfo_take_launch_err(c)) return lrc;
    HIP_TRY(c, hipGetLastError());
    c->have_stereo = true;
    return GFO_OK;
}

extern "C" int gfo_stereo_match_sad_batch(gfo_ctx* c, float mbf, float mb)
{
    if (!c || !(mb > 0)) return GFO_ERR_INVALID;
    if (!c->have_batch) return fail(c, GFO_ERR_STATE, "no batch has been extracted");
    HIP_TRY(c, hipSetDevice(c->device));   // the calling thread may be on another device
    if (c->last_nimg < 2 || (c->last_nimg & 1)) return fail(c, GFO_ERR_STATE, "stereo needs an even number of images (L,R,L,R,...)");
    const int npairs = c->last_nimg / 2;
    GfoStereoLaunch sl{};
    sl.kl = c->d_kp; sl.dl = c->d_desc;
    sl.kr = c->d_kp + c->g.kp_stride; sl.dr = c->d_desc + (size_t)c->g.kp_stride * 32;
    sl.cnt_dev = c->d_kp_cnt;
    sl.pair_stride_kp = 2LL * c->g.kp_stride; sl.npairs = npairs;
    sl.d_scale = c->d_scale;
    sl.p = gfo_stereo_params{c->g.h0, mbf, mb, 0.f};  // nRows = mvImagePyramid[0].rows
    sl.out = c->st; sl.out_stride = c->g.kp_stride;
    sl.sort = c->st_sort; sl.sort_stride = c->g.kp_stride;
    sl.window = gfo_stereo_window(c->scale.data(), c->g.nlevels);
    gfo_launch_stereo_sad(c, sl, c->last_in, c->d_inv_scale);
    if (int lrc = gfo_take_launch_err(c)) return lrc;
    HIP_TRY(c, hipGetLastError());
    c->have_stereo = true;
    return GFO_OK;
}

extern "C" int gfo_stereo_fetch(gfo_ctx* c, int pair, float* u_right, float* depth, int32_t* best_dist,
                                int32_t* best_idx_r, int cap, int* nmatched)
{
    if (!c) return GFO_ERR_INVALID;
    if (!c->have_stereo) return fail(c, GFO_ERR_STATE, "no stereo batch has been matched");
    HIP_TRY(c, hipSetDevice(c->device));   // the calling thread may be on another device
    if (pair < 0 || pair >= c->last_nimg / 2) return fail(c, GFO_ERR_INVALID, "pair out of range");
    int nl = 0;
    HIP_TRY(c, hipMemcpyAsync(&nl, c->d_kp_cnt + 2 * pair, sizeof(int), hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    const int m = nl < cap ? nl : cap;
    const size_t o = (size_t)pair * c->g.kp_stride;
    if (m > 0 && u_right) HIP_TRY(c, hipMemcpyAsync(u_right, c->st.u_right + o, sizeof(float) * m, hipMemcpyDeviceToHost, c->stream));
    if (m > 0 && depth) HIP_TRY(c, hipMemcpyAsync(depth, c->st.depth + o, sizeof(float) * m, hipMemcpyDeviceToHost, c->stream));
    if (m > 0 && best_dist) HIP_TRY(c, hipMemcpyAsync(best_dist, c->st.best_dist + o, sizeof(int) * m, hipMemcpyDeviceToHost, c->stream));
    if (m > 0 && best_idx_r) HIP_TRY(c, hipMemcpyAsync(best_idx_r, c->st.best_idx + o, sizeof(int) * m, hipMemcpyDeviceToHost, c->stream));
    if (nmatched) HIP_TRY(c, hipMemcpyAsync(nmatched, c->st.nmatched + pair, sizeof(int), hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    return nl > cap ? fail(c, GFO_ERR_CAPACITY, "%d left keypoints, caller capacity %d", nl, cap) : GFO_OK;
}

// ---------------------------------------------------------------------------------------------
// profiling + debug
// ---------------------------------------------------------------------------------------------
extern "C" int gfo_profile_enable(gfo_ctx* c, int on)
{
    if (!c) return GFO_ERR_INVALID;
    (void)hipSetDevice(c->device);
    prof_collect(c);
    c->profiling = on != 0;
    return GFO_OK;
}

extern "C" int gfo_profile_read(gfo_ctx* c, gfo_stage_time* out, int cap, int* nstages, int reset)
{
    if (!c || !nstages) return GFO_ERR_INVALID;
    (void)hipSetDevice(c->device);
    prof_collect(c);
    int k = 0;
    for (int s = 0; s < ST_COUNT; s++) {
        if (k < cap && out) {
            memset(&out[k], 0, sizeof out[k]);
            strncpy(out[k].name, k_stage_names[s], sizeof out[k].name - 1);
            out[k].ms = c->stage_ms[s];
            out[k].launches = c->stage_launches[s];
        }
        k++;
    }
    *nstages = k;
    if (reset)
        for (int s = 0; s < ST_COUNT; s++) { c->stage_ms[s] = 0; c->stage_launches[s] = 0; }
    return GFO_OK;
}

extern "C" int gfo_debug_blurred_level(gfo_ctx* c, int image, int level, uint8_t* out, int out_stride)
{
    if (!c || !out) return GFO_ERR_INVALID;
    if (!c->have_batch) return fail(c, GFO_ERR_STATE, "no batch has been extracted");
    HIP_TRY(c, hipSetDevice(c->device));   // the calling thread may be on another device
    if (image < 0 || image >= c->last_nimg || level < 0 || level >= c->g.nlevels) return fail(c, GFO_ERR_INVALID, "out of range");
    const GfoLevel& L = c->g.lv[level];
    HIP_TRY(c, hipMemcpy2DAsync(out, out_stride, c->d_blur + (size_t)image * c->g.blur_img_stride + L.blur_off, L.pitch,
                                L.w, L.h, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    return GFO_OK;
}

extern "C" int gfo_debug_level_candidates(gfo_ctx* c, int image, int level, int32_t* xys, int cap, int* n)
{
    if (!c || !n) return GFO_ERR_INVALID;
    if (!c->have_batch) return fail(c, GFO_ERR_STATE, "no batch has been extracted");
    HIP_TRY(c, hipSetDevice(c->device));   // the calling thread may be on another device
    if (image < 0 || image >= c->last_nimg || level < 0 || level >= c->g.nlevels) return fail(c, GFO_ERR_INVALID, "out of range");
    const GfoLevel& L = c->g.lv[level];
    int cnt = 0;
    HIP_TRY(c, hipMemcpyAsync(&cnt, c->d_cand_cnt + (image * c->g.nlevels + level) * GFO_CNT_STRIDE, sizeof(int), hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    if (cnt > L.cand_cap) cnt = L.cand_cap;
    *n = cnt;
    const int m = cnt < cap ? cnt : cap;
    if (m > 0 && xys) {
        std::vector<uint32_t> tmp(m);
        HIP_TRY(c, hipMemcpy(tmp.data(), c->d_cand + (size_t)image * c->g.cand_img_stride + L.cand_off, sizeof(uint32_t) * m, hipMemcpyDeviceToHost));
        for (int i = 0; i < m; i++) {
            xys[3 * i] = (int)(tmp[i] & 0xFFF);
            xys[3 * i + 1] = (int)((tmp[i] >> 12) & 0xFFF);
            xys[3 * i + 2] = (int)(tmp[i] >> 24);
        }
    }
    return GFO_OK;
}

static void gfo_kernels_api(std::vector<const void*>& v)
{
    v.push_back((const void*)k_pack_results); v.push_back((const void*)k_unpack_pairs);
}
