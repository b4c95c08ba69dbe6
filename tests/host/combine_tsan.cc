// combine_tsan.cc -- the frame combiner and the stereo rigs (gf-orb-slam2_amd/csrc/gfo_combine.hip: engine slots, forming batches,
// the GfoPair rendezvous with its lock-free state mirror, dormancy and wake counts) under ThreadSanitizer, on the CPU.
//
// The product file is #included UNMODIFIED; tests/host/tsan/fake_gfo_internal.h stands in for gfo_internal.h and the HIP runtime
// (the "device" is host memory, a batch takes 30-200 us, results are a pure function of the image bytes).  The callers below do
// what the C ABI's entry points do around the combiner (gfo_extract / gfo_extract_stereo / gfo_stereo_match in gfo_api.hip) and
// what the reference's callers do around those (Frame.cc:84-100: a thread per right image, then the association):
//   A  K = 1 .. 6 cameras, the adapter's pattern with declared rigs -- every result checked against the image it belongs to;
//   B  stereo frames in one submission and monocular frames on one engine, with injected batch failures (members re-run alone);
//   C  a rig whose partner never shows up (solo frames, dormancy, wake counts), then the partner returns;
//   D  gfo_ctx_pair re-declared with another calibration by a third thread while frames run;
//   E  the right context destroyed and re-created while the left side is waiting for it;
//   F  host-array associations of several threads in one batch (kind 3), with and without disparity windows.
// Built and run by tests/test_host_logic.py::test_combiner_under_thread_sanitizer:
//   g++ -std=c++17 -O1 -g -fsanitize=thread -x c++ tests/host/combine_tsan.cc -lpthread
// Exit code 0 and no "ThreadSanitizer" line on stderr = pass.
#include "tsan/fake_gfo_internal.h"

#include <mutex>
#include <random>

#include <pthread.h>
#include <time.h>

// libstdc++ implements condition_variable::wait_until(steady_clock) with pthread_cond_clockwait (glibc >= 2.30), which the
// ThreadSanitizer runtime of GCC 11 does not intercept: it then never sees the mutex released and re-acquired inside the wait and
// reports "double lock of a mutex" and races between threads that both "hold" it.  The executable's own definition below is the one
// libstdc++ binds to; it forwards to pthread_cond_timedwait (CLOCK_REALTIME), which the runtime does know.
extern "C" int pthread_cond_clockwait(pthread_cond_t* cond, pthread_mutex_t* mutex, clockid_t clock, const struct timespec* abstime)
{
    struct timespec now_c, now_r, t;
    clock_gettime(clock, &now_c);
    clock_gettime(CLOCK_REALTIME, &now_r);
    long long d = (abstime->tv_sec - now_c.tv_sec) * 1000000000LL + (abstime->tv_nsec - now_c.tv_nsec);
    if (d < 0) d = 0;
    const long long r = now_r.tv_sec * 1000000000LL + now_r.tv_nsec + d;
    t.tv_sec = r / 1000000000LL;
    t.tv_nsec = r % 1000000000LL;
    return pthread_cond_timedwait(cond, mutex, &t);
}

std::atomic<long> fake_batches{0}, fake_contexts_alive{0};
std::atomic<int> fake_fail_submit_every{0};
static std::atomic<long> g_frames{0}, g_wrong{0}, g_errors{0}, g_injected{0};
static std::atomic<bool> g_expect_overflow{false};   // scenario B: a batch of ONE that the fake fails is that frame's own GFO_ERR_OVERFLOW

// ---------------------------------------------------------------------------------------------------------------------------
// the fake backend
// ---------------------------------------------------------------------------------------------------------------------------
int gfo_fail(gfo_ctx* c, int code, const char* fmt, ...)
{
    char buf[256];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    if (c) c->err = buf;
    return code;
}
extern "C" const char* gfo_last_error(const gfo_ctx* c) { return c ? c->err.c_str() : "context creation failed"; }
extern "C" int gfo_ctx_create(const gfo_params* p, int device, gfo_ctx** out)
{
    gfo_ctx* c = new gfo_ctx();
    c->prm = *p;
    c->device = device;
    float s = 1.f;
    for (int l = 0; l < p->nlevels; l++) { c->scale.push_back(s); s *= p->scale_factor; }
    fake_contexts_alive++;
    *out = c;
    return GFO_OK;
}
extern "C" void gfo_ctx_destroy(gfo_ctx* c)
{
    if (!c) return;
    gfo_pair_release(c);      // the order of the real gfo_ctx_destroy
    gfo_engine_release(c);
    free(c->h_in); free(c->h_out); free(c->d_input);
    fake_contexts_alive--;
    delete c;
}
int gfo_plan(gfo_ctx* c, int w, int h, int batch)
{
    if (c->g.w0 == w && c->g.h0 == h && c->batch_cap >= batch) return GFO_OK;
    c->g.w0 = w; c->g.h0 = h; c->st_rows_cap = h + 64; c->batch_cap = batch;
    free(c->d_input);
    c->d_input = (uint8_t*)malloc((size_t)batch * w * h);
    return GFO_OK;
}
int gfo_small_prepare(gfo_ctx* c, int nimg_cap, GfoSmallLayout* L)
{
    const int ks = c->g.kp_stride, w = c->g.w0, h = c->g.h0;
    L->nimg_cap = nimg_cap; L->pitch = w; L->img_bytes = (size_t)w * h;
    const size_t pairs_need = gfo_pair_block(ks).bytes * (size_t)((nimg_cap + 1) / 2);
    const size_t in_need = L->img_bytes * nimg_cap > pairs_need ? L->img_bytes * nimg_cap : pairs_need;
    if (in_need > c->h_in_bytes) { free(c->h_in); c->h_in = (uint8_t*)malloc(in_need); c->h_in_bytes = in_need; }
    size_t off = 0;
    auto take = [&](size_t bytes) { size_t o = off; off = (off + bytes + 63) & ~(size_t)63; return o; };
    const size_t npair = (size_t)(nimg_cap + 1) / 2;
    L->o_fl = take(16); L->o_cnt = take(16 * (size_t)((nimg_cap + 3) / 4)); L->o_kp = take(sizeof(gfo_keypoint) * (size_t)ks * nimg_cap);
    L->o_ds = take(32 * (size_t)ks * nimg_cap);
    L->o_ur = take(4 * (size_t)ks * npair); L->o_dp = take(4 * (size_t)ks * npair); L->o_bd = take(4 * (size_t)ks * npair);
    L->o_bi = take(4 * (size_t)ks * npair); L->o_nm = take(16 * ((npair + 3) / 4));
    free(c->h_out);
    c->h_out = (uint8_t*)malloc(off);
    return GFO_OK;
}
int gfo_small_upload(gfo_ctx* c, gfo_ctx*, const GfoSmallLayout& L, int first, int count, const uint8_t* const* imgs, int w, int h, int stride, hipStream_t, bool)
{
    for (int i = 0; i < count; i++) {
        uint8_t* stage = c->h_in + (size_t)(first + i) * L.img_bytes;
        for (int y = 0; y < h; y++) memcpy(stage + (size_t)y * L.pitch, imgs[i] + (size_t)y * stride, w);
        memcpy(c->d_input + (size_t)(first + i) * L.img_bytes, stage, L.img_bytes);      // the "DMA"
    }
    return GFO_OK;
}
static void fake_device_time()
{
    static thread_local std::minstd_rand rng((unsigned)std::hash<std::thread::id>()(std::this_thread::get_id()));
    std::this_thread::sleep_for(std::chrono::microseconds(30 + rng() % 170));
}
static void fake_results(const uint8_t* img, gfo_keypoint* kp, uint8_t* desc, int* n)
{
    *n = fake_count(img);
    for (int j = 0; j < *n; j++) fake_keypoint(img, j, kp + j, desc + 32 * (size_t)j);
}
static void fake_stereo(const uint8_t* l, const uint8_t* r, int nl, float* ur, float* dp, int32_t* bd, int32_t* bi, int* nm)
{
    for (int j = 0; j < nl; j++) { ur[j] = fake_uright(l, r, j); dp[j] = ur[j] * 0.5f; bd[j] = (l[7] + j) % 100; bi[j] = (r[7] + j) % 50; }
    *nm = fake_nmatched(l, r);
}
int gfo_small_submit(gfo_ctx* c, const GfoSmallLayout& L, int nimg, const gfo_stereo_params* sp, bool)
{
    fake_device_time();
    const long b = ++fake_batches;
    const int every = fake_fail_submit_every.load();
    if (every > 0 && b % every == 0) return gfo_fail(c, GFO_ERR_OVERFLOW, "internal buffer overflow (injected by the fake backend)");
    const int ks = c->g.kp_stride;
    uint8_t* H = c->h_out;
    for (int i = 0; i < nimg; i++)
        fake_results(c->d_input + (size_t)i * L.img_bytes, reinterpret_cast<gfo_keypoint*>(H + L.o_kp) + (size_t)ks * i, H + L.o_ds + 32 * (size_t)ks * i,
                     reinterpret_cast<int*>(H + L.o_cnt) + i);
    if (sp)
        for (int p = 0; p < nimg / 2; p++) {
            const uint8_t* l = c->d_input + (size_t)(2 * p) * L.img_bytes;
            fake_stereo(l, l + L.img_bytes, fake_count(l), reinterpret_cast<float*>(H + L.o_ur) + (size_t)ks * p, reinterpret_cast<float*>(H + L.o_dp) + (size_t)ks * p,
                        reinterpret_cast<int32_t*>(H + L.o_bd) + (size_t)ks * p, reinterpret_cast<int32_t*>(H + L.o_bi) + (size_t)ks * p,
                        reinterpret_cast<int*>(H + L.o_nm) + p);
        }
    return GFO_OK;
}
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
void gfo_small_collect_stereo(gfo_ctx* c, const GfoSmallLayout& L, int pair, int n_left, int cap, float* u_right, float* depth, int32_t* best_dist,
                              int32_t* best_idx_r, int* nmatched)
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
    auto take = [&](size_t bytes) { size_t o = off; off = (off + bytes + 15) & ~(size_t)15; return o; };
    b.o_kl = take(sizeof(gfo_keypoint) * (size_t)ks); b.o_dl = take(32 * (size_t)ks);
    b.o_kr = take(sizeof(gfo_keypoint) * (size_t)ks); b.o_dr = take(32 * (size_t)ks);
    b.o_min = take(4 * (size_t)ks); b.o_max = take(4 * (size_t)ks);
    b.bytes = (off + 255) & ~(size_t)255;
    return b;
}
// the association of host arrays in the fake: u_right[j] = kl[j].x + kr[0].y + max_d[j], nmatched = nl + nr
static void fake_match_arrays(const gfo_keypoint* kl, int nl, const gfo_keypoint* kr, int nr, const float* mx, float* ur, float* dp, int32_t* bd, int32_t* bi, int* nm)
{
    for (int j = 0; j < nl; j++) { ur[j] = kl[j].x + (nr ? kr[0].y : 0.f) + mx[j]; dp[j] = kl[j].y; bd[j] = j; bi[j] = nr ? j % nr : -1; }
    *nm = nl + nr;
}
int gfo_small_submit_pairs(gfo_ctx* c, const GfoSmallLayout& L, int npairs, const gfo_stereo_params*, const uint8_t* d_stage)
{
    fake_device_time();
    ++fake_batches;
    const int ks = c->g.kp_stride;
    const GfoPairBlock b = gfo_pair_block(ks);
    uint8_t* H = c->h_out;
    for (int p = 0; p < npairs; p++) {
        const uint8_t* S = d_stage + (size_t)p * b.bytes;
        const int nl = reinterpret_cast<const int*>(S)[0], nr = reinterpret_cast<const int*>(S)[1];
        fake_match_arrays(reinterpret_cast<const gfo_keypoint*>(S + b.o_kl), nl, reinterpret_cast<const gfo_keypoint*>(S + b.o_kr), nr,
                          reinterpret_cast<const float*>(S + b.o_max), reinterpret_cast<float*>(H + L.o_ur) + (size_t)ks * p,
                          reinterpret_cast<float*>(H + L.o_dp) + (size_t)ks * p, reinterpret_cast<int32_t*>(H + L.o_bd) + (size_t)ks * p,
                          reinterpret_cast<int32_t*>(H + L.o_bi) + (size_t)ks * p, reinterpret_cast<int*>(H + L.o_nm) + p);
    }
    return GFO_OK;
}
void gfo_note_pinned(const uint8_t*, size_t, bool) {}

// ---------------------------------------------------------------------------------------------------------------------------
// THE PRODUCT FILE, as it is
// ---------------------------------------------------------------------------------------------------------------------------
#include "../../gf-orb-slam2_amd/csrc/gfo_combine.hip"

// ---------------------------------------------------------------------------------------------------------------------------
// the callers: what gfo_api.hip's entry points do around the combiner
// ---------------------------------------------------------------------------------------------------------------------------
static const int W = 32, H = 16, CAP = 64;

static int api_extract(gfo_ctx* c, const uint8_t* img, gfo_keypoint* kp, uint8_t* desc, int* n)
{
    if (c->combining) {
        if (gfo_has_pair(c)) {
            const int prc = gfo_pair_extract(c, img, W, H, W, kp, desc, CAP, n);
            if (prc != GFO_COMBINE_DIRECT) return prc;
        }
        const uint8_t* imgs[1] = {img};
        gfo_keypoint* kps[1] = {kp};
        uint8_t* ds[1] = {desc};
        const int crc = gfo_combined_extract(c, 1, imgs, W, H, W, nullptr, kps, ds, CAP, n, nullptr, nullptr, nullptr, nullptr, nullptr);
        if (crc != GFO_COMBINE_DIRECT) return crc;
    }
    fake_device_time();       // the direct path: alone on the context's own arena
    fake_results(img, kp, desc, n);
    return GFO_OK;
}
static int api_extract_stereo(gfo_ctx* c, const uint8_t* l, const uint8_t* r, const gfo_stereo_params* p, gfo_keypoint* kl, uint8_t* dl, int* nl, gfo_keypoint* kr,
                              uint8_t* dr, int* nr, float* ur, float* dp, int32_t* bd, int32_t* bi, int* nm)
{
    int n[2] = {0, 0};
    if (c->combining) {
        const uint8_t* imgs[2] = {l, r};
        gfo_keypoint* kps[2] = {kl, kr};
        uint8_t* ds[2] = {dl, dr};
        const int crc = gfo_combined_extract(c, 2, imgs, W, H, W, p, kps, ds, CAP, n, ur, dp, bd, bi, nm);
        if (crc != GFO_COMBINE_DIRECT) { *nl = n[0]; *nr = n[1]; return crc; }
    }
    fake_device_time();
    fake_results(l, kl, dl, nl);
    fake_results(r, kr, dr, nr);
    fake_stereo(l, r, *nl, ur, dp, bd, bi, nm);
    return GFO_OK;
}
static int api_stereo_match(gfo_ctx* c, const gfo_keypoint* kl, const uint8_t* dl, int nl, const gfo_keypoint* kr, const uint8_t* dr, int nr,
                            const gfo_stereo_params* p, const float* mn, const float* mx, float* ur, float* dp, int32_t* bd, int32_t* bi, int* nm, int* how)
{
    *how = 0;
    if (gfo_has_pair(c) && gfo_pair_lookup(c, kl, dl, nl, kr, dr, nr, c->scale.data(), (int)c->scale.size(), p, mn, mx, ur, dp, bd, bi, nm) == 0) { *how = 1; return GFO_OK; }
    if (c->combining) {
        int status = 0;
        if (gfo_combined_stereo_match(c, kl, dl, nl, kr, dr, nr, c->scale.data(), (int)c->scale.size(), p, mn, mx, ur, dp, bd, bi, nm, &status) == 0) { *how = 2; return status; }
    }
    *how = 3;
    fake_device_time();
    std::vector<float> whole(nl, p->mbf / p->mb);
    fake_match_arrays(kl, nl, kr, nr, mx ? mx : whole.data(), ur, dp, bd, bi, nm);
    return GFO_OK;
}

struct Arrays {
    gfo_keypoint k[CAP];
    uint8_t d[CAP * 32];
    int n = 0;
};
static void make_image(uint8_t* img, unsigned seed)
{
    std::minstd_rand r(seed * 2654435761u + 12345u);
    for (int i = 0; i < W * H; i++) img[i] = (uint8_t)r();
}
static bool check_image(const uint8_t* img, const Arrays& a)
{
    Arrays want;
    fake_results(img, want.k, want.d, &want.n);
    return a.n == want.n && memcmp(a.k, want.k, sizeof(gfo_keypoint) * want.n) == 0 && memcmp(a.d, want.d, 32 * (size_t)want.n) == 0;
}
#define EXPECT(cond, what)                                                               \
    do {                                                                                 \
        if (!(cond)) {                                                                   \
            g_wrong++;                                                                   \
            if (g_wrong.load() < 20) fprintf(stderr, "WRONG %s (line %d)\n", what, __LINE__); \
        }                                                                                \
    } while (0)

static gfo_ctx* make_ctx(bool combining = true)
{
    gfo_params p = {2000, 1.2f, 8, 20, 7, 1};
    gfo_ctx* c = nullptr;
    gfo_ctx_create(&p, 0, &c);
    gfo_ctx_set_combining(c, combining ? 1 : 0);
    return c;
}

// one camera in the adapter's pattern: left extractor on this thread, right extractor on a thread created per frame (Frame.cc:84-87),
// then the association on the arrays the two calls returned (Frame.cc:100), with the rig declared as adapter/matchers_gfo.cc does
static void camera_adapter(int cam, int frames, gfo_ctx* L, gfo_ctx* R, bool declare, std::atomic<long>* served)
{
    const gfo_stereo_params sp = {H, 40.f + cam, 0.1f, 0.f};
    std::vector<uint8_t> il(W * H), ir(W * H);
    Arrays a, b;
    float ur[CAP], dp[CAP];
    int32_t bd[CAP], bi[CAP];
    for (int f = 0; f < frames; f++) {
        make_image(il.data(), cam * 100000 + 2 * f);
        make_image(ir.data(), cam * 100000 + 2 * f + 1);
        int rcr = 0;
        std::thread tr([&] { rcr = api_extract(R, ir.data(), b.k, b.d, &b.n); });
        const int rcl = api_extract(L, il.data(), a.k, a.d, &a.n);
        tr.join();
        if (rcl || rcr) { g_errors++; continue; }
        EXPECT(check_image(il.data(), a), "left arrays");
        EXPECT(check_image(ir.data(), b), "right arrays");
        if (declare) gfo_ctx_pair(L, R, &sp);
        int nm = -1, how = 0;
        const int rc = api_stereo_match(L, a.k, a.d, a.n, b.k, b.d, b.n, &sp, nullptr, nullptr, ur, dp, bd, bi, &nm, &how);
        if (rc) { g_errors++; continue; }
        if (how == 1) {      // answered from the frame the rig extracted: the association of THESE two images
            if (served) (*served)++;
            EXPECT(nm == fake_nmatched(il.data(), ir.data()) && ur[a.n - 1] == fake_uright(il.data(), ir.data(), a.n - 1), "rig answer");
        } else {
            EXPECT(nm == a.n + b.n && ur[0] == a.k[0].x + b.k[0].y + sp.mbf / sp.mb, "association of host arrays");
        }
        g_frames++;
    }
}

static void camera_stereo_call(int cam, int frames, gfo_ctx* c)
{
    const gfo_stereo_params sp = {H, 40.f + (cam & 1), 0.1f, 0.f};      // two calibrations: batches form per calibration
    std::vector<uint8_t> il(W * H), ir(W * H);
    Arrays a, b;
    float ur[CAP], dp[CAP];
    int32_t bd[CAP], bi[CAP];
    for (int f = 0; f < frames; f++) {
        make_image(il.data(), 7000000 + cam * 100000 + 2 * f);
        make_image(ir.data(), 7000000 + cam * 100000 + 2 * f + 1);
        int nm = -1;
        const int rc = api_extract_stereo(c, il.data(), ir.data(), &sp, a.k, a.d, &a.n, b.k, b.d, &b.n, ur, dp, bd, bi, &nm);
        if (rc == GFO_ERR_OVERFLOW && g_expect_overflow.load()) { g_injected++; continue; }
        if (rc) { g_errors++; continue; }
        EXPECT(check_image(il.data(), a) && check_image(ir.data(), b), "stereo call arrays");
        EXPECT(nm == fake_nmatched(il.data(), ir.data()) && ur[0] == fake_uright(il.data(), ir.data(), 0), "stereo call association");
        g_frames++;
    }
}

static void camera_mono(int cam, int frames, gfo_ctx* c)
{
    std::vector<uint8_t> im(W * H);
    Arrays a;
    for (int f = 0; f < frames; f++) {
        make_image(im.data(), 9000000 + cam * 100000 + f);
        const int rc = api_extract(c, im.data(), a.k, a.d, &a.n);
        if (rc == GFO_ERR_OVERFLOW && g_expect_overflow.load()) { g_injected++; continue; }
        if (rc) { g_errors++; continue; }
        EXPECT(check_image(im.data(), a), "mono arrays");
        g_frames++;
    }
}

int main(int argc, char** argv)
{
    const int scale = argc > 1 ? atoi(argv[1]) : 1;      // frames multiplier
    const int N = 1500 * scale;

    // A: K cameras, declared rigs
    for (int K : {1, 2, 3, 6}) {
        std::vector<gfo_ctx*> ctx;
        std::vector<std::thread> th;
        std::atomic<long> served{0};
        for (int k = 0; k < K; k++) { ctx.push_back(make_ctx()); ctx.push_back(make_ctx()); }
        for (int k = 0; k < K; k++) th.emplace_back(camera_adapter, k, N / K + 50, ctx[2 * k], ctx[2 * k + 1], true, &served);
        for (auto& t : th) t.join();
        fprintf(stderr, "[A] K = %d: %ld associations answered from the rig's own frame\n", K, served.load());
        EXPECT(served.load() > (long)(N / K) * K / 2, "rigs were hardly used");
        for (gfo_ctx* c : ctx) gfo_ctx_destroy(c);
    }

    // B: stereo calls and monocular frames on one engine, batches failing now and then
    {
        fake_fail_submit_every = 5;
        g_expect_overflow = true;
        std::vector<gfo_ctx*> ctx;
        std::vector<std::thread> th;
        for (int k = 0; k < 8; k++) ctx.push_back(make_ctx());
        for (int k = 0; k < 4; k++) th.emplace_back(camera_stereo_call, k, N / 2, ctx[k]);
        for (int k = 4; k < 8; k++) th.emplace_back(camera_mono, k, N / 2, ctx[k]);
        for (auto& t : th) t.join();
        int64_t cnt[8] = {0};
        gfo_combiner_counters(ctx[0], cnt, 8);
        fprintf(stderr, "[B] batches %lld requests %lld re-run alone %lld slots %lld\n", (long long)cnt[0], (long long)cnt[1], (long long)cnt[2], (long long)cnt[3]);
        EXPECT(cnt[2] > 0, "no batch failure was injected");
        fake_fail_submit_every = 0;
        g_expect_overflow = false;
        for (gfo_ctx* c : ctx) gfo_ctx_destroy(c);
    }

    // C: the partner never shows up (three solo frames, then dormant: no more waiting), then it returns
    {
        gfo_ctx* L = make_ctx();
        gfo_ctx* R = make_ctx();
        const gfo_stereo_params sp = {H, 40.f, 0.1f, 0.f};
        gfo_ctx_pair(L, R, &sp);
        std::vector<uint8_t> im(W * H);
        Arrays a;
        for (int f = 0; f < 40; f++) {
            make_image(im.data(), 555 + f);
            if (api_extract(L, im.data(), a.k, a.d, &a.n)) g_errors++;
            EXPECT(check_image(im.data(), a), "solo arrays");
            if (f % 3 == 0) gfo_ctx_pair(L, R, &sp);      // a caller that keeps declaring the rig: wakes it ever more rarely
            g_frames++;
        }
        int64_t cnt[8] = {0};
        gfo_combiner_counters(L, cnt, 8);
        fprintf(stderr, "[C] solo frames %lld\n", (long long)cnt[7]);
        EXPECT(cnt[7] >= 3 && cnt[7] < 20, "dormancy");
        std::atomic<long> served{0};
        camera_adapter(40, 200, L, R, true, &served);
        EXPECT(served.load() > 100, "the rig did not come back");
        gfo_ctx_destroy(L);
        gfo_ctx_destroy(R);
    }

    // D: the rig re-declared with other calibrations from a third thread while frames run
    {
        gfo_ctx* L = make_ctx();
        gfo_ctx* R = make_ctx();
        std::atomic<bool> stop{false};
        std::thread meddler([&] {
            int i = 0;
            while (!stop.load()) {
                const gfo_stereo_params sp = {H, 40.f + (i++ % 3), 0.1f, 0.f};      // (camera 0's own calibration is mbf = 40)
                gfo_ctx_pair(L, R, &sp);
                std::this_thread::sleep_for(std::chrono::microseconds(150));
            }
        });
        camera_adapter(0, N / 2, L, R, true, nullptr);
        stop = true;
        meddler.join();
        gfo_ctx_destroy(L);
        gfo_ctx_destroy(R);
    }

    // E: the right context destroyed and re-created while the left side waits for it
    {
        gfo_ctx* L = make_ctx();
        const gfo_stereo_params sp = {H, 40.f, 0.1f, 0.f};
        std::vector<uint8_t> im(W * H);
        Arrays a;
        for (int round = 0; round < 60; round++) {
            gfo_ctx* R = make_ctx();
            gfo_ctx_pair(L, R, &sp);
            make_image(im.data(), 777 + round);
            std::thread killer([&] {
                std::this_thread::sleep_for(std::chrono::microseconds(100 + 30 * (round % 10)));
                gfo_ctx_destroy(R);      // nobody is inside the library with R: legal, and the left side is waiting for it
            });
            if (api_extract(L, im.data(), a.k, a.d, &a.n)) g_errors++;
            EXPECT(check_image(im.data(), a), "arrays after the partner went away");
            killer.join();
            g_frames++;
        }
        gfo_ctx_destroy(L);
    }

    // F: associations of host arrays from several threads at once (kind 3), some with disparity windows
    {
        const int K = 6;
        std::vector<gfo_ctx*> ctx;
        std::vector<std::thread> th;
        for (int k = 0; k < 2 * K; k++) ctx.push_back(make_ctx());
        for (int k = 0; k < K; k++)
            th.emplace_back([&, k] {
                const gfo_stereo_params sp = {H, 40.f, 0.1f, 0.f};
                std::vector<uint8_t> il(W * H), ir(W * H);
                Arrays a, b;
                float ur[CAP], dp[CAP], mn[CAP], mx[CAP];
                int32_t bd[CAP], bi[CAP];
                for (int f = 0; f < N / 4; f++) {
                    make_image(il.data(), 3000000 + k * 100000 + 2 * f);
                    make_image(ir.data(), 3000000 + k * 100000 + 2 * f + 1);
                    std::thread tr([&] { if (api_extract(ctx[2 * k + 1], ir.data(), b.k, b.d, &b.n)) g_errors++; });
                    if (api_extract(ctx[2 * k], il.data(), a.k, a.d, &a.n)) g_errors++;
                    tr.join();
                    const bool win = (f & 1) != 0;
                    for (int j = 0; j < a.n; j++) { mn[j] = 1.f; mx[j] = 100.f + j; }
                    int nm = -1, how = 0;
                    if (api_stereo_match(ctx[2 * k], a.k, a.d, a.n, b.k, b.d, b.n, &sp, win ? mn : nullptr, win ? mx : nullptr, ur, dp, bd, bi, &nm, &how)) { g_errors++; continue; }
                    EXPECT(nm == a.n + b.n && ur[1] == a.k[1].x + b.k[0].y + (win ? mx[1] : sp.mbf / sp.mb), "host-array association");
                    g_frames++;
                }
            });
        for (auto& t : th) t.join();
        for (gfo_ctx* c : ctx) gfo_ctx_destroy(c);
    }

    fprintf(stderr, "frames %ld, wrong %ld, errors %ld, injected single-frame failures %ld, fake batches %ld, contexts still alive %ld\n", g_frames.load(), g_wrong.load(),
            g_errors.load(), g_injected.load(), fake_batches.load(), fake_contexts_alive.load());
    if (g_wrong.load() || g_errors.load() || fake_contexts_alive.load() != 0) return 1;
    printf("combine_tsan ok: %ld frames\n", g_frames.load());
    return 0;
}
