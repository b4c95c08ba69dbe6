// gfo_api_san.cc -- the HOST side of libgfo.so under AddressSanitizer + UndefinedBehaviorSanitizer, on the CPU.
//
// gf-orb-slam2_amd/csrc/gfo_api.hip and gfo_combine.hip are compiled UNMODIFIED (tests/host/san_api_tu.cc, san_combine_tu.cc include
// them) against tests/host/fakehip/hip/hip_runtime.h: "device" memory is host memory, hipMemcpy is memcpy, a kernel launch runs
// the kernel function thread by thread -- k_pack_results, k_unpack_pairs and the 16-byte copy kernel of gfo_api.hip run for real.
// The kernels of the OTHER translation units (pyramid, blur, FAST, quadtree, orientation + descriptors, stereo) are replaced by the
// stand-ins below, which write what the host code reads afterwards: per-image keypoint counts, keypoints, descriptors, association
// arrays, all a pure function of the image so that the driver can check that every caller got ITS results out of the staging
// blocks.  What this exercises is everything the ABI does on the host: argument checks, plan() (level geometry, resize tables,
// band plans, cell tables, arena sizes), the pinned staging of images and results, caller capacities, the small-batch / batch /
// device-batch / delivery paths, the stereo entry points on caller arrays, context chaining, profiling, the frame combiner and the
// stereo rigs.  Built and run by tests/test_sanitizers.py:
//   g++ -std=c++17 -O1 -g -fsanitize=address,undefined -fno-sanitize-recover=undefined -I tests/host/fakehip -x c++ ... -lpthread
// Exit code 0, "gfo_api_san ok" on stdout and no sanitizer report on stderr = pass.
#include "../../gf-orb-slam2_amd/csrc/gfo_internal.h"

#include <atomic>
#include <map>
#include <mutex>
#include <random>
#include <thread>

// ---------------------------------------------------------------------------------------------------------------------------
// the fake runtime
// ---------------------------------------------------------------------------------------------------------------------------
thread_local dim3 threadIdx, blockIdx, blockDim, gridDim;
static std::mutex g_mu;
static std::map<void*, size_t> g_dev, g_pinned;
static std::atomic<long> g_launch_stubs{0};
static thread_local int g_device = 0;

const char* hipGetErrorString(hipError_t e) { return e == hipSuccess ? "no error" : "fake error"; }
hipError_t hipGetLastError() { return hipSuccess; }
hipError_t hipPeekAtLastError() { return hipSuccess; }
hipError_t hipGetDeviceCount(int* n) { *n = 2; return hipSuccess; }
hipError_t hipSetDevice(int d) { if (d < 0 || d > 1) return hipErrorInvalidValue; g_device = d; return hipSuccess; }
hipError_t hipGetDevice(int* d) { *d = g_device; return hipSuccess; }
hipError_t hipGetDeviceProperties(hipDeviceProp_t* p, int)
{
    memset(p, 0, sizeof *p);
    snprintf(p->name, sizeof p->name, "fake MI355X");
    snprintf(p->gcnArchName, sizeof p->gcnArchName, "gfx950:sramecc+:xnack-");
    p->multiProcessorCount = 256;
    p->totalGlobalMem = (size_t)288 << 30;
    p->sharedMemPerBlock = 64 << 10;
    p->maxSharedMemoryPerMultiProcessor = 160 << 10;
    p->warpSize = 64;
    return hipSuccess;
}
hipError_t hipDeviceGetAttribute(int* v, hipDeviceAttribute_t, int) { *v = 256; return hipSuccess; }
hipError_t hipDeviceSynchronize() { return hipSuccess; }
hipError_t hipMalloc(void** p, size_t n)
{
    *p = malloc(n ? n : 1);          // exact size: an access one byte past a device buffer's computed size is an ASan report
    if (!*p) return hipErrorOutOfMemory;
    memset(*p, 0xCD, n);
    std::lock_guard<std::mutex> lk(g_mu);
    g_dev[*p] = n;
    return hipSuccess;
}
hipError_t hipFree(void* p)
{
    if (!p) return hipSuccess;
    {
        std::lock_guard<std::mutex> lk(g_mu);
        if (!g_dev.erase(p)) { fprintf(stderr, "FAKE HIP: hipFree of a pointer hipMalloc did not return\n"); abort(); }
    }
    free(p);
    return hipSuccess;
}
hipError_t hipHostMalloc(void** p, size_t n, unsigned)
{
    *p = malloc(n ? n : 1);
    if (!*p) return hipErrorOutOfMemory;
    memset(*p, 0xAB, n);
    std::lock_guard<std::mutex> lk(g_mu);
    g_pinned[*p] = n;
    return hipSuccess;
}
hipError_t hipHostFree(void* p)
{
    if (!p) return hipSuccess;
    {
        std::lock_guard<std::mutex> lk(g_mu);
        if (!g_pinned.erase(p)) { fprintf(stderr, "FAKE HIP: hipHostFree of a pointer hipHostMalloc did not return\n"); abort(); }
    }
    free(p);
    return hipSuccess;
}
hipError_t hipHostRegister(void*, size_t, unsigned) { return hipSuccess; }
hipError_t hipHostUnregister(void*) { return hipSuccess; }
hipError_t hipHostGetDevicePointer(void** d, void* h, unsigned) { *d = h; return hipSuccess; }
hipError_t hipMemcpy(void* d, const void* s, size_t n, hipMemcpyKind) { if (n) memcpy(d, s, n); return hipSuccess; }
hipError_t hipMemcpyAsync(void* d, const void* s, size_t n, hipMemcpyKind, hipStream_t) { if (n) memmove(d, s, n); return hipSuccess; }
hipError_t hipMemcpy2DAsync(void* d, size_t dp, const void* s, size_t sp, size_t w, size_t h, hipMemcpyKind, hipStream_t)
{
    for (size_t y = 0; y < h; y++) memcpy((char*)d + y * dp, (const char*)s + y * sp, w);
    return hipSuccess;
}
hipError_t hipMemset(void* d, int v, size_t n) { if (n) memset(d, v, n); return hipSuccess; }
hipError_t hipMemsetAsync(void* d, int v, size_t n, hipStream_t) { if (n) memset(d, v, n); return hipSuccess; }
struct fake_stream { int id; };
struct fake_event { int id; };
struct fake_graph { int id; };
struct fake_graph_exec { int id; };
hipError_t hipStreamCreate(hipStream_t* s) { *s = new fake_stream{1}; return hipSuccess; }
hipError_t hipStreamCreateWithFlags(hipStream_t* s, unsigned) { *s = new fake_stream{1}; return hipSuccess; }
hipError_t hipStreamDestroy(hipStream_t s) { delete s; return hipSuccess; }
hipError_t hipStreamSynchronize(hipStream_t) { return hipSuccess; }
hipError_t hipStreamQuery(hipStream_t) { return hipSuccess; }
hipError_t hipStreamWaitEvent(hipStream_t, hipEvent_t, unsigned) { return hipSuccess; }
hipError_t hipStreamBeginCapture(hipStream_t, hipStreamCaptureMode) { return hipErrorInvalidValue; }   // (GFO_GRAPH stays off)
hipError_t hipStreamEndCapture(hipStream_t, hipGraph_t* g) { *g = nullptr; return hipErrorInvalidValue; }
hipError_t hipGraphInstantiate(hipGraphExec_t* e, hipGraph_t, hipGraphNode_t*, char*, size_t) { *e = nullptr; return hipErrorInvalidValue; }
hipError_t hipGraphLaunch(hipGraphExec_t, hipStream_t) { return hipErrorInvalidValue; }
hipError_t hipGraphDestroy(hipGraph_t) { return hipSuccess; }
hipError_t hipGraphExecDestroy(hipGraphExec_t) { return hipSuccess; }
hipError_t hipEventCreate(hipEvent_t* e) { *e = new fake_event{1}; return hipSuccess; }
hipError_t hipEventCreateWithFlags(hipEvent_t* e, unsigned) { *e = new fake_event{1}; return hipSuccess; }
hipError_t hipEventDestroy(hipEvent_t e) { delete e; return hipSuccess; }
hipError_t hipEventRecord(hipEvent_t, hipStream_t) { return hipSuccess; }
hipError_t hipEventSynchronize(hipEvent_t) { return hipSuccess; }
hipError_t hipEventQuery(hipEvent_t) { return hipSuccess; }
hipError_t hipEventElapsedTime(float* ms, hipEvent_t, hipEvent_t) { *ms = 0.01f; return hipSuccess; }
hipError_t hipFuncSetAttribute(const void*, hipFuncAttribute, int) { return hipSuccess; }
hipError_t hipFuncGetAttributes(hipFuncAttributes* a, const void*) { memset(a, 0, sizeof *a); a->maxThreadsPerBlock = 1024; return hipSuccess; }

// ---------------------------------------------------------------------------------------------------------------------------
// stand-ins for the kernels of the other translation units
// ---------------------------------------------------------------------------------------------------------------------------
void gfo_kernels_fast(std::vector<const void*>&) {}
void gfo_kernels_pyramid(std::vector<const void*>&) {}
void gfo_kernels_blur(std::vector<const void*>&) {}
void gfo_kernels_quadtree(std::vector<const void*>&) {}
void gfo_kernels_orient_desc(std::vector<const void*>&) {}
void gfo_kernels_stereo(std::vector<const void*>&) {}
void gfo_kernels_project(std::vector<const void*>&) {}
void gfo_kernels_bow(std::vector<const void*>&) {}
size_t gfo_quadtree_lds_bytes(int ncap, int klds) { return (size_t)ncap * 16 + (size_t)klds * 4 + 256; }
int gfo_pyramid_bands_prepare(int) { return 0; }
int gfo_few_max() { return 16; }

// every byte of the image `img` of the batch, through the level-0 description the launchers get (base, pitch, stride): reading
// it here is what checks that plan() / the staging code described the input correctly
static uint32_t image_hash(const gfo_ctx* c, const GfoInput& in, int img)
{
    uint32_t h = 2166136261u;
    for (int y = 0; y < c->g.h0; y++) {
        const uint8_t* row = in.base + (long long)img * in.img_stride + (long long)y * in.pitch;
        for (int x = 0; x < c->g.w0; x++) h = (h ^ row[x]) * 16777619u;
    }
    return h;
}
// the pyramid stand-in writes every plane it owns from end to end (a plane the arena is too small for is an ASan report)
static void touch_pyramid(gfo_ctx* c, int nimg)
{
    for (int i = 0; i < nimg; i++)
        for (int l = 1; l < c->g.nlevels; l++) {
            const GfoLevel& L = c->g.lv[l];
            uint8_t* p = c->d_pyr + (long long)i * c->g.pyr_img_stride + L.plane_off;
            for (int y = 0; y < L.h; y++) memset(p + (long long)y * L.pitch, (l * 16 + y) & 255, L.pitch);
        }
}
void gfo_launch_resize(gfo_ctx* c, const GfoInput&, int level, int nimg)
{
    g_launch_stubs++;
    if (level == 1) touch_pyramid(c, nimg);
    (void)gfo_take_zero_cnt(c);
}
void gfo_launch_resize_tail(gfo_ctx* c, const GfoInput&, int, int) { g_launch_stubs++; (void)gfo_take_zero_cnt(c); }
void gfo_launch_pyramid_bands(gfo_ctx* c, const GfoInput&, int nimg) { g_launch_stubs++; touch_pyramid(c, nimg); (void)gfo_take_zero_cnt(c); }
void gfo_launch_blur(gfo_ctx* c, const GfoInput&, int nimg)
{
    g_launch_stubs++;
    for (int i = 0; i < nimg; i++)
        for (int l = 0; l < c->g.nlevels; l++) {
            const GfoLevel& L = c->g.lv[l];
            uint8_t* p = c->d_blur + (long long)i * c->g.blur_img_stride + L.blur_off;
            for (int y = 0; y < L.h; y++) memset(p + (long long)y * L.pitch, 7, L.w);
        }
}
void gfo_launch_fast(gfo_ctx* c, const GfoInput&, int nimg)
{
    g_launch_stubs++;
    for (int i = 0; i < nimg; i++)
        for (int l = 0; l < c->g.nlevels; l++) c->d_cand_cnt[(i * c->g.nlevels + l) * GFO_CNT_STRIDE] = 0;
}
void gfo_launch_quadtree(gfo_ctx* c, int nimg)
{
    g_launch_stubs++;
    for (int i = 0; i < nimg * c->g.nlevels; i++) c->d_sel_cnt[i] = 0;
}
bool gfo_launch_quadtree_blur(gfo_ctx*, const GfoInput&, int) { return false; }
// keypoints of image i: n = hash % (stride + 1) capped, keypoint j = f(hash, j), descriptor byte = f(hash, j, b)
static int fake_count(uint32_t h, int stride) { return (int)(h % (uint32_t)(stride < 300 ? stride + 1 : 301)); }
void gfo_launch_orient_desc(gfo_ctx* c, const GfoInput& in, int nimg)
{
    g_launch_stubs++;
    const int ks = c->g.kp_stride;
    for (int i = 0; i < nimg; i++) {
        const uint32_t h = image_hash(c, in, i);
        const int n = fake_count(h, ks);
        c->d_kp_cnt[i] = n;
        for (int j = 0; j < n; j++) {
            gfo_keypoint& k = c->d_kp[(long long)i * ks + j];
            k.x = (float)((h + 7u * j) % 700u) + 16.f; k.y = (float)((h / 3u + 13u * j) % 400u) + 16.f; k.size = 31.f; k.angle = (float)(j % 360);
            k.response = (float)(j & 255); k.octave = j % c->g.nlevels; k.class_id = -1;
            for (int b = 0; b < 32; b++) c->d_desc[((long long)i * ks + j) * 32 + b] = (uint8_t)(h + 31u * j + b);
        }
    }
    memset(c->d_flags, 0, 4 * sizeof(int));
}
void gfo_launch_stereo(gfo_ctx*, const GfoStereoLaunch& s)
{
    g_launch_stubs++;
    for (int p = 0; p < s.npairs; p++) {
        const int nl = s.cnt_dev ? s.cnt_dev[2 * p] : s.nl_host;
        const gfo_keypoint* kl = s.kl + p * s.pair_stride_kp;
        int nm = 0;
        for (int i = 0; i < nl; i++) {
            const long long o = (long long)p * s.out_stride + i;
            const bool hit = ((int)kl[i].x + i) % 3 != 0;
            s.out.u_right[o] = hit ? kl[i].x - 5.f : -1.f;
            s.out.depth[o] = hit ? 9.5f : -1.f;
            s.out.best_dist[o] = hit ? 30 + i % 40 : -1;
            s.out.best_idx[o] = hit ? i % 7 : -1;
            if (s.out.counted) s.out.counted[o] = 1;
            nm += hit;
            if (s.min_d && s.max_d && s.min_d[p * s.win_stride + i] > s.max_d[p * s.win_stride + i]) abort();   // (reads the windows end to end)
        }
        s.out.nmatched[p] = nm;
        // the sort buffers are sized for the right side: touch them
        const int nr = s.cnt_dev ? s.cnt_dev[2 * p + 1] : s.nr_host;
        for (int j = 0; j < nr; j++) { s.sort.sx[(long long)p * s.sort_stride + j] = 0.f; s.sort.soi[(long long)p * s.sort_stride + j] = 0; }
        for (int r = 0; r <= s.p.n_rows; r++) s.sort.row_start[(long long)p * (s.p.n_rows + 1) + r] = 0;
    }
}
void gfo_launch_pack_cut(gfo_ctx* c, const GfoPack& p, hipStream_t st)
{
    g_launch_stubs++;
    for (int s = 0; s < p.nseg; s++)
        for (int k = 0; k < p.n16[s]; k++) p.dst[s][k] = p.src[s][k];
    for (int q = 0; q < p.cut_pairs; q++) {
        const int nl = p.cut_cnt_dev ? p.cut_cnt_dev[2 * q] : p.cut_nl_host;
        if (p.h_u_right) memcpy(p.h_u_right + (long long)q * p.cut_out_stride, p.cut_out.u_right + (long long)q * p.cut_out_stride, sizeof(float) * nl);
        if (p.h_depth) memcpy(p.h_depth + (long long)q * p.cut_out_stride, p.cut_out.depth + (long long)q * p.cut_out_stride, sizeof(float) * nl);
        if (p.h_nmatched) p.h_nmatched[q] = p.cut_out.nmatched[q];
    }
    (void)c; (void)st;
}
int gfo_stereo_window(const float* scale, int nlevels) { return (int)(2.f * scale[nlevels - 1]) + 2; }
void gfo_launch_stereo_sad(gfo_ctx*, const GfoStereoLaunch& s, const GfoInput&, const float*)
{
    g_launch_stubs++;
    for (int p = 0; p < s.npairs; p++) s.out.nmatched[p] = 0;
}

// ---------------------------------------------------------------------------------------------------------------------------
// the driver: the C ABI as its callers use it
// ---------------------------------------------------------------------------------------------------------------------------
static int g_fail = 0;
#define CHECK(cond, ...)                                                                  \
    do {                                                                                  \
        if (!(cond)) {                                                                    \
            fprintf(stderr, "[gfo_api_san] CHECK FAILED %s:%d: ", __FILE__, __LINE__);    \
            fprintf(stderr, __VA_ARGS__);                                                 \
            fprintf(stderr, "\n");                                                        \
            g_fail++;                                                                     \
        }                                                                                 \
    } while (0)

static uint32_t host_hash(const uint8_t* img, int w, int h, int stride)
{
    uint32_t x = 2166136261u;
    for (int y = 0; y < h; y++)
        for (int i = 0; i < w; i++) x = (x ^ img[(size_t)y * stride + i]) * 16777619u;
    return x;
}
static bool results_belong_to(const uint8_t* img, int w, int h, int stride, const gfo_keypoint* kp, const uint8_t* desc, int n, int ks, int nlevels)
{
    const uint32_t hh = host_hash(img, w, h, stride);
    if (n != fake_count(hh, ks)) return false;
    for (int j = 0; j < n; j++) {
        if (kp[j].x != (float)((hh + 7u * j) % 700u) + 16.f || kp[j].octave != j % nlevels) return false;
        for (int b = 0; b < 32; b++)
            if (desc[(size_t)j * 32 + b] != (uint8_t)(hh + 31u * j + b)) return false;
    }
    return true;
}

int main()
{
    std::mt19937 rng(5);
    // ---- parameters the constructor must refuse before it touches a device
    {
        gfo_ctx* c = nullptr;
        const gfo_params bad[] = {{2000, 1.2f, 0, 20, 7, 1}, {2000, 1.2f, 17, 20, 7, 1}, {0, 1.2f, 8, 20, 7, 1}, {2000, 1.0f, 8, 20, 7, 1},
                                  {2000, 1.2f, 8, 0, 7, 1}, {2000, 1.2f, 8, 20, 300, 1}};   // (max_batch < 1 is taken as 1)
        for (const gfo_params& p : bad) CHECK(gfo_ctx_create(&p, 0, &c) == GFO_ERR_INVALID && c == nullptr, "a bad parameter set was accepted");
        gfo_params ok = {2000, 1.2f, 8, 20, 7, 1};
        CHECK(gfo_ctx_create(&ok, 7, &c) != GFO_OK, "device 7 of 2 was accepted");
        CHECK(gfo_ctx_create(nullptr, 0, &c) == GFO_ERR_INVALID && gfo_ctx_create(&ok, 0, nullptr) == GFO_ERR_INVALID, "null arguments");
    }
    // ---- one context through image sizes, strides and capacities: every plan() is a new arena and new tables
    const int sizes[][2] = {{752, 480}, {640, 480}, {1920, 1080}, {97, 61}, {33, 200}, {1241, 376}, {40, 40}, {19, 19}, {4000, 31}, {752, 480}};
    for (int nfeat : {2000, 500, 7}) {
        for (int nlev : {8, 1, 12}) {
            gfo_params prm = {nfeat, nlev == 12 ? 1.1f : 1.2f, nlev, 20, 7, 1};
            gfo_ctx* c = nullptr;
            CHECK(gfo_ctx_create(&prm, 0, &c) == GFO_OK && c, "create: %s", gfo_last_error(nullptr));
            if (!c) continue;
            std::vector<float> t(4 * nlev);
            std::vector<int> q(nlev);
            CHECK(gfo_ctx_tables(c, t.data(), t.data() + nlev, t.data() + 2 * nlev, t.data() + 3 * nlev, q.data()) == GFO_OK, "tables");
            for (const auto& sz : sizes) {
                const int w = sz[0], h = sz[1], stride = w + (int)(rng() % 3) * 8;
                std::vector<uint8_t> img((size_t)stride * h);
                for (auto& b : img) b = (uint8_t)rng();
                const int cap_full = gfo_ctx_max_keypoints(c) > 0 ? gfo_ctx_max_keypoints(c) : nfeat + 64;
                std::vector<gfo_keypoint> kp(cap_full + 1);
                std::vector<uint8_t> desc((size_t)(cap_full + 1) * 32);
                int n = -1;
                int rc = gfo_extract(c, img.data(), w, h, stride, kp.data(), desc.data(), cap_full, &n);
                if (rc == GFO_ERR_CAPACITY) {      // the arena was re-planned for this size and holds more keypoints per image than the last one
                    const int cap2 = gfo_ctx_max_keypoints(c);
                    CHECK(cap2 > cap_full, "GFO_ERR_CAPACITY although the capacity %d covers the arena's %d", cap_full, cap2);
                    kp.resize(cap2 + 1);
                    desc.resize((size_t)(cap2 + 1) * 32);
                    rc = gfo_extract(c, img.data(), w, h, stride, kp.data(), desc.data(), cap2, &n);
                }
                if (w > 4000 || h > 4000) { CHECK(rc == GFO_ERR_INVALID, "an image beyond the coordinate packing was accepted"); continue; }
                CHECK(rc == GFO_OK, "gfo_extract %dx%d (%d features, %d levels): %s", w, h, nfeat, nlev, gfo_last_error(c));
                if (rc != GFO_OK) continue;
                const int ks = gfo_ctx_max_keypoints(c);
                CHECK(results_belong_to(img.data(), w, h, stride, kp.data(), desc.data(), n, ks, nlev), "gfo_extract %dx%d returned another image's results (n %d)", w, h, n);
                // a capacity one short of the count: refused, nothing written past it
                if (n > 1) {
                    std::vector<gfo_keypoint> k2(n - 1);
                    std::vector<uint8_t> d2((size_t)(n - 1) * 32);
                    int n2 = -1;
                    CHECK(gfo_extract(c, img.data(), w, h, stride, k2.data(), d2.data(), n - 1, &n2) == GFO_ERR_CAPACITY, "cap n-1 must be refused");
                }
                // the pyramid entry points: every level with and without the 19-px frame, exact-size output buffers
                CHECK(gfo_compute_pyramid(c, img.data(), w, h, stride) == GFO_OK, "compute_pyramid: %s", gfo_last_error(c));
                for (int l = 0; l < nlev; l++)
                    for (int border : {0, 19}) {
                        int lw = 0, lh = 0;
                        std::vector<uint8_t> probe((size_t)(w + 38) * (h + 38));
                        if (gfo_pyramid_level(c, 0, l, border, probe.data(), w + 38, &lw, &lh) != GFO_OK) continue;
                        std::vector<uint8_t> exact((size_t)(lw + 2 * border) * (lh + 2 * border));
                        CHECK(exact.empty() || gfo_pyramid_level(c, 0, l, border, exact.data(), lw + 2 * border, &lw, &lh) == GFO_OK, "pyramid_level exact buffer");
                        CHECK(gfo_pyramid_level(c, 0, l, border, exact.data(), lw + 2 * border - 1, &lw, &lh) != GFO_OK || lw == 0, "a short out_stride was accepted");
                    }
                CHECK(gfo_pyramid_level(c, 0, nlev, 0, img.data(), w, &n, &n) == GFO_ERR_INVALID && gfo_pyramid_level(c, 1, 0, 0, img.data(), w, &n, &n) == GFO_ERR_INVALID,
                      "level / image index out of range");
            }
            // bad calls on a live context
            int n = 0;
            uint8_t px[64] = {0};
            gfo_keypoint k1[4];
            uint8_t d1[128];
            // an empty image is not an error (ORBextractor.cc:1115-1116: return, outputs untouched): zero keypoints
            n = 99;
            CHECK(gfo_extract(c, nullptr, 8, 8, 8, k1, d1, 4, &n) == GFO_OK && n == 0, "null image");
            n = 99;
            CHECK(gfo_extract(c, px, 0, 8, 8, k1, d1, 4, &n) == GFO_OK && n == 0, "zero width");
            CHECK(gfo_extract(c, px, 8, 8, 4, k1, d1, 4, &n) == GFO_ERR_INVALID, "a stride below the width was accepted");
            CHECK(gfo_extract(c, px, 8, 8, 8, k1, d1, -1, &n) == GFO_ERR_INVALID, "a negative capacity was accepted");
            CHECK(gfo_extract(c, px, 8, 8, 8, k1, d1, 4, nullptr) == GFO_ERR_INVALID, "a null count pointer was accepted");
            gfo_ctx_destroy(c);
        }
    }
    // ---- host batches, stereo frames, the device-batch path with fetch / deliver, profiling, chaining
    {
        gfo_params prm = {1000, 1.2f, 8, 20, 7, 40};
        gfo_ctx *c = nullptr, *c2 = nullptr;
        CHECK(gfo_ctx_create(&prm, 0, &c) == GFO_OK && gfo_ctx_create(&prm, 0, &c2) == GFO_OK, "create");
        CHECK(gfo_profile_enable(c, 1) == GFO_OK, "profile_enable");
        CHECK(gfo_ctx_chain(c2, c, 1) == GFO_OK && gfo_ctx_chain(c2, c2, 1) == GFO_ERR_INVALID && gfo_ctx_chain(c2, c, 99) == GFO_ERR_INVALID, "chain");
        const int w = 320, h = 240;
        for (int nimg : {1, 2, 3, 8, 17, 40}) {
            std::vector<std::vector<uint8_t>> imgs(nimg, std::vector<uint8_t>((size_t)w * h));
            std::vector<const uint8_t*> ptr(nimg);
            for (int i = 0; i < nimg; i++) { for (auto& b : imgs[i]) b = (uint8_t)rng(); ptr[i] = imgs[i].data(); }
            const int cap = 1100;
            std::vector<gfo_keypoint> kp((size_t)nimg * cap);
            std::vector<uint8_t> desc((size_t)nimg * cap * 32);
            std::vector<int> n(nimg, -1);
            const int rc = gfo_extract_batch(c, ptr.data(), nimg, w, h, w, kp.data(), desc.data(), cap, n.data());
            CHECK(rc == GFO_OK, "extract_batch(%d): %s", nimg, gfo_last_error(c));
            const int ks = gfo_ctx_max_keypoints(c);
            for (int i = 0; rc == GFO_OK && i < nimg; i++)
                CHECK(results_belong_to(ptr[i], w, h, w, kp.data() + (size_t)i * cap, desc.data() + (size_t)i * cap * 32, n[i], ks, 8), "batch %d image %d: wrong results", nimg, i);
            if (nimg >= 2 && (nimg & 1) == 0) {
                gfo_stereo_params sp = {h, 47.9f, 0.11f, 0.f};
                CHECK(gfo_stereo_match_batch(c, &sp) == GFO_OK, "stereo_match_batch: %s", gfo_last_error(c));
                std::vector<float> ur(cap), dp(cap);
                std::vector<int32_t> bd(cap), bi(cap);
                int nm = -1;
                CHECK(gfo_stereo_fetch(c, nimg / 2 - 1, ur.data(), dp.data(), bd.data(), bi.data(), cap, &nm) == GFO_OK && nm >= 0, "stereo_fetch");
                CHECK(gfo_stereo_fetch(c, nimg / 2, ur.data(), dp.data(), bd.data(), bi.data(), cap, &nm) == GFO_ERR_INVALID, "stereo_fetch pair out of range");
                gfo_delivery lay;
                CHECK(gfo_batch_deliver(c, nullptr, 0, &lay) == GFO_OK && lay.bytes > 0, "deliver layout");
                std::vector<uint8_t> block(lay.bytes);
                CHECK(gfo_batch_deliver(c, block.data(), lay.bytes, &lay) == GFO_OK && gfo_deliver_wait(c) == GFO_OK, "deliver: %s", gfo_last_error(c));
                CHECK(gfo_batch_deliver(c, block.data(), lay.bytes - 1, &lay) == GFO_ERR_CAPACITY, "a short delivery block was accepted");
            }
            // one stereo frame per call
            if (nimg >= 2) {
                gfo_stereo_params sp = {h, 47.9f, 0.11f, 0.f};
                std::vector<gfo_keypoint> kl(cap), kr(cap);
                std::vector<uint8_t> dl((size_t)cap * 32), dr((size_t)cap * 32);
                std::vector<float> ur(cap), dp(cap);
                std::vector<int32_t> bd(cap), bi(cap);
                int nl = -1, nr = -1, nm = -1;
                CHECK(gfo_extract_stereo(c2, ptr[0], ptr[1], w, h, w, &sp, kl.data(), dl.data(), kr.data(), dr.data(), cap, &nl, &nr, ur.data(), dp.data(), bd.data(),
                                         bi.data(), &nm) == GFO_OK, "extract_stereo: %s", gfo_last_error(c2));
                CHECK(results_belong_to(ptr[0], w, h, w, kl.data(), dl.data(), nl, gfo_ctx_max_keypoints(c2), 8) &&
                      results_belong_to(ptr[1], w, h, w, kr.data(), dr.data(), nr, gfo_ctx_max_keypoints(c2), 8), "extract_stereo: wrong results");
            }
        }
        // the device-batch path: a "device" buffer with a pitch, fetch per image with exact capacities
        {
            const int nimg = 6, pitch = 384;
            std::vector<uint8_t> dev((size_t)nimg * pitch * h + 64);
            for (auto& b : dev) b = (uint8_t)rng();
            CHECK(gfo_extract_batch_device(c, dev.data(), nimg, w, h, pitch, (size_t)pitch * h) == GFO_OK, "batch_device: %s", gfo_last_error(c));
            CHECK(gfo_extract_batch_device(c, dev.data(), nimg, w, h, w - 1, (size_t)pitch * h) == GFO_ERR_INVALID &&
                  gfo_extract_batch_device(c, dev.data(), nimg, w, h, pitch, (size_t)pitch * h - 1) == GFO_ERR_INVALID, "batch_device bad pitch / stride");
            std::vector<int> cnt(nimg), per(nimg * 8);
            CHECK(gfo_batch_counts(c, cnt.data(), per.data()) == GFO_OK, "batch_counts");
            for (int i = 0; i < nimg; i++) {
                std::vector<gfo_keypoint> kp(cnt[i] > 0 ? cnt[i] : 1);
                std::vector<uint8_t> desc((size_t)(cnt[i] > 0 ? cnt[i] : 1) * 32);
                int n = -1;
                CHECK(gfo_batch_fetch(c, i, kp.data(), desc.data(), cnt[i], &n) == GFO_OK && n == cnt[i], "batch_fetch image %d", i);
                CHECK(results_belong_to(dev.data() + (size_t)i * pitch * h, w, h, pitch, kp.data(), desc.data(), n, gfo_ctx_max_keypoints(c), 8), "batch_fetch: wrong results");
            }
            int n = 0;
            CHECK(gfo_batch_fetch(c, nimg, nullptr, nullptr, 0, &n) == GFO_ERR_INVALID, "batch_fetch image out of range");
        }
        gfo_stage_time st[16];
        int ns = 0;
        CHECK(gfo_profile_read(c, st, 16, &ns, 1) == GFO_OK && ns > 0, "profile_read");
        CHECK(gfo_ctx_chain(c2, nullptr, 0) == GFO_OK, "unchain");
        gfo_ctx_destroy(c2);
        gfo_ctx_destroy(c);
    }
    // ---- the stereo association on caller arrays: sizes from 0 to the 65535 limit, windows, bad octaves
    {
        gfo_params prm = {2000, 1.2f, 8, 20, 7, 1};
        gfo_ctx* c = nullptr;
        CHECK(gfo_ctx_create(&prm, 0, &c) == GFO_OK, "create");
        std::vector<float> sf(8, 1.f);
        for (int i = 1; i < 8; i++) sf[i] = sf[i - 1] * 1.2f;
        for (int nl : {0, 1, 33, 2008, 20000}) {
            for (int nr : {0, 5, 2002, 65535}) {
                std::vector<gfo_keypoint> kl(nl ? nl : 1), kr(nr ? nr : 1);
                std::vector<uint8_t> dl((size_t)(nl ? nl : 1) * 32, 1), dr((size_t)(nr ? nr : 1) * 32, 2);
                for (int i = 0; i < nl; i++) { kl[i] = gfo_keypoint{(float)(i % 700), (float)(i % 470), 31.f, 0.f, 1.f, i % 8, -1}; }
                for (int i = 0; i < nr; i++) { kr[i] = gfo_keypoint{(float)(i % 690), (float)(i % 475), 31.f, 0.f, 1.f, i % 8, -1}; }
                std::vector<float> ur(nl ? nl : 1, 777.f), dp(nl ? nl : 1, 777.f), mn(nl ? nl : 1, 0.f), mx(nl ? nl : 1, 400.f);
                std::vector<int32_t> bd(nl ? nl : 1), bi(nl ? nl : 1);
                gfo_stereo_params sp = {480, 47.9f, 0.11f, 0.f};
                int nm = -1;
                for (int win = 0; win < 2; win++) {
                    const int rc = gfo_stereo_match(c, kl.data(), dl.data(), nl, kr.data(), dr.data(), nr, sf.data(), 8, &sp, win ? mn.data() : nullptr, win ? mx.data() : nullptr,
                                                    ur.data(), dp.data(), bd.data(), bi.data(), &nm);
                    CHECK(rc == GFO_OK, "stereo_match(%d, %d): %s", nl, nr, gfo_last_error(c));
                    int want = 0;
                    for (int i = 0; i < nl; i++) want += ((int)kl[i].x + i) % 3 != 0;
                    CHECK(rc != GFO_OK || nm == want, "stereo_match(%d, %d): nmatched %d, the stand-in wrote %d", nl, nr, nm, want);
                }
                if (nl > 2) {
                    kl[2].octave = 8;
                    CHECK(gfo_stereo_match(c, kl.data(), dl.data(), nl, kr.data(), dr.data(), nr, sf.data(), 8, &sp, nullptr, nullptr, ur.data(), dp.data(), bd.data(), bi.data(),
                                           &nm) == GFO_ERR_INVALID, "a left octave outside the table was accepted");
                }
            }
        }
        std::vector<gfo_keypoint> k(4);
        std::vector<uint8_t> d(128);
        std::vector<float> f(4);
        int nm;
        gfo_stereo_params sp = {480, 47.9f, 0.11f, 0.f};
        CHECK(gfo_stereo_match(c, k.data(), d.data(), 4, k.data(), d.data(), 65536, sf.data(), 8, &sp, nullptr, nullptr, f.data(), f.data(), nullptr, nullptr, &nm) == GFO_ERR_INVALID,
              "65536 right keypoints were accepted");
        gfo_ctx_destroy(c);
    }
    // ---- the frame combiner and a stereo rig from several threads (the product's gfo_combine.hip on top of the product's gfo_api.hip)
    {
        const int K = 4, frames = 30, w = 200, h = 120;
        gfo_params prm = {300, 1.2f, 6, 20, 7, 1};
        std::vector<gfo_ctx*> cl(K), cr(K);
        for (int k = 0; k < K; k++) {
            CHECK(gfo_ctx_create(&prm, 0, &cl[k]) == GFO_OK && gfo_ctx_create(&prm, 0, &cr[k]) == GFO_OK, "create");
            gfo_ctx_set_combining(cl[k], 1);
            gfo_ctx_set_combining(cr[k], 1);
        }
        std::atomic<int> wrong{0};
        std::vector<std::thread> cams;
        for (int k = 0; k < K; k++)
            cams.emplace_back([&, k] {
                std::mt19937 r(100 + k);
                gfo_stereo_params sp = {h, 47.9f, 0.11f, 0.f};
                for (int f = 0; f < frames; f++) {
                    std::vector<uint8_t> il((size_t)w * h), ir((size_t)w * h);
                    for (auto& b : il) b = (uint8_t)r();
                    for (auto& b : ir) b = (uint8_t)r();
                    const int cap = 400;
                    std::vector<gfo_keypoint> kl(cap), kr(cap);
                    std::vector<uint8_t> dl((size_t)cap * 32), dr((size_t)cap * 32);
                    int nl = -1, nr = -1;
                    std::thread tr([&] { if (gfo_extract(cr[k], ir.data(), w, h, w, kr.data(), dr.data(), cap, &nr) != GFO_OK) wrong++; });
                    if (gfo_extract(cl[k], il.data(), w, h, w, kl.data(), dl.data(), cap, &nl) != GFO_OK) wrong++;
                    tr.join();
                    if (f == 1) (void)gfo_ctx_pair(cl[k], cr[k], &sp);
                    const int ks = gfo_ctx_max_keypoints(cl[k]);
                    if (nl < 0 || nr < 0 || !results_belong_to(il.data(), w, h, w, kl.data(), dl.data(), nl, ks, 6) || !results_belong_to(ir.data(), w, h, w, kr.data(), dr.data(), nr, ks, 6)) {
                        wrong++;
                        continue;
                    }
                    std::vector<float> sf(6, 1.f), ur(nl ? nl : 1), dp(nl ? nl : 1);
                    for (int i = 1; i < 6; i++) sf[i] = sf[i - 1] * 1.2f;
                    std::vector<int32_t> bd(nl ? nl : 1), bi(nl ? nl : 1);
                    int nm = -1;
                    if (gfo_stereo_match(cl[k], kl.data(), dl.data(), nl, kr.data(), dr.data(), nr, sf.data(), 6, &sp, nullptr, nullptr, ur.data(), dp.data(), bd.data(), bi.data(), &nm) != GFO_OK)
                        wrong++;
                    int want = 0;
                    for (int i = 0; i < nl; i++) want += ((int)kl[i].x + i) % 3 != 0;
                    if (nm != want) wrong++;
                }
            });
        for (auto& t : cams) t.join();
        CHECK(wrong.load() == 0, "combiner / rigs: %d wrong frames", wrong.load());
        int64_t cnt[8] = {0};
        CHECK(gfo_combiner_counters(cl[0], cnt, 8) == GFO_OK && cnt[1] > 0, "combiner counters");
        for (int k = 0; k < K; k++) { gfo_ctx_destroy(cr[k]); gfo_ctx_destroy(cl[k]); }
    }
    // ---- pinned registration of caller buffers and the helpers
    {
        std::vector<uint8_t> buf(1 << 16);
        CHECK(gfo_host_register(buf.data(), buf.size()) == GFO_OK && gfo_host_unregister(buf.data()) == GFO_OK, "host_register");
        CHECK(gfo_host_register(nullptr, 16) == GFO_ERR_INVALID, "host_register(null)");
        uint8_t a[32], b[32];
        memset(a, 0xF0, 32); memset(b, 0x0F, 32);
        CHECK(gfo_hamming256(a, b) == 256 && gfo_hamming256(a, a) == 0, "hamming");
        CHECK(gfo_version() >= 100 && gfo_build_variant(0) == 0 && gfo_build_variant(6) == 55 && gfo_build_variant(42) == -1, "version / variant");
    }
    {
        std::lock_guard<std::mutex> lk(g_mu);
        CHECK(g_dev.empty() && g_pinned.empty(), "%zu device and %zu pinned allocations were never freed", g_dev.size(), g_pinned.size());
    }
    if (g_fail) { fprintf(stderr, "gfo_api_san FAILED: %d checks\n", g_fail); return 1; }
    printf("gfo_api_san ok: %ld kernel stand-ins ran\n", g_launch_stubs.load());
    return 0;
}
