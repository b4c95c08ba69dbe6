// gfo_combine.hip -- the frame combiner: per-frame calls of several host threads executed as ONE device batch.
//
// The reference extracts one frame per call (Frame::Frame, src/Frame.cc:84-100; ORBextractor::operator(),
// src/ORBextractor.cc:1112-1174), and a process that tracks K cameras does so from K threads.  Each such call is ~10
// kernels of 7-50 us that fill a fraction of the chip; K contexts on K streams only overlap as far as the runtime's four
// hardware queues let them (measured, MI355X, K = 8 stereo streams: never more than 4 kernels in flight, 2.0 on average,
// 92 us per stereo frame = 21.6 k images/s however large K is; more queues -- GPU_MAX_HW_QUEUES=8..32 -- made it worse:
// profiles/boundary_trace_r03.txt).  Every stage of this library is already one launch over all images of a batch, so
// the combiner turns concurrency into batch size instead of into streams:
//
//   * contexts that opted in (gfo_ctx_set_combining) with equal extractor parameters, device and image size share an
//     engine: a few SLOTS, each a batch context (arena for GFO_COMBINE_MAX images + pinned input / result buffers);
//   * a call joins the batch that is FORMING (or opens one and becomes its leader), copies its own image(s) into the
//     slot's pinned input -- all callers do that in parallel -- and waits;
//   * the leader closes the batch as soon as fewer than two batches are on the device, waits for the joiners' copies,
//     and submits the whole batch like any small host batch (gfo_small_submit: one H2D, the launches, one pack kernel, one
//     synchronisation); whoever arrives meanwhile forms the next batch: the batch size follows the load by itself,
//     one caller alone runs a batch of one with the latency of the direct path;
//   * everybody copies their own results out of the slot's pinned buffer, the last one frees the slot.
//
// Results do not depend on the company a frame keeps: every kernel treats the images of a batch independently (that is
// what the batched parity tests establish), so a combined call returns bit for bit what the direct call returns
// (tests/test_gpu_combine.py, and the checksums of tools/c/boundary_throughput.c under load).
#include "gfo_internal.h"

#include <atomic>
#include <chrono>
#include <condition_variable>
#include <map>
#include <memory>
#include <mutex>
#include <string.h>
#include <tuple>

// a polite spin-wait step of the host CPU (this file is also parsed for the device, where the x86 builtin does not exist)
#if defined(__HIP_DEVICE_COMPILE__) || !(defined(__x86_64__) || defined(__i386__))
#define GFO_CPU_RELAX() ((void)0)
#else
#define GFO_CPU_RELAX() __builtin_ia32_pause()
#endif

namespace
{
enum { SLOT_FREE = 0, SLOT_FORMING, SLOT_CLOSED, SLOT_RUNNING, SLOT_DRAINING };
const int NSLOT = 8;          // upper bound; an engine uses `inflight + 4` of them: batches on the device, one forming per kind, one being collected

struct Slot {
    gfo_ctx* bc = nullptr;    // the batch context: arena + pinned buffers + stream
    uint8_t* d_pairs = nullptr;   // device staging of the host-array stereo batches (kind 3): one block per pair
    GfoSmallLayout L{};
    int state = SLOT_FREE;
    int kind = 0;             // images per request: 1 = gfo_extract, 2 = gfo_extract_stereo
    gfo_stereo_params sp{};
    int n = 0, staged = 0, readers = 0;
    unsigned gen = 0;         // batches this slot has completed (a joiner waits for ITS batch)
    int rc = 0;
    std::string err;
    std::condition_variable cv;   // the members of this slot's batch wait here (leader: for the copies; joiners: for the results)
};
}  // namespace

struct GfoEngine {
    std::mutex mu;
    std::condition_variable cv;   // callers that cannot join or open a batch, leaders waiting for room on the device
    int inflight = 2, nslot = 4;
    int device = 0, w = 0, h = 0;
    gfo_params prm{};
    int cap_images = 32;      // images per batch (GFO_COMBINE_MAX)
    Slot slot[NSLOT];
    int forming[4] = {-1, -1, -1, -1};   // per kind of request: the slot that accepts joiners (an extraction batch that is forming
                                         // must not hold up the association calls of the threads that are a phase ahead, and vice versa)
    int running = 0;          // batches submitted and not yet complete
    int members = 0;          // contexts that have attached to this engine so far (engine_for)
    int prepared = 0;         // slots that have their batch context (prepared lazily: two by the first request, one more whenever
                              // a caller finds every prepared slot busy -- concurrent callers observed -- up to nslot)
    int fail_prepare_from = -1, fail_batch_every = 0;   // tests only: injected failures (slot_prepare, run_request)
    bool grow_failed = false; // a later slot could not be prepared: the engine keeps the slots it has
    bool broken = false;      // a slot could not be prepared (out of memory) and none exists: every caller takes the direct path
    long batches = 0, requests = 0, redone = 0;
    ~GfoEngine()
    {
        for (Slot& s : slot) {
            if (s.d_pairs) (void)hipFree(s.d_pairs);
            if (s.bc) gfo_ctx_destroy(s.bc);
        }
    }
};

// Two extractor contexts that the caller declared to be the left and right camera of one stereo rig (gfo_ctx_pair): the
// reference's mpORBextractorLeft / mpORBextractorRight, called from two threads per frame (Frame.cc:84-87), followed by
// Frame::ComputeStereoMatches on the arrays the two calls returned (Frame.cc:100).  The two gfo_extract calls of a frame meet
// here, go to the engine as ONE stereo request (what gfo_extract_stereo would have submitted), and the association the third
// call is going to ask for is computed in the same submission and kept; gfo_stereo_match then only has to establish that it is
// being asked about exactly those arrays.
struct GfoPair {
    std::mutex mu;
    std::condition_variable cv;
    gfo_ctx* ctx[2] = {nullptr, nullptr};
    gfo_stereo_params sp{};
    struct Req {
        const uint8_t* img = nullptr;
        int w = 0, h = 0;
        gfo_keypoint* kp = nullptr;
        uint8_t* desc = nullptr;
        int cap = 0, rc = 0;
        int* n = nullptr;
        bool arrived = false, staged = false;
    } req[2];
    int state = 0;            // 0 idle, 1 one side waits for its partner, 2 the second arrival is executing the frame, 4 the batch is
                              // complete: the waiting side copies its arrays out of the result block (pick_*), 3 the batch failed:
                              // the waiting side takes its status
    std::atomic<int> state_seen{0};   // mirror of `state` for the waiting side's short spin (below): written under mu, read without
    bool spin_ok = false;     // one camera on this engine: the waiting side may spin for the result instead of sleeping
    gfo_ctx* pick_bc = nullptr;
    const GfoSmallLayout* pick_L = nullptr;
    int pick_image = 0, picked = -1;
    uint8_t* h_pair = nullptr;   // pinned: left image | right image, tight rows -- each side copies its own half, ONE copy takes both down
    int w = 0, h = 0;
    // the frame delivered last: what the two calls returned (for the identity check) and its association
    bool valid = false;
    int nl = 0, nr = 0, nm = 0;
    std::vector<gfo_keypoint> rk[2];
    std::vector<uint8_t> rd[2];
    std::vector<float> ur, dp;
    std::vector<int32_t> bd, bi;
    long speculated = 0, served = 0, solo = 0;
    int solo_streak = 0;      // frames in a row whose partner did not show up: after three the rig sleeps (no more waiting) until it is
    bool dormant = false;     // declared again -- which the adapter does at every ComputeStereoMatches, i.e. as soon as stereo frames are back.
    int wake_need = 1, wake_target = 1, wake_count = 0;   // A caller that extracts left and right on ONE thread declares the rig every frame
                              // and never meets its partner: every time the rig falls asleep again without a frame in between that
                              // did pair, it takes twice as many declarations to wake it (1, 2, 4 ... 1024), so the 2-ms waits die out
    ~GfoPair()
    {
        if (h_pair) {
            gfo_note_pinned(h_pair, 0, false);
            (void)hipHostFree(h_pair);
        }
    }
};

// A context's rig (gfo_ctx::pair) may be declared, re-declared or dissolved by one thread (gfo_ctx_pair, gfo_ctx_destroy of the
// partner) while another is entering gfo_extract on that context: the member is read and written through the atomic shared_ptr
// functions only, and which side a context is on is asked of the rig itself, under its lock (round 5: tests/host/combine_tsan.cc
// found the plain accesses racing).
static inline std::shared_ptr<GfoPair> pair_of(const gfo_ctx* c) { return std::atomic_load(&c->pair); }
static inline void set_pair(gfo_ctx* c, const std::shared_ptr<GfoPair>& P) { std::atomic_store(&c->pair, P); }
bool gfo_has_pair(const gfo_ctx* c) { return pair_of(c) != nullptr; }

namespace
{
typedef std::tuple<int, int, int, int, float, int, int, int> EngineKey;
std::mutex g_eng_mu;
std::map<EngineKey, std::weak_ptr<GfoEngine>> g_engines;

std::shared_ptr<GfoEngine> engine_for(gfo_ctx* c, int w, int h)
{
    if (c->engine && c->engine->w == w && c->engine->h == h) return c->engine;
    std::lock_guard<std::mutex> lk(g_eng_mu);
    const EngineKey key(c->device, w, h, c->prm.nfeatures, c->prm.scale_factor, c->prm.nlevels, c->prm.ini_th_fast, c->prm.min_th_fast);
    std::shared_ptr<GfoEngine> e = g_engines[key].lock();
    if (!e) {
        e = std::make_shared<GfoEngine>();
        e->device = c->device; e->w = w; e->h = h; e->prm = c->prm;
        if (const char* m = getenv("GFO_COMBINE_MAX")) {
            const int v = atoi(m);
            if (v >= 2 && v <= 256) e->cap_images = v & ~1;
        }
        if (const char* m = getenv("GFO_COMBINE_INFLIGHT")) {
            const int v = atoi(m);
            if (v >= 1 && v <= NSLOT - 4) e->inflight = v;
        }
        e->nslot = e->inflight + 4 <= NSLOT ? e->inflight + 4 : NSLOT;
        if (const char* m = getenv("GFO_COMBINE_FAIL_PREPARE")) e->fail_prepare_from = atoi(m);
        if (const char* m = getenv("GFO_COMBINE_FAIL_BATCH")) e->fail_batch_every = atoi(m);
        g_engines[key] = e;
    }
    {
        std::lock_guard<std::mutex> lk2(e->mu);
        e->members++;
    }
    c->engine = e;
    return e;
}

// first use of a slot: its batch context, arena and pinned buffers (the caller holds e->mu; happens NSLOT times per engine)
int slot_prepare(GfoEngine* e, Slot& s, gfo_ctx* c)
{
    if (s.bc) return GFO_OK;
    // tests only (tests/test_gpu_combine.py): GFO_COMBINE_FAIL_PREPARE=k makes the k-th and every later slot preparation of an
    // engine fail as an out-of-memory would (k = 0: the engine never gets a slot and its callers take the direct path)
    if (e->fail_prepare_from >= 0 && e->prepared >= e->fail_prepare_from) return gfo_fail(c, GFO_ERR_DEVICE, "combiner: slot preparation failed (injected)");
    gfo_params p = e->prm;
    p.max_batch = e->cap_images;
    int rc = gfo_ctx_create(&p, e->device, &s.bc);
    if (rc) return gfo_fail(c, rc, "combiner: %s", gfo_last_error(nullptr));
    rc = gfo_plan(s.bc, e->w, e->h, e->cap_images);
    if (!rc) rc = gfo_small_prepare(s.bc, e->cap_images, &s.L);
    if (!rc) {
        // host-array stereo batches stage one block per pair in the image staging (idle then) and on the device
        const GfoPairBlock b = gfo_pair_block(s.bc->g.kp_stride);
        const size_t need = b.bytes * (size_t)(e->cap_images / 2);
        if (need <= s.bc->h_in_bytes && hipMalloc(&s.d_pairs, need) != hipSuccess) {
            (void)hipGetLastError();
            s.d_pairs = nullptr;      // those calls then take the direct path
        }
    }
    if (rc) {
        gfo_fail(c, rc, "combiner: %s", gfo_last_error(s.bc));
        gfo_ctx_destroy(s.bc);
        s.bc = nullptr;
    }
    return rc;
}
}  // namespace

void gfo_engine_release(gfo_ctx* c)
{
    if (std::shared_ptr<GfoEngine> e = c->engine) {
        std::lock_guard<std::mutex> lk(e->mu);
        if (e->members > 0) e->members--;      // (a pair of extractors that is deleted and re-created -- Tracking::updateORBExtractor -- stays "one camera")
    }
    c->engine.reset();
}

extern "C" int gfo_ctx_set_combining(gfo_ctx* c, int on)
{
    if (!c) return GFO_ERR_INVALID;
    c->combining = on != 0;
    if (!on) gfo_engine_release(c);
    return GFO_OK;
}

extern "C" int gfo_combiner_stats(const gfo_ctx* c, int64_t* batches, int64_t* requests)
{
    if (!c) return GFO_ERR_INVALID;
    std::shared_ptr<GfoEngine> e = c->engine;
    if (batches) *batches = 0;
    if (requests) *requests = 0;
    if (!e) return GFO_OK;
    std::lock_guard<std::mutex> lk(e->mu);
    if (batches) *batches = e->batches;
    if (requests) *requests = e->requests;
    return GFO_OK;
}

extern "C" int gfo_combiner_counters(const gfo_ctx* c, int64_t* out, int n)
{
    if (!c || !out || n < 1) return GFO_ERR_INVALID;
    int64_t v[GFO_COMBINER_COUNTERS] = {0};
    if (std::shared_ptr<GfoEngine> e = c->engine) {
        std::lock_guard<std::mutex> lk(e->mu);
        v[0] = e->batches; v[1] = e->requests; v[2] = e->redone; v[3] = e->prepared; v[4] = e->broken ? 1 : 0;
    }
    if (std::shared_ptr<GfoPair> P = pair_of(c)) {
        std::lock_guard<std::mutex> lk(P->mu);
        v[5] = P->speculated; v[6] = P->served; v[7] = P->solo;
    }
    for (int i = 0; i < n && i < GFO_COMBINER_COUNTERS; i++) out[i] = v[i];
    return GFO_OK;
}

// One request of one caller through the engine's slots.  kind 1: one image (gfo_extract); 2: a stereo frame
// (gfo_extract_stereo); 3: the host-array stereo association of one pair (gfo_stereo_match -- the third call of the adapter's
// pattern).  `units` = images of the batch capacity a request takes.  upload(slot, idx): the caller's own inputs on their
// way to the device, on the slot's stream; submit(slot, nb): the leader runs the batch of nb requests and synchronises;
// collect(slot, idx): the caller's own results out of the slot's pinned buffer (returns 1 for "truncated").
template <class Upload, class Submit, class Collect>
static int run_request(gfo_ctx* c, GfoEngine* e, int kind, int units, const gfo_stereo_params* sp, Upload upload, Submit submit, Collect collect,
                       int* truncated)
{
    std::unique_lock<std::mutex> lk(e->mu);
    int si = -1, idx = 0;
    bool leader = false;
    for (;;) {
        if (e->forming[kind] >= 0) {
            Slot& f = e->slot[e->forming[kind]];
            if (f.state == SLOT_FORMING && (f.n + 1) * units <= e->cap_images && (kind == 1 || memcmp(&f.sp, sp, sizeof *sp) == 0)) {
                si = e->forming[kind];
                idx = f.n++;
                break;
            }
        } else {
            // Slots are prepared as concurrency shows: a slot is an arena for cap_images images plus pinned buffers --
            // ~130 MB at 752x480, several hundred MB at 1080p.  An engine that only ever served one or two contexts (one camera:
            // the left and right extractor, or one monocular extractor) gets two -- one batch collected while the next forms -- and
            // one more whenever a caller finds every prepared slot busy.  As soon as a third context has attached (several cameras:
            // the case the combiner exists for) the next request prepares all nslot, so that nothing is created or planned once the
            // streams run (gfo_contexts_created / gfo_arenas_planned stay put in steady state).  A slot that cannot be prepared
            // (out of memory) is not fatal: with another slot there the caller waits for it, with none the engine is marked broken
            // and every caller takes the direct path.
            if (e->broken) return GFO_COMBINE_DIRECT;
            const bool want_all = e->members > 2 && e->prepared < e->nslot && !e->grow_failed;
            for (int i = 0; i < e->prepared && si < 0 && !want_all; i++)
                if (e->slot[i].state == SLOT_FREE) si = i;
            if (si < 0 && e->prepared < e->nslot && !e->grow_failed) {
                const int want = want_all ? e->nslot : (e->prepared == 0 ? (e->nslot < 2 ? e->nslot : 2) : e->prepared + 1);
                while (e->prepared < want) {
                    if (slot_prepare(e, e->slot[e->prepared], c)) {
                        e->grow_failed = true;
                        break;
                    }
                    e->prepared++;
                }
                if (e->prepared == 0) {
                    e->broken = true;
                    e->cv.notify_all();
                    return GFO_COMBINE_DIRECT;
                }
                for (int i = 0; i < e->prepared && si < 0; i++)
                    if (e->slot[i].state == SLOT_FREE) si = i;
            }
            if (si >= 0) {
                Slot& f = e->slot[si];
                if (sp && (sp->n_rows < 1 || sp->n_rows > f.bc->st_rows_cap))
                    return gfo_fail(c, GFO_ERR_INVALID, "n_rows %d exceeds the planned %d", sp->n_rows, f.bc->st_rows_cap);
                f.state = SLOT_FORMING; f.kind = kind; f.n = 1; f.staged = 0; f.readers = 0; f.rc = 0;
                if (sp) f.sp = *sp;
                e->forming[kind] = si;
                idx = 0;
                leader = true;
                break;
            }
        }
        e->cv.wait(lk);   // the forming batch of this kind is full or of another calibration, or every slot is busy
    }
    Slot& s = e->slot[si];
    const unsigned my_gen = s.gen;
    e->requests++;
    lk.unlock();

    // every caller stages its own frame, in parallel, and sends it on its way at once: the copy is queued on the SLOT's
    // stream (idle since the slot's previous batch was collected), so it runs while later joiners are still copying and
    // while the leader waits for room on the device; the batch's kernels are queued behind all of them.  ONE copy per
    // request: splitting a stereo frame into two copies (left image on the link while the right one is staged) measured
    // 13-17 % slower at K = 8 / 16 -- a copy costs the DMA engine ~10 us whatever its size, and the copies serialise
    int crc = hipSetDevice(e->device) == hipSuccess ? GFO_OK : GFO_ERR_DEVICE;
    if (!crc) crc = upload(s, idx) ? GFO_ERR_DEVICE : GFO_OK;

    lk.lock();
    s.staged++;
    if (crc && !s.rc) { s.rc = crc; s.err = "host-to-device copy of a request failed"; }
    if (leader) {
        while (e->running >= e->inflight) e->cv.wait(lk);   // joiners keep arriving while the device is busy with earlier batches
        s.state = SLOT_CLOSED;
        e->forming[kind] = -1;
        const int nb = s.n;
        e->cv.notify_all();                              // whoever could not join may open the next batch
        while (s.staged < nb) s.cv.wait(lk);
        s.state = SLOT_RUNNING;
        e->running++;
        e->batches++;
        int rc = s.rc;
        const bool inject = e->fail_batch_every > 0 && nb > 1 && e->batches % e->fail_batch_every == 0;   // tests only
        lk.unlock();
        if (!rc) rc = submit(s, nb);
        else (void)hipStreamSynchronize(s.bc->stream);
        if (!rc && inject) rc = gfo_fail(s.bc, GFO_ERR_OVERFLOW, "batch failure (injected)");
        lk.lock();
        e->running--;
        s.rc = rc;
        if (rc && s.err.empty()) s.err = gfo_last_error(s.bc);
        s.readers = nb;
        s.state = SLOT_DRAINING;
        s.cv.notify_all();
        e->cv.notify_all();                              // room on the device
    } else {
        if (s.state == SLOT_CLOSED && s.staged == s.n) s.cv.notify_all();   // the leader is waiting for this copy
        while (!(s.gen == my_gen && s.state == SLOT_DRAINING)) s.cv.wait(lk);
    }
    int brc = s.rc;
    if (brc && s.n > 1) {
        // a batch-level failure is not every member's failure: one frame that trips the shared overflow flags, or
        // one caller's failed upload, must not fail the other callers' frames -- "a combined call returns what the direct
        // call returns".  Every member of a failed batch of several re-runs its own request alone on the direct path; only the
        // frame that really fails reports the error, from its own context.
        brc = GFO_COMBINE_DIRECT;
        e->redone++;
    } else if (brc) {
        gfo_fail(c, brc, "combined batch: %s", s.err.c_str());
    }
    lk.unlock();

    *truncated = 0;
    if (!brc) *truncated = collect(s, idx);

    lk.lock();
    if (--s.readers == 0) {
        s.state = SLOT_FREE;
        s.n = s.staged = 0;
        s.err.clear();
        s.gen++;
        e->cv.notify_all();
    }
    lk.unlock();
    return brc;
}

int gfo_combined_extract(gfo_ctx* c, int kind, const uint8_t* const* imgs, int w, int h, int stride, const gfo_stereo_params* sp,
                         gfo_keypoint* const* kp, uint8_t* const* desc, int cap, int* n, float* u_right, float* depth,
                         int32_t* best_dist, int32_t* best_idx_r, int* nmatched)
{
    std::shared_ptr<GfoEngine> eh = engine_for(c, w, h);
    int over = 0;
    bool one_camera;   // this context is all the engine serves (a mono camera): see gfo_small_upload
    {
        std::lock_guard<std::mutex> g(eh->mu);
        one_camera = eh->members <= 1;
    }
    const int rc = run_request(
        c, eh.get(), kind, kind, sp,
        [&](Slot& s, int idx) { return gfo_small_upload(s.bc, c, s.L, idx * kind, kind, imgs, w, h, stride, s.bc->stream, one_camera); },
        [&](Slot& s, int nb) { return gfo_small_submit(s.bc, s.L, nb * kind, kind == 2 ? &s.sp : nullptr, false); },
        [&](Slot& s, int idx) {
            int o = 0;
            for (int k = 0; k < kind; k++) o |= gfo_small_collect(s.bc, s.L, idx * kind + k, kp[k], desc[k], cap, &n[k]);
            if (kind == 2) gfo_small_collect_stereo(s.bc, s.L, idx, n[0], cap, u_right, depth, best_dist, best_idx_r, nmatched);
            return o;
        },
        &over);
    if (rc) return rc;      // (GFO_COMBINE_DIRECT: the caller takes the direct path)
    return over ? gfo_fail(c, GFO_ERR_CAPACITY, "an image produced more keypoints than the caller capacity %d", cap) : GFO_OK;
}

// gfo_stereo_match of a combining context: the pairs that several threads associate at the same time -- the third call
// of every Frame constructor in the adapter's pattern -- as ONE launch of the three association kernels.  Returns 1 when
// the request cannot go through the engine (no engine yet, arrays longer than its arena's stride, other scale factors):
// the caller then takes the direct path.
int gfo_combined_stereo_match(gfo_ctx* c, const gfo_keypoint* kl, const uint8_t* dl, int nl, const gfo_keypoint* kr, const uint8_t* dr, int nr,
                              const float* sf, int nlevels, const gfo_stereo_params* p, const float* min_d, const float* max_d,
                              float* u_right, float* depth, int32_t* best_dist, int32_t* best_idx_r, int* nmatched, int* status)
{
    std::shared_ptr<GfoEngine> eh = c->engine;      // the engine of this extractor's frames (set by its last gfo_extract)
    if (!eh || nlevels != c->prm.nlevels || memcmp(sf, c->scale.data(), sizeof(float) * nlevels) != 0) return 1;
    GfoEngine* e = eh.get();
    int ks = 0, rows_cap = 0;
    {
        std::lock_guard<std::mutex> lk(e->mu);
        if (e->broken || !e->slot[0].bc || !e->slot[0].d_pairs) return 1;
        ks = e->slot[0].bc->g.kp_stride;
        rows_cap = e->slot[0].bc->st_rows_cap;
    }
    if (nl > ks || nr > ks || p->n_rows > rows_cap) return 1;
    const GfoPairBlock b = gfo_pair_block(ks);
    const float maxD0 = p->mbf / p->mb;
    int trunc = 0;
    *status = run_request(
        c, e, 3, 2, p,
        [&](Slot& s, int idx) {
            uint8_t* H = s.bc->h_in + (size_t)idx * b.bytes;      // the image staging, idle in a batch of this kind
            int hdr[4] = {nl, nr, 0, 0};
            memcpy(H, hdr, 16);
            memcpy(H + b.o_kl, kl, sizeof(gfo_keypoint) * (size_t)nl);
            memcpy(H + b.o_dl, dl, 32 * (size_t)nl);
            if (nr > 0) {
                memcpy(H + b.o_kr, kr, sizeof(gfo_keypoint) * (size_t)nr);
                memcpy(H + b.o_dr, dr, 32 * (size_t)nr);
            }
            float* wmin = reinterpret_cast<float*>(H + b.o_min);
            float* wmax = reinterpret_cast<float*>(H + b.o_max);
            if (min_d && max_d) {
                memcpy(wmin, min_d, 4 * (size_t)nl);
                memcpy(wmax, max_d, 4 * (size_t)nl);
            } else {
                for (int i = 0; i < nl; i++) { wmin[i] = 0.f; wmax[i] = maxD0; }   // Frame.cc:1199-1200: the whole disparity range
            }
            // one copy: header .. end of the right descriptors are contiguous up to what this pair uses; the windows follow
            // at fixed offsets, so the block goes as a whole
            return hipMemcpyAsync(s.d_pairs + (size_t)idx * b.bytes, H, b.bytes, hipMemcpyHostToDevice, s.bc->stream) == hipSuccess ? 0 : 1;
        },
        [&](Slot& s, int nb) { return gfo_small_submit_pairs(s.bc, s.L, nb, &s.sp, s.d_pairs); },
        [&](Slot& s, int idx) {
            gfo_small_collect_stereo(s.bc, s.L, idx, nl, nl, u_right, depth, best_dist, best_idx_r, nmatched);
            return 0;
        },
        &trunc);
    return *status == GFO_COMBINE_DIRECT ? 1 : 0;      // the engine could not serve it (or its batch failed as a whole): direct path
}

// ---------------------------------------------------------------------------------------------------------------------
// stereo rigs: gfo_ctx_pair
// ---------------------------------------------------------------------------------------------------------------------
extern "C" int gfo_ctx_pair(gfo_ctx* left, gfo_ctx* right, const gfo_stereo_params* p)
{
    if (!left) return GFO_ERR_INVALID;
    if (!right || !p) {      // dissolve
        gfo_pair_release(left);
        return GFO_OK;
    }
    if (left == right) return gfo_fail(left, GFO_ERR_INVALID, "a context cannot be its own stereo partner");
    if (left->device != right->device || memcmp(&left->prm, &right->prm, offsetof(gfo_params, max_batch)) != 0)
        return gfo_fail(left, GFO_ERR_INVALID, "stereo partners need equal extractor parameters and one device");
    if (std::shared_ptr<GfoPair> P = pair_of(left)) {      // the usual call: once per frame, nothing changes
        std::lock_guard<std::mutex> lk(P->mu);
        if (P == pair_of(right) && P->ctx[0] == left && P->ctx[1] == right) {
            if (memcmp(&P->sp, p, sizeof *p) != 0) {
                P->sp = *p;
                P->valid = false;
            }
            if (P->dormant && ++P->wake_count >= P->wake_target) {
                P->dormant = false;
                P->solo_streak = 0;
            }
            return GFO_OK;
        }
    }
    gfo_pair_release(left);
    gfo_pair_release(right);
    std::shared_ptr<GfoPair> P = std::make_shared<GfoPair>();
    P->ctx[0] = left; P->ctx[1] = right; P->sp = *p;
    set_pair(left, P);
    set_pair(right, P);
    return GFO_OK;
}

void gfo_pair_release(gfo_ctx* c)
{
    std::shared_ptr<GfoPair> P = pair_of(c);
    if (!P) return;
    {
        std::unique_lock<std::mutex> lk(P->mu);
        while (P->state >= 2) P->cv.wait(lk);      // a frame is being executed for both sides: let it finish
        for (int s = 0; s < 2; s++)
            if (P->ctx[s] == c) P->ctx[s] = nullptr;
        P->valid = false;
        P->cv.notify_all();
    }
    set_pair(c, std::shared_ptr<GfoPair>());
}

// gfo_extract of a paired, combining context.  Returns GFO_COMBINE_DIRECT when the frame is not taken here (no partner, the
// partner does not show up, other geometry): the caller goes on as if there were no pair.
// the rig's two timing parameters (include/gfo.h gfo_tuning_set): the environment gives the initial values
static std::atomic<long> g_pair_wait_us{getenv("GFO_PAIR_WAIT_US") ? atol(getenv("GFO_PAIR_WAIT_US")) : 2000};
static std::atomic<long> g_pair_spin_us{getenv("GFO_PAIR_SPIN_US") ? atol(getenv("GFO_PAIR_SPIN_US")) : 400};
extern "C" int gfo_tuning_set(const char* key, long value)
{
    if (!key || value < 0) return GFO_ERR_INVALID;
    if (strcmp(key, "pair_wait_us") == 0) { g_pair_wait_us.store(value); return GFO_OK; }
    if (strcmp(key, "pair_spin_us") == 0) { g_pair_spin_us.store(value); return GFO_OK; }
    return GFO_ERR_INVALID;
}
extern "C" long gfo_tuning_get(const char* key)
{
    if (!key) return -1;
    if (strcmp(key, "pair_wait_us") == 0) return g_pair_wait_us.load();
    if (strcmp(key, "pair_spin_us") == 0) return g_pair_spin_us.load();
    return -1;
}

int gfo_pair_extract(gfo_ctx* c, const uint8_t* img, int w, int h, int stride, gfo_keypoint* kp, uint8_t* desc, int cap, int* n)
{
    std::shared_ptr<GfoPair> P = pair_of(c);
    if (!P || (w & 15) != 0) return GFO_COMBINE_DIRECT;      // (tight rows of the staged pair must stay 16-byte aligned)
    const long wait_us = g_pair_wait_us.load(std::memory_order_relaxed);
    if (wait_us <= 0) return GFO_COMBINE_DIRECT;
    std::unique_lock<std::mutex> lk(P->mu);
    const int side = P->ctx[1] == c ? 1 : 0;
    if (P->ctx[side] != c) return GFO_COMBINE_DIRECT;       // this rig was dissolved while the call was on its way in
    GfoPair::Req& me = P->req[side];
    GfoPair::Req& other = P->req[side ^ 1];
    gfo_ctx* partner = P->ctx[side ^ 1];
    if (!partner || !partner->combining || me.arrived || P->state >= 2 || P->dormant) return GFO_COMBINE_DIRECT;
    if (P->sp.n_rows < 1 || P->sp.n_rows > h + 64) return GFO_COMBINE_DIRECT;      // an association no arena of this size is planned for: no speculation
    if (other.arrived && (other.w != w || other.h != h)) return GFO_COMBINE_DIRECT;
    const size_t img_bytes = (size_t)w * h;
    if (!P->h_pair || P->w != w || P->h != h) {
        if (other.arrived) return GFO_COMBINE_DIRECT;
        if (P->h_pair) {
            gfo_note_pinned(P->h_pair, 0, false);
            (void)hipHostFree(P->h_pair);
            P->h_pair = nullptr;
        }
        if (hipSetDevice(c->device) != hipSuccess || hipHostMalloc((void**)&P->h_pair, 2 * img_bytes, hipHostMallocDefault) != hipSuccess) {
            (void)hipGetLastError();
            P->h_pair = nullptr;
            return GFO_COMBINE_DIRECT;
        }
        gfo_note_pinned(P->h_pair, 2 * img_bytes, true);
        P->w = w; P->h = h;
    }
    me.img = img; me.w = w; me.h = h; me.kp = kp; me.desc = desc; me.cap = cap; me.n = n; me.rc = 0;
    me.arrived = true; me.staged = false;
    const bool first = !other.arrived;
    P->state = first ? 1 : 2;            // the second arrival executes the frame for both
    P->state_seen.store(P->state, std::memory_order_release);
    P->cv.notify_all();
    lk.unlock();

    // each side stages its own image, in parallel with the other
    uint8_t* dst = P->h_pair + (size_t)side * img_bytes;
    if (stride == w) memcpy(dst, img, img_bytes);
    else
        for (int y = 0; y < h; y++) memcpy(dst + (size_t)y * w, img + (size_t)y * stride, w);

    // a side's own results out of the batch's pinned result block, and its half of what gfo_stereo_match will compare against
    auto take_mine = [&](gfo_ctx* bc, const GfoSmallLayout& L, int image, int* cnt_out) {
        const int over = gfo_small_collect(bc, L, image, kp, desc, cap, cnt_out);
        const bool keep = !over && kp && desc;
        if (keep) {
            P->rk[side].assign(kp, kp + *cnt_out);
            P->rd[side].assign(desc, desc + 32 * (size_t)*cnt_out);
        }
        return over ? 2 : (keep ? 1 : 0);      // 2 truncated, 1 kept, 0 delivered but not kept (no output arrays)
    };
    lk.lock();
    me.staged = true;
    P->cv.notify_all();
    if (first) {
        const auto deadline = std::chrono::steady_clock::now() + std::chrono::microseconds(wait_us);
        while (!other.arrived) {
            if (P->cv.wait_until(lk, deadline) == std::cv_status::timeout && !other.arrived) {
                // nobody came for the other image: this frame is not a stereo frame after all
                me.arrived = me.staged = false;
                P->state = 0;
                P->valid = false;
                P->solo++;
                if (++P->solo_streak >= 3 && !P->dormant) {      // this extractor is being used on its own: stop waiting for a partner
                    P->dormant = true;
                    P->wake_count = 0;
                    P->wake_target = P->wake_need;
                    P->wake_need = P->wake_need < 1024 ? 2 * P->wake_need : 1024;
                }
                return GFO_COMBINE_DIRECT;
            }
        }
        // the partner executes the frame; it calls this side to the result block as soon as the batch is complete (state 4), so
        // that both sides copy their own arrays at the same time; a failed batch skips that (state 3).
        // One camera alone (the engine serves nobody else): this thread has nothing else to do for the ~0.17 ms the frame takes --
        // in the reference it would be extracting the left image itself -- and a sleeping thread's wake-up costs 5-15 us of the
        // frame's latency: it watches the state for up to GFO_PAIR_SPIN_US (400) before it goes to sleep on the condition
        // variable.  With several cameras the waiting sides sleep at once (their cores belong to the other cameras' threads).
        if (P->spin_ok) {
            const long spin_us = g_pair_spin_us.load(std::memory_order_relaxed);
            lk.unlock();
            const auto until = std::chrono::steady_clock::now() + std::chrono::microseconds(spin_us);
            for (int it = 0;; it++) {
                const int st = P->state_seen.load(std::memory_order_acquire);
                if (st == 3 || st == 4) break;
                GFO_CPU_RELAX();
                if ((it & 63) == 63 && std::chrono::steady_clock::now() >= until) break;
            }
            lk.lock();
        }
        while (P->state != 3 && P->state != 4) P->cv.wait(lk);
        if (P->state == 4) {
            gfo_ctx* bc = P->pick_bc;
            const GfoSmallLayout L = *P->pick_L;
            const int image = P->pick_image + side;
            lk.unlock();
            int cnt = 0;
            const int got = take_mine(bc, L, image, &cnt);
            lk.lock();
            *n = cnt;
            P->picked = got;
            P->cv.notify_all();
            return got == 2 ? gfo_fail(c, GFO_ERR_CAPACITY, "an image produced more keypoints than the caller capacity %d", cap) : GFO_OK;
        }
        const int rc = me.rc;
        me.arrived = me.staged = false;
        other.arrived = other.staged = false;
        P->state = 0;
        P->cv.notify_all();
        return rc;
    }
    // executor: both images are (about to be) staged; ONE stereo request for the frame
    while (!other.staged) P->cv.wait(lk);
    const gfo_stereo_params sp = P->sp;
    P->valid = false;      // the stored frame is about to be overwritten (both sides write their halves without the lock): a
                           // gfo_stereo_match of another thread that is still asking about the previous frame computes its answer
    lk.unlock();
    std::shared_ptr<GfoEngine> eh = engine_for(c, w, h);
    bool alone_on_engine;
    {
        std::lock_guard<std::mutex> g(eh->mu);
        alone_on_engine = eh->members <= 2;
    }
    {
        std::lock_guard<std::mutex> g(P->mu);
        P->spin_ok = alone_on_engine;
    }
    const uint8_t* imgs[2] = {P->h_pair, P->h_pair + img_bytes};
    int cnt_mine = 0, nm = 0, got_mine = 0, got_other = 0, cnt_other = 0, dummy = 0;
    const int rc = run_request(
        c, eh.get(), 2, 2, &sp,
        [&](Slot& s, int idx) { return gfo_small_upload(s.bc, c, s.L, idx * 2, 2, imgs, w, h, w, s.bc->stream); },
        [&](Slot& s, int nb) { return gfo_small_submit(s.bc, s.L, nb * 2, &s.sp, false); },
        [&](Slot& s, int idx) {
            {   // call the waiting side to the result block
                std::lock_guard<std::mutex> g(P->mu);
                P->pick_bc = s.bc; P->pick_L = &s.L; P->pick_image = idx * 2; P->picked = -1;
                P->state = 4;
                P->state_seen.store(4, std::memory_order_release);
                P->cv.notify_all();
            }
            got_mine = take_mine(s.bc, s.L, idx * 2 + side, &cnt_mine);
            const int ks = s.bc->g.kp_stride;
            P->ur.resize(ks); P->dp.resize(ks); P->bd.resize(ks); P->bi.resize(ks);
            const int nl = reinterpret_cast<const int*>(s.bc->h_out + s.L.o_cnt)[idx * 2];
            gfo_small_collect_stereo(s.bc, s.L, idx, nl, ks, P->ur.data(), P->dp.data(), P->bd.data(), P->bi.data(), &nm);
            cnt_other = reinterpret_cast<const int*>(s.bc->h_out + s.L.o_cnt)[idx * 2 + (side ^ 1)];
            std::unique_lock<std::mutex> g(P->mu);
            while (P->picked < 0) P->cv.wait(g);       // the slot's block stays ours until the other side has its arrays
            got_other = P->picked;
            return 0;
        },
        &dummy);
    lk.lock();
    int my_rc = rc;
    if (rc == GFO_OK) {
        *n = cnt_mine;
        if (got_mine == 2) my_rc = gfo_fail(c, GFO_ERR_CAPACITY, "an image produced more keypoints than the caller capacity %d", cap);
        P->nl = side == 0 ? cnt_mine : cnt_other;
        P->nr = side == 0 ? cnt_other : cnt_mine;
        P->nm = nm;
        P->valid = got_mine == 1 && got_other == 1;
        P->speculated++;
        P->solo_streak = 0;
        P->wake_need = 1;
        me.arrived = me.staged = false;
        other.arrived = other.staged = false;
        P->state = 0;
    } else {
        // the batch failed (or the engine could not take it: GFO_COMBINE_DIRECT, both sides then go on alone)
        other.rc = rc;
        if (rc != GFO_COMBINE_DIRECT && P->ctx[side ^ 1]) gfo_fail(P->ctx[side ^ 1], rc, "%s", gfo_last_error(c));
        P->valid = false;
        me.arrived = me.staged = false;
        P->state = 3;          // the waiter takes its status and resets the rendezvous
        P->state_seen.store(3, std::memory_order_release);
    }
    P->cv.notify_all();
    return my_rc;
}

// gfo_stereo_match on the left context of a pair: if the arrays are, bit for bit, what the pair's last two gfo_extract calls
// returned, the calibration is the declared one and no disparity windows are given, the association computed with that frame
// IS the answer (same kernels, same inputs).  Returns 0 when served, 1 otherwise (the caller computes it).
int gfo_pair_lookup(gfo_ctx* c, const gfo_keypoint* kl, const uint8_t* dl, int nl, const gfo_keypoint* kr, const uint8_t* dr, int nr,
                    const float* sf, int nlevels, const gfo_stereo_params* p, const float* min_d, const float* max_d,
                    float* u_right, float* depth, int32_t* best_dist, int32_t* best_idx_r, int* nmatched)
{
    std::shared_ptr<GfoPair> P = pair_of(c);
    if (!P || min_d || max_d) return 1;
    if (nlevels != c->prm.nlevels || memcmp(sf, c->scale.data(), sizeof(float) * nlevels) != 0) return 1;
    std::lock_guard<std::mutex> lk(P->mu);
    if (P->ctx[0] != c) return 1;                            // only the LEFT context of a rig asks for its association
    if (!P->valid || P->state != 0 || nl != P->nl || nr != P->nr || memcmp(p, &P->sp, sizeof *p) != 0) return 1;
    if ((int)P->rk[0].size() != nl || (int)P->rk[1].size() != nr) return 1;
    if (memcmp(kl, P->rk[0].data(), sizeof(gfo_keypoint) * (size_t)nl) != 0 || memcmp(dl, P->rd[0].data(), 32 * (size_t)nl) != 0) return 1;
    if (nr > 0 && (memcmp(kr, P->rk[1].data(), sizeof(gfo_keypoint) * (size_t)nr) != 0 || memcmp(dr, P->rd[1].data(), 32 * (size_t)nr) != 0)) return 1;
    memcpy(u_right, P->ur.data(), 4 * (size_t)nl);
    memcpy(depth, P->dp.data(), 4 * (size_t)nl);
    if (best_dist) memcpy(best_dist, P->bd.data(), 4 * (size_t)nl);
    if (best_idx_r) memcpy(best_idx_r, P->bi.data(), 4 * (size_t)nl);
    *nmatched = P->nm;
    P->served++;
    return 0;
}
