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

#include <condition_variable>
#include <map>
#include <memory>
#include <mutex>
#include <string.h>
#include <tuple>

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
    long batches = 0, requests = 0;
    ~GfoEngine()
    {
        for (Slot& s : slot) {
            if (s.d_pairs) (void)hipFree(s.d_pairs);
            if (s.bc) gfo_ctx_destroy(s.bc);
        }
    }
};

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
        g_engines[key] = e;
    }
    c->engine = e;
    return e;
}

// first use of a slot: its batch context, arena and pinned buffers (the caller holds e->mu; happens NSLOT times per engine)
int slot_prepare(GfoEngine* e, Slot& s, gfo_ctx* c)
{
    if (s.bc) return GFO_OK;
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

void gfo_engine_release(gfo_ctx* c) { c->engine.reset(); }

extern "C" int gfo_ctx_set_combining(gfo_ctx* c, int on)
{
    if (!c) return GFO_ERR_INVALID;
    c->combining = on != 0;
    if (!on) c->engine.reset();
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
            // every slot the engine will ever use is prepared by its first request: nothing is created or planned once the
            // streams are running (gfo_contexts_created / gfo_arenas_planned stay put in steady state)
            for (int i = 0; i < e->nslot; i++) {
                const int rc = slot_prepare(e, e->slot[i], c);
                if (rc) return rc;
            }
            for (int i = 0; i < e->nslot && si < 0; i++)
                if (e->slot[i].state == SLOT_FREE) si = i;
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
        lk.unlock();
        if (!rc) rc = submit(s, nb);
        else (void)hipStreamSynchronize(s.bc->stream);
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
    const int brc = s.rc;
    if (brc) gfo_fail(c, brc, "combined batch: %s", s.err.c_str());
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
    const int rc = run_request(
        c, eh.get(), kind, kind, sp,
        [&](Slot& s, int idx) { return gfo_small_upload(s.bc, c, s.L, idx * kind, kind, imgs, w, h, stride, s.bc->stream); },
        [&](Slot& s, int nb) { return gfo_small_submit(s.bc, s.L, nb * kind, kind == 2 ? &s.sp : nullptr, false); },
        [&](Slot& s, int idx) {
            int o = 0;
            for (int k = 0; k < kind; k++) o |= gfo_small_collect(s.bc, s.L, idx * kind + k, kp[k], desc[k], cap, &n[k]);
            if (kind == 2) gfo_small_collect_stereo(s.bc, s.L, idx, n[0], cap, u_right, depth, best_dist, best_idx_r, nmatched);
            return o;
        },
        &over);
    if (rc) return rc;
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
        if (!e->slot[0].bc || !e->slot[0].d_pairs) return 1;
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
    return 0;
}
