/* boundary_throughput.c -- throughput and latency of the reference's OWN call pattern through the C ABI: one stereo
 * frame per call (Frame::Frame, Frame.cc:84-100), K independent camera streams on K host threads, each with its own
 * context(s).  No Python, no batching the reference's callers could not issue.
 *
 *   stereo  : one context per stream, gfo_extract_stereo per frame (the stereo Frame constructor body in one submission)
 *   adapter : two contexts per stream (the reference's left / right ORBextractor objects), the right image extracted on
 *             a thread created per frame (`thread threadRight(&Frame::ExtractORB, ...)`, Frame.cc:84-87), then
 *             gfo_stereo_match on host arrays (Frame::ComputeStereoMatches_Undistorted, Frame.cc:100) -- what
 *             adapter/ORBextractor_gfo.cc + matchers_gfo.cc do by default
 *
 * Every frame's results are checksummed and compared with the first frame of stream 0 (same input pair everywhere):
 * concurrency must not change a bit.  Contexts created / arenas planned are read before and after the timed region:
 * steady state must create nothing.
 *
 * Build: gcc -O2 -I include tools/c/boundary_throughput.c -o /tmp/boundary_throughput -ldl -lpthread -lm
 * Usage: boundary_throughput <libgfo.so> <golden dir> <seconds per point> <stereo|adapter|both> <K list, e.g. 1,2,4,8,16> [combine 0|1] [pin 0|1] [pair 0|1]
 * pair = 1 (default): in adapter mode the two contexts of a stream are declared a stereo rig (gfo_ctx_pair) before every association,
 * as adapter/matchers_gfo.cc does.
 * combine = 1 (default): every context opts into the frame combiner (gfo_ctx_set_combining), as the drop-in adapter does.
 * pin = 1: the image pair the streams submit is page-locked with gfo_host_register (an application that owns its frame
 * buffers can do that once); default 0 = pageable memory, as cv::imread / a ROS message hand it over. */
#define _GNU_SOURCE
#include <dlfcn.h>
#include <pthread.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

#include "gfo.h"

#define W 752
#define H 480
#define FX 435.2046959714599
#define BF 47.90639384423901
#define CAP 2200
#define MAX_LAT 400000

static double now_ms(void)
{
    struct timespec t;
    clock_gettime(CLOCK_MONOTONIC, &t);
    return t.tv_sec * 1e3 + t.tv_nsec * 1e-6;
}

static int cmp_d(const void* a, const void* b) { return (*(const double*)a > *(const double*)b) - (*(const double*)a < *(const double*)b); }

static struct {
    __typeof__(&gfo_ctx_create) ctx_create;
    __typeof__(&gfo_ctx_destroy) ctx_destroy;
    __typeof__(&gfo_last_error) last_error;
    __typeof__(&gfo_extract) extract;
    __typeof__(&gfo_extract_stereo) extract_stereo;
    __typeof__(&gfo_stereo_match) stereo_match;
    __typeof__(&gfo_ctx_tables) ctx_tables;
    __typeof__(&gfo_contexts_created) contexts_created;
    __typeof__(&gfo_arenas_planned) arenas_planned;
    __typeof__(&gfo_ctx_set_combining) set_combining;
    __typeof__(&gfo_combiner_stats) combiner_stats;
    __typeof__(&gfo_host_register) host_register;
    __typeof__(&gfo_ctx_pair) ctx_pair;
    __typeof__(&gfo_combiner_counters) combiner_counters;
} G;
static int g_combine, g_pin, g_pair = 1;

static uint8_t IMGS[2 * W * H] __attribute__((aligned(4096)));   /* left | right, one buffer: a pinned pair is ONE copy */
#define IML (IMGS)
#define IMR (IMGS + W * H)
static volatile int g_stop, g_go;
static uint64_t g_ref_sum;
static volatile int g_have_ref;
static pthread_mutex_t g_mu = PTHREAD_MUTEX_INITIALIZER;

/* word-wise multiply-xor checksum (every buffer hashed here is a multiple of 4 bytes) */
/* checksum of a result array (every byte of it; sizes are multiples of 4).  Four independent multiply-xor lanes over 64-bit words:
 * the one-lane, 32-bit form of rounds 3-5 was a dependent chain of 77 000 multiplies per checked frame -- ~80 us, 5 us per frame of
 * the throughput the harness reports, spent on the harness's own arithmetic. */
static uint64_t fnv(uint64_t h, const void* p, size_t n)
{
    const uint8_t* b = (const uint8_t*)p;
    uint64_t a0 = h, a1 = h ^ 0x9E3779B97F4A7C15ULL, a2 = h ^ 0xC2B2AE3D27D4EB4FULL, a3 = h ^ 0x165667B19E3779F9ULL;
    size_t i = 0;
    for (; i + 32 <= n; i += 32) {
        uint64_t w[4];
        memcpy(w, b + i, 32);
        a0 = (a0 ^ w[0]) * 1099511628211ULL; a1 = (a1 ^ w[1]) * 1099511628211ULL;
        a2 = (a2 ^ w[2]) * 1099511628211ULL; a3 = (a3 ^ w[3]) * 1099511628211ULL;
    }
    for (; i + 4 <= n; i += 4) {
        uint32_t w;
        memcpy(&w, b + i, 4);
        a0 = (a0 ^ w) * 1099511628211ULL;
    }
    a0 ^= a1 + (a0 << 7); a0 *= 1099511628211ULL;
    a0 ^= a2 + (a0 >> 11); a0 *= 1099511628211ULL;
    a0 ^= a3 + (a0 << 13); a0 *= 1099511628211ULL;
    return a0;
}

typedef struct {
    int mode; /* 0 stereo, 1 adapter */
    int id;
    gfo_ctx *cl, *cr;
    gfo_keypoint* kp;
    uint8_t* desc;
    float *ur, *dp;
    int32_t *bd, *bi;
    float sf[GFO_MAX_LEVELS];
    double* lat;
    int nlat, frames, checked, mismatches, errors;
    int nl, nr, nm;
    double t_first, t_last;
} stream_t;

typedef struct {
    gfo_ctx* c;
    const uint8_t* img;
    gfo_keypoint* kp;
    uint8_t* desc;
    int n, rc;
} job_t;

static void* run_job(void* p)
{
    job_t* j = (job_t*)p;
    j->rc = G.extract(j->c, j->img, W, H, W, j->kp, j->desc, CAP, &j->n);
    return NULL;
}

static int g_nl, g_nr, g_nm;

/* one stereo frame through the boundary; the results stay in the stream's buffers for frame_sum() */
static int one_frame(stream_t* s)
{
    gfo_stereo_params sp = {H, (float)BF, (float)(BF / FX), 0.f};
    int nl = 0, nr = 0, nm = 0, rc;
    if (s->mode == 0) {
        rc = G.extract_stereo(s->cl, IML, IMR, W, H, W, &sp, s->kp, s->desc, s->kp + CAP, s->desc + 32 * (size_t)CAP, CAP, &nl, &nr, s->ur,
                              s->dp, s->bd, s->bi, &nm);
        if (rc) { fprintf(stderr, "stream %d: %s\n", s->id, G.last_error(s->cl)); return rc; }
    } else {
        job_t jl = {s->cl, IML, s->kp, s->desc, 0, 0}, jr = {s->cr, IMR, s->kp + CAP, s->desc + 32 * (size_t)CAP, 0, 0};
        pthread_t th;
        pthread_create(&th, NULL, run_job, &jr);
        run_job(&jl);
        pthread_join(th, NULL);
        if (jl.rc || jr.rc) { fprintf(stderr, "stream %d: %s / %s\n", s->id, G.last_error(s->cl), G.last_error(s->cr)); return -1; }
        nl = jl.n; nr = jr.n;
        if (g_pair) G.ctx_pair(s->cl, s->cr, &sp);   /* what the adapter's ComputeStereoMatches does before the association (idempotent) */
        rc = G.stereo_match(s->cl, s->kp, s->desc, nl, s->kp + CAP, s->desc + 32 * (size_t)CAP, nr, s->sf, 8, &sp, NULL, NULL, s->ur, s->dp,
                            s->bd, s->bi, &nm);
        if (rc) { fprintf(stderr, "stream %d: %s\n", s->id, G.last_error(s->cl)); return rc; }
    }
    s->nl = nl; s->nr = nr; s->nm = nm;
    return 0;
}

static uint64_t frame_sum(const stream_t* s)
{
    const int nl = s->nl, nr = s->nr, nm = s->nm;
    uint64_t h = 1469598103934665603ULL;
    h = fnv(h, &nl, 4); h = fnv(h, &nr, 4); h = fnv(h, &nm, 4);
    h = fnv(h, s->kp, sizeof(gfo_keypoint) * nl); h = fnv(h, s->kp + CAP, sizeof(gfo_keypoint) * nr);
    h = fnv(h, s->desc, 32 * (size_t)nl); h = fnv(h, s->desc + 32 * (size_t)CAP, 32 * (size_t)nr);
    h = fnv(h, s->ur, 4 * (size_t)nl); h = fnv(h, s->dp, 4 * (size_t)nl); h = fnv(h, s->bd, 4 * (size_t)nl); h = fnv(h, s->bi, 4 * (size_t)nl);
    return h;
}

static void* run_stream(void* p)
{
    stream_t* s = (stream_t*)p;
    /* warm-up: plans the arena, pins the staging buffers */
    for (int i = 0; i < 20; i++)
        if (one_frame(s)) { s->errors++; return NULL; }
    pthread_mutex_lock(&g_mu);
    if (!g_have_ref) { g_ref_sum = frame_sum(s); g_have_ref = 1; g_nl = s->nl; g_nr = s->nr; g_nm = s->nm; }
    pthread_mutex_unlock(&g_mu);
    __sync_fetch_and_add(&g_go, 1);
    while (g_go > 0 && !g_stop) { struct timespec ts = {0, 200000}; nanosleep(&ts, NULL); }   /* g_go is set negative by main when all are ready */
    s->t_first = now_ms();
    while (!g_stop) {
        const double t0 = now_ms();
        if (one_frame(s)) { s->errors++; break; }
        const double t1 = now_ms();
        if (s->nlat < MAX_LAT) s->lat[s->nlat++] = t1 - t0;
        s->frames++;
        /* every 16th frame is checksummed in full (outside the latency bracket, inside the throughput), every frame by its counts */
        if (s->nl != g_nl || s->nr != g_nr || s->nm != g_nm) s->mismatches++;
        else if ((s->frames & 15) == 1) { s->checked++; if (frame_sum(s) != g_ref_sum) s->mismatches++; }
        s->t_last = t1;
    }
    return NULL;
}

static int run_point(int mode, int K, double seconds, int first)
{
    gfo_params prm = {2000, 1.2f, 8, 20, 7, 2};
    stream_t* S = (stream_t*)calloc(K, sizeof(stream_t));
    pthread_t* T = (pthread_t*)calloc(K, sizeof(pthread_t));
    for (int k = 0; k < K; k++) {
        stream_t* s = &S[k];
        s->mode = mode; s->id = k;
        if (mode == 1) prm.max_batch = 1;
        if (G.ctx_create(&prm, 0, &s->cl) || (mode == 1 && G.ctx_create(&prm, 0, &s->cr))) { fprintf(stderr, "ctx: %s\n", G.last_error(NULL)); return 1; }
        G.ctx_tables(s->cl, s->sf, NULL, NULL, NULL, NULL);
        if (g_combine) { G.set_combining(s->cl, 1); if (s->cr) G.set_combining(s->cr, 1); }
        s->kp = (gfo_keypoint*)malloc(sizeof(gfo_keypoint) * CAP * 2);
        s->desc = (uint8_t*)malloc(32 * (size_t)CAP * 2);
        s->ur = (float*)malloc(4 * CAP); s->dp = (float*)malloc(4 * CAP);
        s->bd = (int32_t*)malloc(4 * CAP); s->bi = (int32_t*)malloc(4 * CAP);
        s->lat = (double*)malloc(sizeof(double) * MAX_LAT);
    }
    g_stop = 0; g_go = 0;
    for (int k = 0; k < K; k++) pthread_create(&T[k], NULL, run_stream, &S[k]);
    const double t_wait = now_ms();
    while (g_go < K && now_ms() - t_wait < 60000) { struct timespec ts = {0, 1000000}; nanosleep(&ts, NULL); }
    const int created0 = G.contexts_created(), planned0 = G.arenas_planned();
    int64_t cb0 = 0, cr0 = 0;
    G.combiner_stats(S[0].cl, &cb0, &cr0);
    const double t0 = now_ms();
    g_go = -1;
    { struct timespec ts = {(time_t)seconds, (long)((seconds - (long)seconds) * 1e9)}; nanosleep(&ts, NULL); }
    g_stop = 1;
    for (int k = 0; k < K; k++) pthread_join(T[k], NULL);
    const double t1 = now_ms();
    const int created1 = G.contexts_created(), planned1 = G.arenas_planned();
    int64_t cb = 0, cr = 0;
    G.combiner_stats(S[0].cl, &cb, &cr);
    cb -= cb0; cr -= cr0;
    int64_t rig[8] = {0}, rig_sum[3] = {0, 0, 0};
    for (int k = 0; k < K; k++) { G.combiner_counters(S[k].cl, rig, 8); rig_sum[0] += rig[5]; rig_sum[1] += rig[6]; rig_sum[2] += rig[7]; }
    long frames = 0, checked = 0; int mism = 0, errs = 0, nl = 0;
    for (int k = 0; k < K; k++) { frames += S[k].frames; checked += S[k].checked; mism += S[k].mismatches; errs += S[k].errors; nl += S[k].nlat; }
    double* all = (double*)malloc(sizeof(double) * (nl > 0 ? nl : 1));
    int q = 0;
    for (int k = 0; k < K; k++) { memcpy(all + q, S[k].lat, sizeof(double) * S[k].nlat); q += S[k].nlat; }
    qsort(all, nl, sizeof(double), cmp_d);
    const double wall = (t1 - t0) * 1e-3;
    printf("%s  {\"path\": \"%s\", \"caller_buffers_pinned\": %s, \"combining\": %s, \"frames_per_device_batch\": %.2f, \"streams\": %d, \"host_threads\": %d, \"contexts\": %d, \"seconds\": %.2f, \"stereo_frames\": %ld, "
           "\"images_per_s\": %.0f, \"stereo_frames_per_s\": %.0f, \"latency_ms\": {\"p50\": %.4f, \"p90\": %.4f, \"p99\": %.4f, \"max\": %.4f}, "
           "\"stereo_rig\": {\"declared\": %s, \"frames_as_one_submission\": %ld, \"associations_answered_from_them\": %ld, \"frames_alone\": %ld}, \"keypoints\": [%d, %d], \"stereo_candidates\": %d, \"frames_checksummed\": %ld, \"result_mismatches\": %d, \"errors\": %d, \"contexts_created_in_timed_region\": %d, \"arenas_planned_in_timed_region\": %d}",
           first ? "" : ",\n", mode == 0 ? "gfo_extract_stereo" : "adapter: 2 x gfo_extract on two threads + gfo_stereo_match", g_pin ? "true" : "false", g_combine ? "true" : "false",
           cb > 0 ? (double)cr / cb / (mode == 0 ? 1 : 2) : 1.0, K, mode == 0 ? K : 2 * K,
           mode == 0 ? K : 2 * K, wall, frames, 2.0 * frames / wall, frames / wall, nl ? all[nl / 2] : 0.0, nl ? all[(long)nl * 9 / 10] : 0.0,
           nl ? all[(long)nl * 99 / 100] : 0.0, nl ? all[nl - 1] : 0.0, (mode == 1 && g_pair) ? "true" : "false", (long)rig_sum[0], (long)rig_sum[1], (long)rig_sum[2], g_nl, g_nr, g_nm, checked, mism, errs, created1 - created0, planned1 - planned0);
    fflush(stdout);
    for (int k = 0; k < K; k++) {
        G.ctx_destroy(S[k].cl);
        if (S[k].cr) G.ctx_destroy(S[k].cr);
        free(S[k].kp); free(S[k].desc); free(S[k].ur); free(S[k].dp); free(S[k].bd); free(S[k].bi); free(S[k].lat);
    }
    free(all); free(S); free(T);
    return errs || mism ? 1 : 0;
}

#define SYM(field, name) G.field = (__typeof__(G.field))dlsym(lib, #name); if (!G.field) { fprintf(stderr, "missing %s\n", #name); return 2; }

int main(int argc, char** argv)
{
    const char* libpath = argc > 1 ? argv[1] : "gf-orb-slam2_amd/libgfo.so";
    const char* dir = argc > 2 ? argv[2] : "tests/golden";
    const double seconds = argc > 3 ? atof(argv[3]) : 2.0;
    const char* modes = argc > 4 ? argv[4] : "both";
    const char* klist = argc > 5 ? argv[5] : "1,2,4,8,16";
    void* lib = dlopen(libpath, RTLD_NOW);
    if (!lib) { fprintf(stderr, "%s\n", dlerror()); return 2; }
    SYM(ctx_create, gfo_ctx_create) SYM(ctx_destroy, gfo_ctx_destroy) SYM(last_error, gfo_last_error) SYM(extract, gfo_extract)
    SYM(extract_stereo, gfo_extract_stereo) SYM(stereo_match, gfo_stereo_match) SYM(ctx_tables, gfo_ctx_tables)
    SYM(contexts_created, gfo_contexts_created) SYM(arenas_planned, gfo_arenas_planned)
    SYM(set_combining, gfo_ctx_set_combining) SYM(combiner_stats, gfo_combiner_stats)
    g_combine = argc > 6 ? atoi(argv[6]) : 1;
    g_pin = argc > 7 ? atoi(argv[7]) : 0;
    SYM(host_register, gfo_host_register) SYM(ctx_pair, gfo_ctx_pair) SYM(combiner_counters, gfo_combiner_counters)
    g_pair = argc > 8 ? atoi(argv[8]) : 1;
    char path[512];
    snprintf(path, sizeof path, "%s/EuRoC_l_752x480.u8", dir);
    FILE* f = fopen(path, "rb");
    if (!f || fread(IML, 1, W * H, f) != W * H) { fprintf(stderr, "cannot read %s\n", path); return 2; }
    fclose(f);
    snprintf(path, sizeof path, "%s/EuRoC_r_752x480.u8", dir);
    f = fopen(path, "rb");
    if (!f || fread(IMR, 1, W * H, f) != W * H) { fprintf(stderr, "cannot read %s\n", path); return 2; }
    fclose(f);
    if (g_pin && G.host_register(IMGS, sizeof IMGS)) { fprintf(stderr, "gfo_host_register: %s\n", G.last_error(NULL)); return 2; }
    if (getenv("GFO_DUMP_MAPS") && atoi(getenv("GFO_DUMP_MAPS"))) {
        /* so that an abort in this process is symbolisable afterwards: one context created (the HIP / HSA runtimes, and under a
         * profiler its tool libraries, are mapped by then), then the process's address map to stderr */
        gfo_params p0 = {2000, 1.2f, 8, 20, 7, 2};
        gfo_ctx* c0 = NULL;
        if (G.ctx_create(&p0, 0, &c0) == 0) G.ctx_destroy(c0);
        FILE* m = fopen("/proc/self/maps", "r");
        if (m) {
            char line[1024];
            fprintf(stderr, "---- /proc/self/maps (GFO_DUMP_MAPS) ----\n");
            while (fgets(line, sizeof line, m))
                if (strstr(line, " r-xp ") || strstr(line, ".so")) fputs(line, stderr);
            fprintf(stderr, "---- end of maps ----\n");
            fclose(m);
        }
    }
    int bad = 0, first = 1;
    printf("{\"workload\": \"EuRoC stereo pair 752x480 @2000, one frame per call, K independent streams\", \"points\": [\n");
    for (int mode = 0; mode < 2; mode++) {
        if ((mode == 0 && !strcmp(modes, "adapter")) || (mode == 1 && !strcmp(modes, "stereo"))) continue;
        g_have_ref = 0;   /* the two paths agree on keypoints / descriptors / matches, but a checksum per path keeps the harness simple */
        char buf[256];
        strncpy(buf, klist, sizeof buf - 1); buf[sizeof buf - 1] = 0;
        char* save = NULL;   /* strtok_r: the HIP runtime tokenises environment variables with strtok while it initialises */
        for (char* tok = strtok_r(buf, ",", &save); tok; tok = strtok_r(NULL, ",", &save)) {
            const int K = atoi(tok);
            if (K < 1 || K > 64) continue;
            bad |= run_point(mode, K, seconds, first);
            first = 0;
        }
    }
    printf("\n]}\n");
    return bad;
}
