/* latency_pair.c -- wall time of ONE stereo pair through the C ABI (include/gfo.h), no Python in the loop:
 * what a Frame constructor of the reference sees per frame (Frame.cc:84-100).
 *
 *   A  "adapter" path : two contexts on two host threads (the reference's left / right extractor objects,
 *                       Frame.cc:84-87), gfo_extract each, then gfo_stereo_match on host arrays
 *   B  "batched" path : one context, gfo_extract_batch of {L, R}, gfo_stereo_match_batch, gfo_stereo_fetch
 *   C  "frame" path   : one context, gfo_extract_stereo -- both extractions and the association in ONE submission
 *                       (one H2D, one replay of the captured hipGraph, one D2H, one synchronisation)
 *
 * Build (tools/latency_pair.sh):  gcc -O2 -I include tools/c/latency_pair.c -o /tmp/latency_pair -ldl -lpthread -lm
 * The library is loaded with dlopen so that the harness needs no HIP toolchain. */
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

static double now_ms(void)
{
    struct timespec t;
    clock_gettime(CLOCK_MONOTONIC, &t);
    return t.tv_sec * 1e3 + t.tv_nsec * 1e-6;
}

static int cmp_d(const void* a, const void* b) { return (*(const double*)a > *(const double*)b) - (*(const double*)a < *(const double*)b); }

#define SYM(name) __typeof__(&name) p_##name = (__typeof__(&name))dlsym(lib, #name); if (!p_##name) { fprintf(stderr, "missing %s\n", #name); return 2; }

typedef struct {
    int (*extract)(gfo_ctx*, const uint8_t*, int, int, int, gfo_keypoint*, uint8_t*, int, int*);
    gfo_ctx* c;
    const uint8_t* img;
    gfo_keypoint* kp;
    uint8_t* desc;
    int cap, n, rc;
} job_t;

static void* run_job(void* p)
{
    job_t* j = (job_t*)p;
    j->rc = j->extract(j->c, j->img, W, H, W, j->kp, j->desc, j->cap, &j->n);
    return NULL;
}

int main(int argc, char** argv)
{
    const char* libpath = argc > 1 ? argv[1] : "gf-orb-slam2_amd/libgfo.so";
    const char* dir = argc > 2 ? argv[2] : "tests/golden";
    const int reps = argc > 3 ? atoi(argv[3]) : 200;
    void* lib = dlopen(libpath, RTLD_NOW);
    if (!lib) { fprintf(stderr, "%s\n", dlerror()); return 2; }
    SYM(gfo_ctx_create) SYM(gfo_ctx_destroy) SYM(gfo_last_error) SYM(gfo_extract) SYM(gfo_extract_batch) SYM(gfo_ctx_max_keypoints)
    SYM(gfo_stereo_match) SYM(gfo_stereo_match_batch) SYM(gfo_stereo_fetch) SYM(gfo_ctx_tables) SYM(gfo_extract_stereo)
    static uint8_t L[W * H], R[W * H];
    char path[512];
    snprintf(path, sizeof path, "%s/EuRoC_l_752x480.u8", dir);
    FILE* f = fopen(path, "rb");
    if (!f || fread(L, 1, sizeof L, f) != sizeof L) { fprintf(stderr, "cannot read %s\n", path); return 2; }
    fclose(f);
    snprintf(path, sizeof path, "%s/EuRoC_r_752x480.u8", dir);
    f = fopen(path, "rb");
    if (!f || fread(R, 1, sizeof R, f) != sizeof R) { fprintf(stderr, "cannot read %s\n", path); return 2; }
    fclose(f);

    gfo_params prm = {2000, 1.2f, 8, 20, 7, 2};
    gfo_ctx *cl = NULL, *cr = NULL, *cb = NULL, *cf = NULL;
    if (p_gfo_ctx_create(&prm, 0, &cl) || p_gfo_ctx_create(&prm, 0, &cr) || p_gfo_ctx_create(&prm, 0, &cb) || p_gfo_ctx_create(&prm, 0, &cf)) {
        fprintf(stderr, "ctx: %s\n", p_gfo_last_error(NULL));
        return 1;
    }
    int cap = 2200;
    gfo_keypoint* kp = (gfo_keypoint*)malloc(sizeof(gfo_keypoint) * cap * 2);
    uint8_t* desc = (uint8_t*)malloc(32 * (size_t)cap * 2);
    float* ur = (float*)malloc(4 * cap); float* dp = (float*)malloc(4 * cap);
    int32_t* bd = (int32_t*)malloc(4 * cap); int32_t* bi = (int32_t*)malloc(4 * cap);
    float sf[GFO_MAX_LEVELS];
    p_gfo_ctx_tables(cl, sf, NULL, NULL, NULL, NULL);
    gfo_stereo_params sp = {H, (float)BF, (float)(BF / FX), 0.f};
    double* ta = (double*)malloc(sizeof(double) * reps);
    double* tb = (double*)malloc(sizeof(double) * reps);
    double* tb1 = (double*)malloc(sizeof(double) * reps);
    double* tc = (double*)malloc(sizeof(double) * reps);
    int nmA = 0, nmB = 0, nmC = 0, nlA = 0, nlB = 0, nlC = 0, nrC = 0;
    for (int it = -10; it < reps; it++) {
        /* ---- A: two contexts, two threads, host-array stereo ---- */
        double t0 = now_ms();
        job_t jl = {p_gfo_extract, cl, L, kp, desc, cap, 0, 0}, jr = {p_gfo_extract, cr, R, kp + cap, desc + 32 * (size_t)cap, cap, 0, 0};
        pthread_t th;
        pthread_create(&th, NULL, run_job, &jr);
        run_job(&jl);
        pthread_join(th, NULL);
        if (jl.rc || jr.rc) { fprintf(stderr, "extract: %s %s\n", p_gfo_last_error(cl), p_gfo_last_error(cr)); return 1; }
        int rc = p_gfo_stereo_match(cl, kp, desc, jl.n, kp + cap, desc + 32 * (size_t)cap, jr.n, sf, 8, &sp, NULL, NULL, ur, dp, bd, bi, &nmA);
        if (rc) { fprintf(stderr, "stereo: %s\n", p_gfo_last_error(cl)); return 1; }
        double t1 = now_ms();
        nlA = jl.n;
        /* ---- B: one context, one batch of two, device chain ---- */
        const uint8_t* imgs[2] = {L, R};
        int n2[2];
        rc = p_gfo_extract_batch(cb, imgs, 2, W, H, W, kp, desc, cap, n2);
        double t1b = now_ms();
        if (!rc) rc = p_gfo_stereo_match_batch(cb, &sp);
        if (!rc) rc = p_gfo_stereo_fetch(cb, 0, ur, dp, bd, bi, cap, &nmB);
        if (rc) { fprintf(stderr, "batched: %s\n", p_gfo_last_error(cb)); return 1; }
        double t2 = now_ms();
        nlB = n2[0];
        /* ---- C: the whole stereo frame in one submission ---- */
        rc = p_gfo_extract_stereo(cf, L, R, W, H, W, &sp, kp, desc, kp + cap, desc + 32 * (size_t)cap, cap, &nlC, &nrC, ur, dp, bd, bi, &nmC);
        if (rc) { fprintf(stderr, "frame: %s\n", p_gfo_last_error(cf)); return 1; }
        double t3 = now_ms();
        if (it >= 0) { ta[it] = t1 - t0; tb[it] = t2 - t1; tb1[it] = t1b - t1; tc[it] = t3 - t2; }
    }
    qsort(ta, reps, sizeof(double), cmp_d);
    qsort(tb, reps, sizeof(double), cmp_d);
    qsort(tb1, reps, sizeof(double), cmp_d);
    qsort(tc, reps, sizeof(double), cmp_d);
    printf("{\"pair\": \"EuRoC 752x480 @2000\", \"reps\": %d, \"keypoints_left\": [%d, %d, %d], \"stereo_candidates\": [%d, %d, %d],\n"
           " \"adapter_path_two_contexts_two_threads_ms\": {\"median\": %.4f, \"min\": %.4f, \"p90\": %.4f},\n"
           " \"batched_path_one_context_ms\": {\"median\": %.4f, \"min\": %.4f, \"p90\": %.4f, \"of_which_extract_batch_median\": %.4f},\n"
           " \"frame_path_one_submission_ms\": {\"median\": %.4f, \"min\": %.4f, \"p90\": %.4f}}\n",
           reps, nlA, nlB, nlC, nmA, nmB, nmC, ta[reps / 2], ta[0], ta[reps * 9 / 10], tb[reps / 2], tb[0], tb[reps * 9 / 10], tb1[reps / 2],
           tc[reps / 2], tc[0], tc[reps * 9 / 10]);
    p_gfo_ctx_destroy(cl); p_gfo_ctx_destroy(cr); p_gfo_ctx_destroy(cb); p_gfo_ctx_destroy(cf);
    return 0;
}
