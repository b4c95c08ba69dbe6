// fake_gfo_internal.h -- TEST INFRASTRUCTURE: what gf-orb-slam2_amd/csrc/gfo_combine.hip needs from gfo_internal.h and the HIP
// runtime, as plain C++ for the CPU, so that the combiner's and the stereo rig's state machines -- the product file itself,
// unmodified, #included by tests/host/combine_tsan.cc -- can run under ThreadSanitizer (sanitizers are not
// available on the GPU pool).  The "device" is host memory; a batch "runs" for 30-200 us and returns results that are a pure
// function of each image's bytes, so the driver can tell whose results a caller got.  Every buffer access the real backend
// makes on the host side (staging copies, the pinned result block, the callers' arrays) is a real access here: a race on them
// is a race TSAN sees.
#pragma once
#define GFO_INTERNAL_H      // the real header is skipped when gfo_combine.hip includes it

#include <stdarg.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <atomic>
#include <chrono>
#include <memory>
#include <string>
#include <thread>
#include <vector>

#include "../../../include/gfo.h"

// ---- the few HIP runtime names the file uses ---------------------------------------------------------------------------
typedef int hipError_t;
typedef void* hipStream_t;
enum { hipSuccess = 0 };
enum { hipMemcpyHostToDevice = 1 };
enum { hipHostMallocDefault = 0 };
inline hipError_t hipSetDevice(int) { return hipSuccess; }
inline hipError_t hipGetLastError() { return hipSuccess; }
inline hipError_t hipHostMalloc(void** p, size_t n, unsigned) { *p = malloc(n ? n : 1); return *p ? hipSuccess : 2; }
inline hipError_t hipHostFree(void* p) { free(p); return hipSuccess; }
inline hipError_t hipMalloc(void** p, size_t n) { *p = malloc(n ? n : 1); return *p ? hipSuccess : 2; }
template <class T> inline hipError_t hipMalloc(T** p, size_t n) { return hipMalloc((void**)p, n); }
inline hipError_t hipFree(void* p) { free(p); return hipSuccess; }
inline hipError_t hipMemcpyAsync(void* d, const void* s, size_t n, int, hipStream_t) { memcpy(d, s, n); return hipSuccess; }
inline hipError_t hipStreamSynchronize(hipStream_t) { return hipSuccess; }

// ---- gfo_internal.h, the part gfo_combine.hip reads ------------------------------------------------------------------------
struct GfoEngine;
struct GfoPair;
#define GFO_COMBINE_DIRECT 1
#define GFO_COMBINER_COUNTERS 8
#define FAKE_KS 96          // kp_stride of the fake arena

struct gfo_ctx {
    int device = 0;
    gfo_params prm{};
    bool combining = false;
    std::shared_ptr<GfoEngine> engine;
    std::shared_ptr<GfoPair> pair;
    std::vector<float> scale;
    struct { int kp_stride = FAKE_KS, w0 = 0, h0 = 0; } g;
    int st_rows_cap = 0;
    uint8_t* h_in = nullptr;
    size_t h_in_bytes = 0;
    uint8_t* h_out = nullptr;
    uint8_t* d_input = nullptr;       // the fake device input
    hipStream_t stream = nullptr;
    std::string err;
    int batch_cap = 0;
};

struct GfoSmallLayout {
    int nimg_cap;
    int pitch;
    size_t img_bytes;
    size_t o_fl, o_cnt, o_kp, o_ds, o_ur, o_dp, o_bd, o_bi, o_nm;
};
struct GfoPairBlock {
    size_t bytes, o_kl, o_dl, o_kr, o_dr, o_min, o_max;
};

extern std::atomic<long> fake_batches, fake_contexts_alive;
extern std::atomic<int> fake_fail_submit_every;      // > 0: every n-th batch submission fails (GFO_ERR_OVERFLOW), like a tripped overflow flag

int gfo_fail(gfo_ctx* c, int code, const char* fmt, ...);
int gfo_plan(gfo_ctx* c, int w, int h, int batch);
int gfo_small_prepare(gfo_ctx* c, int nimg_cap, GfoSmallLayout* L);
int gfo_small_upload(gfo_ctx* c, gfo_ctx* ec, const GfoSmallLayout& L, int first, int count, const uint8_t* const* imgs, int w, int h, int stride, hipStream_t st,
                     bool lone_caller = false);
int gfo_small_submit(gfo_ctx* c, const GfoSmallLayout& L, int nimg, const gfo_stereo_params* sp, bool copy_in);
int gfo_small_collect(gfo_ctx* c, const GfoSmallLayout& L, int i, gfo_keypoint* kp, uint8_t* desc, int cap, int* n);
void gfo_small_collect_stereo(gfo_ctx* c, const GfoSmallLayout& L, int pair, int n_left, int cap, float* u_right, float* depth, int32_t* best_dist,
                              int32_t* best_idx_r, int* nmatched);
GfoPairBlock gfo_pair_block(int kp_stride);
int gfo_small_submit_pairs(gfo_ctx* c, const GfoSmallLayout& L, int npairs, const gfo_stereo_params* sp, const uint8_t* d_stage);
int gfo_combined_extract(gfo_ctx* c, int kind, const uint8_t* const* imgs, int w, int h, int stride, const gfo_stereo_params* sp, gfo_keypoint* const* kp,
                         uint8_t* const* desc, int cap, int* n, float* u_right, float* depth, int32_t* best_dist, int32_t* best_idx_r, int* nmatched);
int gfo_combined_stereo_match(gfo_ctx* c, const gfo_keypoint* kl, const uint8_t* dl, int nl, const gfo_keypoint* kr, const uint8_t* dr, int nr, const float* sf,
                              int nlevels, const gfo_stereo_params* p, const float* min_d, const float* max_d, float* u_right, float* depth, int32_t* best_dist,
                              int32_t* best_idx_r, int* nmatched, int* status);
void gfo_engine_release(gfo_ctx* c);
void gfo_pair_release(gfo_ctx* c);
bool gfo_has_pair(const gfo_ctx* c);
int gfo_pair_extract(gfo_ctx* c, const uint8_t* img, int w, int h, int stride, gfo_keypoint* kp, uint8_t* desc, int cap, int* n);
int gfo_pair_lookup(gfo_ctx* c, const gfo_keypoint* kl, const uint8_t* dl, int nl, const gfo_keypoint* kr, const uint8_t* dr, int nr, const float* sf, int nlevels,
                    const gfo_stereo_params* p, const float* min_d, const float* max_d, float* u_right, float* depth, int32_t* best_dist, int32_t* best_idx_r,
                    int* nmatched);
void gfo_note_pinned(const uint8_t* p, size_t bytes, bool add);

// what a frame's results are in the fake: a pure function of the image bytes (and, for the association, of both images)
inline int fake_count(const uint8_t* img) { return 8 + img[0] % 40; }
inline void fake_keypoint(const uint8_t* img, int j, gfo_keypoint* k, uint8_t* d)
{
    k->x = (float)(img[1] + j); k->y = (float)(img[2] * 2 + j); k->size = 31.f; k->angle = (float)img[3]; k->response = (float)j; k->octave = j & 7; k->class_id = -1;
    for (int b = 0; b < 32; b++) d[b] = (uint8_t)(img[4 + (b & 3)] + j * 7 + b);
}
inline float fake_uright(const uint8_t* l, const uint8_t* r, int j) { return (float)(l[5] + 3 * r[6] + j); }
inline int fake_nmatched(const uint8_t* l, const uint8_t* r) { return (l[0] + r[0]) % 7 + 1; }
