/* orb_oracle.c -- CPU restatement of the reference ORB front-end.  TEST INFRASTRUCTURE ONLY.
 * See orb_oracle.h for the parity status ("parity unpinned" at the OpenCV boundary).
 *
 * Citations "ORBextractor.cc:N" etc. are relative to /root/reference/src|include.
 * Items tagged [OCV] restate OpenCV 3.4.x behaviour from its published sources (the library
 * is not available in this image); everything else follows the reference's own lines.
 *
 * Build: gcc -O3 -ffp-contract=off (see oracle/Makefile).  No dependencies but libm.
 */
#include "orb_oracle.h"

#include <float.h>
#include <limits.h>
#include <math.h>
#include <stdlib.h>
#include <string.h>

#define PATCH_SIZE 31      /* ORBextractor.cc:72 */
#define HALF_PATCH_SIZE 15 /* ORBextractor.cc:73 */
#define EDGE_THRESHOLD 19  /* ORBextractor.cc:74 */
#define MAX_LEVELS 32

static const signed char k_pattern[256][4] = {
#include "../include/gfo_pattern.inc"
};

typedef struct {
    int x, y, score;
} cand_t;

struct orc_extractor {
    int nfeatures, nlevels, ini_th, min_th;
    float scale_factor;
    float scale[MAX_LEVELS], inv_scale[MAX_LEVELS], sigma2[MAX_LEVELS], inv_sigma2[MAX_LEVELS];
    int quota[MAX_LEVELS];
    int umax[HALF_PATCH_SIZE + 1];
    int trig_mode, rot_mode;
    /* pyramid, unpadded, stride == width */
    int w[MAX_LEVELS], h[MAX_LEVELS];
    uint8_t* level[MAX_LEVELS];
    uint8_t* blurred[MAX_LEVELS];
    /* per-level products of the last extraction */
    cand_t* cand[MAX_LEVELS];
    int ncand[MAX_LEVELS];
    int nkp[MAX_LEVELS];
};

/* ------------------------------------------------------------------------------------------
 * [OCV] variant table.  Every piece of OpenCV 3.4.1 arithmetic this file restates from memory of the published
 * sources is a SINGLE switch here, so that whoever can run cv2 3.4.x (tests/golden/check_against_cv2.py does the
 * comparison stage by stage and names the switch) turns a mismatch into a one-line change of
 * oracle/ocv_variants.json followed by tests/golden/make_golden.py.  The defaults are the variants the golden vectors
 * and the GPU kernels use; "parity unpinned" (orb_oracle.h) stays until that script has run green once.
 *   ORC_OCV_RESIZE      0 = 11-bit fixed-point bilinear (the generic resizeGeneric_ path; default)
 *                       1 = float bilinear, rounded half to even (what an IPP-backed build computes to within 1 LSB)
 *   ORC_OCV_ATAN_FMA    0 = polynomial in separate multiplies and adds (default)   1 = Horner steps fused (v_fma builds)
 *   ORC_OCV_BLUR_ROUND  0 = exact accumulation, one (v + 2^15) >> 16 at the end (default; the classic int path and
 *                           the 8.8 fixed-point path agree on it)
 *                       1 = horizontal pass rounded to 8 bits first ((v + 128) >> 8), then the vertical pass likewise
 *   Gaussian taps       orc_set_gauss_taps(): seven integers, scale 256 ({18,34,49,55,49,34,18}, sum 257, = round(256 k);
 *                           a build whose fixed-point kernel is renormalised to sum 256 needs another centre tap)
 * ---------------------------------------------------------------------------------------- */
static int g_ocv[ORC_OCV_COUNT] = {0, 0, 0};
static int g_gauss7[7] = {18, 34, 49, 55, 49, 34, 18};
int orc_set_ocv_variant(int key, int value)
{
    if (key < 0 || key >= ORC_OCV_COUNT) return -1;
    g_ocv[key] = value;
    return 0;
}
int orc_get_ocv_variant(int key) { return key < 0 || key >= ORC_OCV_COUNT ? -1 : g_ocv[key]; }
void orc_set_gauss_taps(const int* taps7) { memcpy(g_gauss7, taps7, sizeof g_gauss7); }
void orc_get_gauss_taps(int* taps7) { memcpy(taps7, g_gauss7, sizeof g_gauss7); }

/* array forms for the sweep tests (tests/test_oracle.py) */
void orc_fast_atan2_n(const float* y, const float* x, int n, float* out)
{
    for (int i = 0; i < n; i++) out[i] = orc_fast_atan2(y[i], x[i]);
}
void orc_sincos_n(const float* t, int n, float* s, float* c)
{
    for (int i = 0; i < n; i++) orc_sincos(t[i], &s[i], &c[i]);
}

/* [OCV] cvRound: round half to even (SSE cvtss2si under the default rounding mode). */
int orc_cv_round(float v) { return (int)lrintf(v); }
static int cv_round_d(double v) { return (int)lrint(v); }

/* cv::fastAtan2(y, x), OpenCV 3.4.x scalar path (modules/core/src/mathfuncs_core.simd.hpp, atan_f32): a degree-7 odd
 * polynomial of min/max in plain float, folded into [0, 360).  Call site: src/ORBextractor.cc:102.  [OCV]
 * Restated HERE, independently of the product's include/gfo_sincos.h (which the kernels use): the constants below
 * were typed a second time from the published coefficients, so a mistyped digit on either side shows up
 * as a GPU-vs-oracle mismatch instead of passing silently. */
static const float k_rad2deg = 57.295779513082320876798154814105f;            /* (float)(180 / CV_PI) */
static const float k_atan_c1 = 0.9997878412794807f, k_atan_c3 = -0.3258083974640975f;
static const float k_atan_c5 = 0.1555786518463281f, k_atan_c7 = -0.04432655554792128f;
float orc_fast_atan2(float y, float x)
{
    const float q1 = k_atan_c1 * k_rad2deg, q3 = k_atan_c3 * k_rad2deg, q5 = k_atan_c5 * k_rad2deg, q7 = k_atan_c7 * k_rad2deg;
    const float tiny = (float)DBL_EPSILON;
    const float mx = fabsf(x), my = fabsf(y);
    const int steep = !(mx >= my);                      /* ax >= ay takes the first branch, NaNs included */
    const float num = steep ? mx : my, den = (steep ? my : mx) + tiny;
    const float r = num / den, r2 = r * r;
    float deg;
    if (g_ocv[ORC_OCV_ATAN_FMA]) deg = fmaf(fmaf(fmaf(q7, r2, q5), r2, q3), r2, q1) * r;
    else deg = (((q7 * r2 + q5) * r2 + q3) * r2 + q1) * r;
    if (steep) deg = 90.f - deg;
    if (x < 0) deg = 180.f - deg;
    if (y < 0) deg = 360.f - deg;
    return deg;
}

/* a = (float)cos(angle), b = (float)sin(angle) of src/ORBextractor.cc:111-112, evaluated in the widest host type and
 * rounded ONCE to float: the correctly rounded result (up to ties finer than 2^-63), which is what a correctly
 * rounded libm gives the reference.  Independent of include/gfo_sincos.h (the kernels' double-precision evaluation);
 * tests/test_oracle.py asserts the two agree on every angle the fixtures produce and on a 1M-angle sweep. */
void orc_sincos(float t, float* s, float* c)
{
    const long double w = (long double)t;
    *s = (float)sinl(w);
    *c = (float)cosl(w);
}

/* ------------------------------------------------------------------------------------------
 * ORBextractor::ORBextractor -- ORBextractor.cc:409-469
 * ---------------------------------------------------------------------------------------- */
orc_extractor* orc_create(int nfeatures, float scale_factor, int nlevels, int ini_th, int min_th)
{
    if (nlevels < 1 || nlevels > MAX_LEVELS) return NULL;
    orc_extractor* e = (orc_extractor*)calloc(1, sizeof(*e));
    e->nfeatures = nfeatures;
    e->scale_factor = scale_factor;
    e->nlevels = nlevels;
    e->ini_th = ini_th;
    e->min_th = min_th;
    e->scale[0] = 1.0f; /* :416-422, cumulative float product */
    e->sigma2[0] = 1.0f;
    for (int i = 1; i < nlevels; i++) {
        e->scale[i] = e->scale[i - 1] * scale_factor;
        e->sigma2[i] = e->scale[i] * e->scale[i];
    }
    for (int i = 0; i < nlevels; i++) { /* :426-430 */
        e->inv_scale[i] = 1.0f / e->scale[i];
        e->inv_sigma2[i] = 1.0f / e->sigma2[i];
    }
    /* :435-445 per-level quotas */
    /* the header stores scaleFactor as double (ORBextractor.h:155): 1.0f / double */
    float factor = (float)(1.0f / (double)scale_factor);
    float per_scale = nfeatures * (1 - factor) / (1 - (float)pow((double)factor, (double)nlevels));
    int sum = 0;
    for (int l = 0; l < nlevels - 1; l++) {
        e->quota[l] = orc_cv_round(per_scale);
        sum += e->quota[l];
        per_scale *= factor;
    }
    e->quota[nlevels - 1] = nfeatures - sum > 0 ? nfeatures - sum : 0;
    /* :451-468 umax */
    int v, v0;
    int vmax = (int)floor(HALF_PATCH_SIZE * sqrtf(2.f) / 2 + 1);
    int vmin = (int)ceil(HALF_PATCH_SIZE * sqrtf(2.f) / 2);
    const double hp2 = HALF_PATCH_SIZE * HALF_PATCH_SIZE;
    for (v = 0; v <= vmax; ++v) e->umax[v] = cv_round_d(sqrt(hp2 - v * v));
    for (v = HALF_PATCH_SIZE, v0 = 0; v >= vmin; --v) {
        while (e->umax[v0] == e->umax[v0 + 1]) ++v0;
        e->umax[v] = v0;
        ++v0;
    }
    return e;
}

static void free_products(orc_extractor* e)
{
    for (int l = 0; l < MAX_LEVELS; l++) {
        free(e->level[l]);
        free(e->blurred[l]);
        free(e->cand[l]);
        e->level[l] = e->blurred[l] = NULL;
        e->cand[l] = NULL;
        e->ncand[l] = e->nkp[l] = 0;
    }
}

void orc_destroy(orc_extractor* e)
{
    if (!e) return;
    free_products(e);
    free(e);
}

void orc_set_variant(orc_extractor* e, int trig_mode, int rot_mode)
{
    e->trig_mode = trig_mode;
    e->rot_mode = rot_mode;
}

int orc_nlevels(const orc_extractor* e) { return e->nlevels; }
const float* orc_scale_factors(const orc_extractor* e) { return e->scale; }
const float* orc_inv_scale_factors(const orc_extractor* e) { return e->inv_scale; }
const float* orc_level_sigma2(const orc_extractor* e) { return e->sigma2; }
const float* orc_inv_level_sigma2(const orc_extractor* e) { return e->inv_sigma2; }
const int* orc_features_per_level(const orc_extractor* e) { return e->quota; }
const int* orc_umax(const orc_extractor* e) { return e->umax; }

/* ------------------------------------------------------------------------------------------
 * [OCV] cv::resize(..., INTER_LINEAR) for CV_8UC1, the non-IPP fixed-point path:
 * 11-bit coefficients, horizontal pass into int, vertical pass with the >>4, >>16, +2, >>2
 * rounding of VResizeLinear<uchar,int,short,...>.  Called at ORBextractor.cc:1189.
 * ---------------------------------------------------------------------------------------- */
static void resize_axis_tables(int ssize, int dsize, int* ofs, short* coef, int clamp_x)
{
    const double inv_scale = (double)dsize / ssize; /* resize(): inv_scale_x = dsize.width/ssize.width */
    const double scale = 1. / inv_scale;            /* hal::resize(): scale_x = 1./inv_scale_x          */
    for (int d = 0; d < dsize; d++) {
        float f = (float)((d + 0.5) * scale - 0.5);
        int s = (int)floorf(f);
        f -= s;
        if (clamp_x) {
            if (s < 0) { f = 0; s = 0; }
            if (s >= ssize - 1) { f = 0; s = ssize - 1; }
        }
        ofs[d] = s;
        coef[2 * d] = (short)orc_cv_round((1.f - f) * 2048.f);
        coef[2 * d + 1] = (short)orc_cv_round(f * 2048.f);
    }
}

/* variant 1: the same sample positions, interpolated in float and rounded once */
static void resize_linear_float_u8(const uint8_t* src, int sw, int sh, int sstride, uint8_t* dst, int dw, int dh, int dstride)
{
    const double sx_ = 1. / ((double)dw / sw), sy_ = 1. / ((double)dh / sh);
    for (int dy = 0; dy < dh; dy++) {
        float fy = (float)((dy + 0.5) * sy_ - 0.5);
        int y0 = (int)floorf(fy);
        fy -= y0;
        if (y0 < 0) { y0 = 0; fy = 0; }
        if (y0 >= sh - 1) { y0 = sh - 1; fy = 0; }
        const int y1 = y0 + 1 < sh ? y0 + 1 : y0;
        for (int dx = 0; dx < dw; dx++) {
            float fx = (float)((dx + 0.5) * sx_ - 0.5);
            int x0 = (int)floorf(fx);
            fx -= x0;
            if (x0 < 0) { x0 = 0; fx = 0; }
            if (x0 >= sw - 1) { x0 = sw - 1; fx = 0; }
            const int x1 = x0 + 1 < sw ? x0 + 1 : x0;
            const float top = src[(size_t)y0 * sstride + x0] * (1.f - fx) + src[(size_t)y0 * sstride + x1] * fx;
            const float bot = src[(size_t)y1 * sstride + x0] * (1.f - fx) + src[(size_t)y1 * sstride + x1] * fx;
            const int v = orc_cv_round(top * (1.f - fy) + bot * fy);
            dst[(size_t)dy * dstride + dx] = (uint8_t)(v < 0 ? 0 : v > 255 ? 255 : v);
        }
    }
}

void orc_resize_linear_u8(const uint8_t* src, int sw, int sh, int sstride,
                          uint8_t* dst, int dw, int dh, int dstride)
{
    if (g_ocv[ORC_OCV_RESIZE] == 1) {
        resize_linear_float_u8(src, sw, sh, sstride, dst, dw, dh, dstride);
        return;
    }
    int* xofs = (int*)malloc(sizeof(int) * dw);
    int* yofs = (int*)malloc(sizeof(int) * dh);
    short* alpha = (short*)malloc(sizeof(short) * 2 * dw);
    short* beta = (short*)malloc(sizeof(short) * 2 * dh);
    int* row0 = (int*)malloc(sizeof(int) * dw);
    int* row1 = (int*)malloc(sizeof(int) * dw);
    resize_axis_tables(sw, dw, xofs, alpha, 1);
    resize_axis_tables(sh, dh, yofs, beta, 0);
    for (int dy = 0; dy < dh; dy++) {
        int sy0 = yofs[dy], sy1 = yofs[dy] + 1; /* rows clipped to [0, sh-1] */
        if (sy0 < 0) sy0 = 0;
        if (sy0 > sh - 1) sy0 = sh - 1;
        if (sy1 < 0) sy1 = 0;
        if (sy1 > sh - 1) sy1 = sh - 1;
        const uint8_t* S0 = src + (size_t)sy0 * sstride;
        const uint8_t* S1 = src + (size_t)sy1 * sstride;
        for (int dx = 0; dx < dw; dx++) {
            int sx = xofs[dx];
            int sx1 = sx + 1 < sw ? sx + 1 : sx; /* coefficient is 0 there */
            int a0 = alpha[2 * dx], a1 = alpha[2 * dx + 1];
            row0[dx] = S0[sx] * a0 + S0[sx1] * a1;
            row1[dx] = S1[sx] * a0 + S1[sx1] * a1;
        }
        int b0 = beta[2 * dy], b1 = beta[2 * dy + 1];
        uint8_t* D = dst + (size_t)dy * dstride;
        for (int dx = 0; dx < dw; dx++)
            D[dx] = (uint8_t)((((b0 * (row0[dx] >> 4)) >> 16) + ((b1 * (row1[dx] >> 4)) >> 16) + 2) >> 2);
    }
    free(xofs); free(yofs); free(alpha); free(beta); free(row0); free(row1);
}

/* ComputePyramid -- ORBextractor.cc:1176-1201.  The 19-px frame is not stored: nothing on the
 * extraction path reads it (FAST cells start 16 px inside, the angle disc reaches 15 px and the
 * rotated pattern 18 px from keypoints that sit >= 19 px inside); orc_get_level_padded rebuilds
 * it on demand. */
void orc_compute_pyramid(orc_extractor* e, const uint8_t* img, int w, int h, int stride)
{
    free_products(e);
    for (int l = 0; l < e->nlevels; l++) {
        float scale = e->inv_scale[l];
        e->w[l] = orc_cv_round((float)w * scale);
        e->h[l] = orc_cv_round((float)h * scale);
        e->level[l] = (uint8_t*)malloc((size_t)e->w[l] * e->h[l]);
        if (l == 0) {
            for (int y = 0; y < h; y++) memcpy(e->level[0] + (size_t)y * w, img + (size_t)y * stride, w);
        } else {
            orc_resize_linear_u8(e->level[l - 1], e->w[l - 1], e->h[l - 1], e->w[l - 1],
                                 e->level[l], e->w[l], e->h[l], e->w[l]);
        }
    }
}

int orc_level_size(const orc_extractor* e, int level, int* w, int* h)
{
    if (level < 0 || level >= e->nlevels || !e->level[level]) return -1;
    *w = e->w[level];
    *h = e->h[level];
    return 0;
}

void orc_get_level(const orc_extractor* e, int level, uint8_t* out, int out_stride)
{
    for (int y = 0; y < e->h[level]; y++)
        memcpy(out + (size_t)y * out_stride, e->level[level] + (size_t)y * e->w[level], e->w[level]);
}

/* [OCV] BORDER_REFLECT_101: -k -> k, n-1+k -> n-1-k */
static int reflect101(int p, int n)
{
    if (n <= 1) return 0; /* (n == 0: a level that rounds to no pixels at all -- cv::resize would assert; nothing is read through it) */
    while (p < 0 || p >= n) {
        if (p < 0) p = -p;
        else p = 2 * (n - 1) - p;
    }
    return p;
}

void orc_get_level_padded(const orc_extractor* e, int level, uint8_t* out, int out_stride)
{
    const int w = e->w[level], h = e->h[level];
    if (w <= 0 || h <= 0) return; /* an empty level has no frame to reflect into */
    for (int y = -EDGE_THRESHOLD; y < h + EDGE_THRESHOLD; y++) {
        const uint8_t* S = e->level[level] + (size_t)reflect101(y, h) * w;
        uint8_t* D = out + (size_t)(y + EDGE_THRESHOLD) * out_stride;
        for (int x = -EDGE_THRESHOLD; x < w + EDGE_THRESHOLD; x++) D[x + EDGE_THRESHOLD] = S[reflect101(x, w)];
    }
}

/* ------------------------------------------------------------------------------------------
 * [OCV] cv::GaussianBlur(src, dst, Size(7,7), 2, 2, BORDER_REFLECT_101) for CV_8UC1
 * (ORBextractor.cc:1154-1155, applied to a clone of the level, so the border is the level's own
 * reflection).  Taps = round(256 * normalised exp(-x^2/8)) = {18,34,49,55,49,34,18} (sum 257);
 * both the classic integer sepFilter2D path and the 8.8 fixed-point path of 3.4.1 accumulate
 * exactly (no intermediate rounding) and finish with (v + 2^15) >> 16 saturated to 255, so the
 * two give the same bytes.
 * ---------------------------------------------------------------------------------------- */
void orc_gaussian_blur7_u8(const uint8_t* src, int w, int h, int sstride, uint8_t* dst, int dstride)
{
    const int* taps = g_gauss7;                              /* the variant table, default {18,34,49,55,49,34,18} */
    const int per_pass = g_ocv[ORC_OCV_BLUR_ROUND] == 1;
    int* tmp = (int*)malloc(sizeof(int) * (size_t)w * h);
    for (int y = 0; y < h; y++) {
        const uint8_t* S = src + (size_t)y * sstride;
        for (int x = 0; x < w; x++) {
            int acc = 0;
            if (x >= 3 && x < w - 3) {   /* interior: no reflection (same sum, same order of exact integer terms) */
                for (int k = -3; k <= 3; k++) acc += taps[k + 3] * S[x + k];
            } else {
                for (int k = -3; k <= 3; k++) acc += taps[k + 3] * S[reflect101(x + k, w)];
            }
            if (per_pass) { acc = (acc + 128) >> 8; acc = acc > 255 ? 255 : acc; }
            tmp[(size_t)y * w + x] = acc;
        }
    }
    for (int y = 0; y < h; y++) {
        const int* R[7];
        for (int k = -3; k <= 3; k++) R[k + 3] = tmp + (size_t)reflect101(y + k, h) * w;
        for (int x = 0; x < w; x++) {
            unsigned acc = 0;
            for (int k = 0; k < 7; k++) acc += (unsigned)taps[k] * (unsigned)R[k][x];
            unsigned v = per_pass ? (acc + 128u) >> 8 : (acc + 32768u) >> 16;
            dst[(size_t)y * dstride + x] = (uint8_t)(v > 255 ? 255 : v);
        }
    }
    free(tmp);
}

void orc_get_blurred_level(const orc_extractor* e, int level, uint8_t* out, int out_stride)
{
    if (!e->blurred[level]) return; /* a level without keypoints is never blurred (ORBextractor.cc:1150-1152): `out` stays as it is
                                       (found by the sanitizer job of round 6: the copy read from a null plane) */
    for (int y = 0; y < e->h[level]; y++)
        memcpy(out + (size_t)y * out_stride, e->blurred[level] + (size_t)y * e->w[level], e->w[level]);
}

/* ------------------------------------------------------------------------------------------
 * [OCV] cv::FAST(img, kps, threshold, nonmaxSuppression=true) == FAST_t<16>: 9 contiguous of
 * 16 ring pixels all darker than v-t or all brighter than v+t; score = cornerScore<16>;
 * 3x3 strict-max suppression on the score rows; 3-px frame skipped.  In-tree mirror:
 * FAST_NEON.cc:3-7 (ring), :106-108 (threshold table), :121-133,198-260 (scan), :268-285 (NMS).
 * ---------------------------------------------------------------------------------------- */
static const int k_ring[16][2] = {{0, 3}, {1, 3}, {2, 2}, {3, 1}, {3, 0}, {3, -1}, {2, -2}, {1, -3},
                                  {0, -3}, {-1, -3}, {-2, -2}, {-3, -1}, {-3, 0}, {-3, 1}, {-2, 2}, {-1, 3}};

static int corner_score16(const uint8_t* ptr, const int* pixel, int threshold)
{
    int d[25];
    const int v = ptr[0];
    for (int k = 0; k < 25; k++) d[k] = v - ptr[pixel[k]];
    int a0 = threshold;
    for (int k = 0; k < 16; k += 2) {
        int a = d[k + 1] < d[k + 2] ? d[k + 1] : d[k + 2];
        for (int j = 3; j <= 8; j++) a = a < d[k + j] ? a : d[k + j];
        int a1 = a < d[k] ? a : d[k];
        int a2 = a < d[k + 9] ? a : d[k + 9];
        if (a1 > a0) a0 = a1;
        if (a2 > a0) a0 = a2;
    }
    int b0 = -a0;
    for (int k = 0; k < 16; k += 2) {
        int b = d[k + 1] > d[k + 2] ? d[k + 1] : d[k + 2];
        for (int j = 3; j <= 8; j++) b = b > d[k + j] ? b : d[k + j];
        int b1 = b > d[k] ? b : d[k];
        int b2 = b > d[k + 9] ? b : d[k + 9];
        if (b1 < b0) b0 = b1;
        if (b2 < b0) b0 = b2;
    }
    return -b0 - 1;
}

int orc_fast9_nms(const uint8_t* img, int cols, int rows, int stride, int threshold, int* xys, int cap)
{
    int pixel[25];
    for (int k = 0; k < 16; k++) pixel[k] = k_ring[k][0] + k_ring[k][1] * stride;
    for (int k = 16; k < 25; k++) pixel[k] = pixel[k - 16];
    if (threshold < 0) threshold = 0;
    if (threshold > 255) threshold = 255;
    if (cols < 1 || rows < 1) return 0;
    uint8_t* buf = (uint8_t*)calloc((size_t)cols * 3, 1);
    int* cpbuf = (int*)malloc(sizeof(int) * (size_t)(cols + 1) * 3);
    int n = 0;
    for (int i = 3; i < rows - 2; i++) {
        const uint8_t* ptr = img + (size_t)i * stride + 3;
        uint8_t* curr = buf + (size_t)((i - 3) % 3) * cols;
        int* cornerpos = cpbuf + (size_t)((i - 3) % 3) * (cols + 1) + 1;
        memset(curr, 0, cols);
        int ncorners = 0;
        if (i < rows - 3) {
            for (int j = 3; j < cols - 3; j++, ptr++) {
                const int v = ptr[0];
                int is_corner = 0;
                /* the high-speed test cv::FAST runs first (FAST_NEON.cc:121-197 mirrors it as the tab[] look-ups on ring
                 * pixels 0 / 8, then 2 / 10, 4 / 12, 6 / 14): a 9-arc contains a member of every opposite pair, so without a
                 * pixel beyond the threshold -- of ONE sign -- in both (0, 8) and (4, 12) there is no corner.  Rejects ~90 %
                 * of the pixels on four reads; the exact run-length test below decides the rest. */
                {
                    const int lo = v - threshold, hi = v + threshold;
                    const int p0 = ptr[pixel[0]], p8 = ptr[pixel[8]], p4 = ptr[pixel[4]], p12 = ptr[pixel[12]];
                    const int dark = (p0 < lo || p8 < lo) && (p4 < lo || p12 < lo);
                    const int bright = (p0 > hi || p8 > hi) && (p4 > hi || p12 > hi);
                    if (!dark && !bright) continue;
                }
                for (int pol = 0; pol < 2 && !is_corner; pol++) {
                    int count = 0;
                    for (int k = 0; k < 25; k++) {
                        const int x = ptr[pixel[k]];
                        const int hit = pol == 0 ? (x < v - threshold) : (x > v + threshold);
                        if (hit) {
                            if (++count > 8) { is_corner = 1; break; }
                        } else count = 0;
                    }
                }
                if (is_corner) {
                    cornerpos[ncorners++] = j;
                    curr[j] = (uint8_t)corner_score16(ptr, pixel, threshold);
                }
            }
        }
        cornerpos[-1] = ncorners;
        if (i == 3) continue;
        const uint8_t* prev = buf + (size_t)((i - 4 + 3) % 3) * cols;
        const uint8_t* pprev = buf + (size_t)((i - 5 + 3) % 3) * cols;
        cornerpos = cpbuf + (size_t)((i - 4 + 3) % 3) * (cols + 1) + 1;
        ncorners = cornerpos[-1];
        for (int k = 0; k < ncorners; k++) {
            const int j = cornerpos[k];
            const int score = prev[j];
            if (score > prev[j + 1] && score > prev[j - 1] &&
                score > pprev[j - 1] && score > pprev[j] && score > pprev[j + 1] &&
                score > curr[j - 1] && score > curr[j] && score > curr[j + 1]) {
                if (n < cap) {
                    xys[3 * n] = j;
                    xys[3 * n + 1] = i - 1;
                    xys[3 * n + 2] = score;
                }
                n++;
            }
        }
    }
    free(buf);
    free(cpbuf);
    return n;
}

/* ------------------------------------------------------------------------------------------
 * ExtractorNode / DistributeOctTree -- ORBextractor.cc:481-763, restated literally with a
 * doubly linked list.  The one deliberate difference: the reference sorts (size, node address)
 * pairs (:684), which makes the order of equal-sized nodes depend on the allocator; here every
 * node carries its creation sequence number and equal sizes are split NEWEST FIRST -- what a
 * monotonically growing heap would give.
 * ---------------------------------------------------------------------------------------- */
typedef struct qnode {
    int ulx, uly, urx, ury, blx, bly, brx, bry;
    int* keys; /* indices into the candidate array, in insertion order */
    int nkeys;
    int no_more;
    int seq;
    struct qnode *prev, *next;
} qnode;

typedef struct {
    qnode *head, *tail;
    int size, next_seq;
} qlist;

static qnode* qnode_new(int cap)
{
    qnode* n = (qnode*)calloc(1, sizeof(qnode));
    n->keys = (int*)malloc(sizeof(int) * (cap > 0 ? cap : 1));
    return n;
}
static void qnode_free(qnode* n) { free(n->keys); free(n); }
static void qlist_push_front(qlist* l, qnode* n)
{
    n->seq = l->next_seq++;
    n->prev = NULL;
    n->next = l->head;
    if (l->head) l->head->prev = n; else l->tail = n;
    l->head = n;
    l->size++;
}
static void qlist_push_back(qlist* l, qnode* n)
{
    n->seq = l->next_seq++;
    n->next = NULL;
    n->prev = l->tail;
    if (l->tail) l->tail->next = n; else l->head = n;
    l->tail = n;
    l->size++;
}
static qnode* qlist_erase(qlist* l, qnode* n) /* returns the following node */
{
    qnode* nx = n->next;
    if (n->prev) n->prev->next = n->next; else l->head = n->next;
    if (n->next) n->next->prev = n->prev; else l->tail = n->prev;
    l->size--;
    qnode_free(n);
    return nx;
}

/* ExtractorNode::DivideNode -- ORBextractor.cc:481-537 */
static void divide_node(const qnode* p, const cand_t* c, qnode* ch[4])
{
    const int halfX = (int)ceil((float)(p->urx - p->ulx) / 2);
    const int halfY = (int)ceil((float)(p->bry - p->uly) / 2);
    for (int i = 0; i < 4; i++) ch[i] = qnode_new(p->nkeys);
    qnode *n1 = ch[0], *n2 = ch[1], *n3 = ch[2], *n4 = ch[3];
    n1->ulx = p->ulx; n1->uly = p->uly;
    n1->urx = p->ulx + halfX; n1->ury = p->uly;
    n1->blx = p->ulx; n1->bly = p->uly + halfY;
    n1->brx = p->ulx + halfX; n1->bry = p->uly + halfY;
    n2->ulx = n1->urx; n2->uly = n1->ury;
    n2->urx = p->urx; n2->ury = p->ury;
    n2->blx = n1->brx; n2->bly = n1->bry;
    n2->brx = p->urx; n2->bry = p->uly + halfY;
    n3->ulx = n1->blx; n3->uly = n1->bly;
    n3->urx = n1->brx; n3->ury = n1->bry;
    n3->blx = p->blx; n3->bly = p->bly;
    n3->brx = n1->brx; n3->bry = p->bly;
    n4->ulx = n3->urx; n4->uly = n3->ury;
    n4->urx = n2->brx; n4->ury = n2->bry;
    n4->blx = n3->brx; n4->bly = n3->bry;
    n4->brx = p->brx; n4->bry = p->bry;
    for (int i = 0; i < p->nkeys; i++) {
        const cand_t* kp = &c[p->keys[i]];
        qnode* dst;
        if ((float)kp->x < (float)n1->urx) dst = ((float)kp->y < (float)n1->bry) ? n1 : n3;
        else dst = ((float)kp->y < (float)n1->bry) ? n2 : n4;
        dst->keys[dst->nkeys++] = p->keys[i];
    }
    for (int i = 0; i < 4; i++)
        if (ch[i]->nkeys == 1) ch[i]->no_more = 1;
}

typedef struct {
    int size;
    qnode* node;
} size_node;

static int size_node_cmp(const void* a, const void* b)
{
    const size_node* x = (const size_node*)a;
    const size_node* y = (const size_node*)b;
    if (x->size != y->size) return x->size < y->size ? -1 : 1;
    return x->node->seq < y->node->seq ? -1 : (x->node->seq > y->node->seq ? 1 : 0);
}

/* pushes the non-empty children to the front (n1..n4 in that order), records the expandable
 * ones; returns how many of them hold more than one key */
static int push_children(qlist* l, qnode* ch[4], size_node* vec, int* nvec)
{
    int expandable = 0;
    for (int i = 0; i < 4; i++) {
        if (ch[i]->nkeys > 0) {
            qlist_push_front(l, ch[i]);
            if (ch[i]->nkeys > 1) {
                expandable++;
                vec[*nvec].size = ch[i]->nkeys;
                vec[*nvec].node = ch[i];
                (*nvec)++;
            }
        } else qnode_free(ch[i]);
    }
    return expandable;
}

/* returns number of selected keys; sel[] = candidate indices in list order */
static int distribute_oct_tree(const cand_t* c, int nc, int minX, int maxX, int minY, int maxY, int N, int* sel)
{
    if (nc == 0) return 0;
    int nIni = (int)roundf((float)(maxX - minX) / (maxY - minY)); /* :543 */
    if (nIni < 1) nIni = 1; /* the reference divides by zero here for very tall images */
    const float hX = (float)(maxX - minX) / nIni;
    qlist L = {0, 0, 0, 0};
    qnode** ini = (qnode**)malloc(sizeof(qnode*) * nIni);
    for (int i = 0; i < nIni; i++) { /* :551-563 */
        qnode* n = qnode_new(nc);
        n->ulx = (int)(hX * (float)i); n->uly = 0;
        n->urx = (int)(hX * (float)(i + 1)); n->ury = 0;
        n->blx = n->ulx; n->bly = maxY - minY;
        n->brx = n->urx; n->bry = maxY - minY;
        qlist_push_back(&L, n);
        ini[i] = n;
    }
    for (int i = 0; i < nc; i++) { /* :566-570 */
        int idx = (int)((float)c[i].x / hX);
        if (idx >= nIni) idx = nIni - 1; /* cannot happen for in-range keys; guards the array */
        ini[idx]->keys[ini[idx]->nkeys++] = i;
    }
    free(ini);
    for (qnode* it = L.head; it;) { /* :572-585 */
        if (it->nkeys == 1) { it->no_more = 1; it = it->next; }
        else if (it->nkeys == 0) it = qlist_erase(&L, it);
        else it = it->next;
    }
    int finish = 0;
    size_t vcap = (size_t)nc + 16;
    size_node* vec = (size_node*)malloc(sizeof(size_node) * vcap);
    size_node* prevvec = (size_node*)malloc(sizeof(size_node) * vcap);
    int nvec = 0;
    while (!finish) { /* :594 */
        int prevSize = L.size;
        int nToExpand = 0;
        nvec = 0;
        for (qnode* it = L.head; it;) {
            if (it->no_more) { it = it->next; continue; }
            qnode* ch[4];
            divide_node(it, c, ch);
            nToExpand += push_children(&L, ch, vec, &nvec);
            it = qlist_erase(&L, it);
        }
        if (L.size >= N || L.size == prevSize) finish = 1; /* :667-670 */
        else if (L.size + nToExpand * 3 > N) {              /* :671 */
            while (!finish) {
                prevSize = L.size;
                int nprev = nvec;
                memcpy(prevvec, vec, sizeof(size_node) * nprev);
                nvec = 0;
                qsort(prevvec, nprev, sizeof(size_node), size_node_cmp); /* :684, seq instead of address */
                for (int j = nprev - 1; j >= 0; j--) {
                    qnode* ch[4];
                    divide_node(prevvec[j].node, c, ch);
                    push_children(&L, ch, vec, &nvec);
                    qlist_erase(&L, prevvec[j].node);
                    if (L.size >= N) break; /* :730 */
                }
                if (L.size >= N || L.size == prevSize) finish = 1;
            }
        }
    }
    /* :740-760 best response per node, first wins on ties */
    int ns = 0;
    for (qnode* it = L.head; it; it = it->next) {
        int best = it->keys[0];
        int maxr = c[best].score;
        for (int k = 1; k < it->nkeys; k++)
            if (c[it->keys[k]].score > maxr) { best = it->keys[k]; maxr = c[best].score; }
        sel[ns++] = best;
    }
    for (qnode* it = L.head; it;) it = qlist_erase(&L, it);
    free(vec);
    free(prevvec);
    return ns;
}

/* IC_Angle -- ORBextractor.cc:76-103 (image = unblurred level, stride = width) */
static float ic_angle(const uint8_t* img, int stride, int px, int py, const int* umax)
{
    int m_01 = 0, m_10 = 0;
    const uint8_t* center = img + (size_t)py * stride + px;
    for (int u = -HALF_PATCH_SIZE; u <= HALF_PATCH_SIZE; ++u) m_10 += u * center[u];
    for (int v = 1; v <= HALF_PATCH_SIZE; ++v) {
        int v_sum = 0;
        const int d = umax[v];
        for (int u = -d; u <= d; ++u) {
            const int val_plus = center[u + v * stride], val_minus = center[u - v * stride];
            v_sum += (val_plus - val_minus);
            m_10 += u * (val_plus + val_minus);
        }
        m_01 += v * v_sum;
    }
    return orc_fast_atan2((float)m_01, (float)m_10);
}

/* computeOrbDescriptor -- ORBextractor.cc:106-146 */
static void orb_descriptor(const orc_extractor* e, const uint8_t* img, int stride, int px, int py,
                           float angle_deg, uint8_t* desc)
{
    const float factorPI = (float)(3.1415926535897932384626433832795 / 180.f);
    const float angle = angle_deg * factorPI;
    float a, b;
    if (e->trig_mode == ORC_TRIG_LIBM) { a = cosf(angle); b = sinf(angle); }
    else orc_sincos(angle, &b, &a);
    const uint8_t* center = img + (size_t)py * stride + px;
    for (int i = 0; i < 32; i++) {
        int val = 0;
        for (int j = 0; j < 8; j++) {
            const signed char* p = k_pattern[i * 8 + j];
            int t[2];
            for (int s = 0; s < 2; s++) {
                const float x = (float)p[2 * s], y = (float)p[2 * s + 1];
                float fy, fx;
                if (e->rot_mode == ORC_ROT_FMA) { fy = fmaf(x, b, y * a); fx = fmaf(x, a, -(y * b)); }
                else { fy = x * b + y * a; fx = x * a - y * b; }
                t[s] = center[orc_cv_round(fy) * stride + orc_cv_round(fx)];
            }
            val |= (t[0] < t[1]) << j;
        }
        desc[i] = (uint8_t)val;
    }
}

/* ORBextractor::operator() -- ORBextractor.cc:1112-1174 with ComputeKeyPointsOctTree :767-855 */
int orc_extract(orc_extractor* e, const uint8_t* img, int w, int h, int stride,
                orc_keypoint* kp, uint8_t* desc, int cap)
{
    if (!img || w <= 0 || h <= 0) return 0; /* :1115 */
    orc_compute_pyramid(e, img, w, h, stride);
    const float W = 30;
    int total = 0;
    for (int level = 0; level < e->nlevels; ++level) {
        const int lw = e->w[level], lh = e->h[level];
        const uint8_t* L = e->level[level];
        const int minBorderX = EDGE_THRESHOLD - 3, minBorderY = minBorderX;
        const int maxBorderX = lw - EDGE_THRESHOLD + 3, maxBorderY = lh - EDGE_THRESHOLD + 3;
        const float width = (float)(maxBorderX - minBorderX), height = (float)(maxBorderY - minBorderY);
        int ncand = 0, ccap = 0;
        cand_t* cand = NULL;
        if (width >= W && height >= W) { /* nCols/nRows == 0 divides by zero in the reference */
            const int nCols = (int)(width / W), nRows = (int)(height / W);
            const int wCell = (int)ceil(width / nCols), hCell = (int)ceil(height / nRows);
            int* xys = (int*)malloc(sizeof(int) * 3 * (size_t)(wCell + 6) * (hCell + 6));
            for (int i = 0; i < nRows; i++) {
                const float iniY = (float)(minBorderY + i * hCell);
                float maxY = iniY + hCell + 6;
                if (iniY >= maxBorderY - 3) continue;
                if (maxY > maxBorderY) maxY = (float)maxBorderY;
                for (int j = 0; j < nCols; j++) {
                    const float iniX = (float)(minBorderX + j * wCell);
                    float maxX = iniX + wCell + 6;
                    if (iniX >= maxBorderX - 6) continue;
                    if (maxX > maxBorderX) maxX = (float)maxBorderX;
                    const int x0 = (int)iniX, y0 = (int)iniY, cw = (int)maxX - x0, chh = (int)maxY - y0;
                    const uint8_t* sub = L + (size_t)y0 * lw + x0;
                    int n = orc_fast9_nms(sub, cw, chh, lw, e->ini_th, xys, cw * chh);
                    if (n == 0) n = orc_fast9_nms(sub, cw, chh, lw, e->min_th, xys, cw * chh);
                    for (int k = 0; k < n; k++) {
                        if (ncand == ccap) {
                            ccap = ccap ? ccap * 2 : 4096;
                            cand = (cand_t*)realloc(cand, sizeof(cand_t) * ccap);
                        }
                        cand[ncand].x = xys[3 * k] + j * wCell;
                        cand[ncand].y = xys[3 * k + 1] + i * hCell;
                        cand[ncand].score = xys[3 * k + 2];
                        ncand++;
                    }
                }
            }
            free(xys);
        }
        e->cand[level] = cand;
        e->ncand[level] = ncand;
        int* sel = (int*)malloc(sizeof(int) * (ncand + 1));
        const int nsel = distribute_oct_tree(cand, ncand, minBorderX, maxBorderX, minBorderY, maxBorderY,
                                             e->quota[level], sel);
        e->nkp[level] = nsel;
        if (nsel > 0) {
            /* :1154-1155 blur a clone of the level */
            e->blurred[level] = (uint8_t*)malloc((size_t)lw * lh);
            orc_gaussian_blur7_u8(L, lw, lh, lw, e->blurred[level], lw);
        }
        const int scaledPatchSize = (int)(PATCH_SIZE * e->scale[level]);
        for (int k = 0; k < nsel; k++) {
            const cand_t* c = &cand[sel[k]];
            orc_keypoint q;
            q.x = (float)c->x + minBorderX;
            q.y = (float)c->y + minBorderY;
            q.size = (float)scaledPatchSize;
            q.response = (float)c->score;
            q.octave = level;
            q.class_id = -1;
            const int px = orc_cv_round(q.x), py = orc_cv_round(q.y);
            q.angle = ic_angle(L, lw, px, py, e->umax);
            if (total + k < cap) {
                if (desc) orb_descriptor(e, e->blurred[level], lw, px, py, q.angle, desc + (size_t)(total + k) * 32);
                if (level != 0) { /* :1164-1170 */
                    q.x *= e->scale[level];
                    q.y *= e->scale[level];
                }
                if (kp) kp[total + k] = q;
            }
        }
        total += nsel;
        free(sel);
    }
    return total;
}

int orc_level_candidates(const orc_extractor* e, int level, int* xys, int cap)
{
    const int n = e->ncand[level];
    for (int i = 0; i < n && i < cap; i++) {
        xys[3 * i] = e->cand[level][i].x;
        xys[3 * i + 1] = e->cand[level][i].y;
        xys[3 * i + 2] = e->cand[level][i].score;
    }
    return n;
}

int orc_level_keypoint_count(const orc_extractor* e, int level) { return e->nkp[level]; }

/* ------------------------------------------------------------------------------------------
 * ORBmatcher::DescriptorDistance -- ORBmatcher.cc:1768-1784 (SWAR popcount over 8 x u32)
 * ---------------------------------------------------------------------------------------- */
int orc_hamming256(const uint8_t* a, const uint8_t* b)
{
    int dist = 0;
    for (int i = 0; i < 8; i++) {
        uint32_t pa, pb;
        memcpy(&pa, a + 4 * i, 4);
        memcpy(&pb, b + 4 * i, 4);
        uint32_t v = pa ^ pb;
        v = v - ((v >> 1) & 0x55555555u);
        v = (v & 0x33333333u) + ((v >> 2) & 0x33333333u);
        dist += (int)((((v + (v >> 4)) & 0xF0F0F0Fu) * 0x1010101u) >> 24);
    }
    return dist;
}

/* ------------------------------------------------------------------------------------------
 * Stereo association -- Frame.h:230-263 (PrepareStereoCandidates) and Frame.cc:1167-1316
 * (ComputeStereoMatches_Undistorted, isOnline=false, DELAYED_STEREO_MATCHING bookkeeping
 * left to the caller).  TH_HIGH=100, TH_LOW=50 (ORBmatcher.cc:57-58).
 * ---------------------------------------------------------------------------------------- */
typedef struct {
    int dist, il;
} dist_idx;
static int dist_idx_cmp(const void* a, const void* b)
{
    const dist_idx* x = (const dist_idx*)a;
    const dist_idx* y = (const dist_idx*)b;
    if (x->dist != y->dist) return x->dist < y->dist ? -1 : 1;
    return x->il < y->il ? -1 : (x->il > y->il ? 1 : 0);
}

/* Frame::ComputeStereoMatches_Undistorted(bool isOnline): the outlier cut is under `if (!isOnline)` (Frame.cc:1290), outside every
 * #ifdef -- an online call keeps every accepted match.  (The reference's default build only ever passes false; with
 * DELAYED_STEREO_MATCHING the online call also restricts WHICH keypoints are visited, :1186-1199, which is not restated here.) */
static int g_stereo_online = 0;
void orc_set_stereo_online(int on) { g_stereo_online = on != 0; }

int orc_stereo_match(const orc_keypoint* kl, const uint8_t* dl, int nl,
                     const orc_keypoint* kr, const uint8_t* dr, int nr,
                     const float* scale_factors, const orc_stereo_params* p,
                     const float* min_d_in, const float* max_d_in,
                     float* u_right, float* depth, int* best_dist_out, int* best_idx_out)
{
    const int TH_HIGH = 100, TH_LOW = 50;
    const int thOrbDist = (TH_HIGH + TH_LOW) / 2;
    const int nRows = p->n_rows;
    /* row table (PrepareStereoCandidates): CSR, each row keeps insertion order by iR */
    int* start = (int*)calloc((size_t)nRows + 1, sizeof(int));
    int* minr_a = (int*)malloc(sizeof(int) * (nr > 0 ? nr : 1));
    int* maxr_a = (int*)malloc(sizeof(int) * (nr > 0 ? nr : 1));
    for (int iR = 0; iR < nr; iR++) {
        const float kpY = kr[iR].y;
        const float r = 2.0f * scale_factors[kr[iR].octave];
        const float fmaxr = ceilf(kpY + r), fminr = floorf(kpY - r);
        maxr_a[iR] = (int)((float)(nRows - 1) < fmaxr ? (float)(nRows - 1) : fmaxr);
        minr_a[iR] = (int)(0.0f > fminr ? 0.0f : fminr);
        for (int yi = minr_a[iR]; yi <= maxr_a[iR]; yi++) start[yi + 1]++;
    }
    for (int r = 0; r < nRows; r++) start[r + 1] += start[r];
    int* items = (int*)malloc(sizeof(int) * (start[nRows] > 0 ? start[nRows] : 1));
    int* fill = (int*)calloc((size_t)nRows, sizeof(int));
    for (int iR = 0; iR < nr; iR++)
        for (int yi = minr_a[iR]; yi <= maxr_a[iR]; yi++) items[start[yi] + fill[yi]++] = iR;
    free(fill); free(minr_a); free(maxr_a);

    const float minZ = p->mb;
    int nmatched = 0;
    dist_idx* di = (dist_idx*)malloc(sizeof(dist_idx) * (nl > 0 ? nl : 1));
    int ndi = 0;
    for (int iL = 0; iL < nl; iL++) {
        u_right[iL] = -1.0f;
        depth[iL] = -1.0f;
        if (best_dist_out) best_dist_out[iL] = -1;
        if (best_idx_out) best_idx_out[iL] = -1;
    }
    for (int iL = 0; iL < nl; iL++) {
        float minD = 0;
        float maxD = p->mbf / minZ;
        const int levelL = kl[iL].octave;
        const float vL = kl[iL].y, uL = kl[iL].x;
        if (vL < 0 || vL > nRows - 1) continue;
        const int row = (int)vL;
        const int nC = start[row + 1] - start[row];
        if (nC == 0) continue;
        if (min_d_in && max_d_in) { minD = min_d_in[iL]; maxD = max_d_in[iL]; }
        const float minU = uL - maxD, maxU = uL - minD;
        if (maxU < p->min_x) continue;
        int bestDist = TH_HIGH;
        int bestIdxR = 0;
        for (int iC = 0; iC < nC; iC++) {
            const int iR = items[start[row] + iC];
            if (kr[iR].octave < levelL - 1 || kr[iR].octave > levelL + 1) continue;
            const float uR = kr[iR].x;
            if (uR >= minU && uR <= maxU) {
                const int dist = orc_hamming256(dl + (size_t)iL * 32, dr + (size_t)iR * 32);
                if (dist < bestDist) { bestDist = dist; bestIdxR = iR; }
            }
        }
        if (bestDist < thOrbDist) {
            float bestuR = kr[bestIdxR].x;
            float disparity = uL - bestuR;
            if (disparity >= minD && disparity < maxD) {
                if (disparity <= 0) { disparity = 0.01f; bestuR = (float)((double)uL - 0.01); /* double arithmetic, then narrowed: Frame.cc:1054,1278 */ }
                depth[iL] = p->mbf / disparity;
                u_right[iL] = bestuR;
                if (best_dist_out) best_dist_out[iL] = bestDist;
                if (best_idx_out) best_idx_out[iL] = bestIdxR;
                di[ndi].dist = bestDist;
                di[ndi].il = iL;
                ndi++;
            }
        }
        nmatched++;
    }
    if (ndi > 0 && !g_stereo_online) { /* :1290-1313 */
        qsort(di, ndi, sizeof(dist_idx), dist_idx_cmp);
        const float median = (float)di[ndi / 2].dist;
        const float thDist = 1.5f * 1.4f * median;
        for (int i = ndi - 1; i >= 0; i--) {
            if ((float)di[i].dist < thDist) break;
            u_right[di[i].il] = -1;
            depth[di[i].il] = -1;
            nmatched--;
        }
    }
    free(di); free(start); free(items);
    return nmatched;
}

/* ------------------------------------------------------------------------------------------
 * The stereo MEMBERS of one Frame across several calls -- Frame.h:230-263,335-339, Frame.cc:1167-1316.
 * orc_stereo_match above is one call on a fresh frame.  The reference's member keeps state between calls on one frame:
 * mvuRight / mvDepth / mvStereoMatched / mvDistIdx are reset ONLY inside PrepareStereoCandidates, which the member runs only
 * when `mvRowIndices.size() != nRows` (:1173-1176).  Tracking.cc:941-954 (default build: ALTER_STEREO_MATCHING on, DELAYED off,
 * ORB_SLAM_BASELINE off) calls it a second time on mCurrentFrame after map points narrowed the windows: a keypoint whose narrowed
 * window rejects its match KEEPS the first call's uRight / depth, every accepted match is appended to mvDistIdx again (:1282), the
 * list is sorted and cut as a whole (:1296-1313; entries the first call cut are still in it), and nmatched is decremented for
 * every cut entry, duplicates included.  This object is that state, literally; `delayed` != 0 is the member compiled with
 * DELAYED_STEREO_MATCHING (:1186-1199: the online call visits only unvisited keypoints that carry a map point, the offline call
 * the other unvisited ones).
 * ---------------------------------------------------------------------------------------- */
struct orc_stereo_frame {
    int n;                 /* N */
    float* u_right;        /* mvuRight */
    float* depth;          /* mvDepth */
    unsigned char* matched;/* mvStereoMatched */
    dist_idx* di;          /* mvDistIdx */
    int ndi, cap_di;
    int rows;              /* mvRowIndices.size() */
    int* start;            /* mvRowIndices as CSR: each row keeps insertion order by iR */
    int* items;
};

orc_stereo_frame* orc_stereo_frame_new(void) { return (orc_stereo_frame*)calloc(1, sizeof(orc_stereo_frame)); }

void orc_stereo_frame_free(orc_stereo_frame* s)
{
    if (!s) return;
    free(s->u_right); free(s->depth); free(s->matched); free(s->di); free(s->start); free(s->items);
    free(s);
}

/* Frame::PrepareStereoCandidates -- Frame.h:230-263 */
void orc_stereo_frame_prepare(orc_stereo_frame* s, int nl, const orc_keypoint* kr, int nr, const float* scale_factors, int nRows)
{
    free(s->u_right); free(s->depth); free(s->matched); free(s->start); free(s->items);
    s->n = nl;
    s->u_right = (float*)malloc(sizeof(float) * (nl > 0 ? nl : 1));
    s->depth = (float*)malloc(sizeof(float) * (nl > 0 ? nl : 1));
    s->matched = (unsigned char*)calloc((size_t)(nl > 0 ? nl : 1), 1);
    for (int i = 0; i < nl; i++) { s->u_right[i] = -1.0f; s->depth[i] = -1.0f; }
    s->rows = nRows;
    s->start = (int*)calloc((size_t)nRows + 1, sizeof(int));
    int* minr_a = (int*)malloc(sizeof(int) * (nr > 0 ? nr : 1));
    int* maxr_a = (int*)malloc(sizeof(int) * (nr > 0 ? nr : 1));
    for (int iR = 0; iR < nr; iR++) {
        const float kpY = kr[iR].y;
        const float r = 2.0f * scale_factors[kr[iR].octave];
        const float fmaxr = ceilf(kpY + r), fminr = floorf(kpY - r);
        maxr_a[iR] = (int)((float)(nRows - 1) < fmaxr ? (float)(nRows - 1) : fmaxr);
        minr_a[iR] = (int)(0.0f > fminr ? 0.0f : fminr);
        for (int yi = minr_a[iR]; yi <= maxr_a[iR]; yi++) s->start[yi + 1]++;
    }
    for (int r = 0; r < nRows; r++) s->start[r + 1] += s->start[r];
    s->items = (int*)malloc(sizeof(int) * (s->start[nRows] > 0 ? s->start[nRows] : 1));
    int* fill = (int*)calloc((size_t)(nRows > 0 ? nRows : 1), sizeof(int));
    for (int iR = 0; iR < nr; iR++)
        for (int yi = minr_a[iR]; yi <= maxr_a[iR]; yi++) s->items[s->start[yi] + fill[yi]++] = iR;
    free(fill); free(minr_a); free(maxr_a);
    s->ndi = 0; /* mvDistIdx.clear() */
}

/* `mvStereoMatched = vector<bool>(N,false)` of the Frame constructors, AFTER their own association call (Frame.cc:118,216) */
void orc_stereo_frame_clear_matched(orc_stereo_frame* s)
{
    if (s->matched) memset(s->matched, 0, (size_t)(s->n > 0 ? s->n : 1));
}

static void stereo_frame_push(orc_stereo_frame* s, int dist, int il)
{
    if (s->ndi == s->cap_di) {
        s->cap_di = s->cap_di ? 2 * s->cap_di : 1024;
        s->di = (dist_idx*)realloc(s->di, sizeof(dist_idx) * (size_t)s->cap_di);
    }
    s->di[s->ndi].dist = dist;
    s->di[s->ndi].il = il;
    s->ndi++;
}

/* Frame::ComputeStereoMatches_Undistorted(isOnline) -- Frame.cc:1167-1316.  has_mp[iL] = (mvpMapPoints[iL] != NULL), NULL = none;
 * min_d / max_d = the window of :1220-1231 per keypoint (NULL = 0 and mbf/mb everywhere), as the adapter flattens it. */
int orc_stereo_frame_match(orc_stereo_frame* s, const orc_keypoint* kl, const uint8_t* dl, int nl,
                           const orc_keypoint* kr, const uint8_t* dr, int nr, const float* scale_factors,
                           const orc_stereo_params* p, const float* min_d_in, const float* max_d_in,
                           const unsigned char* has_mp, int is_online, int delayed)
{
    const int TH_HIGH = 100, TH_LOW = 50;
    const int thOrbDist = (TH_HIGH + TH_LOW) / 2;
    const int nRows = p->n_rows;
    if (s->rows != nRows) orc_stereo_frame_prepare(s, nl, kr, nr, scale_factors, nRows); /* :1173-1176 */
    const float minZ = p->mb;
    int nmatched = 0;
    for (int iL = 0; iL < nl; iL++) {
        if (delayed) { /* :1186-1199 */
            if (is_online) {
                if (!(has_mp && has_mp[iL]) || s->matched[iL]) continue;
            } else {
                if (s->matched[iL]) continue;
            }
        }
        float minD = 0;
        float maxD = p->mbf / minZ;
        s->matched[iL] = 1;
        const int levelL = kl[iL].octave;
        const float vL = kl[iL].y, uL = kl[iL].x;
        if (vL < 0 || vL > nRows - 1) continue;
        const int row = (int)vL;
        const int nC = s->start[row + 1] - s->start[row];
        if (nC == 0) continue;
        if (min_d_in && max_d_in) { minD = min_d_in[iL]; maxD = max_d_in[iL]; }
        const float minU = uL - maxD, maxU = uL - minD;
        if (maxU < p->min_x) continue;
        int bestDist = TH_HIGH;
        int bestIdxR = 0;
        for (int iC = 0; iC < nC; iC++) {
            const int iR = s->items[s->start[row] + iC];
            if (kr[iR].octave < levelL - 1 || kr[iR].octave > levelL + 1) continue;
            const float uR = kr[iR].x;
            if (uR >= minU && uR <= maxU) {
                const int dist = orc_hamming256(dl + (size_t)iL * 32, dr + (size_t)iR * 32);
                if (dist < bestDist) { bestDist = dist; bestIdxR = iR; }
            }
        }
        if (bestDist < thOrbDist) {
            float bestuR = kr[bestIdxR].x;
            float disparity = uL - bestuR;
            if (disparity >= minD && disparity < maxD) {
                if (disparity <= 0) { disparity = 0.01f; bestuR = (float)((double)uL - 0.01); }
                s->depth[iL] = p->mbf / disparity;
                s->u_right[iL] = bestuR;
                stereo_frame_push(s, bestDist, iL); /* :1282 */
            }
        }
        nmatched++;
    }
    if (!is_online) { /* :1290-1313 */
        if (s->ndi == 0) return nmatched;
        qsort(s->di, (size_t)s->ndi, sizeof(dist_idx), dist_idx_cmp);
        const float median = (float)s->di[s->ndi / 2].dist;
        const float thDist = 1.5f * 1.4f * median;
        for (int i = s->ndi - 1; i >= 0; i--) {
            if ((float)s->di[i].dist < thDist) break;
            s->u_right[s->di[i].il] = -1;
            s->depth[s->di[i].il] = -1;
            nmatched--;
        }
    }
    return nmatched;
}

int orc_stereo_frame_n(const orc_stereo_frame* s) { return s->n; }
const float* orc_stereo_frame_uright(const orc_stereo_frame* s) { return s->u_right; }
const float* orc_stereo_frame_depth(const orc_stereo_frame* s) { return s->depth; }
const unsigned char* orc_stereo_frame_matched(const orc_stereo_frame* s) { return s->matched; }
int orc_stereo_frame_dist_idx(const orc_stereo_frame* s, int* pairs, int cap)
{
    for (int i = 0; i < s->ndi && i < cap; i++) { pairs[2 * i] = s->di[i].dist; pairs[2 * i + 1] = s->di[i].il; }
    return s->ndi;
}

/* ------------------------------------------------------------------------------------------
 * Frame grid -- Frame.cc:461-476 (AssignFeaturesToGrid), :648-658 (PosInGrid),
 * :593-646 (GetFeaturesInArea).  FRAME_GRID_COLS=64, FRAME_GRID_ROWS=48 (Frame.h:92-93).
 * ---------------------------------------------------------------------------------------- */
#define GRID_COLS 64
#define GRID_ROWS 48

typedef struct {
    int* start; /* [GRID_COLS*GRID_ROWS + 1], cell = ix*GRID_ROWS + iy */
    int* items;
    float inv_w, inv_h;
} grid_t;

/* float -> int as the reference's x86 build converts (cvttss2si): NaN, the infinities and anything beyond +-2^31 give INT_MIN.  C
 * leaves those conversions undefined; stated here so that what a garbage projection does is a definition, not an accident of -O3. */
static int cvt_x86(float f) { return (f >= 2147483648.f || f < -2147483648.f || f != f) ? (-2147483647 - 1) : (int)f; }

static void grid_build(grid_t* g, const orc_keypoint* kp, int n, const orc_frame_bounds* fb)
{
    g->inv_w = (float)GRID_COLS / (fb->max_x - fb->min_x); /* Frame.cc:129-130 */
    g->inv_h = (float)GRID_ROWS / (fb->max_y - fb->min_y);
    const int ncell = GRID_COLS * GRID_ROWS;
    int* cell = (int*)malloc(sizeof(int) * (n > 0 ? n : 1));
    g->start = (int*)calloc((size_t)ncell + 1, sizeof(int));
    for (int i = 0; i < n; i++) {
        const int px = cvt_x86(roundf((kp[i].x - fb->min_x) * g->inv_w));
        const int py = cvt_x86(roundf((kp[i].y - fb->min_y) * g->inv_h));
        if (px < 0 || px >= GRID_COLS || py < 0 || py >= GRID_ROWS) cell[i] = -1;
        else { cell[i] = px * GRID_ROWS + py; g->start[cell[i] + 1]++; }
    }
    for (int c = 0; c < ncell; c++) g->start[c + 1] += g->start[c];
    g->items = (int*)malloc(sizeof(int) * (n > 0 ? n : 1));
    int* fill = (int*)calloc(ncell, sizeof(int));
    for (int i = 0; i < n; i++)
        if (cell[i] >= 0) g->items[g->start[cell[i]] + fill[cell[i]]++] = i;
    free(fill);
    free(cell);
}
static void grid_free(grid_t* g) { free(g->start); free(g->items); }

static int grid_query(const grid_t* g, const orc_keypoint* kp, const orc_frame_bounds* fb,
                      float x, float y, float r, int minLevel, int maxLevel, int* out, int cap)
{
    int n = 0;
    int nMinCellX = cvt_x86(floorf((x - fb->min_x - r) * g->inv_w));
    if (nMinCellX < 0) nMinCellX = 0;
    if (nMinCellX >= GRID_COLS) return 0;
    int nMaxCellX = cvt_x86(ceilf((x - fb->min_x + r) * g->inv_w));
    if (nMaxCellX > GRID_COLS - 1) nMaxCellX = GRID_COLS - 1;
    if (nMaxCellX < 0) return 0;
    int nMinCellY = cvt_x86(floorf((y - fb->min_y - r) * g->inv_h));
    if (nMinCellY < 0) nMinCellY = 0;
    if (nMinCellY >= GRID_ROWS) return 0;
    int nMaxCellY = cvt_x86(ceilf((y - fb->min_y + r) * g->inv_h));
    if (nMaxCellY > GRID_ROWS - 1) nMaxCellY = GRID_ROWS - 1;
    if (nMaxCellY < 0) return 0;
    const int bCheckLevels = (minLevel > 0) || (maxLevel >= 0);
    for (int ix = nMinCellX; ix <= nMaxCellX; ix++)
        for (int iy = nMinCellY; iy <= nMaxCellY; iy++) {
            const int c = ix * GRID_ROWS + iy;
            for (int j = g->start[c]; j < g->start[c + 1]; j++) {
                const int i = g->items[j];
                if (bCheckLevels) {
                    if (kp[i].octave < minLevel) continue;
                    if (maxLevel >= 0 && kp[i].octave > maxLevel) continue;
                }
                const float distx = kp[i].x - x, disty = kp[i].y - y;
                if (fabsf(distx) < r && fabsf(disty) < r) {
                    if (n < cap) out[n] = i;
                    n++;
                }
            }
        }
    return n;
}

int orc_features_in_area(const orc_keypoint* kp_un, int n, const orc_frame_bounds* fb,
                         float x, float y, float r, int min_level, int max_level, int* out_idx, int cap)
{
    grid_t g;
    grid_build(&g, kp_un, n, fb);
    const int k = grid_query(&g, kp_un, fb, x, y, r, min_level, max_level, out_idx, cap);
    grid_free(&g);
    return k;
}

/* ORBmatcher::SearchForInitialization -- ORBmatcher.cc:520-633 (monocular bootstrap).  prev_matched = vbPrevMatched as (x, y) pairs, in and
 * out; fb = F2's image bounds (its grid).  rotHist is kept as the reference keeps it: a list per bin, entries never removed. */
int orc_search_for_initialization(const orc_keypoint* kp1, const uint8_t* desc1, int n1, float* prev_matched, const orc_keypoint* kp2,
                                  const uint8_t* desc2, int n2, const orc_frame_bounds* fb, int window_size, float nn_ratio,
                                  int check_orientation, int* vnMatches12)
{
    enum { HISTO_LENGTH = 30, TH_LOW_ = 50 };
    int nmatches = 0;
    for (int i = 0; i < n1; i++) vnMatches12[i] = -1;
    int* rot_list = (int*)malloc(sizeof(int) * (n1 > 0 ? n1 : 1));     /* (i1, bin) in push order */
    int* rot_bin = (int*)malloc(sizeof(int) * (n1 > 0 ? n1 : 1));
    int nrot = 0;
    int histo[HISTO_LENGTH];
    memset(histo, 0, sizeof histo);
    const float factor = 1.0f / HISTO_LENGTH;
    int* vMatchedDistance = (int*)malloc(sizeof(int) * (n2 > 0 ? n2 : 1));
    int* vnMatches21 = (int*)malloc(sizeof(int) * (n2 > 0 ? n2 : 1));
    for (int i = 0; i < n2; i++) { vMatchedDistance[i] = INT_MAX; vnMatches21[i] = -1; }
    int* vIndices2 = (int*)malloc(sizeof(int) * (n2 > 0 ? n2 : 1));
    grid_t g;
    grid_build(&g, kp2, n2, fb);
    for (int i1 = 0; i1 < n1; i1++) {
        const int level1 = kp1[i1].octave;
        if (level1 > 0) continue;
        const int nv = grid_query(&g, kp2, fb, prev_matched[2 * i1], prev_matched[2 * i1 + 1], (float)window_size, level1, level1, vIndices2, n2);
        if (nv == 0) continue;
        int bestDist = INT_MAX, bestDist2 = INT_MAX, bestIdx2 = -1;
        for (int k = 0; k < nv; k++) {
            const int i2 = vIndices2[k];
            const int dist = orc_hamming256(desc1 + (size_t)i1 * 32, desc2 + (size_t)i2 * 32);
            if (vMatchedDistance[i2] <= dist) continue;
            if (dist < bestDist) { bestDist2 = bestDist; bestDist = dist; bestIdx2 = i2; }
            else if (dist < bestDist2) bestDist2 = dist;
        }
        if (bestDist <= TH_LOW_) {
            if (bestDist < (float)bestDist2 * nn_ratio) {
                if (vnMatches21[bestIdx2] >= 0) {
                    vnMatches12[vnMatches21[bestIdx2]] = -1;
                    nmatches--;
                }
                vnMatches12[i1] = bestIdx2;
                vnMatches21[bestIdx2] = i1;
                vMatchedDistance[bestIdx2] = bestDist;
                nmatches++;
                if (check_orientation) {
                    float rot = kp1[i1].angle - kp2[bestIdx2].angle;
                    if (rot < 0.0) rot += 360.0f;
                    int bin = (int)roundf(rot * factor);
                    if (bin == HISTO_LENGTH) bin = 0;
                    rot_list[nrot] = i1; rot_bin[nrot] = bin; nrot++;
                    histo[bin]++;
                }
            }
        }
    }
    if (check_orientation) {
        int ind1 = -1, ind2 = -1, ind3 = -1;
        orc_three_maxima(histo, HISTO_LENGTH, &ind1, &ind2, &ind3);
        for (int k = 0; k < nrot; k++) {
            if (rot_bin[k] == ind1 || rot_bin[k] == ind2 || rot_bin[k] == ind3) continue;
            const int idx1 = rot_list[k];
            if (vnMatches12[idx1] >= 0) { vnMatches12[idx1] = -1; nmatches--; }
        }
    }
    for (int i1 = 0; i1 < n1; i1++)
        if (vnMatches12[i1] >= 0) {
            prev_matched[2 * i1] = kp2[vnMatches12[i1]].x;
            prev_matched[2 * i1 + 1] = kp2[vnMatches12[i1]].y;
        }
    grid_free(&g);
    free(vIndices2); free(vnMatches21); free(vMatchedDistance); free(rot_bin); free(rot_list);
    return nmatches;
}

/* ORBmatcher::SearchByProjection(Frame&, vector<MapPoint*>&, th) -- ORBmatcher.cc:155-249 */
int orc_search_by_projection(const orc_keypoint* kp, const uint8_t* desc, const float* u_right, int n,
                             const float* sf, int nlevels, const orc_frame_bounds* fb,
                             const orc_map_point* mps, const uint8_t* mp_desc, int m,
                             float th, float nn_ratio, const uint8_t* kp_taken,
                             int* out_mp, int* out_score)
{
    const int TH_HIGH = 100;
    grid_t g;
    grid_build(&g, kp, n, fb);
    int* idx = (int*)malloc(sizeof(int) * (n > 0 ? n : 1));
    uint8_t* blocked = (uint8_t*)malloc(n > 0 ? n : 1);
    for (int i = 0; i < n; i++) {
        blocked[i] = kp_taken ? kp_taken[i] : 0;
        out_mp[i] = -1;
        out_score[i] = 0;
    }
    int nmatches = 0;
    const int bFactor = th != 1.0f;
    for (int iMP = 0; iMP < m; iMP++) {
        const orc_map_point* mp = &mps[iMP];
        if (!(mp->flags & 1)) continue; /* mbTrackInView */
        if (mp->flags & 2) continue;    /* isBad()       */
        const int lvl = mp->level;
        /* The reference indexes F.mvScaleFactors[nPredictedLevel] unchecked (:177-180); MapPoint::PredictScale keeps the level inside the
           table, so it never reads beside it.  A level outside [0, nlevels) is therefore not a case the reference defines: this statement
           skips the point -- what the library does -- instead of reading whatever lies beside sf[] (a fuzz run found the oracle's answer
           changing from run to run on such inputs). */
        if (lvl < 0 || lvl >= nlevels) continue;
        float r = mp->view_cos > 0.998 ? 2.5f : 4.0f; /* RadiusByViewingCos :243-249 (double compare) */
        if (bFactor) r *= th;
        const float rs = r * sf[lvl];
        const int nidx = grid_query(&g, kp, fb, mp->proj_x, mp->proj_y, rs, lvl - 1, lvl, idx, n);
        if (nidx == 0) continue;
        int bestDist = 256, bestLevel = -1, bestDist2 = 256, bestLevel2 = -1, bestIdx = -1;
        for (int k = 0; k < nidx; k++) {
            const int i = idx[k];
            if (blocked[i]) continue;
            if (u_right && u_right[i] > 0) {
                const float er = fabsf(mp->proj_xr - u_right[i]);
                if (er > rs) continue;
            }
            const int dist = orc_hamming256(mp_desc + (size_t)iMP * 32, desc + (size_t)i * 32);
            if (dist < bestDist) {
                bestDist2 = bestDist; bestDist = dist;
                bestLevel2 = bestLevel; bestLevel = kp[i].octave;
                bestIdx = i;
            } else if (dist < bestDist2) {
                bestLevel2 = kp[i].octave;
                bestDist2 = dist;
            }
        }
        if (bestDist <= TH_HIGH) {
            if (bestLevel == bestLevel2 && (float)bestDist > nn_ratio * (float)bestDist2) continue;
            out_mp[bestIdx] = iMP;
            out_score[bestIdx] = bestDist;
            blocked[bestIdx] = (mp->flags & 4) ? 1 : 0; /* later points skip it only if Observations()>0 */
            nmatches++;
        }
    }
    free(idx);
    free(blocked);
    grid_free(&g);
    return nmatches;
}

/* ------------------------------------------------------------------------------------------
 * The good-feature matchers: the loop body above exists three more times in the reference.
 *   ORBmatcher::SearchByProjection_OnePoint(F, pMP, th)            include/ORBmatcher.h:71-150
 *   ORBmatcher::GetCandidates / MatchCandidates                    include/ORBmatcher.h:152-250
 *   ORBmatcher::SearchByProjection_Budget(F, MPs, th, time_constr) src/ORBmatcher.cc:45-153
 * A "projection frame" is the part of Frame these read and write between calls: the grid, mvpMapPoints (as the label of the point a
 * slot holds and whether that point has observations), mvpMatchScore.
 * ---------------------------------------------------------------------------------------- */
struct orc_proj_frame {
    const orc_keypoint* kp; const uint8_t* desc; const float* u_right; int n;
    float sf[32]; int nlevels;
    orc_frame_bounds fb;
    grid_t g;
    int* idx;            /* scratch for GetFeaturesInArea */
    int* slot_mp;        /* F.mvpMapPoints[i]: label of the point, -1 = none of this session's */
    uint8_t* slot_obs;   /* F.mvpMapPoints[i] && F.mvpMapPoints[i]->Observations() > 0 */
    int* slot_score;     /* F.mvpMatchScore[i] */
};

orc_proj_frame* orc_proj_frame_new(const orc_keypoint* kp, const uint8_t* desc, const float* u_right, int n, const float* sf, int nlevels,
                                   const orc_frame_bounds* fb, const uint8_t* kp_taken)
{
    if (nlevels < 1 || nlevels > 32) return NULL;
    orc_proj_frame* f = (orc_proj_frame*)calloc(1, sizeof *f);
    f->kp = kp; f->desc = desc; f->u_right = u_right; f->n = n; f->nlevels = nlevels; f->fb = *fb;
    memcpy(f->sf, sf, sizeof(float) * (size_t)nlevels);
    grid_build(&f->g, kp, n, fb);
    f->idx = (int*)malloc(sizeof(int) * (n > 0 ? n : 1));
    f->slot_mp = (int*)malloc(sizeof(int) * (n > 0 ? n : 1));
    f->slot_obs = (uint8_t*)malloc(n > 0 ? n : 1);
    f->slot_score = (int*)calloc(n > 0 ? n : 1, sizeof(int));
    for (int i = 0; i < n; i++) { f->slot_mp[i] = -1; f->slot_obs[i] = kp_taken ? kp_taken[i] : 0; }
    return f;
}
void orc_proj_frame_free(orc_proj_frame* f)
{
    if (!f) return;
    grid_free(&f->g);
    free(f->idx); free(f->slot_mp); free(f->slot_obs); free(f->slot_score);
    free(f);
}
void orc_proj_frame_get(const orc_proj_frame* f, int* out_mp, int* out_score)
{
    for (int i = 0; i < f->n; i++) { out_mp[i] = f->slot_mp[i]; out_score[i] = f->slot_mp[i] >= 0 ? f->slot_score[i] : 0; }
}

/* the window of a point: RadiusByViewingCos (* th), times the scale factor of the predicted level (ORBmatcher.h:84-90) */
static int proj_point_window(const orc_proj_frame* f, const orc_map_point* mp, float th, float* rs)
{
    if (!(mp->flags & 1)) return 0; /* mbTrackInView, :75 */
    if (mp->flags & 2) return 0;    /* isBad(), :78 */
    const int lvl = mp->level;
    if (lvl < 0 || lvl >= f->nlevels) return 0; /* (see orc_search_by_projection: not a case the reference defines) */
    float r = mp->view_cos > 0.998 ? 2.5f : 4.0f;
    if (th != 1.0f) r *= th;
    *rs = r * f->sf[lvl];
    return 1;
}

/* ORBmatcher::GetCandidates (ORBmatcher.h:152-172): pMP->mvMatchCandidates = F.GetFeaturesInArea(...) */
int orc_proj_frame_candidates(orc_proj_frame* f, const orc_map_point* mp, float th, int* out_idx, int cap)
{
    float rs;
    if (!proj_point_window(f, mp, th, &rs)) return 0;
    const int k = grid_query(&f->g, f->kp, &f->fb, mp->proj_x, mp->proj_y, rs, mp->level - 1, mp->level, f->idx, f->n);
    for (int i = 0; i < k && i < cap; i++) out_idx[i] = f->idx[i];
    return k;
}

/* the candidate loop and the acceptance rule shared by _OnePoint (ORBmatcher.h:101-149) and MatchCandidates (:205-249).
 * why: 0 matched, 1 ratio-rejected, 2 nothing within TH_HIGH */
static int proj_match_list(orc_proj_frame* f, const orc_map_point* mp, const uint8_t* mp_desc, float rs, const int* cand, int ncand,
                           float nn_ratio, int label, int* why)
{
    const int TH_HIGH = 100;
    int bestDist = 256, bestLevel = -1, bestDist2 = 256, bestLevel2 = -1, bestIdx = -1;
    for (int k = 0; k < ncand; k++) {
        const int i = cand[k];
        if (f->slot_obs[i]) continue; /* F.mvpMapPoints[idx] && ->Observations() > 0 */
        if (f->u_right && f->u_right[i] > 0) {
            const float er = fabsf(mp->proj_xr - f->u_right[i]);
            if (er > rs) continue;
        }
        const int dist = orc_hamming256(mp_desc, f->desc + (size_t)i * 32);
        if (dist < bestDist) {
            bestDist2 = bestDist; bestDist = dist;
            bestLevel2 = bestLevel; bestLevel = f->kp[i].octave;
            bestIdx = i;
        } else if (dist < bestDist2) {
            bestLevel2 = f->kp[i].octave;
            bestDist2 = dist;
        }
    }
    if (bestDist <= TH_HIGH) {
        if (bestLevel == bestLevel2 && (float)bestDist > nn_ratio * (float)bestDist2) { if (why) *why = 1; return -1; }
        f->slot_mp[bestIdx] = label;                        /* F.mvpMapPoints[bestIdx] = pMP */
        f->slot_obs[bestIdx] = (mp->flags & 4) ? 1 : 0;
        f->slot_score[bestIdx] = bestDist;                  /* F.mvpMatchScore[bestIdx] = bestDist */
        if (why) *why = 0;
        return bestIdx;
    }
    if (why) *why = 2;
    return -1;
}

/* ORBmatcher::SearchByProjection_OnePoint (ORBmatcher.h:71-150).  *why as above, 3 = left before the candidate loop (:75-99) */
int orc_proj_frame_one_point(orc_proj_frame* f, const orc_map_point* mp, const uint8_t* mp_desc, float th, float nn_ratio, int label, int* why)
{
    float rs;
    if (why) *why = 3;
    if (!proj_point_window(f, mp, th, &rs)) return -1;
    const int k = grid_query(&f->g, f->kp, &f->fb, mp->proj_x, mp->proj_y, rs, mp->level - 1, mp->level, f->idx, f->n);
    if (k == 0) return -1; /* vIndices.empty(), :96 */
    return proj_match_list(f, mp, mp_desc, rs, f->idx, k, nn_ratio, label, why);
}

/* ORBmatcher::MatchCandidates (ORBmatcher.h:176-250) on a list orc_proj_frame_candidates returned earlier */
int orc_proj_frame_match_candidates(orc_proj_frame* f, const orc_map_point* mp, const uint8_t* mp_desc, const int* cand, int ncand,
                                    float th, float nn_ratio, int label)
{
    float rs;
    if (!proj_point_window(f, mp, th, &rs)) return -1;
    if (ncand == 0) return -1; /* :185 */
    return proj_match_list(f, mp, mp_desc, rs, cand, ncand, nn_ratio, label, NULL);
}

/* ORBmatcher::SearchByProjection_Budget (ORBmatcher.cc:45-153).  The wall clock of :96-102 is an argument: `clock_trip` = k > 0 makes
 * the k-th reading of the clock the one that finds the budget spent (0: never).  out_point[p]: what point p did -- keypoint |
 * distance << 16, -1 / -2 / -3 as include/gfo.h's GFO_POINT_*, -4 = the loop had ended before it.  found[p] = IncreaseFound() calls. */
int orc_search_by_projection_budget(const orc_keypoint* kp, const uint8_t* desc, const float* u_right, int n,
                                    const float* sf, int nlevels, const orc_frame_bounds* fb,
                                    const orc_map_point* mps, const uint8_t* mp_desc, int m,
                                    float th, float nn_ratio, const uint8_t* kp_taken, int clock_trip,
                                    int* out_mp, int* out_score, int* out_point, int* found)
{
    orc_proj_frame* f = orc_proj_frame_new(kp, desc, u_right, n, sf, nlevels, fb, kp_taken);
    int nmatches = 0, readings = 0;
    for (int p = 0; p < m; p++) { out_point[p] = -4; if (found) found[p] = 0; }
    for (int iMP = 0; iMP < m; iMP++) {
        int why = 3;
        const int best = orc_proj_frame_one_point(f, &mps[iMP], mp_desc + (size_t)iMP * 32, th, nn_ratio, iMP, &why);
        if (why == 3) { out_point[iMP] = -1; continue; }   /* :60-77 */
        if (why == 1) { out_point[iMP] = -2; continue; }   /* :85-86 */
        if (best >= 0) {
            out_point[iMP] = best | (f->slot_score[best] << 16);
            if (found) found[iMP]++;                       /* pMP->IncreaseFound(), :91 */
            nmatches++;
        } else out_point[iMP] = -3;
        readings++;                                        /* time_Match = timer.toc(), :97 */
        if (clock_trip > 0 && readings >= clock_trip) break;
    }
    orc_proj_frame_get(f, out_mp, out_score);
    orc_proj_frame_free(f);
    return nmatches;
}

/* ------------------------------------------------------------------------------------------
 * ORBmatcher::ComputeThreeMaxima -- ORBmatcher.cc:1723-1764
 * ---------------------------------------------------------------------------------------- */
void orc_three_maxima(const int* histo, int L, int* ind1, int* ind2, int* ind3)
{
    int max1 = 0, max2 = 0, max3 = 0;
    for (int i = 0; i < L; i++) {
        const int s = histo[i];
        if (s > max1) {
            max3 = max2; max2 = max1; max1 = s;
            *ind3 = *ind2; *ind2 = *ind1; *ind1 = i;
        } else if (s > max2) {
            max3 = max2; max2 = s;
            *ind3 = *ind2; *ind2 = i;
        } else if (s > max3) {
            max3 = s;
            *ind3 = i;
        }
    }
    if ((float)max2 < 0.1f * (float)max1) { *ind2 = -1; *ind3 = -1; }
    else if ((float)max3 < 0.1f * (float)max1) *ind3 = -1;
}

/* BUDGETING_FEATURE_MATCHING (ORBmatcher.h:36-37, off in the reference's default build; MAX_NUM_FEATURE_MATCHING 150): two matchers
 * stop adding matches once nmatches reaches the budget -- SearchByBoW breaks out of the CURRENT node's keyframe loop (ORBmatcher.cc:360-365;
 * the next common node starts again and breaks after its first accepted match), SearchByProjection(Cur, Last) out of the whole loop over
 * the last frame's points, BEFORE the match that reached the budget enters the rotation histogram (:1547-1552).  0 = compiled without. */
static int g_feature_budget = 0;
void orc_set_feature_budget(int max_matches) { g_feature_budget = max_matches > 0 ? max_matches : 0; }

/* ------------------------------------------------------------------------------------------
 * ORBmatcher::SearchByBoW(KeyFrame*, Frame&, vpMapPointMatches) -- ORBmatcher.cc:270-404.
 * TH_LOW = 50, HISTO_LENGTH = 30; note factor = 1/HISTO_LENGTH (:284) is applied to a rotation in
 * degrees, as the reference does (bins 0..12 are the only ones that can fill).
 * ---------------------------------------------------------------------------------------- */
int orc_search_by_bow(const uint8_t* kf_desc, const float* kf_angle, const uint8_t* kf_mp_valid, int n_kf,
                      const orc_feature_vector* kfv, const uint8_t* f_desc, const float* f_angle, int n_f,
                      const orc_feature_vector* ffv, float nn_ratio, int check_orientation, int* out)
{
    enum { HISTO_LENGTH = 30, TH_LOW_ = 50 };
    (void)n_kf;
    for (int i = 0; i < n_f; i++) out[i] = -1;
    int nmatches = 0;
    int* rot_bin = (int*)malloc(sizeof(int) * (n_f > 0 ? n_f : 1)); /* bin of the match stored at F index */
    int histo[HISTO_LENGTH];
    memset(histo, 0, sizeof histo);
    for (int i = 0; i < n_f; i++) rot_bin[i] = -1;
    const float factor = 1.0f / HISTO_LENGTH;
    int a = 0, b = 0;
    while (a < kfv->n_nodes && b < ffv->n_nodes) { /* :292-380 */
        if (kfv->node_ids[a] == ffv->node_ids[b]) {
            for (int ik = kfv->node_start[a]; ik < kfv->node_start[a + 1]; ik++) {
                const unsigned realIdxKF = kfv->items[ik];
                if (!kf_mp_valid[realIdxKF]) continue;
                int bestDist1 = 256, bestIdxF = -1, bestDist2 = 256;
                for (int jf = ffv->node_start[b]; jf < ffv->node_start[b + 1]; jf++) {
                    const unsigned realIdxF = ffv->items[jf];
                    if (out[realIdxF] >= 0) continue;
                    const int dist = orc_hamming256(kf_desc + (size_t)realIdxKF * 32, f_desc + (size_t)realIdxF * 32);
                    if (dist < bestDist1) { bestDist2 = bestDist1; bestDist1 = dist; bestIdxF = (int)realIdxF; }
                    else if (dist < bestDist2) bestDist2 = dist;
                }
                if (bestDist1 <= TH_LOW_) {
                    if ((float)bestDist1 < nn_ratio * (float)bestDist2) {
                        out[bestIdxF] = (int)realIdxKF;
                        if (check_orientation) {
                            float rot = kf_angle[realIdxKF] - f_angle[bestIdxF];
                            if (rot < 0.0) rot += 360.0f;
                            int bin = (int)roundf(rot * factor);
                            if (bin == HISTO_LENGTH) bin = 0;
                            rot_bin[bestIdxF] = bin;
                            histo[bin]++;
                        }
                        nmatches++;
                        if (g_feature_budget > 0 && nmatches >= g_feature_budget) break; /* :360-365: leaves this node's keyframe loop only */
                    }
                }
            }
            a++; b++;
        } else if (kfv->node_ids[a] < ffv->node_ids[b]) a++; /* lower_bound on a sorted map */
        else b++;
    }
    if (check_orientation) { /* :382-401 */
        int ind1 = -1, ind2 = -1, ind3 = -1;
        orc_three_maxima(histo, HISTO_LENGTH, &ind1, &ind2, &ind3);
        for (int i = 0; i < n_f; i++) {
            const int bin = rot_bin[i];
            if (bin < 0 || bin == ind1 || bin == ind2 || bin == ind3) continue;
            out[i] = -1;
            nmatches--;
        }
    }
    free(rot_bin);
    return nmatches;
}

/* ORBmatcher::SearchByBoW(KeyFrame* pKF1, KeyFrame* pKF2, vector<MapPoint*>& vpMatches12) -- ORBmatcher.cc:635-768 (loop closing,
 * LoopClosing.cc:287).  Differences from the (KeyFrame, Frame) overload above, all kept: BOTH sides carry a map-point validity mask
 * (:672-676, :690-696), the taken state is a flag per keypoint of the second keyframe (:648, :689), the distance test is strict
 * (`bestDist1 < TH_LOW`, :713), results and rotation histogram are indexed by the FIRST keyframe's keypoint (:717, :730, :757).
 * out12[n1]: keypoint of pKF2 whose map point is left in vpMatches12[i], -1 = NULL. */
int orc_search_by_bow_keyframes(const uint8_t* desc1, const float* angle1, const uint8_t* valid1, int n1, const orc_feature_vector* fv1,
                                const uint8_t* desc2, const float* angle2, const uint8_t* valid2, int n2, const orc_feature_vector* fv2,
                                float nn_ratio, int check_orientation, int* out12)
{
    enum { HISTO_LENGTH = 30, TH_LOW_ = 50 };
    for (int i = 0; i < n1; i++) out12[i] = -1;
    uint8_t* matched2 = (uint8_t*)calloc(n2 > 0 ? n2 : 1, 1);   /* vbMatched2 */
    int* rot_bin = (int*)malloc(sizeof(int) * (n1 > 0 ? n1 : 1));
    for (int i = 0; i < n1; i++) rot_bin[i] = -1;
    int histo[HISTO_LENGTH];
    memset(histo, 0, sizeof histo);
    const float factor = 1.0f / HISTO_LENGTH;
    int nmatches = 0, a = 0, b = 0;
    while (a < fv1->n_nodes && b < fv2->n_nodes) {
        if (fv1->node_ids[a] == fv2->node_ids[b]) {
            for (int i1 = fv1->node_start[a]; i1 < fv1->node_start[a + 1]; i1++) {
                const unsigned idx1 = fv1->items[i1];
                if (!valid1[idx1]) continue;                                   /* !pMP1 || pMP1->isBad() */
                int bestDist1 = 256, bestIdx2 = -1, bestDist2 = 256;
                for (int i2 = fv2->node_start[b]; i2 < fv2->node_start[b + 1]; i2++) {
                    const unsigned idx2 = fv2->items[i2];
                    if (matched2[idx2] || !valid2[idx2]) continue;             /* vbMatched2[idx2] || !pMP2 || pMP2->isBad() */
                    const int dist = orc_hamming256(desc1 + (size_t)idx1 * 32, desc2 + (size_t)idx2 * 32);
                    if (dist < bestDist1) { bestDist2 = bestDist1; bestDist1 = dist; bestIdx2 = (int)idx2; }
                    else if (dist < bestDist2) bestDist2 = dist;
                }
                if (bestDist1 < TH_LOW_) {
                    if ((float)bestDist1 < nn_ratio * (float)bestDist2) {
                        out12[idx1] = bestIdx2;
                        matched2[bestIdx2] = 1;
                        if (check_orientation) {
                            float rot = angle1[idx1] - angle2[bestIdx2];
                            if (rot < 0.0) rot += 360.0f;
                            int bin = (int)roundf(rot * factor);
                            if (bin == HISTO_LENGTH) bin = 0;
                            rot_bin[idx1] = bin;
                            histo[bin]++;
                        }
                        nmatches++;
                    }
                }
            }
            a++; b++;
        } else if (fv1->node_ids[a] < fv2->node_ids[b]) a++;
        else b++;
    }
    if (check_orientation) {
        int ind1 = -1, ind2 = -1, ind3 = -1;
        orc_three_maxima(histo, HISTO_LENGTH, &ind1, &ind2, &ind3);
        for (int i = 0; i < n1; i++) {
            const int bin = rot_bin[i];
            if (bin < 0 || bin == ind1 || bin == ind2 || bin == ind3) continue;
            out12[i] = -1;
            nmatches--;
        }
    }
    free(rot_bin);
    free(matched2);
    return nmatches;
}

/* ORBmatcher::CheckDistEpipolarLine -- ORBmatcher.cc:251-268 (f12 row-major: F12.at<float>(r, c) = f12[3 r + c]) */
static int check_dist_epipolar_line(float x1, float y1, float x2, float y2, const float* f12, float sigma2)
{
    const float a = x1 * f12[0] + y1 * f12[3] + f12[6];
    const float b = x1 * f12[1] + y1 * f12[4] + f12[7];
    const float c = x1 * f12[2] + y1 * f12[5] + f12[8];
    const float num = a * x2 + b * y2 + c;
    const float den = a * a + b * b;
    if (den == 0) return 0;
    const float dsqr = num * num / den;
    return dsqr < 3.84 * sigma2;
}

/* ORBmatcher::SearchForTriangulation -- ORBmatcher.cc:770-935.  has_mp = GetMapPoint(i) != NULL; u_right = mvuRight (may be NULL: monocular);
 * (ex, ey) the epipole the caller computed (:777-783).  vbMatched2 is declared and tested in the reference but never set: kept, unset.
 * out12[n1] = vMatches12. */
int orc_search_for_triangulation(const orc_keypoint* kp1, const uint8_t* desc1, const uint8_t* has_mp1, const float* u_right1, int n1,
                                 const orc_feature_vector* fv1, const orc_keypoint* kp2, const uint8_t* desc2, const uint8_t* has_mp2,
                                 const float* u_right2, int n2, const orc_feature_vector* fv2, const float* scale_factors2, const float* level_sigma2_2,
                                 const float* f12, float ex, float ey, int only_stereo, int check_orientation, int* out12)
{
    enum { HISTO_LENGTH = 30, TH_LOW_ = 50 };
    int nmatches = 0;
    uint8_t* vbMatched2 = (uint8_t*)calloc(n2 > 0 ? n2 : 1, 1);
    for (int i = 0; i < n1; i++) out12[i] = -1;
    int* rot_bin = (int*)malloc(sizeof(int) * (n1 > 0 ? n1 : 1));
    for (int i = 0; i < n1; i++) rot_bin[i] = -1;
    int histo[HISTO_LENGTH];
    memset(histo, 0, sizeof histo);
    const float factor = 1.0f / HISTO_LENGTH;
    int a = 0, b = 0;
    while (a < fv1->n_nodes && b < fv2->n_nodes) {
        if (fv1->node_ids[a] == fv2->node_ids[b]) {
            for (int i1 = fv1->node_start[a]; i1 < fv1->node_start[a + 1]; i1++) {
                const unsigned idx1 = fv1->items[i1];
                if (has_mp1[idx1]) continue;
                const int bStereo1 = u_right1 && u_right1[idx1] >= 0;
                if (only_stereo && !bStereo1) continue;
                int bestDist = TH_LOW_, bestIdx2 = -1;
                for (int i2 = fv2->node_start[b]; i2 < fv2->node_start[b + 1]; i2++) {
                    const unsigned idx2 = fv2->items[i2];
                    if (vbMatched2[idx2] || has_mp2[idx2]) continue;
                    const int bStereo2 = u_right2 && u_right2[idx2] >= 0;
                    if (only_stereo && !bStereo2) continue;
                    const int dist = orc_hamming256(desc1 + (size_t)idx1 * 32, desc2 + (size_t)idx2 * 32);
                    if (dist > TH_LOW_ || dist > bestDist) continue;
                    if (!bStereo1 && !bStereo2) {
                        const float distex = ex - kp2[idx2].x, distey = ey - kp2[idx2].y;
                        if (distex * distex + distey * distey < 100 * scale_factors2[kp2[idx2].octave]) continue;
                    }
                    if (check_dist_epipolar_line(kp1[idx1].x, kp1[idx1].y, kp2[idx2].x, kp2[idx2].y, f12, level_sigma2_2[kp2[idx2].octave])) {
                        bestIdx2 = (int)idx2;
                        bestDist = dist;
                    }
                }
                if (bestIdx2 >= 0) {
                    out12[idx1] = bestIdx2;
                    nmatches++;
                    if (check_orientation) {
                        float rot = kp1[idx1].angle - kp2[bestIdx2].angle;
                        if (rot < 0.0) rot += 360.0f;
                        int bin = (int)roundf(rot * factor);
                        if (bin == HISTO_LENGTH) bin = 0;
                        rot_bin[idx1] = bin;
                        histo[bin]++;
                    }
                }
            }
            a++; b++;
        } else if (fv1->node_ids[a] < fv2->node_ids[b]) a++;
        else b++;
    }
    if (check_orientation) {
        int ind1 = -1, ind2 = -1, ind3 = -1;
        orc_three_maxima(histo, HISTO_LENGTH, &ind1, &ind2, &ind3);
        for (int i = 0; i < n1; i++) {
            const int bin = rot_bin[i];
            if (bin < 0 || bin == ind1 || bin == ind2 || bin == ind3) continue;
            out12[i] = -1;
            nmatches--;
        }
    }
    free(rot_bin);
    free(vbMatched2);
    return nmatches;
}

/* ------------------------------------------------------------------------------------------
 * ORBmatcher::SearchByProjection(Frame& CurrentFrame, const Frame& LastFrame, th, bMono, ...)
 * -- ORBmatcher.cc:1440-1593 -- on pre-projected queries; with use_ratio it is the map-point overload
 * (:155-241) again (used to cross-check the two oracle functions against each other).
 * ---------------------------------------------------------------------------------------- */
/* optional per-QUERY record of the call below (orc_search_by_projection_queries_points): what query iq did at its turn, before any
 * rotation check -- keypoint | distance << 16, -1 inactive / no keypoint in its window, -2 ratio test, -3 nothing usable within th_dist */
static int* g_query_outcome = NULL;
/* ORBmatcher::Fuse(KeyFrame*, MapPoints, th) (ORBmatcher.cc:1019-1051) gates a candidate by its reprojection error instead of the mvuRight
 * window of the tracking overloads: set (orc_search_for_fusion) = the inverse level sigma^2 table of the keyframe */
static const float* g_fuse_inv_sigma2 = NULL;

int orc_search_by_projection_queries(const orc_keypoint* kp, const uint8_t* desc, const float* u_right,
                                     const float* kp_angle, int n, const orc_frame_bounds* fb,
                                     const orc_proj_query* q, const uint8_t* q_desc, int m, const orc_proj_mode* mode,
                                     const uint8_t* kp_taken, int* out_q, int* out_score)
{
    enum { HISTO_LENGTH = 30 };
    grid_t g;
    grid_build(&g, kp, n, fb);
    int* idx = (int*)malloc(sizeof(int) * (n > 0 ? n : 1));
    uint8_t* blocked = (uint8_t*)malloc(n > 0 ? n : 1);
    int* hist_kp = (int*)malloc(sizeof(int) * (m > 0 ? m : 1));  /* rotHist entries: keypoint index ... */
    int* hist_bin = (int*)malloc(sizeof(int) * (m > 0 ? m : 1)); /* ... and its bin, in push order      */
    int nh = 0, histo[HISTO_LENGTH];
    memset(histo, 0, sizeof histo);
    for (int i = 0; i < n; i++) {
        blocked[i] = kp_taken ? kp_taken[i] : 0;
        out_q[i] = -1;
        out_score[i] = 0;
    }
    const float factor = 1.0f / HISTO_LENGTH;
    int nmatches = 0;
    if (g_query_outcome) for (int iq = 0; iq < m; iq++) g_query_outcome[iq] = -1;
    for (int iq = 0; iq < m; iq++) {
        const orc_proj_query* p = &q[iq];
        if (!(p->flags & 1)) continue;
        const int nidx = grid_query(&g, kp, fb, p->u, p->v, p->radius, p->min_level, p->max_level, idx, n);
        if (nidx == 0) continue;
        if (g_query_outcome) g_query_outcome[iq] = -3;
        int bestDist = 256, bestLevel = -1, bestDist2 = 256, bestLevel2 = -1, bestIdx = -1;
        for (int k = 0; k < nidx; k++) {
            const int i = idx[k];
            if (blocked[i]) continue;
            if (g_fuse_inv_sigma2) {                       /* :1026-1050 */
                const float kpx = kp[i].x, kpy = kp[i].y;
                const float ex = p->u - kpx, ey = p->v - kpy;
                if (u_right && u_right[i] >= 0) {
                    const float kpr = u_right[i];
                    const float er = p->ur - kpr;
                    const float e2 = ex * ex + ey * ey + er * er;
                    if (e2 * g_fuse_inv_sigma2[kp[i].octave] > 7.8) continue;
                } else {
                    const float e2 = ex * ex + ey * ey;
                    if (e2 * g_fuse_inv_sigma2[kp[i].octave] > 5.99) continue;
                }
            } else if (u_right && u_right[i] > 0) {
                const float er = fabsf(p->ur - u_right[i]);
                if (er > p->radius) continue;
            }
            const int dist = orc_hamming256(q_desc + (size_t)iq * 32, desc + (size_t)i * 32);
            if (dist < bestDist) {
                bestDist2 = bestDist; bestDist = dist;
                bestLevel2 = bestLevel; bestLevel = kp[i].octave;
                bestIdx = i;
            } else if (dist < bestDist2) {
                bestLevel2 = kp[i].octave;
                bestDist2 = dist;
            }
        }
        if (bestIdx >= 0 && bestDist <= mode->th_dist) { /* th_dist < 256 in every caller of the reference: no candidate leaves bestDist = 256 */
            if (mode->use_ratio && bestLevel == bestLevel2 && (float)bestDist > mode->nn_ratio * (float)bestDist2) {
                if (g_query_outcome) g_query_outcome[iq] = -2;
                continue;
            }
            if (g_query_outcome) g_query_outcome[iq] = bestIdx | (bestDist << 16);
            out_q[bestIdx] = iq;
            out_score[bestIdx] = bestDist;
            blocked[bestIdx] = (p->flags & 4) ? 1 : 0;
            nmatches++;
            if (g_feature_budget > 0 && nmatches >= g_feature_budget) break; /* :1547-1552: before the rotation histogram */
            if (mode->check_orientation) {
                float rot = p->angle - kp_angle[bestIdx];
                if (rot < 0.0) rot += 360.0f;
                int bin = (int)roundf(rot * factor);
                if (bin == HISTO_LENGTH) bin = 0;
                hist_kp[nh] = bestIdx;
                hist_bin[nh] = bin;
                nh++;
                histo[bin]++;
            }
        }
    }
    if (mode->check_orientation) {
        int ind1 = -1, ind2 = -1, ind3 = -1;
        orc_three_maxima(histo, HISTO_LENGTH, &ind1, &ind2, &ind3);
        for (int e = 0; e < nh; e++) {
            const int b = hist_bin[e];
            if (b != ind1 && b != ind2 && b != ind3) {
                out_q[hist_kp[e]] = -2; /* mvpMapPoints[..] = NULL: written by this call, then cleared (:1586) */
                nmatches--;
            }
        }
    }
    free(idx); free(blocked); free(hist_kp); free(hist_bin);
    grid_free(&g);
    return nmatches;
}

/* ORBmatcher::Fuse(KeyFrame* pKF, const vector<MapPoint*>&, th), its search (ORBmatcher.cc:1000-1063) on pre-projected points: the best
 * keypoint of levels [l - 1, l] in the window whose reprojection error passes the chi-square test, accepted up to th_dist (TH_LOW); no
 * point hides a keypoint from another (flags bit 2 is ignored).  out_point[m]: keypoint | distance << 16, or negative. */
int orc_search_for_fusion(const orc_keypoint* kp, const uint8_t* desc, const float* u_right, int n, const orc_frame_bounds* fb,
                          const float* inv_level_sigma2, const orc_proj_query* q, const uint8_t* q_desc, int m, int th_dist, int* out_point)
{
    orc_proj_query* qq = (orc_proj_query*)malloc(sizeof(orc_proj_query) * (m > 0 ? m : 1));
    for (int i = 0; i < m; i++) { qq[i] = q[i]; qq[i].flags &= ~4; }
    int* out_q = (int*)malloc(sizeof(int) * (n > 0 ? n : 1));
    int* out_s = (int*)malloc(sizeof(int) * (n > 0 ? n : 1));
    orc_proj_mode mode = {0, 0.f, th_dist, 0};
    g_fuse_inv_sigma2 = inv_level_sigma2;
    g_query_outcome = out_point;
    orc_search_by_projection_queries(kp, desc, u_right, NULL, n, fb, qq, q_desc, m, &mode, NULL, out_q, out_s);
    g_query_outcome = NULL;
    g_fuse_inv_sigma2 = NULL;
    free(qq); free(out_q); free(out_s);
    return 0;
}

/* ... the same call, also reporting what every query did at its turn (a query that blocks nothing -- flags bit 2 clear -- and finds the
 * frame's slots as they were on entry is ONE independent best-match search: the form ORBmatcher::Fuse and SearchBySim3 use) */
int orc_search_by_projection_queries_points(const orc_keypoint* kp, const uint8_t* desc, const float* u_right,
                                            const float* kp_angle, int n, const orc_frame_bounds* fb,
                                            const orc_proj_query* q, const uint8_t* q_desc, int m, const orc_proj_mode* mode,
                                            const uint8_t* kp_taken, int* out_q, int* out_score, int* out_point)
{
    g_query_outcome = out_point;
    const int nm = orc_search_by_projection_queries(kp, desc, u_right, kp_angle, n, fb, q, q_desc, m, mode, kp_taken, out_q, out_score);
    g_query_outcome = NULL;
    return nm;
}

/* ------------------------------------------------------------------------------------------
 * ORBmatcher::SearchByProjection(Frame& CurrentFrame, KeyFrame* pKF, const set<MapPoint*>& sAlreadyFound,
 * th, ORBdist) -- ORBmatcher.cc:1595-1721 -- literally, on pre-projected map points (the projection, the distance
 * test and PredictScale of :1617-1650 are the caller's): q[i].flags bit0 = "reaches the window search".
 * Unlike the two overloads above, a keypoint with ANY map point is skipped (:1664-1665) and there is no mvuRight
 * gate.  kp_set[i] = 1 where CurrentFrame.mvpMapPoints[i] != NULL on entry.  out_q as above (-2 = cleared).
 * ---------------------------------------------------------------------------------------- */
int orc_search_by_projection_kf(const orc_keypoint* kp, const uint8_t* desc, const float* kp_angle, int n,
                                const orc_frame_bounds* fb, const orc_proj_query* q, const uint8_t* q_desc, int m,
                                int orb_dist, int check_orientation, const uint8_t* kp_set, int* out_q, int* out_score)
{
    enum { HISTO_LENGTH = 30 };
    grid_t g;
    grid_build(&g, kp, n, fb);
    int* idx = (int*)malloc(sizeof(int) * (n > 0 ? n : 1));
    uint8_t* set = (uint8_t*)malloc(n > 0 ? n : 1);
    int* hist_kp = (int*)malloc(sizeof(int) * (m > 0 ? m : 1));
    int* hist_bin = (int*)malloc(sizeof(int) * (m > 0 ? m : 1));
    int nh = 0, histo[HISTO_LENGTH];
    memset(histo, 0, sizeof histo);
    for (int i = 0; i < n; i++) {
        set[i] = kp_set ? kp_set[i] : 0;
        out_q[i] = -1;
        out_score[i] = 0;
    }
    const float factor = 1.0f / HISTO_LENGTH;
    int nmatches = 0;
    for (int iq = 0; iq < m; iq++) {
        const orc_proj_query* p = &q[iq];
        if (!(p->flags & 1)) continue;
        const int nidx = grid_query(&g, kp, fb, p->u, p->v, p->radius, p->min_level, p->max_level, idx, n);
        if (nidx == 0) continue;
        int bestDist = 256, bestIdx2 = -1;
        for (int k = 0; k < nidx; k++) {
            const int i2 = idx[k];
            if (set[i2]) continue; /* :1664 */
            const int dist = orc_hamming256(q_desc + (size_t)iq * 32, desc + (size_t)i2 * 32);
            if (dist < bestDist) { bestDist = dist; bestIdx2 = i2; }
        }
        if (bestIdx2 >= 0 && bestDist <= orb_dist) { /* :1679; ORBdist < 256 in every caller */
            set[bestIdx2] = 1;
            out_q[bestIdx2] = iq;
            out_score[bestIdx2] = bestDist;
            nmatches++;
            if (check_orientation) {
                float rot = p->angle - kp_angle[bestIdx2];
                if (rot < 0.0) rot += 360.0f;
                int bin = (int)roundf(rot * factor);
                if (bin == HISTO_LENGTH) bin = 0;
                hist_kp[nh] = bestIdx2;
                hist_bin[nh] = bin;
                nh++;
                histo[bin]++;
            }
        }
    }
    if (check_orientation) {
        int ind1 = -1, ind2 = -1, ind3 = -1;
        orc_three_maxima(histo, HISTO_LENGTH, &ind1, &ind2, &ind3);
        for (int e = 0; e < nh; e++) {
            const int b = hist_bin[e];
            if (b != ind1 && b != ind2 && b != ind3) {
                out_q[hist_kp[e]] = -2;
                nmatches--;
            }
        }
    }
    free(idx); free(set); free(hist_kp); free(hist_bin);
    grid_free(&g);
    return nmatches;
}

/* ------------------------------------------------------------------------------------------
 * DBoW2 TemplatedVocabulary<FORB>::transform(feature, word_id, weight, nid, levelsup)
 * -- Thirdparty/DBoW2/DBoW2/TemplatedVocabulary.h:1231-1272, distance = FORB::distance (FORB.cpp:81)
 * ---------------------------------------------------------------------------------------- */
static void bow_descend(const orc_vocabulary* voc, const uint8_t* d, int levelsup, int* word, double* weight, int* node)
{
    const int nid_level = voc->depth - levelsup;
    int nid = 0; /* root when nid_level <= 0 */
    int final_id = 0, current_level = 0;
    do {
        ++current_level;
        const int fc = voc->first_child[final_id], nc = voc->n_children[final_id];
        final_id = fc;
        int best_d = orc_hamming256(d, voc->descriptors + (size_t)fc * 32);
        for (int c = 1; c < nc; c++) {
            const int dd = orc_hamming256(d, voc->descriptors + (size_t)(fc + c) * 32);
            if (dd < best_d) { best_d = dd; final_id = fc + c; }
        }
        if (current_level == nid_level) nid = final_id;
    } while (voc->n_children[final_id] > 0);
    *word = voc->word_id[final_id];
    *weight = voc->weight64 ? voc->weight64[final_id] : (double)voc->weight[final_id];
    *node = nid;
}

void orc_bow_transform(const orc_vocabulary* voc, const uint8_t* desc, int n, int levelsup,
                       int32_t* word_id, float* weight, int32_t* node_id)
{
    for (int i = 0; i < n; i++) {
        int w, nd;
        double wt;
        bow_descend(voc, desc + (size_t)i * 32, levelsup, &w, &wt, &nd);
        word_id[i] = w;
        weight[i] = (float)wt;
        node_id[i] = nd;
    }
}

/* The fold half of TemplatedVocabulary::transform(features, BowVector&, FeatureVector&, levelsup) --
 * TemplatedVocabulary.h:1161-1212 -- on the per-feature (word, weight, node) stream the descent produced: BowVector::addWeight /
 * addIfNotExist / normalize (BowVector.cpp:34-84) and FeatureVector::addFeature (FeatureVector.cpp:31-46), the two std::maps kept
 * as sorted arrays.  This is the one piece of the path with a reference-compiled pin: oracle/_ref/libdbow2_fold.so is the
 * reference's own BowVector.cpp / FeatureVector.cpp, and tests/test_oracle.py compares the two bit for bit. */
int orc_bow_fold(const uint32_t* word, const double* weight, const uint32_t* node, int n, int weighting, int norm,
                 uint32_t* bow_words, double* bow_values, uint32_t* fv_node_ids, int32_t* fv_start, uint32_t* fv_items,
                 int* n_fv_nodes)
{
    /* fv: per node a growing list (kept as (node, feature) pairs, stable) */
    int nw = 0, npairs = 0;
    uint32_t* pn = (uint32_t*)malloc(sizeof(uint32_t) * (n > 0 ? n : 1));
    uint32_t* pf = (uint32_t*)malloc(sizeof(uint32_t) * (n > 0 ? n : 1));
    const int tf = weighting == 0 || weighting == 1;
    for (int i = 0; i < n; i++) { /* :1161-1176 / :1187-1202 */
        const uint32_t id = word[i];
        const double w = weight[i];
        if (!(w > 0)) continue; /* stopped */
        /* lower_bound in the BowVector */
        int lo = 0, hi = nw;
        while (lo < hi) { const int mid = (lo + hi) / 2; if (bow_words[mid] < id) lo = mid + 1; else hi = mid; }
        if (lo < nw && bow_words[lo] == id) {
            if (tf) bow_values[lo] += w; /* addWeight; addIfNotExist leaves the first value */
        } else {
            memmove(bow_words + lo + 1, bow_words + lo, sizeof(uint32_t) * (size_t)(nw - lo));
            memmove(bow_values + lo + 1, bow_values + lo, sizeof(double) * (size_t)(nw - lo));
            bow_words[lo] = id;
            bow_values[lo] = w;
            nw++;
        }
        pn[npairs] = node[i]; /* fv.addFeature(nid, i_feature) */
        pf[npairs] = (uint32_t)i;
        npairs++;
    }
    if (tf && nw > 0 && norm == 0) { /* :1177-1183 */
        const double nd = (double)nw;
        for (int k = 0; k < nw; k++) bow_values[k] /= nd;
    }
    if (norm != 0) { /* BowVector::normalize */
        double nrm = 0.0;
        if (norm == 1) for (int k = 0; k < nw; k++) nrm += fabs(bow_values[k]);
        else {
            for (int k = 0; k < nw; k++) nrm += bow_values[k] * bow_values[k];
            nrm = sqrt(nrm);
        }
        if (nrm > 0.0)
            for (int k = 0; k < nw; k++) bow_values[k] /= nrm;
    }
    /* FeatureVector in map order: nodes ascending, features in insertion (= ascending) order */
    int nseg = 0, pos = 0;
    uint8_t* used = (uint8_t*)calloc(npairs > 0 ? npairs : 1, 1);
    for (;;) {
        uint32_t best = 0;
        int found = 0;
        for (int k = 0; k < npairs; k++)
            if (!used[k] && (!found || pn[k] < best)) { best = pn[k]; found = 1; }
        if (!found) break;
        fv_node_ids[nseg] = best;
        fv_start[nseg] = pos;
        for (int k = 0; k < npairs; k++)
            if (!used[k] && pn[k] == best) { fv_items[pos++] = pf[k]; used[k] = 1; }
        nseg++;
    }
    fv_start[nseg] = pos;
    *n_fv_nodes = nseg;
    free(pn); free(pf); free(used);
    return nw;
}

/* the per-feature stream of the descent with the weight as DBoW2 holds it (WordValue = double): the input of orc_bow_fold */
void orc_bow_stream(const orc_vocabulary* voc, const uint8_t* desc, int n, int levelsup, uint32_t* word, double* weight, uint32_t* node)
{
    for (int i = 0; i < n; i++) {
        int id, nid;
        double w;
        bow_descend(voc, desc + (size_t)i * 32, levelsup, &id, &w, &nid);
        word[i] = (uint32_t)id; weight[i] = w; node[i] = (uint32_t)nid;
    }
}

int orc_compute_bow(const orc_vocabulary* voc, const uint8_t* desc, int n, int levelsup, int weighting, int norm,
                    uint32_t* bow_words, double* bow_values, uint32_t* fv_node_ids, int32_t* fv_start, uint32_t* fv_items,
                    int* n_fv_nodes)
{
    const size_t m = (size_t)(n > 0 ? n : 1);
    uint32_t* word = (uint32_t*)malloc(sizeof(uint32_t) * m);
    uint32_t* node = (uint32_t*)malloc(sizeof(uint32_t) * m);
    double* weight = (double*)malloc(sizeof(double) * m);
    orc_bow_stream(voc, desc, n, levelsup, word, weight, node);
    const int nw = orc_bow_fold(word, weight, node, n, weighting, norm, bow_words, bow_values, fv_node_ids, fv_start, fv_items, n_fv_nodes);
    free(word); free(node); free(weight);
    return nw;
}

/* ------------------------------------------------------------------------------------------
 * Frame::ComputeStereoMatches -- Frame.cc:889-1078: best Hamming match along the row band, then an
 * 11x11 SAD search over 11 horizontal shifts on the keypoint's pyramid level and a parabola fit.
 * ---------------------------------------------------------------------------------------- */
int orc_stereo_match_sad(const orc_extractor* el, const orc_extractor* er,
                         const orc_keypoint* kl, const uint8_t* dl, int nl,
                         const orc_keypoint* kr, const uint8_t* dr, int nr,
                         float mbf, float mb, float* u_right, float* depth, int* best_dist_out)
{
    const int TH_HIGH = 100, TH_LOW = 50;
    const int thOrbDist = (TH_HIGH + TH_LOW) / 2;
    const int nRows = el->h[0];
    int* start = (int*)calloc((size_t)nRows + 1, sizeof(int));
    int* minr_a = (int*)malloc(sizeof(int) * (nr > 0 ? nr : 1));
    int* maxr_a = (int*)malloc(sizeof(int) * (nr > 0 ? nr : 1));
    for (int iR = 0; iR < nr; iR++) { /* :905-917 */
        const float kpY = kr[iR].y;
        const float r = 2.0f * el->scale[kr[iR].octave];
        int maxr = (int)ceilf(kpY + r), minr = (int)floorf(kpY - r);
        if (minr < 0) minr = 0;
        if (maxr > nRows - 1) maxr = nRows - 1;
        minr_a[iR] = minr; maxr_a[iR] = maxr;
        for (int yi = minr; yi <= maxr; yi++) start[yi + 1]++;
    }
    for (int r = 0; r < nRows; r++) start[r + 1] += start[r];
    int* items = (int*)malloc(sizeof(int) * (start[nRows] > 0 ? start[nRows] : 1));
    int* fill = (int*)calloc((size_t)nRows, sizeof(int));
    for (int iR = 0; iR < nr; iR++)
        for (int yi = minr_a[iR]; yi <= maxr_a[iR]; yi++) items[start[yi] + fill[yi]++] = iR;
    free(fill); free(minr_a); free(maxr_a);
    const float minZ = mb, minD = 0, maxD = mbf / minZ;
    dist_idx* di = (dist_idx*)malloc(sizeof(dist_idx) * (nl > 0 ? nl : 1));
    int ndi = 0;
    for (int iL = 0; iL < nl; iL++) { u_right[iL] = -1.0f; depth[iL] = -1.0f; if (best_dist_out) best_dist_out[iL] = -1; }
    for (int iL = 0; iL < nl; iL++) {
        const int levelL = kl[iL].octave;
        const float vL = kl[iL].y, uL = kl[iL].x;
        int row = (int)vL;
        if (row < 0 || row > nRows - 1) continue; /* the reference would index out of bounds */
        const int nC = start[row + 1] - start[row];
        if (nC == 0) continue;
        const float minU = uL - maxD, maxU = uL - minD;
        if (maxU < 0) continue;
        int bestDist = TH_HIGH, bestIdxR = 0;
        for (int iC = 0; iC < nC; iC++) {
            const int iR = items[start[row] + iC];
            if (kr[iR].octave < levelL - 1 || kr[iR].octave > levelL + 1) continue;
            const float uR = kr[iR].x;
            if (uR >= minU && uR <= maxU) {
                const int dist = orc_hamming256(dl + (size_t)iL * 32, dr + (size_t)iR * 32);
                if (dist < bestDist) { bestDist = dist; bestIdxR = iR; }
            }
        }
        if (bestDist < thOrbDist) { /* :973-1057 */
            const float uR0 = kr[bestIdxR].x;
            const float scaleFactor = el->inv_scale[levelL];
            const float scaleduL = roundf(uL * scaleFactor);
            const float scaledvL = roundf(vL * scaleFactor);
            const float scaleduR0 = roundf(uR0 * scaleFactor);
            const int w = 5, L = 5;
            const uint8_t* IL = el->level[levelL];
            const uint8_t* IR = er->level[levelL];
            const int lw = el->w[levelL], rw = er->w[levelL];
            const int cu = (int)scaleduL, cv = (int)scaledvL, cr = (int)scaleduR0;
            const float iniu = scaleduR0 + L - w, endu = scaleduR0 + L + w + 1;
            if (iniu < 0 || endu >= (float)rw) continue;
            int sadBest = 2147483647, bestincR = 0;
            float vDists[11];
            const float ilc = (float)IL[(size_t)cv * lw + cu];
            for (int incR = -L; incR <= L; incR++) {
                const float irc = (float)IR[(size_t)cv * rw + cr + incR];
                double acc = 0; /* cv::norm(NORM_L1) on CV_32F accumulates in double */
                for (int dy = -w; dy <= w; dy++)
                    for (int dx = -w; dx <= w; dx++) {
                        const float a = (float)IL[(size_t)(cv + dy) * lw + cu + dx] - ilc;
                        const float b = (float)IR[(size_t)(cv + dy) * rw + cr + incR + dx] - irc;
                        acc += fabs((double)(a - b));
                    }
                const float dist = (float)acc;
                if (dist < (float)sadBest) { sadBest = (int)dist; bestincR = incR; }
                vDists[L + incR] = dist;
            }
            if (bestincR == -L || bestincR == L) continue;
            const float dist1 = vDists[L + bestincR - 1], dist2 = vDists[L + bestincR], dist3 = vDists[L + bestincR + 1];
            const float deltaR = (dist1 - dist3) / (2.0f * (dist1 + dist3 - 2.0f * dist2));
            if (deltaR < -1 || deltaR > 1) continue;
            float bestuR = el->scale[levelL] * ((float)scaleduR0 + (float)bestincR + deltaR);
            float disparity = uL - bestuR;
            if (disparity >= minD && disparity < maxD) {
                if (disparity <= 0) { disparity = 0.01f; bestuR = (float)((double)uL - 0.01); /* double arithmetic, then narrowed: Frame.cc:1054,1278 */ }
                depth[iL] = mbf / disparity;
                u_right[iL] = bestuR;
                if (best_dist_out) best_dist_out[iL] = sadBest;
                di[ndi].dist = sadBest; di[ndi].il = iL; ndi++;
            }
        }
    }
    int kept = ndi;
    if (ndi > 0) { /* :1059-1076 (the reference reads vDistIdx[0] even when empty) */
        qsort(di, ndi, sizeof(dist_idx), dist_idx_cmp);
        const float median = (float)di[ndi / 2].dist;
        const float thDist = 1.5f * 1.4f * median;
        for (int i = ndi - 1; i >= 0; i--) {
            if ((float)di[i].dist < thDist) break;
            u_right[di[i].il] = -1;
            depth[di[i].il] = -1;
            kept--;
        }
    }
    free(di); free(start); free(items);
    return kept;
}
