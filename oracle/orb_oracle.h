/* orb_oracle.h -- CPU restatement of the GF-ORB-SLAM2 ORB front-end (TEST INFRASTRUCTURE).
 *
 * This is the parity oracle for the MI355X path.  Only tests/, __graft_entry__.smoke()
 * and bench.py's cpu_baseline leg may load it; the product (libgfo.so) never does.
 *
 * PARITY STATUS: "parity unpinned" at the OpenCV boundary.  The reference's CPU extractor
 * calls OpenCV 3.4.1 (cv::FAST, cv::resize, cv::GaussianBlur, cv::fastAtan2, cvRound), which
 * is neither vendored in /root/reference nor installed in this image, and the reference's
 * tests hold no golden keypoints/descriptors.  Everything that lives
 * in the reference's own sources (cell grid, quadtree, angle, descriptor, matchers) follows
 * the cited lines; the OpenCV pieces restate the published 3.4.x algorithms from memory.
 * The only known answers this oracle is pinned against are the level geometry / quotas /
 * umax table of SURVEY.md 8 and the popcount definition of DescriptorDistance.
 */
#ifndef ORB_ORACLE_H
#define ORB_ORACLE_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* layout-identical to cv::KeyPoint (7 x 4 bytes) */
typedef struct {
    float x, y, size, angle, response;
    int octave, class_id;
} orc_keypoint;

typedef struct orc_extractor orc_extractor;

/* variant switches (default 0 everywhere = the variant the golden vectors use) */
enum {
    ORC_TRIG_SHARED = 0, /* correctly rounded sin/cos (long double, rounded once); the kernels' include/gfo_sincos.h agrees with it */
    ORC_TRIG_LIBM = 1    /* host libm cosf/sinf, what the reference literally calls      */
};
enum {
    ORC_ROT_UNFUSED = 0, /* x*b + y*a as two roundings of the products then one add      */
    ORC_ROT_FMA = 1      /* fmaf(x, b, y*a): what gcc -O3 -march=native may contract to  */
};

/* [OCV] variant table (process-wide; orb_oracle.c documents each value).  tests/golden/check_against_cv2.py compares
 * every stage with cv2 3.4.x where that exists and names the switch to flip; oracle/ocv_variants.json holds the values
 * the Python binding applies at load. */
enum { ORC_OCV_RESIZE = 0, ORC_OCV_ATAN_FMA = 1, ORC_OCV_BLUR_ROUND = 2, ORC_OCV_COUNT = 3 };
int orc_set_ocv_variant(int key, int value);
int orc_get_ocv_variant(int key);
void orc_set_gauss_taps(const int* taps7);
void orc_get_gauss_taps(int* taps7);

orc_extractor* orc_create(int nfeatures, float scale_factor, int nlevels, int ini_th, int min_th);
void orc_destroy(orc_extractor* e);
void orc_set_variant(orc_extractor* e, int trig_mode, int rot_mode);

/* tables (ORBextractor.cc:409-469) */
int orc_nlevels(const orc_extractor* e);
const float* orc_scale_factors(const orc_extractor* e);
const float* orc_inv_scale_factors(const orc_extractor* e);
const float* orc_level_sigma2(const orc_extractor* e);
const float* orc_inv_level_sigma2(const orc_extractor* e);
const int* orc_features_per_level(const orc_extractor* e);
const int* orc_umax(const orc_extractor* e);

/* ORBextractor::operator() (ORBextractor.cc:1112-1174).  Returns the number of keypoints
 * produced (may exceed nfeatures, see DistributeOctTree); writes min(n, cap) of them. */
int orc_extract(orc_extractor* e, const uint8_t* img, int w, int h, int stride,
                orc_keypoint* kp, uint8_t* desc, int cap);

/* ComputePyramid only (ORBextractor.cc:1176-1201) */
void orc_compute_pyramid(orc_extractor* e, const uint8_t* img, int w, int h, int stride);

/* inspection of the last extraction / pyramid */
int orc_level_size(const orc_extractor* e, int level, int* w, int* h);
void orc_get_level(const orc_extractor* e, int level, uint8_t* out, int out_stride);
/* level with the 19-px BORDER_REFLECT_101 frame the reference keeps around it */
void orc_get_level_padded(const orc_extractor* e, int level, uint8_t* out, int out_stride);
void orc_get_blurred_level(const orc_extractor* e, int level, uint8_t* out, int out_stride);
/* FAST candidates handed to DistributeOctTree for one level, in the reference's order
 * (cell-major); coordinates relative to minBorder.  xys = {x, y, score} triplets. */
int orc_level_candidates(const orc_extractor* e, int level, int* xys, int cap);
int orc_level_keypoint_count(const orc_extractor* e, int level);

/* building blocks exposed for unit tests */
void orc_resize_linear_u8(const uint8_t* src, int sw, int sh, int sstride,
                          uint8_t* dst, int dw, int dh, int dstride);
void orc_gaussian_blur7_u8(const uint8_t* src, int w, int h, int sstride, uint8_t* dst, int dstride);
/* cv::FAST(img, kps, threshold, true) on a w x h sub-image; out = {x, y, score} triplets */
int orc_fast9_nms(const uint8_t* img, int w, int h, int stride, int threshold, int* xys, int cap);
float orc_fast_atan2(float y, float x);
void orc_sincos(float t, float* s, float* c);
void orc_fast_atan2_n(const float* y, const float* x, int n, float* out);
void orc_sincos_n(const float* t, int n, float* s, float* c);
int orc_cv_round(float v);

/* ORBmatcher::DescriptorDistance (ORBmatcher.cc:1768-1784) */
int orc_hamming256(const uint8_t* a, const uint8_t* b);

/* Frame::PrepareStereoCandidates + ComputeStereoMatches_Undistorted(false)
 * (Frame.h:230-263, Frame.cc:1167-1316) with every mvpMapPoints[] NULL unless min_d/max_d
 * are given per left keypoint (the adapter's flattening of Frame.cc:1220-1231).
 * Outputs: u_right[nl], depth[nl] (-1 where unmatched), best_dist[nl] (-1 where no
 * accepted match), best_idx_r[nl].  Returns nmatched as the reference counts it. */
typedef struct {
    int n_rows;    /* mvImagePyramid[0].rows                       */
    float mbf, mb; /* baseline*fx and baseline                       */
    float min_x;   /* mnMinX                                         */
} orc_stereo_params;
/* BUDGETING_FEATURE_MATCHING (ORBmatcher.h:36-37): max_matches > 0 makes orc_search_by_bow and orc_search_by_projection_queries stop
 * as ORBmatcher.cc:360-365 / :1547-1552 do; 0 (default) = compiled without it */
void orc_set_feature_budget(int max_matches);
/* 1: the call is ComputeStereoMatches_Undistorted(true): no outlier cut (Frame.cc:1290 `if (!isOnline)`); 0 (default): (false) */
void orc_set_stereo_online(int on);
int orc_stereo_match(const orc_keypoint* kl, const uint8_t* dl, int nl,
                     const orc_keypoint* kr, const uint8_t* dr, int nr,
                     const float* scale_factors, const orc_stereo_params* p,
                     const float* min_d, const float* max_d,
                     float* u_right, float* depth, int* best_dist, int* best_idx_r);

/* The stereo members of one Frame ACROSS calls (mvuRight, mvDepth, mvStereoMatched, mvDistIdx, mvRowIndices): state is reset only
 * by PrepareStereoCandidates, which ComputeStereoMatches_Undistorted runs only when mvRowIndices.size() != nRows (Frame.cc:1173-1176);
 * a second call on the same frame (Tracking.cc:941-954) keeps what it does not overwrite and cuts over the accumulated mvDistIdx.
 * delayed != 0: the member as compiled with DELAYED_STEREO_MATCHING (Frame.cc:1186-1199).  See orb_oracle.c. */
typedef struct orc_stereo_frame orc_stereo_frame;
orc_stereo_frame* orc_stereo_frame_new(void);
void orc_stereo_frame_free(orc_stereo_frame* s);
void orc_stereo_frame_prepare(orc_stereo_frame* s, int nl, const orc_keypoint* kr, int nr, const float* scale_factors, int n_rows);
void orc_stereo_frame_clear_matched(orc_stereo_frame* s);
int orc_stereo_frame_match(orc_stereo_frame* s, const orc_keypoint* kl, const uint8_t* dl, int nl,
                           const orc_keypoint* kr, const uint8_t* dr, int nr, const float* scale_factors,
                           const orc_stereo_params* p, const float* min_d, const float* max_d,
                           const unsigned char* has_mp, int is_online, int delayed);
int orc_stereo_frame_n(const orc_stereo_frame* s);
const float* orc_stereo_frame_uright(const orc_stereo_frame* s);
const float* orc_stereo_frame_depth(const orc_stereo_frame* s);
const unsigned char* orc_stereo_frame_matched(const orc_stereo_frame* s);
int orc_stereo_frame_dist_idx(const orc_stereo_frame* s, int* pairs, int cap);   /* returns mvDistIdx.size(); pairs = (dist, iL) */

/* Frame::ComputeStereoMatches (the SAD sub-pixel variant, Frame.cc:889-1078; compiled out by
 * ALTER_STEREO_MATCHING in the reference's default build).  el / er hold the pyramids of the left and right
 * image (orc_extract / orc_compute_pyramid was run on them); kl/kr are mvKeys / mvKeysRight.
 * Rows outside the image are clamped in the row table (the reference indexes it unchecked).
 * best_dist[nl] = the SAD value pushed into vDistIdx (-1 if none).  Returns the number of matches kept. */
int orc_stereo_match_sad(const orc_extractor* el, const orc_extractor* er,
                         const orc_keypoint* kl, const uint8_t* dl, int nl,
                         const orc_keypoint* kr, const uint8_t* dr, int nr,
                         float mbf, float mb, float* u_right, float* depth, int* best_dist);

/* Frame grid (Frame.cc:461-476, 593-658) + ORBmatcher::SearchByProjection(Frame&, MapPoints, th)
 * (ORBmatcher.cc:155-249) on flattened arrays. */
typedef struct {
    float min_x, min_y, max_x, max_y; /* mnMinX.. image bounds */
} orc_frame_bounds;
typedef struct {
    float proj_x, proj_y, proj_xr; /* mTrackProjX/Y/XR   */
    float view_cos;                /* mTrackViewCos      */
    int level;                     /* mnTrackScaleLevel  */
    int flags;                     /* bit0 mbTrackInView, bit1 isBad(), bit2 Observations()>0 */
} orc_map_point;
/* kp_taken[n]: 1 where F.mvpMapPoints[i] is set with Observations()>0 on entry.
 * out_mp[n]: index of the map point finally stored in F.mvpMapPoints[i] by this call (-1 none);
 * out_score[n]: F.mvpMatchScore[i] for those.  Returns nmatches. */
int orc_search_by_projection(const orc_keypoint* kp_un, const uint8_t* desc, const float* u_right, int n,
                             const float* scale_factors, int nlevels, const orc_frame_bounds* fb,
                             const orc_map_point* mps, const uint8_t* mp_desc, int m,
                             float th, float nn_ratio, const uint8_t* kp_taken,
                             int* out_mp, int* out_score);   /* a map point whose level is outside [0, nlevels) is skipped (see the .c) */
/* The good-feature matchers (see the .c): SearchByProjection_OnePoint / GetCandidates / MatchCandidates (include/ORBmatcher.h:71-250)
 * on a "projection frame" that carries mvpMapPoints / mvpMatchScore between calls, and SearchByProjection_Budget
 * (src/ORBmatcher.cc:45-153) with its wall clock as an argument. */
typedef struct orc_proj_frame orc_proj_frame;
orc_proj_frame* orc_proj_frame_new(const orc_keypoint* kp_un, const uint8_t* desc, const float* u_right, int n, const float* scale_factors,
                                   int nlevels, const orc_frame_bounds* fb, const uint8_t* kp_taken);   /* the arrays must outlive the frame */
void orc_proj_frame_free(orc_proj_frame* f);
void orc_proj_frame_get(const orc_proj_frame* f, int* out_mp, int* out_score);
int orc_proj_frame_candidates(orc_proj_frame* f, const orc_map_point* mp, float th, int* out_idx, int cap);
int orc_proj_frame_one_point(orc_proj_frame* f, const orc_map_point* mp, const uint8_t* mp_desc32, float th, float nn_ratio, int label, int* why);
int orc_proj_frame_match_candidates(orc_proj_frame* f, const orc_map_point* mp, const uint8_t* mp_desc32, const int* cand, int ncand,
                                    float th, float nn_ratio, int label);
int orc_search_by_projection_budget(const orc_keypoint* kp_un, const uint8_t* desc, const float* u_right, int n,
                                    const float* scale_factors, int nlevels, const orc_frame_bounds* fb,
                                    const orc_map_point* mps, const uint8_t* mp_desc, int m,
                                    float th, float nn_ratio, const uint8_t* kp_taken, int clock_trip,
                                    int* out_mp, int* out_score, int* out_point, int* found);
/* ORBmatcher::SearchForInitialization -- ORBmatcher.cc:520-633; prev_matched (x, y pairs) in and out */
int orc_search_for_initialization(const orc_keypoint* kp1, const uint8_t* desc1, int n1, float* prev_matched, const orc_keypoint* kp2,
                                  const uint8_t* desc2, int n2, const orc_frame_bounds* fb, int window_size, float nn_ratio,
                                  int check_orientation, int* vnMatches12);
int orc_features_in_area(const orc_keypoint* kp_un, int n, const orc_frame_bounds* fb,
                         float x, float y, float r, int min_level, int max_level,
                         int* out_idx, int cap);

/* ORBmatcher::SearchByProjection(Frame& Cur, const Frame& Last, th, bMono, ...) -- ORBmatcher.cc:1440-1593 --
 * and the map-point overload, on explicit queries (one per projected map point, visited in array order).
 * The adapter-side projection (:1467-1497: u, v, invzc, ur = u - mbf*invzc, radius = th*scale[octave], the
 * forward/backward level window) is done by the caller.  out_q[n] = query left in mvpMapPoints[i] (-1 NULL). */
typedef struct {
    float u, v, ur, radius;
    int32_t min_level, max_level;
    float angle;
    int32_t flags; /* bit0 active, bit2 Observations() > 0 */
} orc_proj_query;
typedef struct {
    int32_t use_ratio;
    float nn_ratio;
    int32_t th_dist;
    int32_t check_orientation;
} orc_proj_mode;
int orc_search_by_projection_queries(const orc_keypoint* kp_un, const uint8_t* desc, const float* u_right,
                                     const float* kp_angle, int n, const orc_frame_bounds* fb,
                                     const orc_proj_query* q, const uint8_t* q_desc, int m, const orc_proj_mode* mode,
                                     const uint8_t* kp_taken, int* out_q, int* out_score);
/* the same, with what every query did at its turn: keypoint | distance << 16, -1 / -2 / -3 as gfo.h's GFO_POINT_* (before any rotation check) */
int orc_search_by_projection_queries_points(const orc_keypoint* kp, const uint8_t* desc, const float* u_right,
                                            const float* kp_angle, int n, const orc_frame_bounds* fb,
                                            const orc_proj_query* q, const uint8_t* q_desc, int m, const orc_proj_mode* mode,
                                            const uint8_t* kp_taken, int* out_q, int* out_score, int* out_point);
/* the search of ORBmatcher::Fuse(KeyFrame*, MapPoints, th) (ORBmatcher.cc:1000-1063) on pre-projected points; out_point as above */
int orc_search_for_fusion(const orc_keypoint* kp, const uint8_t* desc, const float* u_right, int n, const orc_frame_bounds* fb,
                          const float* inv_level_sigma2, const orc_proj_query* q, const uint8_t* q_desc, int m, int th_dist, int* out_point);

/* ORBmatcher::SearchByProjection(Frame& Cur, KeyFrame*, const set<MapPoint*>& sAlreadyFound, th, ORBdist) --
 * ORBmatcher.cc:1595-1721 on pre-projected map points: any set keypoint is skipped, no mvuRight gate, no ratio.
 * kp_set[i] = 1 where Cur.mvpMapPoints[i] != NULL on entry. */
int orc_search_by_projection_kf(const orc_keypoint* kp_un, const uint8_t* desc, const float* kp_angle, int n,
                                const orc_frame_bounds* fb, const orc_proj_query* q, const uint8_t* q_desc, int m,
                                int orb_dist, int check_orientation, const uint8_t* kp_set, int* out_q, int* out_score);

/* ORBmatcher::SearchByBoW(KeyFrame*, Frame&, vector<MapPoint*>&) -- ORBmatcher.cc:270-404, with the
 * two DBoW2::FeatureVector maps flattened to CSR (node ids ascending, as std::map iterates them).
 * kf_mp_valid[i] = 1 where vpMapPointsKF[i] is set and not bad.  out_kf_idx[n_f]: index of the KF
 * keypoint whose map point ends up in vpMapPointMatches[i], -1 = NULL.  Returns nmatches. */
typedef struct {
    const uint32_t* node_ids;   /* [n_nodes] ascending                       */
    const int32_t* node_start;  /* [n_nodes + 1] offsets into items           */
    const uint32_t* items;      /* keypoint indices, in the vector's order    */
    int32_t n_nodes;
} orc_feature_vector;
int orc_search_by_bow(const uint8_t* kf_desc, const float* kf_angle, const uint8_t* kf_mp_valid, int n_kf,
                      const orc_feature_vector* kf_fv, const uint8_t* f_desc, const float* f_angle, int n_f,
                      const orc_feature_vector* f_fv, float nn_ratio, int check_orientation, int* out_kf_idx);
/* ORBmatcher::SearchByBoW(KeyFrame*, KeyFrame*, vector<MapPoint*>&) -- ORBmatcher.cc:635-768; out12[n1] = keypoint of the second keyframe, -1 */
int orc_search_by_bow_keyframes(const uint8_t* desc1, const float* angle1, const uint8_t* valid1, int n1, const orc_feature_vector* fv1,
                                const uint8_t* desc2, const float* angle2, const uint8_t* valid2, int n2, const orc_feature_vector* fv2,
                                float nn_ratio, int check_orientation, int* out12);
/* ORBmatcher::SearchForTriangulation -- ORBmatcher.cc:770-935 (+ CheckDistEpipolarLine :251-268); out12[n1] = vMatches12 */
int orc_search_for_triangulation(const orc_keypoint* kp1, const uint8_t* desc1, const uint8_t* has_mp1, const float* u_right1, int n1,
                                 const orc_feature_vector* fv1, const orc_keypoint* kp2, const uint8_t* desc2, const uint8_t* has_mp2,
                                 const float* u_right2, int n2, const orc_feature_vector* fv2, const float* scale_factors2, const float* level_sigma2_2,
                                 const float* f12, float ex, float ey, int only_stereo, int check_orientation, int* out12);
/* ORBmatcher::ComputeThreeMaxima -- ORBmatcher.cc:1723-1764 on the bin sizes */
void orc_three_maxima(const int* histo_sizes, int L, int* ind1, int* ind2, int* ind3);

/* DBoW2 TemplatedVocabulary<FORB>::transform, per-feature descent (TemplatedVocabulary.h:1231-1272) on a
 * flattened tree (node 0 = root, children contiguous in the vocabulary's order, leaf: n_children == 0). */
typedef struct {
    const int32_t* first_child;
    const int32_t* n_children;
    const uint8_t* descriptors;
    const int32_t* word_id;
    const float* weight;
    int32_t n_nodes, depth;
    const double* weight64; /* optional: Node::weight as DBoW2 holds it (WordValue = double); NULL = weight[] */
} orc_vocabulary;
void orc_bow_transform(const orc_vocabulary* voc, const uint8_t* desc, int n, int levelsup,
                       int32_t* word_id, float* weight, int32_t* node_id);

/* TemplatedVocabulary::transform(features, BowVector&, FeatureVector&, levelsup) -- TemplatedVocabulary.h:1140-1212 --
 * with BowVector::addWeight / addIfNotExist / normalize (BowVector.cpp:34-84) and FeatureVector::addFeature, literally:
 * the two std::maps are kept as sorted arrays.  weighting: 0 TF_IDF, 1 TF, 2 IDF, 3 BINARY; norm: 0 none, 1 L1, 2 L2.
 * Outputs in map order; bow_* / fv_node_ids / fv_items take n entries, fv_start n + 1.  Returns the word count. */
int orc_compute_bow(const orc_vocabulary* voc, const uint8_t* desc, int n, int levelsup, int weighting, int norm,
                    uint32_t* bow_words, double* bow_values, uint32_t* fv_node_ids, int32_t* fv_start, uint32_t* fv_items,
                    int* n_fv_nodes);
/* the two halves of orc_compute_bow: the per-feature (word, double weight, node) stream of the descent, and the fold of such a
 * stream into the two maps.  The fold is what oracle/_ref/libdbow2_fold.so (the reference's own BowVector.cpp / FeatureVector.cpp
 * behind the same signature, ref_bow_fold) pins. */
void orc_bow_stream(const orc_vocabulary* voc, const uint8_t* desc, int n, int levelsup, uint32_t* word, double* weight, uint32_t* node);
int orc_bow_fold(const uint32_t* word, const double* weight, const uint32_t* node, int n, int weighting, int norm,
                 uint32_t* bow_words, double* bow_values, uint32_t* fv_node_ids, int32_t* fv_start, uint32_t* fv_items,
                 int* n_fv_nodes);

#ifdef __cplusplus
}
#endif
#endif
