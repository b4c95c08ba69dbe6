/* gfo.h -- C ABI of the MI355X-native ORB front-end (libgfo.so).
 *
 * Drop-in boundary for the ORB extract + match path of GF-ORB-SLAM2.  Plain pointers and
 * sizes only; no C++/torch/OpenCV types cross this boundary.  Every entry point names the
 * reference interface it replaces (paths relative to the reference repository).
 *
 * Conventions
 *   - every function returning int returns GFO_OK (0) or a negative gfo_status; nothing
 *     throws or aborts across the ABI.  gfo_last_error(ctx) gives the message.
 *   - one context = one GPU + one HIP stream + one pre-sized HBM arena; contexts are
 *     independent (two contexts may be driven from two host threads, as the reference does
 *     with its left/right extractors, Frame.cc:84-87); a single context is not re-entrant.
 *   - there is NO CPU fallback: without a usable gfx950 device gfo_ctx_create fails.
 */
#ifndef GFO_H
#define GFO_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define GFO_VERSION 100 /* 0.1.0 */
#define GFO_MAX_LEVELS 16

typedef enum {
    GFO_OK = 0,
    GFO_ERR_INVALID = -1,   /* bad argument                                   */
    GFO_ERR_DEVICE = -2,    /* HIP error / no usable device                   */
    GFO_ERR_CAPACITY = -3,  /* caller buffer or batch capacity too small      */
    GFO_ERR_OVERFLOW = -4,  /* an internal fixed-capacity buffer overflowed   */
    GFO_ERR_STATE = -5      /* call order violated (e.g. no batch extracted)  */
} gfo_status;

/* Layout-identical to cv::KeyPoint (pt.x, pt.y, size, angle, response, octave, class_id):
 * the adapter can reinterpret a std::vector<cv::KeyPoint>'s storage.  28 bytes. */
typedef struct {
    float x, y, size, angle, response;
    int32_t octave, class_id;
} gfo_keypoint;

/* ORBextractor::ORBextractor(nfeatures, scaleFactor, nlevels, iniThFAST, minThFAST)
 * include/ORBextractor.h:81-82, src/ORBextractor.cc:409-469.
 * Limits of this implementation (refused with GFO_ERR_INVALID and a message when an image is planned):
 * images up to 4000 px a side (12-bit coordinate packing), FAST cells up to 64 px, at most 60000 quadtree nodes
 * per level, thresholds 1..254.  A level that takes more than 2040 features (with the usual 8 levels at scale
 * 1.2: nfeatures above ~9300) still runs, but its quadtree tables no longer fit the 160 KB of LDS and live in
 * HBM -- markedly slower. */
typedef struct {
    int32_t nfeatures;
    float scale_factor;
    int32_t nlevels;
    int32_t ini_th_fast;
    int32_t min_th_fast;
    int32_t max_batch; /* images per device batch the arena is sized for (>=1) */
} gfo_params;

typedef struct gfo_ctx gfo_ctx;

int gfo_version(void);
/* The [OCV] arithmetic variant this library was COMPILED with (`make EXTRA=-DGFO_OCV_RESIZE=1 ...`, csrc/gfo_internal.h): the four
 * pieces of OpenCV 3.4.1 that ORBextractor.cc:102,1155,1189 call and this library restates -- each a switch with the same values as
 * oracle/ocv_variants.json, so that a run of tests/golden/check_against_cv2.py on a machine with cv2 3.4.x that names a switch is
 * answered by a rebuild, not a rewrite.  key 0: resize (0 = 11-bit fixed point, 1 = float bilinear rounded once), 1: fastAtan2's
 * polynomial (0 = separate multiplies and adds, 1 = fused), 2: GaussianBlur rounding (0 = once, (v + 2^15) >> 16; 1 = each pass to
 * 8 bits), 3..9: the seven Gaussian taps (scale 256).  Returns -1 for an unknown key.  The default build is all zeros and
 * {18,34,49,55,49,34,18}. */
int gfo_build_variant(int key);

/* Creates a context on HIP device `device`.  Arena memory is allocated lazily on the first
 * image (its size fixes the level geometry; a different size re-plans the arena). */
int gfo_ctx_create(const gfo_params* params, int device, gfo_ctx** out);
void gfo_ctx_destroy(gfo_ctx* ctx);
const char* gfo_last_error(const gfo_ctx* ctx); /* ctx may be NULL: last create error */

/* Use an externally owned HIP stream (hipStream_t passed as void*); NULL restores the
 * context's own stream.  Lets the host framework time/order work on its current stream. */
int gfo_ctx_set_stream(gfo_ctx* ctx, void* hip_stream);
int gfo_ctx_synchronize(gfo_ctx* ctx);

/* Pipelining aid for applications that alternate batches between several contexts (no counterpart in the reference,
 * which extracts one frame at a time): after gfo_ctx_chain(ctx, after, stage) every extraction submitted to `ctx` starts
 * on the device only once the extraction most recently submitted to `after` -- at the time of the call -- has finished
 * `stage`.  It fixes the phase between the two contexts' kernel chains, which otherwise settles at random after every
 * synchronisation (two contexts, stereo 752x480: 214 k or 225 k frames/s per run; chained after the pyramid: 227 k every
 * run).  Purely an ordering edge between work that is already submitted: it cannot deadlock, changes no result, and a
 * context that has not run yet imposes nothing.  after = NULL removes the edge; destroying `after` removes it too.
 * One stage per `after` context (the last call wins).  Set the chain up before the contexts are used from other threads. */
enum { GFO_STAGE_PYRAMID = 1, GFO_STAGE_FAST = 2, GFO_STAGE_SELECT = 3, GFO_STAGE_DESCRIPTORS = 4 };
int gfo_ctx_chain(gfo_ctx* ctx, gfo_ctx* after, int stage);

/* Frame combiner.  The reference extracts one frame per call, and a process that tracks K cameras does so from K threads
 * (Frame::Frame, src/Frame.cc:84-100).  With combining on, the per-frame host entry points of this context --
 * gfo_extract (one image) and gfo_extract_stereo -- may execute inside ONE device batch together with the frames other
 * threads submit at the same time through other combining contexts of equal parameters, device and image size: the
 * callers' concurrency becomes batch size (every stage of the library is one launch over all images of a batch) instead
 * of competing streams.  Each call still blocks until ITS results are in the caller's arrays and returns exactly what
 * the direct call returns, bit for bit; one caller alone runs a batch of one.  What changes: the device-side state of
 * the frame (pyramid, batch views, debug hooks) lives in the shared engine, not in this context -- gfo_pyramid_level,
 * gfo_batch_* and gfo_stereo_match_batch after a combined call return GFO_ERR_STATE.  Off by default; the drop-in
 * adapter turns it on (unless GFO_FULL_PYRAMID asks for the levels).  GFO_COMBINE_MAX (default 32) bounds the images of a
 * batch.  gfo_combiner_stats: device batches and requests the context's engine has served so far. */
int gfo_ctx_set_combining(gfo_ctx* ctx, int on);
int gfo_combiner_stats(const gfo_ctx* ctx, int64_t* batches, int64_t* requests);

/* Getters of include/ORBextractor.h:93-119 (GetLevels, GetScaleFactor, GetScaleFactors,
 * GetInverseScaleFactors, GetScaleSigmaSquares, GetInverseScaleSigmaSquares) and the
 * per-level quotas mnFeaturesPerLevel (ORBextractor.cc:435-445).  Each out array takes
 * nlevels entries; any may be NULL. */
int gfo_ctx_tables(const gfo_ctx* ctx, float* scale, float* inv_scale, float* sigma2,
                   float* inv_sigma2, int32_t* features_per_level);
/* Upper bound on keypoints one image can yield (DistributeOctTree may exceed the quota by a
 * few nodes per level, ORBextractor.cc:667-737); size caller buffers with this. */
int gfo_ctx_max_keypoints(const gfo_ctx* ctx);

/* ORBextractor::operator()(image, mask, keypoints, descriptors)
 * include/ORBextractor.h:89-91, src/ORBextractor.cc:1112-1174.  8-bit grey host image,
 * `stride` bytes per row.  Writes min(n, cap) keypoints / 32-byte descriptors, *n = count
 * produced; returns GFO_ERR_CAPACITY if n > cap (outputs truncated).  Empty image
 * (w or h <= 0 or img NULL): returns GFO_OK with *n = 0 and outputs untouched (:1115). */
int gfo_extract(gfo_ctx* ctx, const uint8_t* img, int w, int h, int stride,
                gfo_keypoint* kp, uint8_t* desc, int cap, int* n);

/* Optional: pin the buffers frames are handed over in.  The per-frame entry points (gfo_extract, gfo_extract_stereo,
 * small gfo_extract_batch) copy every image into pinned staging before the DMA engine can take it (~35 us of host memcpy
 * per 752x480 stereo frame); an image that lies entirely inside a range registered here, with stride == width, goes to
 * the device straight from the caller's memory instead (left and right image contiguous in one buffer: one copy).
 * gfo_host_register page-locks [p, p + bytes) (hipHostRegister); the range must stay valid until gfo_host_unregister.
 * Results are unaffected.  Callers that cannot pin anything -- cv::Mat from imread, ROS messages -- simply do not call it. */
int gfo_host_register(void* p, size_t bytes);
int gfo_host_unregister(void* p);

/* Batched form: `nimg` host images of identical size (one arena pass, every stage one
 * launch over all images).  imgs[i] points at image i.  kp/desc are [nimg][cap] arrays. */
int gfo_extract_batch(gfo_ctx* ctx, const uint8_t* const* imgs, int nimg, int w, int h, int stride,
                      gfo_keypoint* kp, uint8_t* desc, int cap, int* n);

/* Device-resident form: `d_imgs` is a device pointer to nimg images, `pitch` bytes per row,
 * `img_stride` bytes between images (>= pitch * h).  Results stay in the arena; fetch with
 * gfo_batch_counts / gfo_batch_fetch or chain gfo_stereo_match_batch / gfo_search_by_projection_batch.
 * Asynchronous on the context stream.
 * Lifetime: level 0 of the pyramid IS the caller's buffer (it is never copied), so d_imgs must stay valid and
 * unchanged until the next gfo_extract* / gfo_compute_pyramid call on this context or gfo_ctx_destroy --
 * gfo_pyramid_level(level 0) and gfo_stereo_match_sad_batch read it after this call has returned. */
int gfo_extract_batch_device(gfo_ctx* ctx, const uint8_t* d_imgs, int nimg, int w, int h,
                             size_t pitch, size_t img_stride);
int gfo_batch_counts(gfo_ctx* ctx, int* n /*[nimg]*/, int* per_level /*[nimg][nlevels] or NULL*/);
int gfo_batch_fetch(gfo_ctx* ctx, int image, gfo_keypoint* kp, uint8_t* desc, int cap, int* n);
/* device addresses of the batch results (for chaining device-side consumers):
 * keypoints [nimg][kp_stride] gfo_keypoint, descriptors [nimg][kp_stride][32], counts [nimg] */
int gfo_batch_device_views(gfo_ctx* ctx, const gfo_keypoint** d_kp, const uint8_t** d_desc,
                           const int32_t** d_counts, int* kp_stride);

/* Delivery of a device-resident batch to the host: ORBextractor::operator() ends with its results in host containers
 * (src/ORBextractor.cc:1137-1173: _descriptors.create, the keypoint vector) and the stereo Frame constructor with mvuRight /
 * mvDepth (src/Frame.cc:1167-1316).  gfo_batch_deliver queues the copy of EVERYTHING the last batch produced -- overflow
 * flags, counts, keypoints, descriptors and, after gfo_stereo_match_batch, the four stereo arrays and nmatched -- into ONE
 * pinned host block `host_dst` laid out as `layout` says, on a copy stream of the context: the call returns at once, the
 * copies run behind the batch's kernels and beside whatever is submitted next (the next extraction on this context only
 * waits for them before it overwrites the outputs), gfo_deliver_wait blocks until the block is complete.
 * host_dst = NULL only fills `layout` (bytes = size to allocate: hipHostMalloc / a pinned torch tensor).  Arrays keep the
 * arena's strides: keypoints [nimg][kp_stride], descriptors [nimg][kp_stride][32], stereo arrays [nimg/2][kp_stride]. */
typedef struct {
    int32_t nimg, kp_stride, stereo;
    size_t off_flags, off_counts, off_kp, off_desc, off_u_right, off_depth, off_best_dist, off_best_idx, off_nmatched, bytes;
} gfo_delivery;
int gfo_batch_deliver(gfo_ctx* ctx, void* host_dst, size_t host_bytes, gfo_delivery* layout);
int gfo_deliver_wait(gfo_ctx* ctx);

/* ORBextractor::ComputePyramid + public member mvImagePyramid (include/ORBextractor.h:127-132,
 * src/ORBextractor.cc:1176-1201).  Copies level `level` of image `image` of the last batch
 * to host.  border = 0: the w_l x h_l level; border = 19: with the BORDER_REFLECT_101 frame
 * the reference keeps (EDGE_THRESHOLD).  *w,*h receive the level size without border. */
int gfo_compute_pyramid(gfo_ctx* ctx, const uint8_t* img, int w, int h, int stride);
int gfo_pyramid_level(gfo_ctx* ctx, int image, int level, int border, uint8_t* out, int out_stride,
                      int* w, int* h);

/* ORBmatcher::DescriptorDistance(a, b) -- include/ORBmatcher.h:49, src/ORBmatcher.cc:1768-1784.
 * Host helper (popcount of the 256-bit XOR); the device matchers use the same definition. */
int gfo_hamming256(const void* a, const void* b);

/* Frame::PrepareStereoCandidates + Frame::ComputeStereoMatches_Undistorted(false)
 * include/Frame.h:230-263, src/Frame.cc:1167-1316 (ALTER_STEREO_MATCHING path). */
typedef struct {
    int32_t n_rows; /* mpORBextractorLeft->mvImagePyramid[0].rows */
    float mbf;      /* Frame::mbf                                 */
    float mb;       /* Frame::mb  (minZ; maxD = mbf/mb)           */
    float min_x;    /* Frame::mnMinX                              */
} gfo_stereo_params;
/* Host-array form used by the Frame adapter.  kl/kr are mvKeysUn / mvKeysRightUn, dl/dr the
 * descriptor rows.  min_d/max_d (each nl floats) may be NULL (= 0 and mbf/mb); they carry the
 * per-keypoint disparity window of Frame.cc:1220-1231.  Outputs: u_right[nl] = mvuRight,
 * depth[nl] = mvDepth (-1 where unmatched), best_dist[nl]/best_idx_r[nl] = the accepted match
 * before the median cut (-1 if none; rebuilds mvDistIdx).  *nmatched as the reference counts. */
int gfo_stereo_match(gfo_ctx* ctx, const gfo_keypoint* kl, const uint8_t* dl, int nl,
                     const gfo_keypoint* kr, const uint8_t* dr, int nr,
                     const float* scale_factors, int nlevels, const gfo_stereo_params* p,
                     const float* min_d, const float* max_d,
                     float* u_right, float* depth, int32_t* best_dist, int32_t* best_idx_r,
                     int* nmatched);
/* Device-chained form: images (2k, 2k+1) of the last batch are the left/right views of pair k.
 * Results stay on the device; fetch with gfo_stereo_fetch. */
int gfo_stereo_match_batch(gfo_ctx* ctx, const gfo_stereo_params* p);
/* One stereo frame in ONE submission: the body of the reference's stereo Frame constructor --
 * ExtractORB(0, imLeft) and ExtractORB(1, imRight) on two threads (src/Frame.cc:84-87, 478-484), then
 * ComputeStereoMatches_Undistorted (:100, 1167-1316) -- as one H2D copy of both images, one replay of the captured
 * launch sequence (both extractions + the association), one D2H burst and one synchronisation.  The extracted
 * keypoints stand for mvKeysUn / mvKeysRightUn (rectified input, as the reference's stereo examples provide).
 * kp/desc buffers take `cap` entries per image; u_right/depth/best_dist/best_idx_r take cap entries (left keypoints). */
int gfo_extract_stereo(gfo_ctx* ctx, const uint8_t* img_l, const uint8_t* img_r, int w, int h, int stride,
                       const gfo_stereo_params* p, gfo_keypoint* kp_l, uint8_t* desc_l, gfo_keypoint* kp_r,
                       uint8_t* desc_r, int cap, int* n_l, int* n_r, float* u_right, float* depth,
                       int32_t* best_dist, int32_t* best_idx_r, int* nmatched);

/* Frame::ComputeStereoMatches (src/Frame.cc:889-1078): the SAD sub-pixel variant the reference compiles out
 * with ALTER_STEREO_MATCHING (include/Frame.h:38).  It reads pyramid pixels of BOTH cameras, so it exists in the
 * batched form only: images (2k, 2k+1) of the last batch, keypoints as extracted (mvKeys / mvKeysRight).
 * gfo_stereo_fetch then returns mvuRight, mvDepth, best_dist = the SAD value kept in vDistIdx, and
 * *nmatched = matches surviving the median cut. */
int gfo_stereo_match_sad_batch(gfo_ctx* ctx, float mbf, float mb);
int gfo_stereo_fetch(gfo_ctx* ctx, int pair, float* u_right, float* depth, int32_t* best_dist,
                     int32_t* best_idx_r, int cap, int* nmatched);

/* Frame::AssignFeaturesToGrid / GetFeaturesInArea (src/Frame.cc:461-476, 593-658) +
 * ORBmatcher::SearchByProjection(Frame&, const vector<MapPoint*>&, th)
 * include/ORBmatcher.h:64, src/ORBmatcher.cc:155-249, on flattened arrays. */
typedef struct {
    float min_x, min_y, max_x, max_y; /* Frame::mnMinX, mnMinY, mnMaxX, mnMaxY */
} gfo_frame_bounds;
typedef struct {
    float proj_x, proj_y, proj_xr; /* MapPoint::mTrackProjX / Y / XR  (MapPoint.h:107-115) */
    float view_cos;                /* MapPoint::mTrackViewCos                              */
    int32_t level;                 /* MapPoint::mnTrackScaleLevel                          */
    int32_t flags;                 /* bit0 mbTrackInView, bit1 isBad(), bit2 Observations()>0 */
} gfo_map_point;
/* kp_un = F.mvKeysUn, desc = F.mDescriptors, u_right = F.mvuRight (may be NULL),
 * kp_taken[n] = 1 where F.mvpMapPoints[i] is set with Observations()>0 on entry (may be NULL).
 * out_mp[n]: index into mps of the map point left in F.mvpMapPoints[i] by the call, -1 = none;
 * out_score[n] = F.mvpMatchScore[i] for those.  *nmatches = return value of the reference. */
int gfo_search_by_projection(gfo_ctx* ctx, const gfo_keypoint* kp_un, const uint8_t* desc,
                             const float* u_right, int n, const float* scale_factors, int nlevels,
                             const gfo_frame_bounds* bounds, const gfo_map_point* mps,
                             const uint8_t* mp_desc, int m, float th, float nn_ratio,
                             const uint8_t* kp_taken, int32_t* out_mp, int32_t* out_score,
                             int* nmatches);

/* Generalised projection search on explicit queries: the common core of
 *   ORBmatcher::SearchByProjection(Frame& F, const vector<MapPoint*>&, th)                 ORBmatcher.cc:155-241
 *   ORBmatcher::SearchByProjection(Frame& Cur, const Frame& Last, th, bMono, nVisible)     ORBmatcher.cc:1440-1593
 * A query is one projected map point: where it lands (u, v, ur), how far to look (radius, the
 * GetFeaturesInArea level window), its descriptor, and whether taking a keypoint blocks later queries
 * (Observations() > 0).  Queries are processed in array order with the reference's serial semantics.
 * Mode: use_ratio/nn_ratio = best-vs-second test of :228-231 (same-level rule included); th_dist = TH_HIGH
 * (or ORBdist); check_orientation = the 30-bin rotation histogram of :1548-1591 between query.angle and
 * kp_angle[] (ComputeThreeMaxima :1723-1764): a keypoint is cleared if ANY accepted query that took it falls
 * in a discarded bin, and *nmatches drops once per such query, as the reference's loop does.
 * out_q[i]: the query whose map point the call leaves in mvpMapPoints[i]; -1: the call did not touch the slot;
 * -2: the call matched the slot and its rotation check then cleared it (the reference stores NULL there, :1586). */
typedef struct {
    float u, v, ur;                /* projection; ur is compared with u_right[] when that is > 0        */
    float radius;                  /* window half-size = r passed to GetFeaturesInArea, and the ur gate */
    int32_t min_level, max_level;  /* level window of GetFeaturesInArea (Frame.cc:593), -1 = open       */
    float angle;                   /* source keypoint angle in degrees (rotation check only)            */
    int32_t flags;                 /* bit0 active (visible, not bad), bit2 Observations() > 0           */
} gfo_proj_query;
typedef struct {
    int32_t use_ratio;
    float nn_ratio;
    int32_t th_dist;            /* 0..255: TH_HIGH (100) or ORBdist; 256 is the reference's "no candidate" value of bestDist */
    int32_t check_orientation;
    int32_t max_matches;        /* 0: no limit.  > 0: BUDGETING_FEATURE_MATCHING (include/ORBmatcher.h:36-37, MAX_NUM_FEATURE_MATCHING):
                                 * the loop over the queries ends with the query whose match makes nmatches reach this number, and that
                                 * match stays out of the rotation histogram (src/ORBmatcher.cc:1547-1552)                                */
} gfo_proj_mode;
int gfo_search_by_projection_queries(gfo_ctx* ctx, const gfo_keypoint* kp_un, const uint8_t* desc,
                                     const float* u_right, const float* kp_angle, int n,
                                     const gfo_frame_bounds* bounds, const gfo_proj_query* queries,
                                     const uint8_t* q_desc, int m, const gfo_proj_mode* mode,
                                     const uint8_t* kp_taken, int32_t* out_q, int32_t* out_score, int* nmatches);

/* The good-feature matchers (GOOD_FEATURE_MAP_MATCHING is the reference's default build, include/Tracking.h:75): the loop body of
 * SearchByProjection(Frame&, vector<MapPoint*>&, th) exists three more times in the reference, word for word --
 *   ORBmatcher::SearchByProjection_Budget(F, vpMapPoints, th, time_constr)      include/ORBmatcher.h:67, src/ORBmatcher.cc:45-153
 *       (Tracking::SearchAdditionalMatchesInFrame, src/Tracking.cc:2166): the same loop, `pMP->IncreaseFound()` per match, and a
 *       wall clock read at the END of the body (:96-102) that breaks the loop;
 *   ORBmatcher::SearchByProjection_OnePoint(F, pMP, th)                         include/ORBmatcher.h:71-150
 *       one point; called point after point by Observability::runBaselineMapMatching (src/Observability.cc:1233-1262, in a sorted
 *       order, until a weighted match budget is met) and runActiveMapMatching (:975);
 *   ORBmatcher::GetCandidates / MatchCandidates                                 include/ORBmatcher.h:152-250
 * gfo_search_by_projection_points = gfo_search_by_projection that ALSO reports what every point did at its turn:
 *   out_point[p] >= 0 : SearchByProjection_OnePoint's return value for point p (the keypoint it took) in bits 0-15, its distance
 *                       (what the call stored in mvpMatchScore at that moment) in bits 16-24
 *   GFO_POINT_NONE    : not in view / bad / no keypoint in its window (the reference `continue`s before computing a distance)
 *   GFO_POINT_RATIO   : best candidate within TH_HIGH, rejected by the ratio test (`continue`, :85-86)
 *   GFO_POINT_FAR     : keypoints in the window, none free and within TH_HIGH (falls through to the end of the loop body)
 * A point's outcome depends on the points IN FRONT of it only, so every early exit of the reference is a PREFIX of this answer:
 * gfo_projection_points_prefix rebuilds out_mp / out_score / nmatches for the first `prefix` points (host bookkeeping, no device
 * call).  SearchByProjection_Budget's clock is read after a point whose outcome is >= 0 or GFO_POINT_FAR -- the other two leave the
 * body early -- which is what a caller needs to place the cut for a given clock. */
#define GFO_POINT_NONE (-1)
#define GFO_POINT_RATIO (-2)
#define GFO_POINT_FAR (-3)
int gfo_search_by_projection_points(gfo_ctx* ctx, const gfo_keypoint* kp_un, const uint8_t* desc,
                                    const float* u_right, int n, const float* scale_factors, int nlevels,
                                    const gfo_frame_bounds* bounds, const gfo_map_point* mps,
                                    const uint8_t* mp_desc, int m, float th, float nn_ratio,
                                    const uint8_t* kp_taken, int32_t* out_mp, int32_t* out_score,
                                    int32_t* out_point, int* nmatches);
int gfo_projection_points_prefix(const int32_t* out_point, int m, int prefix, int n, int32_t* out_mp,
                                 int32_t* out_score, int* nmatches);

/* ORBmatcher::GetCandidates (include/ORBmatcher.h:152-172) for EVERY map point in one call -- pMP->mvMatchCandidates =
 * F.GetFeaturesInArea(mTrackProjX, mTrackProjY, r * mvScaleFactors[level], level - 1, level) -- together with the two things
 * ORBmatcher::MatchCandidates (:176-250) and SearchByProjection_OnePoint (:71-150) compute per candidate that do not depend on what the
 * frame's slots hold at the moment of the call: the mvuRight gate (:118-123) and the descriptor distance (:127).  What remains of
 * those two functions is gfo_match_candidates: a few compares per candidate, interleaved by the caller with its own selection loop
 * (Observability::runActiveMapMatching picks the next point from the outcome of the previous ones, src/Observability.cc:900-1100 --
 * an order no batch can know in advance).
 *   cand_start[m + 1] : CSR offsets; point p's candidates are cand[cand_start[p] .. cand_start[p + 1])
 *   cand[]            : in GetFeaturesInArea's order (grid column, grid row, keypoint index):
 *                       bits 0-15 keypoint index | 16-19 octave | 20-28 descriptor distance 0..256 | bit 31 the mvuRight gate rejects it
 *                       (such an entry is part of mvMatchCandidates -- its size() is the cost term of INFORMATION_EFFICIENCY_SCORE --
 *                       and is skipped by the match)
 *   cap               : entries the caller's cand[] holds.  *total = entries of the table; more than cap: GFO_ERR_CAPACITY, cand_start is
 *                       valid, cand[] is not (call again with room for *total; cap = 0 just asks for the size)
 * A point that is not in view, bad, or whose level lies outside the table has no candidates.
 * gfo_match_candidates(cand + cand_start[p], cand_start[p + 1] - cand_start[p], slot_taken, n, nn_ratio, &dist): slot_taken[n], n = the frame's
 * keypoints: slot_taken[i] = 1 where F.mvpMapPoints[i] is set with Observations() > 0 NOW (may be NULL: nothing taken); an entry whose
 * keypoint index is not below n is ignored.  Returns the keypoint index the reference's
 * function returns (the caller then stores the point and its *best_dist in the slot, :143-146, and marks slot_taken if the point has
 * observations), or GFO_POINT_NONE / _RATIO / _FAR.  Host arithmetic only; no device call, no descriptor read. */
int gfo_projection_candidates(gfo_ctx* ctx, const gfo_keypoint* kp_un, const uint8_t* desc, const float* u_right, int n,
                              const float* scale_factors, int nlevels, const gfo_frame_bounds* bounds,
                              const gfo_map_point* mps, const uint8_t* mp_desc, int m, float th,
                              int32_t* cand_start, uint32_t* cand, int cap, int* total);
int gfo_match_candidates(const uint32_t* cand, int ncand, const uint8_t* slot_taken, int n, float nn_ratio, int* best_dist);

/* gfo_search_by_projection_queries that also reports what every QUERY did at its turn (out_point[m]: keypoint | distance << 16, or
 * GFO_POINT_NONE / _RATIO / _FAR; before any rotation check; mode->max_matches must be 0).  With queries that block nothing (flags bit 2
 * clear) every query is an independent best-match search over the frame as it was on entry -- the search inside
 * ORBmatcher::Fuse(KF, Scw, ...) (src/ORBmatcher.cc:1089-1212: window, the two predicted levels, TH_LOW) and inside either direction of
 * SearchBySim3 (:1214-1438), whose side effects (Replace / AddObservation, the mutual-agreement pass) stay with the caller.
 * (Fuse(KF, MapPoints, th), :937-1087, gates its candidates by reprojection error before picking the best: not this form.) */
int gfo_search_by_projection_queries_points(gfo_ctx* ctx, const gfo_keypoint* kp_un, const uint8_t* desc,
                                            const float* u_right, const float* kp_angle, int n,
                                            const gfo_frame_bounds* bounds, const gfo_proj_query* queries,
                                            const uint8_t* q_desc, int m, const gfo_proj_mode* mode,
                                            const uint8_t* kp_taken, int32_t* out_q, int32_t* out_score,
                                            int32_t* out_point, int* nmatches);

/* The search of ORBmatcher::Fuse(KeyFrame* pKF, const vector<MapPoint*>& vpMapPoints, th) -- include/ORBmatcher.h, src/ORBmatcher.cc:937-1087,
 * local mapping's matcher (LocalMapping::SearchInNeighbors) -- on points the caller has projected into the keyframe (:955-1004: u, v,
 * ur = u - bf / z, radius = th * mvScaleFactors[predicted level], levels [predicted - 1, predicted]): per point the most similar keypoint
 * among those whose reprojection error passes the chi-square test with the KEYPOINT's level (:1019-1051: e2 * mvInvLevelSigma2[kpLevel]
 * against 5.99, or 7.8 with the right-image coordinate where mvuRight[idx] >= 0), accepted up to th_dist (TH_LOW).  No point hides a
 * keypoint from another (flags bit 2 is ignored); what is done with a find -- Replace / AddObservation + AddMapPoint, :1067-1083, which
 * may turn a later point of the vector bad -- stays with the caller, in the vector's order.
 * kp_un / desc / u_right = pKF->mvKeysUn / mDescriptors / mvuRight; inv_level_sigma2 = pKF->mvInvLevelSigma2 (every keypoint octave below
 * nlevels).  out_point[m]: keypoint | distance << 16, or negative (GFO_POINT_NONE / GFO_POINT_FAR).  The sums of squares are evaluated left
 * to right without fused multiply-adds, the products compared in double, as the expressions stand in the reference. */
int gfo_search_for_fusion(gfo_ctx* ctx, const gfo_keypoint* kp_un, const uint8_t* desc, const float* u_right, int n,
                          const gfo_frame_bounds* bounds, const float* inv_level_sigma2, int nlevels,
                          const gfo_proj_query* queries, const uint8_t* q_desc, int m, int th_dist, int32_t* out_point);

/* Device-resident, batched form of SearchByProjection(Frame&, vector<MapPoint*>&, th) -- the chain
 *   gfo_extract_batch_device -> [gfo_stereo_match_batch] -> gfo_search_by_projection_batch
 * never leaves the GPU: keypoints, descriptors (and mvuRight) are read where the extractor / stereo matcher
 * left them in the arena, the Frame grid (Frame.cc:461-476) is built per frame on the device, and the local
 * map's descriptors stay resident between frames as Tracking::mvpLocalMapPoints does (Tracking.cc:2335-2346
 * calls the matcher once per frame with the same map and fresh projections).
 *
 * gfo_map_upload: descriptors [m][32] of the local map (MapPoint::GetDescriptor() in vector order); resident
 * until replaced by the next upload.  The host buffer is free on return.
 * gfo_search_by_projection_batch: one search per frame of the last batch.  mps is [frames][m]: the per-frame
 * projection of every map point (Frame::isInFrustum fills mTrackProjX/Y/XR, mnTrackScaleLevel, mTrackViewCos,
 * mbTrackInView per frame, Frame.cc:512-590).  The extractor's keypoints stand for mvKeysUn (rectified or
 * distortion-free input); bounds = mnMinX..mnMaxY.  stereo = 1: frame k is the pair k of the last
 * gfo_stereo_match_batch (left keypoints + mvuRight), otherwise frame k = image k with mvuRight = -1.
 * on_device = 1: mps / kp_taken are device pointers used in place (they must stay valid until the stream has
 * consumed them); 0: host arrays, copied.  kp_taken (optional) is [frames][kp_stride] with kp_stride from
 * gfo_batch_device_views.  Asynchronous on the context stream; results stay on the device. */
typedef struct {
    const gfo_map_point* mps;
    const uint8_t* kp_taken;
    int32_t on_device;
    int32_t stereo;
    float th;               /* the `th` argument (window factor) */
    float nn_ratio;         /* ORBmatcher::mfNNratio             */
    gfo_frame_bounds bounds;
} gfo_projection_batch;
int gfo_map_upload(gfo_ctx* ctx, const uint8_t* mp_desc, int m);
int gfo_search_by_projection_batch(gfo_ctx* ctx, const gfo_projection_batch* p);
/* out_mp / out_score as gfo_search_by_projection, for the first min(n, cap) keypoints of the frame */
int gfo_projection_fetch(gfo_ctx* ctx, int frame, int32_t* out_mp, int32_t* out_score, int cap, int* nmatches);
/* device views of the batch results: out_mp / out_score are [frames][*stride], counters [frames][*counters_stride]
 * int32 with [0] live points, [1] fixed-point rounds, [2] nmatches */
int gfo_projection_device_views(gfo_ctx* ctx, const int32_t** d_out_mp, const int32_t** d_out_score,
                                const int32_t** d_counters, int* stride, int* counters_stride);

/* ORBmatcher::SearchByBoW(KeyFrame* pKF, Frame& F, vector<MapPoint*>& vpMapPointMatches)
 * include/ORBmatcher.h:272, src/ORBmatcher.cc:270-404.  The two DBoW2::FeatureVector maps
 * (Thirdparty/DBoW2/DBoW2/FeatureVector.h: map<NodeId, vector<unsigned>>) are passed flattened to CSR in
 * the map's own (ascending node id) order; the DBoW2 transform that produces them stays on the host.
 * kf_mp_valid[i] = 1 where pKF->GetMapPointMatches()[i] is set and !isBad().  kf_angle = pKF->mvKeysUn[i].angle,
 * f_angle = F.mvKeys[i].angle (used when check_orientation, = mbCheckOrientation).
 * out_kf_idx[n_f]: index i of the keyframe keypoint whose map point is left in vpMapPointMatches[j], -1 = NULL. */
typedef struct {
    const uint32_t* node_ids;   /* [n_nodes] ascending                    */
    const int32_t* node_start;  /* [n_nodes + 1] offsets into items        */
    const uint32_t* items;      /* keypoint indices in the vector's order  */
    int32_t n_nodes;
} gfo_feature_vector;
int gfo_search_by_bow(gfo_ctx* ctx, const uint8_t* kf_desc, const float* kf_angle, const uint8_t* kf_mp_valid,
                      int n_kf, const gfo_feature_vector* kf_fv, const uint8_t* f_desc, const float* f_angle,
                      int n_f, const gfo_feature_vector* f_fv, float nn_ratio, int check_orientation,
                      int32_t* out_kf_idx, int* nmatches);
/* The same with BUDGETING_FEATURE_MATCHING compiled in (include/ORBmatcher.h:36-37, src/ORBmatcher.cc:360-365): once nmatches has reached
 * max_matches the loop over a node's keyframe keypoints ends -- the reference breaks out of THAT loop only, so every later common node
 * still contributes its first accepted match.  max_matches <= 0: gfo_search_by_bow. */
int gfo_search_by_bow_budget(gfo_ctx* ctx, const uint8_t* kf_desc, const float* kf_angle, const uint8_t* kf_mp_valid,
                             int n_kf, const gfo_feature_vector* kf_fv, const uint8_t* f_desc, const float* f_angle,
                             int n_f, const gfo_feature_vector* f_fv, float nn_ratio, int check_orientation, int max_matches,
                             int32_t* out_kf_idx, int* nmatches);

/* ORBmatcher::SearchByBoW(KeyFrame* pKF1, KeyFrame* pKF2, vector<MapPoint*>& vpMatches12) -- include/ORBmatcher.h:273,
 * src/ORBmatcher.cc:635-768 (loop closing, src/LoopClosing.cc:287).  As the (KeyFrame, Frame) overload with: a map-point mask on BOTH
 * sides (mp_valid[i] = 1 where GetMapPointMatches()[i] is set and !isBad(), :672-676, :690-696), a strict distance test
 * (bestDist1 < TH_LOW, :713), angles = mvKeysUn[i].angle of either keyframe, and the answer indexed by the FIRST keyframe:
 * out_idx2[n1] = keypoint of pKF2 whose map point is left in vpMatches12[i], -1 = NULL. */
int gfo_search_by_bow_keyframes(gfo_ctx* ctx, const uint8_t* desc1, const float* angle1, const uint8_t* mp_valid1, int n1,
                                const gfo_feature_vector* fv1, const uint8_t* desc2, const float* angle2,
                                const uint8_t* mp_valid2, int n2, const gfo_feature_vector* fv2, float nn_ratio,
                                int check_orientation, int32_t* out_idx2, int* nmatches);

/* ORBmatcher::SearchForInitialization(Frame& F1, Frame& F2, vector<cv::Point2f>& vbPrevMatched, vector<int>& vnMatches12, int windowSize)
 * -- include/ORBmatcher.h, src/ORBmatcher.cc:520-633; the monocular bootstrap's matcher (Tracking::MonocularInitialization,
 * src/Tracking.cc:1322).  kp1 / desc1 = F1.mvKeysUn / mDescriptors, kp2 / desc2 = F2's, fb = F2's image bounds (its grid is rebuilt from them,
 * Frame.cc:461-476); prev_matched = vbPrevMatched as n1 (x, y) pairs, IN and OUT (:626-629).  Only level-0 keypoints of F1 search, and only
 * F2's level-0 keypoints within window_size of prev_matched[i] are looked at (GetFeaturesInArea, :538); best and second-best distance,
 * TH_LOW, the ratio test; a later keypoint of F1 takes a keypoint of F2 from an earlier one when strictly closer (:557, :575-581);
 * rotation histogram when check_orientation (a robbed keypoint still counts in its bin, as in the reference).  matches12[n1] =
 * vnMatches12.  The windows and every candidate's descriptor distance come from one device call (the candidate table of
 * gfo_projection_candidates); the ordered pass over the table runs on the host inside this call. */
int gfo_search_for_initialization(gfo_ctx* ctx, const gfo_keypoint* kp1, const uint8_t* desc1, int n1, float* prev_matched, const gfo_keypoint* kp2,
                                  const uint8_t* desc2, int n2, const gfo_frame_bounds* fb, int window_size, float nn_ratio,
                                  int check_orientation, int32_t* matches12, int* nmatches);

/* ORBmatcher::SearchForTriangulation(KeyFrame* pKF1, KeyFrame* pKF2, cv::Mat F12, vector<pair<size_t, size_t>>& vMatchedPairs, bOnlyStereo)
 * -- include/ORBmatcher.h, src/ORBmatcher.cc:770-935; local mapping's matcher for NEW map points (LocalMapping::CreateNewMapPoints,
 * src/LocalMapping.cc:435).  Keypoints WITHOUT a map point (has_mp = GetMapPoint(i) != NULL) of two keyframes, node by node of their
 * feature vectors: for a keypoint of pKF1 the candidate of pKF2 with the smallest distance <= TH_LOW among those that pass the gates of
 * :836-860 -- both stereo when only_stereo (u_right >= 0), not within 100 * mvScaleFactors[octave] (squared pixels) of the epipole
 * (ex, ey) when neither is stereo, CheckDistEpipolarLine (:251-268: distance^2 to the epipolar line x1' F12 below 3.84 * mvLevelSigma2[
 * octave]; f12 row-major) -- and on a tie the LAST one of the node's list.  Nothing a keypoint finds hides a candidate from the next one
 * (the reference declares vbMatched2 and never sets it).  Rotation histogram over the pairs as in the other matchers when
 * check_orientation.  out_idx2[n1] = vMatches12 (-1 none); vMatchedPairs = the (i, out_idx2[i]) with out_idx2[i] >= 0 in ascending i.
 * Float expressions are evaluated as written (left to right, no fused multiply-adds, a correctly rounded division). */
int gfo_search_for_triangulation(gfo_ctx* ctx, const gfo_keypoint* kp1, const uint8_t* desc1, const uint8_t* has_mp1, const float* u_right1, int n1,
                                 const gfo_feature_vector* fv1, const gfo_keypoint* kp2, const uint8_t* desc2, const uint8_t* has_mp2,
                                 const float* u_right2, int n2, const gfo_feature_vector* fv2, const float* scale_factors2,
                                 const float* level_sigma2_2, int nlevels, const float* f12, float ex, float ey, int only_stereo,
                                 int check_orientation, int32_t* out_idx2, int* nmatches);

/* Frame::ComputeBoW -> DBoW2 TemplatedVocabulary<FORB>::transform(features, BowVector&, FeatureVector&, levelsup)
 * src/Frame.cc:661-668, Thirdparty/DBoW2/DBoW2/TemplatedVocabulary.h:1140-1212 and the per-feature tree descent
 * :1231-1272 with FORB::distance (FORB.cpp:81).  The vocabulary tree is passed flattened: node 0 is the root,
 * the children of a node are the contiguous range [first_child, first_child + n_children) in the vocabulary's
 * own child order (ties keep the FIRST minimum, :1251-1260), a leaf has n_children == 0.
 * Per descriptor i the call returns the word id, the word weight and the node id at level L - levelsup
 * (0 = root when that level is <= 0); the adapter folds them into the two maps (only weight > 0 entries,
 * in feature order, :1169-1175).  The tree stays resident in the context until replaced. */
typedef struct {
    const int32_t* first_child;   /* [n_nodes]                               */
    const int32_t* n_children;    /* [n_nodes] 0 = leaf (word)               */
    const uint8_t* descriptors;   /* [n_nodes][32] node centres (root unused) */
    const int32_t* word_id;       /* [n_nodes] valid on leaves               */
    const float* weight;          /* [n_nodes] valid on leaves               */
    int32_t n_nodes;
    int32_t depth;                /* m_L                                     */
    const double* weight64;       /* [n_nodes] optional: Node::weight as DBoW2 holds it (WordValue = double); NULL = weight[] */
} gfo_vocabulary;
int gfo_vocabulary_upload(gfo_ctx* ctx, const gfo_vocabulary* voc);
/* Residency is a property of the CONTEXT, never of its address: gfo_vocabulary_nodes returns the node count of the tree
 * resident in ctx (0 = none -- a context that has just been created holds none, whatever lived at its address before),
 * gfo_ctx_id a process-wide serial number that no two contexts ever share (a `new` right after a `delete` returns the
 * same pointer; the id still differs).  The adapter asks these instead of remembering "uploaded" per pointer. */
int gfo_vocabulary_nodes(const gfo_ctx* ctx);
uint64_t gfo_ctx_id(const gfo_ctx* ctx);
int gfo_bow_transform(gfo_ctx* ctx, const uint8_t* desc, int n, int levelsup,
                      int32_t* word_id, float* weight, int32_t* node_id);

/* Frame::ComputeBoW in full -- mpORBvocabulary->transform(vCurrentDesc, mBowVec, mFeatVec, levelsup), src/Frame.cc:661-668,
 * Thirdparty/DBoW2/DBoW2/TemplatedVocabulary.h:1140-1212: the tree descent of every descriptor AND the fold of the
 * (word, weight, node) triples into the two maps, on the device.  The maps come back flattened in their own
 * (std::map, ascending key) order:
 *   BowVector     bow_words[k] ascending, bow_values[k] = the WordValue the reference ends up with: weights of a word
 *                 summed in feature order (TF_IDF, TF; BowVector::addWeight) or the first one kept (IDF, BINARY;
 *                 addIfNotExist), divided by the number of words when the scoring does not normalise (:1177-1183), then
 *                 BowVector::normalize(norm) (BowVector.cpp:62-84) -- every sum in the reference's order, in double;
 *   FeatureVector fv_node_ids[j] ascending, fv_items[fv_start[j] .. fv_start[j+1]) = the feature indices of that node,
 *                 ascending (FeatureVector::addFeature appends in feature order).
 * Stopped words (weight <= 0) enter neither map.  bow_* and fv_items / fv_node_ids take n entries, fv_start n + 1.
 * Up to 8192 descriptors are folded in LDS; more (up to 2^20 per call) by the same kernel on device memory -- slower, never refused. */
typedef struct {
    int32_t weighting;   /* DBoW2::WeightingType: 0 TF_IDF, 1 TF, 2 IDF, 3 BINARY (BowVector.h:26-32)  */
    int32_t norm;        /* 0 = the scoring does not normalise, 1 = L1, 2 = L2 (mustNormalize, ScoringObject.h:69-80) */
} gfo_bow_mode;
int gfo_compute_bow(gfo_ctx* ctx, const uint8_t* desc, int n, int levelsup, const gfo_bow_mode* mode,
                    uint32_t* bow_words, double* bow_values, int* n_words,
                    uint32_t* fv_node_ids, int32_t* fv_start, uint32_t* fv_items, int* n_fv_nodes);

/* ---- measurement hooks (bench.py / rocprof cross-check) ---------------------------------- */
/* When enabled, every kernel launch of the extract / stereo pipelines is bracketed by HIP
 * events on the context stream; gfo_profile_read returns, per stage, the accumulated device
 * milliseconds and launch count since the last reset. */
#define GFO_STAGE_MAX 16
typedef struct {
    char name[32];
    double ms;
    int launches;
} gfo_stage_time;
int gfo_profile_enable(gfo_ctx* ctx, int on);
int gfo_profile_read(gfo_ctx* ctx, gfo_stage_time* out, int cap, int* nstages, int reset);
/* Declares `left` and `right` the two extractors of ONE stereo rig -- the reference's mpORBextractorLeft / mpORBextractorRight, which
 * Frame::Frame calls from two threads per frame (Frame.cc:84-87) before Frame::ComputeStereoMatches works on what they returned
 * (Frame.cc:100) -- and `p` the calibration that association will be asked for with.  A HINT: no call returns anything else than
 * without it.  With both contexts combining (gfo_ctx_set_combining), the two gfo_extract calls of a frame meet inside the library
 * and go to the device as ONE stereo submission (what gfo_extract_stereo issues: one input copy, one launch chain), which also
 * computes the frame's association; a gfo_stereo_match on `left` that passes, bit for bit, the keypoints and descriptors those two
 * calls returned, the same `p`, the context's own scale factors and no disparity windows is then answered from that result without
 * touching the device (the arrays are compared in full: rectified input, where the reference's mvKeysUn equals mvKeys, qualifies;
 * anything else is computed as before).  A side whose partner does not call within GFO_PAIR_WAIT_US (2000) extracts alone; after
 * three such frames in a row the rig stops waiting altogether until it is declared again (the extractor is being used on its own;
 * a caller that keeps declaring a rig whose two sides never meet -- left and right extracted on one thread -- wakes it ever more
 * rarely: 1, 2, 4 ... 1024 declarations).
 * Both contexts need equal extractor parameters and one device.  right == NULL or p == NULL dissolves the rig; destroying either
 * context does too.  The adapter declares the rig from Frame::ComputeStereoMatches_Undistorted (adapter/matchers_gfo.cc). */
int gfo_ctx_pair(gfo_ctx* left, gfo_ctx* right, const gfo_stereo_params* p);
/* Counters of the frame combiner and of a context's stereo rig, for tests and harnesses: out[0] device batches, [1] requests,
 * [2] batches whose members were re-run alone after a batch-level failure, [3] batch slots prepared, [4] 1 if the engine could
 * prepare no slot and its callers take the direct path, [5] stereo frames extracted as one submission through gfo_ctx_pair,
 * [6] gfo_stereo_match calls answered from such a frame, [7] frames whose partner did not show up.  n <= 8 entries are written. */
int gfo_combiner_counters(const gfo_ctx* ctx, int64_t* out, int n);
/* The two timing parameters of a stereo rig that a CALLER may need to change (everything else behind a GFO_* environment variable only
 * picks between code paths that give the same results, and keeps its measured default): key "pair_wait_us" -- how long the first
 * extractor of a declared rig waits for its partner's image before it extracts alone (default 2000; 0 switches the rigs off; the
 * environment's GFO_PAIR_WAIT_US is the initial value), key "pair_spin_us" -- how long a lone camera's waiting side watches for its
 * result before it sleeps (default 400, GFO_PAIR_SPIN_US).  Process-wide, effective from the next frame; results never depend on
 * either (a frame whose partner is late is extracted alone and associated on request).  gfo_tuning_set returns GFO_ERR_INVALID for
 * an unknown key or a negative value; gfo_tuning_get returns -1 for an unknown key. */
int gfo_tuning_set(const char* key, long value);
long gfo_tuning_get(const char* key);
/* Process-wide monotonic counters: contexts created by gfo_ctx_create and arenas (re)planned -- hipMalloc of a whole
 * arena -- since the library was loaded.  A steady-state per-frame loop must leave both unchanged
 * (tools/c/boundary_throughput.c and the adapter's context table assert it). */
int gfo_contexts_created(void);
int gfo_arenas_planned(void);
/* Kernels resolved ahead of their first launch.  The first gfo_ctx_create on a device loads every code object of the library and
 * registers every kernel, serialised under one mutex, so that K host threads issuing their first frames at once (the reference's
 * K Frame constructors, Frame.cc:84-87) never race through the HIP runtime's lazy first-launch path.  Returns the number resolved
 * so far (kernels x devices used); a per-frame loop leaves it unchanged. */
int gfo_kernels_preloaded(void);

/* ---- inspection hooks for the parity tests (device -> host copies of intermediates) ------ */
int gfo_debug_blurred_level(gfo_ctx* ctx, int image, int level, uint8_t* out, int out_stride);
/* FAST candidates of one level handed to the quadtree: {x, y, score} int32 triplets, x/y
 * relative to minBorder as in ORBextractor.cc:824-825; order unspecified (a set). */
int gfo_debug_level_candidates(gfo_ctx* ctx, int image, int level, int32_t* xys, int cap, int* n);

#ifdef __cplusplus
}
#endif
#endif /* GFO_H */
