// matchers_gfo.cc -- the matcher bodies a maintainer swaps in on the reference side.
//
// Compile this file inside the reference tree next to (a trimmed) src/Frame.cc / src/ORBmatcher.cc and link
// libgfo.so.  Every function below has the reference's own signature (include/Frame.h, include/ORBmatcher.h,
// unchanged) and replaces the body the reference defines at the cited lines; each is guarded by its own macro so the
// swap can be done one function at a time (-DGFO_ADAPTER_ALL takes all of them):
//
//   GFO_ADAPTER_STEREO        Frame::ComputeStereoMatches_Undistorted(bool)                     src/Frame.cc:1167-1316
//   GFO_ADAPTER_PROJECTION    ORBmatcher::SearchByProjection(Frame&, vector<MapPoint*>&, th)    src/ORBmatcher.cc:155-241
//   GFO_ADAPTER_PROJ_BUDGET   ORBmatcher::SearchByProjection_Budget(F, MapPoints, th, time)     src/ORBmatcher.cc:45-153
//   GFO_ADAPTER_PROJ_LAST     ORBmatcher::SearchByProjection(Cur, Last, th, bMono, nVisible)    src/ORBmatcher.cc:1440-1593
//   GFO_ADAPTER_PROJ_KF       ORBmatcher::SearchByProjection(Cur, KF*, sAlreadyFound, th, dist) src/ORBmatcher.cc:1595-1721
//   GFO_ADAPTER_BOW           ORBmatcher::SearchByBoW(KeyFrame*, Frame&, vector<MapPoint*>&)    src/ORBmatcher.cc:270-404
//   GFO_ADAPTER_BOW_KF        ORBmatcher::SearchByBoW(KeyFrame*, KeyFrame*, vector<MapPoint*>&) src/ORBmatcher.cc:635-768
//   GFO_ADAPTER_PROJ_SCW      ORBmatcher::SearchByProjection(KeyFrame*, Scw, points, matched, th)  src/ORBmatcher.cc:406-518
//   GFO_ADAPTER_FUSE_SCW      ORBmatcher::Fuse(KeyFrame*, Scw, points, th, vpReplacePoint)         src/ORBmatcher.cc:1089-1212
//   GFO_ADAPTER_FUSE          ORBmatcher::Fuse(KeyFrame*, vector<MapPoint*>&, th)                  src/ORBmatcher.cc:937-1087
//   GFO_ADAPTER_SIM3          ORBmatcher::SearchBySim3(KF1, KF2, vpMatches12, s12, R12, t12, th)    src/ORBmatcher.cc:1214-1438
//   GFO_ADAPTER_TRIANGULATION ORBmatcher::SearchForTriangulation(KF1, KF2, F12, vMatchedPairs, bOnlyStereo) src/ORBmatcher.cc:770-935
//   GFO_ADAPTER_INIT          ORBmatcher::SearchForInitialization(F1, F2, vbPrevMatched, vnMatches12, windowSize) src/ORBmatcher.cc:520-633
//   GFO_ADAPTER_COMPUTE_BOW   Frame::ComputeBoW()                                               src/Frame.cc:661-668
//
// Each body flattens the reference's objects into plain arrays, calls the C ABI (include/gfo.h) and writes the
// results back into the same members the reference fills.  Error behaviour follows the reference (no exceptions,
// no status): a gfo error is reported on stderr and the function returns "nothing matched".
//
// The device context of a frame is the one its left extractor owns (adapter/ORBextractor_gfo.cc keeps it in a side
// table because include/ORBextractor.h cannot carry a new member): gfo_context_pin() / GfoUse.
#ifdef GFO_ADAPTER_ALL
#define GFO_ADAPTER_STEREO
#define GFO_ADAPTER_PROJECTION
#define GFO_ADAPTER_PROJ_BUDGET
#define GFO_ADAPTER_PROJ_LAST
#define GFO_ADAPTER_PROJ_KF
#define GFO_ADAPTER_BOW
#define GFO_ADAPTER_BOW_KF
#define GFO_ADAPTER_PROJ_SCW
#define GFO_ADAPTER_FUSE_SCW
#define GFO_ADAPTER_FUSE
#define GFO_ADAPTER_SIM3
#define GFO_ADAPTER_TRIANGULATION
#define GFO_ADAPTER_INIT
#define GFO_ADAPTER_COMPUTE_BOW
#endif

#include "Frame.h"
#include "KeyFrame.h"
#include "MapPoint.h"
#include "ORBmatcher.h"

// Compile-time variants of the reference, both reproduced -- compile this file with the same macros as the rest of the tree:
//   DELAYED_STEREO_MATCHING (include/Frame.h:40, off by default): ComputeStereoMatches_Undistorted visits what src/Frame.cc:1186-1199
//     visits (tests/_build/adapter_run_delayed);
//   BUDGETING_FEATURE_MATCHING (include/ORBmatcher.h:36-37, off by default): SearchByBoW / SearchByProjection(Cur, Last) stop adding
//     matches at MAX_NUM_FEATURE_MATCHING exactly where src/ORBmatcher.cc:360-365 / 1547-1552 break (gfo_search_by_bow_budget,
//     gfo_proj_mode::max_matches; tests/_build/adapter_run_budget).
#ifdef BUDGETING_FEATURE_MATCHING
#ifndef MAX_NUM_FEATURE_MATCHING
#error "BUDGETING_FEATURE_MATCHING needs MAX_NUM_FEATURE_MATCHING (include/ORBmatcher.h:37)"
#endif
#define GFO_FEATURE_BUDGET (MAX_NUM_FEATURE_MATCHING)
#else
#define GFO_FEATURE_BUDGET 0
#endif

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <map>
#include <mutex>
#include <set>
#include <vector>

#include "gfo.h"

namespace ORB_SLAM2
{

gfo_ctx* gfo_context_pin(const ORBextractor* e);   // adapter/ORBextractor_gfo.cc
void gfo_context_unpin(const ORBextractor* e, gfo_ctx* c);
bool gfo_context_colocate(const ORBextractor* follower, const ORBextractor* leader);
gfo_ctx* gfo_context_pin_thread();                 // a context of the calling thread's own, for matcher calls without a Frame
void gfo_context_unpin_thread(gfo_ctx* c);

namespace
{
static_assert(sizeof(cv::KeyPoint) == sizeof(gfo_keypoint), "gfo_keypoint must mirror cv::KeyPoint");

// the device context of an extractor, pinned in the adapter's table for the duration of one matcher call (another thread that
// creates a context may reclaim idle ones meanwhile: never one that is in use)
struct GfoUse {
    const ORBextractor* e;
    gfo_ctx* c;
    explicit GfoUse(const ORBextractor* e_) : e(e_), c(e_ ? gfo_context_pin(e_) : NULL) {}
    ~GfoUse() { gfo_context_unpin(e, c); }
private:
    GfoUse(const GfoUse&);
    GfoUse& operator=(const GfoUse&);
};

inline const gfo_keypoint* as_gfo(const std::vector<cv::KeyPoint>& v) { return reinterpret_cast<const gfo_keypoint*>(v.data()); }

void report(gfo_ctx* c, const char* where) { fprintf(stderr, "[gfo] %s: %s\n", where, gfo_last_error(c)); }

// DBoW2::FeatureVector (std::map<NodeId, std::vector<unsigned int>>, iterated in ascending node id) as the CSR
// gfo_search_by_bow takes
struct FlatFeatVec {
    std::vector<uint32_t> ids, items;
    std::vector<int32_t> start;
    gfo_feature_vector view;
    explicit FlatFeatVec(const DBoW2::FeatureVector& fv)
    {
        start.push_back(0);
        for (DBoW2::FeatureVector::const_iterator it = fv.begin(); it != fv.end(); ++it) {
            ids.push_back(it->first);
            items.insert(items.end(), it->second.begin(), it->second.end());
            start.push_back((int32_t)items.size());
        }
        view.node_ids = ids.data();
        view.node_start = start.data();
        view.items = items.data();
        view.n_nodes = (int32_t)ids.size();
    }
};

// One descriptor into row i of a block.  MapPoint::GetDescriptor() (src/MapPoint.cc:464-468) is `lock(mMutexFeatures); return
// mDescriptor.clone();` -- an allocation and a release per point, a few thousand times per matcher call.  The same 32 bytes under the
// same mutex, without the temporary: both members are protected (include/MapPoint.h:173,193), a derived type reads them -- the device
// VocabularyView below uses for the vocabulary's tree.  (A descriptor of another shape goes through GetDescriptor() as before.)
struct MapPointView : public MapPoint {
    bool descriptor_into(uint8_t* dst)
    {
        std::unique_lock<std::mutex> lock(mMutexFeatures);
        if (!(mDescriptor.data && mDescriptor.rows * mDescriptor.cols >= 32 && mDescriptor.isContinuous())) return false;
        memcpy(dst, mDescriptor.data, 32);
        return true;
    }
};
inline void descriptor_row(MapPoint* pMP, cv::Mat& block, int i)
{
    if (static_cast<MapPointView*>(pMP)->descriptor_into(block.data + (size_t)i * 32)) return;
    const cv::Mat d = pMP->GetDescriptor();
    if (d.data && d.rows * d.cols >= 32 && d.isContinuous()) memcpy(block.data + (size_t)i * 32, d.data, 32);
    else d.copyTo(block.row(i));
}

// continuous descriptor rows (mDescriptors is created by the extractor adapter as one N x 32 block; a row view
// handed in from elsewhere may not be)
inline const uint8_t* rows32(const cv::Mat& m, cv::Mat& keep)
{
    if (m.isContinuous()) return m.data;
    keep = m.clone();
    return keep.data;
}
}  // namespace

// ---------------------------------------------------------------------------------------------------------------
#ifdef GFO_ADAPTER_STEREO
// The member keeps state between calls on one frame, as the reference's does: mvuRight / mvDepth / mvStereoMatched / mvDistIdx are
// reset only by PrepareStereoCandidates, which runs only when `mvRowIndices.size() != nRows` (Frame.cc:1173-1176, Frame.h:230-263).
//   * first call on a frame (every Frame constructor, Frame.cc:100,198): state reset, the library's answer IS the member's;
//   * a later call on the same frame (Tracking.cc:941-954 in the default build, after map points narrowed the windows; every call
//     of a DELAYED_STEREO_MATCHING build after the caller's own PrepareStereoCandidates, Tracking.cc:613,649,681): a keypoint the
//     call does not match KEEPS what it had, every accepted match is appended to mvDistIdx (:1282), and the outlier cut runs over
//     the accumulated list (:1290-1313; entries an earlier call cut are still in it, nmatched goes down for each, duplicates
//     included).  The library answers "this call on a fresh frame" and returns the accepted matches BEFORE its own cut
//     (best_dist / best_idx); the member's state is derived from those here, with the reference's own expressions (:1271-1281).
//   * isOnline: no cut at all (`if (!isOnline)`, :1290).
//   * DELAYED_STEREO_MATCHING (:1186-1199): the online call visits only unvisited keypoints that carry a map point, the offline
//     call the other unvisited ones; a keypoint that is not visited gets an empty disparity window (nothing can match it and the
//     library does not count it).
// mvRowIndices itself is read by nothing else in a build without DELAYED_STEREO_MATCHING (ComputeStereoMatch_OnePoint returns
// before it, :1086), so it is only SIZED there -- the flag the reference tests; -DGFO_ADAPTER_FULL_ROW_INDEX (implied by
// DELAYED_STEREO_MATCHING, whose Observability.cc:995,1259 read the lists) runs the reference's own PrepareStereoCandidates instead.
int Frame::ComputeStereoMatches_Undistorted(bool isOnline)
{
    const int nRows = mpORBextractorLeft->mvImagePyramid[0].rows;
    const bool first = mvRowIndices.size() != (size_t)nRows;     // :1173
    if (first) {
#if defined(GFO_ADAPTER_FULL_ROW_INDEX) || defined(DELAYED_STEREO_MATCHING)
        PrepareStereoCandidates();
#else
        mvuRight.assign(N, -1.0f);
        mvDepth.assign(N, -1.0f);
        mvStereoMatched.assign(N, false);
        mvRowIndices.assign(nRows, std::vector<size_t>());
        mvDistIdx.reserve(N);
        mvDistIdx.clear();
#endif
    }
    if ((int)mvuRight.size() != N) mvuRight.resize(N, -1.0f);    // (the reference would index past the end)
    if ((int)mvDepth.size() != N) mvDepth.resize(N, -1.0f);
    if ((int)mvStereoMatched.size() != N) mvStereoMatched.resize(N, false);

    std::vector<float> minD, maxD;
    bool windows = false;                      // Frame.cc:1220-1231: per-keypoint disparity window
#ifdef DELAYED_STEREO_MATCHING
    std::vector<uint8_t> visit(N);
    for (int iL = 0; iL < N; iL++) {           // :1186-1199
        visit[iL] = isOnline ? (mvpMapPoints[iL] != NULL && !mvStereoMatched[iL]) : !mvStereoMatched[iL];
        if (!visit[iL]) windows = true;
    }
#endif
    for (int iL = 0; iL < N && !windows; iL++) windows = mvpMapPoints[iL] != NULL;
    if (windows) {
        minD.assign(N, 0.f);
        maxD.assign(N, mbf / mb);
        for (int iL = 0; iL < N; iL++) {
#ifdef DELAYED_STEREO_MATCHING
            if (!visit[iL]) {                  // uL - minD < mnMinX whatever uL: not visited, not counted (:1235)
                minD[iL] = 3.0e38f;
                maxD[iL] = -3.0e38f;
                continue;
            }
#endif
            MapPoint* pMP = mvpMapPoints[iL];
            if (!pMP || pMP->isBad()) continue;
            cv::Mat Pw = pMP->GetWorldPos(), Pc;
            if (!WorldToCameraPoint(Pw, Pc)) continue;
            const float disp = float(mbf) / Pc.at<float>(2);
            minD[iL] = std::max(disp - float(DISPARITY_THRES), 0.0f);
            maxD[iL] = std::min(disp + float(DISPARITY_THRES), float(mbf) / float(mb));
        }
    }
#ifdef DELAYED_STEREO_MATCHING
    for (int iL = 0; iL < N; iL++) if (visit[iL]) mvStereoMatched[iL] = true;   // :1203
#else
    mvStereoMatched.assign(N, true);
#endif
    gfo_stereo_params p = {nRows, mbf, mb, mnMinX};
    std::vector<int32_t> bestDist(N), bestIdx(N);
    int nmatched = 0;
    cv::Mat keepL, keepR;
    // several GPUs (GFO_DEVICES): the rig's right extractor lives where its left one does -- the pair below needs one device, and
    // every matcher call of this frame runs in the left extractor's context.  Moves the right one once, on the rig's first frame.
    if (mpORBextractorRight) (void)gfo_context_colocate(mpORBextractorRight, mpORBextractorLeft);
    GfoUse use(mpORBextractorLeft), use_r(mpORBextractorRight);
    gfo_ctx* c = use.c;
    // the two extractors are one stereo rig with this calibration: from the next frame on their two operator() calls go to the
    // device as one stereo submission that also computes this association, and the call below -- on rectified input, where
    // mvKeysUn == mvKeys -- is answered from it (gfo_ctx_pair: a hint, idempotent, a few nanoseconds when nothing changed)
    if (c && use_r.c) (void)gfo_ctx_pair(c, use_r.c, &p);
    const bool direct = first && !isOnline;    // a fresh frame, offline: the library's arrays are the member's
    std::vector<float> ur, dp;
    if (!direct) { ur.resize(N); dp.resize(N); }
    float* const outU = direct ? mvuRight.data() : ur.data();
    float* const outD = direct ? mvDepth.data() : dp.data();
    const int rc = gfo_stereo_match(c, as_gfo(mvKeysUn), rows32(mDescriptors, keepL), N, as_gfo(mvKeysRightUn),
                                    rows32(mDescriptorsRight, keepR), (int)mvKeysRightUn.size(), mvScaleFactors.data(),
                                    (int)mvScaleFactors.size(), &p, windows ? minD.data() : NULL, windows ? maxD.data() : NULL,
                                    outU, outD, bestDist.data(), bestIdx.data(), &nmatched);
    if (rc != GFO_OK) {
        report(c, "ComputeStereoMatches_Undistorted");
        if (direct) { mvuRight.assign(N, -1.0f); mvDepth.assign(N, -1.0f); }
        return 0;
    }
    if (direct) {
        for (int iL = 0; iL < N; iL++)         // mvDistIdx as :1282 fills it, :1296 sorts it
            if (bestDist[iL] >= 0) mvDistIdx.push_back(std::pair<int, int>(bestDist[iL], iL));
        std::sort(mvDistIdx.begin(), mvDistIdx.end());
        return nmatched;
    }
    // the library counted as the reference does for ONE fresh call: keypoints that reach the end of the loop body (:1287), minus
    // one per entry of its own cut; the member's count starts from the former
    for (int iL = 0; iL < N; iL++) {
        if (bestDist[iL] < 0) continue;        // nothing accepted for this keypoint: it keeps what it had
        if (dp[iL] >= 0) {                     // accepted and not cut by the library (an accepted match has depth > 0)
            mvuRight[iL] = ur[iL];
            mvDepth[iL] = dp[iL];
        } else {                               // accepted, cleared by the library's cut: :1271-1281
            nmatched++;
            const float uL = mvKeysUn[iL].pt.x;
            float bestuR = mvKeysRightUn[bestIdx[iL]].pt.x;
            float disparity = uL - bestuR;
            if (disparity <= 0) {
                disparity = 0.01;
                bestuR = uL - 0.01;
            }
            mvDepth[iL] = mbf / disparity;
            mvuRight[iL] = bestuR;
        }
        mvDistIdx.push_back(std::pair<int, int>(bestDist[iL], iL));            // :1282
    }
    if (!isOnline) {                           // :1290-1313 over everything the frame has accumulated
        if (mvDistIdx.empty()) return nmatched;
        std::sort(mvDistIdx.begin(), mvDistIdx.end());
        const float median = mvDistIdx[mvDistIdx.size() / 2].first;
        const float thDist = 1.5f * 1.4f * median;
        for (int i = (int)mvDistIdx.size() - 1; i >= 0; i--) {
            if (mvDistIdx[i].first < thDist) break;
            mvuRight[mvDistIdx[i].second] = -1;
            mvDepth[mvDistIdx[i].second] = -1;
            nmatched--;
        }
    }
    return nmatched;                           // (an online call leaves mvDistIdx unsorted, as the reference does)
}
#endif

// ---------------------------------------------------------------------------------------------------------------
#if defined(GFO_ADAPTER_PROJECTION) || defined(GFO_ADAPTER_PROJ_BUDGET)
namespace
{
// The loop of ORBmatcher.cc:155-241 (= :45-153 = include/ORBmatcher.h:71-150) for all of vpMapPoints in ONE device call.  outPoint (optional):
// what every point did at its turn (gfo_search_by_projection_points).  false: the library refused the call (reported).
bool project_map_points(Frame& F, const std::vector<MapPoint*>& vpMapPoints, const float th, const float nnRatio, std::vector<int32_t>& outMp,
                        std::vector<int32_t>& outScore, std::vector<int32_t>* outPoint, int& nmatches, const char* who)
{
    const int M = (int)vpMapPoints.size(), N = F.N;
    std::vector<gfo_map_point> mps(M);
    cv::Mat mpDesc(M > 0 ? M : 1, 32, CV_8U);
    memset(mpDesc.data, 0, (size_t)(M > 0 ? M : 1) * 32);   // rows of the points that are skipped below
    for (int i = 0; i < M; i++) {
        MapPoint* pMP = vpMapPoints[i];
        gfo_map_point& m = mps[i];
        m.proj_x = pMP->mTrackProjX; m.proj_y = pMP->mTrackProjY; m.proj_xr = pMP->mTrackProjXR;
        m.view_cos = pMP->mTrackViewCos; m.level = pMP->mnTrackScaleLevel;
        // As the reference's loop (:163-168): a point that is not in view, or bad, is skipped before anything else is asked of it --
        // most of a local map, usually.  Only a point that will be searched pays for GetDescriptor() (a clone under the point's
        // mutex) and Observations(); the library never looks at the descriptor row or the third flag of the others.
        m.flags = pMP->mbTrackInView ? 1 : 0;
        if (!m.flags) continue;
        if (pMP->isBad()) { m.flags |= 2; continue; }
        if (pMP->Observations() > 0) m.flags |= 4;
        descriptor_row(pMP, mpDesc, i);
    }
    std::vector<uint8_t> taken(N);
    for (int i = 0; i < N; i++) taken[i] = F.mvpMapPoints[i] && F.mvpMapPoints[i]->Observations() > 0;
    gfo_frame_bounds fb = {Frame::mnMinX, Frame::mnMinY, Frame::mnMaxX, Frame::mnMaxY};
    outMp.assign(N, -1);
    outScore.assign(N, 0);
    if (outPoint) outPoint->assign(M, GFO_POINT_NONE);
    nmatches = 0;
    cv::Mat keep;
    GfoUse use(F.mpORBextractorLeft);
    gfo_ctx* c = use.c;
    const int rc = outPoint
        ? gfo_search_by_projection_points(c, as_gfo(F.mvKeysUn), rows32(F.mDescriptors, keep), F.mvuRight.data(), N, F.mvScaleFactors.data(),
                                          (int)F.mvScaleFactors.size(), &fb, mps.data(), mpDesc.data, M, th, nnRatio, taken.data(),
                                          outMp.data(), outScore.data(), outPoint->data(), &nmatches)
        : gfo_search_by_projection(c, as_gfo(F.mvKeysUn), rows32(F.mDescriptors, keep), F.mvuRight.data(), N, F.mvScaleFactors.data(),
                                   (int)F.mvScaleFactors.size(), &fb, mps.data(), mpDesc.data, M, th, nnRatio, taken.data(),
                                   outMp.data(), outScore.data(), &nmatches);
    if (rc != GFO_OK) {
        report(c, who);
        return false;
    }
    return true;
}
}  // namespace
#endif

#ifdef GFO_ADAPTER_PROJECTION
int ORBmatcher::SearchByProjection(Frame& F, const std::vector<MapPoint*>& vpMapPoints, const float th)
{
    std::vector<int32_t> outMp, outScore;
    int nmatches = 0;
    if (!project_map_points(F, vpMapPoints, th, mfNNratio, outMp, outScore, NULL, nmatches, "SearchByProjection(F, MapPoints)")) return 0;
    for (int i = 0; i < F.N; i++)
        if (outMp[i] >= 0) {
            F.mvpMapPoints[i] = vpMapPoints[outMp[i]];   // ORBmatcher.cc:233
            F.mvpMatchScore[i] = outScore[i];            // :235
        }
    return nmatches;
}
#endif

// ---------------------------------------------------------------------------------------------------------------
#ifdef GFO_ADAPTER_PROJ_BUDGET
// Tracking::SearchAdditionalMatchesInFrame's matcher (Tracking.cc:2166; GOOD_FEATURE_MAP_MATCHING, the reference's default build): the loop
// of the overload above with IncreaseFound() per match (:91) and a wall clock read at the end of the loop body that ends the loop once
// time_constr is spent (:96-102).  Here ALL points are answered by one device call and the clock is read once, after it:
//   * the call took less than time_constr (every frame: a call is ~0.2 ms, the budget a few ms): the answer is what the reference gives
//     whenever ITS clock does not run out -- every point visited;
//   * time_constr <= 0 on entry (the visibility check of Tracking.cc:2127-2151 used up the frame's budget): the reference's FIRST clock
//     reading ends the loop whatever the host's speed -- the first point that reaches the end of the body (a match, or candidates and
//     none within TH_HIGH) is the last one; the same prefix is taken here (gfo_projection_points_prefix);
//   * the call itself outlasted a positive time_constr: the reference would have stopped somewhere inside the list, where depends on
//     its host; the work is done, every match is kept (a superset of any such prefix, each entry what the reference's loop computes
//     for that point given the points before it).
int ORBmatcher::SearchByProjection_Budget(Frame& F, const std::vector<MapPoint*>& vpMapPoints, const float th, const double time_constr)
{
    const int M = (int)vpMapPoints.size(), N = F.N;
    std::vector<int32_t> outMp, outScore, outPoint;
    int nmatches = 0;
    if (!project_map_points(F, vpMapPoints, th, mfNNratio, outMp, outScore, &outPoint, nmatches, "SearchByProjection_Budget")) return 0;
    int prefix = M;
    if (!(time_constr > 0)) {
        for (int p = 0; p < M; p++)
            if (outPoint[p] >= 0 || outPoint[p] == GFO_POINT_FAR) { prefix = p + 1; break; }
        if (gfo_projection_points_prefix(outPoint.data(), M, prefix, N, outMp.data(), outScore.data(), &nmatches) != GFO_OK) return 0;
    }
    for (int p = 0; p < prefix; p++)
        if (outPoint[p] >= 0) vpMapPoints[p]->IncreaseFound();   // :91, once per match (also for a slot a later point takes over)
    for (int i = 0; i < N; i++)
        if (outMp[i] >= 0) {
            F.mvpMapPoints[i] = vpMapPoints[outMp[i]];   // :89
            F.mvpMatchScore[i] = outScore[i];            // :94
        }
    return nmatches;
}
#endif

// ---------------------------------------------------------------------------------------------------------------
#ifdef GFO_ADAPTER_PROJ_LAST
// TrackWithMotionModel's matcher (Tracking.cc:1516).  The projection of the last frame's map points (:1465-1497)
// stays on the host -- it is the reference's own cv::Mat arithmetic, a few thousand 3x3 products -- and becomes the
// query array; window search, Hamming, ordered resolution and the rotation histogram run on the device.
// With BUDGETING_FEATURE_MATCHING the loop ends with the MAX_NUM_FEATURE_MATCHING-th match (:1547-1552): gfo_proj_mode::max_matches.
int ORBmatcher::SearchByProjection(Frame& CurrentFrame, const Frame& LastFrame, const float th, const bool bMono,
                                   double& numVisibleMpt)
{
    const cv::Mat Rcw = CurrentFrame.mTcw.rowRange(0, 3).colRange(0, 3);
    const cv::Mat tcw = CurrentFrame.mTcw.rowRange(0, 3).col(3);
    const cv::Mat twc = -Rcw.t() * tcw;
    const cv::Mat Rlw = LastFrame.mTcw.rowRange(0, 3).colRange(0, 3);
    const cv::Mat tlw = LastFrame.mTcw.rowRange(0, 3).col(3);
    const cv::Mat tlc = Rlw * twc + tlw;
    const bool bForward = tlc.at<float>(2) > CurrentFrame.mb && !bMono;
    const bool bBackward = -tlc.at<float>(2) > CurrentFrame.mb && !bMono;

    std::vector<gfo_proj_query> q;
    std::vector<MapPoint*> qmp;
    q.reserve(LastFrame.N);
    qmp.reserve(LastFrame.N);
    for (int i = 0; i < LastFrame.N; i++) {
        MapPoint* pMP = LastFrame.mvpMapPoints[i];
        if (!pMP || LastFrame.mvbOutlier[i]) continue;
        cv::Mat x3Dw = pMP->GetWorldPos();
        cv::Mat x3Dc = Rcw * x3Dw + tcw;
        const float xc = x3Dc.at<float>(0);
        const float yc = x3Dc.at<float>(1);
        const float invzc = 1.0 / x3Dc.at<float>(2);
        if (invzc < 0) continue;
        float u = CurrentFrame.fx * xc * invzc + CurrentFrame.cx;
        float v = CurrentFrame.fy * yc * invzc + CurrentFrame.cy;
        if (u < CurrentFrame.mnMinX || u > CurrentFrame.mnMaxX) continue;
        if (v < CurrentFrame.mnMinY || v > CurrentFrame.mnMaxY) continue;
        numVisibleMpt++;
        const int nLastOctave = LastFrame.mvKeys[i].octave;
        gfo_proj_query e;
        e.u = u;
        e.v = v;
        e.ur = u - CurrentFrame.mbf * invzc;                                // :1523
        e.radius = th * CurrentFrame.mvScaleFactors[nLastOctave];            // :1502
        if (bForward) { e.min_level = nLastOctave; e.max_level = -1; }       // GetFeaturesInArea(u, v, r, octave)
        else if (bBackward) { e.min_level = 0; e.max_level = nLastOctave; }
        else { e.min_level = nLastOctave - 1; e.max_level = nLastOctave + 1; }
        e.angle = LastFrame.mvKeysUn[i].angle;                               // :1557
        e.flags = 1 | (pMP->Observations() > 0 ? 4 : 0);                     // what later queries see at :1520-1522
        q.push_back(e);
        qmp.push_back(pMP);
    }
    const int M = (int)q.size(), N = CurrentFrame.N;
    cv::Mat qDesc(M > 0 ? M : 1, 32, CV_8U);
    for (int i = 0; i < M; i++) descriptor_row(qmp[i], qDesc, i);
    std::vector<uint8_t> taken(N);
    std::vector<float> angle(N);
    for (int i = 0; i < N; i++) {
        taken[i] = CurrentFrame.mvpMapPoints[i] && CurrentFrame.mvpMapPoints[i]->Observations() > 0;
        angle[i] = CurrentFrame.mvKeysUn[i].angle;
    }
    gfo_frame_bounds fb = {Frame::mnMinX, Frame::mnMinY, Frame::mnMaxX, Frame::mnMaxY};
    gfo_proj_mode mode = {0, 0.f, TH_HIGH, mbCheckOrientation ? 1 : 0, GFO_FEATURE_BUDGET};      // no ratio test in this overload (:1541)
    std::vector<int32_t> outQ(N), outScore(N);
    int nmatches = 0;
    cv::Mat keep;
    GfoUse use(CurrentFrame.mpORBextractorLeft);
    gfo_ctx* c = use.c;
    const int rc = gfo_search_by_projection_queries(c, as_gfo(CurrentFrame.mvKeysUn), rows32(CurrentFrame.mDescriptors, keep),
                                                    CurrentFrame.mvuRight.data(), angle.data(), N, &fb, q.data(), qDesc.data, M, &mode,
                                                    taken.data(), outQ.data(), outScore.data(), &nmatches);
    if (rc != GFO_OK) {
        report(c, "SearchByProjection(Cur, Last)");
        return 0;
    }
    // out_q[i]: >= 0 the query left in mvpMapPoints[i]; -2 a slot this call wrote and its rotation check cleared
    // (:1579-1586 store NULL); -1 untouched
    for (int i = 0; i < N; i++) {
        if (outQ[i] >= 0) CurrentFrame.mvpMapPoints[i] = qmp[outQ[i]];
        else if (outQ[i] == -2) CurrentFrame.mvpMapPoints[i] = static_cast<MapPoint*>(NULL);
    }
    return nmatches;
}
#endif

// ---------------------------------------------------------------------------------------------------------------
#ifdef GFO_ADAPTER_PROJ_KF
// Relocalization's matcher (Tracking.cc, Relocalization()).  Differences from the overload above, all of them in
// the flattening: a keypoint with ANY map point is skipped (:1664-1665), so every accepted match blocks the slot for
// later map points (flag bit 2 always set, taken = "slot is set"); no mvuRight gate; the threshold is ORBdist; the
// level window is [predicted - 1, predicted + 1].
int ORBmatcher::SearchByProjection(Frame& CurrentFrame, KeyFrame* pKF, const std::set<MapPoint*>& sAlreadyFound, const float th,
                                   const int ORBdist)
{
    const cv::Mat Rcw = CurrentFrame.mTcw.rowRange(0, 3).colRange(0, 3);
    const cv::Mat tcw = CurrentFrame.mTcw.rowRange(0, 3).col(3);
    const cv::Mat Ow = -Rcw.t() * tcw;
    const std::vector<MapPoint*> vpMPs = pKF->GetMapPointMatches();
    std::vector<gfo_proj_query> q;
    std::vector<MapPoint*> qmp;
    for (size_t i = 0, iend = vpMPs.size(); i < iend; i++) {
        MapPoint* pMP = vpMPs[i];
        if (!pMP || pMP->isBad() || sAlreadyFound.count(pMP)) continue;
        cv::Mat x3Dw = pMP->GetWorldPos();
        cv::Mat x3Dc = Rcw * x3Dw + tcw;
        const float xc = x3Dc.at<float>(0);
        const float yc = x3Dc.at<float>(1);
        const float invzc = 1.0 / x3Dc.at<float>(2);
        const float u = CurrentFrame.fx * xc * invzc + CurrentFrame.cx;
        const float v = CurrentFrame.fy * yc * invzc + CurrentFrame.cy;
        if (u < CurrentFrame.mnMinX || u > CurrentFrame.mnMaxX) continue;
        if (v < CurrentFrame.mnMinY || v > CurrentFrame.mnMaxY) continue;
        cv::Mat PO = x3Dw - Ow;
        float dist3D = cv::norm(PO);
        const float maxDistance = pMP->GetMaxDistanceInvariance();
        const float minDistance = pMP->GetMinDistanceInvariance();
        if (dist3D < minDistance || dist3D > maxDistance) continue;
        const int nPredictedLevel = pMP->PredictScale(dist3D, &CurrentFrame);
        gfo_proj_query e;
        e.u = u;
        e.v = v;
        e.ur = 0.f;
        e.radius = th * CurrentFrame.mvScaleFactors[nPredictedLevel];       // :1650
        e.min_level = nPredictedLevel - 1;
        e.max_level = nPredictedLevel + 1;
        e.angle = pKF->mvKeysUn[i].angle;                                    // :1689
        e.flags = 1 | 4;
        q.push_back(e);
        qmp.push_back(pMP);
    }
    const int M = (int)q.size(), N = CurrentFrame.N;
    cv::Mat qDesc(M > 0 ? M : 1, 32, CV_8U);
    for (int i = 0; i < M; i++) descriptor_row(qmp[i], qDesc, i);
    std::vector<uint8_t> taken(N);
    std::vector<float> angle(N);
    for (int i = 0; i < N; i++) {
        taken[i] = CurrentFrame.mvpMapPoints[i] != NULL;
        angle[i] = CurrentFrame.mvKeysUn[i].angle;
    }
    gfo_frame_bounds fb = {Frame::mnMinX, Frame::mnMinY, Frame::mnMaxX, Frame::mnMaxY};
    gfo_proj_mode mode = {0, 0.f, ORBdist, mbCheckOrientation ? 1 : 0, 0};     // (this overload has no budget in the reference either)
    std::vector<int32_t> outQ(N), outScore(N);
    int nmatches = 0;
    cv::Mat keep;
    GfoUse use(CurrentFrame.mpORBextractorLeft);
    gfo_ctx* c = use.c;
    const int rc = gfo_search_by_projection_queries(c, as_gfo(CurrentFrame.mvKeysUn), rows32(CurrentFrame.mDescriptors, keep), NULL,
                                                    angle.data(), N, &fb, q.data(), qDesc.data, M, &mode, taken.data(), outQ.data(),
                                                    outScore.data(), &nmatches);
    if (rc != GFO_OK) {
        report(c, "SearchByProjection(Cur, KF)");
        return 0;
    }
    for (int i = 0; i < N; i++) {
        if (outQ[i] >= 0) CurrentFrame.mvpMapPoints[i] = qmp[outQ[i]];
        else if (outQ[i] == -2) CurrentFrame.mvpMapPoints[i] = NULL;         // written by this call, cleared by its rotation check (:1713)
    }
    return nmatches;
}
#endif

// ---------------------------------------------------------------------------------------------------------------
#ifdef GFO_ADAPTER_BOW
int ORBmatcher::SearchByBoW(KeyFrame* pKF, Frame& F, std::vector<MapPoint*>& vpMapPointMatches)
{
    const std::vector<MapPoint*> vpMapPointsKF = pKF->GetMapPointMatches();
    vpMapPointMatches = std::vector<MapPoint*>(F.N, static_cast<MapPoint*>(NULL));
    const int nKF = (int)vpMapPointsKF.size(), nF = F.N;
    FlatFeatVec kfv(pKF->mFeatVec), ffv(F.mFeatVec);
    std::vector<uint8_t> valid(nKF);
    std::vector<float> kfAngle(nKF), fAngle(nF);
    for (int i = 0; i < nKF; i++) {
        MapPoint* pMP = vpMapPointsKF[i];
        valid[i] = pMP && !pMP->isBad();                 // :303-309
        kfAngle[i] = pKF->mvKeysUn[i].angle;             // :348
    }
    for (int i = 0; i < nF; i++) fAngle[i] = F.mvKeys[i].angle;   // :352
    std::vector<int32_t> out(nF > 0 ? nF : 1);
    int nmatches = 0;
    cv::Mat keepK, keepF;
    GfoUse use(F.mpORBextractorLeft);
    gfo_ctx* c = use.c;
    const int rc = gfo_search_by_bow_budget(c, rows32(pKF->mDescriptors, keepK), kfAngle.data(), valid.data(), nKF, &kfv.view,
                                            rows32(F.mDescriptors, keepF), fAngle.data(), nF, &ffv.view, mfNNratio,
                                            mbCheckOrientation ? 1 : 0, GFO_FEATURE_BUDGET, out.data(), &nmatches);
    if (rc != GFO_OK) {
        report(c, "SearchByBoW");
        return 0;
    }
    for (int i = 0; i < nF; i++)
        if (out[i] >= 0) vpMapPointMatches[i] = vpMapPointsKF[out[i]];      // :343
    return nmatches;
}
#endif

// ---------------------------------------------------------------------------------------------------------------
#if defined(GFO_ADAPTER_PROJ_SCW) || defined(GFO_ADAPTER_FUSE_SCW)
namespace
{
// Loop closing projects candidate map points into a keyframe under a similarity Scw (ORBmatcher.cc:406-518 and :1089-1212 share this
// part word for word, but for the literal `1 / z` against `1.0 / z`): the decomposition (:415-419 = :1098-1102) and, per point, the
// visibility chain up to the search window (:441-477 = :1124-1161).  All of it is the reference's own cv::Mat arithmetic, on the host.
struct ScwPose {
    cv::Mat Rcw, tcw, Ow;
    explicit ScwPose(const cv::Mat& Scw)
    {
        cv::Mat sRcw = Scw.rowRange(0, 3).colRange(0, 3);
        const float scw = sqrt(sRcw.row(0).dot(sRcw.row(0)));
        Rcw = sRcw / scw;
        tcw = Scw.rowRange(0, 3).col(3) / scw;
        Ow = -Rcw.t() * tcw;
    }
};
// false: the reference `continue`s.  DOUBLE_INV: Fuse writes `1.0 / z` (a double quotient rounded to float), SearchByProjection `1 / z`
template <bool DOUBLE_INV, class TH>
bool project_under_scw(KeyFrame* pKF, MapPoint* pMP, const ScwPose& S, const TH th, gfo_proj_query& e)
{
    const float &fx = pKF->fx, &fy = pKF->fy, &cx = pKF->cx, &cy = pKF->cy;
    cv::Mat p3Dw = pMP->GetWorldPos();
    cv::Mat p3Dc = S.Rcw * p3Dw + S.tcw;
    if (p3Dc.at<float>(2) < 0.0f) return false;                       // depth must be positive
    const float invz = DOUBLE_INV ? (float)(1.0 / p3Dc.at<float>(2)) : 1 / p3Dc.at<float>(2);
    const float x = p3Dc.at<float>(0) * invz;
    const float y = p3Dc.at<float>(1) * invz;
    const float u = fx * x + cx;
    const float v = fy * y + cy;
    if (!pKF->IsInImage(u, v)) return false;
    const float maxDistance = pMP->GetMaxDistanceInvariance();
    const float minDistance = pMP->GetMinDistanceInvariance();
    cv::Mat PO = p3Dw - S.Ow;
    const float dist = cv::norm(PO);
    if (dist < minDistance || dist > maxDistance) return false;       // inside the scale invariance region of the point
    cv::Mat Pn = pMP->GetNormal();
    if (PO.dot(Pn) < 0.5 * dist) return false;                        // viewing angle below 60 degrees
    const int nPredictedLevel = pMP->PredictScale(dist, pKF);
    e.u = u;
    e.v = v;
    e.ur = 0.f;
    e.radius = th * pKF->mvScaleFactors[nPredictedLevel];
    e.min_level = nPredictedLevel - 1;                                // kpLevel < nPredictedLevel - 1 || kpLevel > nPredictedLevel: skipped
    e.max_level = nPredictedLevel;
    e.angle = 0.f;
    e.flags = 1;
    return true;
}
}  // namespace
#endif

#ifdef GFO_ADAPTER_PROJ_SCW
// LoopClosing::ComputeSim3 / CorrectLoop's matcher (LoopClosing.cc:397): map points seen from the loop candidates, projected into the
// current keyframe; a keypoint that holds a match (vpMatched) is skipped and every new match takes its keypoint (:487, :511), no ratio test,
// TH_LOW.  The calling thread's own device context (no Frame in the call).
int ORBmatcher::SearchByProjection(KeyFrame* pKF, cv::Mat Scw, const std::vector<MapPoint*>& vpPoints, std::vector<MapPoint*>& vpMatched, int th)
{
    const ScwPose S(Scw);
    std::set<MapPoint*> spAlreadyFound(vpMatched.begin(), vpMatched.end());     // (a snapshot: matches made below do not enter it, :422)
    spAlreadyFound.erase(static_cast<MapPoint*>(NULL));
    std::vector<gfo_proj_query> q;
    std::vector<MapPoint*> qmp;
    for (int iMP = 0, iendMP = (int)vpPoints.size(); iMP < iendMP; iMP++) {
        MapPoint* pMP = vpPoints[iMP];
        if (pMP->isBad() || spAlreadyFound.count(pMP)) continue;
        gfo_proj_query e;
        if (!project_under_scw<false>(pKF, pMP, S, th, e)) continue;
        e.flags = 1 | 4;                                                // vpMatched[bestIdx] = pMP hides the keypoint from the points behind
        q.push_back(e);
        qmp.push_back(pMP);
    }
    const int M = (int)q.size(), N = pKF->N;
    if ((int)vpMatched.size() < N) return 0;
    cv::Mat qDesc(M > 0 ? M : 1, 32, CV_8U);
    for (int i = 0; i < M; i++) descriptor_row(qmp[i], qDesc, i);
    std::vector<uint8_t> taken(N);
    for (int i = 0; i < N; i++) taken[i] = vpMatched[i] != NULL;        // :487
    gfo_frame_bounds fb = {(float)pKF->mnMinX, (float)pKF->mnMinY, (float)pKF->mnMaxX, (float)pKF->mnMaxY};
    gfo_proj_mode mode = {0, 0.f, TH_LOW, 0, 0};
    std::vector<int32_t> outQ(N > 0 ? N : 1), outScore(N > 0 ? N : 1);
    int nmatches = 0;
    cv::Mat keep;
    gfo_ctx* c = gfo_context_pin_thread();
    const int rc = gfo_search_by_projection_queries(c, as_gfo(pKF->mvKeysUn), rows32(pKF->mDescriptors, keep), NULL, NULL, N, &fb, q.data(),
                                                    qDesc.data, M, &mode, taken.data(), outQ.data(), outScore.data(), &nmatches);
    if (rc != GFO_OK) report(c, "SearchByProjection(KF, Scw)");
    gfo_context_unpin_thread(c);
    if (rc != GFO_OK) return 0;
    for (int i = 0; i < N; i++)
        if (outQ[i] >= 0) vpMatched[i] = qmp[outQ[i]];                 // :511
    return nmatches;
}
#endif

#ifdef GFO_ADAPTER_FUSE_SCW
// LoopClosing::SearchAndFuse's matcher (LoopClosing.cc:621): the loop's map points projected into a keyframe of the current neighbourhood.
// Every point looks for its best keypoint of the two predicted levels on its own -- nothing it finds hides a keypoint from the next point
// (gfo_search_by_projection_queries_points with queries that block nothing) -- and what is done with the find (:1194-1208) happens here,
// point after point in the vector's order, against the keyframe as the points before have left it.
int ORBmatcher::Fuse(KeyFrame* pKF, cv::Mat Scw, const std::vector<MapPoint*>& vpPoints, float th, std::vector<MapPoint*>& vpReplacePoint)
{
    const ScwPose S(Scw);
    const std::set<MapPoint*> spAlreadyFound = pKF->GetMapPoints();
    const int nPoints = (int)vpPoints.size();
    std::vector<gfo_proj_query> q;
    std::vector<int> qpoint;
    for (int iMP = 0; iMP < nPoints; iMP++) {
        MapPoint* pMP = vpPoints[iMP];
        if (pMP->isBad() || spAlreadyFound.count(pMP)) continue;
        gfo_proj_query e;
        if (!project_under_scw<true>(pKF, pMP, S, th, e)) continue;
        q.push_back(e);
        qpoint.push_back(iMP);
    }
    const int M = (int)q.size(), N = pKF->N;
    cv::Mat qDesc(M > 0 ? M : 1, 32, CV_8U);
    for (int i = 0; i < M; i++) descriptor_row(vpPoints[qpoint[i]], qDesc, i);
    gfo_frame_bounds fb = {(float)pKF->mnMinX, (float)pKF->mnMinY, (float)pKF->mnMaxX, (float)pKF->mnMaxY};
    gfo_proj_mode mode = {0, 0.f, TH_LOW, 0, 0};
    std::vector<int32_t> outQ(N > 0 ? N : 1), outScore(N > 0 ? N : 1), outPoint(M > 0 ? M : 1);
    int nmatches = 0;
    cv::Mat keep;
    gfo_ctx* c = gfo_context_pin_thread();
    const int rc = gfo_search_by_projection_queries_points(c, as_gfo(pKF->mvKeysUn), rows32(pKF->mDescriptors, keep), NULL, NULL, N, &fb, q.data(),
                                                           qDesc.data, M, &mode, NULL, outQ.data(), outScore.data(), outPoint.data(), &nmatches);
    if (rc != GFO_OK) report(c, "Fuse(KF, Scw)");
    gfo_context_unpin_thread(c);
    if (rc != GFO_OK) return 0;
    int nFused = 0;
    for (int i = 0; i < M; i++) {
        if (outPoint[i] < 0) continue;                                  // bestDist > TH_LOW, or nothing in the window
        const int bestIdx = outPoint[i] & 0xFFFF;
        const int iMP = qpoint[i];
        MapPoint* pMP = vpPoints[iMP];
        MapPoint* pMPinKF = pKF->GetMapPoint(bestIdx);                  // :1196: as the points in front of this one have left it
        if (pMPinKF) {
            if (!pMPinKF->isBad()) vpReplacePoint[iMP] = pMPinKF;
        } else {
            pMP->AddObservation(pKF, bestIdx);
            pKF->AddMapPoint(pMP, bestIdx);
        }
        nFused++;
    }
    return nFused;
}
#endif

#ifdef GFO_ADAPTER_FUSE
// LocalMapping::SearchInNeighbors' matcher (LocalMapping.cc:673, 698): the map points of one keyframe projected into a neighbour, a few dozen
// calls per new keyframe.  The search of every point (ORBmatcher.cc:1000-1063: window, two levels, the reprojection-error gate, TH_LOW) depends
// on the keyframe's keypoints only, so all of them are ONE device call (gfo_search_for_fusion); what a find leads to (:1067-1083) is applied
// here, point after point in the vector's order -- Replace() can turn a LATER point of the vector bad, and the reference looks at isBad() /
// IsInKeyFrame() at a point's own turn (:959), so that test is made again where the reference makes it.  (A point that is skipped there never
// becomes searchable again, and nothing a point's turn does changes what a later point's search reads: position, normal, distance range and
// descriptor of a point change only by Replace() INTO it, which puts it into the keyframe.)
int ORBmatcher::Fuse(KeyFrame* pKF, const std::vector<MapPoint*>& vpMapPoints, const float th)
{
    cv::Mat Rcw = pKF->GetRotation();
    cv::Mat tcw = pKF->GetTranslation();
    const float &fx = pKF->fx, &fy = pKF->fy, &cx = pKF->cx, &cy = pKF->cy, &bf = pKF->mbf;
    cv::Mat Ow = pKF->GetCameraCenter();
    const int nMPs = (int)vpMapPoints.size();
    std::vector<gfo_proj_query> q;
    std::vector<int> query_of(nMPs, -1);
    std::vector<MapPoint*> qmp;
    for (int i = 0; i < nMPs; i++) {
        MapPoint* pMP = vpMapPoints[i];
        if (!pMP) continue;
        if (pMP->isBad() || pMP->IsInKeyFrame(pKF)) continue;             // (tested again at the point's turn below)
        cv::Mat p3Dw = pMP->GetWorldPos();
        cv::Mat p3Dc = Rcw * p3Dw + tcw;
        if (p3Dc.at<float>(2) < 0.0f) continue;                           // depth must be positive
        const float invz = 1 / p3Dc.at<float>(2);
        const float x = p3Dc.at<float>(0) * invz;
        const float y = p3Dc.at<float>(1) * invz;
        const float u = fx * x + cx;
        const float v = fy * y + cy;
        if (!pKF->IsInImage(u, v)) continue;
        const float ur = u - bf * invz;
        const float maxDistance = pMP->GetMaxDistanceInvariance();
        const float minDistance = pMP->GetMinDistanceInvariance();
        cv::Mat PO = p3Dw - Ow;
        const float dist3D = cv::norm(PO);
        if (dist3D < minDistance || dist3D > maxDistance) continue;
        cv::Mat Pn = pMP->GetNormal();
        if (PO.dot(Pn) < 0.5 * dist3D) continue;                          // viewing angle below 60 degrees
        const int nPredictedLevel = pMP->PredictScale(dist3D, pKF);
        gfo_proj_query e;
        e.u = u; e.v = v; e.ur = ur;
        e.radius = th * pKF->mvScaleFactors[nPredictedLevel];
        e.min_level = nPredictedLevel - 1;
        e.max_level = nPredictedLevel;
        e.angle = 0.f;
        e.flags = 1;
        query_of[i] = (int)q.size();
        q.push_back(e);
        qmp.push_back(pMP);
    }
    const int M = (int)q.size(), N = pKF->N;
    cv::Mat qDesc(M > 0 ? M : 1, 32, CV_8U);
    for (int i = 0; i < M; i++) descriptor_row(qmp[i], qDesc, i);
    gfo_frame_bounds fb = {(float)pKF->mnMinX, (float)pKF->mnMinY, (float)pKF->mnMaxX, (float)pKF->mnMaxY};
    std::vector<int32_t> outPoint(M > 0 ? M : 1);
    cv::Mat keep;
    gfo_ctx* c = gfo_context_pin_thread();
    const int rc = gfo_search_for_fusion(c, as_gfo(pKF->mvKeysUn), rows32(pKF->mDescriptors, keep), pKF->mvuRight.data(), N, &fb,
                                         pKF->mvInvLevelSigma2.data(), (int)pKF->mvInvLevelSigma2.size(), q.data(), qDesc.data, M, TH_LOW, outPoint.data());
    if (rc != GFO_OK) report(c, "Fuse(KF, MapPoints)");
    gfo_context_unpin_thread(c);
    if (rc != GFO_OK) return 0;
    int nFused = 0;
    for (int i = 0; i < nMPs; i++) {
        MapPoint* pMP = vpMapPoints[i];
        if (!pMP || query_of[i] < 0) continue;
        if (pMP->isBad() || pMP->IsInKeyFrame(pKF)) continue;             // :959, at the point's own turn
        const int found = outPoint[query_of[i]];
        if (found < 0) continue;                                           // nothing within TH_LOW
        const int bestIdx = found & 0xFFFF;
        MapPoint* pMPinKF = pKF->GetMapPoint(bestIdx);
        if (pMPinKF) {
            if (!pMPinKF->isBad()) {
                if (pMPinKF->Observations() > pMP->Observations()) pMP->Replace(pMPinKF);
                else pMPinKF->Replace(pMP);
            }
        } else {
            pMP->AddObservation(pKF, bestIdx);
            pKF->AddMapPoint(pMP, bestIdx);
        }
        nFused++;
    }
    return nFused;
}
#endif

#ifdef GFO_ADAPTER_SIM3
// LoopClosing::ComputeSim3's guided matcher (LoopClosing.cc:345): the map points of each keyframe projected into the other under the
// estimated similarity, every point looking for its most similar keypoint of the two predicted levels on its own (TH_HIGH; nothing blocks),
// and a pair kept only where both directions agree (:1419-1435).  Two device calls (gfo_search_by_projection_queries_points with queries
// that block nothing), the projections (:1258-1298, :1338-1378: the reference's cv::Mat arithmetic, intrinsics of pKF1 in both directions
// as written there) and the agreement on the host.
namespace
{
// one direction: the points of `vpMapPoints` (skipping those flagged in vbAlready) through `proj` into pKFto; vnMatch[i] = keypoint of pKFto or -1
template <class Project>
bool sim3_direction(ORBmatcher* self, KeyFrame* pKFfrom, KeyFrame* pKFto, const std::vector<MapPoint*>& vpMapPoints, const std::vector<bool>& vbAlready,
                    const float th, const float fx, const float fy, const float cx, const float cy, Project proj, std::vector<int>& vnMatch, gfo_ctx* c)
{
    (void)self; (void)pKFfrom;
    const int N = (int)vpMapPoints.size();
    std::vector<gfo_proj_query> q;
    std::vector<int> src;
    std::vector<MapPoint*> qmp;
    for (int i = 0; i < N; i++) {
        MapPoint* pMP = vpMapPoints[i];
        if (!pMP || vbAlready[i]) continue;
        if (pMP->isBad()) continue;
        cv::Mat p3Dw = pMP->GetWorldPos();
        cv::Mat p3Dc = proj(p3Dw);                                        // in the target keyframe's camera
        if (p3Dc.at<float>(2) < 0.0) continue;
        const float invz = 1.0 / p3Dc.at<float>(2);
        const float x = p3Dc.at<float>(0) * invz;
        const float y = p3Dc.at<float>(1) * invz;
        const float u = fx * x + cx;
        const float v = fy * y + cy;
        if (!pKFto->IsInImage(u, v)) continue;
        const float maxDistance = pMP->GetMaxDistanceInvariance();
        const float minDistance = pMP->GetMinDistanceInvariance();
        const float dist3D = cv::norm(p3Dc);
        if (dist3D < minDistance || dist3D > maxDistance) continue;
        const int nPredictedLevel = pMP->PredictScale(dist3D, pKFto);
        gfo_proj_query e;
        e.u = u; e.v = v; e.ur = 0.f;
        e.radius = th * pKFto->mvScaleFactors[nPredictedLevel];
        e.min_level = nPredictedLevel - 1;
        e.max_level = nPredictedLevel;
        e.angle = 0.f;
        e.flags = 1;
        q.push_back(e);
        src.push_back(i);
        qmp.push_back(pMP);
    }
    const int M = (int)q.size(), Nto = pKFto->N;
    cv::Mat qDesc(M > 0 ? M : 1, 32, CV_8U);
    for (int i = 0; i < M; i++) descriptor_row(qmp[i], qDesc, i);
    gfo_frame_bounds fb = {(float)pKFto->mnMinX, (float)pKFto->mnMinY, (float)pKFto->mnMaxX, (float)pKFto->mnMaxY};
    gfo_proj_mode mode = {0, 0.f, ORBmatcher::TH_HIGH, 0, 0};
    std::vector<int32_t> outQ(Nto > 0 ? Nto : 1), outScore(Nto > 0 ? Nto : 1), outPoint(M > 0 ? M : 1);
    int nm = 0;
    cv::Mat keep;
    const int rc = gfo_search_by_projection_queries_points(c, as_gfo(pKFto->mvKeysUn), rows32(pKFto->mDescriptors, keep), NULL, NULL, Nto, &fb, q.data(),
                                                           qDesc.data, M, &mode, NULL, outQ.data(), outScore.data(), outPoint.data(), &nm);
    if (rc != GFO_OK) {
        report(c, "SearchBySim3");
        return false;
    }
    for (int k = 0; k < M; k++)
        if (outPoint[k] >= 0) vnMatch[src[k]] = outPoint[k] & 0xFFFF;
    return true;
}
}  // namespace

int ORBmatcher::SearchBySim3(KeyFrame* pKF1, KeyFrame* pKF2, std::vector<MapPoint*>& vpMatches12, const float& s12, const cv::Mat& R12, const cv::Mat& t12,
                             const float th)
{
    const float &fx = pKF1->fx, &fy = pKF1->fy, &cx = pKF1->cx, &cy = pKF1->cy;
    cv::Mat R1w = pKF1->GetRotation();
    cv::Mat t1w = pKF1->GetTranslation();
    cv::Mat R2w = pKF2->GetRotation();
    cv::Mat t2w = pKF2->GetTranslation();
    cv::Mat sR12 = s12 * R12;
    cv::Mat sR21 = (1.0 / s12) * R12.t();
    cv::Mat t21 = -sR21 * t12;
    const std::vector<MapPoint*> vpMapPoints1 = pKF1->GetMapPointMatches();
    const int N1 = (int)vpMapPoints1.size();
    const std::vector<MapPoint*> vpMapPoints2 = pKF2->GetMapPointMatches();
    const int N2 = (int)vpMapPoints2.size();
    std::vector<bool> vbAlreadyMatched1(N1, false), vbAlreadyMatched2(N2, false);
    for (int i = 0; i < N1; i++) {
        MapPoint* pMP = vpMatches12[i];
        if (pMP) {
            vbAlreadyMatched1[i] = true;
            int idx2 = pMP->GetIndexInKeyFrame(pKF2);
            if (idx2 >= 0 && idx2 < N2) vbAlreadyMatched2[idx2] = true;
        }
    }
    std::vector<int> vnMatch1(N1, -1), vnMatch2(N2, -1);
    gfo_ctx* c = gfo_context_pin_thread();
    bool ok = sim3_direction(this, pKF1, pKF2, vpMapPoints1, vbAlreadyMatched1, th, fx, fy, cx, cy,
                             [&](const cv::Mat& p3Dw) { cv::Mat p3Dc1 = R1w * p3Dw + t1w; return cv::Mat(sR21 * p3Dc1 + t21); }, vnMatch1, c);
    ok = ok && sim3_direction(this, pKF2, pKF1, vpMapPoints2, vbAlreadyMatched2, th, fx, fy, cx, cy,
                              [&](const cv::Mat& p3Dw) { cv::Mat p3Dc2 = R2w * p3Dw + t2w; return cv::Mat(sR12 * p3Dc2 + t12); }, vnMatch2, c);
    gfo_context_unpin_thread(c);
    if (!ok) return 0;
    int nFound = 0;                                                       // :1419-1435
    for (int i1 = 0; i1 < N1; i1++) {
        const int idx2 = vnMatch1[i1];
        if (idx2 >= 0) {
            const int idx1 = vnMatch2[idx2];
            if (idx1 == i1) {
                vpMatches12[i1] = vpMapPoints2[idx2];
                nFound++;
            }
        }
    }
    return nFound;
}
#endif

// ---------------------------------------------------------------------------------------------------------------
#ifdef GFO_ADAPTER_BOW_KF
// Loop closing's matcher (LoopClosing.cc:287): the keyframe-pair overload, ORBmatcher.cc:635-768.  No Frame, hence no extractor to take a
// device context from: the calling thread's own (gfo_context_pin_thread, adapter/ORBextractor_gfo.cc).
int ORBmatcher::SearchByBoW(KeyFrame* pKF1, KeyFrame* pKF2, std::vector<MapPoint*>& vpMatches12)
{
    const std::vector<MapPoint*> vpMapPoints1 = pKF1->GetMapPointMatches();
    const std::vector<MapPoint*> vpMapPoints2 = pKF2->GetMapPointMatches();
    vpMatches12 = std::vector<MapPoint*>(vpMapPoints1.size(), static_cast<MapPoint*>(NULL));   // :647
    const int n1 = (int)vpMapPoints1.size(), n2 = (int)vpMapPoints2.size();
    FlatFeatVec fv1(pKF1->mFeatVec), fv2(pKF2->mFeatVec);
    std::vector<uint8_t> valid1(n1), valid2(n2);
    std::vector<float> angle1(n1), angle2(n2);
    for (int i = 0; i < n1; i++) {
        valid1[i] = vpMapPoints1[i] && !vpMapPoints1[i]->isBad();      // :672-676
        angle1[i] = pKF1->mvKeysUn[i].angle;                           // :722
    }
    for (int i = 0; i < n2; i++) {
        valid2[i] = vpMapPoints2[i] && !vpMapPoints2[i]->isBad();      // :690-696
        angle2[i] = pKF2->mvKeysUn[i].angle;
    }
    std::vector<int32_t> out(n1 > 0 ? n1 : 1);
    int nmatches = 0;
    cv::Mat keep1, keep2;
    gfo_ctx* c = gfo_context_pin_thread();
    const int rc = gfo_search_by_bow_keyframes(c, rows32(pKF1->mDescriptors, keep1), angle1.data(), valid1.data(), n1, &fv1.view,
                                               rows32(pKF2->mDescriptors, keep2), angle2.data(), valid2.data(), n2, &fv2.view, mfNNratio,
                                               mbCheckOrientation ? 1 : 0, out.data(), &nmatches);
    if (rc != GFO_OK) report(c, "SearchByBoW(KF, KF)");
    gfo_context_unpin_thread(c);
    if (rc != GFO_OK) return 0;
    for (int i = 0; i < n1; i++)
        if (out[i] >= 0) vpMatches12[i] = vpMapPoints2[out[i]];      // :717
    return nmatches;
}
#endif

// ---------------------------------------------------------------------------------------------------------------
#ifdef GFO_ADAPTER_TRIANGULATION
// Local mapping's matcher for new map points (LocalMapping::CreateNewMapPoints, LocalMapping.cc:435), ORBmatcher.cc:770-935: keypoints
// WITHOUT a map point of two keyframes under the epipolar constraint.  The epipole is computed here with the same cv::Mat expressions
// the reference uses (:777-783); the sweep, CheckDistEpipolarLine and the rotation histogram run on the device.
int ORBmatcher::SearchForTriangulation(KeyFrame* pKF1, KeyFrame* pKF2, cv::Mat F12, std::vector<std::pair<size_t, size_t> >& vMatchedPairs,
                                       const bool bOnlyStereo)
{
    // the first camera's centre seen from the second: the same cv::Mat product, sum and float expressions as :777-783, so the two
    // coordinates carry the same roundings whatever OpenCV's small-matrix code does
    const cv::Mat centre1_in_2 = pKF2->GetRotation() * pKF1->GetCameraCenter() + pKF2->GetTranslation();
    const float inv_depth = 1.0f / centre1_in_2.at<float>(2);
    const float ex = pKF2->fx * centre1_in_2.at<float>(0) * inv_depth + pKF2->cx;
    const float ey = pKF2->fy * centre1_in_2.at<float>(1) * inv_depth + pKF2->cy;

    vMatchedPairs.clear();
    const int n1 = pKF1->N, n2 = pKF2->N;
    if (n1 <= 0 || n2 <= 0) return 0;
    float f12[9];
    for (int r = 0; r < 3; r++)
        for (int q = 0; q < 3; q++) f12[3 * r + q] = F12.at<float>(r, q);
    FlatFeatVec fv1(pKF1->mFeatVec), fv2(pKF2->mFeatVec);
    // GetMapPoint(idx) != NULL for every keypoint: one locked copy of each keyframe's vector instead of a lock per keypoint (the
    // reference reads them one at a time while tracking may add points; either is a snapshot local mapping tolerates, :812, :835)
    const std::vector<MapPoint*> mp1 = pKF1->GetMapPointMatches(), mp2 = pKF2->GetMapPointMatches();
    std::vector<uint8_t> has1(n1, 0), has2(n2, 0);
    for (int i = 0; i < n1 && i < (int)mp1.size(); i++) has1[i] = mp1[i] != NULL;
    for (int i = 0; i < n2 && i < (int)mp2.size(); i++) has2[i] = mp2[i] != NULL;
    std::vector<int32_t> out(n1);
    int nmatches = 0;
    cv::Mat keep1, keep2;
    gfo_ctx* c = gfo_context_pin_thread();
    const int rc = gfo_search_for_triangulation(c, as_gfo(pKF1->mvKeysUn), rows32(pKF1->mDescriptors, keep1), has1.data(), pKF1->mvuRight.data(), n1,
                                                &fv1.view, as_gfo(pKF2->mvKeysUn), rows32(pKF2->mDescriptors, keep2), has2.data(),
                                                pKF2->mvuRight.data(), n2, &fv2.view, pKF2->mvScaleFactors.data(), pKF2->mvLevelSigma2.data(),
                                                (int)pKF2->mvScaleFactors.size(), f12, ex, ey, bOnlyStereo ? 1 : 0, mbCheckOrientation ? 1 : 0,
                                                out.data(), &nmatches);
    if (rc != GFO_OK) report(c, "SearchForTriangulation");
    gfo_context_unpin_thread(c);
    if (rc != GFO_OK) return 0;
    vMatchedPairs.reserve(nmatches > 0 ? nmatches : 0);                    // :922-930
    for (int i = 0; i < n1; i++)
        if (out[i] >= 0) vMatchedPairs.push_back(std::make_pair((size_t)i, (size_t)out[i]));
    return nmatches;
}
#endif

// ---------------------------------------------------------------------------------------------------------------
#ifdef GFO_ADAPTER_INIT
// The monocular bootstrap's matcher (Tracking::MonocularInitialization, Tracking.cc:1322), ORBmatcher.cc:520-633.  One call: the windows
// of F2's grid around vbPrevMatched and every candidate's distance on the device, the ordered pass (a closer keypoint takes a match from an
// earlier one) inside the library; vbPrevMatched is updated in place as :626-629 do.
int ORBmatcher::SearchForInitialization(Frame& F1, Frame& F2, std::vector<cv::Point2f>& vbPrevMatched, std::vector<int>& vnMatches12, int windowSize)
{
    static_assert(sizeof(cv::Point2f) == 2 * sizeof(float), "vbPrevMatched is handed over as (x, y) pairs");
    static_assert(sizeof(int) == sizeof(int32_t), "vnMatches12 is handed over as int32");
    const int n1 = (int)F1.mvKeysUn.size(), n2 = (int)F2.mvKeysUn.size();
    vnMatches12 = std::vector<int>(n1, -1);                               // :523
    if (n1 == 0 || n2 == 0) return 0;
    if ((int)vbPrevMatched.size() < n1) {                                 // (the reference indexes it unchecked, :538)
        fprintf(stderr, "[gfo] SearchForInitialization: vbPrevMatched has %zu entries for %d keypoints\n", vbPrevMatched.size(), n1);
        return 0;
    }
    gfo_frame_bounds fb = {Frame::mnMinX, Frame::mnMinY, Frame::mnMaxX, Frame::mnMaxY};
    cv::Mat keep1, keep2;
    int nmatches = 0;
    GfoUse use(F2.mpORBextractorLeft);
    const int rc = gfo_search_for_initialization(use.c, as_gfo(F1.mvKeysUn), rows32(F1.mDescriptors, keep1), n1, reinterpret_cast<float*>(vbPrevMatched.data()),
                                                 as_gfo(F2.mvKeysUn), rows32(F2.mDescriptors, keep2), n2, &fb, windowSize, mfNNratio,
                                                 mbCheckOrientation ? 1 : 0, reinterpret_cast<int32_t*>(vnMatches12.data()), &nmatches);
    if (rc != GFO_OK) {
        report(use.c, "SearchForInitialization");
        std::fill(vnMatches12.begin(), vnMatches12.end(), -1);
        return 0;
    }
    return nmatches;
}
#endif

// ---------------------------------------------------------------------------------------------------------------
#ifdef GFO_ADAPTER_COMPUTE_BOW
namespace
{
// TemplatedVocabulary keeps its tree in protected members (Thirdparty/DBoW2/DBoW2/TemplatedVocabulary.h:428-441);
// a derived type reads them.  The tree is flattened breadth first so that the children of a node are consecutive
// (gfo_vocabulary), keeping the vocabulary's own child order (ties in the descent keep the FIRST minimum, :1251-1260).
struct VocabularyView : public ORBVocabulary {
    struct Flat {
        uint64_t print;      // fingerprint() of the vocabulary this copy was made from
        std::vector<int32_t> first_child, n_children, word_id, orig_id;
        std::vector<float> weight;
        std::vector<double> weight64;
        std::vector<uint8_t> desc;
        int depth;
    };
    void flatten(Flat& f) const
    {
        const size_t n = m_nodes.size();
        f.first_child.assign(n, 0); f.n_children.assign(n, 0); f.word_id.assign(n, 0); f.orig_id.assign(n, 0);
        f.weight.assign(n, 0.f); f.weight64.assign(n, 0.0); f.desc.assign(n * 32, 0);
        f.depth = m_L;
        std::vector<DBoW2::NodeId> order(1, 0);                      // new index -> original NodeId, root first
        for (size_t head = 0; head < order.size(); head++) {
            const Node& nd = m_nodes[order[head]];
            f.first_child[head] = (int32_t)order.size();
            f.n_children[head] = (int32_t)nd.children.size();
            for (size_t k = 0; k < nd.children.size(); k++) order.push_back(nd.children[k]);
        }
        for (size_t i = 0; i < order.size() && i < n; i++) {
            const Node& nd = m_nodes[order[i]];
            f.orig_id[i] = (int32_t)nd.id;
            f.word_id[i] = (int32_t)nd.word_id;
            f.weight[i] = (float)nd.weight;
            f.weight64[i] = nd.weight;                         // WordValue is a double: the fold sums it unrounded
            if (i > 0 && nd.descriptor.cols == 32) memcpy(&f.desc[i * 32], nd.descriptor.data, 32);
        }
    }
    // A cheap identity of the vocabulary's CONTENT: the flattened copy is cached per object address, and an application that deletes
    // a vocabulary and loads another may get the same address back (round 5: tests/host/adapter_run.cc builds four vocabularies in
    // one stack slot).  Shape, parameters and a sample of 16 nodes (weight, word id, eight descriptor bytes): a few hundred bytes per
    // ComputeBoW call, against re-reading a million nodes.  (Round 6, ORBvoc's size -- 1 111 111 Node objects, a quarter of a
    // gigabyte: every sampled node is three dependent cache misses; 64 samples were 45 us of a 119-us call, 16 are what a call can
    // afford.  Shape, parameters and the word count are in the hash whatever the samples say.)
    uint64_t fingerprint() const
    {
        uint64_t hsh = 1469598103934665603ull;
        auto mix = [&](uint64_t v) { hsh = (hsh ^ v) * 1099511628211ull; };
        const size_t n = m_nodes.size();
        mix(n); mix((uint64_t)m_k); mix((uint64_t)m_L); mix((uint64_t)m_weighting); mix((uint64_t)m_scoring); mix(m_words.size());
        const size_t step = n > 16 ? n / 16 : 1;
        for (size_t i = 0; i < n; i += step) {
            const Node& nd = m_nodes[i];
            uint64_t w;
            memcpy(&w, &nd.weight, 8);
            mix(w); mix((uint64_t)nd.word_id); mix(nd.children.size());
            if (nd.descriptor.cols == 32 && nd.descriptor.data) { uint64_t d8; memcpy(&d8, nd.descriptor.data, 8); mix(d8); }
        }
        return hsh;
    }
    using ORBVocabulary::m_weighting;
    using ORBVocabulary::m_scoring_object;
};

// One flattened copy per vocabulary (shared by every context), and per CONTEXT ID which vocabulary it holds.  A context's
// id is never reused (gfo_ctx_id), so an extractor re-created at the address of a deleted one (Tracking::updateORBExtractor,
// src/Tracking.cc:298-320) -- new context, same pointer -- is seen as holding nothing and gets its upload; and the context
// itself has the last word (gfo_vocabulary_nodes): nothing here can claim residency the device does not have.
std::mutex g_voc_mu;
std::map<const ORBVocabulary*, VocabularyView::Flat> g_voc_flat;
std::map<uint64_t, const ORBVocabulary*> g_voc_on_ctx;
}  // namespace

void Frame::ComputeBoW()
{
    if (!mBowVec.empty()) return;
    mFeatVec.clear();
    if (mpORBvocabulary->empty() || N == 0) return;
    GfoUse use(mpORBextractorLeft);
    gfo_ctx* c = use.c;
    if (!c) return;
    const VocabularyView::Flat* flat;
    {
        std::lock_guard<std::mutex> lk(g_voc_mu);
        const uint64_t print = static_cast<const VocabularyView*>(mpORBvocabulary)->fingerprint();
        std::map<const ORBVocabulary*, VocabularyView::Flat>::iterator it = g_voc_flat.find(mpORBvocabulary);
        bool fresh = false;
        if (it == g_voc_flat.end() || it->second.print != print) {      // first sight, or another vocabulary at a known address
            if (it == g_voc_flat.end()) it = g_voc_flat.insert(std::make_pair(mpORBvocabulary, VocabularyView::Flat())).first;
            static_cast<const VocabularyView*>(mpORBvocabulary)->flatten(it->second);
            it->second.print = print;
            fresh = true;
            for (std::map<uint64_t, const ORBVocabulary*>::iterator o = g_voc_on_ctx.begin(); o != g_voc_on_ctx.end();)   // what the contexts hold of this address is stale
                if (o->second == mpORBvocabulary) g_voc_on_ctx.erase(o++); else ++o;
        }
        flat = &it->second;
        const uint64_t id = gfo_ctx_id(c);
        std::map<uint64_t, const ORBVocabulary*>::iterator on = g_voc_on_ctx.find(id);
        const bool resident = !fresh && on != g_voc_on_ctx.end() && on->second == mpORBvocabulary &&
                              gfo_vocabulary_nodes(c) == (int)flat->first_child.size();
        if (!resident) {
            gfo_vocabulary v = {flat->first_child.data(), flat->n_children.data(), flat->desc.data(), flat->word_id.data(),
                                flat->weight.data(), (int32_t)flat->first_child.size(), flat->depth, flat->weight64.data()};
            if (gfo_vocabulary_upload(c, &v) != GFO_OK) {
                report(c, "ComputeBoW (vocabulary upload)");
                return;
            }
            if (g_voc_on_ctx.size() > 4096) g_voc_on_ctx.clear();   // ids of long-gone contexts; the survivors re-upload once
            g_voc_on_ctx[id] = mpORBvocabulary;
        }
    }
    const VocabularyView* voc = static_cast<const VocabularyView*>(mpORBvocabulary);
    DBoW2::LNorm norm;
    const bool must = voc->m_scoring_object->mustNormalize(norm);
    cv::Mat keep;
    const uint8_t* desc = rows32(mDescriptors, keep);
    // descent AND fold on the device, whatever N (beyond 8192 descriptors the fold sorts in device memory instead of LDS): both
    // maps come back flattened in std::map order (TemplatedVocabulary.h:1140-1212)
    std::vector<uint32_t> bw(N), fn(N), fi(N);
    std::vector<double> bv(N);
    std::vector<int32_t> fs(N + 1);
    int nw = 0, nf = 0;
    gfo_bow_mode mode = {(int32_t)voc->m_weighting, must ? (norm == DBoW2::L1 ? 1 : 2) : 0};
    if (gfo_compute_bow(c, desc, N, 4, &mode, bw.data(), bv.data(), &nw, fn.data(), fs.data(), fi.data(), &nf) != GFO_OK) {   // levelsup = 4, Frame.cc:666
        report(c, "ComputeBoW");
        return;
    }
    for (int k = 0; k < nw; k++) mBowVec.insert(mBowVec.end(), DBoW2::BowVector::value_type((DBoW2::WordId)bw[k], bv[k]));
    for (int j = 0; j < nf; j++) {
        std::vector<unsigned int>& items = mFeatVec[(DBoW2::NodeId)flat->orig_id[fn[j]]];
        items.assign(fi.begin() + fs[j], fi.begin() + fs[j + 1]);
    }
}
#endif

}  // namespace ORB_SLAM2
