// good_feature_matching_gfo.h -- the good-feature selection loops of the reference on the device's candidate table.
//
// GOOD_FEATURE_MAP_MATCHING is the reference's default build (include/Tracking.h:75).  In it Tracking::SearchLocalPoints hands the
// local map to Observability::runActiveMapMatching (src/Observability.cc:830-1100; runBaselineMapMatching :1171-1275 in the two
// baseline builds), which picks map points one at a time -- each pick depends on whether the previous ones matched -- and calls
//     ORBmatcher::SearchByProjection_OnePoint(F, pMP, th)                       include/ORBmatcher.h:71-150
// (GetCandidates + MatchCandidates, :152-250, with INFORMATION_EFFICIENCY_SCORE) for every pick.  Those three are INLINE members of the
// reference's header: no object file defines them, so there is nothing to swap at link time (the other matchers: adapter/matchers_gfo.cc,
// adapter/weaken_symbols.txt).  What a maintainer changes instead is the call site, two lines in src/Observability.cc:
//
//     #include "good_feature_matching_gfo.h"
//     ...
//     ORB_SLAM2::GfoCandidateTable table(*pFrame, *mMapPoints, th, 0.8f);      // before the selection loop: ONE device call
//     ...
//     int bestIdx = table.OnePoint(*pFrame, heapTop.idx);                      // was: mORBMatcher.SearchByProjection_OnePoint(*pFrame, mMapPoints->at(heapTop.idx), th)
//
// (0.8f is the ratio the reference constructs that matcher with, src/Tracking.cc:2326; ORBmatcher keeps it protected.)
//
// The table is ORBmatcher::GetCandidates for every point of the vector at once, with each candidate's descriptor distance
// (gfo_projection_candidates): everything SearchByProjection_OnePoint computes that does not depend on the frame's slots.  OnePoint()
// does what is left, reading the frame LIVE exactly where the reference does -- F.mvpMapPoints[idx]->Observations() (:113-115) and
// F.mvuRight[idx] (:118-123; a DELAYED_STEREO_MATCHING build changes it between picks) -- and writes F.mvpMapPoints / F.mvpMatchScore
// as :143-146 do.  Valid for as long as the frame's keypoints and the points' mTrackProj* / mnTrackScaleLevel / mTrackViewCos /
// mbTrackInView / descriptors stay what they were when the table was built, which is the whole selection loop (isInFrustum has
// run before it, Tracking.cc:2282-2303).
//
// Error behaviour follows the other adapters: a refused device call is reported on stderr and every OnePoint() answers -1.
#ifndef GFO_GOOD_FEATURE_MATCHING_H
#define GFO_GOOD_FEATURE_MATCHING_H

#include <cmath>
#include <cstdio>
#include <cstring>
#include <mutex>
#include <vector>

#include "Frame.h"
#include "MapPoint.h"
#include "ORBmatcher.h"
#include "gfo.h"

namespace ORB_SLAM2
{

gfo_ctx* gfo_context_pin(const ORBextractor* e);   // adapter/ORBextractor_gfo.cc
void gfo_context_unpin(const ORBextractor* e, gfo_ctx* c);

// MapPoint::GetDescriptor() without its temporary: the same 32 bytes under the same mutex (src/MapPoint.cc:464-468 locks mMutexFeatures and
// clones; both members are protected, include/MapPoint.h:173,193 -- a derived type reads them)
struct GfoMapPointView : public MapPoint {
    bool descriptor_into(uint8_t* dst)
    {
        std::unique_lock<std::mutex> lock(mMutexFeatures);
        if (!(mDescriptor.data && mDescriptor.rows * mDescriptor.cols >= 32 && mDescriptor.isContinuous())) return false;
        memcpy(dst, mDescriptor.data, 32);
        return true;
    }
};

class GfoCandidateTable
{
public:
    GfoCandidateTable(Frame& F, const std::vector<MapPoint*>& vpMapPoints, const float th, const float nnratio)
        : mpPoints(&vpMapPoints), mTh(th), mfNNratio(nnratio), mbOk(false)
    {
        static_assert(sizeof(cv::KeyPoint) == sizeof(gfo_keypoint), "gfo_keypoint must mirror cv::KeyPoint");
        const int M = (int)vpMapPoints.size(), N = F.N;
        mStart.assign(M + 1, 0);
        if (M == 0 || N == 0) { mbOk = true; return; }
        std::vector<gfo_map_point> mps(M);
        std::vector<uint8_t> desc((size_t)M * 32, 0);
        for (int i = 0; i < M; i++) {
            MapPoint* pMP = vpMapPoints[i];
            gfo_map_point& m = mps[i];
            memset(&m, 0, sizeof m);
            if (!pMP) continue;                                  // (runActiveMapMatching skips NULL entries itself, Observability.cc:872)
            m.proj_x = pMP->mTrackProjX; m.proj_y = pMP->mTrackProjY; m.proj_xr = pMP->mTrackProjXR;
            m.view_cos = pMP->mTrackViewCos; m.level = pMP->mnTrackScaleLevel;
            m.flags = pMP->mbTrackInView ? 1 : 0;                // ORBmatcher.h:75-79: not in view / bad: no candidates
            if (!m.flags) continue;
            if (pMP->isBad()) { m.flags |= 2; continue; }
            if (static_cast<GfoMapPointView*>(pMP)->descriptor_into(&desc[(size_t)i * 32])) continue;
            const cv::Mat d = pMP->GetDescriptor();
            if (d.data && d.rows * d.cols >= 32 && d.isContinuous()) memcpy(&desc[(size_t)i * 32], d.data, 32);
            else { cv::Mat row(1, 32, CV_8U, &desc[(size_t)i * 32]); d.copyTo(row); }
        }
        cv::Mat keep;
        const uint8_t* fdesc = F.mDescriptors.data;
        if (!F.mDescriptors.isContinuous()) { keep = F.mDescriptors.clone(); fdesc = keep.data; }
        gfo_frame_bounds fb = {Frame::mnMinX, Frame::mnMinY, Frame::mnMaxX, Frame::mnMaxY};
        gfo_ctx* c = gfo_context_pin(F.mpORBextractorLeft);
        int cap = 32 * M, total = 0, rc = GFO_OK;
        for (int attempt = 0; attempt < 2; attempt++) {          // a table larger than the guess: the library says how large
            mCand.assign(cap > 0 ? cap : 1, 0);
            rc = gfo_projection_candidates(c, reinterpret_cast<const gfo_keypoint*>(F.mvKeysUn.data()), fdesc, F.mvuRight.data(), N,
                                           F.mvScaleFactors.data(), (int)F.mvScaleFactors.size(), &fb, mps.data(), desc.data(), M, th,
                                           mStart.data(), mCand.data(), cap, &total);
            if (rc != GFO_ERR_CAPACITY) break;
            cap = total;
        }
        if (rc != GFO_OK) {
            fprintf(stderr, "[gfo] GfoCandidateTable: %s\n", gfo_last_error(c));
            mStart.assign(M + 1, 0);
        } else mbOk = true;
        gfo_context_unpin(F.mpORBextractorLeft, c);
    }

    bool ok() const { return mbOk; }

    // pMP->mvMatchCandidates.size() after ORBmatcher::GetCandidates (the cost term of INFORMATION_EFFICIENCY_SCORE, Observability.cc:958)
    size_t Candidates(const size_t i) const { return i + 1 < mStart.size() ? (size_t)(mStart[i + 1] - mStart[i]) : 0; }
    // ... and the list itself, as GetCandidates leaves it in the point
    void GetCandidates(const size_t i, std::vector<size_t>& out) const
    {
        out.clear();
        for (int k = mStart[i]; i + 1 < mStart.size() && k < mStart[i + 1]; k++) out.push_back(mCand[k] & 0xFFFFu);
    }

    // ORBmatcher::SearchByProjection_OnePoint(F, vpMapPoints[i], th) / MatchCandidates(F, vpMapPoints[i], th)
    int OnePoint(Frame& F, const size_t i)
    {
        if (!mbOk || i + 1 >= mStart.size()) return -1;
        MapPoint* pMP = (*mpPoints)[i];
        const int n = mStart[i + 1] - mStart[i];
        if (!pMP || n == 0) return -1;                           // :75-99 (a point without candidates was filtered when the table was built)
        // the two live reads of the candidate loop, :113-123, on this point's entries only
        float r = (double)pMP->mTrackViewCos > 0.998 ? 2.5f : 4.0f;      // RadiusByViewingCos, ORBmatcher.cc:243-249
        if (mTh != 1.0f) r *= mTh;
        const float rs = r * F.mvScaleFactors[pMP->mnTrackScaleLevel];
        mEntries.assign(mCand.begin() + mStart[i], mCand.begin() + mStart[i + 1]);
        mTaken.resize(F.N);
        for (int k = 0; k < n; k++) {
            const size_t idx = mEntries[k] & 0xFFFFu;
            mTaken[idx] = F.mvpMapPoints[idx] && F.mvpMapPoints[idx]->Observations() > 0;
            const bool gated = F.mvuRight[idx] > 0 && fabs(pMP->mTrackProjXR - F.mvuRight[idx]) > rs;
            mEntries[k] = (mEntries[k] & 0x7FFFFFFFu) | (gated ? 0x80000000u : 0u);
        }
        int bestDist = 256;
        const int bestIdx = gfo_match_candidates(mEntries.data(), n, mTaken.data(), F.N, mfNNratio, &bestDist);
        if (bestIdx < 0) return -1;
        F.mvpMapPoints[bestIdx] = pMP;                           // :143
        F.mvpMatchScore[bestIdx] = bestDist;                     // :145
        return bestIdx;
    }

private:
    const std::vector<MapPoint*>* mpPoints;
    float mTh, mfNNratio;
    bool mbOk;
    std::vector<int32_t> mStart;
    std::vector<uint32_t> mCand, mEntries;
    std::vector<uint8_t> mTaken;
};

}  // namespace ORB_SLAM2

#endif
