// adapter_link_support.cc -- TEST INFRASTRUCTURE for tests/host/adapter_run.cc, nothing else.
//
// adapter_run links gf-orb-slam2_amd/adapter/*.cc (our code) against the reference's UNCHANGED headers.  The reference's own
// src/Frame.cc, MapPoint.cc, ORBmatcher.cc cannot be compiled in this image (OpenCV 3.4.1, Armadillo, g2o, DBoW2's cv-dependent
// files are absent) and are NOT compiled, copied or imitated here.  What the link still needs are the few out-of-line members
// and static data members the adapters and the harness name; they are given the smallest bodies that carry the data the
// harness put into the objects -- getters return the member, the constructor does nothing.  They are not restatements of the
// reference's algorithms, pin nothing and are never part of the product or of the oracle.
#include "Frame.h"
#include "KeyFrame.h"
#include "MapPoint.h"
#include "ORBmatcher.h"

// ---- DBoW2: an ORBVocabulary OBJECT (the harness builds one for Frame::ComputeBoW) instantiates every virtual member of
//      TemplatedVocabulary -- create / transform / save / load -- and with them names FORB's descriptor functions (FORB.cpp) and the six
//      scoring classes' score() (ScoringObject.cpp), neither of which this image can build.  The adapter never calls any of them (the
//      descent and the fold run on the device); the definitions below exist for the linker and abort if they are ever reached.
namespace DBoW2
{
static void unreachable(const char* what)
{
    fprintf(stderr, "[adapter_link_support] %s was called: it is a link stub, not DBoW2\n", what);
    abort();
}
int FORB::distance(const cv::Mat&, const cv::Mat&) { unreachable("FORB::distance"); return 0; }
void FORB::meanValue(const std::vector<FORB::pDescriptor>&, FORB::TDescriptor&) { unreachable("FORB::meanValue"); }
std::string FORB::toString(const FORB::TDescriptor&) { unreachable("FORB::toString"); return std::string(); }
void FORB::fromString(FORB::TDescriptor&, const std::string&) { unreachable("FORB::fromString"); }
double L1Scoring::score(const BowVector&, const BowVector&) const { unreachable("L1Scoring::score"); return 0; }
double L2Scoring::score(const BowVector&, const BowVector&) const { unreachable("L2Scoring::score"); return 0; }
double ChiSquareScoring::score(const BowVector&, const BowVector&) const { unreachable("ChiSquareScoring::score"); return 0; }
double KLScoring::score(const BowVector&, const BowVector&) const { unreachable("KLScoring::score"); return 0; }
double BhattacharyyaScoring::score(const BowVector&, const BowVector&) const { unreachable("BhattacharyyaScoring::score"); return 0; }
double DotProductScoring::score(const BowVector&, const BowVector&) const { unreachable("DotProductScoring::score"); return 0; }
}  // namespace DBoW2

namespace ORB_SLAM2
{

// ---- Frame: static data members (src/Frame.cc defines them) and the members the harness / the adapters call -----------------
long unsigned int Frame::nNextId = 0;
bool Frame::mbInitialComputations = true;
float Frame::cx, Frame::cy, Frame::fx, Frame::fy, Frame::invfx, Frame::invfy;
float Frame::mnMinX, Frame::mnMinY, Frame::mnMaxX, Frame::mnMaxY;
float Frame::mfGridElementWidthInv, Frame::mfGridElementHeightInv;

Frame::Frame() {}

void Frame::SetPose(cv::Mat Tcw)
{
    mTcw = Tcw.clone();
    UpdatePoseMatrices();
}

void Frame::UpdatePoseMatrices()
{
    mRcw = mTcw.rowRange(0, 3).colRange(0, 3);
    mRwc = mRcw.t();
    mtcw = mTcw.rowRange(0, 3).col(3);
    mOw = -mRcw.t() * mtcw;
}

// ---- MapPoint: the four accessors the matcher adapters flatten (include/MapPoint.h:60-96) -------------------------------------
long unsigned int MapPoint::nNextId = 0;
std::mutex MapPoint::mGlobalMutex;

// For the keyframe overload (ORBmatcher.cc:1595-1721) the adapter asks a map point for its distance range and for the pyramid level
// it predicts at a distance.  The reference computes those from the point's observation history (MapPoint.cc); here they simply
// hand back what the harness stored in the object -- the adapter's flattening is what is under test, not that arithmetic.
float MapPoint::GetMinDistanceInvariance() { return mfMinDistance; }
float MapPoint::GetMaxDistanceInvariance() { return mfMaxDistance; }
int MapPoint::PredictScale(const float&, Frame*) { return mnTrackScaleLevel; }
int MapPoint::PredictScale(const float&, KeyFrame*) { return mnTrackScaleLevel; }
cv::Mat MapPoint::GetNormal() { return mNormalVector.clone(); }
bool MapPoint::IsInKeyFrame(KeyFrame* pKF) { return mObservations.count(pKF) != 0; }
MapPoint* MapPoint::GetReplaced() { return mpReplaced; }
int MapPoint::GetIndexInKeyFrame(KeyFrame* pKF) { return mObservations.count(pKF) ? (int)mObservations[pKF] : -1; }
// MapPoint::Replace as far as ORBmatcher::Fuse can see it afterwards: this point is bad and names its replacement, its observations move to
// the replacement (or the keyframe's slot is cleared where the replacement is there already).  The statistics, the descriptor recomputation
// and the Map's bookkeeping of src/MapPoint.cc:332-371 are not the adapter's to test.
void MapPoint::Replace(MapPoint* pMP)
{
    if (pMP == this) return;
    std::map<KeyFrame*, size_t> obs = mObservations;
    mObservations.clear();
    mbBad = true;
    mpReplaced = pMP;
    for (std::map<KeyFrame*, size_t>::iterator it = obs.begin(); it != obs.end(); ++it) {
        if (!pMP->IsInKeyFrame(it->first)) {
            it->first->ReplaceMapPointMatch(it->second, pMP);
            pMP->AddObservation(it->first, it->second);
        } else
            it->first->EraseMapPointMatch(it->second);
    }
}
void MapPoint::AddObservation(KeyFrame* pKF, size_t idx)
{
    if (mObservations.count(pKF)) return;
    mObservations[pKF] = idx;
    nObs++;
}

cv::Mat MapPoint::GetWorldPos() { return mWorldPos.clone(); }
cv::Mat MapPoint::GetDescriptor() { return mDescriptor.clone(); }
int MapPoint::Observations() { return nObs; }
void MapPoint::IncreaseFound(int n) { mnFound += n; }
bool MapPoint::isBad() { return mbBad; }

// ---- KeyFrame: a keyframe is a frame's arrays frozen (include/KeyFrame.h:165-268: const members copied from the Frame) ---------
long unsigned int KeyFrame::nNextId = 0;
KeyFrame::KeyFrame(Frame& F, Map* pMap, KeyFrameDatabase* pKFDB)
    : mnFrameId(F.mnId), mTimeStamp(F.mTimeStamp), mnGridCols(FRAME_GRID_COLS), mnGridRows(FRAME_GRID_ROWS),
      mfGridElementWidthInv(F.mfGridElementWidthInv), mfGridElementHeightInv(F.mfGridElementHeightInv), fx(F.fx), fy(F.fy), cx(F.cx), cy(F.cy),
      invfx(F.invfx), invfy(F.invfy), mbf(F.mbf), mb(F.mb), mThDepth(F.mThDepth), N(F.N), mvKeys(F.mvKeys), mvKeysUn(F.mvKeysUn), mvuRight(F.mvuRight),
      mvDepth(F.mvDepth), mDescriptors(F.mDescriptors.clone()), mBowVec(F.mBowVec), mFeatVec(F.mFeatVec), mnScaleLevels(F.mnScaleLevels),
      mfScaleFactor(F.mfScaleFactor), mfLogScaleFactor(F.mfLogScaleFactor), mvScaleFactors(F.mvScaleFactors), mvLevelSigma2(F.mvLevelSigma2),
      mvInvLevelSigma2(F.mvInvLevelSigma2), mnMinX(F.mnMinX), mnMinY(F.mnMinY), mnMaxX(F.mnMaxX), mnMaxY(F.mnMaxY), mK(F.mK),
      mvpMapPoints(F.mvpMapPoints), mpKeyFrameDB(pKFDB), mpORBvocabulary(F.mpORBvocabulary), mbFirstConnection(true), mpParent(NULL), mbNotErase(false),
      mbToBeErased(false), mbBad(false), mHalfBaseline(F.mb / 2), mpMap(pMap)
{
    mnId = nNextId++;
}
std::vector<MapPoint*> KeyFrame::GetMapPointMatches() { return mvpMapPoints; }
std::set<MapPoint*> KeyFrame::GetMapPoints()
{
    std::set<MapPoint*> s;
    for (size_t i = 0; i < mvpMapPoints.size(); i++)
        if (mvpMapPoints[i] && !mvpMapPoints[i]->isBad()) s.insert(mvpMapPoints[i]);
    return s;
}
MapPoint* KeyFrame::GetMapPoint(const size_t& idx) { return mvpMapPoints[idx]; }
void KeyFrame::AddMapPoint(MapPoint* pMP, const size_t& idx) { mvpMapPoints[idx] = pMP; }
void KeyFrame::ReplaceMapPointMatch(const size_t& idx, MapPoint* pMP) { mvpMapPoints[idx] = pMP; }
void KeyFrame::EraseMapPointMatch(const size_t& idx) { mvpMapPoints[idx] = static_cast<MapPoint*>(NULL); }
cv::Mat KeyFrame::GetRotation() { return Tcw.rowRange(0, 3).colRange(0, 3).clone(); }
cv::Mat KeyFrame::GetTranslation() { return Tcw.rowRange(0, 3).col(3).clone(); }
cv::Mat KeyFrame::GetCameraCenter() { return Ow.clone(); }
bool KeyFrame::IsInImage(const float& x, const float& y) const { return x >= mnMinX && x < mnMaxX && y >= mnMinY && y < mnMaxY; }

// ---- ORBmatcher: constructor and the three public constants (include/ORBmatcher.h:46,294-296) --------------------------------
const int ORBmatcher::TH_HIGH = 100;
const int ORBmatcher::TH_LOW = 50;
const int ORBmatcher::HISTO_LENGTH = 30;

ORBmatcher::ORBmatcher(float nnratio, bool checkOri) : mfNNratio(nnratio), mbCheckOrientation(checkOri) {}

}  // namespace ORB_SLAM2
