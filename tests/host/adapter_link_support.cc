// adapter_link_support.cc -- TEST INFRASTRUCTURE for tests/host/adapter_run.cc, nothing else.
//
// adapter_run links gf-orb-slam2_amd/adapter/*.cc (our code) against the reference's UNCHANGED headers.  The reference's own
// src/Frame.cc, MapPoint.cc, ORBmatcher.cc cannot be compiled in this image (OpenCV 3.4.1, Armadillo, g2o, DBoW2's cv-dependent
// files are absent) and are NOT compiled, copied or imitated here.  What the link still needs are the few out-of-line members
// and static data members the adapters and the harness name; they are given the smallest bodies that carry the data the
// harness put into the objects -- getters return the member, the constructor does nothing.  They are not restatements of the
// reference's algorithms, pin nothing and are never part of the product or of the oracle.
#include "Frame.h"
#include "MapPoint.h"
#include "ORBmatcher.h"

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

cv::Mat MapPoint::GetWorldPos() { return mWorldPos.clone(); }
cv::Mat MapPoint::GetDescriptor() { return mDescriptor.clone(); }
int MapPoint::Observations() { return nObs; }
bool MapPoint::isBad() { return mbBad; }

// ---- ORBmatcher: constructor and the three public constants (include/ORBmatcher.h:46,294-296) --------------------------------
const int ORBmatcher::TH_HIGH = 100;
const int ORBmatcher::TH_LOW = 50;
const int ORBmatcher::HISTO_LENGTH = 30;

ORBmatcher::ORBmatcher(float nnratio, bool checkOri) : mfNNratio(nnratio), mbCheckOrientation(checkOri) {}

}  // namespace ORB_SLAM2
