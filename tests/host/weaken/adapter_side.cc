// stands for adapter/matchers_gfo.cc: the same twelve members, every body answers 2
#include "decl.h"
namespace ORB_SLAM2
{
int Frame::ComputeStereoMatches_Undistorted(bool) { return 2; }
void Frame::ComputeBoW() { bow = 2; }
int ORBmatcher::SearchByProjection(Frame&, const std::vector<MapPoint*>&, const float) { return 2; }
int ORBmatcher::SearchByProjection_Budget(Frame&, const std::vector<MapPoint*>&, const float, const double) { return 2; }
int ORBmatcher::SearchByProjection(Frame&, const Frame&, const float, const bool, double&) { return 2; }
int ORBmatcher::SearchByProjection(Frame&, KeyFrame*, const std::set<MapPoint*>&, const float, const int) { return 2; }
int ORBmatcher::SearchByBoW(KeyFrame*, Frame&, std::vector<MapPoint*>&) { return 2; }
int ORBmatcher::SearchByBoW(KeyFrame*, KeyFrame*, std::vector<MapPoint*>&) { return 2; }
int ORBmatcher::SearchByProjection(KeyFrame*, cv::Mat, const std::vector<MapPoint*>&, std::vector<MapPoint*>&, int) { return 2; }
int ORBmatcher::Fuse(KeyFrame*, cv::Mat, const std::vector<MapPoint*>&, float, std::vector<MapPoint*>&) { return 2; }
int ORBmatcher::Fuse(KeyFrame*, const std::vector<MapPoint*>&, const float) { return 2; }
int ORBmatcher::SearchBySim3(KeyFrame*, KeyFrame*, std::vector<MapPoint*>&, const float&, const cv::Mat&, const cv::Mat&, const float) { return 2; }
int ORBmatcher::SearchForTriangulation(KeyFrame*, KeyFrame*, cv::Mat, std::vector<std::pair<std::size_t, std::size_t> >&, const bool) { return 2; }
int ORBmatcher::SearchForInitialization(Frame&, Frame&, std::vector<cv::Point2f>&, std::vector<int>&, int) { return 2; }
}  // namespace ORB_SLAM2
