// Declarations with the reference's names and signatures (include/Frame.h:267, include/ORBmatcher.h:64,67,259-272), nothing else of
// the classes: a mangled name depends on the names in a signature only.  tests/test_adapter_compiles.py checks that the symbols
// this miniature defines ARE the ones in adapter/weaken_symbols.txt.
#include <cstddef>
#include <set>
#include <utility>
#include <vector>
namespace cv { class Mat { public: int x; }; }   // (by value in two signatures: only the name enters the mangling)
namespace cv { template <class T> struct Point_ { T x, y; }; typedef Point_<float> Point2f; }   // (cv::Point2f is a typedef of a template: the name that enters the mangling)
namespace ORB_SLAM2
{
class MapPoint;
class KeyFrame;
class Frame
{
public:
    int ComputeStereoMatches_Undistorted(bool isOnline);
    void ComputeBoW();
    int construct();   // stands for Frame::Frame, which calls ComputeStereoMatches_Undistorted from inside Frame.o (Frame.cc:100)
    int bow = 0;
};
class ORBmatcher
{
public:
    int SearchByProjection(Frame& F, const std::vector<MapPoint*>& vpMapPoints, const float th);
    int SearchByProjection_Budget(Frame& F, const std::vector<MapPoint*>& vpMapPoints, const float th, const double time_constr);
    int SearchByProjection(Frame& CurrentFrame, const Frame& LastFrame, const float th, const bool bMono, double& numVisibleMpt);
    int SearchByProjection(Frame& CurrentFrame, KeyFrame* pKF, const std::set<MapPoint*>& sAlreadyFound, const float th, const int ORBdist);
    int SearchByBoW(KeyFrame* pKF, Frame& F, std::vector<MapPoint*>& vpMapPointMatches);
    int SearchByBoW(KeyFrame* pKF1, KeyFrame* pKF2, std::vector<MapPoint*>& vpMatches12);
    int SearchByProjection(KeyFrame* pKF, cv::Mat Scw, const std::vector<MapPoint*>& vpPoints, std::vector<MapPoint*>& vpMatched, int th);
    int Fuse(KeyFrame* pKF, cv::Mat Scw, const std::vector<MapPoint*>& vpPoints, float th, std::vector<MapPoint*>& vpReplacePoint);
    int Fuse(KeyFrame* pKF, const std::vector<MapPoint*>& vpMapPoints, const float th = 3.0);
    int SearchBySim3(KeyFrame* pKF1, KeyFrame* pKF2, std::vector<MapPoint*>& vpMatches12, const float& s12, const cv::Mat& R12, const cv::Mat& t12, const float th);
    int SearchForTriangulation(KeyFrame* pKF1, KeyFrame* pKF2, cv::Mat F12, std::vector<std::pair<std::size_t, std::size_t> >& vMatchedPairs, const bool bOnlyStereo);
    int SearchForInitialization(Frame& F1, Frame& F2, std::vector<cv::Point2f>& vbPrevMatched, std::vector<int>& vnMatches12, int windowSize = 10);
    int untouched();   // a member the adapter does not replace
};
}  // namespace ORB_SLAM2
