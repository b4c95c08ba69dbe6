// stands for Tracking.cc: calls through the unchanged public signatures
#include <cstdio>
#include "decl.h"
using namespace ORB_SLAM2;
int main()
{
    Frame f, g;
    ORBmatcher m;
    std::vector<MapPoint*> v;
    std::set<MapPoint*> s;
    double nv = 0;
    std::vector<cv::Point2f> pm;
    std::vector<int> vi;
    std::vector<std::pair<std::size_t, std::size_t> > pp;
    f.ComputeBoW();
    printf("%d %d %d %d %d %d %d %d %d %d %d %d %d %d %d %d\n", f.ComputeStereoMatches_Undistorted(false), f.construct(), f.bow, m.SearchByProjection(f, v, 3.f),
           m.SearchByProjection(f, g, 3.f, false, nv), m.SearchByProjection(f, (KeyFrame*)0, s, 3.f, 100), m.SearchByBoW((KeyFrame*)0, f, v), m.SearchByProjection_Budget(f, v, 0.5f, 0.01), m.SearchByBoW((KeyFrame*)0, (KeyFrame*)0, v), m.SearchByProjection((KeyFrame*)0, cv::Mat(), v, v, 10), m.Fuse((KeyFrame*)0, cv::Mat(), v, 4.f, v), m.Fuse((KeyFrame*)0, v, 3.f), m.SearchBySim3((KeyFrame*)0, (KeyFrame*)0, v, 1.f, cv::Mat(), cv::Mat(), 7.5f), m.SearchForTriangulation((KeyFrame*)0, (KeyFrame*)0, cv::Mat(), pp, false), m.SearchForInitialization(f, g, pm, vi, 100), m.untouched());
    return 0;
}
