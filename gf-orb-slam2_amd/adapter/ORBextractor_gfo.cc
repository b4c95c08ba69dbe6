// ORBextractor_gfo.cc -- drop-in replacement for the reference's src/ORBextractor.cc.
//
// Compile this file INSTEAD of src/ORBextractor.cc inside the reference tree (same
// include/ORBextractor.h, unchanged) and link libgfo.so: Frame.cc, Tracking.cc and every other
// caller link unchanged.  It implements exactly the public surface of include/ORBextractor.h:52-168:
//   ORBextractor::ORBextractor(int, float, int, int, int)     (:81-82)
//   void ORBextractor::operator()(InputArray, InputArray, vector<KeyPoint>&, OutputArray)   (:89-91)
//   void ORBextractor::ComputePyramid(cv::Mat)                (:132)
// and keeps the public member mvImagePyramid (:127) filled with host copies of the levels
// (views into 19-px-framed buffers, as the reference builds them, ORBextractor.cc:1182-1197), because
// Frame.h:237 and Frame.cc:994,1016,1171 read it.
//
// The header cannot carry a new member, so the gfo context of an extractor lives in a side table
// keyed by the object's address (the header's inline destructor is empty; a context is released when
// the same address is constructed again, and at process exit).
//
// Error behaviour follows the reference: no exceptions, no return codes.  Empty image -> return with
// the outputs untouched (ORBextractor.cc:1115-1116); any gfo error -> message on stderr and the
// "zero keypoints" result (_descriptors.release(), :1133-1134).
#include "ORBextractor.h"

#include <cassert>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <mutex>

#include "gfo.h"

namespace ORB_SLAM2
{

namespace
{
std::mutex g_mu;
std::map<const ORBextractor*, gfo_ctx*> g_ctx;

gfo_ctx* ctx_of(const ORBextractor* self)
{
    std::lock_guard<std::mutex> lk(g_mu);
    auto it = g_ctx.find(self);
    return it == g_ctx.end() ? nullptr : it->second;
}

struct AtExit {
    ~AtExit()
    {
        for (auto& kv : g_ctx) gfo_ctx_destroy(kv.second);
    }
} g_at_exit;
}  // namespace

// used by the matcher adapters (matchers_gfo.inc) to reach the device context of a frame's extractor
gfo_ctx* gfo_context_of(const ORBextractor* e) { return ctx_of(e); }

ORBextractor::ORBextractor(int _nfeatures, float _scaleFactor, int _nlevels, int _iniThFAST, int _minThFAST)
    : nfeatures(_nfeatures), scaleFactor(_scaleFactor), nlevels(_nlevels), iniThFAST(_iniThFAST), minThFAST(_minThFAST)
{
    gfo_params p;
    p.nfeatures = _nfeatures;
    p.scale_factor = _scaleFactor;
    p.nlevels = _nlevels;
    p.ini_th_fast = _iniThFAST;
    p.min_th_fast = _minThFAST;
    p.max_batch = 1;
    int dev = 0;
    if (const char* e = getenv("GFO_DEVICE")) dev = atoi(e);
    gfo_ctx* c = nullptr;
    if (gfo_ctx_create(&p, dev, &c) != GFO_OK) {
        fprintf(stderr, "[gfo] ORBextractor: %s\n", gfo_last_error(nullptr));
    }
    {
        std::lock_guard<std::mutex> lk(g_mu);
        auto it = g_ctx.find(this);
        if (it != g_ctx.end()) gfo_ctx_destroy(it->second);
        g_ctx[this] = c;
    }
    mvScaleFactor.resize(nlevels);
    mvInvScaleFactor.resize(nlevels);
    mvLevelSigma2.resize(nlevels);
    mvInvLevelSigma2.resize(nlevels);
    mnFeaturesPerLevel.resize(nlevels);
    mvImagePyramid.resize(nlevels);
    if (c)
        gfo_ctx_tables(c, mvScaleFactor.data(), mvInvScaleFactor.data(), mvLevelSigma2.data(), mvInvLevelSigma2.data(),
                       mnFeaturesPerLevel.data());
}

static void fetch_pyramid(gfo_ctx* c, std::vector<cv::Mat>& pyr, int nlevels, int w0, int h0)
{
    const int EDGE = 19;
    for (int l = 0; l < nlevels; ++l) {
        cv::Mat whole(h0 + 2 * EDGE, w0 + 2 * EDGE, CV_8UC1);  // generous; trimmed below
        int w = 0, h = 0;
        if (gfo_pyramid_level(c, 0, l, EDGE, whole.data, (int)whole.step, &w, &h) != GFO_OK) return;
        pyr[l] = whole(cv::Rect(EDGE, EDGE, w, h));  // ROI view, like ORBextractor.cc:1184
    }
}

void ORBextractor::ComputePyramid(cv::Mat image)
{
    gfo_ctx* c = ctx_of(this);
    if (!c || image.empty()) return;
    if (gfo_compute_pyramid(c, image.data, image.cols, image.rows, (int)image.step) != GFO_OK) {
        fprintf(stderr, "[gfo] ComputePyramid: %s\n", gfo_last_error(c));
        return;
    }
    fetch_pyramid(c, mvImagePyramid, nlevels, image.cols, image.rows);
}

void ORBextractor::operator()(cv::InputArray _image, cv::InputArray /*_mask*/, std::vector<cv::KeyPoint>& _keypoints,
                              cv::OutputArray _descriptors)
{
    if (_image.empty()) return;
    cv::Mat image = _image.getMat();
    assert(image.type() == CV_8UC1);
    gfo_ctx* c = ctx_of(this);
    static_assert(sizeof(cv::KeyPoint) == sizeof(gfo_keypoint), "gfo_keypoint must mirror cv::KeyPoint");
    int n = 0;
    int cap = c ? gfo_ctx_max_keypoints(c) : 0;
    _keypoints.clear();
    if (!c) {
        _descriptors.release();
        return;
    }
    std::vector<gfo_keypoint> kp(cap);
    cv::Mat desc(cap, 32, CV_8U);
    int rc = gfo_extract(c, image.data, image.cols, image.rows, (int)image.step, kp.data(), desc.data, cap, &n);
    if (rc == GFO_ERR_CAPACITY) {  // first call planned the geometry: retry with the exact bound
        cap = n;
        kp.resize(cap);
        desc.create(cap, 32, CV_8U);
        rc = gfo_extract(c, image.data, image.cols, image.rows, (int)image.step, kp.data(), desc.data, cap, &n);
    }
    if (rc != GFO_OK) {
        fprintf(stderr, "[gfo] ORBextractor::operator(): %s\n", gfo_last_error(c));
        n = 0;
    }
    if (n == 0) {
        _descriptors.release();
    } else {
        _descriptors.create(n, 32, CV_8U);
        cv::Mat out = _descriptors.getMat();
        memcpy(out.data, desc.data, (size_t)n * 32);
        _keypoints.resize(n);
        memcpy(static_cast<void*>(_keypoints.data()), kp.data(), (size_t)n * sizeof(gfo_keypoint));
    }
    // mvImagePyramid[0].rows is read by the stereo code (Frame.h:237, Frame.cc:1171); the SAD variant
    // (Frame.cc:994,1016) reads pixels of every level.  Host copies cost a D2H of the pyramid; set
    // GFO_LAZY_PYRAMID=1 to publish only correctly sized headers when ALTER_STEREO_MATCHING is on.
    static const bool lazy = getenv("GFO_LAZY_PYRAMID") && getenv("GFO_LAZY_PYRAMID")[0] == '1';
    if (!lazy) fetch_pyramid(c, mvImagePyramid, nlevels, image.cols, image.rows);
    else
        for (int l = 0; l < nlevels; ++l) {
            const float s = mvInvScaleFactor[l];
            mvImagePyramid[l] = cv::Mat(cvRound((float)image.rows * s), cvRound((float)image.cols * s), CV_8UC1, cv::Scalar(0));
        }
}

// Kept so that translation units which still name them link; the work happens on the device.
void ExtractorNode::DivideNode(ExtractorNode&, ExtractorNode&, ExtractorNode&, ExtractorNode&) {}
void ORBextractor::ComputeKeyPointsOctTree(std::vector<std::vector<cv::KeyPoint> >&) {}
std::vector<cv::KeyPoint> ORBextractor::DistributeOctTree(const std::vector<cv::KeyPoint>&, const int&, const int&, const int&,
                                                          const int&, const int&, const int&)
{
    return std::vector<cv::KeyPoint>();
}
void ORBextractor::ComputeKeyPointsOld(std::vector<std::vector<cv::KeyPoint> >&) {}

}  // namespace ORB_SLAM2
