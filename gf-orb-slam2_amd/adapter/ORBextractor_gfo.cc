// ORBextractor_gfo.cc -- drop-in replacement for the reference's src/ORBextractor.cc.
//
// Compile this file INSTEAD of src/ORBextractor.cc inside the reference tree (same
// include/ORBextractor.h, unchanged) and link libgfo.so: Frame.cc, Tracking.cc and every other
// caller link unchanged.  It implements exactly the public surface of include/ORBextractor.h:52-168:
//   ORBextractor::ORBextractor(int, float, int, int, int)     (:81-82)
//   void ORBextractor::operator()(InputArray, InputArray, vector<KeyPoint>&, OutputArray)   (:89-91)
//   void ORBextractor::ComputePyramid(cv::Mat)                (:132)
// and keeps the public member mvImagePyramid (:127) usable:
//   - operator() publishes correctly SIZED levels (what the default build reads: only mvImagePyramid[0].rows,
//     Frame.h:237, Frame.cc:1171, under ALTER_STEREO_MATCHING, Frame.h:37) without copying a pixel;
//     GFO_FULL_PYRAMID=1 makes every call copy the levels back to the host (the SAD stereo variant reads their
//     pixels, Frame.cc:994,1016);
//   - ComputePyramid(), whose callers want the pixels (Frame.cc:182-183), always copies them, as views into
//     19-px-framed buffers like the reference builds them (ORBextractor.cc:1182-1197).
//
// The header cannot carry a new member and its inline destructor is empty, so the gfo context of an extractor lives
// in a side table keyed by the object's address, as a CACHE: the table remembers the constructor arguments of every
// extractor and (re)creates a context on demand, and only the GFO_MAX_CONTEXTS (default 4) most recently used
// contexts stay alive.  Tracking::updateORBExtractor (src/Tracking.cc:298-320) deletes and re-creates both
// extractors at run time: the contexts of the deleted pair fall out of the cache as soon as the new pair is in use,
// whether or not the allocator hands the new objects the old addresses.
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
struct Entry {
    gfo_params prm;
    gfo_ctx* ctx;
    unsigned long stamp;
};
std::mutex g_mu;
std::map<const ORBextractor*, Entry> g_tab;
unsigned long g_clock = 0;

int max_contexts()
{
    static const int n = getenv("GFO_MAX_CONTEXTS") ? atoi(getenv("GFO_MAX_CONTEXTS")) : 4;
    return n < 2 ? 2 : n;
}

// the caller holds g_mu
void evict_lru(const ORBextractor* keep)
{
    for (;;) {
        int alive = 0;
        std::map<const ORBextractor*, Entry>::iterator oldest = g_tab.end();
        for (std::map<const ORBextractor*, Entry>::iterator it = g_tab.begin(); it != g_tab.end(); ++it) {
            if (!it->second.ctx) continue;
            alive++;
            if (it->first != keep && (oldest == g_tab.end() || it->second.stamp < oldest->second.stamp)) oldest = it;
        }
        if (alive <= max_contexts() || oldest == g_tab.end()) return;
        gfo_ctx_destroy(oldest->second.ctx);   // its owner (if it still exists) gets a fresh one on its next call
        oldest->second.ctx = NULL;
    }
}

gfo_ctx* ctx_of(const ORBextractor* self)
{
    std::lock_guard<std::mutex> lk(g_mu);
    std::map<const ORBextractor*, Entry>::iterator it = g_tab.find(self);
    if (it == g_tab.end()) return NULL;   // not constructed through this file
    Entry& e = it->second;
    e.stamp = ++g_clock;
    if (!e.ctx) {
        int dev = 0;
        if (const char* d = getenv("GFO_DEVICE")) dev = atoi(d);
        if (gfo_ctx_create(&e.prm, dev, &e.ctx) != GFO_OK) {
            fprintf(stderr, "[gfo] ORBextractor: %s\n", gfo_last_error(NULL));
            e.ctx = NULL;
            return NULL;
        }
        evict_lru(self);
    }
    return e.ctx;
}

struct AtExit {
    ~AtExit()
    {
        for (std::map<const ORBextractor*, Entry>::iterator it = g_tab.begin(); it != g_tab.end(); ++it)
            if (it->second.ctx) gfo_ctx_destroy(it->second.ctx);
    }
} g_at_exit;
}  // namespace

// used by the matcher adapters (matchers_gfo.cc) to reach the device context of a frame's extractor
gfo_ctx* gfo_context_of(const ORBextractor* e) { return ctx_of(e); }

ORBextractor::ORBextractor(int _nfeatures, float _scaleFactor, int _nlevels, int _iniThFAST, int _minThFAST)
    : nfeatures(_nfeatures), scaleFactor(_scaleFactor), nlevels(_nlevels), iniThFAST(_iniThFAST), minThFAST(_minThFAST)
{
    Entry e;
    e.prm.nfeatures = _nfeatures;
    e.prm.scale_factor = _scaleFactor;
    e.prm.nlevels = _nlevels;
    e.prm.ini_th_fast = _iniThFAST;
    e.prm.min_th_fast = _minThFAST;
    e.prm.max_batch = 1;
    e.ctx = NULL;
    e.stamp = 0;
    {
        std::lock_guard<std::mutex> lk(g_mu);
        std::map<const ORBextractor*, Entry>::iterator it = g_tab.find(this);
        if (it != g_tab.end() && it->second.ctx) gfo_ctx_destroy(it->second.ctx);   // the address of a deleted extractor, reused
        g_tab[this] = e;
    }
    mvScaleFactor.resize(nlevels);
    mvInvScaleFactor.resize(nlevels);
    mvLevelSigma2.resize(nlevels);
    mvInvLevelSigma2.resize(nlevels);
    mnFeaturesPerLevel.resize(nlevels);
    mvImagePyramid.resize(nlevels);
    if (gfo_ctx* c = ctx_of(this))
        gfo_ctx_tables(c, mvScaleFactor.data(), mvInvScaleFactor.data(), mvLevelSigma2.data(), mvInvLevelSigma2.data(),
                       mnFeaturesPerLevel.data());
}

// host copies of the levels of the last pyramid, each a view into its own (w + 38) x (h + 38) framed buffer
static void fetch_pyramid(gfo_ctx* c, std::vector<cv::Mat>& pyr, const std::vector<float>& inv_scale, int nlevels, int w0, int h0)
{
    const int EDGE = 19;
    for (int l = 0; l < nlevels; ++l) {
        const int wl = cvRound((float)w0 * inv_scale[l]), hl = cvRound((float)h0 * inv_scale[l]);   // ORBextractor.cc:1180-1181
        cv::Mat whole(hl + 2 * EDGE, wl + 2 * EDGE, CV_8UC1);
        int w = 0, h = 0;
        if (gfo_pyramid_level(c, 0, l, EDGE, whole.data, (int)whole.step, &w, &h) != GFO_OK || w != wl || h != hl) {
            fprintf(stderr, "[gfo] mvImagePyramid[%d]: %s\n", l, gfo_last_error(c));
            return;
        }
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
    fetch_pyramid(c, mvImagePyramid, mvInvScaleFactor, nlevels, image.cols, image.rows);
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
    // the results land directly in the caller's containers: cv::KeyPoint and gfo_keypoint share their layout
    _keypoints.resize(cap);
    _descriptors.create(cap, 32, CV_8U);
    cv::Mat desc = _descriptors.getMat();
    int rc = gfo_extract(c, image.data, image.cols, image.rows, (int)image.step, reinterpret_cast<gfo_keypoint*>(_keypoints.data()),
                         desc.data, cap, &n);
    if (rc == GFO_ERR_CAPACITY) {  // the first call planned the geometry: retry with the exact bound
        cap = n;
        _keypoints.resize(cap);
        _descriptors.create(cap, 32, CV_8U);
        desc = _descriptors.getMat();
        rc = gfo_extract(c, image.data, image.cols, image.rows, (int)image.step, reinterpret_cast<gfo_keypoint*>(_keypoints.data()),
                         desc.data, cap, &n);
    }
    if (rc != GFO_OK) {
        fprintf(stderr, "[gfo] ORBextractor::operator(): %s\n", gfo_last_error(c));
        n = 0;
    }
    _keypoints.resize(n);
    if (n == 0) {
        _descriptors.release();
    } else if (n < cap) {
        cv::Mat exact(n, 32, CV_8U);                       // the reference hands back exactly n rows (:1137)
        memcpy(exact.data, desc.data, (size_t)n * 32);
        _descriptors.create(n, 32, CV_8U);
        memcpy(_descriptors.getMat().data, exact.data, (size_t)n * 32);
    }
    static const bool full = getenv("GFO_FULL_PYRAMID") && getenv("GFO_FULL_PYRAMID")[0] == '1';
    if (full) {
        fetch_pyramid(c, mvImagePyramid, mvInvScaleFactor, nlevels, image.cols, image.rows);
    } else {
        // sized headers only (allocated once per image size): the default build reads mvImagePyramid[0].rows and nothing else
        for (int l = 0; l < nlevels; ++l) {
            const int wl = cvRound((float)image.cols * mvInvScaleFactor[l]), hl = cvRound((float)image.rows * mvInvScaleFactor[l]);
            if (mvImagePyramid[l].rows != hl || mvImagePyramid[l].cols != wl) mvImagePyramid[l] = cv::Mat(hl, wl, CV_8UC1, cv::Scalar(0));
        }
    }
}

// Kept so that translation units which still name them link; the work happens on the device and these are never
// reached through operator().  Anything that calls them directly gets an empty result and a message.
void ExtractorNode::DivideNode(ExtractorNode&, ExtractorNode&, ExtractorNode&, ExtractorNode&)
{
    fprintf(stderr, "[gfo] ExtractorNode::DivideNode: the quadtree runs on the device (libgfo); this host stub does nothing\n");
}
void ORBextractor::ComputeKeyPointsOctTree(std::vector<std::vector<cv::KeyPoint> >& allKeypoints)
{
    fprintf(stderr, "[gfo] ComputeKeyPointsOctTree: use operator(); this host stub returns no keypoints\n");
    allKeypoints.assign(nlevels, std::vector<cv::KeyPoint>());
}
std::vector<cv::KeyPoint> ORBextractor::DistributeOctTree(const std::vector<cv::KeyPoint>&, const int&, const int&, const int&,
                                                          const int&, const int&, const int&)
{
    fprintf(stderr, "[gfo] DistributeOctTree: the quadtree runs on the device (libgfo); this host stub returns no keypoints\n");
    return std::vector<cv::KeyPoint>();
}
void ORBextractor::ComputeKeyPointsOld(std::vector<std::vector<cv::KeyPoint> >& allKeypoints)
{
    fprintf(stderr, "[gfo] ComputeKeyPointsOld: not part of the accelerated path; this host stub returns no keypoints\n");
    allKeypoints.assign(nlevels, std::vector<cv::KeyPoint>());
}

}  // namespace ORB_SLAM2
