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
// in a side table keyed by the object's address (adapter/gfo_context_table.h, which states the rules): a live extractor
// keeps its context however many extractors there are -- a rig of K stereo cameras holds 2K contexts and never
// re-creates one (gfo_contexts_created() stays put; tools/c/boundary_throughput.c asserts it through the C ABI) --
// a constructor at a known address retires the old object's context at once (Tracking::updateORBExtractor,
// src/Tracking.cc:298-320), and a context whose owner has not called for thousands of lookups is reclaimed.
//
// Error behaviour follows the reference: no exceptions, no return codes.  Empty image -> return with
// the outputs untouched (ORBextractor.cc:1115-1116); any gfo error -> message on stderr and the
// "zero keypoints" result (_descriptors.release(), :1133-1134).
#include "ORBextractor.h"

#include <cassert>
#include <cstdio>
#include <cstdlib>
#include <cstring>

#include "gfo.h"
#include "gfo_context_table.h"

namespace ORB_SLAM2
{

namespace
{
bool full_pyramid()
{
    static const bool full = getenv("GFO_FULL_PYRAMID") && getenv("GFO_FULL_PYRAMID")[0] == '1';
    return full;
}

// Frames of several extractors -- the left / right pair of a Frame constructor (Frame.cc:84-87), the cameras of a rig --
// arrive on several threads at once: with the frame combiner on they share one device batch instead of competing for the
// runtime's four hardware queues (include/gfo.h, gfo_ctx_set_combining; GFO_COMBINE=0 opts out).  Not with
// GFO_FULL_PYRAMID=1: the levels fetched after operator() must be in the extractor's own context.
void configure_context(gfo_ctx* c)
{
    static const bool combine = !(getenv("GFO_COMBINE") && getenv("GFO_COMBINE")[0] == '0');
    gfo_ctx_set_combining(c, combine && !full_pyramid() ? 1 : 0);
}

struct Table : gfo_adapter::ContextTable {
    Table() { on_create = configure_context; }
} g_tab;

typedef gfo_adapter::ContextTable::Use Use;   // a context pinned for the duration of one call (reclaim() on another thread skips it)

struct AtExit {
    ~AtExit() { g_tab.destroy_all(); }
} g_at_exit;
}  // namespace

// used by the matcher adapters (matchers_gfo.cc) to reach the device context of a frame's extractor: pinned while the matcher
// call is inside the library (GfoUse there), released afterwards
gfo_ctx* gfo_context_pin(const ORBextractor* e) { return g_tab.acquire(e); }
void gfo_context_unpin(const ORBextractor* e, gfo_ctx* c) { if (c) g_tab.release(e, c); }
// A context for a matcher call that has no Frame to take one from -- ORBmatcher::SearchByBoW between two KEYFRAMES runs in the loop-closing
// thread, and a KeyFrame carries no extractor (include/KeyFrame.h): one context per calling thread, declared on its first use (it never
// extracts: no arena is ever planned for it), pinned for the call like an extractor's, reclaimed when idle like any other.
namespace
{
const void* thread_context_key()
{
    static thread_local char key;
    static thread_local bool declared = false;
    if (!declared) {
        gfo_params prm;
        prm.nfeatures = 1000; prm.scale_factor = 1.2f; prm.nlevels = 8; prm.ini_th_fast = 20; prm.min_th_fast = 7; prm.max_batch = 1;
        g_tab.declare(&key, prm);
        declared = true;
    }
    return &key;
}
}  // namespace
gfo_ctx* gfo_context_pin_thread() { return g_tab.acquire(thread_context_key()); }
void gfo_context_unpin_thread(gfo_ctx* c) { if (c) g_tab.release(thread_context_key(), c); }
// several GPUs (GFO_DEVICES): the right extractor of a stereo rig follows its left one (gfo_context_table.h); the HIP ordinal an
// extractor is placed on, and how many extractors have been moved so far
bool gfo_context_colocate(const ORBextractor* follower, const ORBextractor* leader) { return g_tab.colocate(follower, leader); }
int gfo_context_device(const ORBextractor* e) { return g_tab.device_of(e); }
int gfo_context_slot(const ORBextractor* e) { return g_tab.slot_of(e); }       // index into GFO_DEVICES
unsigned long gfo_contexts_moved() { return g_tab.moved(); }

ORBextractor::ORBextractor(int _nfeatures, float _scaleFactor, int _nlevels, int _iniThFAST, int _minThFAST)
    : nfeatures(_nfeatures), scaleFactor(_scaleFactor), nlevels(_nlevels), iniThFAST(_iniThFAST), minThFAST(_minThFAST)
{
    gfo_params prm;
    prm.nfeatures = _nfeatures;
    prm.scale_factor = _scaleFactor;
    prm.nlevels = _nlevels;
    prm.ini_th_fast = _iniThFAST;
    prm.min_th_fast = _minThFAST;
    prm.max_batch = 1;
    g_tab.declare(this, prm);   // an address the table knows: the previous owner is gone, its context is retired
    mvScaleFactor.resize(nlevels);
    mvInvScaleFactor.resize(nlevels);
    mvLevelSigma2.resize(nlevels);
    mvInvLevelSigma2.resize(nlevels);
    mnFeaturesPerLevel.resize(nlevels);
    mvImagePyramid.resize(nlevels);
    Use use(g_tab, this);
    if (gfo_ctx* c = use.ctx())
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
    Use use(g_tab, this);
    gfo_ctx* c = use.ctx();
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
    Use use(g_tab, this);
    gfo_ctx* c = use.ctx();
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
        // the reference hands back exactly n rows (:1137).  `desc` keeps the cap-row block alive (a header sharing its storage) while
        // create() gives the output its own n-row block: one copy, no temporary
        _descriptors.create(n, 32, CV_8U);
        cv::Mat exact = _descriptors.getMat();
        if (exact.data != desc.data) memcpy(exact.data, desc.data, (size_t)n * 32);
    }
    if (full_pyramid()) {
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
