// Minimal OpenCV type stand-in used ONLY to syntax-check gf-orb-slam2_amd/adapter/ORBextractor_gfo.cc
// against the reference's unchanged include/ORBextractor.h (tests/test_adapter_compiles.py).
// It is NOT OpenCV, nothing is linked or run with it, no reference source file is compiled with it,
// and the oracle does not use it.
#pragma once
#include <cstddef>
#include <cstring>
#include <vector>

#define CV_8U 0
#define CV_8UC1 0

namespace cv
{
struct Point { int x, y; Point() : x(0), y(0) {} Point(int a, int b) : x(a), y(b) {} };
typedef Point Point2i;
struct Point2f { float x, y; };
struct Rect { int x, y, width, height; Rect(int a, int b, int c, int d) : x(a), y(b), width(c), height(d) {} };
struct Scalar { double v; Scalar(double a = 0) : v(a) {} };
struct KeyPoint { Point2f pt; float size, angle, response; int octave, class_id; };

class Mat
{
public:
    unsigned char* data = nullptr;
    int rows = 0, cols = 0;
    size_t step = 0;
    Mat() {}
    Mat(int r, int c, int /*type*/) { create(r, c, 0); }
    Mat(int r, int c, int /*type*/, const Scalar&) { create(r, c, 0); }
    void create(int r, int c, int /*type*/) { rows = r; cols = c; step = (size_t)c; store.assign((size_t)r * c, 0); data = store.data(); }
    bool empty() const { return data == nullptr || rows == 0 || cols == 0; }
    int type() const { return CV_8UC1; }
    Mat operator()(const Rect& r) const { Mat m; m.data = data + (size_t)r.y * step + r.x; m.rows = r.height; m.cols = r.width; m.step = step; m.store = store; return m; }
private:
    std::vector<unsigned char> store;
};

class _InputArray
{
public:
    _InputArray(const Mat& m) : m_(&m) {}
    bool empty() const { return m_->empty(); }
    Mat getMat() const { return *m_; }
private:
    const Mat* m_;
};
typedef const _InputArray& InputArray;

class _OutputArray
{
public:
    _OutputArray(Mat& m) : m_(&m) {}
    void create(int r, int c, int t) const { m_->create(r, c, t); }
    void release() const { *m_ = Mat(); }
    Mat getMat() const { return *m_; }
private:
    Mat* m_;
};
typedef const _OutputArray& OutputArray;
}  // namespace cv

static inline int cvRound(float v) { return (int)(v + (v >= 0 ? 0.5f : -0.5f)); }
