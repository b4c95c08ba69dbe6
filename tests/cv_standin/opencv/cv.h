// Minimal OpenCV TYPE stand-in for building gf-orb-slam2_amd/adapter/*.cc (OUR code) against the reference's unchanged headers
// (include/ORBextractor.h, Frame.h, ORBmatcher.h, ...) in an image that has no OpenCV:
//   * tests/test_adapter_compiles.py parses the adapters with it (-fsyntax-only);
//   * tests/host/adapter_run.cc links the adapters with it and libgfo.so and RUNS them on the GPU box
//     (tests/test_gpu_adapter_run.py): the container semantics the adapters rely on are therefore real here --
//     reference-counted storage shared by copies and views, create() that keeps a matching allocation, ROI / row / col
//     views, clone(), copyTo() into a view, at<T>(), and small CV_32F matrix arithmetic (the host-side projections the
//     matcher adapters do with the reference's own expressions).
// It is NOT OpenCV and pins nothing about OpenCV's arithmetic: no resize, blur, FAST or fastAtan2 lives here, no reference
// SOURCE file is compiled with it, the oracle does not use it, and libgfo.so never sees it (plain pointers cross the C ABI).
#pragma once
#include <algorithm>
#include <climits>
#include <cmath>
#include <cstddef>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <iostream>
#include <list>
#include <map>
#include <memory>
#include <set>
#include <sstream>
#include <string>
#include <vector>

#define CV_8U 0
#define CV_8UC1 0
#define CV_32F 5
#define CV_64F 6

namespace cv
{
// (templates as in OpenCV, so that a signature with cv::Point2f mangles to the name a real build has: N2cv6Point_IfEE)
template <class T> struct Point_ { T x, y; Point_() : x(0), y(0) {} Point_(T a, T b) : x(a), y(b) {} };
typedef Point_<int> Point2i;
typedef Point2i Point;
typedef Point_<float> Point2f;
struct Rect { int x, y, width, height; Rect(int a, int b, int c, int d) : x(a), y(b), width(c), height(d) {} };
struct Scalar { double v; Scalar(double a = 0) : v(a) {} };
struct KeyPoint {
    Point2f pt; float size, angle, response; int octave, class_id;
    KeyPoint() : size(0), angle(-1), response(0), octave(0), class_id(-1) {}
    KeyPoint(float x, float y, float s, float a = -1, float r = 0, int o = 0, int c = -1) : pt(x, y), size(s), angle(a), response(r), octave(o), class_id(c) {}
};

class _OutputArray;

class Mat
{
public:
    unsigned char* data;
    int rows, cols;
    size_t step;
    Mat() : data(NULL), rows(0), cols(0), step(0), type_(CV_8U) {}
    Mat(int r, int c, int type) : data(NULL), rows(0), cols(0), step(0), type_(CV_8U) { create(r, c, type); }
    Mat(int r, int c, int type, const Scalar& s) : data(NULL), rows(0), cols(0), step(0), type_(CV_8U) { create(r, c, type); fill(s.v); }
    // user-owned memory (what cv::Mat(rows, cols, type, ptr, step) is): a header, nothing allocated
    Mat(int r, int c, int type, void* p, size_t st = 0) : data((unsigned char*)p), rows(r), cols(c), step(st ? st : (size_t)c * esz(type)), type_(type) {}
    // copies and views share the allocation (shared_ptr = the reference count)
    void create(int r, int c, int type)
    {
        if (data && r == rows && c == cols && type == type_) return;      // cv::Mat::create keeps a matching allocation
        type_ = type; rows = r; cols = c; step = (size_t)c * esz(type);
        store = std::make_shared<std::vector<unsigned char> >((size_t)r * step + 8, 0);
        data = store->data();
    }
    bool empty() const { return data == NULL || rows == 0 || cols == 0; }
    int type() const { return type_; }
    size_t elemSize() const { return esz(type_); }
    Mat operator()(const Rect& r) const { return view(r.y, r.x, r.height, r.width); }
    template <class T> T& at(int i) { return rows == 1 ? *reinterpret_cast<T*>(data + (size_t)i * sizeof(T)) : *reinterpret_cast<T*>(data + (size_t)i * step); }
    template <class T> const T& at(int i) const { return const_cast<Mat*>(this)->at<T>(i); }
    template <class T> T& at(int i, int j) { return *reinterpret_cast<T*>(data + (size_t)i * step + (size_t)j * sizeof(T)); }
    template <class T> const T& at(int i, int j) const { return const_cast<Mat*>(this)->at<T>(i, j); }
    template <class T> T* ptr(int i = 0) { return reinterpret_cast<T*>(data + (size_t)i * step); }
    template <class T> const T* ptr(int i = 0) const { return reinterpret_cast<const T*>(data + (size_t)i * step); }
    Mat clone() const
    {
        Mat m;
        if (empty()) return m;
        m.create(rows, cols, type_);
        for (int i = 0; i < rows; i++) memcpy(m.data + (size_t)i * m.step, data + (size_t)i * step, (size_t)cols * esz(type_));
        return m;
    }
    Mat row(int i) const { return view(i, 0, 1, cols); }
    Mat col(int j) const { return view(0, j, rows, 1); }
    Mat rowRange(int a, int b) const { return view(a, 0, b - a, cols); }
    Mat colRange(int a, int b) const { return view(0, a, rows, b - a); }
    Mat t() const
    {
        need32("t");
        Mat m(cols, rows, CV_32F);
        for (int i = 0; i < rows; i++) for (int j = 0; j < cols; j++) m.at<float>(j, i) = at<float>(i, j);
        return m;
    }
    // (CV_32F only) the sum of products in double, as cv::Mat::dot returns it
    double dot(const Mat& b) const
    {
        need32("dot"); b.need32("dot");
        if (rows * cols != b.rows * b.cols) { fprintf(stderr, "[cv stand-in] dot: size mismatch\n"); abort(); }
        double s = 0;
        const bool av = rows >= cols, bv = b.rows >= b.cols;     // vectors either way round
        for (int i = 0; i < rows * cols; i++) s += (double)(av ? at<float>(i, 0) : at<float>(0, i)) * (double)(bv ? b.at<float>(i, 0) : b.at<float>(0, i));
        return s;
    }
    Mat inv() const { fprintf(stderr, "[cv stand-in] Mat::inv is not provided\n"); abort(); }
    inline void copyTo(const _OutputArray& dst) const;
    bool isContinuous() const { return rows <= 1 || step == (size_t)cols * esz(type_); }
    void release() { store.reset(); data = NULL; rows = cols = 0; step = 0; }
    static Mat zeros(int r, int c, int t) { return Mat(r, c, t); }
    static Mat eye(int r, int c, int t)
    {
        Mat m(r, c, t);
        for (int i = 0; i < std::min(r, c); i++) { if (t == CV_32F) m.at<float>(i, i) = 1.f; else if (t == CV_64F) m.at<double>(i, i) = 1.0; else m.at<unsigned char>(i, i) = 1; }
        return m;
    }
    void need32(const char* what) const { if (type_ != CV_32F) { fprintf(stderr, "[cv stand-in] %s: CV_32F only\n", what); abort(); } }
    static size_t esz(int type) { return type == CV_32F ? 4 : type == CV_64F ? 8 : 1; }
private:
    Mat view(int y, int x, int h, int w) const
    {
        Mat m;
        m.type_ = type_; m.rows = h; m.cols = w; m.step = step; m.store = store;
        m.data = data + (size_t)y * step + (size_t)x * esz(type_);
        return m;
    }
    void fill(double v)
    {
        for (int i = 0; i < rows; i++) for (int j = 0; j < cols; j++) {
            if (type_ == CV_32F) at<float>(i, j) = (float)v; else if (type_ == CV_64F) at<double>(i, j) = v; else at<unsigned char>(i, j) = (unsigned char)v;
        }
    }
    int type_;
    std::shared_ptr<std::vector<unsigned char> > store;
};

// CV_32F only, plain loops in float (the adapters' host-side projections: 3x3 by 3x1 products and sums)
inline Mat operator*(const Mat& a, const Mat& b)
{
    a.need32("operator*"); b.need32("operator*");
    if (a.cols != b.rows) { fprintf(stderr, "[cv stand-in] operator*: %dx%d by %dx%d\n", a.rows, a.cols, b.rows, b.cols); abort(); }
    Mat m(a.rows, b.cols, CV_32F);
    for (int i = 0; i < a.rows; i++) for (int j = 0; j < b.cols; j++) {
        float s = 0.f;
        for (int k = 0; k < a.cols; k++) s += a.at<float>(i, k) * b.at<float>(k, j);
        m.at<float>(i, j) = s;
    }
    return m;
}
inline Mat scaled(const Mat& a, float f)
{
    a.need32("scale");
    Mat m(a.rows, a.cols, CV_32F);
    for (int i = 0; i < a.rows; i++) for (int j = 0; j < a.cols; j++) m.at<float>(i, j) = a.at<float>(i, j) * f;
    return m;
}
inline Mat combine(const Mat& a, const Mat& b, float sb)
{
    a.need32("operator+-"); b.need32("operator+-");
    if (a.rows != b.rows || a.cols != b.cols) { fprintf(stderr, "[cv stand-in] operator+-: size mismatch\n"); abort(); }
    Mat m(a.rows, a.cols, CV_32F);
    for (int i = 0; i < a.rows; i++) for (int j = 0; j < a.cols; j++) m.at<float>(i, j) = a.at<float>(i, j) + sb * b.at<float>(i, j);
    return m;
}
inline Mat operator/(const Mat& a, double f)
{
    a.need32("operator/");
    Mat m(a.rows, a.cols, CV_32F);
    for (int i = 0; i < a.rows; i++) for (int j = 0; j < a.cols; j++) m.at<float>(i, j) = a.at<float>(i, j) / (float)f;
    return m;
}
inline Mat operator*(const Mat& a, double f) { return scaled(a, (float)f); }
inline Mat operator*(double f, const Mat& a) { return scaled(a, (float)f); }
inline Mat operator+(const Mat& a, const Mat& b) { return combine(a, b, 1.f); }
inline Mat operator-(const Mat& a, const Mat& b) { return combine(a, b, -1.f); }
inline Mat operator-(const Mat& a) { return scaled(a, -1.f); }
inline double norm(const Mat& a)
{
    a.need32("norm");
    double s = 0;
    for (int i = 0; i < a.rows; i++) for (int j = 0; j < a.cols; j++) s += (double)a.at<float>(i, j) * a.at<float>(i, j);
    return std::sqrt(s);
}

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

// an lvalue Mat can be re-allocated by the callee; a temporary (a row / ROI view) is "fixed size": written in place
class _OutputArray
{
public:
    _OutputArray(Mat& m) : m_(&m), fixed_(false) {}
    _OutputArray(const Mat& m) : m_(const_cast<Mat*>(&m)), fixed_(true) {}
    void create(int r, int c, int t) const
    {
        if (fixed_) {
            if (m_->rows != r || m_->cols != c || m_->type() != t) { fprintf(stderr, "[cv stand-in] create on a fixed-size view: %dx%d vs %dx%d\n", r, c, m_->rows, m_->cols); abort(); }
            return;
        }
        m_->create(r, c, t);
    }
    void release() const { m_->release(); }
    Mat getMat() const { return *m_; }
    Mat& getMatRef() const { return *m_; }
private:
    Mat* m_;
    bool fixed_;
};
typedef const _OutputArray& OutputArray;

inline void Mat::copyTo(const _OutputArray& dst) const
{
    if (empty()) { dst.release(); return; }
    dst.create(rows, cols, type_);
    Mat d = dst.getMat();
    for (int i = 0; i < rows; i++) memcpy(d.data + (size_t)i * d.step, data + (size_t)i * step, (size_t)cols * esz(type_));
}

struct FileNode {
    FileNode operator[](const char*) const { return FileNode(); }
    FileNode operator[](int) const { return FileNode(); }
    operator int() const { return 0; }
    operator float() const { return 0; }
    operator double() const { return 0; }
    operator std::string() const { return ""; }
    size_t size() const { return 0; }
    int type() const { return 0; }
    bool empty() const { return true; }
    enum { SEQ = 5 };
};
struct FileStorage {
    enum { READ = 0, WRITE = 1 };
    FileStorage() {}
    FileStorage(const std::string&, int) {}
    bool isOpened() const { return false; }
    void release() {}
    FileNode operator[](const char*) const { return FileNode(); }
    FileNode operator[](const std::string&) const { return FileNode(); }
    template <class T> FileStorage& operator<<(const T&) { return *this; }
};
}  // namespace cv

// round half to even, like cvRound on every x86 build of OpenCV (cvtss2si under the default rounding mode)
static inline int cvRound(float v) { return (int)lrintf(v); }
static inline int cvRound(double v) { return (int)lrint(v); }
static inline int cvRound(int v) { return v; }
