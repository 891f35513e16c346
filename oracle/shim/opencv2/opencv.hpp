// stand-in for <opencv2/opencv.hpp> (OpenCV is not installed).  cv::Mat is a real, reference-counted 2-D array (the depth
// images of generateDepthWithVoxel / fetchDepthBilinear live in it); the image codecs, drawing and filtering calls either throw
// (never reached on the paths oracle/ref_glue*.cpp run) or explicit no-ops where the reference calls them
// unconditionally on a tested path (imwrite of the depth preview, initUndistortRectifyMap in the DatasetIO constructor).
// TEST INFRASTRUCTURE ONLY.
#pragma once
#include <array>
#include "../lvba_unavailable.h"
#include <cmath>
#include <cstdint>
#include <cstring>
#include <iomanip>
#include <iostream>
#include <limits>
#include <memory>
#include <regex>
#include <sstream>
#include <string>
#include <vector>
#define CV_8UC1 0
#define CV_8UC3 16
#define CV_16UC1 2
#define CV_16SC2 11
#define CV_32FC1 5
#define CV_64FC1 6
#define CV_RGB(r, g, b) cv::Scalar((b), (g), (r), 0)
namespace cv {
inline int lvba_elem_size(int type)
{
    const int depth = type & 7, cn = (type >> 3) + 1;
    static const int bytes[8] = {1, 1, 2, 2, 4, 4, 8, 2};
    return bytes[depth] * cn;
}
struct Size { int width, height; Size(int w = 0, int h = 0) : width(w), height(h) {} Size(double w, double h) : width((int)w), height((int)h) {} };
struct Scalar { double val[4]; Scalar(double a = 0, double b = 0, double c = 0, double d = 0) : val{a, b, c, d} {} };
template <class T> struct Point_ {
    T x, y;
    Point_() : x(0), y(0) {}
    Point_(T a, T b) : x(a), y(b) {}
    template <class U> Point_(const Point_<U> &o) : x((T)o.x), y((T)o.y) {} // NOLINT: implicit, like cv::Point_
    Point_ operator-(const Point_ &o) const { return Point_(x - o.x, y - o.y); }
    Point_ operator+(const Point_ &o) const { return Point_(x + o.x, y + o.y); }
};
typedef Point_<int> Point;
typedef Point_<float> Point2f;
typedef Point_<double> Point2d;
template <class T> double norm(const Point_<T> &p) { return std::sqrt((double)p.x * p.x + (double)p.y * p.y); }
struct Rect { int x, y, width, height; Rect(int a = 0, int b = 0, int w = 0, int h = 0) : x(a), y(b), width(w), height(h) {} };
struct Vec3b { unsigned char v[3]; unsigned char &operator[](int i) { return v[i]; } const unsigned char &operator[](int i) const { return v[i]; } };
struct RNG { explicit RNG(uint64_t = 0) {} int uniform(int, int) { lvba_unavailable("cv::RNG"); } };
class Mat {
  public:
    int rows = 0, cols = 0;
    unsigned char *data = nullptr;
    Mat() {}
    Mat(int r, int c, int type) { create(r, c, type); }
    Mat(int r, int c, int type, const Scalar &s) { create(r, c, type); fill(s); }
    void create(int r, int c, int type)
    {
        rows = r; cols = c; type_ = type;
        buf_ = std::make_shared<std::vector<unsigned char>>((size_t)(r > 0 ? r : 0) * (size_t)(c > 0 ? c : 0) * lvba_elem_size(type), 0);
        data = buf_->data();
    }
    bool empty() const { return rows == 0 || cols == 0 || data == nullptr; }
    int type() const { return type_; }
    int channels() const { return (type_ >> 3) + 1; }
    Mat clone() const
    {
        Mat m;
        m.rows = rows; m.cols = cols; m.type_ = type_;
        if (buf_) { m.buf_ = std::make_shared<std::vector<unsigned char>>(*buf_); m.data = m.buf_->data(); }
        return m;
    }
    template <class T> T &at(int y, int x) { return reinterpret_cast<T *>(data)[(size_t)y * cols + x]; }
    template <class T> const T &at(int y, int x) const { return reinterpret_cast<const T *>(data)[(size_t)y * cols + x]; }
    template <class T> T *ptr(int y) { return reinterpret_cast<T *>(data) + (size_t)y * cols; }
    template <class T> const T *ptr(int y) const { return reinterpret_cast<const T *>(data) + (size_t)y * cols; }
    void convertTo(Mat &, int, double = 1.0, double = 0.0) const {} // previews only
    void copyTo(Mat) const { lvba_unavailable("cv::Mat::copyTo"); }
    Mat operator()(const Rect &) const { lvba_unavailable("cv::Mat::operator()(Rect)"); }

  protected:
    void fill(const Scalar &s)
    {
        const size_t n = (size_t)rows * cols;
        if (type_ == CV_32FC1) for (size_t i = 0; i < n; ++i) reinterpret_cast<float *>(data)[i] = (float)s.val[0];
        else if (type_ == CV_64FC1) for (size_t i = 0; i < n; ++i) reinterpret_cast<double *>(data)[i] = s.val[0];
        else if (type_ == CV_8UC3) for (size_t i = 0; i < n; ++i) for (int c = 0; c < 3; ++c) data[3 * i + c] = (unsigned char)s.val[c];
        else std::memset(data, (int)s.val[0], n * lvba_elem_size(type_));
    }
    int type_ = 0;
    std::shared_ptr<std::vector<unsigned char>> buf_;
};
template <class T> struct LvbaDepthOf;
template <> struct LvbaDepthOf<float> { enum { value = CV_32FC1 }; };
template <> struct LvbaDepthOf<double> { enum { value = CV_64FC1 }; };
template <class T> class Mat_;
template <class T> struct MatCommaInit_ {
    Mat_<T> *m;
    size_t k;
    template <class V> MatCommaInit_ &operator,(V v);
    operator Mat() const;
};
template <class T> class Mat_ : public Mat {
  public:
    Mat_() {}
    Mat_(int r, int c) : Mat(r, c, LvbaDepthOf<T>::value) {}
    template <class V> MatCommaInit_<T> operator<<(V v) { reinterpret_cast<T *>(data)[0] = (T)v; return MatCommaInit_<T>{this, 1}; }
    static Mat_ eye(int r, int c) { Mat_ m(r, c); for (int i = 0; i < (r < c ? r : c); ++i) m.template at<T>(i, i) = (T)1; return m; }
};
template <class T> template <class V> MatCommaInit_<T> &MatCommaInit_<T>::operator,(V v) { reinterpret_cast<T *>(m->data)[k++] = (T)v; return *this; }
template <class T> MatCommaInit_<T>::operator Mat() const { return *m; }
template <class T> using Ptr = std::shared_ptr<T>;
struct CLAHE { virtual ~CLAHE() {} virtual void apply(const Mat &, Mat &) = 0; };
enum { INTER_LINEAR = 1, INTER_CUBIC = 2, COLOR_BGR2Lab = 44, COLOR_Lab2BGR = 56, IMREAD_COLOR = 1, IMREAD_UNCHANGED = -1, LINE_AA = 16,
       FONT_HERSHEY_SIMPLEX = 0 };
// no-ops on tested paths
inline void initUndistortRectifyMap(const Mat &, const Mat &, const Mat &, const Mat &, Size, int, Mat &, Mat &) {}
inline bool imwrite(const std::string &, const Mat &) { return true; }
// unreachable on the tested paths: every call throws
// no image codec: when the test glue has set a synthetic image size, imread hands out a deterministic BGR pattern of that size
// (b, g, r) = (x mod 256, y mod 256, (x + y) mod 256) whatever the path -- enough for the reference's colourising export to
// run; otherwise it throws
inline int (&lvba_synthetic_image_size())[2] { static int wh[2] = {0, 0}; return wh; }
inline Mat imread(const std::string &, int = IMREAD_COLOR)
{
    const int w = lvba_synthetic_image_size()[0], h = lvba_synthetic_image_size()[1];
    if (w <= 0 || h <= 0) lvba_unavailable("cv::imread");
    Mat m(h, w, CV_8UC3);
    for (int y = 0; y < h; ++y)
        for (int x = 0; x < w; ++x) {
            unsigned char *p = m.data + 3 * ((size_t)y * w + x);
            p[0] = (unsigned char)(x % 256); p[1] = (unsigned char)(y % 256); p[2] = (unsigned char)((x + y) % 256);
        }
    return m;
}
inline void remap(const Mat &src, Mat &dst, const Mat &, const Mat &, int) { dst = src; } // undistorted previews are never read back
inline void resize(const Mat &, Mat &, Size, double = 0, double = 0, int = 1) { lvba_unavailable("cv::resize"); }
inline void cvtColor(const Mat &, Mat &, int) { lvba_unavailable("cv::cvtColor"); }
inline void split(const Mat &, std::vector<Mat> &) { lvba_unavailable("cv::split"); }
inline void merge(const std::vector<Mat> &, Mat &) { lvba_unavailable("cv::merge"); }
inline Ptr<CLAHE> createCLAHE(double = 40.0, Size = Size(8, 8)) { lvba_unavailable("cv::createCLAHE"); }
inline void GaussianBlur(const Mat &, Mat &, Size, double, double = 0) { lvba_unavailable("cv::GaussianBlur"); }
inline void addWeighted(const Mat &, double, const Mat &, double, double, Mat &) { lvba_unavailable("cv::addWeighted"); }
inline void line(Mat &, Point2d, Point2d, const Scalar &, int = 1, int = 8, int = 0) { lvba_unavailable("cv::line"); }
inline void circle(Mat &, Point2d, int, const Scalar &, int = 1, int = 8, int = 0) { lvba_unavailable("cv::circle"); }
inline void rectangle(Mat &, Point, Point, const Scalar &, int = 1, int = 8, int = 0) { lvba_unavailable("cv::rectangle"); }
inline void putText(Mat &, const std::string &, Point, int, double, Scalar, int = 1, int = 8, bool = false) { lvba_unavailable("cv::putText"); }
} // namespace cv
