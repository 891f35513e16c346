// stand-in for <opencv2/opencv.hpp>: declarations only, enough for include/utils.hpp of the reference to PARSE.  None of
// the image functions is ever called by oracle/ref_glue_visual.cpp (test infrastructure only).
#pragma once
#include <array>
#include <cmath>
#include <cstdint>
#include <iomanip>
#include <iostream>
#include <limits>
#include <memory>
#include <regex>
#include <sstream>
#include <string>
#include <vector>
#define CV_32FC1 5
#define CV_16UC1 2
namespace cv {
struct Size { int width, height; Size(int w = 0, int h = 0) : width(w), height(h) {} };
class Mat {
  public:
    int rows = 0, cols = 0;
    bool empty() const { return rows == 0 || cols == 0; }
    int type() const { return type_; }
    Mat clone() const { return *this; }
    template <class T> T &at(int y, int x) { return reinterpret_cast<T *>(data_.data())[(size_t)y * cols + x]; }
    template <class T> const T &at(int y, int x) const { return reinterpret_cast<const T *>(data_.data())[(size_t)y * cols + x]; }
    int type_ = 0;
    std::vector<unsigned char> data_;
};
template <class T> using Ptr = std::shared_ptr<T>;
struct CLAHE { virtual ~CLAHE() {} virtual void apply(const Mat &, Mat &) = 0; };
enum { INTER_CUBIC = 2, COLOR_BGR2Lab = 44, COLOR_Lab2BGR = 56 };
void resize(const Mat &, Mat &, Size, double = 0, double = 0, int = 1);
void cvtColor(const Mat &, Mat &, int);
void split(const Mat &, std::vector<Mat> &);
void merge(const std::vector<Mat> &, Mat &);
Ptr<CLAHE> createCLAHE(double = 40.0, Size = Size(8, 8));
void GaussianBlur(const Mat &, Mat &, Size, double, double = 0);
void addWeighted(const Mat &, double, const Mat &, double, double, Mat &);
} // namespace cv
