// stand-in for <pcl/point_cloud.h>: a vector of points with the members the reference's BALM headers touch
// (test infrastructure only)
#pragma once
#include <cstddef>
#include <cstdint>
#include <memory>
#include <utility>
#include <vector>
namespace pcl {
template <class PointT>
class PointCloud {
  public:
    typedef std::shared_ptr<PointCloud<PointT>> Ptr;
    typedef std::shared_ptr<const PointCloud<PointT>> ConstPtr;
    std::vector<PointT> points;
    uint32_t width = 0, height = 0;
    bool is_dense = true;
    size_t size() const { return points.size(); }
    bool empty() const { return points.empty(); }
    void reserve(size_t n) { points.reserve(n); }
    void resize(size_t n) { points.resize(n); width = (uint32_t)n; height = 1; }
    void clear() { points.clear(); width = height = 0; }
    template <class... A> void emplace_back(A &&...a) { points.emplace_back(std::forward<A>(a)...); width = (uint32_t)points.size(); height = 1; }
    PointCloud &operator+=(const PointCloud &o)
    {
        points.insert(points.end(), o.points.begin(), o.points.end());
        width = (uint32_t)points.size(); height = 1;
        return *this;
    }
    void push_back(const PointT &p) { points.push_back(p); width = (uint32_t)points.size(); height = 1; }
    void swap(PointCloud &o) { points.swap(o.points); std::swap(width, o.width); std::swap(height, o.height); std::swap(is_dense, o.is_dense); }
    PointT &operator[](size_t i) { return points[i]; }
    const PointT &operator[](size_t i) const { return points[i]; }
    typename std::vector<PointT>::iterator begin() { return points.begin(); }
    typename std::vector<PointT>::iterator end() { return points.end(); }
    typename std::vector<PointT>::const_iterator begin() const { return points.begin(); }
    typename std::vector<PointT>::const_iterator end() const { return points.end(); }
};
} // namespace pcl
