// stand-in for <pcl/point_types.h>: only the point layouts the reference's BALM headers name (test infrastructure only)
#pragma once
namespace pcl {
struct PointXYZ { float x = 0, y = 0, z = 0, pad_ = 1; PointXYZ() {} PointXYZ(float a, float b, float c) : x(a), y(b), z(c) {} };
struct PointXYZI { float x = 0, y = 0, z = 0, pad_ = 1, intensity = 0, pad2_[3] = {0, 0, 0}; };
struct PointXYZINormal {
    float x = 0, y = 0, z = 0, pad_ = 1;
    float normal_x = 0, normal_y = 0, normal_z = 0, pad2_ = 0;
    float intensity = 0, curvature = 0, pad3_[2] = {0, 0};
};
struct PointXYZRGB { float x = 0, y = 0, z = 0, pad_ = 1; unsigned char b = 0, g = 0, r = 0, a = 255; float pad2_[3] = {0, 0, 0}; };
struct PointXYZRGBA { float x = 0, y = 0, z = 0, pad_ = 1; unsigned char b = 0, g = 0, r = 0, a = 255; float pad2_[3] = {0, 0, 0}; };
} // namespace pcl
