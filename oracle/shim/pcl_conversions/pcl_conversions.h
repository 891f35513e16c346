// stand-in for <pcl_conversions/pcl_conversions.h>: messages are dropped (test infrastructure only)
#pragma once
#include <pcl/point_cloud.h>
#include <sensor_msgs/PointCloud2.h>
namespace pcl {
template <class C> void toROSMsg(const C &, sensor_msgs::PointCloud2 &) {}
}
