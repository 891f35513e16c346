// stand-in for <sensor_msgs/PointCloud2.h> (test infrastructure only)
#pragma once
#include <string>
#include <ros/ros.h>
namespace sensor_msgs {
struct PointCloud2 { struct { std::string frame_id; ros::Time stamp; } header; };
}
