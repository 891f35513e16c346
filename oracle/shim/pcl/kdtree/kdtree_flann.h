// stand-in for <pcl/kdtree/kdtree_flann.h>: the real header pulls in pcl::PointCloud, which is all utils.hpp needs from it
// (test infrastructure only)
#pragma once
#include "../point_cloud.h"
#include "../point_types.h"
