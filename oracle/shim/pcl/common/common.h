// stand-in for <pcl/common/common.h> (test infrastructure only)
#pragma once
#include <pcl/point_cloud.h>
#include <pcl/point_types.h>
