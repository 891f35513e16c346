// stand-in for <pcl/filters/voxel_grid.h>: every call throws; never called on the paths under test (test infrastructure only)
#pragma once
#include "../../lvba_unavailable.h"
#include <pcl/point_cloud.h>
namespace pcl {
template <class P> class VoxelGrid {
  public:
    void setInputCloud(const typename PointCloud<P>::Ptr &) { lvba_unavailable("pcl::VoxelGrid"); }
    void setLeafSize(float, float, float) { lvba_unavailable("pcl::VoxelGrid"); }
    void filter(PointCloud<P> &) { lvba_unavailable("pcl::VoxelGrid"); }
};
}
