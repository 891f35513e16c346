// stand-in for <pcl/visualization/pcl_visualizer.h>: nothing of it is used on the tested path (test infrastructure only)
#pragma once
