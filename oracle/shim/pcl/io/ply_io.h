// stand-in for <pcl/io/ply_io.h> (test infrastructure only)
#pragma once
