/* Linux counterpart of the reference's precompiled header (ICP/stdafx.h:3-16): same standard headers,
 * PCL names from the header-only stand-in, no OpenCV (only the out-of-scope regist.h used it). */
#pragma once
#include <iostream>
#include <vector>
#include <string>
#include <assert.h>
#include <pcl/io/pcd_io.h>
#include <pcl/point_cloud.h>
#include <pcl/point_types.h>
#include <pcl/visualization/pcl_visualizer.h>
#include <pcl/PCLPointCloud2.h>
#include <pcl/common/transforms.h>
#include <pcl/features/normal_3d.h>
using std::cout;
using std::endl;
