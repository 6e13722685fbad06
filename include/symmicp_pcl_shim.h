/*
 * symmicp_pcl_shim.h -- the few pcl:: names the reference's driver touches
 * (ICP/main.cpp:13-40, ICP/myicp.h:5,15-16), so that a main.cpp with the reference's call
 * sequence compiles and runs headless on Linux without PCL / VTK / Boost.
 * Types only; no PCL algorithm is re-implemented here (search and normals live in libsymmicp).
 *   pcl::PointXYZ       16-byte x,y,z,pad   (PCL's layout)
 *   pcl::PointNormal    48-byte x,y,z,pad, normal_x,y,z,pad, curvature,pad[3]
 *   pcl::PointCloud<T>  points / width / height / Ptr (std::shared_ptr here, boost::shared_ptr in PCL 1.9.1)
 *   pcl::visualization::PCLVisualizer   no-op; wasStopped() is true so the spin loop of main.cpp:37-40 ends
 */
#ifndef SYMMICP_PCL_SHIM_H
#define SYMMICP_PCL_SHIM_H
#include <cstdint>
#include <memory>
#include <string>
#include <vector>

namespace pcl {
struct alignas(16) PointXYZ {
    float x = 0.f, y = 0.f, z = 0.f, pad_ = 1.f;
};
struct alignas(16) PointNormal {
    float x = 0.f, y = 0.f, z = 0.f, pad0_ = 1.f;
    float normal_x = 0.f, normal_y = 0.f, normal_z = 0.f, pad1_ = 0.f;
    float curvature = 0.f, pad2_[3] = {0.f, 0.f, 0.f};
};
template <typename PointT>
class PointCloud {
public:
    typedef std::shared_ptr<PointCloud<PointT>> Ptr;
    typedef std::shared_ptr<const PointCloud<PointT>> ConstPtr;
    std::vector<PointT> points;
    uint32_t width = 0, height = 1;
    bool is_dense = true;
    size_t size() const { return points.size(); }
    void push_back(const PointT &p) { points.push_back(p); width = (uint32_t)points.size(); }
};
namespace visualization {
class PCLVisualizer {
public:
    explicit PCLVisualizer(const std::string & = "") {}
    template <typename CloudPtr> bool addPointCloud(const CloudPtr &, const std::string & = "cloud") { return true; }
    bool wasStopped() const { return true; }      // headless: the viewer loop of main.cpp:37-40 exits at once
    void spinOnce(int = 1) {}
};
}  // namespace visualization
}  // namespace pcl
#endif
