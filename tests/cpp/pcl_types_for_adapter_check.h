// TEST-ONLY stand-ins so that include/scl/scan_context_hip_descriptor.hpp can be type-checked and
// exercised in a container without PCL/ROS.  This is NOT a build of the reference: nothing of the
// reference is compiled here.  It declares (a) a 32-byte point record and a cloud with a `points`
// vector, which is all the adapter touches of pcl::PointCloud<pcl::PointXYZI>, and (b) the plugin
// interface the adapter derives from -- the six pure virtuals of `class scan_descriptor`
// (reference include/descriptor.h:21-36), which is the API surface the adapter must match.
#pragma once
#include <cstdint>
#include <utility>
#include <vector>

namespace pcl {
struct alignas(16) PointXYZI { float x, y, z, pad0, intensity, pad1, pad2, pad3; };
template <class P> struct PointCloud { std::vector<P> points; };
}  // namespace pcl
static_assert(sizeof(pcl::PointXYZI) == 32, "pcl::PointXYZI is a 32-byte record");

class scan_descriptor
{
public:
    virtual std::vector<float> makeAndSaveDescriptorAndKey(const pcl::PointCloud<pcl::PointXYZI> &scan, const int8_t robot, const int index) = 0;
    virtual void saveDescriptorAndKey(const float *descriptorMat, const int8_t robot, const int index) = 0;
    virtual std::pair<int, float> detectIntraLoopClosureID(const int currentPtr) = 0;
    virtual std::pair<int, float> detectInterLoopClosureID(const int currentPtr) = 0;
    virtual std::pair<int8_t, int> getIndex(const int key) = 0;
    virtual int getSize(const int idIn = -1) = 0;
};
