// lidar_iris_hip_descriptor.hpp -- header-only adapter that plugs the MI355X LiDAR-Iris engine (scl_iris.h) into the
// reference's descriptor plugin interface, beside scan_context_hip_descriptor.hpp.
//
// Include it AFTER the reference's descriptor.h (it needs `class scan_descriptor`, descriptor.h:21-36, and
// pcl::PointCloud<pcl::PointXYZI>).  Same constructor arguments, defaults and return conventions as
// lidar_iris_descriptor (descriptor.h:462-1302); the DescriptorType switch changes by one line:
//
//   distributedMapping.h:408   scanDescriptor = std::unique_ptr<scan_descriptor>(new lidar_iris_descriptor(80, 360, N_SCAN, 0.4, ...));
//   becomes                    scanDescriptor = std::unique_ptr<scan_descriptor>(new lidar_iris_hip_descriptor(80, 360, N_SCAN, 0.4, ...));
//
// What differs from the reference's class, on purpose (scl_iris.h has the details):
//   * compare() follows the reference (FFT shift estimate, two five-shift Hamming windows, matchNum), but the estimate is OpenCV's
//     algorithms restated, not OpenCV's binaries: an estimate within rounding of a whole column can fall on the other side and move
//     a window by one column (parity unpinned).  shiftSearch = 1 searches every column shift instead -- then the distance is <= the
//     reference's and distThres 0.32 accepts loops the reference would not: an opt-in, re-derive the threshold for it;
//   * saveDescriptorAndKey decodes the wire vector with the reference's own indexing by default (wireDecode = 0,
//     descriptor.h:1035 -- it shears the received image); wireDecode = 1 reads the layout makeAndSave emits.
// Errors are written to stderr and mapped to "no loop" / empty results, as the reference only logs.
// Lifetime: as for scan_context_hip_descriptor -- scan_descriptor has no virtual destructor, call close() before
// dropping the object if the host re-creates descriptors.
#pragma once

#include <cfloat>
#include <cstdint>
#include <cstdio>
#include <utility>
#include <vector>

#include "scl_iris.h"

class lidar_iris_hip_descriptor : public scan_descriptor
{
public:
    // parameter list and defaults of lidar_iris_descriptor's ctor, descriptor.h:473-486
    lidar_iris_hip_descriptor(
        int rows               = 80,
        int cols               = 360,
        int nscan              = 64,
        double distThres       = 0.32,
        int numExcludeRecent   = 30,
        int matchNum           = 2,
        int numCandidates      = 10,
        int nscale             = 4,
        int minWaveLength      = 18,
        float mult             = 1.6f,
        float sigmaOnf         = 0.75f,
        int robotNum           = 1,
        int thisID             = 0,
        int device             = 0,
        int wireDecode         = 0,
        float knnExcludeEps    = FLT_EPSILON,
        int shiftSearch        = 0)
    {
        scl_iris_config cfg;
        scl_iris_default_config(&cfg);
        cfg.rows = rows; cfg.cols = cols; cfg.nscan = nscan; cfg.dist_thres = distThres;
        cfg.num_exclude_recent = numExcludeRecent; cfg.match_num = matchNum; cfg.num_candidates = numCandidates;
        cfg.nscale = nscale; cfg.min_wavelength = minWaveLength; cfg.mult = mult; cfg.sigma_onf = sigmaOnf;
        cfg.robot_num = robotNum; cfg.this_id = thisID; cfg.device = device; cfg.wire_decode = wireDecode;
        cfg.knn_exclude_eps = knnExcludeEps; cfg.shift_search = shiftSearch;
        values_ = rows * cols + rows;
        const int rc = scl_iris_create(&cfg, &iris_);
        if (rc != SCL_OK) {
            std::fprintf(stderr, "[lidar_iris_hip_descriptor] engine creation failed: %s\n", scl_status_string(rc));
            iris_ = nullptr;
        }
    }

    ~lidar_iris_hip_descriptor() { close(); }
    void close()
    {
        if (iris_) scl_iris_destroy(iris_);
        iris_ = nullptr;
    }
    lidar_iris_hip_descriptor(const lidar_iris_hip_descriptor &) = delete;
    lidar_iris_hip_descriptor &operator=(const lidar_iris_hip_descriptor &) = delete;

    // descriptor.h:25 / 1062-1083: image values row-major, then the row key
    std::vector<float> makeAndSaveDescriptorAndKey(const pcl::PointCloud<pcl::PointXYZI> &scan,
                                                   const int8_t robot, const int index) override
    {
        std::vector<float> vT(static_cast<size_t>(values_), 0.0f);
        report(scl_iris_make_and_save(iris_, scan.points.data(), static_cast<int>(scan.points.size()),
                                      static_cast<int>(sizeof(pcl::PointXYZI)), robot, index, vT.data()),
               "makeAndSaveDescriptorAndKey");
        return vT;
    }

    // descriptor.h:27 / 1026-1044 (iris = global_descriptor.values.data(), DM.h:627)
    void saveDescriptorAndKey(const float *iris, const int8_t robot, const int index) override
    {
        report(scl_iris_save_from_wire(iris_, iris, robot, index), "saveDescriptorAndKey");
    }

    // descriptor.h:29 / 1085-1151: {local index of the loop keyframe or -1, column shift}
    std::pair<int, float> detectIntraLoopClosureID(const int curPtr) override
    {
        int loop_id = -1; float bias = 0.0f;
        if (!report(scl_iris_detect_intra(iris_, curPtr, &loop_id, &bias, nullptr), "detectIntraLoopClosureID"))
            return std::pair<int, float>(-1, 0.0f);
        return std::pair<int, float>(loop_id, bias);
    }

    // descriptor.h:31 / 1153-1253: {global key of the loop keyframe or -1, column shift}
    std::pair<int, float> detectInterLoopClosureID(const int curPtr) override
    {
        int loop_id = -1; float bias = 0.0f;
        if (!report(scl_iris_detect_inter(iris_, curPtr, &loop_id, &bias, nullptr), "detectInterLoopClosureID"))
            return std::pair<int, float>(-1, 0.0f);
        return std::pair<int, float>(loop_id, bias);
    }

    // descriptor.h:33 / 1255-1258
    std::pair<int8_t, int> getIndex(const int key) override
    {
        int8_t robot = 0; int index = -1;
        report(scl_iris_get_index(iris_, key, &robot, &index), "getIndex");
        return std::pair<int8_t, int>(robot, index);
    }

    // descriptor.h:35 / 1260-1270
    int getSize(const int idIn = -1) override
    {
        if (!iris_) return 0;
        const int n = scl_iris_get_size_of(iris_, idIn);
        return n < 0 ? 0 : n;
    }

    scl_iris *engine() { return iris_; }

private:
    bool report(int rc, const char *where) const
    {
        if (!iris_) {
            std::fprintf(stderr, "[lidar_iris_hip_descriptor] %s: no engine (creation failed or close() was called)\n", where);
            return false;
        }
        if (rc == SCL_OK) return true;
        std::fprintf(stderr, "[lidar_iris_hip_descriptor] %s: %s (%s)\n", where, scl_status_string(rc), scl_iris_last_error(iris_));
        return false;
    }

    scl_iris *iris_ = nullptr;
    int values_ = 0;
};
